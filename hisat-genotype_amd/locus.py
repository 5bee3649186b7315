"""Packed locus (8a-0): turns the reference's per-gene dicts into the arrays libhgx consumes.

Mirrors the per-locus set-up of typing() (hisatgenotype_typing_core.py:384-401, 476-491, 559-569):
``Genes / Gene_names / Vars / Var_list / Links / refGene_loci`` in, a host ``hgx_locus`` (variant
tables, allele->variant lists, exon representatives, alternatives) and a device ``hgx_index``
(word-major link bit matrix + level masks) out.
"""
import ctypes as C

import numpy as np

from . import capi

import threading as _threading

_INDEX_LOCK = _threading.Lock()
_TYPE = {"insertion": capi.VAR_INSERTION, "single": capi.VAR_SINGLE, "deletion": capi.VAR_DELETION}


class Batch:
    """Host-side result of the front-end for one locus: distinct pieces + per-pair references."""

    def __init__(self, handle):
        self.h = handle
        L = capi.lib()
        np_, nm, npairs, nrefs, nreads = C.c_int32(), C.c_int64(), C.c_int32(), C.c_int64(), C.c_int32()
        capi.check(L.hgx_batch_dims(self.h, C.byref(np_), C.byref(nm), C.byref(npairs), C.byref(nrefs), C.byref(nreads)))
        self.n_pieces, self.n_mask_u32, self.n_pairs = np_.value, nm.value, npairs.value
        self.n_refs, self.n_reads = nrefs.value, nreads.value
        self._arrays = None

    def _load(self):
        """Host copies of the batch arrays, made on first use (the typing path never needs them: libhgx uploads its own)."""
        if self._arrays is None:
            pp, pm, po, pr = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
            capi.check(capi.lib().hgx_batch_arrays(self.h, C.byref(pp), C.byref(pm), C.byref(po), C.byref(pr)))

            def view(p, n, dt):
                if n == 0 or not p.value:
                    return np.zeros(0, dtype=dt)
                buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(p.value)
                return np.frombuffer(buf, dtype=dt, count=n).copy()

            self._arrays = (view(pp, self.n_pieces, capi.PIECE_DTYPE), view(pm, self.n_mask_u32, np.uint32),
                            view(po, self.n_pairs + 1, np.int32), view(pr, self.n_refs, np.uint32))
        return self._arrays

    pieces = property(lambda self: self._load()[0])
    masks = property(lambda self: self._load()[1])
    pair_off = property(lambda self: self._load()[2])
    pair_ref = property(lambda self: self._load()[3])

    def trace_text(self):
        L = capi.lib()
        need = C.c_size_t(0)
        capi.check(L.hgx_batch_trace_text(self.h, None, C.c_size_t(0), C.byref(need)))
        buf = C.create_string_buffer(need.value + 1)
        capi.check(L.hgx_batch_trace_text(self.h, buf, C.c_size_t(need.value + 1), C.byref(need)))
        return buf.value.decode()

    def pileup(self, length):
        nt = np.zeros(length, dtype=np.uint8)
        cnt = np.zeros((length, 6), dtype=np.uint32)
        capi.check(capi.lib().hgx_batch_pileup(self.h, capi.ptr(nt), capi.ptr(cnt)))
        return nt, cnt

    def close(self):
        if self.h:
            capi.lib().hgx_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PackedLocus:
    """One gene's index.  ``names`` = Gene_names[gene] without the backbone (allele index order)."""

    def __init__(self, gene, base_fname, ref_seq, Vars, Var_list, Links, Gene_names, Gene_lengths, exons,
                 Genes_keys=None):
        self.gene, self.base_fname = gene, base_fname
        self.ref_allele = gene + "*BACKBONE"        # RNAME of this locus in a graph alignment (refGenes[gene], core:385)
        self.ref_seq = ref_seq
        names = [n for n in Gene_names if n.find("BACKBONE") == -1 and
                 not (base_fname == "genome" and n.find("GRCh38") != -1)]
        if Genes_keys is not None:
            names = [n for n in names if n in Genes_keys]
        self.names = names
        self.aidx = {n: i for i, n in enumerate(names)}
        A = len(names)
        V = len(Var_list)
        self.var_ids = [vid for _, vid in Var_list]
        self.var_index = {vid: i for i, vid in enumerate(self.var_ids)}
        pos = np.zeros(V, np.int32)
        typ = np.zeros(V, np.uint8)
        ln = np.ones(V, np.int32)
        base = np.zeros(V, np.uint8)
        linked = np.zeros(V, np.uint8)
        ins = []
        off = np.zeros(V + 1, np.int32)
        flat = []
        for i, vid in enumerate(self.var_ids):
            t, p, d = Vars[vid]
            pos[i] = p
            typ[i] = _TYPE[t]
            if t == "deletion":
                ln[i] = int(d)
                ins.append("")
            elif t == "insertion":
                ln[i] = len(d)
                ins.append(d)
            else:
                base[i] = ord(d)
                ins.append("")
            if vid in Links:
                linked[i] = 1
                flat += [self.aidx[a] for a in Links[vid] if a in self.aidx]
            off[i + 1] = len(flat)
        link_order = np.array([self.var_index[v] for v in Links.keys() if v in self.var_index] or [0], np.int32)
        n_link_order = sum(1 for v in Links.keys() if v in self.var_index)
        rank = np.zeros(A, np.int32)
        for r, i in enumerate(sorted(range(A), key=lambda i: names[i])):
            rank[i] = r
        lengths = np.array([Gene_lengths[n] for n in names], np.int32)
        ex = np.array(exons, np.int32).reshape(-1, 2) if len(exons) else np.zeros((0, 2), np.int32)
        self.name_rank, self.allele_len = rank, lengths
        self._keep = dict(pos=pos, typ=typ, ln=ln, base=base, linked=linked, off=off,
                          flat=np.array(flat or [0], np.int32), link_order=link_order, ex=np.ascontiguousarray(ex),
                          rank=rank, lengths=lengths,
                          names_pool=("\0".join(self.var_ids) + "\0").encode(), ins_pool=("\0".join(ins) + "\0").encode(),
                          bb=ref_seq.encode())
        self._n_link_order = n_link_order
        self._finish_init()

    def _finish_init(self):
        """Build the native locus (hgx_locus_create) from the packed arrays in self._keep."""
        k = self._keep
        A, V = len(self.names), len(self.var_ids)
        d = capi.LocusDesc(capi.BASE_KIND.get(self.base_fname, 3), len(self.ref_seq), k["bb"], V, capi.ptr(k["pos"]),
                           capi.ptr(k["typ"]), capi.ptr(k["ln"]), capi.ptr(k["base"]), capi.ptr(k["linked"]), k["names_pool"],
                           k["ins_pool"], A, capi.ptr(k["off"]), capi.ptr(k["flat"]), self._n_link_order,
                           capi.ptr(k["link_order"]), len(k["ex"]), capi.ptr(k["ex"]), capi.ptr(k["lengths"]), capi.ptr(k["rank"]))
        h = C.c_void_p()
        capi.check(capi.lib().hgx_locus_create(C.byref(h), C.byref(d)))
        self.h = h
        na, ap, nv, nw = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        capi.check(capi.lib().hgx_locus_dims(self.h, C.byref(na), C.byref(ap), C.byref(nv), C.byref(nw)))
        self.n_alleles, self.a_pad, self.n_vars, self.n_words = na.value, ap.value, nv.value, nw.value
        self.w64 = self.a_pad // 64
        self._tables = None
        self._index = None

    # ---- packed binary cache -----------------------------------------------------------
    CACHE_VERSION = 1

    def save_cache(self, path):
        """Write the packed form of this locus (everything hgx_locus_create consumes) as one .npz file: loading it skips
        the text parsers and the per-allele / per-variant dict walks of the constructor."""
        k = self._keep
        np.savez_compressed(
            path, version=np.int32(self.CACHE_VERSION), gene=np.array(self.gene), base_fname=np.array(self.base_fname),
            ref_allele=np.array(self.ref_allele), ref_seq=np.frombuffer(k["bb"], np.uint8), names=np.array("\0".join(self.names)),
            names_pool=np.frombuffer(k["names_pool"], np.uint8), ins_pool=np.frombuffer(k["ins_pool"], np.uint8),
            n_link_order=np.int32(self._n_link_order),
            **{n: k[n] for n in ("pos", "typ", "ln", "base", "linked", "off", "flat", "link_order", "ex", "rank", "lengths")})

    @classmethod
    def load_cache(cls, path):
        z = np.load(path, allow_pickle=False)
        if int(z["version"]) != cls.CACHE_VERSION:
            raise ValueError("index cache %s has version %d, expected %d" % (path, int(z["version"]), cls.CACHE_VERSION))
        self = cls.__new__(cls)
        self.gene, self.base_fname = str(z["gene"]), str(z["base_fname"])
        self.ref_allele = str(z["ref_allele"]) if "ref_allele" in z.files else self.gene + "*BACKBONE"
        self.ref_seq = z["ref_seq"].tobytes().decode()
        self.names = str(z["names"]).split("\0") if str(z["names"]) else []
        self.aidx = {n: i for i, n in enumerate(self.names)}
        names_pool = z["names_pool"].tobytes()
        self.var_ids = names_pool.decode().split("\0")[:-1] if len(z["pos"]) else []
        self.var_index = {vid: i for i, vid in enumerate(self.var_ids)}
        self._keep = {n: np.ascontiguousarray(z[n]) for n in
                      ("pos", "typ", "ln", "base", "linked", "off", "flat", "link_order", "ex", "rank", "lengths")}
        self._keep.update(names_pool=names_pool, ins_pool=z["ins_pool"].tobytes(), bb=self.ref_seq.encode())
        self.name_rank, self.allele_len = self._keep["rank"], self._keep["lengths"]
        self._n_link_order = int(z["n_link_order"])
        self._finish_init()
        return self

    # ---- constructors -------------------------------------------------------------------------
    @classmethod
    def from_reference_dicts(cls, gene, base_fname, refGenes, Genes, Gene_names, Gene_lengths, refGene_loci, Vars,
                             Var_list, Links):
        ref_allele = refGenes[gene]
        self = cls(gene, base_fname, Genes[gene][ref_allele], Vars.get(gene, {}), Var_list.get(gene, []), Links,
                   Gene_names[gene], Gene_lengths[gene], refGene_loci[gene][-2], set(Genes[gene].keys()))
        self.ref_allele = ref_allele
        return self

    @classmethod
    def cached_from_reference_dicts(cls, gene, base_fname, refGenes, Genes, Gene_names, Gene_lengths, refGene_loci, Vars,
                                    Var_list, Links):
        """from_reference_dicts through the in-process locus cache (LocusCache below): typing() is called once per sample with
        the SAME index dicts (/root/reference/hisatgenotype:613-665 forks them into every worker), and packing a 7 000-allele
        locus from them costs 40-90 ms of Python -- many times the GPU's share of a sample.  A cached locus keeps its device
        index and pattern tables; it must not be close()d by the caller."""
        return LOCUS_CACHE.get(gene, base_fname, refGenes, Genes, Gene_names, Gene_lengths, refGene_loci, Vars, Var_list, Links)

    @classmethod
    def from_synth(cls, locus):
        d = locus.reference_dicts()
        return cls.from_reference_dicts(locus.gene, locus.base_fname, d["refGenes"], d["Genes"], d["Gene_names"],
                                        d["Gene_lengths"], d["refGene_loci"], d["Vars"], d["Var_list"], d["Links"])

    # ---- derived tables -------------------------------------------------------------------------
    def tables(self):
        if self._tables is None:
            bits = np.zeros((self.n_words, self.a_pad), np.uint32)
            em = np.zeros(self.w64, np.uint64)
            gm = np.zeros(self.w64, np.uint64)
            rep = np.zeros(self.n_alleles, np.int32)
            capi.check(capi.lib().hgx_locus_tables(self.h, capi.ptr(bits), capi.ptr(em), capi.ptr(gm), capi.ptr(rep)))
            self._tables = dict(link_bits=bits, exon_mask=em, gene_mask=gm, rep_of=rep)
        return self._tables

    def rep_groups(self):
        """allele_rep_groups (core:86-115) as {rep index: [member indices in first-met order]}."""
        if getattr(self, "_rep_groups", None) is not None:
            return self._rep_groups
        rep = self.tables()["rep_of"]
        groups = {}
        # members are listed in the order get_rep_alleles met them = order of first appearance scanning Links;
        # the hand-off only uses the groups as sets (core:1745-1749), so index order is sufficient
        for a, r in enumerate(rep):
            if r >= 0:
                groups.setdefault(int(r), []).append(a)
        self._rep_groups = groups
        return groups

    def index(self):
        """Device-resident hgx_index (created on first use on the current device; a cached locus may be asked by several threads at once)."""
        if self._index is None:
            with _INDEX_LOCK:
                if self._index is None:
                    h = C.c_void_p()
                    capi.check(capi.lib().hgx_index_from_locus(C.byref(h), self.h))
                    self._index = h
        return self._index

    def alternatives_text(self):
        need = C.c_size_t(0)
        capi.check(capi.lib().hgx_locus_alternatives_text(self.h, None, C.c_size_t(0), C.byref(need)))
        buf = C.create_string_buffer(need.value + 1)
        capi.check(capi.lib().hgx_locus_alternatives_text(self.h, buf, C.c_size_t(need.value + 1), C.byref(need)))
        return buf.value.decode()

    # ---- front-end ---------------------------------------------------------------------------------
    def batch_from_haplotypes(self, pair_off, level, left, right, id_off, ids):
        pair_off = np.ascontiguousarray(pair_off, np.int32)
        level = np.ascontiguousarray(level, np.uint8)
        left = np.ascontiguousarray(left, np.int32)
        right = np.ascontiguousarray(right, np.int32)
        id_off = np.ascontiguousarray(id_off, np.int32)
        ids = np.ascontiguousarray(ids if len(ids) else [0], np.int32)
        h = C.c_void_p()
        capi.check(capi.lib().hgx_batch_from_haplotypes(C.byref(h), self.h, C.c_int32(len(pair_off) - 1), capi.ptr(pair_off),
                                                        capi.ptr(level), capi.ptr(left), capi.ptr(right), capi.ptr(id_off),
                                                        capi.ptr(ids)))
        return Batch(h)

    def parse_sam(self, sam_text, num_editdist=2, error_correction=True, allow_discordant=False, simulation=False,
                  base_locus=0, keep_trace=False, n_threads=0, pileup_exchange=None, interdist_exchange=None, last_shard=True):
        """`pileup_exchange(counts)`: intra-locus read sharding (dist.type_locus_sharded) -- called once with this shard's
        pileup counts (numpy uint32 [L*6], a view of the front-end's table) and must turn them into the sum over all shards
        in place (an all-reduce).  `interdist_exchange(hist)`: the same for the inter-distance histogram of a sharded CODIS
        D18S51 sample (int64 [HGX_INTERDIST_BINS]); `last_shard`: this shard ends with the stream's last pair -- the only one
        choose_pairs is applied to (typing_core.py:1547-1552)."""
        data = sam_text if isinstance(sam_text, (bytes, bytearray)) else sam_text.encode()
        o = capi.ParseOpts(num_editdist, int(error_correction), int(allow_discordant), int(simulation), base_locus,
                           int(keep_trace), int(self.base_fname == "codis" and self.gene == "D18S51" and last_shard), int(n_threads))
        cb, failure = None, []
        if pileup_exchange is not None:
            def _cb(_ctx, ptr, n):
                try:
                    pileup_exchange(np.ctypeslib.as_array(ptr, shape=(n,)))
                    return 0
                except BaseException as e:          # never let an exception cross the C frame
                    failure.append(e)
                    return 1
            cb = capi.PILEUP_EXCHANGE(_cb)
            o.pileup_exchange = C.cast(cb, C.c_void_p)
        cb2 = None
        if interdist_exchange is not None:
            def _cb2(_ctx, ptr, n):
                try:
                    interdist_exchange(np.ctypeslib.as_array(ptr, shape=(n,)))
                    return 0
                except BaseException as e:
                    failure.append(e)
                    return 1
            cb2 = capi.INTERDIST_EXCHANGE(_cb2)
            o.interdist_exchange = C.cast(cb2, C.c_void_p)
        h = C.c_void_p()
        rc = capi.lib().hgx_parse_sam(C.byref(h), self.h, data, C.c_size_t(len(data)), C.byref(o))
        if failure:
            raise failure[0]
        capi.check(rc)
        return Batch(h)

    def parse_alignment_file(self, path, regions=None, num_editdist=2, error_correction=True, allow_discordant=False,
                             simulation=False, base_locus=0, n_threads=0):
        """Front-end straight from a SAM / BAM file (hgx_parse_alignment_file): read, inflate, decode, region filter, name
        grouping and piece extraction without the text ever passing through Python.  `regions`: samtools region strings
        ("name" or "name:left-right", 1-based; a list or one newline-separated string), None = every record."""
        if regions is not None and not isinstance(regions, (str, bytes)):
            regions = "\n".join(regions)
        if isinstance(regions, str):
            regions = regions.encode()
        o = capi.ParseOpts(num_editdist, int(error_correction), int(allow_discordant), int(simulation), base_locus, 0,
                           int(self.base_fname == "codis" and self.gene == "D18S51"), int(n_threads))
        h = C.c_void_p()
        capi.check(capi.lib().hgx_parse_alignment_file(C.byref(h), self.h, path.encode(), regions or None, C.byref(o)))
        return Batch(h)

    @staticmethod
    def _install_exchanges(o, pileup_exchange, pileup_exchange_dev, interdist_exchange=None):
        """Hook the python callbacks of a sharded parse into an hgx_parse_opts; returns (objects to keep alive, failure list)."""
        keep, failure = [], []
        if pileup_exchange is not None:
            def _cb(_ctx, ptr, n):
                try:
                    pileup_exchange(np.ctypeslib.as_array(ptr, shape=(n,)))
                    return 0
                except BaseException as e:          # never let an exception cross the C frame
                    failure.append(e)
                    return 1
            cb = capi.PILEUP_EXCHANGE(_cb)
            o.pileup_exchange = C.cast(cb, C.c_void_p)
            keep.append(cb)
        if pileup_exchange_dev is not None:
            def _cbd(_ctx, dptr, n, stream):
                try:
                    pileup_exchange_dev(int(dptr), int(n), C.c_void_p(stream) if stream else None)   # (a bare int would travel as a C int)
                    return 0
                except BaseException as e:
                    failure.append(e)
                    return 1
            cbd = capi.PILEUP_EXCHANGE_DEV(_cbd)
            o.pileup_exchange_dev = C.cast(cbd, C.c_void_p)
            keep.append(cbd)
        if interdist_exchange is not None:
            def _cb2(_ctx, ptr, n):
                try:
                    interdist_exchange(np.ctypeslib.as_array(ptr, shape=(n,)))
                    return 0
                except BaseException as e:
                    failure.append(e)
                    return 1
            cb2 = capi.INTERDIST_EXCHANGE(_cb2)
            o.interdist_exchange = C.cast(cb2, C.c_void_p)
            keep.append(cb2)
        return keep, failure

    def parse_sam_dev(self, sam_text, num_editdist=2, error_correction=True, allow_discordant=False, simulation=False, base_locus=0,
                      n_threads=0, stream=None, keep_trace=False, pileup_exchange=None, pileup_exchange_dev=None, interdist_exchange=None,
                      last_shard=True):
        """SAM text -> piece batch in HBM through the DEVICE front end (hgx_parse_sam_dev): record fields, filters, key grouping,
        pileup, decode, piece table and pair protocol run as kernels; inputs the kernels decline are finished by the host stages
        inside the same call.  Returns an engine.DeviceBatch (engine.front_last() tells which route ran).
        A shard of a sharded locus (dist.type_locus_sharded): `pileup_exchange(counts)` = the host form (numpy uint32 [L*6], summed
        in place over the shards), `pileup_exchange_dev(device pointer, n, stream)` = the device form (n = L*6 counters + one
        spare element, summed in place in HBM); `interdist_exchange` / `last_shard`: as in parse_sam (CODIS D18S51: the kernels count
        the inner distances, the histogram is exchanged after the pileup); see hgx_parse_opts in include/hgx.h."""
        from . import engine
        data = sam_text if isinstance(sam_text, (bytes, bytearray)) else sam_text.encode()
        o = capi.ParseOpts(num_editdist, int(error_correction), int(allow_discordant), int(simulation), base_locus, int(keep_trace),
                           int(self.base_fname == "codis" and self.gene == "D18S51" and last_shard), int(n_threads))
        keep, failure = self._install_exchanges(o, pileup_exchange, pileup_exchange_dev, interdist_exchange)
        h = C.c_void_p()
        rc = capi.lib().hgx_parse_sam_dev(C.byref(h), self.h, data, C.c_size_t(len(data)), C.byref(o), stream)
        del keep
        if failure:
            raise failure[0]
        capi.check(rc)
        return engine.DeviceBatch.from_handle(h)

    def parse_alignment_file_dev(self, path, regions=None, num_editdist=2, error_correction=True, allow_discordant=False,
                                 simulation=False, base_locus=0, n_threads=0, stream=None, keep_trace=False, pileup_exchange=None,
                                 pileup_exchange_dev=None, interdist_exchange=None, last_shard=True):
        """parse_alignment_file through the device front end (hgx_parse_alignment_file_dev) -> engine.DeviceBatch."""
        from . import engine
        if regions is not None and not isinstance(regions, (str, bytes)):
            regions = "\n".join(regions)
        if isinstance(regions, str):
            regions = regions.encode()
        o = capi.ParseOpts(num_editdist, int(error_correction), int(allow_discordant), int(simulation), base_locus, int(keep_trace),
                           int(self.base_fname == "codis" and self.gene == "D18S51" and last_shard), int(n_threads))
        keep, failure = self._install_exchanges(o, pileup_exchange, pileup_exchange_dev, interdist_exchange)
        h = C.c_void_p()
        rc = capi.lib().hgx_parse_alignment_file_dev(C.byref(h), self.h, path.encode(), regions or None, C.byref(o), stream)
        del keep
        if failure:
            raise failure[0]
        capi.check(rc)
        return engine.DeviceBatch.from_handle(h)

    def close(self):
        if self._index is not None:
            capi.lib().hgx_index_destroy(self._index)
            self._index = None
        if self.h:
            capi.lib().hgx_locus_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LocusCache:
    """Packed loci (host tables + device index) of this process, kept between typing() calls.

    Two keys.  The IDENTITY key is the tuple of id()s of the per-gene dict objects a call passes (the entry holds references
    to them, so an id cannot be recycled while the entry lives) together with a cheap fingerprint -- the container sizes and
    the total number of (variant, allele) links -- that catches an index edited in place between calls when the edit adds or
    removes anything (an edit that keeps every size, e.g. a variant moved to another position in the SAME dict objects, is not
    seen: call LOCUS_CACHE.clear() after such surgery; the reference itself never edits these dicts, typing_core.py:384-401 copies).  A miss there falls
    back to the CONTENT key: a digest of everything the PackedLocus constructor reads (names in order, backbone, variants in
    Var_list order, the links of those variants in Links' key order, lengths, exons), so that dicts re-read from the same index
    files (driver.genotyping_locus does that once per call) still meet their packed form.  Least recently used entries are
    dropped -- and closed -- beyond `limit`."""

    per_device = True

    def __init__(self, limit=16):
        import threading
        self.limit = limit
        self._lock = threading.Lock()
        self._by_id, self._by_content = {}, {}       # key -> entry; entry = [locus, id key, content key, kept references, tick]
        self._tick = 0
        self.hits_identity = self.hits_content = self.misses = 0

    @staticmethod
    def _parts(gene, refGenes, Genes, Gene_names, Gene_lengths, refGene_loci, Vars, Var_list):
        return (Vars.get(gene, None), Var_list.get(gene, None), Genes[gene], Gene_names[gene], Gene_lengths[gene], refGene_loci[gene])

    @staticmethod
    def _fingerprint(parts, Links):
        v, vl, g, gn, gl, rl = parts
        return (len(v) if v is not None else -1, len(vl) if vl is not None else -1, len(g), len(gn), len(gl), len(Links),
                sum(map(len, Links.values())))

    @staticmethod
    def _digest(gene, base_fname, ref_allele, parts, Links):
        import hashlib
        v, vl, g, gn, gl, rl = parts
        v, vl = v or {}, vl or []
        h = hashlib.blake2b(digest_size=20)
        sep = "\x1f"
        h.update(("%s\0%s\0%s\0%r\0" % (gene, base_fname, ref_allele, rl[-2])).encode())
        h.update(g[ref_allele].encode())
        h.update(sep.join(gn).encode())
        h.update(sep.join(sorted(g.keys())).encode())
        h.update(repr([gl[n] for n in gn if n in gl]).encode())
        h.update(repr(vl).encode())
        h.update(repr([v[vid] for _, vid in vl]).encode())
        mine = set(vid for _, vid in vl)
        for vid, alleles in Links.items():           # in Links' own key order (link_order of the packed form)
            if vid in mine:
                h.update(("\x1e" + vid + sep).encode())
                h.update(sep.join(alleles).encode())
        return h.digest()

    def get(self, gene, base_fname, refGenes, Genes, Gene_names, Gene_lengths, refGene_loci, Vars, Var_list, Links):
        parts = self._parts(gene, refGenes, Genes, Gene_names, Gene_lengths, refGene_loci, Vars, Var_list)
        ref_allele = refGenes[gene]
        dev = capi.current_device() if LocusCache.per_device else -1        # (the device index of a cached locus lives on ONE GPU)
        idk = (gene, base_fname, ref_allele, dev, id(Links)) + tuple(id(x) for x in parts) + self._fingerprint(parts, Links)
        with self._lock:
            self._tick += 1
            e = self._by_id.get(idk)
            if e is not None:
                e[4] = self._tick
                self.hits_identity += 1
                return e[0]
        ck = (dev, self._digest(gene, base_fname, ref_allele, parts, Links))
        with self._lock:
            e = self._by_content.get(ck)
            if e is not None:                        # the same index in other dict objects: re-key the entry to them
                self._by_id.pop(e[1], None)
                e[1], e[3], e[4] = idk, (parts, Links), self._tick
                self._by_id[idk] = e
                self.hits_content += 1
                return e[0]
        pl = PackedLocus.from_reference_dicts(gene, base_fname, refGenes, Genes, Gene_names, Gene_lengths, refGene_loci, Vars,
                                              Var_list, Links)
        pl.cached = True
        with self._lock:
            self.misses += 1
            e = [pl, idk, ck, (parts, Links), self._tick]
            self._by_id[idk] = e
            self._by_content[ck] = e
            while len(self._by_content) > max(self.limit, 1):
                old = min(self._by_content.values(), key=lambda x: x[4])
                self._by_id.pop(old[1], None)
                self._by_content.pop(old[2], None)
                old[0].cached = False                # (a caller may still hold it: it is closed when the last reference goes)
        return pl

    def clear(self):
        with self._lock:
            for e in self._by_content.values():
                e[0].cached = False
                e[0].close()
            self._by_id.clear()
            self._by_content.clear()


LOCUS_CACHE = LocusCache()
