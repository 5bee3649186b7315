"""Device side of the hot path: piece scoring, class dedup, allele counts and EM on the MI355X.

Thin, explicit wrappers over the C-ABI (include/hgx.h); every call runs hand-written HIP kernels in
libhgx.so.  There is no CPU fallback here -- a missing library or a failing call raises.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import DevArray


class DeviceBatch:
    """A front-end batch resident in HBM (distinct pieces, masks, per-pair refs): hgx_dbatch."""

    def __init__(self, batch, stream=None):
        self.h = C.c_void_p()
        capi.check(capi.lib().hgx_dbatch_create(C.byref(self.h), batch.h, stream))
        self.n_pieces, self.n_pairs, self.n_refs, self.n_reads = batch.n_pieces, batch.n_pairs, batch.n_refs, batch.n_reads
        spw, ngr = C.c_int64(), C.c_int64()
        capi.check(capi.lib().hgx_dbatch_dims(self.h, None, None, None, None, C.byref(spw), C.byref(ngr)))
        # algorithmic byte model inputs
        self.sum_piece_words, self.n_gene_refs = spw.value, ngr.value
        self._batch = batch
        self._dev = None

    @classmethod
    def from_handle(cls, h):
        """Wrap an hgx_dbatch made by the library (the device front end: hgx_parse_sam_dev / hgx_parse_alignment_file_dev)."""
        self = cls.__new__(cls)
        self.h = h
        npc, npr, nrf, nrd, spw, ngr = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int32(), C.c_int64(), C.c_int64()
        capi.check(capi.lib().hgx_dbatch_dims(self.h, C.byref(npc), C.byref(npr), C.byref(nrf), C.byref(nrd), C.byref(spw), C.byref(ngr)))
        self.n_pieces, self.n_pairs, self.n_refs, self.n_reads = npc.value, npr.value, nrf.value, nrd.value
        self.sum_piece_words, self.n_gene_refs = spw.value, ngr.value
        self._batch = None
        self._dev = None
        return self

    def to_host(self):
        """The batch as a host Batch (hgx_dbatch_to_host): pieces, masks, refs and -- if the kernels made them -- the pileup tables."""
        from . import locus
        h = C.c_void_p()
        capi.check(capi.lib().hgx_dbatch_to_host(self.h, C.byref(h)))
        return locus.Batch(h)

    def _arrays(self):
        """Separate device copies for the callers that drive the stages one by one (tests, tools)."""
        if self._batch is None:
            self._batch = self.to_host()
        if self._dev is None:
            b = self._batch
            self._dev = (DevArray.from_host(b.pieces if b.n_pieces else np.zeros(1, capi.PIECE_DTYPE)),
                         DevArray.from_host(b.masks if b.n_mask_u32 else np.zeros(2, np.uint32)),
                         DevArray.from_host(b.pair_off),
                         DevArray.from_host(b.pair_ref if b.n_refs else np.zeros(1, np.uint32)))
        return self._dev

    pieces = property(lambda self: self._arrays()[0])
    masks = property(lambda self: self._arrays()[1])
    pair_off = property(lambda self: self._arrays()[2])
    pair_ref = property(lambda self: self._arrays()[3])

    def close(self):
        if self.h:
            capi.lib().hgx_dbatch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Alignment:
    """An alignment file read ONCE, its bytes resident in HBM (hgx_alignment_open): the SAM text, or the BAM stream inflated on the
    device.  parse_dev(locus, regions) is the per-locus rest of the device front end over those bytes -- the loci of a panel may
    call it side by side from threads with streams of their own.  `resident` False (a file below the front end's size gate, or one
    the reader could not leave to the device): every parse_dev goes through hgx_parse_alignment_file_dev on the path instead."""

    def __init__(self, path, n_threads=0, stream=None):
        self.path = path
        self.h = C.c_void_p()
        capi.check(capi.lib().hgx_alignment_open(C.byref(self.h), path.encode(), C.c_int32(n_threads), stream))
        r, t, nb, up = C.c_int32(), C.c_int32(), C.c_size_t(), C.c_longlong()
        capi.check(capi.lib().hgx_alignment_dims(self.h, C.byref(r), C.byref(t), C.byref(nb), C.byref(up)))
        self.resident, self.is_text, self.stream_bytes, self.bytes_to_device = bool(r.value), bool(t.value), nb.value, up.value

    def parse_dev(self, locus, regions=None, num_editdist=2, error_correction=True, allow_discordant=False, simulation=False, base_locus=0,
                  n_threads=0, stream=None):
        if regions is not None and not isinstance(regions, (str, bytes)):
            regions = "\n".join(regions)
        if isinstance(regions, str):
            regions = regions.encode()
        o = capi.ParseOpts(num_editdist, int(error_correction), int(allow_discordant), int(simulation), base_locus, 0,
                           int(locus.base_fname == "codis" and locus.gene == "D18S51"), int(n_threads))
        h = C.c_void_p()
        capi.check(capi.lib().hgx_alignment_parse_dev(C.byref(h), self.h, locus.h, regions or None, C.byref(o), stream))
        return DeviceBatch.from_handle(h)

    def close(self):
        if self.h:
            capi.lib().hgx_alignment_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def stream_sets_info():
    """Stream placement of the current device (hgx_stream_sets_info): {'sets', 'queue_classes', 'probes', 'probe_ms', 'free_sets': [(em
    class, gene class)]}."""
    n, nc, npb, ms = C.c_int32(), C.c_int32(), C.c_int32(), C.c_double()
    cls = np.full(64, -9, np.int32)
    capi.check(capi.lib().hgx_stream_sets_info(C.byref(n), C.byref(nc), C.byref(npb), C.byref(ms), capi.ptr(cls), C.c_int32(64)))
    pairs = [(int(cls[2 * i]), int(cls[2 * i + 1])) for i in range(32) if cls[2 * i] != -9]
    return {"sets": n.value, "queue_classes": nc.value, "probes": npb.value, "probe_ms": round(ms.value, 2), "free_sets": pairs}


def front_last():
    """(route, decline code) of the calling thread's last hgx_parse_*_dev / hgx_type_file call: route 2 = the device took the
    records themselves (fields, filters, key grouping as kernels), 1 = the host made the key table and the device the rest,
    0 = the host stages finished the job (decline code says why)."""
    ran, code = C.c_int32(0), C.c_int32(0)
    capi.check(capi.lib().hgx_front_last(C.byref(ran), C.byref(code), None))
    return ran.value, code.value


def front_last_parts():
    """2 / 3 = that call took a big SAM text in that many parts (all but the last beside the upload's tail), 0 = every kernel behind the last byte."""
    n = C.c_int32(0)
    capi.check(capi.lib().hgx_front_last_parts(C.byref(n)))
    return n.value


def front_last_bytes():
    """Bytes that call sent to the device (SAM text / inflated BAM stream + line table, or the key table of the key route)."""
    n = C.c_int64(0)
    capi.check(capi.lib().hgx_front_last(None, None, C.byref(n)))
    return n.value


def em_last_order(n_alleles):
    """Insertion order of the dict the last em() / em_ordered() of this thread returned (position per allele, -1 outside), or None
    when that EM did not run on the reference-order kernel (hgx_em_last_order)."""
    order = np.zeros(max(n_alleles, 1), np.int32)
    ok = capi.lib().hgx_em_last_order(capi.ptr(order), C.c_int32(n_alleles))
    return order[:n_alleles] if ok else None


def emx_set_timing(on):
    capi.check(capi.lib().hgx_emx_set_timing(C.c_int(1 if on else 0)))


def emx_get_timing(fast):
    """(kernel ms, launches, jobs, applications of the EM map, algorithmic bytes) of the k_emx launches timed so far."""
    ms, nl, nj, na, nb = C.c_double(0), C.c_longlong(0), C.c_longlong(0), C.c_longlong(0), C.c_longlong(0)
    capi.check(capi.lib().hgx_emx_get_timing(C.c_int(1 if fast else 0), C.byref(ms), C.byref(nl), C.byref(nj), C.byref(na), C.byref(nb)))
    return ms.value, nl.value, nj.value, na.value, nb.value


def emx_cluster_stats():
    """(problems launched on a cluster of workgroups, those whose cluster gave up and were re-run on one workgroup) since the library was loaded."""
    a, b = C.c_longlong(0), C.c_longlong(0)
    capi.check(capi.lib().hgx_emx_cluster_stats(C.byref(a), C.byref(b)))
    return a.value, b.value


def em_tie_reruns():
    """EM calls whose table-lookup result held a near-tie it could not order (two alleles of different class membership closer than 1e-8
    relative) and was recomputed in the reference's own order of operations, since the library was loaded (hgx_em_tie_reruns)."""
    f = capi.lib().hgx_em_tie_reruns
    f.restype = C.c_longlong
    return int(f())


def em_set_fast(on):
    """Arithmetic of Classes.em / em_ordered on this thread for problems the one-workgroup kernel takes: False = the reference's own
    order of operations (default, bit-identical), True = table lookups (~5x faster, within rounding).  Returns the old setting."""
    mode = -1 if (on is not True and on is not False and int(on) < 0) else (1 if on else 0)    # -1: the reference's order at EVERY size
    old = capi.lib().hgx_em_set_fast(C.c_int32(mode))
    return -1 if old < 0 else bool(old)


def _parse_opts(locus, num_editdist=2, error_correction=True, allow_discordant=False, simulation=False, base_locus=0, n_threads=0):
    return capi.ParseOpts(num_editdist, int(error_correction), int(allow_discordant), int(simulation), base_locus, 0,
                          int(locus.base_fname == "codis" and locus.gene == "D18S51"), int(n_threads))


class ManyBatch:
    """The piece batches of many tasks of ONE locus merged and resident in HBM (hgx_many): what hgx_type_many types at once."""

    def __init__(self, locus, batches, stream=None):
        self.h = C.c_void_p()
        arr = (C.c_void_p * max(len(batches), 1))(*[b.h for b in batches])
        capi.check(capi.lib().hgx_many_create(C.byref(self.h), locus.h, arr, C.c_int32(len(batches)), stream))
        self._dims()

    def _dims(self):
        nt, npc, npr, nrf, nrd = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64()
        capi.check(capi.lib().hgx_many_dims(self.h, C.byref(nt), C.byref(npc), C.byref(npr), C.byref(nrf), C.byref(nrd)))
        self.n_tasks, self.n_pieces, self.n_pairs, self.n_refs, self.n_reads = nt.value, npc.value, npr.value, nrf.value, nrd.value
        n = self.n_tasks
        base, reads = (C.c_int32 * (n + 1))(), (C.c_int32 * max(n, 1))()
        pieces, refs = (C.c_int32 * max(n, 1))(), (C.c_int64 * max(n, 1))()
        capi.check(capi.lib().hgx_many_tasks(self.h, None, base, reads, pieces, refs))
        self.pair_base = list(base)
        self.task_reads = list(reads)[:n]
        self.task_pairs = [base[t + 1] - base[t] for t in range(n)]
        self.task_pieces = list(pieces)[:n]
        self.task_refs = list(refs)[:n]

    @classmethod
    def from_files(cls, locus, paths, regions=None, stream=None, **opts):
        """hgx_many_create_files: the samples' alignment files (SAM text or BAM) of one locus -> merged batch, in one pass of the
        device front end (or, where it declines, through the host front end per task: `engine.front_last()` says which)."""
        self = cls.__new__(cls)
        self.h = C.c_void_p()
        n = len(paths)
        p_arr = (C.c_char_p * max(n, 1))(*[str(p).encode() for p in paths])
        r_arr = None
        if regions is not None:
            r_arr = (C.c_char_p * max(n, 1))(*[(r.encode() if r else None) for r in regions])
        o = _parse_opts(locus, **opts)
        capi.check(capi.lib().hgx_many_create_files(C.byref(self.h), locus.h, p_arr, r_arr, C.c_int32(n), C.byref(o), stream))
        self._dims()
        return self

    @classmethod
    def from_sams(cls, locus, sams, stream=None, **opts):
        """hgx_many_create_sams: name-grouped SAM texts (bytes) in memory, one per task."""
        self = cls.__new__(cls)
        self.h = C.c_void_p()
        n = len(sams)
        keep = [s if isinstance(s, bytes) else s.encode() for s in sams]
        s_arr = (C.c_char_p * max(n, 1))(*keep)
        n_arr = (C.c_size_t * max(n, 1))(*[len(s) for s in keep])
        o = _parse_opts(locus, **opts)
        capi.check(capi.lib().hgx_many_create_sams(C.byref(self.h), locus.h, s_arr, n_arr, C.c_int32(n), C.byref(o), stream))
        self._dims()
        return self

    @classmethod
    def from_dbatch(cls, locus, dbatch, stream=None):
        """hgx_many_from_dbatch: ONE task from a DeviceBatch the front end left in HBM (the batch moves into the ManyBatch: `dbatch` is
        empty afterwards).  The loci of one sample go into typing.type_many_loci this way."""
        self = cls.__new__(cls)
        self.h = C.c_void_p()
        capi.check(capi.lib().hgx_many_from_dbatch(C.byref(self.h), locus.h, dbatch.h, stream))
        dbatch.h = None
        self._dims()
        return self

    def merged(self):
        """The merged device batch as a host Batch-like object (tests, tools): pieces, masks, pair_off, pair_ref."""
        db = C.c_void_p()
        capi.check(capi.lib().hgx_many_tasks(self.h, C.byref(db), None, None, None, None))
        from . import locus
        h = C.c_void_p()
        capi.check(capi.lib().hgx_dbatch_to_host(db, C.byref(h)))
        return locus.Batch(h)

    def close(self):
        if self.h:
            capi.lib().hgx_many_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Gate:
    """hgx_gate: shared by the samples in flight on one GPU; hgx_type_* holds it from entry until the sample's exon-level
    classes exist, so that one bandwidth-bound front runs at a time, beside the other samples' EM phases."""

    def __init__(self):
        self.h = C.c_void_p()
        capi.check(capi.lib().hgx_gate_create(C.byref(self.h)))

    def __del__(self):
        try:
            capi.lib().hgx_gate_destroy(self.h)
        except Exception:
            pass


class ScoreBuffers:
    """Output/scratch buffers of one scoring pass, reusable across steps."""

    def __init__(self, locus, dbatch, exon=True):
        w64 = locus.w64
        n = max(dbatch.n_pairs, 1)
        self.compat = DevArray((max(dbatch.n_pieces, 1), w64), np.uint64)
        self.gene_bits = DevArray((n, w64), np.uint64)
        self.gene_hash = DevArray(n, np.uint64)
        self.exon_bits = DevArray((n, w64), np.uint64) if exon else None
        self.exon_hash = DevArray(n, np.uint64) if exon else None


def score_pairs(locus, dbatch, bufs, stream=None):
    """8a-5/6: pieces x alleles -> per-pair class rows (+ hashes) for the exon and gene levels."""
    L = capi.lib()
    capi.check(L.hgx_score_pairs(locus.index(), capi.ptr(dbatch.pieces), capi.ptr(dbatch.masks), C.c_int32(dbatch.n_pieces),
                                 capi.ptr(dbatch.pair_off), capi.ptr(dbatch.pair_ref), C.c_int32(dbatch.n_pairs),
                                 capi.ptr(bufs.compat), capi.ptr(bufs.exon_bits), capi.ptr(bufs.gene_bits),
                                 capi.ptr(bufs.exon_hash), capi.ptr(bufs.gene_hash), stream))


def piece_compat(locus, dbatch, bufs, stream=None):
    """8a-5 stage 1 alone: the allele bitset of every distinct piece."""
    capi.check(capi.lib().hgx_piece_compat(locus.index(), capi.ptr(dbatch.pieces), capi.ptr(dbatch.masks),
                                           C.c_int32(dbatch.n_pieces), capi.ptr(bufs.compat), stream))


def pair_classes(locus, dbatch, bufs, stream=None, exon=True, gene=True):
    """8a-5/6 stage 2 alone (after piece_compat): per-pair class rows + hashes of the chosen levels."""
    e, g = exon and bufs.exon_bits is not None, gene
    capi.check(capi.lib().hgx_pair_classes(locus.index(), capi.ptr(bufs.compat), capi.ptr(dbatch.pair_off),
                                           capi.ptr(dbatch.pair_ref), C.c_int32(dbatch.n_pairs),
                                           capi.ptr(bufs.exon_bits) if e else None, capi.ptr(bufs.gene_bits) if g else None,
                                           capi.ptr(bufs.exon_hash) if e else None, capi.ptr(bufs.gene_hash) if g else None, stream))


class Groups:
    """Pairs grouped by their list of piece refs at one level (hgx_group_pairs): needs no piece bitsets, so it can be queued on
    its own stream beside piece_compat.  Keep it alive until the classes made from it have been consumed."""

    def __init__(self, dbatch, level, stream=None):
        self.h = C.c_void_p()
        self.level = level
        capi.check(capi.lib().hgx_group_pairs(C.byref(self.h), capi.ptr(dbatch.pair_off), capi.ptr(dbatch.pair_ref),
                                              C.c_int32(dbatch.n_pairs), C.c_int32(level), stream))      # queued, not waited for

    @property
    def n_groups(self):
        n = C.c_int64()
        capi.check(capi.lib().hgx_groups_dims(self.h, C.byref(n), None))
        return n.value

    def close(self):
        if self.h:
            capi.lib().hgx_groups_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class Classes:
    """Distinct compatibility classes in first-seen order (Gene_cmpt / Gene_exons_cmpt as a bit matrix)."""

    def __init__(self, handle, owned=True):
        self.h = handle
        self.owned = owned           # False: the handle belongs to an hgx_typing result
        n, ap = C.c_int32(), C.c_int32()
        capi.check(capi.lib().hgx_classes_dims(self.h, C.byref(n), C.byref(ap)))
        self.n_classes, self.a_pad = n.value, ap.value
        self.w64 = self.a_pad // 64

    @staticmethod
    def dedup(rows, n_rows, a_pad, hashes=None, weights=None, and_mask=None, stream=None):
        """8a-7: group rows by content (optionally AND and_mask first; empty rows dropped)."""
        h = C.c_void_p()
        capi.check(capi.lib().hgx_dedup_classes(C.byref(h), capi.ptr(rows), capi.ptr(hashes), capi.ptr(weights),
                                                 C.c_int64(n_rows), C.c_int32(a_pad), capi.ptr(and_mask), stream))
        return Classes(h)

    def set_allele_rank(self, name_rank):
        """Name order of the alleles (rank per allele index): lets em_masked follow the reference's summation order exactly."""
        r = np.ascontiguousarray(name_rank, dtype=np.int32)
        capi.check(capi.lib().hgx_classes_set_allele_rank(self.h, capi.ptr(r), C.c_int32(len(r))))

    @staticmethod
    def of_level(locus, dbatch, bufs, level, stream=None, groups=None):
        """Classes of one level (0 exon, 1 gene) straight from the piece refs (after piece_compat): hgx_level_classes, or its
        second step when the pairs were grouped beforehand (`groups`, which must outlive the returned classes' use)."""
        rows, hashes = (bufs.exon_bits, bufs.exon_hash) if level == 0 else (bufs.gene_bits, bufs.gene_hash)
        h = C.c_void_p()
        if groups is not None:
            assert groups.level == level
            capi.check(capi.lib().hgx_level_classes_grouped(C.byref(h), locus.index(), capi.ptr(bufs.compat),
                                                             capi.ptr(dbatch.pair_off), capi.ptr(dbatch.pair_ref), groups.h,
                                                             capi.ptr(rows), capi.ptr(hashes), stream))
            return Classes(h)
        capi.check(capi.lib().hgx_level_classes(C.byref(h), locus.index(), capi.ptr(bufs.compat), capi.ptr(dbatch.pair_off),
                                                 capi.ptr(dbatch.pair_ref), C.c_int32(dbatch.n_pairs), C.c_int32(level),
                                                 capi.ptr(rows), capi.ptr(hashes), stream))
        return Classes(h)

    @staticmethod
    def of_pairs_fused(locus, dbatch, bufs, level, stream=None):
        """Classes of one level from per-pair rows that never reach memory (hgx_pair_classes_dedup, after piece_compat)."""
        rows = bufs.exon_bits if level == 0 else bufs.gene_bits
        h = C.c_void_p()
        capi.check(capi.lib().hgx_pair_classes_dedup(C.byref(h), locus.index(), capi.ptr(bufs.compat), capi.ptr(dbatch.pair_off),
                                                      capi.ptr(dbatch.pair_ref), C.c_int32(dbatch.n_pairs), C.c_int32(level),
                                                      capi.ptr(rows), stream))
        return Classes(h)

    @staticmethod
    def from_host(bits, counts, a_pad):
        bits = np.ascontiguousarray(bits, np.uint64)
        counts = np.ascontiguousarray(counts, np.int64)
        h = C.c_void_p()
        capi.check(capi.lib().hgx_classes_from_host(C.byref(h), capi.ptr(bits), capi.ptr(counts), C.c_int32(len(counts)),
                                                     C.c_int32(a_pad)))
        return Classes(h)

    def device_ptrs(self):
        b, c, f = C.c_void_p(), C.c_void_p(), C.c_void_p()
        capi.check(capi.lib().hgx_classes_device(self.h, C.byref(b), C.byref(c), C.byref(f)))
        return b.value, c.value, f.value

    def to_host(self):
        bits = np.zeros((self.n_classes, self.w64), np.uint64)
        cnt = np.zeros(self.n_classes, np.int64)
        first = np.zeros(self.n_classes, np.int64)
        capi.check(capi.lib().hgx_classes_to_host(self.h, capi.ptr(bits), capi.ptr(cnt), capi.ptr(first)))
        return bits, cnt, first

    def allele_counts(self, stream=None):
        """Gene_counts (core:1187-1190) and, per allele, the first class containing it."""
        cnt = np.zeros(self.a_pad, np.int64)
        first = np.zeros(self.a_pad, np.int32)
        capi.check(capi.lib().hgx_allele_counts_on(self.h, capi.ptr(cnt), capi.ptr(first), stream))
        return cnt, first

    def first_classes(self, alleles, stream=None):
        """First class (dict order) containing each of a few alleles (-1 if none)."""
        al = np.ascontiguousarray(alleles, np.int32)
        out = np.full(len(al), -1, np.int32)
        if len(al):
            capi.check(capi.lib().hgx_first_classes(self.h, capi.ptr(al), C.c_int32(len(al)), capi.ptr(out), stream))
        return out

    def em(self, n_alleles, remove_low=False, lengths=None, stream=None):
        """8a-8: single_abundance.  Returns (prob[n_alleles] with -1 for absent alleles, n_iter)."""
        prob = np.zeros(n_alleles, np.float64)
        it = C.c_int32(0)
        ln = None if lengths is None else np.ascontiguousarray(lengths, np.int32)
        capi.check(capi.lib().hgx_em(self.h, C.c_int32(n_alleles), C.c_int32(1 if remove_low else 0), capi.ptr(ln),
                                     capi.ptr(prob), C.byref(it), stream))
        return prob, it.value

    def em_ordered(self, n_alleles, remove_low=False, lengths=None, stream=None):
        """em() plus, for every allele of the result, the first class containing it (the tie order of the result list)."""
        prob = np.zeros(n_alleles, np.float64)
        first = np.zeros(n_alleles, np.int32)
        it = C.c_int32(0)
        ln = None if lengths is None else np.ascontiguousarray(lengths, np.int32)
        capi.check(capi.lib().hgx_em_ordered(self.h, C.c_int32(n_alleles), C.c_int32(1 if remove_low else 0), capi.ptr(ln),
                                             capi.ptr(prob), capi.ptr(first), C.byref(it), stream))
        return prob, first, it.value

    def em_masked(self, mask, n_alleles, remove_low=True, lengths=None, stream=None):
        """The exon -> gene hand-off (core:1752-1782): EM on this class set filtered to the alleles of `mask` (a_pad/64
        uint64 words).  Returns (prob, first_class, n_iter, n_merged_classes)."""
        mask = np.ascontiguousarray(mask, np.uint64)
        prob = np.zeros(n_alleles, np.float64)
        first = np.zeros(n_alleles, np.int32)
        it, nc = C.c_int32(0), C.c_int32(0)
        ln = None if lengths is None else np.ascontiguousarray(lengths, np.int32)
        capi.check(capi.lib().hgx_em_masked(self.h, capi.ptr(mask), C.c_int32(n_alleles), C.c_int32(1 if remove_low else 0),
                                            capi.ptr(ln), capi.ptr(prob), capi.ptr(first), C.byref(it), C.byref(nc), stream))
        return prob, first, it.value, nc.value

    def close(self):
        if self.h and self.owned:
            capi.lib().hgx_classes_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _RawDev:
    """Borrowed device pointer (e.g. the class matrix owned by a Classes handle)."""

    def __init__(self, p):
        self.ptr = p


def em_order(first_class, name_rank, present):
    """Insertion order of the EM's dict (common:1300-1305): alleles by (first class containing them,
    position inside that class' sorted key)."""
    idx = np.nonzero(np.asarray(present))[0]
    fc = np.asarray(first_class)[idx]
    nr = np.asarray(name_rank)[idx]
    return idx[np.lexsort((nr, fc))].tolist()


def em_last_exact():
    """True if the last EM on this thread ran in the reference's own order of operations (bit-identical abundances)."""
    return bool(capi.lib().hgx_em_last_exact())


def em_set_timing(on):
    """0 / False = off (totals kept), 1 / True = sampled passes, 2 = every plain mat-vec pass."""
    capi.check(capi.lib().hgx_em_set_timing(C.c_int(int(on))))


def em_get_timing():
    """{kernel name: (ms_total, launches, executed, bytes_total)} accumulated since em_set_timing(True)."""
    out = {}
    # slots 0/1 = rows pass (vector <= 8192 / larger), 2/3 = cols pass; the default (table-lookup) backend runs both sizes
    # with the same kernel, k_lutmatvec<0> / k_lutmatvec<1> in rocprofv3's naming
    # slot 4 = the resident-block EM (k_em_grid): whole launches; executed = applications of the EM map they ran
    for name, slots in (("k_lutmatvec<0>", (0, 1)), ("k_lutmatvec<1>", (2, 3)), ("k_em_grid", (4,))):
        tot = [0.0, 0, 0, 0]
        for slot in slots:
            ms, n, ex, by = C.c_double(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
            capi.check(capi.lib().hgx_em_get_timing(C.c_int(slot), C.byref(ms), C.byref(n), C.byref(ex), C.byref(by)))
            tot[0] += ms.value; tot[1] += n.value; tot[2] += ex.value; tot[3] += by.value
        out[name] = tuple(tot)
    return out


def test_switch(name, value="1"):
    """Test hook (hgx.h hgx_test_switch_set): set (value a string) or clear (value None) a named path-forcing switch;
    name None clears all.  The library reads no path-selecting environment variable."""
    capi.check(capi.lib().hgx_test_switch_set(None if name is None else name.encode(),
                                              None if value is None else str(value).encode()))


class test_switches:
    """with engine.test_switches(em_skip="wave"): ...  -- switches set inside the block, cleared after it"""
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        for k, v in self.kw.items():
            test_switch(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.kw:
            test_switch(k, None)
        return False


def em_set_backend(backend):
    """0 = auto (table lookup), 1 = EXEC-masked FP64 VALU mat-vec, 3 = table-lookup mat-vec (256 subset sums per 8 columns in
    LDS); 2 = int8 MFMA mat-vec (128-bit fixed point), lab build only (capi.use_lab())."""
    capi.check(capi.lib().hgx_em_set_backend(C.c_int(backend)))
