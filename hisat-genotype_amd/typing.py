"""Drop-in mirrors of the reference's entry points for the hot path.

  single_abundance(Gene_cmpt, remove_low_abundance_allele=False, Gene_length={})
        same signature / return value as hisatgenotype_typing_common.single_abundance (common:1282-1410);
        can be assigned over it (``typing_common.single_abundance = hgx.single_abundance``).
  type_locus(packed_locus, sam_text, ...)
        the per-locus body of typing() (typing_core.py:370-2122 minus assembly): SAM stream in,
        counts / classes / abundances / report lines out.  Scoring, dedup and EM run on the GPU.
  typing(...)
        same 38-parameter signature as hisatgenotype_typing_core.typing (core:249-286); writes the same
        report file and returns ``test_passed`` in simulation mode.

Nothing here computes the hot path on the CPU: all of it goes through libhgx (HIP kernels).
"""
import ctypes as C
import os
import sys
import time

import numpy as np

from . import capi, engine
from .locus import PackedLocus


TIE_REL_TOL = 1e-11
class _TypingOptions:
    """What typing() / genotyping_locus() have no parameter for (the reference's signatures are kept as they are).
    em_fast: the EM arithmetic of the per-locus body -- False (default): the reference's own order of floating-point operations
    wherever the one-workgroup kernel takes the problem (abundances equal to the reference's doubles; a 1 700-class EM #1 takes
    ~10 ms that way), True: table lookups (within 1e-8 of the reference, bar 1e-5; ~2 ms), None / -1: see typing._em_mode."""
    em_fast = False
    loci_side_by_side = True       # several loci of one typing() call: a host thread and stream per locus over ONE read of the file
    loci_together = False          # ... their batches typed by ONE hgx_type_many_loci call instead (measured slower at six loci)


typing_options = _TypingOptions()
last_profile = []        # typing(): one dict per locus of the last call -- where its wall time went (bench.py's drop-in legs print it)


def _stable_desc(lst, exact=False):
    """The reference's ``sorted(..., key=prob, reverse=True)`` (a STABLE sort: equal abundances keep dict insertion order).
    Alleles the data cannot tell apart come out of the reference's EM bit-identical; here they agree to ~1e-14 only (same
    arithmetic, different summation order on the GPU), so abundances within a relative 1e-11 of each other count as tied and
    keep their insertion order, as an exact tie does in the reference.  `exact`: the values ARE the reference's (the EM ran
    on one wavefront in the reference's order of operations, hgx_em_last_exact): plain stable sort, no tolerance."""
    tol = 0.0 if exact else TIE_REL_TOL
    idx = sorted(range(len(lst)), key=lambda i: -lst[i][1])            # stable on equal values
    out, i = [], 0
    while i < len(idx):
        j = i + 1
        top = lst[idx[i]][1]
        while j < len(idx) and top - lst[idx[j]][1] <= tol * abs(top):
            j += 1
        out.extend(lst[k] for k in sorted(idx[i:j]))                       # the tied run, back in insertion order
        i = j
    return out


def _keys_to_classes(keys):
    """Class keys ('-'.join(sorted(names))) -> (names in first-appearance order, bit rows [C][a_pad/64], a_pad, name rank or None):
    one pass over the joined text inside libhgx (hgx_keyset_*), not a Python loop per allele name -- a 5 000-class dict of a
    7 000-allele locus spells out millions of names."""
    L = capi.lib()
    text = "\n".join(keys).encode()
    if text.count(b"\n") != len(keys) - 1:
        raise ValueError("a compatibility-class key contains a newline")
    h = C.c_void_p()
    capi.check(L.hgx_keyset_create(C.byref(h), text, C.c_size_t(len(text)), C.c_int32(len(keys))))
    try:
        n, ap, nb, srt = C.c_int32(), C.c_int32(), C.c_size_t(), C.c_int32()
        capi.check(L.hgx_keyset_dims(h, C.byref(n), C.byref(ap), C.byref(nb), C.byref(srt)))
        bits = np.empty((len(keys), ap.value // 64), np.uint64)
        pool = C.create_string_buffer(max(nb.value, 1))
        rank = np.zeros(max(n.value, 1), np.int32)
        capi.check(L.hgx_keyset_fill(h, capi.ptr(bits), pool, capi.ptr(rank)))
        names = pool.raw[:nb.value].decode().split("\0")[:-1] if nb.value else []
        return names, bits, ap.value, (rank[:n.value] if srt.value else None)
    finally:
        L.hgx_keyset_destroy(h)


def single_abundance(Gene_cmpt, remove_low_abundance_allele=False, Gene_length={}):
    """EM abundance of alleles from compatibility classes ``{'a-b-c': count}`` (drop-in, GPU)."""
    if len(Gene_cmpt) == 0:
        return []
    names, bits, ap, rank = _keys_to_classes(list(Gene_cmpt.keys()))
    A = len(names)
    counts = np.fromiter(Gene_cmpt.values(), np.int64, len(Gene_cmpt))
    lengths = None
    if len(Gene_length) > 0:
        lengths = np.array([Gene_length[a] for a in names], np.int32)
    cl = engine.Classes.from_host(bits, counts, ap)
    try:
        if rank is not None:                               # keys as the reference builds them: '-'.join(sorted(names))
            cl.set_allele_rank(rank)
        prob, _ = cl.em(A, bool(remove_low_abundance_allele), lengths)
        exact = bool(capi.lib().hgx_em_last_exact())
        order = engine.em_last_order(A) if exact else None
    finally:
        cl.close()
    # allele index == first-appearance order here, which is the dict order of the reference -- unless the EM skipped classes
    # whose alleles_prob was 0 (common:1321): the kernel then reports the returned dict's own insertion order
    idx = [a for a in range(A) if prob[a] >= 0.0]
    if order is not None:
        idx.sort(key=lambda a: int(order[a]))
    res = [[names[a], float(prob[a])] for a in idx]
    return _stable_desc(res, exact)


class LocusResult:
    def __init__(self):
        self.num_reads = self.num_pairs = 0
        self._names = None
        self.counts_order = None    # allele indices ranked like the reference's sorted Gene_counts (core:1650-1651)
        self.counts = None          # Gene_counts per allele index
        self.exon_classes = None    # (bits[C][w64], counts[C]) host copies, first-seen order
        self.gene_classes = None
        self.em = []                # [{'n_classes', 'remove_low', 'use_length', 'result', 'n_iter'}]
        self.gene_prob = []         # final [[allele name, prob]] (core:1732-1789)
        self.n_pieces = self.n_refs = 0
        self.t_em = 0.0             # seconds spent inside the EM calls (bench)

    @property
    def counts_sorted(self):
        """[[allele name, count]] in the reference's print order (materialised on demand)."""
        if self.counts_order is None:
            return []
        return [[self._names[a], int(self.counts[a])] for a in self.counts_order]


def _em_mode(em_fast):
    """hgx_type_opts.em_fast: None = the entry point's default (one-task calls: the reference's order where the one-workgroup kernel
    takes the problem; many-task calls: table lookups), False = the reference's order (2: also in the many-task calls), True / 1 =
    table lookups, -1 = the reference's order at every size (validation mode: slow beyond 4096 classes)."""
    if em_fast is None:
        return 0
    if em_fast is False or (em_fast is not True and int(em_fast) == 0):      # False, 0, numpy.bool_(False): all mean "the reference's order"
        return 2
    return -1 if (em_fast is not True and int(em_fast) < 0) else (1 if (em_fast is True or int(em_fast) == 1) else int(em_fast))


def type_locus(pl, sam_text, num_editdist=2, error_correction=True, allow_discordant=False,
               remove_low_abundance_alleles=True, simulation=False, base_locus=0, keep_classes=False, stream=None,
               alignment_file=None, regions=None, gate=None, per_pair_exon=False, em_fast=False, profile=None, alignment=None):
    """Per-locus typing: the reference's loop body with the O(alleles) work on the GPU.  Input: name-grouped SAM text
    (`sam_text`), or `alignment_file` (SAM / BAM; `regions` = samtools region strings, see read_alignment_text) read inside
    libhgx.  `gate` (engine.Gate): shared by the samples in flight on one GPU, see _type_batch.  `em_fast`: see _em_mode
    (False = the library's default for one task; -1 = the reference's order of floating-point operations at every size)."""
    res = LocusResult()
    t0 = time.perf_counter()
    # the front end on the device (hgx_parse_*_dev: record fields, filters, key grouping, pileup, decode, piece table and pair protocol
    # as kernels; small or unusual inputs are finished by the host stages inside the same call): the batch is born in HBM
    if alignment is not None:                                   # engine.Alignment: the file was read once, its bytes are in HBM
        dbatch = alignment.parse_dev(pl, regions, num_editdist=num_editdist, error_correction=error_correction,
                                     allow_discordant=allow_discordant, simulation=simulation, base_locus=base_locus, stream=stream)
    elif alignment_file is not None:
        dbatch = pl.parse_alignment_file_dev(alignment_file, regions, num_editdist=num_editdist, error_correction=error_correction,
                                             allow_discordant=allow_discordant, simulation=simulation, base_locus=base_locus, stream=stream)
    else:
        dbatch = pl.parse_sam_dev(sam_text, num_editdist=num_editdist, error_correction=error_correction,
                                  allow_discordant=allow_discordant, simulation=simulation, base_locus=base_locus, stream=stream)
    t1 = time.perf_counter()
    try:
        res.num_reads, res.num_pairs = dbatch.n_reads, dbatch.n_pairs
        res.n_pieces, res.n_refs = dbatch.n_pieces, dbatch.n_refs
        if dbatch.n_reads <= 0:                                 # core:1589-1590
            return res
        return _type_batch(pl, None, res, remove_low_abundance_alleles, keep_classes, stream, dbatch=dbatch, gate=gate, per_pair_exon=per_pair_exon,
                           em_fast=em_fast)
    finally:
        dbatch.close()
        if profile is not None:
            profile["file_read_and_front_end_ms"] = (t1 - t0) * 1e3
            profile["front_end_route"] = list(engine.front_last())
            profile["gpu_typing_and_result_ms"] = (time.perf_counter() - t1) * 1e3


class TypeOpts(C.Structure):
    """hgx_type_opts (include/hgx.h)."""
    _fields_ = [("remove_low", C.c_int32), ("keep_classes", C.c_int32), ("overlap", C.c_int32), ("per_pair_exon", C.c_int32),
                ("gate", C.c_void_p), ("ev_compat_begin", C.c_void_p), ("ev_compat_end", C.c_void_p),
                ("ev_pairs_begin", C.c_void_p), ("ev_pairs_end", C.c_void_p), ("em_fast", C.c_int32)]


def _result_from_handle(h, pl, res, keep_classes):
    """Copy an hgx_typing result into a LocusResult (allele indices -> names)."""
    L = capi.lib()
    names = pl.names
    n_counted, n_em, n_gp, t_em = C.c_int32(), C.c_int32(), C.c_int32(), C.c_double()
    capi.check(L.hgx_typing_dims(h, None, None, None, None, C.byref(n_counted), C.byref(n_em), C.byref(n_gp), C.byref(t_em)))
    order = np.zeros(max(n_counted.value, 1), np.int32)
    cnt = np.zeros(max(pl.n_alleles, 1), np.int64)
    capi.check(L.hgx_typing_counts(h, capi.ptr(order), capi.ptr(cnt)))
    res._names, res.counts_order, res.counts = names, order[:n_counted.value], cnt[:pl.n_alleles]
    res.t_em = t_em.value
    res.em = []
    for k in range(n_em.value):
        nc, it, low, ln, nr = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        capi.check(L.hgx_typing_em(h, C.c_int32(k), C.byref(nc), C.byref(it), C.byref(low), C.byref(ln), C.byref(nr), None, None))
        al, pr = np.zeros(max(nr.value, 1), np.int32), np.zeros(max(nr.value, 1), np.float64)
        capi.check(L.hgx_typing_em(h, C.c_int32(k), None, None, None, None, None, capi.ptr(al), capi.ptr(pr)))
        res.em.append({"n_classes": nc.value, "remove_low": bool(low.value), "use_length": bool(ln.value),
                       "result": [[names[a], float(p)] for a, p in zip(al[:nr.value].tolist(), pr[:nr.value].tolist())],
                       "n_iter": it.value})
    al, pr = np.zeros(max(n_gp.value, 1), np.int32), np.zeros(max(n_gp.value, 1), np.float64)
    capi.check(L.hgx_typing_gene_prob(h, capi.ptr(al), capi.ptr(pr)))
    res.gene_prob = [[names[a], float(p)] for a, p in zip(al[:n_gp.value].tolist(), pr[:n_gp.value].tolist())]
    if keep_classes:
        for level, attr in ((0, "exon_classes"), (1, "gene_classes")):
            ch = C.c_void_p()
            capi.check(L.hgx_typing_classes(h, C.c_int32(level), C.byref(ch)))
            if ch:
                setattr(res, attr, engine.Classes(ch, owned=False).to_host()[:2])
    return res


def _type_batch(pl, batch, res, remove_low, keep_classes=False, stream=None, dbatch=None, overlap=None, gate=None, events=None,
                em_fast=False, per_pair_exon=False):
    """The per-locus body of typing() for one piece batch: ONE call into libhgx (hgx_type_dbatch / hgx_type_batch, which
    orchestrate scoring, grouping, dedup, Gene_counts, both EMs and the hand-off on the GPU; typing_core.py:1589-1789).
    `dbatch`: the batch already resident in HBM (engine.DeviceBatch; bench.py types it repeatedly); `overlap`: None = the
    library's default (gene side beside the exon-level EM when `stream` is None), else forced; `gate` (engine.Gate): several
    samples in flight on one GPU take turns with their bandwidth-bound front; `events` = (compat begin, compat end, pairs
    begin, pairs end) capi.Event objects recorded around hgx_piece_compat and the gene-level hgx_pair_classes launch."""
    o = TypeOpts(int(bool(remove_low)), int(bool(keep_classes)), -1 if overlap is None else int(bool(overlap)), int(bool(per_pair_exon)),
                 gate.h if gate is not None else None, *[(e.h if e is not None else None) for e in (events or (None,) * 4)],
                 _em_mode(em_fast))
    h = C.c_void_p()
    L = capi.lib()
    if dbatch is not None:
        rc = L.hgx_type_dbatch(C.byref(h), pl.h, pl.index(), dbatch.h, C.byref(o), stream)
    else:
        rc = L.hgx_type_batch(C.byref(h), pl.h, pl.index(), batch.h, C.byref(o), stream)
    if rc == -7:                                               # HGX_ETYPE: the reference's quirk Q3 (typing_core.py:1787)
        raise TypeError(L.hgx_last_error().decode(errors="replace"))
    capi.check(rc)
    try:
        return _result_from_handle(h, pl, res, keep_classes)
    finally:
        L.hgx_typing_destroy(h)


def type_file(pl, alignment_fname, regions=None, num_editdist=2, error_correction=True, allow_discordant=False,
              remove_low_abundance_alleles=True, simulation=False, base_locus=0, stream=None, em_fast=False):
    """Alignment file (SAM text or BAM, any order) -> typing result in ONE call into libhgx (hgx_type_file): read / inflate /
    decode, region filter, name grouping, front-end, upload, device path, result -- the whole per-locus work of typing()
    (typing_core.py:436-468 + 800-1789).  `regions`: samtools region strings (list or newline-separated); None = the locus'
    backbone, as the reference's `samtools view F ref_allele` (core:443-444)."""
    if regions is None:
        regions = [pl.ref_allele]
    if not isinstance(regions, (str, bytes)):
        regions = "\n".join(regions)
    reg = regions.encode() if isinstance(regions, str) else regions
    po = capi.ParseOpts(int(num_editdist), int(bool(error_correction)), int(bool(allow_discordant)), int(bool(simulation)), int(base_locus),
                        0, int(pl.base_fname == "codis" and pl.gene == "D18S51"), 0)
    to = TypeOpts(int(bool(remove_low_abundance_alleles)), 0, -1, 0, None, None, None, None, None, _em_mode(em_fast))
    h = C.c_void_p()
    L = capi.lib()
    rc = L.hgx_type_file(C.byref(h), pl.h, pl.index(), alignment_fname.encode(), reg or None, C.byref(po), C.byref(to), stream)
    if rc == -7:
        raise TypeError(L.hgx_last_error().decode(errors="replace"))
    capi.check(rc)
    try:
        res = LocusResult()
        nr, npair, npc, nrf = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
        capi.check(L.hgx_typing_dims(h, C.byref(nr), C.byref(npair), C.byref(npc), C.byref(nrf), None, None, None, None))
        res.num_reads, res.num_pairs, res.n_pieces, res.n_refs = nr.value, npair.value, npc.value, nrf.value
        if res.num_reads > 0:
            _result_from_handle(h, pl, res, False)
        return res
    finally:
        L.hgx_typing_destroy(h)


def type_many(pl, many, remove_low=True, stream=None, return_errors=False, em_fast=None):
    """Every task of a merged batch (engine.ManyBatch: many samples of ONE locus) in one call into libhgx (hgx_type_many): one
    launch chain for all tasks instead of one per task -- the many-samples form of the per-locus body of typing()
    (typing_core.py:370, /root/reference/hisatgenotype:613-665).  Returns one LocusResult per task, each identical to
    _type_batch on that task alone.  Where the reference would raise on a task (quirk Q3 / Q6) the call raises, unless
    `return_errors`: then that task's entry is the exception instance."""
    L = capi.lib()
    n = many.n_tasks
    o = TypeOpts(int(bool(remove_low)), 0, 0, 0, None, None, None, None, None, _em_mode(em_fast))
    hs = (C.c_void_p * max(n, 1))()
    rcs = (C.c_int32 * max(n, 1))()
    capi.check(L.hgx_type_many(hs, rcs, pl.h, pl.index(), many.h, C.byref(o), stream))
    out = []
    try:
        for t in range(n):
            if rcs[t] != 0:
                err = TypeError("'dict_keys' object is not subscriptable (reference quirk Q3, typing_core.py:1787)") if rcs[t] == -7 else \
                    capi.HgxKeyError(rcs[t], "EM: allele missing from the next estimate (common:1365-1369)") if rcs[t] == -4 else \
                    capi.HgxError(rcs[t], "task %d failed" % t)
                if not return_errors:
                    raise err
                out.append(err)
                continue
            res = LocusResult()
            res.num_reads, res.num_pairs = many.task_reads[t], many.task_pairs[t]
            if res.num_reads > 0:
                _result_from_handle(C.c_void_p(hs[t]), pl, res, False)
            out.append(res)
    finally:
        for t in range(n):
            if hs[t]:
                L.hgx_typing_destroy(C.c_void_p(hs[t]))
    return out


def type_many_loci(pls, manies, remove_low=True, stream=None, light=False, em_fast=None):
    """A whole panel in one call (hgx_type_many_loci): `manies[i]` = the merged batch of locus `pls[i]`'s samples.  The loci are
    scored side by side (a host thread and stream pair per locus); the EMs of all their tasks go out in ONE launch.  Returns a list (per locus) of lists (per task)
    of LocusResult -- or, with `light`, of (num_reads, [top-2 allele names], EM iterations): what a throughput run looks at."""
    L = capi.lib()
    nl = len(pls)
    o = TypeOpts(int(bool(remove_low)), 0, 0, 0, None, None, None, None, None, _em_mode(em_fast))
    hs = [(C.c_void_p * max(m.n_tasks, 1))() for m in manies]
    rcs = [(C.c_int32 * max(m.n_tasks, 1))() for m in manies]
    out_pp = (C.POINTER(C.c_void_p) * max(nl, 1))(*[C.cast(h, C.POINTER(C.c_void_p)) for h in hs])
    rc_pp = (C.POINTER(C.c_int32) * max(nl, 1))(*[C.cast(r, C.POINTER(C.c_int32)) for r in rcs])
    loc_p = (C.c_void_p * max(nl, 1))(*[pl.h for pl in pls])
    ix_p = (C.c_void_p * max(nl, 1))(*[pl.index() for pl in pls])
    many_p = (C.c_void_p * max(nl, 1))(*[m.h for m in manies])
    prof = "HGX_TYPE_PROFILE" in os.environ
    t0 = time.perf_counter()
    capi.check(L.hgx_type_many_loci(C.c_int32(nl), out_pp, rc_pp, loc_p, ix_p, many_p, C.byref(o), stream))
    t1 = time.perf_counter()
    out = []
    try:
        for i, (pl, many) in enumerate(zip(pls, manies)):
            row = []
            for t in range(many.n_tasks):
                if rcs[i][t] != 0:
                    raise capi.HgxError(rcs[i][t], "task %d of locus %d failed (the reference would raise here)" % (t, i))
            if light:                                        # one call for the locus' tasks (hgx_typing_top)
                n = many.n_tasks
                nr, ne = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
                al, pr = np.zeros((max(n, 1), 2), np.int32), np.zeros((max(n, 1), 2), np.float64)
                capi.check(L.hgx_typing_top(hs[i], C.c_int32(n), C.c_int32(2), capi.ptr(nr), capi.ptr(ne), capi.ptr(al), capi.ptr(pr)))
                names = pl.names
                out.append([(int(nr[t]), [names[a] for a in al[t].tolist() if a >= 0], int(ne[t])) for t in range(n)])
                continue
            for t in range(many.n_tasks):
                h = C.c_void_p(hs[i][t])
                res = LocusResult()
                res.num_reads, res.num_pairs = many.task_reads[t], many.task_pairs[t]
                if res.num_reads > 0:
                    _result_from_handle(h, pl, res, False)
                row.append(res)
            out.append(row)
    finally:
        t2 = time.perf_counter()
        for i, many in enumerate(manies):
            for t in range(many.n_tasks):
                if hs[i][t]:
                    L.hgx_typing_destroy(C.c_void_p(hs[i][t]))
        if prof:
            print("[type_many_loci] call %.2f ms | results %.2f | destroy %.2f" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (time.perf_counter() - t2) * 1e3),
                  file=sys.stderr)
    return out


def report_lines(res, simulation=False, true_alleles=(), output_allele_counts=False, best_alleles=False):
    """Report body (core:1593, 1650-1677, 2076-2121)."""
    out = ["\t\t\t%d reads and %d pairs are aligned" % (res.num_reads, res.num_pairs)]
    if not simulation and res.counts_order is not None:
        # real data: the first ten counted alleles, or all of them (--output-allele-counts: thousands of lines) -- one formatting pass
        order = res.counts_order if output_allele_counts else res.counts_order[:10]
        names, counts = res._names, res.counts
        out.extend(map("\t\t\t\t%d %s (count: %d)".__mod__,
                       zip(range(1, len(order) + 1), [names[a] for a in order.tolist()], counts[order].tolist())))
    for i, (a, c) in enumerate(res.counts_sorted if simulation else ()):
        if simulation:
            found = False
            for t in true_alleles:
                if a == t:
                    out.append("\t\t\t*** %d ranked %s (count: %d)" % (i + 1, t, c))
                    found = True
            if i < 5 and not found:
                out.append("\t\t\t\t%d %s (count: %d)" % (i + 1, a, c))
        else:
            out.append("\t\t\t\t%d %s (count: %d)" % (i + 1, a, c))
            if i >= 9 and not output_allele_counts:
                break
    out.append("\n")
    success = [False] * len(true_alleles)
    found_list = [False] * len(true_alleles)
    for i, (a, p) in enumerate(res.gene_prob):
        if p < 0.01:
            break
        found = False
        if simulation:
            for k, t in enumerate(true_alleles):
                if a == t:
                    out.append("\t\t\t*** %d ranked %s (abundance: %.2f%%)" % (i + 1, t, p * 100.0))
                    if i < len(success):
                        success[i] = True
                    found_list[k] = True
                    found = True
            if False not in found_list and i >= 10:
                break
        if not found:
            out.append("\t\t\t\t%d ranked %s (abundance: %.2f%%)" % (i + 1, a, p * 100.0))
            if best_alleles and i < 2:
                out.append("SingleModel %s (abundance: %.2f%%)" % (a, p * 100.0))
        if not simulation and i >= 9:
            break
        if i >= 19:
            break
    return out, success


def read_alignment_text(alignment_fname, regions=None, n_threads=0, native=True):
    """The record stream the reference's loop consumes: ``samtools view F [chr:l-r] ref_allele`` piped through
    ``sort -k1,1 -s`` (core:436-468), as bytes.  SAM text and BAM files are read by the native reader of libhgx
    (hgx_read_alignments: parallel BGZF inflate, BAM decode, region overlap filter and name grouping; no samtools needed).
    `regions`: samtools region strings (list, or newline-separated), None = every record.  ``native=False`` uses the
    pure-Python statement of the same formats and rules (bamio.py), kept for tests."""
    from . import bamio
    if regions is not None and not isinstance(regions, (str, bytes)):
        regions = "\n".join(regions)
    if native:
        import ctypes as C
        text, nbytes = C.c_void_p(), C.c_size_t(0)
        reg = regions.encode() if isinstance(regions, str) else regions
        capi.check(capi.lib().hgx_read_alignments(alignment_fname.encode(), reg or None, C.c_int32(n_threads),
                                                  C.byref(text), C.byref(nbytes)))
        try:
            return C.string_at(text.value, nbytes.value)
        finally:
            capi.lib().hgx_free_text(text)
    with open(alignment_fname, "rb") as f:
        head = f.read(4)
    if head[:2] == b"\x1f\x8b" or head == b"BAM\x01":
        lines = [l.encode() for l in bamio.read_bam(alignment_fname, regions)]
    else:
        with open(alignment_fname, "rb") as f:
            data = f.read()
        lines = [l[:-1] if l.endswith(b"\r") else l for l in data.split(b"\n")]
        lines = [l for l in lines if l and not l.startswith(b"@")]
        regs = bamio.normalise_regions(regions)
        if regs is not None:
            per_region = [[] for _ in regs]
            for l in lines:
                f = l.split(b"\t")
                if len(f) < 7:
                    continue
                flag, pos0 = int(f[1]), int(f[3]) - 1
                reflen = 0 if (flag & 4) else bamio.cigar_reflen(f[5].decode())
                for g, r in enumerate(regs):
                    if bamio.region_hit(r, f[2].decode(), pos0, pos0 + max(reflen, 1) - 1):
                        per_region[g].append(l)
            lines = [l for v in per_region for l in v]
    lines.sort(key=lambda l: l.split(b"\t", 1)[0])      # stable; bytewise like LC_ALL=C sort -k1,1 -s
    return b"".join(l + b"\n" for l in lines)


def typing(simulation, full_path_base_fname, locus_list, genotype_genome, partial, partial_alleles, refGenes, Genes,
           Gene_names, Gene_lengths, refGene_loci, Vars, Var_list, Links, aligners, num_editdist, assembly, output_base,
           error_correction, keep_alignment, allow_discordant, type_primary_exons, remove_low_abundance_alleles,
           display_alleles, fastq, read_fname, alignment_fname, num_frag_list, read_len, fragment_len, threads,
           best_alleles, verbose, assembly_verbose, out_dir, dbversion, output_allele_counts, test_i=0):
    """Same contract as hisatgenotype_typing_core.typing (core:249-286): writes
    ``<out_dir>/<output_base>-<base>.<id>.report``; returns ``test_passed`` in simulation mode.  With
    ``alignment_fname == ""`` the reads are aligned first (simulate.align_reads: HISAT2 when installed; simulated reads are
    placed by the alignment their names spell).  --assembly is outside this path."""
    if assembly:
        raise NotImplementedError("--assembly (assembly graph) is outside the accelerated path")
    base_fname = full_path_base_fname.split("/")[-1]
    report_base = "%s/%s-%s." % (out_dir, output_base, base_fname)
    test_passed = {}
    del last_profile[:]
    if simulation:
        core_fid = str(test_i + 1)
        report_base += "test-"
    else:
        core_fid = "_".join(read_fname[0].split("/")[-1].split(".")[:-1])
    report_base += core_fid
    with open("%s.report" % report_base, "w") as report_file:
        msg_out = [sys.stderr, report_file] if (verbose or assembly_verbose or simulation) else [report_file]

        def say(s):
            for f_ in msg_out:
                print(s, file=f_)

        say("# VERSIONS:")
        say("# HISAT-genotype hot path: hgx %s (MI355X)" % __import__("hisatgenotype_amd").__version__)
        say("# Database - %s" % dbversion)
        say("# COMMAND:\n%s" % " ".join(sys.argv))
        for aligner, index_type in aligners:
            if index_type != "graph":
                raise NotImplementedError("only graph alignments are on the accelerated path")
            say("\n\t\t%s %s" % (aligner, index_type))
            remove_alignment_file = False
            if alignment_fname == "":                     # core:346-367: align the reads first
                from . import simulate
                remove_alignment_file = True
                alignment_fname = "%s_output.bam" % base_fname if simulation else "%s.bam" % core_fid
                gegenome = genotype_genome if genotype_genome != "" else full_path_base_fname + "." + index_type
                simulate.align_reads(aligner, simulation, gegenome, index_type, base_fname, read_fname, fastq, threads,
                                     alignment_fname, verbose, truth=(Genes, Vars, refGenes))
            # The reference's loop `for test_Gene_names in locus_list` (core:370) runs `samtools view F ref_allele | sort` per locus:
            # here the file is read ONCE (engine.Alignment: its bytes stay in HBM) and the loci -- independent of each other in the
            # reference too: every per-locus structure is rebuilt -- are typed side by side, one host thread and stream per locus;
            # their report sections are written in locus_list order afterwards.
            jobs = []
            for test_Gene_names in locus_list:
                gene = test_Gene_names[0].split("*")[0] if simulation else test_Gene_names
                prof = {"gene": gene}
                t_ = time.perf_counter()
                # the packed locus and its device index come from the in-process cache when this process has seen these dicts --
                # or dicts with the same content -- before (locus.LocusCache): typing() runs once per sample on one index
                pl = PackedLocus.cached_from_reference_dicts(gene, base_fname, refGenes, Genes, Gene_names, Gene_lengths,
                                                             refGene_loci, Vars, Var_list, Links)
                prof["locus_packing_ms"] = (time.perf_counter() - t_) * 1e3
                t_ = time.perf_counter()
                pl.index()
                prof["index_upload_ms"] = (time.perf_counter() - t_) * 1e3
                # samtools view F [chr:l-r] ref_allele (core:436-444): the view is ALWAYS restricted to this gene's backbone
                # sequence -- a multi-locus alignment (hla graph: A, B, C, ... in one BAM) never leaks other genes' reads
                # into this gene's decode -- and in genotype-genome mode to the locus span on the chromosome as well
                regions, base_locus = [refGenes[gene]], 0
                if genotype_genome != "":
                    _, chr_, left, right = refGene_loci[gene][:4]
                    regions, base_locus = ["%s:%d-%d" % (chr_, left + 1, right + 1), refGenes[gene]], left
                jobs.append([test_Gene_names, pl, regions, base_locus, prof, None, None])
            t_ = time.perf_counter()
            al = engine.Alignment(alignment_fname) if len(jobs) > 1 else None
            t_open = (time.perf_counter() - t_) * 1e3

            def run_job(job, stream=None):
                # ... piped through sort -k1,1 -s (core:458-468), and the loop's decode: all inside libhgx
                try:
                    job[5] = type_locus(job[1], None, num_editdist=num_editdist, error_correction=error_correction,
                                        allow_discordant=allow_discordant, remove_low_abundance_alleles=remove_low_abundance_alleles,
                                        simulation=simulation, base_locus=job[3], alignment_file=alignment_fname, regions=job[2],
                                        profile=job[4], em_fast=typing_options.em_fast, alignment=al, stream=stream)
                except BaseException as e:          # re-raised on the caller's thread, at this locus' place in the loop
                    job[6] = e
            def parse_job(job, stream):
                # phase 1 of the side-by-side form: this locus' records out of the resident file -> its piece batch in HBM
                t0 = time.perf_counter()
                try:
                    job[7] = al.parse_dev(job[1], job[2], num_editdist=num_editdist, error_correction=error_correction,
                                          allow_discordant=allow_discordant, simulation=simulation, base_locus=job[3], stream=stream)
                    job[4]["front_end_route"] = list(engine.front_last())
                except BaseException as e:
                    job[6] = e
                job[4]["file_read_and_front_end_ms"] = (time.perf_counter() - t0) * 1e3
            try:
                if al is not None and al.resident and typing_options.loci_side_by_side and not typing_options.loci_together:
                    # every locus' whole body -- its records out of the resident file, scoring, dedup, both EMs -- on a thread and
                    # stream set of its own (the stream sets keep the loci's EM chains on different hardware lanes)
                    import threading
                    dev = capi.current_device()
                    n_workers = min(len(jobs), 8)

                    def worker(k):
                        capi.set_device(dev)
                        capi.set_stream_slot(("typing loci", k))
                        st = capi.get_stream(2)
                        for job in jobs[k::n_workers]:
                            run_job(job, st)
                        capi.sync(st)
                    ths = [threading.Thread(target=worker, args=(k,)) for k in range(n_workers)]
                    for t in ths:
                        t.start()
                    for t in ths:
                        t.join()
                elif al is not None and al.resident and typing_options.loci_side_by_side:
                    # measured and NOT the default (bench.py sample6: 23.9 ms against 16.6): the loci parsed side by side, then typed
                    # TOGETHER through hgx_type_many_loci (one task per locus: the many-task path's set-up costs more than one
                    # launch for all EMs saves at six tasks)
                    import threading
                    dev = capi.current_device()
                    for job in jobs:
                        job.append(None)                    # [7]: the locus' device batch
                    n_workers = min(len(jobs), 8)

                    def worker(k):
                        capi.set_device(dev)
                        capi.set_stream_slot(("typing loci", k))
                        st = capi.get_stream(2)
                        for job in jobs[k::n_workers]:
                            parse_job(job, st)
                        capi.sync(st)
                    ths = [threading.Thread(target=worker, args=(k,)) for k in range(n_workers)]
                    for t in ths:
                        t.start()
                    for t in ths:
                        t.join()
                    # phase 2: every locus that has reads, typed TOGETHER (hgx_type_many_loci: scoring side by side, the EMs of all
                    # loci in ONE launch -- a lone sizeable reference-order problem each: workgroup clusters side by side)
                    t_ = time.perf_counter()
                    live = [job for job in jobs if job[6] is None and job[7] is not None and job[7].n_reads > 0]
                    for job in jobs:
                        if job[6] is None and job not in live:
                            job[5] = LocusResult()
                    manies = []
                    try:
                        for job in live:
                            manies.append(engine.ManyBatch.from_dbatch(job[1], job[7]))
                        rows = type_many_loci([job[1] for job in live], manies, remove_low=remove_low_abundance_alleles,
                                              em_fast=typing_options.em_fast) if live else []
                        for job, row in zip(live, rows):
                            job[5] = row[0]
                    except (capi.HgxError, TypeError, KeyError):
                        # a locus the reference would raise on (quirks Q3 / Q6): locus after locus instead, so that the loci before it
                        # still get their report sections and the exception is this locus' own
                        for job in jobs:
                            if job[5] is None and job[6] is None:
                                run_job(job)
                                if job[6] is not None:
                                    break
                    finally:
                        for m in manies:
                            m.close()
                        for job in jobs:
                            if job[7] is not None:
                                job[7].close()
                    for job in jobs:
                        job[4]["gpu_typing_all_loci_ms_shared"] = (time.perf_counter() - t_) * 1e3
                else:
                    for job in jobs:
                        run_job(job)
                        if job[6] is not None:
                            break
            finally:
                if al is not None:
                    al.close()
            for test_Gene_names, pl, _, _, prof, res, err, *_rest in jobs:
                if not getattr(pl, "cached", False):
                    pl.close()
                if err is not None:
                    raise err
                if res is None:
                    break
                if al is not None:
                    prof["alignment_open_ms_shared"] = t_open
                last_profile.append(prof)
                if res.num_reads <= 0:
                    continue
                t_ = time.perf_counter()
                lines, success = report_lines(res, simulation, test_Gene_names if simulation else (),
                                              output_allele_counts, best_alleles)
                say("\n".join(lines))
                prof["report_ms"] = (time.perf_counter() - t_) * 1e3
                if simulation:
                    for ok in success:
                        if ok:
                            key = "%s %s" % (aligner, index_type)
                            test_passed[key] = test_passed.get(key, 0) + 1
            if not keep_alignment and remove_alignment_file:                   # core:2144-2145
                for f in [alignment_fname] + [alignment_fname + ext for ext in (".bai", ".unsorted")]:
                    if os.path.exists(f):
                        os.remove(f)
    if simulation:
        return test_passed
