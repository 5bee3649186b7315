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
import os
import subprocess
import sys
import threading
import time

import numpy as np

from . import capi, engine
from .locus import PackedLocus


TIE_REL_TOL = 1e-11


def _stable_desc(lst):
    """The reference's ``sorted(..., key=prob, reverse=True)`` (a STABLE sort: equal abundances keep dict insertion order).
    Alleles the data cannot tell apart come out of the reference's EM bit-identical; here they agree to ~1e-14 only (same
    arithmetic, different summation order on the GPU), so abundances within a relative 1e-11 of each other count as tied and
    keep their insertion order, as an exact tie does in the reference."""
    idx = sorted(range(len(lst)), key=lambda i: -lst[i][1])            # stable on equal values
    out, i = [], 0
    while i < len(idx):
        j = i + 1
        top = lst[idx[i]][1]
        while j < len(idx) and top - lst[idx[j]][1] <= TIE_REL_TOL * abs(top):
            j += 1
        out.extend(lst[k] for k in sorted(idx[i:j]))                       # the tied run, back in insertion order
        i = j
    return out


def _sorted_result(prob, order):
    """[[allele index, prob]] for present alleles in dict order, then the reference's stable
    descending sort (common:1408-1409)."""
    lst = [[a, float(prob[a])] for a in order]
    return _stable_desc(lst)


def _em_on_classes(classes, n_alleles, name_rank, remove_low, lengths, stream=None):
    classes.set_allele_rank(name_rank)      # small problems are then summed in the reference's own order (bit-identical)
    prob, first, n_iter = classes.em_ordered(n_alleles, remove_low, lengths, stream)
    present = np.nonzero(prob >= 0.0)[0]
    # dict insertion order of the survivors (common:1300-1305): first class containing each, then name order
    order = present[np.lexsort((np.asarray(name_rank)[present], first[present]))].tolist()
    return _sorted_result(prob, order), n_iter


def single_abundance(Gene_cmpt, remove_low_abundance_allele=False, Gene_length={}):
    """EM abundance of alleles from compatibility classes ``{'a-b-c': count}`` (drop-in, GPU)."""
    names, idx = [], {}
    split = []
    for key in Gene_cmpt.keys():
        al = key.split("-")
        for a in al:
            if a not in idx:
                idx[a] = len(names)
                names.append(a)
        split.append(al)
    A = len(names)
    if A == 0:
        return []
    ap = capi.a_pad(A)
    bits = np.zeros((len(split), ap // 64), np.uint64)
    for c, al in enumerate(split):
        ii = np.fromiter((idx[a] for a in al), np.int64, len(al))
        np.bitwise_or.at(bits[c], ii >> 6, np.uint64(1) << (ii & 63).astype(np.uint64))
    counts = np.fromiter(Gene_cmpt.values(), np.int64, len(split))
    lengths = None
    if len(Gene_length) > 0:
        lengths = np.array([Gene_length[a] for a in names], np.int32)
    cl = engine.Classes.from_host(bits, counts, ap)
    try:
        if all(al == sorted(al) for al in split):      # keys as the reference builds them: '-'.join(sorted(names))
            rank = np.zeros(A, np.int32)
            rank[np.array(sorted(range(A), key=lambda i: names[i]), np.int64)] = np.arange(A, dtype=np.int32)
            cl.set_allele_rank(rank)
        prob, _ = cl.em(A, bool(remove_low_abundance_allele), lengths)
    finally:
        cl.close()
    # allele index == first-appearance order here, which is the dict order of the reference
    res = [[names[a], float(prob[a])] for a in range(A) if prob[a] >= 0.0]
    return _stable_desc(res)


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


def type_locus(pl, sam_text, num_editdist=2, error_correction=True, allow_discordant=False,
               remove_low_abundance_alleles=True, simulation=False, base_locus=0, keep_classes=False, stream=None,
               alignment_file=None, regions=None, heavy_lock=None):
    """Per-locus typing: the reference's loop body with the O(alleles) work on the GPU.  Input: name-grouped SAM text
    (`sam_text`), or `alignment_file` (SAM / BAM; `regions` = samtools region strings, see read_alignment_text) read inside
    libhgx."""
    res = LocusResult()
    if alignment_file is not None:
        batch = pl.parse_alignment_file(alignment_file, regions, num_editdist=num_editdist, error_correction=error_correction,
                                        allow_discordant=allow_discordant, simulation=simulation, base_locus=base_locus)
    else:
        batch = pl.parse_sam(sam_text, num_editdist=num_editdist, error_correction=error_correction,
                             allow_discordant=allow_discordant, simulation=simulation, base_locus=base_locus)
    res.num_reads, res.num_pairs = batch.n_reads, batch.n_pairs
    res.n_pieces, res.n_refs = batch.n_pieces, batch.n_refs
    if batch.n_reads <= 0:                                  # core:1589-1590
        return res
    if heavy_lock is None:
        return _type_batch(pl, batch, res, remove_low_abundance_alleles, keep_classes, stream)
    held = _HeldOnce(heavy_lock)        # several samples in flight: one bandwidth-bound front at a time (see _type_batch)
    try:
        return _type_batch(pl, batch, res, remove_low_abundance_alleles, keep_classes, stream, heavy_lock=held)
    finally:
        held.release()


class _HeldOnce:
    """A held lock that is released exactly once (by _type_batch as soon as it can, or by the caller on the way out)."""

    def __init__(self, lock):
        lock.acquire()
        self.lock = lock

    def release(self):
        lock, self.lock = self.lock, None
        if lock is not None:
            lock.release()


_tls = threading.local()


def _fork_event():
    ev = getattr(_tls, "fork_event", None)
    if ev is None:
        ev = _tls.fork_event = capi.Event()
    return ev


def _type_batch(pl, batch, res, remove_low, keep_classes=False, stream=None, dbatch=None, bufs=None, scored=False, overlap=None,
                heavy_lock=None, pc_events=None):
    """`scored`: False = nothing computed yet; "compat" = the caller queued hgx_piece_compat on `stream`; True = the caller
    also queued hgx_pair_classes for both levels (the per-pair form).  For HLA-like loci (two levels) the exon-level classes
    come from hgx_level_classes (pairs grouped by ref list first, HGX_NO_SIG=1 selects the per-pair form) and only the
    gene-level rows are materialised per pair, beside the exon-level EM.  `pc_events` = (before, after) events recorded
    around that gene-level hgx_pair_classes launch (bench.py).

    `heavy_lock` (held by the caller on entry, released here): with several samples in flight per GPU, the bandwidth-bound
    front of the path (scoring, the exon-level dedup) of one sample should not run beside another sample's -- it would only
    share the HBM -- but beside the other samples' EM phases, which are chains of short launches.  The lock is released as soon
    as this sample's exon-level classes exist."""
    hla = pl.base_fname == "hla"
    A, names = pl.n_alleles, pl.names
    db = dbatch if dbatch is not None else engine.DeviceBatch(batch, stream)
    bufs = bufs if bufs is not None else engine.ScoreBuffers(pl, db, exon=hla)
    by_list = hla and scored is not True and not os.environ.get("HGX_NO_SIG")
    if scored is False:
        if by_list:
            engine.piece_compat(pl, db, bufs, stream)
        else:
            engine.score_pairs(pl, db, bufs, stream)
    elif scored == "compat" and not by_list:
        if pc_events:
            pc_events[0].record(stream)
        engine.pair_classes(pl, db, bufs, stream)
        if pc_events:
            pc_events[1].record(stream)
    # ---- Gene_counts (core:1187-1190, 1650-1651) --------------------------------------------------
    # The gene-level side (dedup -> counts -> ranking) is independent of the exon-level EM until the hand-off, and the
    # EM is a long chain of short launches that leaves most of the GPU idle: for HLA the two run concurrently, the gene
    # side on a worker thread with its own non-blocking stream.
    gene = {}

    def gene_side(st):
        if by_list:
            if pc_events:
                pc_events[0].record(st)
            engine.pair_classes(pl, db, bufs, st, exon=False)
            if pc_events:
                pc_events[1].record(st)
        gcl_ = engine.Classes.dedup(bufs.gene_bits, db.n_pairs, pl.a_pad, hashes=bufs.gene_hash, stream=st)
        cnt, first = gcl_.allele_counts(st)
        fr = np.zeros(gcl_.n_classes, np.int64)                     # first pair of every class
        capi.check(capi.lib().hgx_classes_to_host(gcl_.h, None, None, capi.ptr(fr)))
        cnt_a, first_a = cnt[:A], first[:A]
        counted = np.nonzero(cnt_a > 0)[0]
        # dict insertion order of Gene_counts = (first pair that counted the allele, Gene_names order); then the
        # reference's stable descending sort on the count (core:1650-1651)
        ins = fr[first_a[counted]]
        counted = counted[np.lexsort((counted, ins, -cnt_a[counted]))]
        gene.update(gcl=gcl_, counted=counted, cnt=cnt_a)

    if overlap is None:
        overlap = stream is None
    overlap = hla and overlap and db.n_pairs >= 4096
    pair_groups = None
    if overlap:
        # scoring is complete before either side reads its output: a device-side dependency, the host keeps running ahead
        ev = _fork_event()
        ev.record(stream)
        em_stream_, gene_stream_ = capi.get_stream(0), capi.get_stream(1)     # this host thread's pair of side streams
        ev.make_wait(gene_stream_)
    state = {"worker": None}

    def start_gene():
        """Run the gene side: beside the caller on a worker thread + its own stream when overlapping, otherwise right here."""
        if not overlap:
            gene_side(stream)
            return
        dev = capi.current_device()

        def run():
            capi.set_device(dev)
            gene_side(gene_stream_)

        # a fresh thread per sample: measured faster than a pooled executor when several samples are in flight
        from concurrent.futures import Future
        fut = state["worker"] = Future()

        def run_t():
            try:
                fut.set_result(run())
            except BaseException as e:      # re-raised on the calling thread by worker.result()
                fut.set_exception(e)
        threading.Thread(target=run_t).start()

    if overlap and by_list:
        # grouping the pairs by exon-level ref list does not read the piece bitsets: queued on the EM stream right away it runs
        # BESIDE hgx_piece_compat (queued above on `stream`); the stream is ordered behind the scoring only further down
        pair_groups = engine.Groups(db, 0, em_stream_)
    start_gene()
    if overlap:
        if pair_groups is not None:
            pair_groups.n_groups                     # host wait for the grouping alone: the stream is not yet behind the scoring
        ev.make_wait(em_stream_)

    def finish_gene():
        if state["worker"] is not None:
            state["worker"].result()                 # re-raises on this thread
        res._names, res.counts_order, res.counts = names, gene["counted"], gene["cnt"]
        if keep_classes:
            res.gene_classes = gene["gcl"].to_host()[:2]
        return gene["gcl"]

    def run_em(classes, low, lengths):
        t0 = time.perf_counter()
        out, n_iter = _em_on_classes(classes, A, pl.name_rank, low, lengths, stream)
        res.t_em += time.perf_counter() - t0
        res.em.append({"n_classes": classes.n_classes, "remove_low": bool(low), "use_length": lengths is not None,
                       "result": [[names[a], p] for a, p in out], "n_iter": n_iter})
        return out

    if hla:
        em_stream = em_stream_ if overlap else stream
        if by_list:
            # the per-pair gene-level rows run beside the exon-level grouping (queueing them behind it, under the EM, was
            # measured slower: 2.62 vs 2.37 ms -- the EM's short launches then wait for wave slots)
            ecl = engine.Classes.of_level(pl, db, bufs, 0, em_stream, pair_groups)
        else:
            ecl = engine.Classes.dedup(bufs.exon_bits, db.n_pairs, pl.a_pad, hashes=bufs.exon_hash, stream=em_stream)
        if heavy_lock is not None:
            heavy_lock.release()
            heavy_lock = None
        if keep_classes:
            res.exon_classes = ecl.to_host()[:2]
        stream_saved, stream = stream, em_stream
        exon_prob = gene_prob = run_em(ecl, remove_low, None)                    # core:1732-1737
        stream = stream_saved
        gcl = finish_gene()
        groups = pl.rep_groups()
        exon_alleles, psum = set(), 0.0
        for i, (a, p) in enumerate(exon_prob):                                   # core:1739-1749
            if i >= 10 and p < 0.03:
                break
            g = groups.get(a, [a])
            if len(g) <= 1:
                continue
            psum += p
            exon_alleles |= set(g)
        if exon_alleles:                                                         # core:1752-1782
            mask = np.zeros(pl.w64, np.uint64)
            ea = np.fromiter(exon_alleles, np.int64, len(exon_alleles))
            np.bitwise_or.at(mask, ea >> 6, np.uint64(1) << (ea & 63).astype(np.uint64))
            # Gene_cmpt2 (gene classes filtered to exon_alleles, merged) and EM #2 in one call (hgx_em_masked)
            t0 = time.perf_counter()
            gcl.set_allele_rank(pl.name_rank)
            prob2, first2, n_iter2, n_cls2 = gcl.em_masked(mask, A, True, pl.allele_len, stream)
            res.t_em += time.perf_counter() - t0
            present = np.nonzero(prob2 >= 0.0)[0]
            order2 = present[np.lexsort((np.asarray(pl.name_rank)[present], first2[present]))].tolist()
            gp = _sorted_result(prob2, order2)
            res.em.append({"n_classes": n_cls2, "remove_low": True, "use_length": True,
                           "result": [[names[a], p] for a, p in gp], "n_iter": n_iter2})
            comb = {}
            for a, p in exon_prob:
                if a not in exon_alleles:
                    comb[a] = p
            for a, p in gp:
                comb[a] = p * psum
            gene_prob = _stable_desc([[a, p] for a, p in comb.items()])
        ecl.close()
        if pair_groups is not None:
            pair_groups.close()
    else:
        gcl = finish_gene()
        if heavy_lock is not None:
            heavy_lock.release()
            heavy_lock = None
        if gcl.n_classes <= 1:                                                   # core:1784-1787 (quirk Q3)
            if gcl.n_classes == 1:
                raise TypeError("'dict_keys' object is not subscriptable (reference quirk Q3, typing_core.py:1787)")
            gene_prob = []
        else:
            gene_prob = run_em(gcl, False, None)
    res.gene_prob = [[names[a], p] for a, p in gene_prob]
    gcl.close()
    return res


def report_lines(res, simulation=False, true_alleles=(), output_allele_counts=False, best_alleles=False):
    """Report body (core:1593, 1650-1677, 2076-2121)."""
    out = ["\t\t\t%d reads and %d pairs are aligned" % (res.num_reads, res.num_pairs)]
    for i, (a, c) in enumerate(res.counts_sorted):
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
            for test_Gene_names in locus_list:
                gene = test_Gene_names[0].split("*")[0] if simulation else test_Gene_names
                pl = PackedLocus.from_reference_dicts(gene, base_fname, refGenes, Genes, Gene_names, Gene_lengths,
                                                      refGene_loci, Vars, Var_list, Links)
                # samtools view F [chr:l-r] ref_allele (core:436-444): the view is ALWAYS restricted to this gene's backbone
                # sequence -- a multi-locus alignment (hla graph: A, B, C, ... in one BAM) never leaks other genes' reads
                # into this gene's decode -- and in genotype-genome mode to the locus span on the chromosome as well
                regions, base_locus = [refGenes[gene]], 0
                if genotype_genome != "":
                    _, chr_, left, right = refGene_loci[gene][:4]
                    regions, base_locus = ["%s:%d-%d" % (chr_, left + 1, right + 1), refGenes[gene]], left
                # ... piped through sort -k1,1 -s (core:458-468), and the loop's decode: all inside libhgx
                res = type_locus(pl, None, num_editdist=num_editdist, error_correction=error_correction,
                                 allow_discordant=allow_discordant,
                                 remove_low_abundance_alleles=remove_low_abundance_alleles, simulation=simulation,
                                 base_locus=base_locus, alignment_file=alignment_fname, regions=regions)
                pl.close()
                if res.num_reads <= 0:
                    continue
                lines, success = report_lines(res, simulation, test_Gene_names if simulation else (),
                                              output_allele_counts, best_alleles)
                for l in lines:
                    say(l)
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
