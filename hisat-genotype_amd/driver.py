"""Callers of the hot path: a `genotyping_locus` with the reference's signature, and a panel runner
that shards independent (sample, locus) tasks over the GPUs of a node.

`genotyping_locus` mirrors typing_core.genotyping_locus (typing_core.py:2278-2691): stand-alone and genotype-genome
indexes (read with hisatgenotype_amd.indexio), real reads / existing alignments, and the simulation self-test loop.
Downloading/building databases and the assembly graph are not on this path; HISAT2 is used when it is installed
(simulate.align_reads).
"""
import os
import random
import sys
from copy import deepcopy
from datetime import datetime

from . import indexio
from .locus import PackedLocus
from .typing import read_alignment_text, type_locus, typing  # noqa: F401


def genotyping_locus(base_fname, locus_list, genotype_genome, ix_dir, only_locus_list, partial, aligners, read_fname, fastq,
                     alignment_fname, threads, simulate_interval, read_len, fragment_len, best_alleles, num_editdist,
                     perbase_errorrate, perbase_snprate, skip_fragment_regions, assembly, output_base, error_correction,
                     keep_alignment, discordant, type_primary_exons, remove_low_abundance_alleles, display_alleles, verbose,
                     assembly_verbose, out_dir, output_allele_counts, debug_instr):
    """Same 32 parameters as the reference (typing_core.py:2278-2309).

    * index: `<ix_dir>/<base_fname>.*` (stand-alone) or, with `genotype_genome`, `<ix_dir>/<genotype_genome>.*` restricted to
      the family `base_fname` (typing_core.py:2326-2397) -- read with hisatgenotype_amd.indexio; cloning / downloading /
      building databases and HISAT2 indexes (typing_core.py:2402-2414) need the network and are not done here;
    * real reads / an existing alignment: one typing() call over the locus list (typing_core.py:2650-2691);
    * no reads and no alignment: the simulation self-test loop (typing_core.py:2488-2648) -- `test_size` tests (default
      200) of one allele ("basic") or a sorted allele pair (`debug_instr["pair"]`) per locus drawn with
      `random.seed(set_seed); random.sample(...)`, reads from simulate.simulate_reads, typing(simulation=True) per test,
      "Passed so far" after every test and the totals at the end, on stderr like the reference.  Returns test_passed in that
      mode (the reference returns None; the totals are also printed)."""
    assert isinstance(base_fname, str) and "," not in base_fname
    assert os.path.exists(ix_dir)
    simulation = (read_fname == [] and alignment_fname == "")
    full_gg_path = ix_dir + "/" + (genotype_genome if genotype_genome else base_fname)
    ix = indexio.load_index_memo(ix_dir, base_fname, genotype_genome)      # (read once per process and index-file state)
    refGene_loci, Gene_names, partial_alleles = ix["refGene_loci"], ix["Gene_names"], ix["partial_alleles"]
    if len(locus_list) == 0:
        locus_list = list(refGene_loci.keys())

    def run_typing(sim, loci, reads, num_frag_list, test_i=0):
        return typing(sim, full_gg_path, loci, genotype_genome, partial, partial_alleles, ix["refGenes"], ix["Genes"],
                      Gene_names, ix["Gene_lengths"], refGene_loci, ix["Vars"], ix["Var_list"], ix["Links"], aligners,
                      num_editdist, assembly, output_base, error_correction, keep_alignment, discordant, type_primary_exons,
                      remove_low_abundance_alleles, display_alleles, False if sim else fastq, reads, alignment_fname,
                      num_frag_list, read_len, fragment_len, threads, best_alleles, verbose, assembly_verbose, out_dir,
                      ix["dbversion"], output_allele_counts, test_i)

    if not simulation:
        print("\t", locus_list if base_fname == "genome" else " ".join(locus_list), file=sys.stderr)
        return run_typing(False, locus_list, read_fname, [])

    # ---- simulation self-test (typing_core.py:2488-2648) -------------------------------------------------------------
    from . import simulate
    basic_test, pair_test, test_size, ranseed = True, False, 200, None
    test_passed, test_list = {}, []
    if debug_instr:
        if "pair" in debug_instr:
            basic_test, pair_test = False, True
        if "test_size" in debug_instr:
            test_size = int(debug_instr["test_size"])
        if "set_seed" in debug_instr:
            ranseed = debug_instr["set_seed"]
        if "test_list" in debug_instr:
            test_list = [[debug_instr["test_list"].split("-")]]
    # (the reference builds this list from a set intersection, typing_core.py:2508: its order -- and with it the order of
    # the loci inside every test -- changes with the interpreter's hash seed; here it is the order of locus_list)
    genes = [g for g in dict.fromkeys(locus_list) if g in Gene_names]
    allele_count = 2 if pair_test else 1
    if not test_list:
        test_list = [[] for _ in range(test_size)]
        for gene in genes:
            candidates = deepcopy(Gene_names[gene])
            candidates.remove(gene + "*BACKBONE")
            random.seed(ranseed)
            picks = random.sample(range(len(candidates)), test_size * allele_count)
            for k in range(0, len(picks), allele_count):
                first, last = candidates[picks[k]], candidates[picks[k + allele_count - 1]]
                test_list[k // allele_count].append([first] if basic_test else sorted([first, last]))
    for test_i, test_locus_list in enumerate(test_list):
        if "test_id" in debug_instr and str(test_i + 1) not in debug_instr["test_id"].split("-"):
            continue
        print("Test %d" % (test_i + 1), str(datetime.now()), file=sys.stderr)
        for names in test_locus_list:                       # load_index keeps names and lengths only: spell these alleles out
            for name in names:
                indexio.allele_sequence(ix, name.split("*")[0], name)
        num_frag_list = simulate.simulate_reads(ix["Genes"], base_fname, test_locus_list, ix["Vars"], ix["Links"],
                                                simulate_interval, read_len, fragment_len, perbase_errorrate, perbase_snprate,
                                                skip_fragment_regions, out_dir, test_i)
        assert len(num_frag_list) == len(test_locus_list)
        for names, frags in zip(test_locus_list, num_frag_list):
            assert len(frags) == len(names)
            for name, n_frag in zip(names, frags):
                print("\t%s - %d bp (%s sequence, %d pairs)" % (
                    name, len(ix["Genes"][name.split("*")[0]][name]), "partial" if name in partial_alleles else "full", n_frag),
                    file=sys.stderr)
        reads = ["%s_input_1.fa" % base_fname] if "single-end" in debug_instr else \
            ["%s_input_1.fa" % base_fname, "%s_input_2.fa" % base_fname]
        got = run_typing(True, test_locus_list, reads, num_frag_list, test_i)
        aligner_type = None
        for aligner_type, passed in got.items():
            test_passed[aligner_type] = test_passed.get(aligner_type, 0) + passed
        expected = (test_i + 1) * allele_count * len(genes)
        if aligner_type is not None:
            print("\t\tPassed so far: %d/%d (%.2f%%)" % (test_passed[aligner_type], expected,
                                                         test_passed[aligner_type] * 100.0 / expected), file=sys.stderr)
        else:
            print("\t\tTest Failed!", file=sys.stderr)
    total = len(test_list) * allele_count * len(genes)
    for aligner_type, passed in test_passed.items():
        print("%s:\t%d/%d passed (%.2f%%)" % (aligner_type, passed, total, passed * 100.0 / total), file=sys.stderr)
    return test_passed


def run_panel(tasks, index, base_fname, rank=0, world=1, weights=None, ix_dir=None, inflight=1, many=False, em_fast=False, **typing_opts):
    """Type independent (sample_id, gene, sam_text_or_path) tasks; rank `rank` of `world` handles its share
    (deterministic greedy split, no communication).  `index` is the dict from indexio.load_index; with `ix_dir` the
    packed loci come through the binary cache next to the index files (indexio.packed_locus).  `inflight` > 1 types that many
    tasks concurrently on this rank's GPU (host threads with their own streams): the EM of one sample is a chain of short
    launches that leaves the GPU to the front-end work and scoring of the next (bench.py --inflight).
    `many`: the rank's tasks of one locus are typed TOGETHER (hgx_type_many: one launch chain per locus instead of one per
    task -- the many-samples form, /root/reference/hisatgenotype:613-665); results are identical to the one-by-one form
    (`em_fast` False: the reference's order of floating-point operations; None: hgx_type_many's default, table-lookup arithmetic --
    ~3x faster per panel, abundances within 1e-8).
    Returns {(sample_id, gene): LocusResult} for this rank's tasks."""
    import threading
    from . import capi
    from . import dist as hdist
    mine = hdist.shard(list(tasks), rank, world, weights)
    packed = {}
    for _, gene, _ in mine:
        if gene in packed:
            continue
        if ix_dir is not None:
            packed[gene] = indexio.packed_locus(ix_dir, base_fname, gene, index)
        else:
            packed[gene] = PackedLocus.from_reference_dicts(gene, base_fname, index["refGenes"], index["Genes"],
                                                            index["Gene_names"], index["Gene_lengths"], index["refGene_loci"],
                                                            index["Vars"], index["Var_list"], index["Links"])
        packed[gene].index()                     # device index created once, before any worker needs it
    out = {}

    from . import engine
    heavy = engine.Gate() if inflight > 1 else None          # one bandwidth-bound front (scoring, exon dedup) at a time

    def one(task, stream):
        sample_id, gene, sam = task
        if isinstance(sam, (bytes, bytearray)) or "\t" in sam:
            return (sample_id, gene), type_locus(packed[gene], sam, stream=stream, gate=heavy, **typing_opts)
        # a SAM / BAM path: read, grouped and decoded inside libhgx
        # (the view is restricted to the gene's backbone, as the reference's `samtools view F ref_allele` does, core:443-444)
        opts = dict(typing_opts)
        opts.setdefault("regions", [packed[gene].ref_allele])
        return (sample_id, gene), type_locus(packed[gene], None, alignment_file=sam, stream=stream, gate=heavy, **opts)

    if many:
        from .typing import type_many
        by_gene = {}
        for task in mine:
            by_gene.setdefault(task[1], []).append(task)
        parse_keys = ("num_editdist", "error_correction", "allow_discordant", "simulation", "base_locus")
        popts = {k: v for k, v in typing_opts.items() if k in parse_keys}
        for gene, group in by_gene.items():
            pl = packed[gene]
            is_text = [isinstance(sam, (bytes, bytearray)) or "\t" in sam for _, _, sam in group]
            if all(is_text) or not any(is_text):
                # the samples of the locus through ONE pass of the device front end (hgx_many_create_sams / _files; where it
                # declines, the library runs the host front end per task and merges: the same batch either way)
                if all(is_text):
                    mb = engine.ManyBatch.from_sams(pl, [sam for _, _, sam in group], **popts)
                else:
                    regions = typing_opts.get("regions", [pl.ref_allele])
                    regions = regions if isinstance(regions, str) else "\n".join(regions)
                    mb = engine.ManyBatch.from_files(pl, [sam for _, _, sam in group], regions=[regions] * len(group), **popts)
            else:
                batches = []
                for (_, _, sam), text in zip(group, is_text):
                    if text:
                        batches.append(pl.parse_sam(sam, **popts))
                    else:
                        batches.append(pl.parse_alignment_file(sam, typing_opts.get("regions", [pl.ref_allele]), **popts))
                mb = engine.ManyBatch(pl, batches)
            try:
                res = type_many(pl, mb, remove_low=typing_opts.get("remove_low_abundance_alleles", True), em_fast=em_fast)
                pieces, refs = mb.task_pieces, mb.task_refs
            finally:
                mb.close()
            for t, ((sample_id, _, _), r) in enumerate(zip(group, res)):
                r.n_pieces, r.n_refs = pieces[t], refs[t]
                out[(sample_id, gene)] = r
        return out
    if inflight <= 1 or len(mine) <= 1:
        for task in mine:
            k, v = one(task, None)
            out[k] = v
    else:
        lock = threading.Lock()
        state = {"next": 0, "err": None}
        dev = capi.current_device()

        def work(slot):
            try:
                capi.set_device(dev)
                capi.set_stream_slot(slot)               # stable stream set per worker index across calls
                stream = capi.get_stream(2)              # this thread's main stream (its side streams are per thread too)
                while True:
                    with lock:
                        i = state["next"]
                        if i >= len(mine) or state["err"] is not None:
                            return
                        state["next"] = i + 1
                    k, v = one(mine[i], stream)
                    with lock:
                        out[k] = v
            except BaseException as e:                   # re-raised on the calling thread
                with lock:
                    state["err"] = e

        threads = [threading.Thread(target=work, args=(i,)) for i in range(min(inflight, len(mine)))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if state["err"] is not None:
            raise state["err"]
    for pl in packed.values():
        pl.close()
    return out
