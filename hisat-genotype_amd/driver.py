"""Callers of the hot path: a `genotyping_locus` with the reference's signature, and a panel runner
that shards independent (sample, locus) tasks over the GPUs of a node.

`genotyping_locus` mirrors typing_core.genotyping_locus (typing_core.py:2278-2691) for the part that leads to the
accelerated path: a stand-alone index (`<ix_dir>/<base_fname>.*`, read with hisatgenotype_amd.indexio) and existing
alignments.  Downloading/building databases, HISAT2 alignment, read simulation and the assembly graph are not on this
path and raise NotImplementedError with the reference line they correspond to.
"""
import os

from . import indexio
from .locus import PackedLocus
from .typing import read_alignment_text, type_locus, typing  # noqa: F401


def genotyping_locus(base_fname, locus_list, genotype_genome, ix_dir, only_locus_list, partial, aligners, read_fname, fastq,
                     alignment_fname, threads, simulate_interval, read_len, fragment_len, best_alleles, num_editdist,
                     perbase_errorrate, perbase_snprate, skip_fragment_regions, assembly, output_base, error_correction,
                     keep_alignment, discordant, type_primary_exons, remove_low_abundance_alleles, display_alleles, verbose,
                     assembly_verbose, out_dir, output_allele_counts, debug_instr):
    """Same 32 parameters as the reference (typing_core.py:2278-2309)."""
    assert isinstance(base_fname, str) and "," not in base_fname
    assert os.path.exists(ix_dir)
    simulation = (read_fname == [] and alignment_fname == "")
    if simulation:
        raise NotImplementedError("simulation self-test needs simulate_reads + HISAT2 (typing_core.py:2488-2648)")
    if genotype_genome:
        raise NotImplementedError("genotype-genome indexes need samtools faidx (typing_core.py:2175-2195)")
    if alignment_fname == "":
        raise NotImplementedError("read alignment needs HISAT2 (typing_common.py:985-1056): pass alignment_fname")
    ix = indexio.load_index(ix_dir, base_fname)
    if len(locus_list) == 0:
        locus_list = list(ix["refGene_loci"].keys())
    return typing(False, os.path.join(ix_dir, base_fname), locus_list, "", partial, ix["partial_alleles"], ix["refGenes"],
                  ix["Genes"], ix["Gene_names"], ix["Gene_lengths"], ix["refGene_loci"], ix["Vars"], ix["Var_list"],
                  ix["Links"], aligners, num_editdist, assembly, output_base, error_correction, keep_alignment, discordant,
                  type_primary_exons, remove_low_abundance_alleles, display_alleles, fastq, read_fname, alignment_fname, [],
                  read_len, fragment_len, threads, best_alleles, verbose, assembly_verbose, out_dir, ix["dbversion"],
                  output_allele_counts)


def run_panel(tasks, index, base_fname, rank=0, world=1, weights=None, ix_dir=None, inflight=1, **typing_opts):
    """Type independent (sample_id, gene, sam_text_or_path) tasks; rank `rank` of `world` handles its share
    (deterministic greedy split, no communication).  `index` is the dict from indexio.load_index; with `ix_dir` the
    packed loci come through the binary cache next to the index files (indexio.packed_locus).  `inflight` > 1 types that many
    tasks concurrently on this rank's GPU (host threads with their own streams): the EM of one sample is a chain of short
    launches that leaves the GPU to the front-end work and scoring of the next (bench.py --inflight).
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

    heavy = threading.Lock() if inflight > 1 else None       # one bandwidth-bound front (scoring, exon dedup) at a time

    def one(task, stream):
        sample_id, gene, sam = task
        if isinstance(sam, (bytes, bytearray)) or "\t" in sam:
            return (sample_id, gene), type_locus(packed[gene], sam, stream=stream, heavy_lock=heavy, **typing_opts)
        # a SAM / BAM path: read, grouped and decoded inside libhgx
        # (the view is restricted to the gene's backbone, as the reference's `samtools view F ref_allele` does, core:443-444)
        opts = dict(typing_opts)
        opts.setdefault("regions", [packed[gene].ref_allele])
        return (sample_id, gene), type_locus(packed[gene], None, alignment_file=sam, stream=stream, heavy_lock=heavy, **opts)

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
