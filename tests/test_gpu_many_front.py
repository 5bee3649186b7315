"""hgx_many_create_files / hgx_many_create_sams: the samples of one locus through ONE pass of the device front end (every record
carries its task; a pileup per task; one shared piece table) against hgx_many_create of the host front end's per-task batches --
the merged batch must be the same array for array, the per-task extents equal, and hgx_type_many's results identical."""
import os
import random
import sys

import numpy as np
import pytest

import golden_util as gu
from hisatgenotype_amd import bamio, capi, engine, locus as hl, synth

htyping = sys.modules["hisatgenotype_amd.typing"]
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def same_merged(a, b):
    assert (a.n_reads, a.n_pairs, a.n_pieces, a.n_refs, a.n_mask_u32) == (b.n_reads, b.n_pairs, b.n_pieces, b.n_refs, b.n_mask_u32)
    assert a.pieces.tobytes() == b.pieces.tobytes()
    assert a.masks.tobytes() == b.masks.tobytes()
    assert a.pair_off.tobytes() == b.pair_off.tobytes()
    assert a.pair_ref.tobytes() == b.pair_ref.tobytes()


def same_many(dev, host):
    assert dev.n_tasks == host.n_tasks
    assert (dev.n_pieces, dev.n_pairs, dev.n_refs, dev.n_reads) == (host.n_pieces, host.n_pairs, host.n_refs, host.n_reads)
    assert dev.pair_base == host.pair_base
    assert dev.task_reads == host.task_reads and dev.task_pieces == host.task_pieces and dev.task_refs == host.task_refs
    same_merged(dev.merged(), host.merged())


def same_results(pl, dev, host):
    for em_fast in (False, None):
        for g, h in zip(htyping.type_many(pl, dev, em_fast=em_fast), htyping.type_many(pl, host, em_fast=em_fast)):
            assert g.num_reads == h.num_reads and g.num_pairs == h.num_pairs
            if g.num_reads == 0:
                continue
            assert np.array_equal(g.counts_order, h.counts_order) and np.array_equal(g.counts, h.counts)
            assert [e["n_iter"] for e in g.em] == [e["n_iter"] for e in h.em]
            assert [e["result"] for e in g.em] == [e["result"] for e in h.em]
            assert g.gene_prob == h.gene_prob


def _samples(loc, n, pairs, seed, err=0.004):
    rng = random.Random(seed)
    return [synth.simulate_sam_fast(loc, synth.pick_sample(loc, rng.randrange(1 << 30)), pairs + 53 * t, err_rate=err, seed=rng.randrange(1 << 30))
            for t in range(n)]


def test_many_samples_in_one_device_pass(tmp_path):
    """Samples with different alleles (the same read text decodes differently under each sample's own pileup), an empty task, the
    same sample twice; SAM texts in memory, SAM files, BAM files (name-grouped and coordinate-sorted with regions)."""
    loc = synth.make_hla_like_locus(n_alleles=300, n_vars=700, seed=9, deletion_frac=0.12)
    pl = hl.PackedLocus.from_synth(loc)
    sams = _samples(loc, 5, 900, 21, err=0.01)
    sams = sams[:2] + [""] + sams[2:] + [sams[0]]
    host = engine.ManyBatch(pl, [pl.parse_sam(s) for s in sams])
    with engine.test_switches(front="device"):
        dev = engine.ManyBatch.from_sams(pl, sams)
        assert engine.front_last() == (2, 0)
    same_many(dev, host)
    same_results(pl, dev, host)
    p_sam, p_bam = [], []
    for t, s in enumerate(sams):
        p_sam.append(str(tmp_path / ("t%d.sam" % t)))
        open(p_sam[-1], "w").write(s)
        p_bam.append(str(tmp_path / ("t%d.bam" % t)))
        bamio.write_bam_native(p_bam[-1], s.encode(), [(loc.ref_allele, len(loc.backbone))], sort_by_coordinate=(t % 2 == 1))
    sent = {}
    for paths, sw in ((p_sam, "device"), (p_bam, "device"), (p_bam, "device,host_inflate")):
        # (BAM: the files travel deflated and one launch inflates every task's BGZF blocks -- or the host's threads inflate)
        with engine.test_switches(front=sw):
            dev = engine.ManyBatch.from_files(pl, paths, regions=[loc.ref_allele] * len(paths))
            assert engine.front_last() == (2, 0)
            sent[(paths is p_bam, sw)] = engine.front_last_bytes()
        same_many(dev, host)
    assert sent[(True, "device")] < sent[(True, "device,host_inflate")] / 2
    with engine.test_switches(front="device,late"):                   # the buffer estimate was too small: sent again after the reads
        dev = engine.ManyBatch.from_files(pl, p_bam, regions=[loc.ref_allele] * len(p_bam))
        assert engine.front_last() == (2, 0)
    same_many(dev, host)
    with engine.test_switches(front="device"):                                        # on a stream of the caller's
        dev = engine.ManyBatch.from_files(pl, p_bam, regions=[loc.ref_allele] * len(p_bam), stream=capi.get_stream(1))
        assert engine.front_last() == (2, 0)
    same_many(dev, host)
    # where the device front end declines, the host front end runs task by task: the same batch again
    for switches, want in ((dict(front="host"), (0, -1)), (dict(), (2, 0))):          # (the library's choice at 9 000 records: the kernels)
        with engine.test_switches(**switches):
            dev = engine.ManyBatch.from_files(pl, p_bam, regions=[loc.ref_allele] * len(p_bam))
            assert engine.front_last() == want, engine.front_last()
        same_many(dev, host)
    # below the size gate (1 000 records / 300 KB in all: the measured break-even, hgx_front.hip FE_MIN_*) the host stages take the call
    small = _samples(loc, 2, 120, 77, err=0.01)
    p_small = []
    for t, s_ in enumerate(small):
        p_small.append(str(tmp_path / ("small%d.bam" % t)))
        bamio.write_bam_native(p_small[-1], s_.encode(), [(loc.ref_allele, len(loc.backbone))])
    dev = engine.ManyBatch.from_files(pl, p_small, regions=[loc.ref_allele] * 2)
    assert engine.front_last() == (0, 6), engine.front_last()                         # (6: fewer records than the size gate)
    same_many(dev, engine.ManyBatch(pl, [pl.parse_sam(s_) for s_ in small]))
    with engine.test_switches(front="device"):
        dev = engine.ManyBatch.from_files(pl, [p_sam[0], p_bam[1]])                  # SAM text and BAM records in one batch
        assert engine.front_last() == (0, 1)
    same_many(dev, engine.ManyBatch(pl, [pl.parse_sam(sams[0]), pl.parse_sam(sams[1])]))
    with pytest.raises(capi.HgxError):
        engine.ManyBatch.from_files(pl, [p_bam[0], str(tmp_path / "missing.bam")])


@pytest.mark.parametrize("name", [n for n in gu.ALL + gu.LEAN if n != "codis_d18s51"])
def test_many_front_on_fixtures(name):
    fx = gu.load(name)
    o = fx["options"]
    pl = hl.PackedLocus.from_synth(fx["_locus"])
    kw = dict(num_editdist=o["num_editdist"], error_correction=o["error_correction"], allow_discordant=o["allow_discordant"],
              simulation=o["simulation"])
    lines = fx["sam"].splitlines(keepends=True)
    cut = len(lines) // 2
    while 0 < cut < len(lines) and lines[cut].split("\t")[0] == lines[cut - 1].split("\t")[0]:
        cut += 1
    sams = [fx["sam"], "".join(lines[:cut]), fx["sam"]]
    host = engine.ManyBatch(pl, [pl.parse_sam(s, **kw) for s in sams])
    with engine.test_switches(front="device"):
        dev = engine.ManyBatch.from_sams(pl, sams, **kw)
        assert engine.front_last() == (2, 0), engine.front_last()
    same_many(dev, host)


def test_many_front_at_panel_size(tmp_path):
    """24 samples x 3000 pairs of a 2500-allele locus from BAM files: above the size gate without a switch; results identical."""
    loc = synth.make_hla_like_locus(n_alleles=2500, n_vars=1500, seed=77)
    pl = hl.PackedLocus.from_synth(loc)
    sams = _samples(loc, 24, 3000, 5)
    paths = []
    for t, s in enumerate(sams):
        paths.append(str(tmp_path / ("s%d.bam" % t)))
        bamio.write_bam_native(paths[-1], s.encode(), [(loc.ref_allele, len(loc.backbone))])
    host = engine.ManyBatch(pl, [pl.parse_sam(s) for s in sams])
    dev = engine.ManyBatch.from_files(pl, paths)
    assert engine.front_last() == (2, 0), engine.front_last()
    same_many(dev, host)
    same_results(pl, dev, host)


def test_many_front_on_fuzz_cases():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity
    n_dev = n_all = 0
    for k in range(60):
        loc, sam0, single = fuzz_parity.make_case(990000, k, 1 + k % 3)
        sams = [sam0] + [fuzz_parity.make_case(990000, k, 1 + (k + j + 1) % 3)[1] for j in range(1 + k % 3)]
        pl = hl.PackedLocus.from_synth(loc)
        try:
            host = engine.ManyBatch(pl, [pl.parse_sam(s, allow_discordant=single) for s in sams])
        except capi.HgxError:
            with pytest.raises(capi.HgxError):
                with engine.test_switches(front="device"):
                    engine.ManyBatch.from_sams(pl, sams, allow_discordant=single)
            continue
        with engine.test_switches(front="device"):
            dev = engine.ManyBatch.from_sams(pl, sams, allow_discordant=single)
            route, _ = engine.front_last()
        n_all += 1
        n_dev += route == 2
        same_many(dev, host)
    assert n_dev >= 0.7 * n_all, (n_dev, n_all)
