"""The DEVICE front end (csrc/hgx_front.hip: pileup, per-key decode, piece table, pair protocol as kernels) against the pinned
host front end (csrc/hgx_sam.cpp): the batch born in HBM must be the host's batch byte for byte -- pieces, masks, pair offsets,
refs, read count, pileup counts and nt_sets -- on every fixture recorded from the real reference, on the fuzz cases of
tools/fuzz_parity.py, on a deep sample, from SAM text, SAM files and BAM files; and hgx_type_file (which now goes through it) must
give the results it gave through the host stages."""
import os
import random
import sys

import numpy as np
import pytest

import golden_util as gu
from hisatgenotype_amd import capi, engine, locus as hl, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def same_batch(a, b, length, pileup=True):
    assert (a.n_reads, a.n_pairs, a.n_pieces, a.n_refs, a.n_mask_u32) == (b.n_reads, b.n_pairs, b.n_pieces, b.n_refs, b.n_mask_u32)
    assert a.pieces.tobytes() == b.pieces.tobytes()
    assert a.masks.tobytes() == b.masks.tobytes()
    assert a.pair_off.tobytes() == b.pair_off.tobytes()
    assert a.pair_ref.tobytes() == b.pair_ref.tobytes()
    if pileup:
        na, ca = a.pileup(length)
        nb, cb = b.pileup(length)
        assert np.array_equal(na, nb) and np.array_equal(ca, cb)


@pytest.mark.parametrize("name", gu.ALL + gu.LEAN)
def test_device_front_end_equals_the_host_front_end_on_every_fixture(name):
    fx = gu.load(name)
    o = fx["options"]
    pl = hl.PackedLocus.from_synth(fx["_locus"])
    kw = dict(num_editdist=o["num_editdist"], error_correction=o["error_correction"], allow_discordant=o["allow_discordant"],
              simulation=o["simulation"])
    host = pl.parse_sam(fx["sam"], **kw)
    hd = engine.DeviceBatch(host)
    for switches, want in ((dict(front="device"), 2), (dict(front="device,keys"), 1)):
        with engine.test_switches(**switches):             # (the fixtures are smaller than the size gate)
            dev = pl.parse_sam_dev(fx["sam"], **kw)
            route, code = engine.front_last()
        if name == "codis_d18s51":
            assert (route, code) == (0, 1)                 # choose_pairs (typing_core.py:1547-1552) stays on the host
        else:
            assert (route, code) == (want, 0), (route, code)
        assert (dev.n_reads, dev.n_pairs, dev.n_pieces, dev.n_refs) == (host.n_reads, host.n_pairs, host.n_pieces, host.n_refs)
        same_batch(host, dev.to_host(), len(fx["_locus"].backbone), pileup=route > 0)
        assert (dev.sum_piece_words, dev.n_gene_refs) == (hd.sum_piece_words, hd.n_gene_refs)   # the byte-model inputs of the bench line


@pytest.mark.parametrize("name", gu.ALL)
def test_record_route_on_files(name, tmp_path):
    """SAM file, name-grouped BAM, coordinate-sorted BAM with regions: the device parses the records (BAM: binary CIGAR, packed SEQ)."""
    from hisatgenotype_amd import bamio
    fx = gu.load(name)
    o = fx["options"]
    loc = fx["_locus"]
    pl = hl.PackedLocus.from_synth(loc)
    kw = dict(num_editdist=o["num_editdist"], error_correction=o["error_correction"], allow_discordant=o["allow_discordant"],
              simulation=o["simulation"])
    host = pl.parse_sam(fx["sam"], **kw)
    p_sam, p_bam, p_sorted = str(tmp_path / "r.sam"), str(tmp_path / "r.bam"), str(tmp_path / "s.bam")
    open(p_sam, "w").write(fx["sam"])
    bamio.write_bam_native(p_bam, fx["sam"].encode(), [(loc.ref_allele, len(loc.backbone))])
    bamio.write_bam_native(p_sorted, fx["sam"].encode(), [(loc.ref_allele, len(loc.backbone))], sort_by_coordinate=True)
    for path in (p_sam, p_bam, p_sorted):
        with engine.test_switches(front="device"):
            dev = pl.parse_alignment_file_dev(path, regions=[loc.ref_allele], **kw)
            route, code = engine.front_last()
        assert (route, code) == ((0, 1) if name == "codis_d18s51" else (2, 0)), (path, route, code)
        same_batch(host, dev.to_host(), len(loc.backbone), pileup=route > 0)


def test_size_gate_and_switches():
    fx = gu.load("hla_small_pair")
    pl = hl.PackedLocus.from_synth(fx["_locus"])
    host = pl.parse_sam(fx["sam"], simulation=True)
    dev = pl.parse_sam_dev(fx["sam"], simulation=True)
    assert engine.front_last() == (0, 6)                   # a few hundred records: the host stages finish the job
    same_batch(host, dev.to_host(), len(fx["_locus"].backbone), pileup=False)
    with engine.test_switches(front="host"):
        dev = pl.parse_sam_dev(fx["sam"], simulation=True)
        assert engine.front_last() == (0, -1)
    same_batch(host, dev.to_host(), len(fx["_locus"].backbone), pileup=False)


def test_device_front_end_on_fuzz_cases():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity
    n_dev = n_all = 0
    why = {}
    for k in range(int(os.environ.get("HGX_FRONT_FUZZ", "150"))):
        loc, sam, single = fuzz_parity.make_case(880000, k, 1 + k % 3)
        pl = hl.PackedLocus.from_synth(loc)
        for ec in (True, False):
            try:
                host = pl.parse_sam(sam, error_correction=ec, allow_discordant=single)
            except capi.HgxError:
                with engine.test_switches(front="device"), pytest.raises(capi.HgxError):
                    pl.parse_sam_dev(sam, error_correction=ec, allow_discordant=single)
                continue
            for extra in ("device", "device,keys"):
                with engine.test_switches(front=extra):
                    dev = pl.parse_sam_dev(sam, error_correction=ec, allow_discordant=single)
                    ran, code = engine.front_last()
                n_all += 1
                n_dev += ran > 0
                why[(ran, code)] = why.get((ran, code), 0) + 1
                same_batch(host, dev.to_host(), len(loc.backbone), pileup=ran > 0)
        pl.close()
    print("device stages took %d of %d inputs; decline codes %s" % (n_dev, n_all, why))
    assert n_dev >= 0.8 * n_all, (n_dev, n_all, why)


@pytest.mark.parametrize("n_pairs,err", [(30000, 0.002), (120000, 0.01)])
def test_device_front_end_on_deep_samples_and_files(tmp_path, n_pairs, err):
    """Past the size gate (no switch): SAM text, SAM file, name-grouped BAM, coordinate-sorted BAM; then hgx_type_file == type_locus
    on the host batch."""
    import hisatgenotype_amd as hgx
    from hisatgenotype_amd import bamio
    loc = synth.make_hla_like_locus(n_alleles=1200, n_vars=1100, seed=77)
    sample = synth.pick_sample(loc, 5)
    sam = synth.simulate_sam_fast(loc, sample, n_pairs, err_rate=err, seed=3)
    pl = hl.PackedLocus.from_synth(loc)
    host = pl.parse_sam(sam)
    dev = pl.parse_sam_dev(sam)
    assert engine.front_last() == (2, 0)
    same_batch(host, dev.to_host(), len(loc.backbone))
    with engine.test_switches(front="keys"):
        dev = pl.parse_sam_dev(sam)
        assert engine.front_last() == (1, 0)
    same_batch(host, dev.to_host(), len(loc.backbone))
    p_sam = str(tmp_path / "r.sam")
    open(p_sam, "w").write(sam)
    p_bam, p_sorted = str(tmp_path / "r.bam"), str(tmp_path / "s.bam")
    bamio.write_bam_native(p_bam, sam.encode(), [(loc.ref_allele, len(loc.backbone))])
    bamio.write_bam_native(p_sorted, sam.encode(), [(loc.ref_allele, len(loc.backbone))], sort_by_coordinate=True)
    for path in (p_sam, p_bam, p_sorted):
        d2 = pl.parse_alignment_file_dev(path, regions=[loc.ref_allele])
        assert engine.front_last() == (2, 0), path
        same_batch(host, d2.to_host(), len(loc.backbone))
    ref = hgx.type_locus(pl, sam)
    for path in (p_sam, p_sorted):
        res = hgx.type_file(pl, path) if hasattr(hgx, "type_file") else None
        if res is None:
            break
        assert engine.front_last() == (2, 0)
        assert (res.num_reads, res.num_pairs) == (ref.num_reads, ref.num_pairs)
        assert res.counts_sorted == ref.counts_sorted and res.gene_prob == ref.gene_prob
        assert [e["n_iter"] for e in res.em] == [e["n_iter"] for e in ref.em]
