"""BAM ingestion (8f-3): BGZF/BAM decode round trips with the writer, on the fixtures' SAM (every tag the path reads)."""
import gzip
import os
import struct
import zlib

import golden_util as gu
from hisatgenotype_amd import bamio


def test_bam_round_trip(tmp_path):
    fx = gu.load("hla_insertions")
    loc = fx["_locus"]
    path = str(tmp_path / "x.bam")
    bamio.write_bam(path, fx["sam"], [(loc.ref_allele, len(loc.backbone))])
    raw = open(path, "rb").read()
    assert raw[:4] == b"\x1f\x8b\x08\x04" and raw.endswith(bamio._BGZF_EOF)
    # a BGZF file is a series of gzip members: a generic gzip reader must inflate it to "BAM\1..."
    assert gzip.decompress(raw)[:4] == b"BAM\x01"
    got = bamio.read_bam(path)
    exp = [l for l in fx["sam"].split("\n") if l]
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        gf, ef = g.split("\t"), e.split("\t")
        assert gf[:11] == ef[:11]
        assert sorted(gf[11:]) == sorted(ef[11:])


def test_bam_region_filter_and_sorting(tmp_path):
    from hisatgenotype_amd.typing import read_alignment_text
    fx = gu.load("hla_small_pair")
    loc = fx["_locus"]
    path = str(tmp_path / "y.bam")
    lines = [l for l in fx["sam"].split("\n") if l]
    bamio.write_bam(path, "\n".join(reversed(lines)) + "\n", [(loc.ref_allele, len(loc.backbone))])
    text = read_alignment_text(path).decode().split("\n")
    names = [l.split("\t")[0] for l in text if l]
    assert names == sorted(names)                        # sort -k1,1 -s equivalent
    sub = bamio.read_bam(path, (loc.ref_allele, 1000, 2000))
    assert sub and all(1000 <= int(l.split("\t")[3]) - 1 <= 2000 for l in sub)


def test_corrupt_block_is_detected(tmp_path):
    fx = gu.load("codis_like")
    loc = fx["_locus"]
    path = str(tmp_path / "z.bam")
    bamio.write_bam(path, fx["sam"], [(loc.ref_allele, len(loc.backbone))])
    raw = bytearray(open(path, "rb").read())
    raw[60] ^= 0xFF
    open(path, "wb").write(bytes(raw))
    try:
        bamio.read_bam(path)
    except (ValueError, zlib.error):
        return
    raise AssertionError("corruption not detected")
