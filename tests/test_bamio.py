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


def _native_vs_python(path, region=None, n_threads=0):
    from hisatgenotype_amd.typing import read_alignment_text
    a = read_alignment_text(path, region, n_threads=n_threads, native=True)
    b = read_alignment_text(path, region, native=False)
    assert a == b
    return a


def test_native_reader_equals_python_reader(tmp_path):
    """hgx_read_alignments (parallel BGZF inflate + BAM decode + stable name sort in C++) against the pure-Python statement
    of the formats: byte-identical record streams for BAM (every fixture with its tags, shuffled record order, a region) and
    for SAM text (headers dropped, CRLF-free), with one thread and with many."""
    import random
    from hisatgenotype_amd import capi
    for name in ("hla_insertions", "hla_errors_filters", "codis_d18s51", "hla_single_end"):
        fx = gu.load(name)
        loc = fx["_locus"]
        lines = [l for l in fx["sam"].split("\n") if l]
        random.Random(5).shuffle(lines)                    # coordinate-order-like input: the reader has to group by name
        bam = str(tmp_path / (name + ".bam"))
        bamio.write_bam(bam, "\n".join(lines) + "\n", [(loc.ref_allele, len(loc.backbone))], block_size=3000)   # many BGZF blocks
        t1 = _native_vs_python(bam, n_threads=1)
        assert t1 == _native_vs_python(bam, n_threads=8)
        assert t1.count(b"\n") == len(lines)
        reg = "%s:%d-%d" % (loc.ref_allele, 200, 900)
        sub = _native_vs_python(bam, reg)
        assert 0 < sub.count(b"\n") < len(lines)
        sam = str(tmp_path / (name + ".sam"))
        with open(sam, "w") as f:
            f.write("@HD\tVN:1.0\n@SQ\tSN:%s\tLN:%d\n" % (loc.ref_allele, len(loc.backbone)) + "\n".join(lines) + "\n")
        assert _native_vs_python(sam).count(b"\n") == len(lines)
    # stability: records of one name keep their file order
    fx = gu.load("hla_small_pair")
    lines = [l for l in fx["sam"].split("\n") if l]
    sam = str(tmp_path / "stable.sam")
    open(sam, "w").write("\n".join(reversed(lines)) + "\n")
    got = _native_vs_python(sam).decode().split("\n")[:-1]
    by_name = {}
    for l in reversed(lines):
        by_name.setdefault(l.split("\t")[0], []).append(l)
    exp = [l for n in sorted(by_name, key=lambda s: s.encode()) for l in by_name[n]]
    assert got == exp


def test_native_reader_tags_and_errors(tmp_path):
    """Every tag type of the BAM specification through both readers; corrupt and truncated files are reported."""
    import pytest
    from hisatgenotype_amd import capi
    from hisatgenotype_amd.typing import read_alignment_text
    # hand-built record with all tag types (the writer only emits the types HISAT2 uses)
    refs = [("chr1", 1000)]
    rec = bytearray()
    qname = b"r1\0"
    tags = (b"XAAQ" + b"Xcc" + struct.pack("<b", -5) + b"XCC" + struct.pack("<B", 200) + b"Xss" + struct.pack("<h", -300) +
            b"XSS" + struct.pack("<H", 60000) + b"Xii" + struct.pack("<i", -70000) + b"XII" + struct.pack("<I", 4000000000) +
            b"Xff" + struct.pack("<f", 0.1) + b"XZZhello\0" + b"XHH1AE3\0" +
            b"XBBs" + struct.pack("<I", 3) + struct.pack("<3h", 1, -2, 3) + b"XGBf" + struct.pack("<I", 2) + struct.pack("<2f", 1.5, 0.1))
    cigar = struct.pack("<2I", (4 << 4) | 0, (2 << 4) | 4)
    seq = bytes([(1 << 4) | 2, (4 << 4) | 8, (15 << 4) | 1])
    qual = bytes([30, 31, 32, 33, 34, 35])
    body = struct.pack("<iiBBHHHiiii", 0, 99, len(qname), 37, 4680, 2, 99, 6, 0, 199, 150) + qname + cigar + seq + qual + tags
    out = bytearray(b"BAM\x01") + struct.pack("<i", 0) + struct.pack("<i", 1)
    out += struct.pack("<i", 5) + b"chr1\0" + struct.pack("<i", 1000)
    out += struct.pack("<i", len(body)) + body
    path = str(tmp_path / "tags.bam")
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    cdata = comp.compress(bytes(out)) + comp.flush()
    with open(path, "wb") as f:
        f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cdata) + 25) + cdata +
                struct.pack("<II", zlib.crc32(bytes(out)) & 0xffffffff, len(out)) + bamio._BGZF_EOF)
    text = _native_vs_python(path).decode()
    f = text.rstrip("\n").split("\t")
    assert f[:11] == ["r1", "99", "chr1", "100", "37", "4M2S", "=", "200", "150", "ACGTNA", "?@ABCD"]
    assert f[11:] == ["XA:A:Q", "Xc:i:-5", "XC:i:200", "Xs:i:-300", "XS:i:60000", "Xi:i:-70000", "XI:i:4000000000", "Xf:f:0.1",
                      "XZ:Z:hello", "XH:H:1AE3", "XB:B:s,1,-2,3", "XG:B:f,1.5,0.1"]
    raw = bytearray(open(path, "rb").read())
    bad = bytearray(raw)
    bad[40] ^= 0xFF                                       # payload byte: inflate or CRC must notice
    open(str(tmp_path / "bad.bam"), "wb").write(bytes(bad))
    with pytest.raises(capi.HgxError):
        read_alignment_text(str(tmp_path / "bad.bam"))
    open(str(tmp_path / "cut.bam"), "wb").write(bytes(raw[:60]))
    with pytest.raises(capi.HgxError):
        read_alignment_text(str(tmp_path / "cut.bam"))
    with pytest.raises(capi.HgxError):
        read_alignment_text(str(tmp_path / "missing.bam"))


def test_parse_alignment_file_equals_text_path(tmp_path):
    """hgx_parse_alignment_file (file -> piece batch inside libhgx) gives exactly the batch of read_alignment_text +
    parse_sam, from BAM and from SAM text, shuffled record order included."""
    import random
    import numpy as np
    from hisatgenotype_amd import locus as hl
    from hisatgenotype_amd.typing import read_alignment_text
    for name in ("hla_errors_filters", "codis_d18s51"):
        fx = gu.load(name)
        loc = fx["_locus"]
        pl = hl.PackedLocus.from_synth(loc)
        lines = [l for l in fx["sam"].split("\n") if l]
        random.Random(2).shuffle(lines)
        bam = str(tmp_path / (name + ".bam"))
        sam = str(tmp_path / (name + ".sam"))
        bamio.write_bam(bam, "\n".join(lines) + "\n", [(loc.ref_allele, len(loc.backbone))], block_size=5000)
        open(sam, "w").write("\n".join(lines) + "\n")
        sim = bool(fx.get("simulation", False))
        for path in (bam, sam):
            a = pl.parse_sam(read_alignment_text(path), simulation=sim)
            b = pl.parse_alignment_file(path, simulation=sim)
            assert (a.n_reads, a.n_pairs, a.n_pieces, a.n_refs) == (b.n_reads, b.n_pairs, b.n_pieces, b.n_refs)
            assert np.array_equal(a.pieces, b.pieces) and np.array_equal(a.masks, b.masks)
            assert np.array_equal(a.pair_off, b.pair_off) and np.array_equal(a.pair_ref, b.pair_ref)


def test_reader_edge_inputs(tmp_path):
    """Empty SAM, header-only SAM, header-only BAM -> empty record stream; plain gzip (not BGZF) is refused."""
    import gzip as gz
    import pytest
    from hisatgenotype_amd import capi
    from hisatgenotype_amd.typing import read_alignment_text
    p = str(tmp_path / "empty.sam")
    open(p, "w").write("")
    assert read_alignment_text(p) == b""
    p = str(tmp_path / "hdr.sam")
    open(p, "w").write("@HD\tVN:1.0\n@SQ\tSN:x\tLN:10\n")
    assert read_alignment_text(p) == b""
    p = str(tmp_path / "hdr.bam")
    bamio.write_bam(p, "", [("x", 10)])
    assert read_alignment_text(p) == b"" and read_alignment_text(p, native=False) in (b"", b"\n")
    p = str(tmp_path / "plain.sam.gz")
    with gz.open(p, "wb") as f:
        f.write(b"r1\t0\tx\t1\t0\t1M\t*\t0\t0\tA\tI\n")
    with pytest.raises(capi.HgxError):
        read_alignment_text(p)
