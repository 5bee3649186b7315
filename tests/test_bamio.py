"""BAM ingestion (8f-3): BGZF/BAM decode round trips with the writer, on the fixtures' SAM (every tag the path reads)."""
import gzip
import os
import struct
import zlib

import numpy as np

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
    sub = bamio.read_bam(path, ["%s:1001-2001" % loc.ref_allele])      # 1-based inclusive = 0-based [1000, 2000]
    assert sub
    for l in sub:                                                       # every kept record OVERLAPS the span ...
        f = l.split("\t")
        pos0 = int(f[3]) - 1
        assert pos0 <= 2000 and pos0 + max(bamio.cigar_reflen(f[5]), 1) - 1 >= 1000
    kept = set(sub)                                                     # ... and every dropped one does not
    for l in bamio.read_bam(path):
        if l not in kept:
            f = l.split("\t")
            pos0 = int(f[3]) - 1
            assert pos0 > 2000 or pos0 + max(bamio.cigar_reflen(f[5]), 1) - 1 < 1000


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


def _rec(qname, flag, rname, pos1, cigar, seq_len=10):
    return "\t".join([qname, str(flag), rname, str(pos1), "60", cigar, "*", "0", "0", "A" * seq_len, "I" * seq_len, "NM:i:0"])


def test_region_semantics_follow_samtools_view(tmp_path):
    """`samtools view F r1 r2` (typing_core.py:436-444): records that OVERLAP a region, region after region; a read that
    starts before `left` but reaches into the span is kept (the POS-only rule dropped it); deletions and skips
    extend the span, insertions / clips do not; unmapped mates count as one base; a reference whose NAME contains ':' is
    matched as a whole before any "name:span" reading.  Native reader == Python reader on BAM and on SAM text."""
    from hisatgenotype_amd.typing import read_alignment_text
    refs = [("A*BACKBONE", 5000), ("HLA:B*07", 5000), ("chr6", 100000)]
    recs = [
        _rec("r01", 0, "chr6", 991, "10M"),            # [990, 999] ends exactly at left0 - 1 of chr6:1001-2000 -> out
        _rec("r02", 0, "chr6", 992, "10M"),            # [991, 1000] touches left0 = 1000 -> in
        _rec("r03", 0, "chr6", 985, "5M10D5M"),        # deletion extends the span: [984, 1003] -> in
        _rec("r04", 0, "chr6", 985, "5M10I5M", 20),    # insertion does not: [984, 993] -> out
        _rec("r05", 0, "chr6", 2000, "10M"),           # starts at right0 = 1999 -> in
        _rec("r06", 0, "chr6", 2001, "10M"),           # starts past the span -> out
        _rec("r07", 4, "chr6", 1500, "*"),             # unmapped, placed at its mate: one base -> in
        _rec("r08", 0, "chr6", 900, "5S10M200N10M", 25),   # spliced over the left edge -> in
        _rec("r09", 0, "A*BACKBONE", 100, "10M"),
        _rec("r10", 0, "HLA:B*07", 100, "10M"),
        _rec("r11", 0, "HLA:B*07", 3000, "10M"),
        _rec("r00", 0, "A*BACKBONE", 4000, "10M"),
    ]
    text = "\n".join(recs) + "\n"
    bam, sam = str(tmp_path / "m.bam"), str(tmp_path / "m.sam")
    bamio.write_bam(bam, text, refs, block_size=300)
    open(sam, "w").write("".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs) + text)

    def names(path, regions):
        a = read_alignment_text(path, regions, native=True)
        assert a == read_alignment_text(path, regions, native=False)
        return [l.split(b"\t")[0].decode() for l in a.split(b"\n") if l]

    for path in (bam, sam):
        assert names(path, ["chr6:1001-2000"]) == ["r02", "r03", "r05", "r07", "r08"]
        assert names(path, ["chr6:1,001-2,000"]) == ["r02", "r03", "r05", "r07", "r08"]
        assert names(path, ["chr6"]) == ["r01", "r02", "r03", "r04", "r05", "r06", "r07", "r08"]
        assert names(path, ["chr6:2001"]) == ["r05", "r06"]                 # open end
        assert names(path, ["chr6:-1000"]) == ["r01", "r02", "r03", "r04", "r08"]  # open start ([0, 999])
        assert names(path, ["A*BACKBONE"]) == ["r00", "r09"]                # name-sorted after the filter
        assert names(path, ["HLA:B*07"]) == ["r10", "r11"]                  # ':' inside a name: whole reference
        assert names(path, ["HLA:B*07:2000-4000"]) == ["r11"]
        assert names(path, ["chr6:1001-2000", "A*BACKBONE"]) == ["r00", "r02", "r03", "r05", "r07", "r08", "r09"]
        assert names(path, ["nosuch", "nosuch:1-10"]) == []
        assert names(path, None) == sorted(r.split("\t")[0] for r in recs)
    # region after region BEFORE the stable name sort: same-named records keep region order, then file order
    dup = "\n".join([_rec("q", 0, "chr6", 1500, "10M"), _rec("q", 16, "A*BACKBONE", 7, "10M"), _rec("q", 0, "chr6", 1400, "10M")]) + "\n"
    bamio.write_bam(bam, dup, refs)
    for native in (True, False):
        got = read_alignment_text(bam, ["A*BACKBONE", "chr6:1-5000"], native=native).decode().split("\n")[:-1]
        assert [(l.split("\t")[2], l.split("\t")[3]) for l in got] == [("A*BACKBONE", "7"), ("chr6", "1500"), ("chr6", "1400")]


def test_multi_locus_alignment_is_restricted_to_the_gene_backbone(tmp_path):
    """typing() must read only the records on this gene's backbone (`alignview_cmd += [ref_allele]`,
    typing_core.py:443-444).  Two genes in one file: the front-end of gene A sees exactly gene A's records, from BAM and
    from SAM text, through parse_alignment_file and through the text path."""
    import numpy as np
    from hisatgenotype_amd import locus as hl
    fx = gu.load("hla_small_pair")
    loc = fx["_locus"]
    pl = hl.PackedLocus.from_synth(loc)
    assert pl.ref_allele == loc.ref_allele
    own = [l for l in fx["sam"].split("\n") if l]
    other_name = "B*BACKBONE"
    other = []
    for l in own:                                         # the same reads, renamed, on another gene's backbone
        f = l.split("\t")
        f[0] = "o" + f[0]
        f[2] = other_name
        other.append("\t".join(f))
    mixed = [x for pair in zip(own, other) for x in pair]
    bam, sam = str(tmp_path / "two.bam"), str(tmp_path / "two.sam")
    refs = [(loc.ref_allele, len(loc.backbone)), (other_name, len(loc.backbone))]
    bamio.write_bam(bam, "\n".join(mixed) + "\n", refs, block_size=4000)
    open(sam, "w").write("\n".join(mixed) + "\n")
    sim = bool(fx.get("simulation", False))
    ref = pl.parse_sam("\n".join(own) + "\n", simulation=sim)
    for path in (bam, sam):
        got = pl.parse_alignment_file(path, [pl.ref_allele], simulation=sim)
        assert (got.n_reads, got.n_pairs, got.n_pieces, got.n_refs) == (ref.n_reads, ref.n_pairs, ref.n_pieces, ref.n_refs)
        assert np.array_equal(got.pieces, ref.pieces) and np.array_equal(got.pair_ref, ref.pair_ref)
        both = pl.parse_alignment_file(path, None, simulation=sim)          # unfiltered: the other gene's reads leak in
        assert both.n_reads == 2 * ref.n_reads


def test_native_bam_writer_round_trip(tmp_path):
    """hgx_write_bam (parallel encoder + BGZF deflate) against the two readers: what it writes decodes -- through the Python
    reader and the native one -- to the records it was given (all fixture tag types, '*' fields), the coordinate-sorted form
    is ordered by (reference, position) with file order kept among equal keys, and the EOF block is in place."""
    refs = [("A*BACKBONE", 5000), ("chr6", 100000)]
    recs = [_rec("r3", 99, "chr6", 500, "10M"), _rec("r1", 0, "A*BACKBONE", 40, "4S6M", 10), _rec("r2", 4, "chr6", 100, "*"),
            "rX\t77\t*\t0\t0\t*\t*\t0\t0\t*\t*\tXA:A:Q\tXf:f:0.5\tXH:H:1AE3\tXB:B:s,1,-2,3\tXI:i:4000000000\tXs:i:-300\tXc:i:-5",
            _rec("r0", 16, "chr6", 100, "5M200N5M")]
    text = "\n".join(recs) + "\n"
    for srt in (False, True):
        path = str(tmp_path / ("w%d.bam" % srt))
        bamio.write_bam_native(path, "@HD\tVN:1.0\n" + text, refs, sort_by_coordinate=srt, n_threads=3)
        raw = open(path, "rb").read()
        assert raw.endswith(bamio._BGZF_EOF)
        got = bamio.read_bam(path)
        exp = recs if not srt else [recs[1], recs[2], recs[4], recs[0], recs[3]]
        assert len(got) == len(exp)
        for g, e in zip(got, exp):
            gf, ef = g.split("\t"), e.split("\t")
            assert gf[:11] == ef[:11] and gf[11:] == ef[11:]
        from hisatgenotype_amd.typing import read_alignment_text
        assert read_alignment_text(path, native=True) == read_alignment_text(path, native=False)
    # a fixture with thousands of records and several BGZF blocks: identical to what the Python writer's file decodes to
    fx = gu.load("hla_mid_real")
    loc = fx["_locus"]
    a, b = str(tmp_path / "py.bam"), str(tmp_path / "nat.bam")
    bamio.write_bam(a, fx["sam"], [(loc.ref_allele, len(loc.backbone))])
    bamio.write_bam_native(b, fx["sam"], [(loc.ref_allele, len(loc.backbone))], n_threads=4)
    assert bamio.read_bam(a) == bamio.read_bam(b)


def test_readers_on_a_hand_assembled_bam(tmp_path):
    """The BAM readers against bytes this package did not write (tests/golden/bam_handmade.json: assembled with struct.pack
    straight from the SAM/BAM specification): several BGZF members whose boundaries fall inside records, a stored member, an
    empty member mid-file and the EOF marker, three references, every tag type and B-array item type, unmapped and unplaced
    reads, '*' SEQ / QUAL, missing qualities, a CG-tag long CIGAR.  The decoded text must equal, LITERALLY, the lines
    `samtools view` prints for such records -- through the Python reader and the native one, unsorted, name-grouped, with
    regions, and a truncated copy must be refused."""
    import json
    import pytest
    from hisatgenotype_amd import capi
    from hisatgenotype_amd.typing import read_alignment_text
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bam_handmade.json")))
    path = str(tmp_path / "hand.bam")
    data = bytes.fromhex(fx["hex"])
    open(path, "wb").write(data)
    exp = fx["expected"]
    assert bamio.read_bam(path) == exp                                           # file order, literal text
    by_name = sorted(range(len(exp)), key=lambda i: exp[i].split("\t")[0].encode())   # stable: sort -k1,1 -s
    want = "".join(exp[i] + "\n" for i in by_name).encode()
    for nt in (1, 4):
        assert read_alignment_text(path, n_threads=nt, native=True) == want
    assert read_alignment_text(path, native=False) == want
    names = lambda regions: [l.split(b"\t")[0].decode() for l in read_alignment_text(path, regions).split(b"\n") if l]
    assert names(["HLA:A*BACKBONE"]) == ["longcigar", "other", "pair1", "pair1"]
    assert names(["HLA:A*BACKBONE:506-506"]) == ["longcigar"]                    # inside the long CIGAR's real span (501..523)
    assert names(["HLA:A*BACKBONE:524-600"]) == []
    assert names(["chr6:1001-1001"]) == ["mateless", "mateless"]                 # the unmapped mate counts as one base
    assert names(["contig_3:100-120"]) == ["noqual"]                             # inside its 100N skip
    assert names(["contig_3", "chr6:29941260-29941260"]) == ["noqual", "noseq", "tags"]
    for native in (True, False):
        assert read_alignment_text(path, ["HLA:A*BACKBONE:506-506"], native=native).decode().split("\t")[5] == "3M1I2M1D4M10N2M3S"
    # every member boundary really falls where the generator says (and inflating member by member gives the stream back)
    offs = fx["member_offsets"]
    assert data[offs[3]:offs[4]][-4:] == b"\0\0\0\0" and len(offs) == 7          # the empty member: ISIZE 0
    cut = str(tmp_path / "cut.bam")
    open(cut, "wb").write(data[:offs[5] - 9])
    with pytest.raises(capi.HgxError):
        read_alignment_text(cut)
    with pytest.raises(Exception):
        bamio.read_bam(cut)


def test_bam_record_chain_walked_in_ranges_is_exact(tmp_path, monkeypatch):
    """Big BAM files have their record chain discovered in parallel: every worker but the first guesses a record start in its
    byte range and the guesses are checked against the chain (hgx_bam.cpp).  (1) With the range walk forced on a fixture, the
    record stream is the same as with one thread.  (2) A decoy: a record whose B:C tag holds a run of bytes that look like a
    chain of valid records, placed over the range boundary -- the worker there synchronises on the decoy, the check notices,
    and the result is still exactly the real records."""
    import struct
    from hisatgenotype_amd.typing import read_alignment_text
    from hisatgenotype_amd import engine
    engine.test_switch("bam_chain_min", "0")
    fx = gu.load("hla_mid_real")
    loc = fx["_locus"]
    path = str(tmp_path / "mid.bam")
    bamio.write_bam_native(path, fx["sam"], [(loc.ref_allele, len(loc.backbone))], sort_by_coordinate=True)
    one = read_alignment_text(path, n_threads=1)
    for nt in (2, 3, 7, 16):
        assert read_alignment_text(path, n_threads=nt) == one
        assert read_alignment_text(path, [loc.ref_allele + ":1000-1400"], n_threads=nt) == read_alignment_text(path, [loc.ref_allele + ":1000-1400"], n_threads=1)
    assert one == read_alignment_text(path, native=False)

    def rec(qname, pos0, tags=b""):
        body = struct.pack("<iiBBHHHiiii", 0, pos0, len(qname) + 1, 60, 4681, 1, 0, 4, -1, -1, 0) + qname.encode() + b"\0"
        body += struct.pack("<I", (4 << 4) | 0) + bytes([0x12, 0x48]) + bytes([30, 30, 30, 30]) + tags
        return struct.pack("<i", len(body)) + body
    fake = rec("decoy", 7)                                    # a perfectly plausible record ... as payload bytes of a tag
    decoy_tag = b"XBBC" + struct.pack("<I", 120 * len(fake)) + fake * 120
    recs = [rec("a%d" % k, 10 + k) for k in range(3)] + [rec("big", 50, decoy_tag)] + [rec("z%d" % k, 90 + k) for k in range(3)]
    head = b"BAM\x01" + struct.pack("<i", 0) + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chr1\0" + struct.pack("<i", 1000)
    stream = head + b"".join(recs)
    out = b""
    for o in range(0, len(stream), 3000):                     # several BGZF members
        raw = stream[o:o + 3000]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        d = c.compress(raw) + c.flush()
        out += (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(d) + 25) + d +
                struct.pack("<II", zlib.crc32(raw) & 0xffffffff, len(raw)))
    dpath = str(tmp_path / "decoy.bam")
    open(dpath, "wb").write(out + bamio._BGZF_EOF)
    want = read_alignment_text(dpath, native=False)
    assert want.count(b"\n") == 7 and b"decoy\t" not in want
    for nt in (1, 2, 3, 4):
        assert read_alignment_text(dpath, n_threads=nt) == want


def test_big_sam_text_is_scanned_while_it_is_read(tmp_path):
    """A SAM file above 8 MB is read in 1 MB pieces per worker and scanned for lines piece by piece (hgx_bam.cpp); the line that runs
    over the end of a worker's byte range is finished afterwards.  Header lines, CRLF line ends, no final newline, a region
    list and 1 ... 32 workers: always the batch of the plain text path."""
    from hisatgenotype_amd import synth, locus as hl
    loc = synth.make_hla_like_locus(n_alleles=300, n_vars=200, seed=9)
    pl = hl.PackedLocus.from_synth(loc)
    sam = synth.simulate_sam_fast(loc, synth.pick_sample(loc, 9), 14000, err_rate=0.003, seed=5)
    lines = ["@HD\tVN:1.0", "@SQ\tSN:x\tLN:10"] + [l + ("\r" if i % 7 == 0 else "") for i, l in enumerate(sam.split("\n")) if l]
    txt = "\n".join(lines)                                     # no final newline
    assert len(txt) > (8 << 20)
    path = str(tmp_path / "big.sam")
    with open(path, "w") as f:
        f.write(txt)
    ref = pl.parse_sam(sam, n_threads=4)
    for nt in (1, 2, 3, 5, 8, 13, 32):
        for regions in (None, [loc.ref_allele]):
            b = pl.parse_alignment_file(path, regions, n_threads=nt)
            assert (b.n_pairs, b.n_reads) == (ref.n_pairs, ref.n_reads), (nt, regions)
            assert np.array_equal(np.asarray(b.pair_ref), np.asarray(ref.pair_ref))
            assert np.array_equal(np.asarray(b.masks), np.asarray(ref.masks))


def test_bgzf_inflate_with_libdeflate_and_with_zlib(tmp_path, monkeypatch):
    """The BGZF blocks are inflated by libdeflate when the system has it (dlopen) and by zlib otherwise (HGX_NO_LIBDEFLATE=1 forces
    it): same records either way, and a block with a damaged payload is refused by both (CRC-32 / ISIZE check)."""
    import pytest
    import sys
    from hisatgenotype_amd import capi
    read = sys.modules["hisatgenotype_amd.typing"].read_alignment_text
    fx = gu.load("hla_small_pair")
    data = fx["sam"].encode()
    path = str(tmp_path / "x.bam")
    bamio.write_bam_native(path, data, [(fx["_locus"].ref_allele, len(fx["_locus"].backbone))], sort_by_coordinate=True)
    texts = []
    for force_zlib in (False, True):
        if force_zlib:
            monkeypatch.setenv("HGX_NO_LIBDEFLATE", "1")
        texts.append(read(path))
    assert texts[0] == texts[1] and texts[0].count(b"\n") == data.count(b"\n")
    raw = bytearray(open(path, "rb").read())
    raw[40] ^= 0x55                                             # inside the first block's compressed payload
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(bytes(raw))
    for force_zlib in (True, False):
        if not force_zlib:
            monkeypatch.delenv("HGX_NO_LIBDEFLATE")
        with pytest.raises(capi.HgxError):
            read(bad)
