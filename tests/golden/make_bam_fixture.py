#!/usr/bin/env python3
"""A BAM file assembled BY HAND from the SAM/BAM specification (sections 4.1 BGZF, 4.2 BAM), byte for byte with struct.pack,
WITHOUT this package's writer or reader -- the independent producer the readers are pinned to.

What is in it, each with the SAM text `samtools view` prints for it written out literally:
  * three reference sequences, one with ':' and '*' in its name
  * nine records: a plain pair; every tag type (A c C s S i I f Z H and B arrays of all seven item types); an unmapped mate
    placed at its mate (FLAG 69/137, '*' CIGAR, SEQ present); a record without SEQ/QUAL ('*', l_seq 0); a secondary record
    with missing qualities (0xFF); a long-CIGAR record (placeholder 12S21N + CG:B:I tag, spec 4.2.2) that must come out with
    its real CIGAR and without the CG tag; a record on the second reference with RNEXT on another reference; an unplaced
    unmapped read (refID -1, POS 0); an odd-length sequence with all 16 base codes
  * BGZF: five members -- one holding only the first 30 bytes of the header, members cut in the MIDDLE of records (block
    boundaries fall inside a record's core fields and inside a tag), one STORED (uncompressed deflate) member, an EMPTY member
    in the middle of the file, and the 28-byte EOF marker at the end
Output: tests/golden/bam_handmade.json  (hex of the file + the expected lines + the byte offsets of the members).
Run: python tests/golden/make_bam_fixture.py"""
import json
import os
import struct
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))

REFS = [("chr6", 170805979), ("HLA:A*BACKBONE", 3569), ("contig_3", 500)]
SEQ_CODE = "=ACMGRSVTWYHKDBN"
CIG_CODE = "MIDNSHP=X"


def cigar(*ops):
    return [(n << 4) | CIG_CODE.index(c) for n, c in ops]


def seq4(seq):
    codes = [SEQ_CODE.index(c) for c in seq] + [0]
    return bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(seq), 2))


def reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def record(qname, flag, ref_id, pos0, mapq, cig, next_ref, next_pos0, tlen, seq, qual, tags=b"", span=None):
    """One alignment record (spec 4.2), block_size first.  qual: bytes of phred values, None = missing (0xFF)."""
    if span is None:
        span = sum(v >> 4 for v in cig if (v & 15) in (0, 2, 3, 7, 8)) or 1
    l_seq = len(seq)
    body = struct.pack("<iiBBHHHiiii", ref_id, pos0, len(qname) + 1, mapq, reg2bin(pos0, pos0 + span) if pos0 >= 0 else 4680,
                       len(cig), flag, l_seq, next_ref, next_pos0, tlen)
    body += qname.encode() + b"\0"
    body += b"".join(struct.pack("<I", v) for v in cig)
    body += seq4(seq)
    body += (bytes([0xFF]) * l_seq) if qual is None else bytes(qual)
    body += tags
    return struct.pack("<i", len(body)) + body


def tag(name, typ, payload):
    return name.encode() + typ.encode() + payload


def barr(name, sub, fmt, vals):
    return name.encode() + b"B" + sub.encode() + struct.pack("<I", len(vals)) + b"".join(struct.pack("<" + fmt, v) for v in vals)


def bgzf_member(raw, stored=False):
    """One BGZF member (spec 4.1): gzip header with the BC extra subfield, raw deflate data, CRC32, ISIZE."""
    if stored:                     # a deflate stream made of one final STORED block: 01 LEN NLEN data
        data = b"\x01" + struct.pack("<HH", len(raw), len(raw) ^ 0xFFFF) + raw
    else:
        c = zlib.compressobj(9, zlib.DEFLATED, -15)
        data = c.compress(raw) + c.flush()
    bsize = 12 + 6 + len(data) + 8 - 1
    assert bsize < 65536
    return (b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize) + data +
            struct.pack("<II", zlib.crc32(raw) & 0xFFFFFFFF, len(raw)))


def main():
    text = "@HD\tVN:1.6\tSO:unsorted\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in REFS)
    head = b"BAM\x01" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(REFS))
    for name, ln in REFS:
        head += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln)
    q30 = [30] * 10
    all_tags = (tag("XA", "A", b"Q") + tag("Xc", "c", struct.pack("<b", -128)) + tag("XC", "C", struct.pack("<B", 255)) +
                tag("Xs", "s", struct.pack("<h", -32768)) + tag("XS", "S", struct.pack("<H", 65535)) +
                tag("Xi", "i", struct.pack("<i", -2147483648)) + tag("XI", "I", struct.pack("<I", 4294967295)) +
                tag("Xf", "f", struct.pack("<f", 3.5)) + tag("XZ", "Z", b"two words\0") + tag("XH", "H", b"1AE301\0") +
                barr("Bc", "c", "b", [-1, 0, 1]) + barr("BC", "C", "B", [0, 200]) + barr("Bs", "s", "h", [-300, 300]) +
                barr("BS", "S", "H", [60000]) + barr("Bi", "i", "i", [-70000, 70000]) + barr("BI", "I", "I", [4000000000]) +
                barr("Bf", "f", "f", [1.5, -0.25]) + barr("Be", "C", "B", []))
    real = cigar((3, "M"), (1, "I"), (2, "M"), (1, "D"), (4, "M"), (10, "N"), (2, "M"), (3, "S"))      # reads 3+1+2+4+2+3 = 15? see below
    long_seq = "ACGTACGTACGTACG"                                                                          # 15 bases
    assert sum(v >> 4 for v in real if (v & 15) in (0, 1, 4, 7, 8)) == len(long_seq)
    real_span = sum(v >> 4 for v in real if (v & 15) in (0, 2, 3, 7, 8))
    recs = [
        record("pair1", 99, 1, 99, 60, cigar((10, "M")), 1, 299, 210, "ACGTACGTAC", q30, tag("NM", "C", b"\0") + tag("MD", "Z", b"10\0")),
        record("pair1", 147, 1, 299, 60, cigar((4, "S"), (6, "M")), 1, 99, -210, "TTTTGGGGCC", [0, 1, 2, 3, 40, 41, 42, 60, 92, 93],
               tag("NM", "C", b"\1") + tag("Zs", "Z", b"2|S|hv17\0")),
        record("tags", 0, 0, 29941259, 37, cigar((5, "M"), (2, "I"), (3, "M")), -1, -1, 0, "ACGTNACGTA", q30, all_tags),
        record("mateless", 73, 0, 1000, 20, cigar((10, "M")), 0, 1000, 0, "GGGGGGGGGG", q30),
        record("mateless", 133, 0, 1000, 0, [], 0, 1000, 0, "ACACACACAC", q30, span=1),
        record("noseq", 256, 2, 10, 0, cigar((20, "M")), -1, -1, 0, "", []),
        record("noqual", 272, 2, 50, 3, cigar((6, "M"), (100, "N"), (4, "M")), -1, -1, 0, "CATGCATGCA", None),
        record("longcigar", 0, 1, 500, 60, cigar((len(long_seq), "S"), (real_span, "N")), -1, -1, 0, long_seq, [20] * 15,
               tag("X0", "C", b"\7") + barr("CG", "I", "I", real) + tag("X1", "Z", b"after\0"), span=real_span),
        record("other", 97, 1, 0, 255, cigar((10, "M")), 2, 489, 0, "AAAAACCCCC", q30),
        record("unplaced", 4, -1, -1, 0, [], -1, -1, 0, "=ACMGRSVTWYHKDBNA", [10] * 17),
    ]
    stream = head + b"".join(recs)
    # member boundaries: inside the header; inside the core of record 1; inside the tags of "tags"; then the rest
    cut1 = 30
    cut2 = len(head) + len(recs[0]) + 20                                  # 20 bytes into the second record's fixed fields
    cut3 = len(head) + sum(len(r) for r in recs[:2]) + 36 + 5 + 8 + 12 + 40          # inside the tag block of "tags"
    cut4 = len(head) + sum(len(r) for r in recs[:7]) + 7                 # inside "longcigar"'s fixed fields
    assert cut1 < cut2 < cut3 < cut4 < len(stream)
    parts = [stream[:cut1], stream[cut1:cut2], stream[cut2:cut3], b"", stream[cut3:cut4], stream[cut4:]]
    members = [bgzf_member(p, stored=(k == 2)) for k, p in enumerate(parts)]
    eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    data = b"".join(members) + eof
    expected = [
        "pair1\t99\tHLA:A*BACKBONE\t100\t60\t10M\t=\t300\t210\tACGTACGTAC\t??????????\tNM:i:0\tMD:Z:10",
        "pair1\t147\tHLA:A*BACKBONE\t300\t60\t4S6M\t=\t100\t-210\tTTTTGGGGCC\t!\"#$IJK]}~\tNM:i:1\tZs:Z:2|S|hv17",
        "tags\t0\tchr6\t29941260\t37\t5M2I3M\t*\t0\t0\tACGTNACGTA\t??????????\tXA:A:Q\tXc:i:-128\tXC:i:255\tXs:i:-32768\tXS:i:65535\t"
        "Xi:i:-2147483648\tXI:i:4294967295\tXf:f:3.5\tXZ:Z:two words\tXH:H:1AE301\tBc:B:c,-1,0,1\tBC:B:C,0,200\tBs:B:s,-300,300\t"
        "BS:B:S,60000\tBi:B:i,-70000,70000\tBI:B:I,4000000000\tBf:B:f,1.5,-0.25\tBe:B:C",
        "mateless\t73\tchr6\t1001\t20\t10M\t=\t1001\t0\tGGGGGGGGGG\t??????????",
        "mateless\t133\tchr6\t1001\t0\t*\t=\t1001\t0\tACACACACAC\t??????????",
        "noseq\t256\tcontig_3\t11\t0\t20M\t*\t0\t0\t*\t*",
        "noqual\t272\tcontig_3\t51\t3\t6M100N4M\t*\t0\t0\tCATGCATGCA\t*",
        "longcigar\t0\tHLA:A*BACKBONE\t501\t60\t3M1I2M1D4M10N2M3S\t*\t0\t0\tACGTACGTACGTACG\t555555555555555\tX0:i:7\tX1:Z:after",
        "other\t97\tHLA:A*BACKBONE\t1\t255\t10M\tcontig_3\t490\t0\tAAAAACCCCC\t??????????",
        "unplaced\t4\t*\t0\t0\t*\t*\t0\t0\t=ACMGRSVTWYHKDBNA\t+++++++++++++++++",
    ]
    out = {"hex": data.hex(), "expected": expected, "refs": REFS,
           "member_offsets": [sum(len(m) for m in members[:k]) for k in range(len(members) + 1)],
           "note": "hand-assembled per SAMv1 sections 4.1 / 4.2 by tests/golden/make_bam_fixture.py; not written by this package"}
    with open(os.path.join(HERE, "bam_handmade.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("%d bytes, %d members (+EOF), %d records" % (len(data), len(members), len(recs)))


if __name__ == "__main__":
    main()
