"""BAM ingestion without samtools.

The reference pipes ``samtools view <bam> [region]`` through ``sort -k1,1 -s`` (typing_core.py:436-468) and walks the
text twice (typing_common.py:1059-1134 for the pileup, then the typing loop).  ``samtools`` is not available on the GPU
box, so this module decodes BGZF/BAM (SAM/BAM specification v1, sections 4.1 and 4.2) directly and yields the same SAM
text lines; `read_alignment_text` then name-groups them.  A minimal writer is included for tests and for turning the
synthetic SAM into BAM fixtures.

Only what the typing path consumes is decoded: the eleven mandatory fields and the tags of types A c C s S i I f Z H B.
"""
import gzip
import re
import struct
import zlib

_CIGAR_OPS = "MIDNSHP=X"
_SEQ = "=ACMGRSVTWYHKDBN"
_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _bgzf_blocks(data):
    """Yield the inflated payload of every BGZF block (concatenated gzip members with a BC extra field)."""
    off, n = 0, len(data)
    while off < n:
        if data[off:off + 4] != b"\x1f\x8b\x08\x04":
            raise ValueError("not a BGZF block at offset %d" % off)
        xlen = struct.unpack_from("<H", data, off + 10)[0]
        extra = data[off + 12:off + 12 + xlen]
        bsize, p = None, 0
        while p + 4 <= len(extra):
            si1, si2, slen = extra[p], extra[p + 1], struct.unpack_from("<H", extra, p + 2)[0]
            if si1 == 66 and si2 == 67 and slen == 2:
                bsize = struct.unpack_from("<H", extra, p + 4)[0]
            p += 4 + slen
        if bsize is None:
            raise ValueError("BGZF block without BC subfield")
        cdata = data[off + 12 + xlen:off + bsize + 1 - 8]
        crc, isize = struct.unpack_from("<II", data, off + bsize + 1 - 8)
        raw = zlib.decompress(cdata, -15) if cdata else b""
        if len(raw) != isize or (zlib.crc32(raw) & 0xffffffff) != crc:
            raise ValueError("corrupt BGZF block at offset %d" % off)
        yield raw
        off += bsize + 1


def _decode_tags(buf, p, end):
    out = []
    while p < end:
        tag = buf[p:p + 2].decode()
        t = chr(buf[p + 2])
        p += 3
        if t == "A":
            out.append("%s:A:%s" % (tag, chr(buf[p]))); p += 1
        elif t in "cCsSiI":
            fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}[t]
            v = struct.unpack_from(fmt, buf, p)[0]
            p += struct.calcsize(fmt)
            out.append("%s:i:%d" % (tag, v))
        elif t == "f":
            out.append("%s:f:%g" % (tag, struct.unpack_from("<f", buf, p)[0])); p += 4
        elif t in "ZH":
            e = buf.index(b"\0", p)
            out.append("%s:%s:%s" % (tag, t, buf[p:e].decode())); p = e + 1
        elif t == "B":
            st = chr(buf[p]); cnt = struct.unpack_from("<I", buf, p + 1)[0]
            fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[st]
            vals = struct.unpack_from("<%d%s" % (cnt, fmt), buf, p + 5)
            p += 5 + cnt * struct.calcsize(fmt)
            out.append("%s:B:%s%s" % (tag, st, "".join("," + (("%g" % v) if st == "f" else str(v)) for v in vals)))
        else:
            raise ValueError("unknown BAM tag type %r" % t)
    return out


def parse_region(text):
    """One samtools region string -> (whole, name, left0, right0): the record matches if RNAME == whole (entire reference)
    or RNAME == name and its span overlaps [left0, right0].  "name:l-r" is 1-based inclusive; "name:l" and "name:-r" are
    open-ended; commas may group digits.  name is None when the text after the last ':' is not a span."""
    name, left0, right0 = None, 0, 1 << 62
    if ":" in text[1:]:
        head, span = text.rsplit(":", 1)
        m = re.fullmatch(r"([0-9,]*)(-([0-9,]*))?", span)
        if m and span and (m.group(1) or m.group(3)):
            lo = m.group(1).replace(",", "")
            hi = (m.group(3) or "").replace(",", "")
            if (lo or hi) and not (m.group(1) and not lo) and not (m.group(3) and not hi):
                name = head
                left0 = max(int(lo) - 1, 0) if lo else 0
                right0 = int(hi) - 1 if hi else 1 << 62
    return text, name, left0, right0


def normalise_regions(regions):
    """None / "" / [] -> None (no filter); a string (regions separated by newlines) or a list of strings -> list of
    parse_region tuples."""
    if regions is None:
        return None
    if isinstance(regions, (str, bytes)):
        if isinstance(regions, bytes):
            regions = regions.decode()
        if regions == "":
            return None
        regions = regions.split("\n")
    return [parse_region(r) for r in regions if r]


def region_hit(reg, rname, pos0, end0):
    whole, name, left0, right0 = reg
    return rname == whole or (name is not None and rname == name and end0 >= left0 and pos0 <= right0)


def cigar_reflen(cigar):
    """Reference bases consumed by a SAM CIGAR string (M D N = X)."""
    return sum(int(n) for n, op in re.findall(r"(\d+)([MIDNSHP=X])", cigar) if op in "MDN=X")


def read_bam(path, regions=None):
    """Decode a BAM file into SAM text lines (no header).  `regions` = samtools region strings ("name", "name:left-right";
    a list, or one string with newlines): the records OVERLAPPING a region (reference span from the CIGAR, one base for
    unmapped records) come out region after region, like ``samtools view file r1 r2`` (typing_core.py:436-444)."""
    regs = normalise_regions(regions)
    with open(path, "rb") as f:
        data = f.read()
    raw = b"".join(_bgzf_blocks(data))
    if raw[:4] != b"BAM\x01":
        raise ValueError("not a BAM file")
    l_text = struct.unpack_from("<i", raw, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, p)[0]
    p += 4
    refs = []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", raw, p)[0]
        refs.append(raw[p + 4:p + 4 + l_name - 1].decode())
        p += 4 + l_name + 4
    lines = []
    per_region = [[] for _ in (regs or [])]
    n = len(raw)
    while p < n:
        bs = struct.unpack_from("<i", raw, p)[0]
        rec_end = p + 4 + bs
        ref_id, pos, l_rn, mapq, _bin, n_cig, flag, l_seq, nref_id, npos, tlen = struct.unpack_from("<iiBBHHHiiii", raw, p + 4)
        q = p + 36
        qname = raw[q:q + l_rn - 1].decode()
        q += l_rn
        cig = []
        for k in range(n_cig):
            v = struct.unpack_from("<I", raw, q + 4 * k)[0]
            cig.append("%d%s" % (v >> 4, _CIGAR_OPS[v & 15]))
        q += 4 * n_cig
        sb = raw[q:q + (l_seq + 1) // 2]
        seq = "".join(_SEQ[b >> 4] + _SEQ[b & 15] for b in sb)[:l_seq]
        q += (l_seq + 1) // 2
        qb = raw[q:q + l_seq]
        qual = "*" if (l_seq == 0 or qb[0] == 0xff) else "".join(chr(b + 33) for b in qb)
        q += l_seq
        tags = _decode_tags(raw, q, rec_end)
        # long CIGARs (spec 4.2.2): placeholder <l_seq>S<span>N in the CIGAR field, the operations in a CG:B:I tag -- read the
        # way htslib reads them (the tag becomes the CIGAR and leaves the tag list)
        if n_cig and ref_id >= 0 and pos >= 0 and cig[0] == "%dS" % l_seq:
            for k, tg in enumerate(tags):
                if tg.startswith("CG:B:") and tg[5] in "Ii":
                    vals = [int(x) for x in tg[7:].split(",")] if len(tg) > 7 else []
                    if len(vals) >= n_cig:
                        cig = ["%d%s" % (v >> 4, _CIGAR_OPS[v & 15]) for v in vals]
                        del tags[k]
                    break
                if tg.startswith("CG:"):
                    break
        p = rec_end
        rname = refs[ref_id] if ref_id >= 0 else "*"
        hits = None
        if regs is not None:
            reflen = 0 if (flag & 4) else sum(int(c[:-1]) for c in cig if c[-1] in "MDN=X")
            end0 = pos + max(reflen, 1) - 1
            hits = [g for g, r in enumerate(regs) if ref_id >= 0 and region_hit(r, rname, pos, end0)]
            if not hits:
                continue
        rnext = "*" if nref_id < 0 else ("=" if nref_id == ref_id else refs[nref_id])
        text = "\t".join([qname, str(flag), rname, str(pos + 1), str(mapq), "".join(cig) or "*", rnext, str(npos + 1),
                          str(tlen), seq or "*", qual] + tags)
        if hits is None:
            lines.append(text)
        else:
            for g in hits:
                per_region[g].append(text)
    if regs is not None:
        for v in per_region:
            lines.extend(v)
    return lines


def _reg2bin(beg, end):
    end -= 1
    for shift, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return off + (beg >> shift)
    return 0


def _encode_tag(field):
    tag, t, val = field.split(":", 2)
    if t == "i":
        v = int(val)
        for code, fmt, lo, hi in (("C", "<B", 0, 255), ("c", "<b", -128, 127), ("S", "<H", 0, 65535), ("s", "<h", -32768, 32767),
                                  ("I", "<I", 0, 2 ** 32 - 1), ("i", "<i", -2 ** 31, 2 ** 31 - 1)):
            if lo <= v <= hi:
                return tag.encode() + code.encode() + struct.pack(fmt, v)
        raise ValueError("integer tag out of range")
    if t == "Z":
        return tag.encode() + b"Z" + val.encode() + b"\0"
    if t == "A":
        return tag.encode() + b"A" + val.encode()
    if t == "f":
        return tag.encode() + b"f" + struct.pack("<f", float(val))
    raise ValueError("unsupported tag type %s" % t)


def write_bam(path, sam_text, refs, block_size=0xff00):
    """Write SAM text (records only) as BAM.  `refs` = [(name, length)]."""
    ref_id = {n: i for i, (n, _) in enumerate(refs)}
    out = bytearray(b"BAM\x01")
    header = "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs).encode()
    out += struct.pack("<i", len(header)) + header + struct.pack("<i", len(refs))
    for name, ln in refs:
        out += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln)
    import re
    cre = re.compile(r"(\d+)([MIDNSHP=X])")
    for line in sam_text.split("\n"):
        if not line or line.startswith("@"):
            continue
        f = line.split("\t")
        qname, flag, rname, pos, mapq, cigar, rnext, pnext, tlen, seq, qual = f[:11]
        ops = cre.findall(cigar) if cigar != "*" else []
        pos0 = int(pos) - 1
        ref_len = sum(int(n) for n, op in ops if op in "MDN=X")
        rid = ref_id.get(rname, -1)
        nid = rid if rnext == "=" else ref_id.get(rnext, -1)
        rec = bytearray()
        rec += struct.pack("<iiBBHHHiiii", rid, pos0, len(qname) + 1, int(mapq), _reg2bin(pos0, pos0 + max(ref_len, 1)), len(ops),
                           int(flag), 0 if seq == "*" else len(seq), nid, int(pnext) - 1, int(tlen))
        rec += qname.encode() + b"\0"
        for n, op in ops:
            rec += struct.pack("<I", (int(n) << 4) | _CIGAR_OPS.index(op))
        if seq != "*":
            codes = [_SEQ.index(c) if c in _SEQ else 15 for c in seq] + [0]
            rec += bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(seq), 2))
            rec += bytes([0xff] * len(seq)) if qual == "*" else bytes(ord(c) - 33 for c in qual)
        for field in f[11:]:
            rec += _encode_tag(field)
        out += struct.pack("<i", len(rec)) + rec
    with open(path, "wb") as fo:
        for i in range(0, len(out), block_size):
            raw = bytes(out[i:i + block_size])
            comp = zlib.compressobj(6, zlib.DEFLATED, -15)
            cdata = comp.compress(raw) + comp.flush()
            bsize = len(cdata) + 25
            fo.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + cdata +
                     struct.pack("<II", zlib.crc32(raw) & 0xffffffff, len(raw)))
        fo.write(_BGZF_EOF)


def write_bam_native(path, sam_text, refs, sort_by_coordinate=False, n_threads=0):
    """write_bam through libhgx (hgx_write_bam: parallel record encoding and BGZF deflate): seconds instead of minutes for a
    million records.  `sort_by_coordinate` orders the records like `samtools sort` first."""
    import ctypes as C
    import numpy as np
    from . import capi
    data = sam_text if isinstance(sam_text, (bytes, bytearray)) else sam_text.encode()
    names = "\n".join(n for n, _ in refs).encode()
    lens = np.array([ln for _, ln in refs] or [0], np.int32)
    capi.check(capi.lib().hgx_write_bam(path.encode(), data, C.c_size_t(len(data)), names, capi.ptr(lens), C.c_int32(len(refs)),
                                        C.c_int32(int(bool(sort_by_coordinate))), C.c_int32(n_threads)))
