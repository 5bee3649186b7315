#!/usr/bin/env python3
"""Randomised parity of the device BGZF inflate (csrc/hgx_inflate.hip) with zlib: payloads of every texture (uniform bytes over
alphabets of 1 - 256 symbols, skewed alphabets that need 15-bit codes, runs, periodic data with periods up to 40 KB -- matches from
far beyond the 8 KB ring --, text, mixtures), deflated per block with a random level / strategy / memLevel and block sizes from 1
byte to 0xff00; every case one BGZF file of several blocks through hgx_bgzf_inflate.  Usage: tools/fuzz_inflate.py [n_cases] [seed]"""
import os, sys, random, struct, zlib, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hisatgenotype_amd import capi
if os.environ.get("INF_LIB"):                              # a variant build of the library (tools/inf_subpool_check.sh)
    capi.LIB_PATH = os.environ["INF_LIB"]
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 500
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1

def payload(rng):
    kind = rng.randrange(8)
    n = rng.choice([0, 1, 2, 7, 100, 1000, 5000, 70000, 200000, 400000])
    n = rng.randint(max(0, n // 2), n) if n else 0
    if kind == 0:
        k = rng.choice([1, 2, 3, 4, 16, 64, 256])
        return bytes(rng.randrange(k) for _ in range(min(n, 120000)))
    if kind == 1:                                            # skewed: a few frequent symbols, a long tail (long Huffman codes)
        pool = bytes([65] * 4000 + [66] * 1500 + [67] * 500 + list(range(256)))
        return bytes(rng.choice(pool) for _ in range(min(n, 120000)))
    if kind == 2:                                            # runs
        out = bytearray()
        while len(out) < n:
            out += bytes([rng.randrange(256)]) * rng.choice([1, 2, 3, 4, 5, 17, 258, 259, 1000, 40000])
        return bytes(out[:n])
    if kind == 3:                                            # periodic: matches at one distance, up to far beyond the ring
        p = rng.choice([1, 2, 3, 5, 64, 255, 256, 257, 4000, 7900, 7935, 8000, 8200, 16000, 32000, 32768, 40000])
        base = bytes(rng.randrange(256) for _ in range(min(p, 40000)))
        return (base * (n // max(len(base), 1) + 1))[:n]
    if kind == 4:                                            # text
        words = [bytes(rng.choice(b"ACGTNacgt=*IHMDS0123456789:\t") for _ in range(rng.randint(1, 12))) for _ in range(rng.randint(2, 400))]
        out = bytearray()
        while len(out) < n:
            out += rng.choice(words)
        return bytes(out[:n])
    if kind == 5:                                            # near copies of earlier stretches with point changes (read-like)
        out = bytearray(rng.randrange(4) + 65 for _ in range(min(n, 300)))
        while len(out) < n:
            a = rng.randrange(max(1, len(out) - 30000), len(out) + 1) if len(out) > 1 else 0
            seg = bytearray(out[max(0, a - rng.randint(20, 300)):a])
            for _ in range(rng.randint(0, 3)):
                if seg: seg[rng.randrange(len(seg))] = rng.randrange(256)
            out += seg or b"x"
        return bytes(out[:n])
    if kind == 6:
        return bytes(rng.getrandbits(8) for _ in range(min(n, 100000)))
    return b"".join(payload(rng)[:rng.randint(0, 30000)] for _ in range(rng.randint(1, 4)))

def bgzf(rng, data):
    out = bytearray()
    i = 0
    while i < len(data) or not out:
        blk = rng.choice([1, 2, 100, 777, 4096, 30000, 0xff00, 0xff00, 0xff00])
        raw = data[i:i + blk]
        i += max(len(raw), 1)
        level = rng.choice([0, 1, 2, 4, 6, 9])
        strat = rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])
        comp = zlib.compressobj(level, zlib.DEFLATED, -15, rng.choice([1, 5, 8, 9]), strat)
        cdata = comp.compress(raw)
        if rng.random() < 0.3:
            cdata += comp.flush(zlib.Z_FULL_FLUSH)           # (several DEFLATE blocks, an empty stored block between them)
            cdata += comp.compress(b"")
        cdata += comp.flush()
        if len(cdata) + 26 > 65536:                          # an incompressible block that outgrew the container: halve it
            i -= len(raw)
            half = raw[:len(raw) // 2]
            comp = zlib.compressobj(0, zlib.DEFLATED, -15)
            cdata = comp.compress(half) + comp.flush()
            raw = half
            i += len(raw)
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cdata) + 25) + cdata
        out += struct.pack("<II", zlib.crc32(raw) & 0xffffffff, len(raw))
        if len(data) == 0:
            break
    if rng.random() < 0.7:
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00\x1b\x00\x03\x00\x00\x00\x00\x00\x00\x00\x00\x00"
    return bytes(out)

t0 = time.time(); bad_cases = 0; total = 0
for k in range(n_cases):
    rng = random.Random(seed0 * 1000003 + k)
    data = payload(rng)
    f = bgzf(rng, data)
    nb = f.count(b"\x1f\x8b\x08\x04")
    cap = max(len(data) + 64, 64)
    buf = np.zeros(cap, np.uint8)
    n_out, bad = C.c_size_t(0), C.c_int32(0)
    capi.check(capi.lib().hgx_bgzf_inflate(f, C.c_size_t(len(f)), capi.ptr(buf), C.c_size_t(cap), C.byref(n_out), C.byref(bad), None))
    ok = bad.value == 0 and n_out.value == len(data) and bytes(buf[:n_out.value]) == data
    total += len(data)
    if not ok:
        bad_cases += 1
        print("case %d seed %d: MISMATCH (bad blocks %d, %d of %d bytes, %d blocks)" % (k, seed0, bad.value, n_out.value, len(data), nb), flush=True)
print("%d cases, %.1f MB of payload, %d mismatches, %.0f s" % (n_cases, total / 1e6, bad_cases, time.time() - t0))
sys.exit(1 if bad_cases else 0)
