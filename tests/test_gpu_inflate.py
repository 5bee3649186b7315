"""BGZF inflate on the device (csrc/hgx_inflate.hip: one wavefront per block, DEFLATE decoded with wave-uniform control flow, CRC-32
and ISIZE checked) against zlib: stored / fixed / dynamic blocks, every compression level, matches at distance 1 (runs), long
matches across the 32 KB window, incompressible bytes, empty blocks, BAM files of the fixtures; damaged blocks are reported."""
import ctypes as C
import os
import random
import struct
import zlib

import numpy as np
import pytest

from hisatgenotype_amd import bamio, capi, synth

pytestmark = pytest.mark.gpu
EOF_BLOCK = bamio._BGZF_EOF


@pytest.fixture(autouse=True, params=[None, "inflate_small_pool", "inflate_v1"], ids=["lane_parallel", "lane_parallel_small_pool", "round4"])
def inflate_form(request):
    """Every test with the forms of the device inflate the library holds: the default (k_bgzf_inflate_w: the symbol loop lane-parallel,
    64 bit offsets per window, second-level tables for the long codes), the same with a second-level pool of 32 entries (test switch
    front=inflate_small_pool: most long codes overflow into the wave-uniform path) and round 4's one-symbol-per-trip kernel
    (k_bgzf_inflate, test switch front=inflate_v1)."""
    from hisatgenotype_amd import engine
    if request.param:
        engine.test_switch("front", request.param)
    yield request.param
    engine.test_switch("front", None)


def bgzf(payload, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, block=0xff00, eof=True):
    out = bytearray()
    for i in range(0, max(len(payload), 1), block):
        raw = payload[i:i + block]
        comp = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
        cdata = comp.compress(raw) + comp.flush()
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cdata) + 25) + cdata
        out += struct.pack("<II", zlib.crc32(raw) & 0xffffffff, len(raw))
    if eof:
        out += EOF_BLOCK
    return bytes(out)


def inflate_dev(data):
    n_out, bad = C.c_size_t(0), C.c_int32(0)
    cap = 64 * 1024 * (data.count(b"\x1f\x8b\x08\x04") + 2)
    buf = np.zeros(cap, np.uint8)
    capi.check(capi.lib().hgx_bgzf_inflate(data, C.c_size_t(len(data)), capi.ptr(buf), C.c_size_t(cap), C.byref(n_out), C.byref(bad), None))
    return bytes(buf[:n_out.value]), bad.value


def payloads():
    rng = random.Random(7)
    yield "empty", b""
    yield "one byte", b"x"
    yield "text", ("".join(rng.choice("ACGT") for _ in range(200000)) + "\n").encode() * 2
    yield "runs", b"A" * 100000 + b"AB" * 40000 + b"ABC" * 30000 + bytes(70000)
    yield "random", bytes(rng.getrandbits(8) for _ in range(150000))
    yield "window", (bytes(rng.getrandbits(8) for _ in range(30000)) * 9)[:250000]          # matches at distances close to 32 KB
    loc = synth.make_hla_like_locus(n_alleles=60, n_vars=200, seed=3)
    yield "sam", synth.simulate_sam_fast(loc, synth.pick_sample(loc, 1), 4000, err_rate=0.01, seed=5).encode()
    yield "skewed", bytes(rng.choice(b"aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaabcdefghijklmnopqrstuvwxyz0123456789" * 3 + bytes(range(256))) for _ in range(180000))


@pytest.mark.parametrize("name,payload", list(payloads()), ids=[n for n, _ in payloads()])
def test_device_inflate_equals_zlib(name, payload):
    for level, strategy, block in ((6, zlib.Z_DEFAULT_STRATEGY, 0xff00), (1, zlib.Z_DEFAULT_STRATEGY, 0xff00), (9, zlib.Z_DEFAULT_STRATEGY, 0xff00),
                                   (0, zlib.Z_DEFAULT_STRATEGY, 30000), (6, zlib.Z_FIXED, 0xff00), (6, zlib.Z_HUFFMAN_ONLY, 20000), (6, zlib.Z_RLE, 0xff00),
                                   (6, zlib.Z_DEFAULT_STRATEGY, 777)):
        data = bgzf(payload, level, strategy, block)
        got, bad = inflate_dev(data)
        assert bad == 0, (name, level, strategy, block, bad)
        assert got == payload, (name, level, strategy, block)


def test_device_inflate_reports_damaged_blocks():
    rng = random.Random(11)
    payload = ("".join(rng.choice("ACGTN") for _ in range(300000))).encode()
    data = bytearray(bgzf(payload))
    good, bad = inflate_dev(bytes(data))
    assert bad == 0 and good == payload
    n_blocks = bytes(data).count(b"\x1f\x8b\x08\x04")
    for trial in range(12):
        d = bytearray(data)
        at = rng.randrange(40, len(d) - 60)
        d[at] ^= 1 << rng.randrange(8)
        try:
            got, bad = inflate_dev(bytes(d))
        except capi.HgxError:
            continue                                   # (the flip hit a container header: refused before anything is launched)
        assert bad >= 1 or got == payload              # (a flip in a header's don't-care field changes nothing)
        assert bad <= 2 and n_blocks >= 4
    # a wrong CRC-32 alone
    d = bytearray(data)
    first_len = struct.unpack_from("<H", d, 16)[0] + 1
    d[first_len - 8] ^= 0x40
    got, bad = inflate_dev(bytes(d))
    assert bad == 1


def test_device_inflate_on_bam_files(tmp_path):
    import golden_util as gu
    for name in gu.ALL[:4]:
        fx = gu.load(name)
        loc = fx["_locus"]
        p = str(tmp_path / (name + ".bam"))
        bamio.write_bam_native(p, fx["sam"].encode(), [(loc.ref_allele, len(loc.backbone))], sort_by_coordinate=True)
        data = open(p, "rb").read()
        got, bad = inflate_dev(data)
        assert bad == 0 and got == b"".join(bamio._bgzf_blocks(data))


def test_device_inflate_stream_that_outruns_its_block():
    """A well-formed container around a DEFLATE stream that asks for far more bits than the block holds -- here the LAST block of
    the file, cut short inside its Huffman data with the size fields rewritten: the bit reader is fed zeros beyond the block (it
    must not walk into whatever lies behind the buffer), the block is reported, the blocks before it are inflated."""
    rng = random.Random(11)
    payload = bytes(rng.getrandbits(8) for _ in range(3 * 0xff00))          # incompressible: ~64 KB of Huffman literals per block
    blocks = []
    for i in range(0, len(payload), 0xff00):
        raw = payload[i:i + 0xff00]
        comp = zlib.compressobj(6, zlib.DEFLATED, -15, 9, zlib.Z_HUFFMAN_ONLY)
        blocks.append((raw, comp.compress(raw) + comp.flush()))
    def member(cdata, raw):
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cdata) + 25) + cdata +
                struct.pack("<II", zlib.crc32(raw) & 0xffffffff, len(raw)))
    for keep in (40, 1000, 30000):
        raw, cdata = blocks[-1]
        assert len(cdata) > 60000
        data = b"".join(member(c, r) for r, c in blocks[:-1]) + member(cdata[:keep], raw)      # no EOF block: this one ends the file
        got, bad = inflate_dev(data)
        assert bad == 1, (keep, bad)
        assert got[:2 * 0xff00] == payload[:2 * 0xff00]


def test_device_inflate_fuzz_cases():
    """200 cases of tools/fuzz_inflate.py (random payload textures, levels, strategies, block sizes, several DEFLATE blocks per member)."""
    import importlib.util
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_inflate.py")
    r = subprocess.run([sys.executable, tool, "200", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "200 cases" in r.stdout and " 0 mismatches" in r.stdout
