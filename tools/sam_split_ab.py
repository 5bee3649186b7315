"""A/B of the SAM text call with its line table + record fields beside the upload's tail (default) and behind the last byte
(front=sam_whole): hgx_type_file on a 1 M-read file, the two settings in turn, median and spread of the wall times.
usage: python tools/sam_split_ab.py [pairs] [rounds]"""
import ctypes as C, os, sys, tempfile, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hisatgenotype_amd  # noqa
from hisatgenotype_amd import capi, engine, synth, locus as hl
ht = sys.modules["hisatgenotype_amd.typing"]
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
loc = synth.make_hla_like_locus(n_alleles=7000, n_vars=2500, seed=101)
pl = hl.PackedLocus.from_synth(loc)
pl.index()
sam = synth.simulate_sam_fast(loc, synth.pick_sample(loc, 101), n_pairs, err_rate=0.002, seed=100)
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "x.sam")
open(path, "w").write(sam)
del sam
L = capi.lib()
def call():
    o = capi.ParseOpts(2, 1, 0, 0, 0, 0, 0, 0)
    to = ht.TypeOpts(1, 0, -1, 0, None, None, None, None, None)
    h = C.c_void_p()
    t0 = time.perf_counter()
    capi.check(L.hgx_type_file(C.byref(h), pl.h, pl.index(), path.encode(), pl.ref_allele.encode(), C.byref(o), C.byref(to), None))
    dt = time.perf_counter() - t0
    L.hgx_typing_destroy(h)
    return dt * 1e3
for _ in range(3):
    call()
t = {"parts": [], "whole": []}
for r in range(rounds):
    for kind in ("parts", "whole"):
        time.sleep(0.3)
        if kind == "whole":
            with engine.test_switches(front="sam_whole"):
                t[kind].append(call())
        else:
            t[kind].append(call())
for kind, v in t.items():
    v = sorted(v)
    print("%-6s median %.2f ms, min %.2f, max %.2f  (%s)" % (kind, statistics.median(v), v[0], v[-1], " ".join("%.1f" % x for x in v)))
os.remove(path)
