"""File -> typing result through ONE C call (hgx_type_file): phase profile and thread scaling on this host.
usage: python tools/e2e_file.py [pairs] [threads ...]"""
import ctypes as C, os, resource, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hisatgenotype_amd  # noqa
from hisatgenotype_amd import capi, synth, locus as hl
if os.environ.get("HGX_BENCH_LIB"):          # an A/B against another build of libhgx
    from hisatgenotype_amd import capi as _capi
    _capi.LIB_PATH = os.environ["HGX_BENCH_LIB"]
ht = sys.modules["hisatgenotype_amd.typing"]
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
threads = [int(x) for x in sys.argv[2:]] or [0]
loc = synth.make_hla_like_locus(n_alleles=7000, n_vars=2500, seed=101)
pl = hl.PackedLocus.from_synth(loc)
pl.index()
sample = synth.pick_sample(loc, 101)
sam = synth.simulate_sam_fast(loc, sample, n_pairs, err_rate=0.002, seed=100)
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "x.sam")
open(path, "w").write(sam)
n_reads = sam.count("\n")
del sam
L = capi.lib()
if os.environ.get("HGX_E2E_FRONT"):          # a test switch for the whole run, e.g. HGX_E2E_FRONT=sam_whole
    from hisatgenotype_amd import engine
    engine.test_switch("front", os.environ["HGX_E2E_FRONT"])
for nt in threads:
    for rep in range(3):
        time.sleep(0.4)          # one call at a time: the container's CPU quota refills between calls
        if rep == 2:
            os.environ["HGX_PARSE_PROFILE"] = "1"
        o = capi.ParseOpts(2, 1, 0, 0, 0, 0, 0, nt)
        to = ht.TypeOpts(1, 0, -1, 0, None, None, None, None, None)
        h = C.c_void_p()
        r0 = resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.perf_counter()
        capi.check(L.hgx_type_file(C.byref(h), pl.h, pl.index(), path.encode(), pl.ref_allele.encode(), C.byref(o), C.byref(to), None))
        dt = time.perf_counter() - t0
        r1 = resource.getrusage(resource.RUSAGE_SELF)
        cpu = (r1.ru_utime - r0.ru_utime, r1.ru_stime - r0.ru_stime)
        L.hgx_typing_destroy(h)
        os.environ.pop("HGX_PARSE_PROFILE", None)
    print("threads %3d: %.1f ms end to end = %.2f M reads/s; CPU time of the call %.0f ms user + %.0f ms system" % (
        nt, dt * 1e3, n_reads / dt / 1e6, cpu[0] * 1e3, cpu[1] * 1e3), flush=True)
    time.sleep(0.5)
os.remove(path)
