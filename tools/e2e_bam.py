"""BAM (coordinate-sorted) -> typing result through hgx_type_file: phase profile on this host."""
import ctypes as C, os, resource, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hisatgenotype_amd  # noqa
from hisatgenotype_amd import capi, synth, bamio, locus as hl
if os.environ.get("HGX_BENCH_LIB"):          # an A/B against another build of libhgx
    from hisatgenotype_amd import capi as _capi
    _capi.LIB_PATH = os.environ["HGX_BENCH_LIB"]
ht = sys.modules["hisatgenotype_amd.typing"]
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 0
loc = synth.make_hla_like_locus(n_alleles=7000, n_vars=2500, seed=101)
pl = hl.PackedLocus.from_synth(loc)
pl.index()
sample = synth.pick_sample(loc, 101)
sam = synth.simulate_sam_fast(loc, sample, n_pairs, err_rate=0.002, seed=100)
if os.environ.get("LONG_NAMES"):         # sequencer-style read names (38 characters: five 8-byte sort chunks) instead of the generator's short ones
    import re
    def long_name(m):
        k = int(m.group(1))
        return "A00123:45:HXXXXXXX:%d:%d:%d:%d\t" % (1 + k % 4, 1101 + (k // 4) % 578, 1000 + (k * 7919) % 30000, 1000 + (k * 104729) % 36000)
    sam = re.sub(r"(?m)^[A-Za-z_]*(\d+)[^\t]*\t", long_name, sam)
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "x.bam")
t0 = time.time()
bamio.write_bam_native(path, sam, [(loc.ref_allele, len(loc.backbone))], sort_by_coordinate=True)
print("native BAM writer: %.2f s, %.1f MB" % (time.time() - t0, os.path.getsize(path) / 1e6))
n_reads = sam.count("\n")
del sam
L = capi.lib()
if os.environ.get("FRONT_SWITCH"):       # e.g. FRONT_SWITCH=name_chunks: a test switch of the front end for an A/B
    from hisatgenotype_amd import engine
    engine.test_switch("front", os.environ["FRONT_SWITCH"])
for rep in range(4):
    if rep == 3:
        os.environ["HGX_PARSE_PROFILE"] = "1"
    o = capi.ParseOpts(2, 1, 0, 0, 0, 0, 0, nt)
    r0 = resource.getrusage(resource.RUSAGE_SELF)
    to = ht.TypeOpts(1, 0, -1, 0, None, None, None, None, None)
    h = C.c_void_p()
    t0 = time.perf_counter()
    capi.check(L.hgx_type_file(C.byref(h), pl.h, pl.index(), path.encode(), pl.ref_allele.encode(), C.byref(o), C.byref(to), None))
    dt = time.perf_counter() - t0
    L.hgx_typing_destroy(h)
    r1 = resource.getrusage(resource.RUSAGE_SELF)
    print("run %d: %.1f ms = %.2f M reads/s; CPU %.0f ms" % (rep, dt * 1e3, n_reads / dt / 1e6,
                                                             (r1.ru_utime + r1.ru_stime - r0.ru_utime - r0.ru_stime) * 1e3), flush=True)
    time.sleep(0.4)
os.remove(path)
