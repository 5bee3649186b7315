"""Front-end (hgx_parse_sam) phase times and thread scaling on this host.  HGX_PARSE_PROFILE=1 prints the phases."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["HGX_PARSE_PROFILE"] = "1"
import hisatgenotype_amd  # noqa
from hisatgenotype_amd import synth, locus as hl
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
loc = synth.make_hla_like_locus(n_alleles=7000, n_vars=2500, seed=101)
pl = hl.PackedLocus.from_synth(loc)
sample = synth.pick_sample(loc, 101)
sam = synth.simulate_sam_fast(loc, sample, n_pairs, err_rate=0.002, seed=100)
print("SAM text %.0f MB, host threads %d" % (len(sam) / 1e6, os.cpu_count()))
for nt in (1, 8, 16, 32, 64):
    t0 = time.time()
    b = pl.parse_sam(sam, n_threads=nt)
    dt = time.time() - t0
    print("threads %3d: %.2f M reads/s (%.3f s)" % (nt, b.n_reads / dt / 1e6, dt), flush=True)
