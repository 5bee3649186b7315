"""Native alignment reader (hgx_read_alignments) phase times and thread scaling on this host."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["HGX_PARSE_PROFILE"] = "1"
import hisatgenotype_amd  # noqa
from hisatgenotype_amd import synth, bamio
ht = sys.modules["hisatgenotype_amd.typing"]
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
loc = synth.make_hla_like_locus(n_alleles=500, n_vars=400, seed=3)
sample = synth.pick_sample(loc, 1)
sam = synth.simulate_sam_fast(loc, sample, n_pairs, err_rate=0.002, seed=1)
d = tempfile.mkdtemp()
t0 = time.time()
bamio.write_bam(os.path.join(d, "big.bam"), sam, [(loc.ref_allele, len(loc.backbone))])
print("python BAM writer %.1f s, %.1f MB" % (time.time() - t0, os.path.getsize(os.path.join(d, "big.bam")) / 1e6))
open(os.path.join(d, "big.sam"), "w").write(sam)
for f in ("big.bam", "big.sam"):
    for nt in (1, 8, 32, 64):
        t0 = time.time()
        a = ht.read_alignment_text(os.path.join(d, f), n_threads=nt)
        dt = time.time() - t0
        print("%s threads %2d: %.3f s  %.2f M reads/s" % (f, nt, dt, 2 * n_pairs / dt / 1e6), flush=True)
t0 = time.time()
b = ht.read_alignment_text(os.path.join(d, "big.bam"), native=False)
print("pure Python reader: %.1f s  equal=%s" % (time.time() - t0, a.count(b"\n") == b.count(b"\n")))
