"""End-to-end rate from an alignment FILE (SAM text / BAM) to the typing result on this box: hgx_parse_alignment_file
(read, inflate, decode, name grouping, front-end) + upload + the GPU path."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hisatgenotype_amd  # noqa
from hisatgenotype_amd import synth, bamio, locus as hl
ht = sys.modules["hisatgenotype_amd.typing"]
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
loc = synth.make_hla_like_locus(n_alleles=7000, n_vars=2500, seed=101)
pl = hl.PackedLocus.from_synth(loc)
pl.index()
sample = synth.pick_sample(loc, 101)
sam = synth.simulate_sam_fast(loc, sample, n_pairs, err_rate=0.002, seed=100)
d = tempfile.mkdtemp()
open(os.path.join(d, "x.sam"), "w").write(sam)
t0 = time.time()
bamio.write_bam(os.path.join(d, "x.bam"), sam, [(loc.ref_allele, len(loc.backbone))])
print("(python BAM writer: %.0f s)" % (time.time() - t0))
for f in ("x.sam", "x.bam"):
    path = os.path.join(d, f)
    for rep in range(3):
        t0 = time.time()
        res = ht.type_locus(pl, None, alignment_file=path)
        dt = time.time() - t0
    print("%s (%.0f MB): %.3f s end to end, %.2f M reads/s, top-2 %s" % (
        f, os.path.getsize(path) / 1e6, dt, res.num_reads / dt / 1e6, [a for a, _ in res.gene_prob[:2]]))
