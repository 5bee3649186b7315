import os, sys, tempfile, time
sys.path.insert(0, "/root/repo")
os.environ["HGX_PARSE_PROFILE"] = "1"
import hisatgenotype_amd
from hisatgenotype_amd import synth, bamio, locus as hl, engine, capi
ht = sys.modules["hisatgenotype_amd.typing"]
loc = synth.make_hla_like_locus(n_alleles=7000, n_vars=2500, seed=101)
pl = hl.PackedLocus.from_synth(loc); pl.index()
sample = synth.pick_sample(loc, 101)
sam = synth.simulate_sam_fast(loc, sample, 250000, err_rate=0.002, seed=100)
d = tempfile.mkdtemp()
open(os.path.join(d, "x.sam"), "w").write(sam)
bamio.write_bam(os.path.join(d, "x.bam"), sam, [(loc.ref_allele, len(loc.backbone))])
for f in ("x.sam", "x.bam"):
    path = os.path.join(d, f)
    for rep in range(2):
        t0 = time.time(); batch = pl.parse_alignment_file(path); t1 = time.time()
        db = engine.DeviceBatch(batch); t2 = time.time()
        bufs = engine.ScoreBuffers(pl, db, exon=True); t3 = time.time()
        res = ht.LocusResult(); res.num_reads, res.num_pairs = batch.n_reads, batch.n_pairs
        ht._type_batch(pl, batch, res, True, dbatch=db, bufs=bufs); capi.sync(); t4 = time.time()
        del bufs, db; t5 = time.time()
    print("%s: parse_file %.1f ms | upload %.1f | alloc bufs %.1f | GPU path %.1f | free %.1f" % (f, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t5-t4)*1e3), flush=True)
