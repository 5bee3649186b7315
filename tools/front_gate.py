"""Where the device front end starts to beat the host stages (the size gate of hgx_front.hip parse_dev): SAM text and BAM files of
N pairs through hgx_parse_sam_dev / hgx_parse_alignment_file_dev with front=device and front=host, median wall time."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hisatgenotype_amd as hgx  # noqa: E402,F401
from hisatgenotype_amd import bamio, engine, locus as hl, synth  # noqa: E402

loc = synth.make_hla_like_locus(n_alleles=int(os.environ.get("ALLELES", "7000")), n_vars=2500, seed=101)
pl = hl.PackedLocus.from_synth(loc)
pl.index()
tmp = tempfile.mkdtemp(dir="/dev/shm")
print("%8s | %28s | %28s | %28s" % ("pairs", "SAM text dev / host / default", "SAM file dev / host / default", "BAM file dev / host / default"))
for n in (50, 150, 400, 1000, 2500, 5000, 10000, 20000):
    sam = synth.simulate_sam_fast(loc, synth.pick_sample(loc, 3), n, err_rate=0.002, seed=n).encode()
    fs = os.path.join(tmp, "s%d.sam" % n)
    fb = os.path.join(tmp, "s%d.bam" % n)
    open(fs, "wb").write(sam)
    bamio.write_bam_native(fb, sam, [(loc.ref_allele, len(loc.backbone))], sort_by_coordinate=True)
    row = []
    for what in ("text", fs, fb):
        cell = []
        for front in ("device", "host", None):
            with engine.test_switches(**({"front": front} if front else {})):
                ts = []
                for _ in range(7):
                    t0 = time.perf_counter()
                    db = pl.parse_sam_dev(sam) if what == "text" else pl.parse_alignment_file_dev(what, [pl.ref_allele])
                    ts.append(time.perf_counter() - t0)
                    db.close()
                ts.sort()
                cell.append("%.2f" % (ts[len(ts) // 2] * 1e3))
            if front is None:
                cell[-1] += " r%d" % engine.front_last()[0]
        row.append(" / ".join(cell))
    print("%8d | %28s | %28s | %28s" % (n, *row))
