"""bench.py's own file -> result leg (end_to_end: SAM text and BAM, 1 M reads) with the SAM call's two-part form (default) and with
front=sam_whole, in turn, in one process that has first run what the default bench runs before it (timed steps, samples in flight:
stream sets and worker threads exist).  usage: python tools/sam_split_ab2.py [rounds]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import hisatgenotype_amd as hgx
from hisatgenotype_amd import engine, synth, locus as hl
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
loc = synth.make_hla_like_locus(n_alleles=7000, n_vars=2500, seed=101)
pl = hl.PackedLocus.from_synth(loc)
pl.index()
sam = synth.simulate_sam_fast(loc, synth.pick_sample(loc, 101), 500000, err_rate=0.002, seed=100)
ref = hgx.type_locus(pl, sam)
if os.environ.get("AB_INFLIGHT", "1") != "0":                  # what the default bench has done by then: three callers in flight
    import threading
    db = pl.parse_sam_dev(sam)
    def work():
        for _ in range(5):
            engine.type_dbatch(pl, db) if hasattr(engine, "type_dbatch") else hgx.type_locus(pl, sam)
    ths = [threading.Thread(target=work) for _ in range(3)]
    [t.start() for t in ths]; [t.join() for t in ths]
for r in range(rounds):
    for kind in ("parts", "whole"):
        if kind == "whole":
            with engine.test_switches(front="sam_whole"):
                e = bench.end_to_end(pl, loc, sam, ref)
        else:
            e = bench.end_to_end(pl, loc, sam, ref)
        print(kind, "sam", e["sam"]["ms"], e["sam"]["runs_ms"], "spaced", e["sam"].get("back_to_back", {}).get("runs_ms"), "| bam", e["bam"]["ms"], flush=True)
