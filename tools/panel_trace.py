#!/usr/bin/env python3
"""Kernel timeline of ONE configs[3] panel call (hgx_type_many_loci over 6 loci x 64 samples).
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ptrace -o t -- python3 tools/panel_trace.py run
  python3 tools/panel_trace.py show gpurun_out/ptrace/*/t_kernel_trace.csv   (per queue: start, duration, gap to the queue's previous kernel)"""
import os, sys, csv, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    from hisatgenotype_amd import engine, locus as hl, synth
    import bench
    htyping = sys.modules["hisatgenotype_amd.typing"]
    loci = [synth.make_hla_like_locus(gene=g, n_alleles=a, length=ln, n_vars=v, seed=500 + i, var_id_base=10000 * i) for i, (g, a, ln, v) in enumerate(bench.PANEL)]
    pls, manies = [], []
    for k, loc in enumerate(loci):
        pl = hl.PackedLocus.from_synth(loc); pl.index()
        bs = [pl.parse_sam(synth.simulate_sam_fast(loc, synth.pick_sample(loc, 1000 * s + k), 5000, err_rate=0.002, seed=100 * s + k)) for s in range(64)]
        pls.append(pl); manies.append(engine.ManyBatch(pl, bs))
    for _ in range(4):
        htyping.type_many_loci(pls, manies, light=True)
else:
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "k_piece_compat_pat" in r["Kernel_Name"]]
    i0 = starts[-6]
    t0 = int(rows[i0]["Start_Timestamp"])
    last_end, busy = {}, collections.defaultdict(int)
    for r in rows[i0:]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        q = r.get("Queue_Id", "?")
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:60]
        print("%9.1f us  +%7.1f us  q%-3s gap %6.1f  %s" % (s / 1e3, (e - s) / 1e3, q, (s - last_end.get(q, s)) / 1e3, name))
        last_end[q] = e
        busy[q] += e - s
    print({q: round(v / 1e3, 1) for q, v in busy.items()})
