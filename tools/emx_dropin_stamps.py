"""Where the reference-order EM #1 of configs[0] (1 730 classes x 4 500 alleles) spends its time, by the kernel's own phase stamps (lab
build: emx_stamps): one workgroup vs the cluster mode a lone problem gets since round 6 (CLUSTER_WG = workgroups)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import golden_util as gu
import hisatgenotype_amd  # noqa
from hisatgenotype_amd import capi
capi.use_lab()
from hisatgenotype_amd import engine, locus as hl
capi.set_device(0)
fx = gu.load("hla_7000_10k")
pl = hl.PackedLocus.from_synth(fx["_locus"])
A = pl.n_alleles
em = fx["em"][0]
rows = np.stack([gu.class_bits(fx, cid, A) for cid, _ in em["cmpt"]])
bits = np.zeros((len(rows), pl.w64), np.uint64)
bits[:, :rows.shape[1]] = rows
counts = np.array([n for _, n in em["cmpt"]], np.int64)
for mode in ("one workgroup", "cluster"):
    engine.test_switch(None)
    engine.test_switch("emx_stamps", "1")
    if mode == "one workgroup":
        engine.test_switch("emx_cluster_lone", "0")
    elif os.environ.get("CLUSTER_WG"):
        engine.test_switch("emx_cluster_wg", os.environ["CLUSTER_WG"])
    cl = engine.Classes.from_host(bits, counts, pl.a_pad)
    cl.set_allele_rank(pl.name_rank)
    for rep in range(3):
        t0 = time.perf_counter()
        p, it = cl.em(A, em["remove_low"], pl.allele_len if em["use_length"] else None)
        dt = time.perf_counter() - t0
    print("%s: %d iterations, call %.2f ms" % (mode, it, dt * 1e3), flush=True)
    cl.close()
