"""Randomised parity of the DEVICE front end on the GPU box: the batch the kernels build in HBM (record route and key route) against
the pinned host front end's batch, byte for byte, on the cases of tools/fuzz_parity.py (HLA-like and STR loci, errors, soft clips,
novel indels, duplicates, multi-hit and single-end records).  usage: tools/fuzz_front.py [n_cases] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import hisatgenotype_amd  # noqa
from hisatgenotype_amd import capi, engine, locus as hl
import fuzz_parity

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 990000
bad = 0
routes = {}
t0 = time.time()
for k in range(n_cases):
    loc, sam, single = fuzz_parity.make_case(seed0, k, 1 + k % 4)
    pl = hl.PackedLocus.from_synth(loc)
    for ec in (True, False):
        try:
            host = pl.parse_sam(sam, error_correction=ec, allow_discordant=single)
        except capi.HgxError as e:
            for extra in ("device", "device,keys"):
                try:
                    with engine.test_switches(front=extra):
                        pl.parse_sam_dev(sam, error_correction=ec, allow_discordant=single)
                    print("case %d seed %d: host raised %r, device did not" % (k, seed0 + k, e)); bad += 1
                except capi.HgxError:
                    pass
            continue
        L = len(loc.backbone)
        for extra in ("device", "device,keys"):
            with engine.test_switches(front=extra):
                dev = pl.parse_sam_dev(sam, error_correction=ec, allow_discordant=single)
                route, code = engine.front_last()
            routes[(route, code)] = routes.get((route, code), 0) + 1
            hb = dev.to_host()
            same = all(getattr(hb, f).tobytes() == getattr(host, f).tobytes() for f in ("pieces", "masks", "pair_off", "pair_ref")) and \
                hb.n_reads == host.n_reads
            if same and route > 0:
                (na, ca), (nb, cb) = hb.pileup(L), host.pileup(L)
                same = np.array_equal(na, nb) and np.array_equal(ca, cb)
            if not same:
                print("case %d seed %d ec %s %s: MISMATCH (route %d, code %d)" % (k, seed0 + k, ec, extra, route, code)); bad += 1
    pl.close()
    if (k + 1) % 100 == 0:
        print("%d cases, %d mismatches, %.0f s" % (k + 1, bad, time.time() - t0), flush=True)
print("%d cases (x 2 error-correction settings x 2 routes), %d mismatches; (route, decline code) -> inputs: %s; %.0f s" % (
    n_cases, bad, dict(sorted(routes.items())), time.time() - t0))
sys.exit(1 if bad else 0)
