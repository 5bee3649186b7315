"""Where a typing() call on the configs[0] fixture spends its time: the C call, the result copy, the report."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu  # noqa: E402
import hisatgenotype_amd as hgx  # noqa: E402
from hisatgenotype_amd import capi, engine, locus as hl  # noqa: E402
T = sys.modules["hisatgenotype_amd.typing"]

name = sys.argv[1] if len(sys.argv) > 1 else "hla_7000_10k"
fx = gu.load(name)
pl = hl.PackedLocus.from_synth(fx["_locus"])
pl.index()
sam = fx["sam"].encode()
if os.environ.get("CLUSTER_WG"):
    engine.test_switch("emx_cluster_wg", os.environ["CLUSTER_WG"])
for front in (None, "device", "host")[:1 if os.environ.get("CLUSTER_WG") else 3]:
    with engine.test_switches(**({"front": front} if front else {})):
        for em_fast in (False, True):
            ts = []
            for it in range(6):
                t0 = time.perf_counter()
                db = pl.parse_sam_dev(sam)
                t1 = time.perf_counter()
                o = T.TypeOpts(1, 0, -1, 0, None, None, None, None, None, T._em_mode(em_fast))
                h = C.c_void_p()
                capi.check(capi.lib().hgx_type_dbatch(C.byref(h), pl.h, pl.index(), db.h, C.byref(o), None))
                t2 = time.perf_counter()
                res = T.LocusResult()
                res.num_reads, res.num_pairs = db.n_reads, db.n_pairs
                T._result_from_handle(h, pl, res, False)
                capi.lib().hgx_typing_destroy(h)
                t3 = time.perf_counter()
                lines, _ = hgx.report_lines(res, False, (), True)
                text = "\n".join(lines)
                t4 = time.perf_counter()
                db.close()
                ts.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
            ts.sort(key=sum)
            m = ts[len(ts) // 2]
            print("front=%-6s route=%s em_fast=%-5s | parse %.2f ms | hgx_type_dbatch %.2f | result copy %.2f | report lines %.2f | EMs %s" % (
                front, engine.front_last(), em_fast, m[0] * 1e3, m[1] * 1e3, m[2] * 1e3, m[3] * 1e3,
                [(e["n_classes"], e["n_iter"]) for e in res.em]))
