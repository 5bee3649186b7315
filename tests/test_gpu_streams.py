"""Stream placement (hgx_type.hip, round 6): callers in flight each hold a stream set (an EM stream: the chain of short dependent
launches on the critical path; a gene-side stream).  Two such chains on one hardware LANE take 2.3x as long, and which lane a new
stream lands on depends on how many streams the process -- libhgx, torch, RCCL, the caller -- created before.  The library therefore
MEASURES (chain beside chain, the kernels' own clock) and keeps candidates whose lanes fit.  Here: whatever number of foreign streams
exists first, the EM streams of three callers in flight sit on three different lanes, checked with the library's probe AND with an
independent chain-against-chain measurement; results are those of the one-at-a-time path."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import ctypes as C, sys, threading
    sys.path.insert(0, %r)
    import numpy as np
    import hisatgenotype_amd as hgx
    from hisatgenotype_amd import capi, engine, synth, locus as hl
    n_foreign, kind = int(sys.argv[1]), sys.argv[2]
    foreign = []
    if kind == "torch":                                     # (torch initialises the GPU first, as in a torch.distributed program)
        import torch
        torch.cuda.set_device(0)
        foreign = [torch.cuda.Stream(device=0, priority=-1 if k %% 2 else 0) for k in range(n_foreign)]
        x = torch.ones(8, device="cuda")
        for s_ in foreign:
            with torch.cuda.stream(s_):
                x = x + 1
        torch.cuda.synchronize()
    capi.set_device(0)
    if kind != "torch":
        for k in range(n_foreign):
            p = C.c_void_p()
            capi.check(capi.lib().hgx_stream_create_prio(C.byref(p), C.c_int(k %% 2)))
            foreign.append(p)
    loc = synth.make_hla_like_locus(n_alleles=1500, n_vars=1200, seed=41)
    pl = hl.PackedLocus.from_synth(loc)
    pl.index()
    sams = [synth.simulate_sam_fast(loc, synth.pick_sample(loc, 9 + k), 9000, err_rate=0.004, seed=17 + k) for k in range(3)]
    alone = [hgx.type_locus(pl, s_) for s_ in sams]
    assert engine.stream_sets_info()["sets"] == 1
    out, errs, bar = [None] * 3, [], threading.Barrier(3)
    def work(k):
        try:
            capi.set_device(0)
            capi.set_stream_slot(("streams test", k))
            st = capi.get_stream(2)
            db = pl.parse_sam_dev(sams[k], stream=st)
            T = sys.modules["hisatgenotype_amd.typing"]
            for it in range(6):
                bar.wait()                                  # all three inside hgx_type_dbatch at the same time: three sets out
                res = T.LocusResult()
                res.num_reads, res.num_pairs = db.n_reads, db.n_pairs
                out[k] = T._type_batch(pl, None, res, True, dbatch=db, stream=st, overlap=True)
            db.close()
        except BaseException as e:
            errs.append(e)
            bar.abort()
    ths = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs, errs
    for k in range(3):
        assert out[k].gene_prob == alone[k].gene_prob and out[k].counts_sorted == alone[k].counts_sorted and out[k].em == alone[k].em
    info = engine.stream_sets_info()
    assert info["sets"] == 3 and len(info["free_sets"]) == 3, info
    em_lanes = [e for e, _ in info["free_sets"]]
    assert len(set(em_lanes)) == 3, info                     # three EM chains, three lanes
    assert all(g != e for e, g in info["free_sets"]), info   # no gene side on its own EM chain's lane
    # ... and an independent look: chain against chain on every pair of the EM streams
    buf = (C.c_void_p * 16)()
    ns = C.c_int32()
    capi.check(capi.lib().hgx_stream_sets_streams(buf, C.c_int32(16), C.byref(ns)))
    ems = [C.c_void_p(buf[2 * i]) for i in range(ns.value)]
    for i in range(3):
        for j in range(3):
            if i != j:
                us = np.zeros(2)
                capi.check(capi.lib().hgx_stream_probe_chain(ems[i], ems[j], C.c_int32(1), capi.ptr(us)))
                assert us[1] < 1.5 * us[0], (i, j, us, info)
                same = C.c_int32()
                capi.check(capi.lib().hgx_stream_probe_pair(ems[i], ems[j], C.byref(same)))
                assert same.value == 0
    # placed caller streams may be destroyed: one that represents a lane is kept alive inside the library, later placements still probe beside it
    made = []
    for k in range(5):
        p = C.c_void_p()
        capi.check(capi.lib().hgx_stream_create_placed(C.byref(p), C.c_int(0)))
        made.append(p)
    for p in made:
        capi.check(capi.lib().hgx_stream_destroy(p))
    p = C.c_void_p()
    capi.check(capi.lib().hgx_stream_create_placed(C.byref(p), C.c_int(0)))
    again = hgx.type_locus(pl, sams[0], stream=p)
    assert again.gene_prob == alone[0].gene_prob
    print("placed ok", n_foreign, kind, info)
''') % ROOT


@pytest.mark.parametrize("n_foreign,kind", [(0, "hgx"), (1, "hgx"), (2, "hgx"), (3, "hgx"), (5, "hgx"), (3, "torch"), (6, "torch")])
def test_em_chains_of_callers_in_flight_get_lanes_of_their_own(tmp_path, n_foreign, kind):
    w = tmp_path / "w.py"
    w.write_text(WORKER)
    r = subprocess.run([sys.executable, str(w), str(n_foreign), kind], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "placed ok" in r.stdout
