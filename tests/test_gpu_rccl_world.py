"""libhgx's RCCL entry points with a world of 2-4 ranks -- on ONE GPU.  RCCL itself refuses two ranks on one device, so the transport is
a stand-in (tests/fake_rccl/fake_rccl.cpp: RCCL's signatures and datatype codes, the bytes moved between the processes through /dev/shm):
everything AROUND the transport is the product's own code and runs for real -- hgx_index_broadcast into an uninitialised index,
hgx_allreduce_sum_u32 on the pileup counters in HBM (the device front end of every shard), hgx_classes_allgather (sizes gather, padded
gather, rank-order unpack, weighted merge), hgx_allreduce_sum_i64, dist.RcclComm, dist.type_locus_sharded -- and bench.py's class-I body
with `comm_kind: rccl`.  Every rank's result must be the unsharded result.  (What stays unexecuted on this pool: RCCL's own kernels
between two devices.)"""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory):
    d = tmp_path_factory.mktemp("fake_rccl")
    lib = str(d / "libfake_rccl.so")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", lib, os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cpp")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return lib


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


WORKER = textwrap.dedent('''
    import ctypes as C, os, sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch.distributed as dist
    import hisatgenotype_amd as hgx
    from hisatgenotype_amd import synth, locus as hl, dist as hdist, capi, engine
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    capi.set_device(0)                                   # every rank on the one GPU of the box
    hdist.RcclComm.lib_path = sys.argv[1]
    comm = hdist.RcclComm.from_torch()
    assert (comm.rank, comm.world) == (rank, world)
    # plain sums through hgx_allreduce_sum_i64 / _u32 (device buffers made by the wrapper)
    a = comm.allreduce_sum(np.arange(1000, dtype=np.int64) * (rank + 1))
    assert np.array_equal(a, np.arange(1000, dtype=np.int64) * (world * (world + 1) // 2))
    u = comm.allreduce_u32(np.array([7, 0xFFFFFFF0 if rank == 0 else 3], np.uint32))
    assert u.tolist() == [7 * world, (0xFFFFFFF0 + 3 * (world - 1)) & 0xFFFFFFFF]
    cases = []
    loc = synth.make_hla_like_locus(n_alleles=1500, n_vars=1200, seed=41)
    sam = synth.simulate_sam_fast(loc, synth.pick_sample(loc, 9), 6000, err_rate=0.004, seed=17)
    cases.append(("natural split", loc, sam, hdist.split_name_grouped(sam, world)))
    cases.append(("the last rank holds everything", loc, sam, [b""] * (world - 1) + [sam.encode()]))
    cases.append(("rank 0 holds everything", loc, sam, [sam.encode()] + [b""] * (world - 1)))
    str_loc = synth.make_str_like_locus(gene="TH01", unit="AATG", max_repeats=12, min_repeats=4, seed=5)
    names = [a_ for a_ in str_loc.allele_names if "BACKBONE" not in a_]
    str_sam = synth.simulate_sam_fast(str_loc, [names[2], names[-3]], 900, read_len=100, frag_len=(200, 300), err_rate=0.002, seed=8)
    cases.append(("STR locus", str_loc, str_sam, hdist.split_name_grouped(str_sam, world)))
    # CODIS D18S51: choose_pairs needs the MEDIAN inner distance of the whole sample -- a histogram all-reduce (hgx_allreduce_sum_i64)
    d18 = synth.make_str_like_locus(gene="D18S51", unit="AGAA", max_repeats=22, min_repeats=9, flank=180, seed=71)
    d18.base_fname = "codis"
    dn = [a_ for a_ in d18.allele_names if "BACKBONE" not in a_]
    d18_sam = synth.simulate_sam_fast(d18, [dn[3], dn[-3]], 3000, read_len=100, frag_len=(200, 280), err_rate=0.002, seed=17)
    cases.append(("CODIS D18S51", d18, d18_sam, hdist.split_name_grouped(d18_sam, world)))
    L = capi.lib()
    for what, loc, sam, shards in cases:
        pl = hl.PackedLocus.from_synth(loc)
        ref = hgx.type_locus(pl, sam)                     # the unsharded result, on this rank's own index
        # the index broadcast: every rank but the root swaps in an index of the same shape with UNINITIALISED tables and receives into it
        if rank != 0:
            h = C.c_void_p()
            capi.check(L.hgx_index_create_device(C.byref(h), C.c_int32(pl.n_alleles), C.c_int32(pl.n_vars)))
            L.hgx_index_destroy(pl._index)
            pl._index = h
            blk, nb = C.c_void_p(), C.c_size_t()
            capi.check(L.hgx_index_device_block(h, C.byref(blk), C.byref(nb)))
            capi.check(L.hgx_memset(blk, 0, nb, None))      # (zeros: a result equal to the reference's can only come from the received tables)
            capi.sync()
        n0 = C.c_uint64()
        capi.check(L.hgx_rccl_stats(C.byref(n0), None, None, C.c_int32(0)))
        comm.broadcast_index(pl, 0)
        again = hgx.type_locus(pl, sam)                   # ... and types with what it received
        assert again.gene_prob == ref.gene_prob and again.counts_sorted == ref.counts_sorted, (what, rank, "after the broadcast")
        for sw in (dict(front="device"), dict()):
            with engine.test_switches(**sw):
                res = hdist.type_locus_sharded(pl, shards[rank], comm)
                if sw and shards[rank]:
                    assert engine.front_last() == (2, 0), (what, rank, engine.front_last())
            assert (res.num_reads, res.num_pairs) == (ref.num_reads, ref.num_pairs), (what, rank)
            assert res.counts_sorted == ref.counts_sorted and res.em == ref.em and res.gene_prob == ref.gene_prob, (what, rank)
        n1 = C.c_uint64()
        capi.check(L.hgx_rccl_stats(C.byref(n1), None, None, C.c_int32(0)))
        assert n1.value - n0.value >= 1 + 2 * 4, (what, n0.value, n1.value)      # the broadcast, then per sharded call: pileup, class tables, totals
        got = [None] * world
        dist.all_gather_object(got, res.gene_prob)
        assert all(g == got[0] for g in got), what
        pl.close()
    # a rank whose front end raises (a record without NM: quirk Q8) fails EVERY rank -- the flag travels with the exchanges
    loc = synth.make_hla_like_locus(n_alleles=300, n_vars=400, seed=8)
    pl = hl.PackedLocus.from_synth(loc)
    sam = synth.simulate_sam_fast(loc, synth.pick_sample(loc, 1), 600, seed=2)
    shards = hdist.split_name_grouped(sam, world)
    shards[world - 1] = shards[world - 1].replace(b"NM:i:", b"XM:i:", 1)
    for sw in (dict(), dict(front="device")):
        raised = False
        try:
            with engine.test_switches(**sw):
                hdist.type_locus_sharded(pl, shards[rank], comm)
        except Exception:
            raised = True
        assert raised, (rank, sw)
    res = hdist.type_locus_sharded(pl, hdist.split_name_grouped(sam, world)[rank], comm)      # ... and the communicator is still in step
    assert res.gene_prob == hgx.type_locus(pl, sam).gene_prob
    pl.close()
    comm.close()
    dist.barrier()
    dist.destroy_process_group()
    print("rank %%d ok" %% rank)
''') % ROOT


@pytest.mark.parametrize("world", [2, 3])
def test_c_abi_collectives_and_a_sharded_locus_with_a_world_of_n(tmp_path, fake_rccl, world):
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), str(w), fake_rccl], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-4000:]
    assert r.stdout.count("ok") >= world


@pytest.mark.parametrize("n", [4, 5])
def test_bench_class1_body_over_the_rccl_entry_points(fake_rccl, n):
    """bench.py --workload class1 with more ranks than loci on one GPU, the ranks of a sharded locus exchanging through dist.RcclComm
    (`--comm rccl`; the line says which library stood behind the entry points): hgx_classes_allgather and hgx_allreduce_sum_* per step on
    device buffers, the pileup all-reduce in HBM at parse -- and every rank of a sharded locus equal to the locus typed alone."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", HGX_BENCH_RCCL_LIB=fake_rccl)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--backend", "gloo", "--share-gpu", "--comm", "rccl", "--workload", "class1",
                        "--pairs", "30000", "--steps", "2", "--warmup", "1", "--check-unsharded", "--no-cpu-baseline"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    cfg = json.loads(lines[0])["config"]
    assert cfg["comm_kind"]["A"].startswith("rccl entry points over libfake_rccl.so"), cfg["comm_kind"]
    assert cfg["sharded_equals_unsharded"] is True and all(c["correct"] for c in cfg["calls"].values())
    ex = cfg["exchange"]
    assert ex["collectives_per_step"] >= 5 and ex["bytes_received_per_step"] >= 2 * ex["bytes_sent_per_step"] - 64 > 0      # (two ranks in HLA-A's group)
    assert cfg["front_end_route_of_my_shards"]["A"][0] == 2
    assert cfg["e2e_shards"]["device_front_end_on_every_rank_and_results_identical_to_the_resident_path"] is True
