"""Two PROCESSES, one GPU, control plane over gloo: type_locus_sharded end to end on real shards -- the exchanges of dist.TorchComm
(pileup all-reduce inside the front end, class tables of DIFFERENT sizes gathered in rank order, totals) between ranks that score their
shares on the same MI355X.  Cases: a natural split, a rank with NO pairs at all, an STR locus.  Every rank's result must be the
unsharded result.  (A second GPU is not needed for any of this; the 2-GPU RCCL test stays gated on device_count >= 2.)"""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch.distributed as dist
    import hisatgenotype_amd as hgx
    from hisatgenotype_amd import synth, locus as hl, dist as hdist, capi, engine
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    capi.set_device(0)                                   # both ranks on the one GPU of the box
    comm = hdist.TorchComm()
    cases = []
    loc = synth.make_hla_like_locus(n_alleles=1500, n_vars=1200, seed=41)
    sam = synth.simulate_sam_fast(loc, synth.pick_sample(loc, 9), 6000, err_rate=0.004, seed=17)
    cases.append(("natural split", loc, sam, hdist.split_name_grouped(sam, world)))
    cases.append(("rank 1 holds no pairs", loc, sam, [sam.encode()] + [b""] * (world - 1)))
    cases.append(("rank 0 holds no pairs", loc, sam, [b""] * (world - 1) + [sam.encode()]))
    str_loc = synth.make_str_like_locus(gene="TH01", unit="AATG", max_repeats=12, min_repeats=4, seed=5)
    names = [a for a in str_loc.allele_names if "BACKBONE" not in a]
    str_sam = synth.simulate_sam_fast(str_loc, [names[2], names[-3]], 900, read_len=100, frag_len=(200, 300), err_rate=0.002, seed=8)
    cases.append(("STR locus", str_loc, str_sam, hdist.split_name_grouped(str_sam, world)))
    for what, loc, sam, shards in cases:
        pl = hl.PackedLocus.from_synth(loc)
        # every shard through the DEVICE front end (forced: they are below its size gate); the pileup counters are summed between the
        # two processes out of and back into HBM (gloo has no device path); then once more on the host route
        with engine.test_switches(front="device"):
            res = hdist.type_locus_sharded(pl, shards[rank], comm)
            if shards[rank]:
                assert engine.front_last() == (2, 0), (what, rank, engine.front_last())
        res_h = hdist.type_locus_sharded(pl, shards[rank], comm, front="host")
        assert res_h.gene_prob == res.gene_prob and res_h.counts_sorted == res.counts_sorted and res_h.em == res.em, (what, rank)
        ref = hgx.type_locus(pl, sam)
        assert (res.num_reads, res.num_pairs) == (ref.num_reads, ref.num_pairs), (what, rank)
        assert res.counts_sorted == ref.counts_sorted, (what, rank)
        assert [e["n_iter"] for e in res.em] == [e["n_iter"] for e in ref.em], (what, rank)
        assert res.gene_prob == ref.gene_prob, (what, rank)
        got = [None] * world
        dist.all_gather_object(got, res.gene_prob)
        assert all(g == got[0] for g in got), what
        pl.close()
    dist.barrier()
    dist.destroy_process_group()
    print("rank %%d ok" %% rank)
''') % ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_processes_share_one_gpu_over_gloo(tmp_path):
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), str(w)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") >= 2


RCCL_ORDER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", sys.argv[2])
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    import numpy as np
    import torch, torch.distributed as dist
    import hisatgenotype_amd
    from hisatgenotype_amd import synth, locus as hl, dist as hdist, capi
    torch_first = sys.argv[1] == "torch_first"
    torch.cuda.set_device(0)
    capi.set_device(0)
    if torch_first:
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
        t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
        comm = hdist.RcclComm.from_torch()
    else:
        comm = hdist.RcclComm(0, 1, lambda x: x)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
        t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
    loc = synth.make_hla_like_locus(n_alleles=300, n_vars=500, seed=21)
    pl = hl.PackedLocus.from_synth(loc)
    comm.broadcast_index(pl, 0)
    a = comm.allreduce_sum(np.arange(10, dtype=np.int64))
    assert a.tolist() == list(range(10))
    sam = synth.simulate_sam_fast(loc, synth.pick_sample(loc, 3), 900, err_rate=0.003, seed=5)
    import hisatgenotype_amd as hgx
    res = hdist.type_locus_sharded(pl, sam, comm)
    ref = hgx.type_locus(pl, sam)
    assert res.gene_prob == ref.gene_prob and res.counts_sorted == ref.counts_sorted
    t2 = torch.ones(4, device="cuda"); dist.all_reduce(t2); torch.cuda.synchronize()     # torch's communicator still works afterwards
    comm.close()
    dist.destroy_process_group()
    print("order ok")
''') % ROOT


@pytest.mark.parametrize("order", ["torch_first", "hgx_first"])
def test_rccl_communicator_beside_a_torch_nccl_group(tmp_path, order):
    """libhgx's own RCCL communicator (dist.RcclComm: ncclCommInitRank through ctypes, exchanges on device buffers through the C-ABI)
    created AFTER and BEFORE a torch.distributed nccl group in the same process: both keep working (world size 1 here; the two
    runtimes share the HIP context and load librccl once)."""
    w = tmp_path / "w.py"
    w.write_text(RCCL_ORDER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(w), order, str(_free_port())], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "order ok" in r.stdout


def _bench(args, timeout=1500):
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]                 # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 4])
def test_bench_body_with_ranks_sharing_the_gpu(n):
    """The REAL bench body -- not --dry-run -- with N ranks on the one GPU of the box, control plane over gloo: bench.py launches
    the ranks itself, rank 0's index is broadcast, every rank types its own sample, the time is the max over ranks, the reads are
    summed, every rank's file -> result leg runs at the same time as the others'.  (RCCL refuses two ranks on one device; the nccl
    form of the same body differs in make_comm() only.)"""
    line = _bench(["--gpus", str(n), "--backend", "gloo", "--share-gpu", "--pairs", "20000", "--steps", "2", "--warmup", "1", "--no-workloads"])
    assert line["n_gpus"] == n and line["steps"] == 2 and line["scaling"] == "weak"
    cfg = line["config"]
    assert cfg["shared_gpu"] is True and cfg["comm_kind"].startswith("torch-gloo") and cfg["index_broadcast_bytes"] > 1_000_000
    assert cfg["device_front_end_batch_identical_to_host"] is True
    assert sorted(cfg["top2"]) == sorted(cfg["true_alleles"])
    # value = the reads of ALL ranks over the slowest rank's time
    assert abs(line["value"] - n * cfg["pairs_per_gpu"] * 2 * line["steps"] / (line["ms_per_step"] * line["steps"] * 1e-3)) / line["value"] < 0.02
    e2e = line["e2e"]
    assert e2e["results_identical_to_hbm_path_on_every_rank"] is True and e2e["ms_per_call"] > 0
    assert line["roofline"]["frac"] > 0


@pytest.mark.parametrize("n", [4, 5])
def test_bench_class1_body_with_sharded_loci_on_a_shared_gpu(n):
    """bench.py --workload class1 with more ranks than loci, all on GPU 0 over gloo: HLA-A (and, with 5 ranks, B) has its pairs
    sharded over a rank group -- device front end per shard, pileup all-reduce at parse, class tables gathered and merged per
    step -- and every rank of a sharded locus also types the whole locus alone: identical result required."""
    line = _bench(["--gpus", str(n), "--backend", "gloo", "--share-gpu", "--workload", "class1", "--pairs", "30000", "--steps", "2", "--warmup", "1",
                   "--check-unsharded", "--no-cpu-baseline"])
    cfg = line["config"]
    assert line["n_gpus"] == n and cfg["shared_gpu"] is True
    groups = cfg["rank_groups"]
    assert sorted(r for g in groups.values() for r in g) == list(range(n)) and len(groups["A"]) == 2
    assert cfg["sharded_equals_unsharded"] is True
    assert all(c["correct"] for c in cfg["calls"].values()) and len(cfg["calls"]) == 3
    assert cfg["comm_kind"] == {"A": "torch-gloo (host control plane)"}
    assert cfg["exchange"]["collectives_per_step"] >= 3 and cfg["exchange"]["bytes_received_per_step"] > cfg["exchange"]["bytes_sent_per_step"] > 0
    assert cfg["front_end_route_of_my_shards"]["A"][0] == 2         # rank 0's shard went through the device front end
    assert cfg["e2e_shards"]["device_front_end_on_every_rank_and_results_identical_to_the_resident_path"] is True


def _bench_nccl_world1(args, force):
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", HGX_FORCE_DIST=force, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    return json.loads(lines[0])


def test_bench_nccl_body_with_a_world_of_one():
    """The nccl form of the multi-rank body on the one GPU there is: torch.distributed over RCCL with ONE rank (HGX_FORCE_DIST) -- process
    group on the device, the index broadcast on a tensor aliasing the index block, the max / sum reductions on device tensors, the
    per-rank file -> result leg."""
    line = _bench_nccl_world1(["--pairs", "20000", "--steps", "2", "--warmup", "1", "--no-workloads"], "1")
    cfg = line["config"]
    assert line["n_gpus"] == 1 and cfg["comm_kind"].startswith("torch-nccl") and cfg["index_broadcast_bytes"] > 1_000_000
    assert sorted(cfg["top2"]) == sorted(cfg["true_alleles"]) and cfg["device_front_end_batch_identical_to_host"] is True
    assert line["e2e"]["results_identical_to_hbm_path_on_every_rank"] is True


def test_bench_class1_exchanges_through_the_library_s_own_rccl_communicator():
    """bench.py --workload class1 under nccl with every locus given a communicator of its own (a group of ONE rank: HGX_FORCE_DIST=comm):
    make_comm() must come back with dist.RcclComm (`comm_kind: rccl`), and the per-step exchanges -- hgx_classes_allgather of both
    levels, the totals -- run on device buffers through the C-ABI (`exchange` counts them from hgx_rccl_stats); the results equal the
    unsharded path's."""
    line = _bench_nccl_world1(["--workload", "class1", "--pairs", "30000", "--steps", "2", "--warmup", "1", "--check-unsharded", "--no-cpu-baseline"], "comm")
    cfg = line["config"]
    assert cfg["comm_kind"] == {"A": "rccl", "B": "rccl", "C": "rccl"}, cfg["comm_kind"]
    assert cfg["sharded_equals_unsharded"] is True and all(c["correct"] for c in cfg["calls"].values())
    ex = cfg["exchange"]
    assert ex["collectives_per_step"] >= 9 and ex["bytes_sent_per_step"] > 100_000          # 3 loci x (2 x 2 all-gathers + the totals)
    assert all(r[0] == 2 for r in cfg["front_end_route_of_my_shards"].values())
    assert cfg["e2e_shards"]["device_front_end_on_every_rank_and_results_identical_to_the_resident_path"] is True
