"""N > 1 path on CPU: world_size 2 over gloo -- index broadcast and task sharding (no GPU needed)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch.distributed as dist
    import hisatgenotype_amd
    from hisatgenotype_amd import synth, locus as hl, dist as hdist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    loc = synth.make_hla_like_locus(n_alleles=300, n_vars=500, seed=21)
    pl = hl.PackedLocus.from_synth(loc)
    ref = {k: v.copy() for k, v in pl.tables().items()}
    if rank != 0:                       # wipe the device-bound tables on the receiving rank
        pl._tables["link_bits"][:] = 0
        pl._tables["exon_mask"][:] = 0
    nbytes = hdist.broadcast_index(pl, src=0)
    t = pl.tables()
    assert np.array_equal(t["link_bits"], ref["link_bits"]) and np.array_equal(t["exon_mask"], ref["exon_mask"])
    assert np.array_equal(t["gene_mask"], ref["gene_mask"]) and nbytes == (pl.n_words * pl.a_pad + 4 * pl.w64) * 4
    tasks = [(s, l) for s in range(8) for l in "ABC"]
    weights = [{"A": 5, "B": 7, "C": 3}[l] for _, l in tasks]
    mine = hdist.shard(tasks, rank, world, weights)
    got = [None] * world
    dist.all_gather_object(got, mine)
    flat = sorted(x for part in got for x in part)
    assert flat == sorted(tasks), "tasks lost or duplicated"
    loads = [sum({"A": 5, "B": 7, "C": 3}[l] for _, l in part) for part in got]
    assert max(loads) - min(loads) <= 7
    # intra-locus read sharding: the pileup of a sharded parse (all-reduced inside the front-end) is the whole sample's, and
    # gathered class tables come back in rank order -- no GPU needed for either
    sample = synth.pick_sample(loc, 3)
    sam = synth.simulate_sam_fast(loc, sample, 1200, err_rate=0.004, seed=5)
    comm = hdist.TorchComm()
    shard_text = hdist.split_name_grouped(sam, world)[rank]
    whole = pl.parse_sam(sam)
    mine_b = pl.parse_sam(shard_text, pileup_exchange=comm.allreduce_u32)
    L = len(loc.backbone)
    nt_w, cnt_w = whole.pileup(L)
    nt_s, cnt_s = mine_b.pileup(L)
    assert np.array_equal(cnt_w, cnt_s) and np.array_equal(nt_w, nt_s), "sharded pileup differs from the whole sample's"
    pairs = [None] * world
    dist.all_gather_object(pairs, (mine_b.n_pairs, mine_b.n_reads))
    assert sum(p for p, _ in pairs) == whole.n_pairs and sum(r for _, r in pairs) == whole.n_reads
    bits = np.full((3 + rank, pl.w64), rank + 1, np.uint64)
    cnt = np.arange(3 + rank, dtype=np.int64) + 10 * rank
    got_t = comm.all_gather_tables(bits, cnt)
    assert [b.shape[0] for b, _ in got_t] == [3, 4][:world] and all(int(b[0, 0]) == r + 1 for r, (b, _) in enumerate(got_t))
    assert [c.tolist() for _, c in got_t] == [[0, 1, 2], [10, 11, 12, 13]][:world]
    tot = comm.allreduce_sum(np.array([mine_b.n_reads, 7], np.int64))
    assert tot.tolist() == [whole.n_reads, 7 * world]
    big = comm.allreduce_u32(np.array([0xC0000000, 5], np.uint32))          # uint32 sums travel as int32: wrap-around is the same
    assert big.tolist() == [(0xC0000000 * world) & 0xFFFFFFFF, 5 * world]
    # the whole sharded typing sequence on the host form (no GPU here: parse_shard takes the host front end): exchange order and
    # failure protocol -- a rank whose shard is unparseable fails BOTH ranks, nobody hangs
    _, db_or_none = None, None
    try:
        hdist.parse_shard(pl, shard_text if rank == 0 else shard_text.replace(b"NM:i:", b"XM:i:", 1), comm)
        raise SystemExit("a failing rank went unnoticed")
    except SystemExit:
        raise
    except Exception as e:
        assert ("front-end" in str(e)) or ("reference would fail" in str(e)) or ("NM" in str(e)), e
    assert hdist.assign_ranks_to_loci([7000, 8000, 7000], 4) == {0: [0], 1: [1, 2], 2: [3]}
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
''') % ROOT


def test_broadcast_and_shard_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29577", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count("ok") == 2


def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` (no launcher) starts two ranks itself -- as a child process, before anything touches the
    GPU -- and rank 0 reports the world it saw; with the RCCL backend and fewer GPUs than asked for it fails loudly instead
    of printing n_gpus = 1 (Missing #2)."""
    import json
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["world_size"] == 2 and rec["gpus_arg"] == 2
    # no GPU here: asking for two of them over RCCL must fail, not fall back to one
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode != 0 and "refusing" in (out.stderr + out.stdout)
    # under a launcher whose world size disagrees with --gpus
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"],
                         env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)


def test_bench_dry_run_class1_rank_groups():
    """`bench.py --gpus 4 --backend gloo --dry-run --workload class1`: four ranks meet over gloo, every rank computes the rank groups of
    configs[2] (dist.assign_ranks_to_loci: three loci on four ranks, the largest locus gets two), and they agree."""
    import json
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--backend", "gloo", "--dry-run", "--workload", "class1"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 4 and rec["my_loci_agree"] and rec["every_rank_has_work"]
    groups = rec["rank_groups"]
    assert sorted(groups) == ["A", "B", "C"] and sorted(r for v in groups.values() for r in v) == [0, 1, 2, 3]
    assert max(len(v) for v in groups.values()) == 2
