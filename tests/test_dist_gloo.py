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
