"""Multi-GPU plumbing (8e): one process per GPU, loci / samples shard with no data-path collective.

The only exchange is the start-up broadcast of the shared locus index (the word-major link bit matrix plus
the two level masks, a few MB) from the rank that built it: ``torch.distributed`` with backend ``nccl`` (= RCCL
over xGMI) on GPUs, ``gloo`` in the CPU tests.  Work assignment is static and deterministic."""
import ctypes as C

import numpy as np

from . import capi


def shard(items, rank, world, weights=None):
    """Greedy longest-processing-time assignment of independent (sample, locus) tasks to ranks.
    Returns the items of ``rank`` (in input order).  Deterministic on every rank."""
    n = len(items)
    w = list(weights) if weights is not None else [1] * n
    load = [0] * world
    owner = [0] * n
    for i in sorted(range(n), key=lambda i: (-w[i], i)):
        r = min(range(world), key=lambda r: (load[r], r))
        owner[i] = r
        load[r] += w[i]
    return [items[i] for i in range(n) if owner[i] == rank]


def broadcast_index(pl, src=0, group=None):
    """Make ``pl``'s device index on every rank hold rank ``src``'s packed tables.

    Every rank passes a PackedLocus of the same locus (host tables are cheap to rebuild from the
    reference's text files); the device-resident bit matrix is what gets broadcast, so only ``src``
    needs to have packed it.  Returns the number of bytes broadcast."""
    import torch
    import torch.distributed as dist
    backend = dist.get_backend(group)
    on_gpu = backend == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    rank = dist.get_rank(group)
    dims = torch.zeros(4, dtype=torch.int64, device=dev)
    if rank == src:
        t = pl.tables()
        dims[:] = torch.tensor([pl.n_alleles, pl.a_pad, pl.n_vars, pl.n_words], dtype=torch.int64)
    dist.broadcast(dims, src, group=group)
    n_alleles, a_pad, n_vars, n_words = (int(x) for x in dims.tolist())
    if (n_alleles, a_pad, n_vars, n_words) != (pl.n_alleles, pl.a_pad, pl.n_vars, pl.n_words):
        raise ValueError("rank %d holds a different locus than rank %d" % (rank, src))
    nb = n_words * a_pad
    w64 = a_pad // 64
    # one flat int32 buffer: link bits, exon mask, gene mask
    buf = torch.zeros(nb + 4 * w64, dtype=torch.int32, device=dev)
    if rank == src:
        flat = np.concatenate([t["link_bits"].reshape(-1).view(np.int32), t["exon_mask"].view(np.int32),
                               t["gene_mask"].view(np.int32)])
        buf.copy_(torch.from_numpy(flat))
    dist.broadcast(buf, src, group=group)
    host = buf.cpu().numpy()
    bits = np.ascontiguousarray(host[:nb].view(np.uint32))
    em = np.ascontiguousarray(host[nb:nb + 2 * w64].view(np.uint64))
    gm = np.ascontiguousarray(host[nb + 2 * w64:].view(np.uint64))
    rep = pl.tables()["rep_of"]          # host-only table, identical on every rank
    pl._tables = dict(link_bits=bits.reshape(n_words, a_pad), exon_mask=em, gene_mask=gm, rep_of=rep)
    if on_gpu or capi_has_device():
        h = C.c_void_p()
        capi.check(capi.lib().hgx_index_create(C.byref(h), C.c_int32(n_alleles), C.c_int32(n_vars), capi.ptr(bits),
                                               capi.ptr(em), capi.ptr(gm)))
        if pl._index is not None:
            capi.lib().hgx_index_destroy(pl._index)
        pl._index = h
    return int(buf.numel()) * 4


def capi_has_device():
    try:
        return capi.device_count() > 0
    except Exception:
        return False
