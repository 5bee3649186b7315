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


class _DeviceBlock:
    """A device allocation of libhgx presented through __cuda_array_interface__ so that torch can alias it (no copy)."""

    def __init__(self, ptr, n_int32):
        self.__cuda_array_interface__ = {"shape": (int(n_int32),), "typestr": "<i4", "data": (int(ptr), False), "version": 2}


def index_block_tensor(index_handle):
    """torch int32 tensor aliasing the device tables [link bits | exon mask | gene mask] of an hgx_index."""
    import torch
    p, nb = C.c_void_p(), C.c_size_t()
    capi.check(capi.lib().hgx_index_device_block(index_handle, C.byref(p), C.byref(nb)))
    return torch.as_tensor(_DeviceBlock(p.value, nb.value // 4), device=torch.device("cuda", torch.cuda.current_device()))


def broadcast_index(pl, src=0, group=None):
    """Make ``pl``'s device index on every rank hold rank ``src``'s packed tables; returns the number of bytes broadcast.

    On GPUs (backend ``nccl`` = RCCL) the collective runs on the index memory itself: rank ``src`` sends the device block of
    its index, every other rank creates an index of the same shape with uninitialised tables (hgx_index_create_device) and
    receives INTO its block (torch tensors aliasing the allocations, hgx_index_device_block) -- device to device over xGMI,
    no host bounce, no second upload.  With ``gloo`` (CPU tests) the packed host tables travel instead and a device index is
    only created where a GPU exists."""
    import torch
    import torch.distributed as dist
    backend = dist.get_backend(group)
    on_gpu = backend == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    rank = dist.get_rank(group)
    dims = torch.zeros(4, dtype=torch.int64, device=dev)
    if rank == src:
        dims[:] = torch.tensor([pl.n_alleles, pl.a_pad, pl.n_vars, pl.n_words], dtype=torch.int64)
    dist.broadcast(dims, src, group=group)
    n_alleles, a_pad, n_vars, n_words = (int(x) for x in dims.tolist())
    if (n_alleles, a_pad, n_vars, n_words) != (pl.n_alleles, pl.a_pad, pl.n_vars, pl.n_words):
        raise ValueError("rank %d holds a different locus than rank %d" % (rank, src))
    if on_gpu:
        if rank == src:
            h = pl.index()                                  # built here from the packed host tables
        else:
            h = C.c_void_p()
            capi.check(capi.lib().hgx_index_create_device(C.byref(h), C.c_int32(n_alleles), C.c_int32(n_vars)))
            if pl._index is not None:
                capi.lib().hgx_index_destroy(pl._index)
            pl._index = h
        block = index_block_tensor(h)
        dist.broadcast(block, src, group=group)
        torch.cuda.synchronize()
        return int(block.numel()) * 4
    nb = n_words * a_pad
    w64 = a_pad // 64
    # one flat int32 buffer: link bits, exon mask, gene mask
    buf = torch.zeros(nb + 4 * w64, dtype=torch.int32)
    if rank == src:
        t = pl.tables()
        flat = np.concatenate([t["link_bits"].reshape(-1).view(np.int32), t["exon_mask"].view(np.int32),
                               t["gene_mask"].view(np.int32)])
        buf.copy_(torch.from_numpy(flat))
    dist.broadcast(buf, src, group=group)
    host = buf.numpy()
    bits = np.ascontiguousarray(host[:nb].view(np.uint32))
    em = np.ascontiguousarray(host[nb:nb + 2 * w64].view(np.uint64))
    gm = np.ascontiguousarray(host[nb + 2 * w64:].view(np.uint64))
    rep = pl.tables()["rep_of"]          # host-only table, identical on every rank
    pl._tables = dict(link_bits=bits.reshape(n_words, a_pad), exon_mask=em, gene_mask=gm, rep_of=rep)
    if capi_has_device():
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


# ----------------------------------------------------------------------------------------------------------------------
# intra-locus read sharding (optional part): one sample's pairs of ONE locus split over several ranks
# ----------------------------------------------------------------------------------------------------------------------
def split_name_grouped(sam_text, k, simulation=False):
    """Cut name-grouped SAM text into `k` consecutive shards at read boundaries (mates stay together), near-equal in bytes.
    `simulation`: the read id is the QNAME up to its first '|' (typing_core.py:808)."""
    data = sam_text.encode() if isinstance(sam_text, str) else bytes(sam_text)
    n = len(data)
    # The shards are typed in rank order and their class tables concatenated in rank order: that reproduces the unsharded
    # first-seen order for ANY name-grouped stream (sorted or not), because every shard keeps its records in stream order
    # (the front-end does not re-sort a grouped stream: hgx_sam.cpp recognises it and leaves it alone).

    def read_id(ls):
        name = data[ls:data.find(b"\t", ls)]
        return name.split(b"|", 1)[0] if simulation else name

    cuts = [0]
    for i in range(1, k):
        p = max(cuts[-1], n * i // k)
        while p < n:
            e = data.find(b"\n", p)
            if e < 0:
                p = n
                break
            nxt = e + 1
            if nxt >= n:
                p = n
                break
            # the line starting at nxt begins a new read iff its name differs from the name of the line containing p
            ls = data.rfind(b"\n", 0, e) + 1
            if read_id(ls) != read_id(nxt):
                p = nxt
                break
            p = nxt
        cuts.append(p)
    cuts.append(n)
    return [data[cuts[i]:cuts[i + 1]] for i in range(k)]


class TorchComm:
    """The three exchanges of a sharded locus over torch.distributed (nccl = RCCL on GPUs, gloo in the CPU tests)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.on_gpu = dist.get_backend(group) == "nccl"
        self.stats = [0, 0, 0]             # collectives, bytes handed to them, bytes received (bench.py: exchange bytes per step)

    def _count(self, sent, received):
        self.stats[0] += 1
        self.stats[1] += int(sent)
        self.stats[2] += int(received)

    def _dev(self):
        import torch
        return torch.device("cuda", torch.cuda.current_device()) if self.on_gpu else torch.device("cpu")

    def allreduce_sum(self, arr):
        """In-place element-wise sum of an integer numpy array over the ranks (counts < 2^31 travel as int64)."""
        import torch
        t = torch.from_numpy(arr.astype(np.int64)).to(self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        self._count(t.numel() * 8, t.numel() * 8)
        arr[...] = t.cpu().numpy().astype(arr.dtype)
        return arr

    def allreduce_u32(self, arr):
        """In-place sum of a uint32 numpy array over the ranks: the pileup exchange of a sharded parse, host form.  Travels as
        int32 (two's-complement sums are the uint32 sums bit for bit) -- the SAME collective as allreduce_u32_dev, so that ranks
        on the host route and ranks on the device route of one parse meet in it."""
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr, np.uint32).view(np.int32).copy()).to(self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        self._count(t.numel() * 4, t.numel() * 4)
        arr[...] = t.cpu().numpy().view(np.uint32).reshape(arr.shape)
        return arr

    def allreduce_u32_dev(self, dptr, n, stream=None):
        """The device form: `n` uint32 at device pointer `dptr`, summed in place in HBM (nccl = RCCL on an aliasing tensor: no
        host bounce).  Returns the LAST element of the result (the failure flag of dist.type_locus_sharded)."""
        import torch
        if not self.on_gpu:                                 # gloo between processes that do have GPUs: through the host
            h = np.zeros(n, np.uint32)
            capi.sync(stream)
            capi.check(capi.lib().hgx_memcpy_d2h(capi.ptr(h), C.c_void_p(dptr), C.c_size_t(4 * n), stream))
            self.allreduce_u32(h)
            capi.check(capi.lib().hgx_memcpy_h2d(C.c_void_p(dptr), capi.ptr(h), C.c_size_t(4 * n), stream))
            return int(h[-1])
        capi.sync(stream)                                   # the pileup kernels wrote the buffer on the library's stream
        t = torch.as_tensor(_DeviceBlock(dptr, n), device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        self._count(4 * n, 4 * n)
        flag = int(t[-1].item())                            # (synchronises torch's stream: the sum is complete)
        return flag & 0xFFFFFFFF

    def all_gather_tables(self, bits, counts):
        """Every rank's class table (bits [C][w64] uint64, counts [C] int64) -> list in rank order."""
        import torch
        dev = self._dev()
        n = torch.tensor([bits.shape[0]], dtype=torch.int64, device=dev)
        sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(self.world)]
        self.dist.all_gather(sizes, n, group=self.group)
        sizes = [int(x.item()) for x in sizes]
        w64, cap = bits.shape[1], max(max(sizes), 1)
        pad = np.zeros((cap, w64 + 1), np.int64)                      # [row bits | count], padded to the largest table
        pad[:bits.shape[0], :w64] = bits.view(np.int64)
        pad[:bits.shape[0], w64] = counts
        mine = torch.from_numpy(pad).to(dev)
        parts = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine, group=self.group)
        self.stats[0] += 1                                            # (+ the one-element size gather above)
        self._count(pad.nbytes + 8, self.world * (pad.nbytes + 8))
        out = []
        for k, p in zip(sizes, parts):
            a = p.cpu().numpy()
            out.append((np.ascontiguousarray(a[:k, :w64]).view(np.uint64), np.ascontiguousarray(a[:k, w64])))
        return out


class _NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


class RcclComm:
    """An RCCL communicator held by this library's caller (ncclCommInitRank through ctypes) and the exchanges of a sharded locus
    on DEVICE buffers through the C-ABI (hgx_allreduce_sum_*, hgx_classes_allgather, hgx_index_broadcast): no
    `.cpu().numpy()` per step.  The 128-byte unique id of rank 0 reaches the other ranks through `share` -- e.g.
    torch.distributed.broadcast_object_list (any backend), a file, MPI."""

    _lib = None
    lib_path = None          # tests: another library with RCCL's entry points (tests/fake_rccl: ranks that share ONE GPU); set before first use

    @classmethod
    def rccl(cls):
        if cls._lib is None:
            if cls.lib_path:
                from . import engine
                engine.test_switch("rccl_lib", cls.lib_path)       # (libhgx's own loader takes the same library)
                cls._lib = C.CDLL(cls.lib_path, mode=C.RTLD_LOCAL)
                return cls._lib
            for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
                try:
                    cls._lib = C.CDLL(name, mode=C.RTLD_GLOBAL)
                    break
                except OSError:
                    continue
            if cls._lib is None:
                raise ImportError("librccl.so not found")
        return cls._lib

    def __init__(self, rank, world, share):
        """`share(obj)`: returns rank 0's `obj` on every rank (identity for world 1)."""
        import os
        # ranks of ONE node: the bootstrap sockets go over the loopback interface (a container's hostname may not resolve);
        # a multi-node caller sets NCCL_SOCKET_IFNAME itself
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        R = self.rccl()
        uid = _NcclUniqueId()
        if rank == 0:
            rc = R.ncclGetUniqueId(C.byref(uid))
            if rc:
                raise RuntimeError("ncclGetUniqueId failed: %d" % rc)
        if world > 1:                                   # (string_at: a c_char array read as bytes would stop at the first NUL)
            raw = share(C.string_at(C.byref(uid), 128))
            assert len(raw) == 128
            C.memmove(C.byref(uid), raw, 128)
        self.h = C.c_void_p()
        R.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _NcclUniqueId, C.c_int]
        rc = R.ncclCommInitRank(C.byref(self.h), C.c_int(world), uid, C.c_int(rank))
        if rc:
            raise RuntimeError("ncclCommInitRank failed: %d" % rc)
        self.rank, self.world, self.on_gpu = rank, world, True

    @classmethod
    def from_torch(cls, group=None):
        """One communicator per rank of a torch.distributed group (the id travels with broadcast_object_list)."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)

        def share(obj):
            box = [obj]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            return box[0]
        return cls(rank, world, share)

    def close(self):
        if self.h:
            self.rccl().ncclCommDestroy(self.h)
            self.h = None

    def broadcast_index(self, pl, src=0, stream=None):
        capi.check(capi.lib().hgx_index_broadcast(pl.index(), C.c_int32(src), self.h, stream))
        capi.sync(stream)

    def allreduce_sum(self, arr, stream=None):
        """In-place element-wise sum of an integer numpy array over the ranks (the front-end's pileup / histogram live on the host)."""
        d = capi.DevArray.from_host(np.ascontiguousarray(arr.astype(np.int64)).ravel(), stream)
        capi.check(capi.lib().hgx_allreduce_sum_i64(capi.ptr(d), C.c_size_t(arr.size), self.h, stream))
        arr[...] = d.to_host(stream).reshape(arr.shape).astype(arr.dtype)
        return arr

    def allreduce_u32(self, arr, stream=None):
        """The pileup exchange of a sharded parse, host form (the same collective as allreduce_u32_dev)."""
        d = capi.DevArray.from_host(np.ascontiguousarray(arr, np.uint32).ravel(), stream)
        capi.check(capi.lib().hgx_allreduce_sum_u32(capi.ptr(d), C.c_size_t(arr.size), self.h, stream))
        arr[...] = d.to_host(stream).reshape(arr.shape)
        return arr

    def allreduce_u32_dev(self, dptr, n, stream=None):
        """The device form: hgx_allreduce_sum_u32 on the front end's own counter table, on the front end's stream.  Returns the
        last element of the result (the failure flag)."""
        capi.check(capi.lib().hgx_allreduce_sum_u32(C.c_void_p(dptr), C.c_size_t(n), self.h, stream))
        flag = np.zeros(1, np.uint32)
        capi.sync(stream)
        capi.check(capi.lib().hgx_memcpy_d2h(capi.ptr(flag), C.c_void_p(dptr + 4 * (n - 1)), C.c_size_t(4), stream))
        return int(flag[0])

    def merge_classes(self, cl, a_pad, stream=None):
        """This rank's class set of one level (or None) -> the merged class set of the whole sample, on every rank."""
        from . import engine
        h = C.c_void_p()
        capi.check(capi.lib().hgx_classes_allgather(C.byref(h), cl.h if cl is not None else None, C.c_int32(a_pad), self.h, stream))
        return engine.Classes(h)


class LocalComm:
    """`world` shards of one locus typed by `world` threads of ONE process (tests, and a single GPU standing in for several):
    the same three exchanges through a barrier."""

    class _Shared:
        def __init__(self, world):
            import threading
            self.world, self.barrier, self.slots = world, threading.Barrier(world), [None] * world

    def __init__(self, shared, rank):
        self.sh, self.rank, self.world, self.on_gpu = shared, rank, shared.world, False

    @staticmethod
    def make(world):
        sh = LocalComm._Shared(world)
        return [LocalComm(sh, r) for r in range(world)]

    def _exchange(self, value):
        self.sh.slots[self.rank] = value
        self.sh.barrier.wait()
        got = list(self.sh.slots)
        self.sh.barrier.wait()
        return got

    def allreduce_sum(self, arr):
        parts = self._exchange(arr.copy())
        arr[...] = np.sum(np.stack([p.astype(np.int64) for p in parts]), axis=0).astype(arr.dtype)
        return arr

    def allreduce_u32(self, arr):
        parts = self._exchange(np.ascontiguousarray(arr, np.uint32).copy())
        arr[...] = np.sum(np.stack(parts), axis=0, dtype=np.uint32).reshape(arr.shape)
        return arr

    def allreduce_u32_dev(self, dptr, n, stream=None):
        """Threads of one process standing in for ranks: every thread's table comes down, is summed, goes back up."""
        h = np.zeros(n, np.uint32)
        capi.sync(stream)
        capi.check(capi.lib().hgx_memcpy_d2h(capi.ptr(h), C.c_void_p(dptr), C.c_size_t(4 * n), stream))
        self.allreduce_u32(h)
        capi.check(capi.lib().hgx_memcpy_h2d(C.c_void_p(dptr), capi.ptr(h), C.c_size_t(4 * n), stream))
        return int(h[-1])

    def all_gather_tables(self, bits, counts):
        return self._exchange((bits.copy(), counts.copy()))


def merge_class_tables(tables, a_pad, stream=None):
    """Class tables of the shards (rank order = stream order of their pairs) -> ONE class set: rows concatenated, equal rows
    merged with summed counts by hgx_dedup_classes, which keeps first-seen order -- the dict the reference would have built
    over all pairs (typing_core.py:1229-1234)."""
    from . import engine
    bits = np.concatenate([b for b, _ in tables]) if tables else np.zeros((0, a_pad // 64), np.uint64)
    cnt = np.concatenate([c for _, c in tables]) if tables else np.zeros(0, np.int64)
    if len(cnt) == 0:
        return engine.Classes.from_host(np.zeros((0, a_pad // 64), np.uint64), np.zeros(0, np.int64), a_pad)
    rows = capi.DevArray.from_host(np.ascontiguousarray(bits, np.uint64), stream)
    w = capi.DevArray.from_host(np.ascontiguousarray(cnt, np.int64), stream)
    merged = engine.Classes.dedup(rows, len(cnt), a_pad, weights=w, stream=stream)
    capi.sync(stream)
    return merged


def type_shard(pl, batch, db, comm, remove_low_abundance_alleles=True, stream=None):
    """The device side of a sharded locus for a shard whose piece batch is resident (`db`: engine.DeviceBatch from a parse that
    already exchanged the pileup; `batch`: its host form, or None for a batch born in HBM): this rank's class tables, the two
    exchanges, the EMs on the merged tables.  bench.py --workload class1 times this per step."""
    from . import engine
    from .typing import LocusResult, TypeOpts, _result_from_handle
    hla = pl.base_fname == "hla"
    if batch is None:
        batch = db
    if batch.n_pairs > 0:
        bufs = engine.ScoreBuffers(pl, db, exon=hla)
        engine.piece_compat(pl, db, bufs, stream)
        engine.pair_classes(pl, db, bufs, stream, exon=False)
        levels = ([("exon", engine.Classes.of_level(pl, db, bufs, 0, stream))] if hla else []) + \
            [("gene", engine.Classes.dedup(bufs.gene_bits, db.n_pairs, pl.a_pad, hashes=bufs.gene_hash, stream=stream))]
    else:
        levels = [(name, None) for name in (("exon", "gene") if hla else ("gene",))]
    merged = {}
    for name, cl in levels:
        if hasattr(comm, "merge_classes"):                  # device to device over RCCL (hgx_classes_allgather)
            merged[name] = comm.merge_classes(cl, pl.a_pad, stream)
            if cl is not None:
                cl.close()
            continue
        if cl is None:
            bits, cnt = np.zeros((0, pl.w64), np.uint64), np.zeros(0, np.int64)
        else:
            bits, cnt, _ = cl.to_host()
            cl.close()
        merged[name] = merge_class_tables(comm.all_gather_tables(bits, cnt), pl.a_pad, stream)
    totals = comm.allreduce_sum(np.array([batch.n_reads, batch.n_pairs], np.int64))
    res = LocusResult()
    res.num_reads, res.num_pairs = int(totals[0]), int(totals[1])
    res.n_pieces, res.n_refs = batch.n_pieces, batch.n_refs
    try:
        if res.num_reads > 0:
            o = TypeOpts(int(bool(remove_low_abundance_alleles)), 0, 0, 0, None, None, None, None, None)
            h = C.c_void_p()
            ecl = merged.get("exon")
            rc = capi.lib().hgx_type_classes(C.byref(h), pl.h, ecl.h if ecl is not None else None, merged["gene"].h,
                                             C.c_int32(res.num_reads), C.c_int32(res.num_pairs), C.byref(o), stream)
            if rc == -7:
                raise TypeError(capi.lib().hgx_last_error().decode(errors="replace"))
            capi.check(rc)
            try:
                _result_from_handle(h, pl, res, False)
            finally:
                capi.lib().hgx_typing_destroy(h)
    finally:
        for cl in merged.values():
            cl.close()
    return res


def parse_shard(pl, sam_shard, comm, num_editdist=2, error_correction=True, allow_discordant=False, simulation=False, base_locus=0,
                stream=None, alignment_file=None, regions=None, front=None):
    """The front end of one shard of a sharded locus, with the pileup exchange (and, for CODIS D18S51, the inter-distance exchange)
    and the failure protocol -> (host batch or None, engine.DeviceBatch).  On a GPU the shard goes through the DEVICE front end and
    the exchange runs on the counter table in HBM (comm.allreduce_u32_dev; D18S51's inter-distance histogram, counted by the kernels,
    goes through the host form); `front="host"` and GPU-less processes take the host front end.  Collective: every rank of `comm` must call it."""
    from . import engine
    # A rank whose front-end fails must not leave its peers waiting in an exchange.  Every exchange of the parse carries one
    # extra element, the failure flag (sum > 0 = some rank failed: every rank raises out of that exchange and skips the later
    # ones); a rank that fails LOCALLY before an exchange its peers will enter contributes the flag to that one exchange; and
    # after the parse -- whatever happened -- every rank takes part in one status exchange, so that a failure behind the last
    # exchange of the parse reaches the peers before they enter the class-table gather.
    state = {"local": None, "remote": False, "n": 0}

    def exchange(arr):
        state["n"] += 1
        ext = np.zeros(arr.size + 1, np.int64)
        ext[:arr.size] = arr.astype(np.int64).ravel()
        comm.allreduce_sum(ext)
        if ext[-1] != 0:
            state["remote"] = True
            raise RuntimeError("another rank of this locus failed in its front-end")
        arr[...] = ext[:arr.size].reshape(arr.shape).astype(arr.dtype)
        return arr
    def exchange_u32(arr):                                  # pileup counters, host form: uint32 [L*6] + the flag
        state["n"] += 1
        ext = np.zeros(arr.size + 1, np.uint32)
        ext[:arr.size] = arr.ravel()
        comm.allreduce_u32(ext)
        if ext[-1] != 0:
            state["remote"] = True
            raise RuntimeError("another rank of this locus failed in its front-end")
        arr[...] = ext[:arr.size].reshape(arr.shape)
        return arr

    def exchange_u32_dev(dptr, n, st):                      # ... device form: the same collective on the table in HBM
        state["n"] += 1
        if comm.allreduce_u32_dev(dptr, n, st) != 0:
            state["remote"] = True
            raise RuntimeError("another rank of this locus failed in its front-end")
    d18 = pl.base_fname == "codis" and pl.gene == "D18S51"
    n_exchanges = 2 if d18 else 1
    on_device = front != "host" and capi_has_device()
    batch = db = None
    try:
        if on_device:
            # the DEVICE front end (round 5): record fields, filters, key grouping, pileup -- summed over the shards where the
            # kernels left it, in HBM -- decode, piece table and pair protocol as kernels; the batch is born in HBM.  A shard the
            # kernels decline (too few records, a record the reference would raise on) is finished by the host stages inside the
            # same call, through the host form of the same exchange.
            kw = dict(num_editdist=num_editdist, error_correction=error_correction, allow_discordant=allow_discordant,
                      simulation=simulation, base_locus=base_locus, stream=stream, pileup_exchange=exchange_u32,
                      pileup_exchange_dev=exchange_u32_dev if hasattr(comm, "allreduce_u32_dev") else None,
                      interdist_exchange=exchange if d18 else None, last_shard=comm.rank == comm.world - 1)
            if alignment_file is not None:
                db = pl.parse_alignment_file_dev(alignment_file, regions, **kw)
            else:
                db = pl.parse_sam_dev(sam_shard, **kw)
        else:
            # GPU-less processes (the gloo tests) and `front="host"`: the host front end
            batch = pl.parse_sam(sam_shard, num_editdist=num_editdist, error_correction=error_correction,
                                 allow_discordant=allow_discordant, simulation=simulation, base_locus=base_locus,
                                 pileup_exchange=exchange_u32, interdist_exchange=exchange if d18 else None,
                                 last_shard=comm.rank == comm.world - 1)
    except BaseException as e:
        state["local"] = e
        if not state["remote"] and state["n"] < n_exchanges:      # the peers are in (or heading for) the next exchange: flag it
            state["n"] += 1
            if state["n"] == 1:
                comm.allreduce_u32(_flag_only(pl, 1))
            else:
                comm.allreduce_sum(_flag_only(pl, 2))
    status = np.array([0 if state["local"] is None else 1], np.int64)
    try:
        comm.allreduce_sum(status)
        if state["local"] is not None:
            raise state["local"]
        if status[0] != 0:
            raise RuntimeError("another rank of this locus failed in its front-end")
        if db is None:
            db = engine.DeviceBatch(batch, stream)
    except BaseException:
        #  a batch this rank DID parse stays in HBM until the garbage collector finds it -- in a retry loop over loci or
        # samples that is memory growing exactly while other ranks are failing: give it back before the raise
        if db is not None:
            db.close()
        raise
    return batch, db


def type_locus_sharded(pl, sam_shard, comm, num_editdist=2, error_correction=True, allow_discordant=False,
                       remove_low_abundance_alleles=True, simulation=False, base_locus=0, stream=None, alignment_file=None, regions=None,
                       front=None):
    """Type one sample at one locus with its name-grouped reads split over the ranks of `comm` (shard r = the r-th
    consecutive stretch of the stream, cut at read boundaries: split_name_grouped).  Three exchanges, no other coupling:
      1. pileup counts, summed (get_mpileup covers the whole alignment; error correction needs all of it),
      2. the ranks' exon- and gene-level class tables, gathered and merged in rank order (= first-seen order),
      3. read / pair counts, summed.
    Every rank then runs the (small) EMs on the merged tables and returns the same LocusResult as the unsharded path.
    `alignment_file` (+ `regions`): this rank's shard as a SAM / BAM file instead of text.  `front="host"`: the host front end
    (what a process without a GPU takes anyway); engine.front_last() tells which route a rank's parse ran."""
    batch, db = parse_shard(pl, sam_shard, comm, num_editdist=num_editdist, error_correction=error_correction,
                            allow_discordant=allow_discordant, simulation=simulation, base_locus=base_locus, stream=stream,
                            alignment_file=alignment_file, regions=regions, front=front)
    try:
        return type_shard(pl, batch, db, comm, remove_low_abundance_alleles, stream)
    finally:
        db.close()


def _flag_only(pl, k):
    """The array a FAILED rank contributes to the k-th exchange of a sharded parse: zeros of the shape and type its peers send,
    with the failure flag set (exchange 1 = pileup counts, uint32 [L][6]; exchange 2 = the D18S51 inter-distance histogram, int64)."""
    if k == 1:
        ext = np.zeros(len(pl.ref_seq) * 6 + 1, np.uint32)
    else:
        ext = np.zeros(2 * 65536 + 2 + 1, np.int64)
    ext[-1] = 1
    return ext


def assign_ranks_to_loci(weights, world):
    """configs[2]-style jobs (a few big loci on more GPUs than loci): rank groups per locus, sizes proportional to the loci's
    weights, every locus at least one rank; with fewer ranks than loci the loci are packed instead (shard()).  Returns
    {locus index: [ranks]}; a locus with several ranks is typed with type_locus_sharded over that group."""
    n = len(weights)
    if world <= n:
        owner = {}
        for r in range(world):
            for i in shard(list(range(n)), r, world, weights):
                owner[i] = [r]
        return owner
    share = [1] * n
    for _ in range(world - n):
        i = max(range(n), key=lambda i: (weights[i] / share[i], -i))
        share[i] += 1
    out, nxt = {}, 0
    for i in range(n):
        out[i] = list(range(nxt, nxt + share[i]))
        nxt += share[i]
    return out
