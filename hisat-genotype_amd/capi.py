"""ctypes binding of libhgx.so (include/hgx.h).  No fallback: if the library is missing or a
call fails, an exception is raised."""
import ctypes as C
import threading
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libhgx.so")
LAB_PATH = os.path.join(HERE, "csrc", "lab", "libhgx_lab.so")     # same sources with -DHGX_LAB (build.build_lab): lab tools / tests only

VAR_INSERTION, VAR_SINGLE, VAR_DELETION = 0, 1, 2
BASE_KIND = {"hla": 0, "codis": 1, "genome": 2}


class HgxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libhgx error %d: %s" % (code, msg))
        self.code = code


class HgxKeyError(HgxError, KeyError):
    """The reference raises KeyError at this point (EM quirk Q6)."""


class Piece(C.Structure):
    _fields_ = [("mask_off", C.c_uint32), ("lo_word", C.c_uint16), ("n_words", C.c_uint16)]


PIECE_DTYPE = np.dtype([("mask_off", np.uint32), ("lo_word", np.uint16), ("n_words", np.uint16)])


class LocusDesc(C.Structure):
    _fields_ = [("base_kind", C.c_int32), ("backbone_len", C.c_int32), ("backbone", C.c_char_p),
                ("n_vars", C.c_int32), ("var_pos", C.c_void_p), ("var_type", C.c_void_p), ("var_len", C.c_void_p),
                ("var_base", C.c_void_p), ("var_linked", C.c_void_p), ("var_name_pool", C.c_char_p),
                ("var_ins_pool", C.c_char_p), ("n_alleles", C.c_int32), ("link_off", C.c_void_p),
                ("link_allele", C.c_void_p), ("n_link_order", C.c_int32), ("link_order", C.c_void_p),
                ("n_exons", C.c_int32), ("exons", C.c_void_p), ("allele_len", C.c_void_p), ("name_rank", C.c_void_p)]


class ParseOpts(C.Structure):
    _fields_ = [("num_editdist", C.c_int32), ("error_correction", C.c_int32), ("allow_discordant", C.c_int32),
                ("simulation", C.c_int32), ("base_locus", C.c_int32), ("keep_trace", C.c_int32),
                ("codis_choose_pairs", C.c_int32), ("n_threads", C.c_int32),
                ("pileup_exchange", C.c_void_p), ("pileup_ctx", C.c_void_p),
                ("interdist_exchange", C.c_void_p), ("interdist_ctx", C.c_void_p),
                ("pileup_exchange_dev", C.c_void_p), ("pileup_dev_ctx", C.c_void_p)]


PILEUP_EXCHANGE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.c_int64)
PILEUP_EXCHANGE_DEV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
INTERDIST_EXCHANGE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int64)


# every symbol include/hgx.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "hgx_last_error", "hgx_version", "hgx_device_count", "hgx_set_device", "hgx_dev_alloc", "hgx_dev_free",
    "hgx_memcpy_h2d", "hgx_memcpy_h2d_async", "hgx_memcpy_d2h", "hgx_memset", "hgx_stream_sync", "hgx_stream_create", "hgx_stream_create_prio", "hgx_stream_destroy", "hgx_pool_trim", "hgx_event_create", "hgx_event_destroy",
    "hgx_event_record", "hgx_stream_wait_event", "hgx_event_elapsed_ms", "hgx_a_pad", "hgx_index_create",
    "hgx_index_destroy", "hgx_index_dims", "hgx_index_device_bits", "hgx_piece_compat", "hgx_pair_classes",
    "hgx_score_pairs", "hgx_level_classes", "hgx_group_pairs", "hgx_groups_dims", "hgx_groups_destroy",
    "hgx_level_classes_grouped", "hgx_classes_set_allele_rank", "hgx_dedup_classes", "hgx_classes_destroy", "hgx_classes_dims", "hgx_classes_device",
    "hgx_classes_to_host", "hgx_classes_from_host", "hgx_allele_counts", "hgx_allele_counts_on", "hgx_first_classes", "hgx_em", "hgx_em_ordered", "hgx_em_masked", "hgx_em_set_backend", "hgx_test_switch_set", "hgx_debug_matvec", "hgx_em_set_timing", "hgx_em_get_timing", "hgx_locus_create",
    "hgx_locus_destroy", "hgx_locus_dims", "hgx_locus_tables", "hgx_index_from_locus",
    "hgx_locus_alternatives_text", "hgx_batch_destroy", "hgx_batch_dims", "hgx_batch_arrays",
    "hgx_batch_from_haplotypes", "hgx_parse_sam", "hgx_read_alignments", "hgx_free_text", "hgx_parse_alignment_file", "hgx_batch_trace_text", "hgx_batch_pileup",
    "hgx_dbatch_create", "hgx_dbatch_destroy", "hgx_dbatch_dims", "hgx_gate_create", "hgx_gate_destroy", "hgx_type_dbatch",
    "hgx_type_batch", "hgx_type_file", "hgx_typing_destroy", "hgx_typing_dims", "hgx_typing_counts", "hgx_typing_em",
    "hgx_typing_gene_prob", "hgx_typing_classes", "hgx_write_bam", "hgx_em_last_exact", "hgx_index_device_block", "hgx_index_create_device", "hgx_type_classes", "hgx_pair_classes_dedup",
    "hgx_bgzf_inflate", "hgx_bgzf_scan_compare", "hgx_many_create", "hgx_many_create_files", "hgx_many_create_sams", "hgx_many_tasks", "hgx_many_destroy", "hgx_many_dims", "hgx_type_many", "hgx_type_many_loci", "hgx_em_set_fast", "hgx_em_last_order", "hgx_typing_top", "hgx_emx_set_timing", "hgx_emx_get_timing",
    "hgx_index_broadcast", "hgx_allreduce_sum_u32", "hgx_allreduce_sum_i64", "hgx_classes_allgather",
    "hgx_classes_pack_rows", "hgx_classes_merge_gathered", "hgx_rccl_stats",
    "hgx_keyset_create", "hgx_keyset_dims", "hgx_keyset_fill", "hgx_keyset_destroy",
    "hgx_many_from_dbatch", "hgx_alignment_open", "hgx_alignment_dims", "hgx_alignment_parse_dev", "hgx_alignment_close", "hgx_stream_sets_info", "hgx_stream_create_placed", "hgx_stream_probe_matrix", "hgx_stream_probe_pair", "hgx_stream_probe_chain", "hgx_stream_sets_streams",
    "hgx_parse_sam_dev", "hgx_parse_alignment_file_dev", "hgx_front_last", "hgx_front_last_parts", "hgx_dbatch_to_host",
    "hgx_emx_cluster_stats", "hgx_em_tie_reruns",
]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libhgx.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        _lib = C.CDLL(LIB_PATH)
        _lib.hgx_last_error.restype = C.c_char_p
        _lib.hgx_a_pad.restype = C.c_int32
    return _lib


def use_lab():
    """Bind the lab build (opt-in EM back-ends of rounds 1-2) instead of the product library; before the first call only."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("libhgx.so is already loaded in this process")
    LIB_PATH = LAB_PATH


def check(rc):
    if rc != 0:
        msg = lib().hgx_last_error().decode(errors="replace")
        if rc == -4:
            raise HgxKeyError(rc, msg)
        raise HgxError(rc, msg)


def ptr(a):
    if a is None:
        return None
    if hasattr(a, "ptr") and not hasattr(a, "ctypes"):      # DevArray or a borrowed device pointer
        return C.c_void_p(a.ptr)
    return a.ctypes.data_as(C.c_void_p)


def a_pad(n_alleles):
    return int(lib().hgx_a_pad(C.c_int32(n_alleles)))


def device_count():
    n = C.c_int(0)
    check(lib().hgx_device_count(C.byref(n)))
    return n.value


_current_device = 0
_streams = {}


def set_device(dev):
    """hipSetDevice for the calling thread (HIP's current device is per thread)."""
    global _current_device
    check(lib().hgx_set_device(C.c_int(dev)))
    _current_device = dev


def current_device():
    return _current_device


_slot = threading.local()


def set_stream_slot(k):
    """Streams are cached per (device, slot, index); a slot defaults to the calling thread.  Worker threads that come and go
    (one generation for warm-up, one for the timed region, ...) name a stable slot so that they reuse the streams -- creating a
    stream means creating a hardware queue, which costs milliseconds."""
    _slot.k = ("slot", k)


def get_stream(i):
    """A cached non-blocking stream of the current device (created on first use).  Stream 0 (the EM chain: short
    dependent launches on the critical path) has the highest priority, stream 1 (overlapped side work) the lowest."""
    key = (_current_device, getattr(_slot, "k", threading.get_ident()), i)     # per host thread (or named slot): samples in flight never share a stream
    if key not in _streams:
        p = C.c_void_p()
        # worker main streams (i == 2) run chains of short kernels (the device front end): placed on the lane with the fewest chains
        fn = lib().hgx_stream_create_placed if i == 2 else lib().hgx_stream_create_prio
        check(fn(C.byref(p), C.c_int(1 if i == 0 else 0)))
        _streams[key] = p
    return _streams[key]


class DevArray:
    """A typed device allocation owned by Python (hipMalloc through the C-ABI)."""

    def __init__(self, shape, dtype):
        self.shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        check(lib().hgx_dev_alloc(C.byref(p), C.c_size_t(self.nbytes)))
        self.ptr = p.value

    @staticmethod
    def from_host(arr, stream=None, sync=True):
        """Upload `arr`.  sync=False: no host round trip (small arrays are staged through pinned memory and travel in
        stream order; the result may only be used on `stream`)."""
        arr = np.ascontiguousarray(arr)
        d = DevArray(arr.shape, arr.dtype)
        if arr.nbytes:
            fn = lib().hgx_memcpy_h2d if sync else lib().hgx_memcpy_h2d_async
            check(fn(C.c_void_p(d.ptr), ptr(arr), C.c_size_t(arr.nbytes), stream))
        return d

    def to_host(self, stream=None):
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            check(lib().hgx_memcpy_d2h(ptr(out), C.c_void_p(self.ptr), C.c_size_t(self.nbytes), stream))
        return out

    def zero(self, stream=None):
        check(lib().hgx_memset(C.c_void_p(self.ptr), 0, C.c_size_t(self.nbytes), stream))

    def free(self):
        if self.ptr:
            lib().hgx_dev_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def sync(stream=None):
    check(lib().hgx_stream_sync(stream))


class Event:
    """hipEvent on a stream (timing of individual kernels)."""

    def __init__(self):
        p = C.c_void_p()
        check(lib().hgx_event_create(C.byref(p)))
        self.h = p

    def record(self, stream=None):
        check(lib().hgx_event_record(self.h, stream))

    def make_wait(self, stream):
        """Work queued on `stream` from now on waits for this event (device-side dependency, no host sync)."""
        check(lib().hgx_stream_wait_event(stream, self.h))

    def elapsed_ms(self, stop):
        ms = C.c_float(0)
        check(lib().hgx_event_elapsed_ms(self.h, stop.h, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            lib().hgx_event_destroy(self.h)
        except Exception:
            pass
