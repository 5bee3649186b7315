"""Create a one-rank RCCL communicator through ctypes (what dist.RcclComm does) with NCCL_DEBUG=INFO: bootstrap diagnostics."""
import ctypes as C, os, sys, time
os.environ["NCCL_DEBUG"] = "INFO"
for k, v in [a.split("=", 1) for a in sys.argv[1:]]:
    os.environ[k] = v
R = C.CDLL("librccl.so.1", mode=C.RTLD_GLOBAL)
class Uid(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]
uid = Uid()
print("getUniqueId", R.ncclGetUniqueId(C.byref(uid)), flush=True)
h = C.c_void_p()
R.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
t0 = time.time()
rc = R.ncclCommInitRank(C.byref(h), 1, uid, 0)
print("commInitRank", rc, "%.1fs" % (time.time() - t0), flush=True)
