"""Build libhgx.so (HIP kernels + C-ABI + host front-end) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this also runs in the build container.  The shared object
lands next to the sources (hisat-genotype_amd/csrc/libhgx.so), is git-ignored and travels to the
GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libhgx.so")
ARCH = "gfx950"


def _sources():
    out = []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".cpp")):
            out.append(os.path.join(CSRC, f))
    return out


def _headers():
    return [os.path.join(ROOT, "include", "hgx.h")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]


def _object_fresh(src, newest_header):
    """An object is reused only if it is newer than its source AND every header (struct layouts are shared between units)."""
    obj = src.rsplit(".", 1)[0] + ".o"
    return os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(src) and os.path.getmtime(obj) > newest_header


def needs_build():
    if not os.path.exists(LIB):
        return True
    newest_header = max(os.path.getmtime(h) for h in _headers())
    t = os.path.getmtime(LIB)
    for src in _sources():
        if not _object_fresh(src, newest_header) or os.path.getmtime(src.rsplit(".", 1)[0] + ".o") > t:
            return True
    return False


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    newest_header = max(os.path.getmtime(h) for h in _headers())
    for src in _sources():
        obj = src.rsplit(".", 1)[0] + ".o"
        if not force and _object_fresh(src, newest_header):
            objs.append(obj)
            continue
        if src.endswith(".hip"):
            cmd = [hipcc, "--offload-arch=" + ARCH]
        else:                                    # host-only translation units: plain C++ (hipcc would also run a device pass)
            cmd = [os.environ.get("CXX", "g++"), "-pthread"]
        cmd += ["-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
                "-I", CSRC, "-Wall", "-Wno-unused-result", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-pthread", "-o", LIB] + objs + ["-lz", "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


LAB_DIR = os.path.join(CSRC, "lab")
LAB_LIB = os.path.join(LAB_DIR, "libhgx_lab.so")


def build_lab(force=False, verbose=True):
    """libhgx_lab.so: the product objects with hgx_em.hip recompiled under -DHGX_LAB (the only unit that has lab code: the int8-MFMA,
    persistent and resident-grid EM back-ends in csrc/lab/*.inc).  Lab tools and tests/test_gpu_lab.py bind it (capi.use_lab());
    nothing in the product path does."""
    build(force=force, verbose=verbose)
    src = os.path.join(CSRC, "hgx_em.hip")
    obj = os.path.join(LAB_DIR, "hgx_em_lab.o")
    deps = _headers() + [src] + [os.path.join(LAB_DIR, f) for f in os.listdir(LAB_DIR) if f.endswith(".inc")]
    newest = max(os.path.getmtime(d) for d in deps)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest:
        cmd = [hipcc, "--offload-arch=" + ARCH, "-DHGX_LAB", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
               "-I", CSRC, "-Wall", "-Wno-unused-result", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    if force or not os.path.exists(LAB_LIB) or os.path.getmtime(LAB_LIB) < max(os.path.getmtime(obj), os.path.getmtime(LIB)):
        objs = [s_.rsplit(".", 1)[0] + ".o" for s_ in _sources() if os.path.basename(s_) != "hgx_em.hip"] + [obj]
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-pthread", "-o", LAB_LIB] + objs + ["-lz", "-ldl"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LAB_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    if "--lab" in sys.argv:
        build_lab(force="--force" in sys.argv)
