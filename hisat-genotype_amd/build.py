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


def _deps(path, seen=None):
    """`path` and every local header it includes, directly or not (#include "..." resolved in csrc/, csrc/lab/ and include/)."""
    import re
    seen = set() if seen is None else seen
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    with open(path, errors="replace") as f:
        for m in re.finditer(r'^\s*#\s*include\s*"([^"]+)"', f.read(), re.M):
            for d in (os.path.dirname(path), CSRC, os.path.join(CSRC, "lab"), os.path.join(ROOT, "include")):
                _deps(os.path.join(d, m.group(1)), seen)
    return seen


def _object_fresh(src, newest_header=None):
    """An object is reused only if it is newer than its source AND every header that source includes (struct layouts are shared
    between units)."""
    obj = src.rsplit(".", 1)[0] + ".o"
    return os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(d) for d in _deps(src))


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


LAB_UNITS = ["hgx_em.hip", "hgx_device.hip", "hgx_dedup.hip", "hgx_type.hip", "hgx_emx.hip", "hgx_front_host.cpp"]      # the units with #ifdef HGX_LAB code


def build_lab(force=False, verbose=True):
    """libhgx_lab.so: the product objects with the LAB_UNITS recompiled under -DHGX_LAB -- hgx_em.hip (the int8-MFMA, persistent and
    resident-grid EM back-ends in csrc/lab/*.inc) and hgx_front_host.cpp (the CPU emulation of the device front end's kernels, which
    the CPU test-suite compares with the host front end).  Lab tools and tests bind it (capi.use_lab()); nothing in the product
    path does."""
    build(force=force, verbose=verbose)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    incs = [os.path.join(LAB_DIR, f) for f in os.listdir(LAB_DIR) if f.endswith(".inc")]
    lab_objs = []
    for unit in LAB_UNITS:
        src = os.path.join(CSRC, unit)
        obj = os.path.join(LAB_DIR, unit.rsplit(".", 1)[0] + "_lab.o")
        newest = max(os.path.getmtime(d) for d in list(_deps(src)) + incs)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest:
            if unit.endswith(".hip"):
                cmd = [hipcc, "--offload-arch=" + ARCH]
            else:
                cmd = [os.environ.get("CXX", "g++"), "-pthread"]
            cmd += ["-DHGX_LAB", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
                    "-I", CSRC, "-Wall", "-Wno-unused-result", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
        lab_objs.append(obj)
    if force or not os.path.exists(LAB_LIB) or os.path.getmtime(LAB_LIB) < max([os.path.getmtime(o) for o in lab_objs] + [os.path.getmtime(LIB)]):
        objs = [s_.rsplit(".", 1)[0] + ".o" for s_ in _sources() if os.path.basename(s_) not in LAB_UNITS] + lab_objs
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-pthread", "-o", LAB_LIB] + objs + ["-lz", "-ldl"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LAB_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    if "--lab" in sys.argv:
        build_lab(force="--force" in sys.argv)
