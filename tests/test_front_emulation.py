"""The device front end's per-key / per-pair logic (csrc/hgx_front_core.hpp), emulated on the CPU by the lab library, against the
pinned host front end: tests/lab_front_cases.py in a child process (the lab library is a different shared object).  The kernels
themselves are compared with the host front end on the GPU box (tests/test_gpu_front.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_emulated_device_front_end_equals_the_host_front_end():
    from hisatgenotype_amd import capi
    assert os.path.exists(capi.LAB_PATH), "libhgx_lab.so is not built (__graft_entry__.build())"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "lab_front_cases.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-6000:] + r.stderr[-2000:]
