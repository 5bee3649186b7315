"""The lab build (libhgx_lab.so: the opt-in EM back-ends kept out of the product library) still agrees with the product's default
path.  It is a different shared object, so its cases (tests/lab_cases.py) run in a child process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lab_backends_agree_with_the_product_path():
    from hisatgenotype_amd import capi
    assert os.path.exists(capi.LAB_PATH), "libhgx_lab.so is not built (__graft_entry__.build())"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "lab_cases.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]


def test_product_library_has_no_lab_backend():
    """The product library answers a request for a lab back-end with an error, not with a silent other path."""
    from hisatgenotype_amd import capi, engine
    with pytest.raises(capi.HgxError):
        engine.em_set_backend(2)
    engine.em_set_backend(0)


def test_product_library_has_no_fused_gene_level_form():
    """hgx_pair_classes_dedup (round 2's fused gene-level form, measured slower) lives in the lab library; libhgx.so says so."""
    import ctypes as C
    from hisatgenotype_amd import capi
    h = C.c_void_p()
    rc = capi.lib().hgx_pair_classes_dedup(C.byref(h), None, None, None, None, C.c_int32(0), C.c_int32(1), None, None)
    assert rc != 0 and b"lab" in capi.lib().hgx_last_error()
