"""The wave-wide reductions of hgx_common.hpp (permlane-swap / DPP butterflies) are bit-identical to the __shfl_xor form."""
import os
import subprocess
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_butterfly_reductions_bit_identical_to_shuffle_form():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "wrt")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.join(ROOT, "hisat-genotype_amd", "csrc"), "-o", exe,
                        os.path.join(ROOT, "tools", "wave_reduce_test.hip")], check=True, capture_output=True, timeout=300)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "bad=0" in r.stdout
