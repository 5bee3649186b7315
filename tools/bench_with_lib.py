#!/usr/bin/env python3
"""bench.py against ANOTHER build of libhgx (an A/B on the same box): tools/bench_with_lib.py <path/to/libhgx.so> [bench.py flags].
The library path is set before anything loads it; everything else is bench.py's own main()."""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hisatgenotype_amd                      # registers the package directory
from hisatgenotype_amd import capi
capi.LIB_PATH = os.path.abspath(sys.argv[1])
os.environ["HGX_BENCH_LIB"] = capi.LIB_PATH        # (bench.py's child processes load it too)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
