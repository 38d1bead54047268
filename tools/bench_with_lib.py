"""bench.py (or another tool) against another build of the library: usage  bench_with_lib.py <path to .so> <script> [args] -- for A/B runs of
two builds on the same box (libmmdeer_var_*.so: build.build_variant, or a saved copy of an earlier build)."""
import os
import runpy
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmdeer import build
build.LIB_PATH = os.path.abspath(sys.argv[1])
build.needs_build = lambda: False
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
