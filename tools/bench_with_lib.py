#!/usr/bin/env python3
"""bench.py on another build of the library (variants/liblasso_<name>.so, tools/build_variant.sh):
usage: tools/bench_with_lib.py <lib.so> <bench.py arguments ...>"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd")]
from lasso_amd import _native as nat
nat.use_library(os.path.abspath(sys.argv[1]))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
