#!/usr/bin/env python3
"""bench.py's N=1 FISTA leg against an A/B build of the library.  usage: bench_variant.py <lib.so> [bench.py args]"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd")]
from lasso_amd import _native as nat
nat.use_library(os.path.abspath(sys.argv[1]))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
