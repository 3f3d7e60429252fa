#!/usr/bin/env python3
"""Per-iteration wall-clock durations of the fused FISTA kernel's first tile (all workgroups), to see what a launch
costs before / after its iterations.  Needs the debug build
  tools/build_variant.sh fista_t fista_tile_sp.hip -DLASSO_FISTA_TIMING
usage: fista_timeline.py [rows] [iterations] [zero|given]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
from lasso_amd import _native as nat
LIB = os.path.join(ROOT, 'variants', 'liblasso_fista_t.so')
nat.use_library(LIB)
from lasso_amd.linear.solvers import ista
from lasso_amd.linear import sparse_encode
from recipes import recipe_xw, LAMBDA_MAX_C2

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
M = int(sys.argv[2]) if len(sys.argv) > 2 else 10
given = len(sys.argv) > 3 and sys.argv[3] == "given"
X, W = recipe_xw(n)
Xg, Wg = X.cuda(), W.cuda()
z0 = torch.zeros(n, 1024, device='cuda')
f = (lambda: ista(Xg, z0, Wg, 0.5, lr=1.0 / LAMBDA_MAX_C2, maxiter=M, tol=0.0)) if given else \
    (lambda: sparse_encode(Xg, Wg, alpha=0.5, lr=1.0 / LAMBDA_MAX_C2, maxiter=M, tol=0.0))
for _ in range(6):
    f()
torch.cuda.synchronize()
lib = C.CDLL(LIB)
buf = (C.c_uint64 * (1024 * 64))()
assert lib.lasso_debug_fista_stamps(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 64)[:256, :M + 4].astype(np.float64) / 100.0
t0 = t[:, 0].min()
print("n=%d, %d iterations, z0 %s: us from the first workgroup's entry (min / median / max over 256 workgroups)" % (n, M, "given" if given else "NULL"))
names = ["entry", "first tile staged"] + ["iteration %d starts" % i for i in range(M)] + ["iterations done", "tile written"]
for i, nm in enumerate(names):
    print("%-22s %8.2f %8.2f %8.2f" % (nm, t[:, i].min() - t0, np.median(t[:, i]) - t0, t[:, i].max() - t0))
d = np.diff(t[:, 2:M + 3], axis=1)
print("per-iteration durations (median over workgroups):", " ".join("%.1f" % v for v in np.median(d, axis=0)))
