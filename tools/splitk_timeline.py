#!/usr/bin/env python3
"""Per-workgroup time stamps of one iteration of the split-k FISTA kernel (csrc/fista_splitk.hip).  Needs the debug
build  tools/build_variant.sh splitk_t fista_splitk.hip -DLASSO_SPLITK_TIMING=50  (stamps of iteration 50).
usage: splitk_timeline.py [n] [kernel]   (default 2048 splitk4)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from lasso_amd import _native as nat
LIB = os.path.join(ROOT, 'variants', 'liblasso_splitk_t.so')
nat.use_library(LIB)
from lasso_amd.linear.solvers import ista
from recipes import recipe_xw
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
kern = sys.argv[2] if len(sys.argv) > 2 else 'splitk4'
T = {'splitk1': 1, 'splitk2': 2, 'splitk4': 4, 'splitk2g': 2, 'splitk4g': 4, 'splitk1s': 1}[kern]
X, W = recipe_xw(n, 256, 1024)
Xg, Wg = X.cuda(), W.cuda()
z0 = torch.zeros(n, 1024, device='cuda')
for _ in range(3):
    ista(Xg, z0, Wg, 0.5, lr=0.05, maxiter=100, tol=0.0, kernel=kern)
torch.cuda.synchronize()
lib = C.CDLL(LIB)
buf = (C.c_uint64 * (2048 * 32))()
assert lib.lasso_debug_splitk_stamps(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 32).astype(np.float64) / 100.0
used = t[:, 0] > 0
t = t[used]
print('workgroups with stamps:', int(used.sum()))
t0 = t[:, 0].copy()
def col(i): return t[:, i] - t0
names = [(0, 'iteration start')] + [(1 + i, 'tile %d GEMM-1 issued' % i) for i in range(T)] + [(5, 'all partials flagged'), (6, 'flags of every tile seen')]
for i in range(T):
    names += [(7 + 5 * i, 'tile %d partials summed' % i),
              (8 + 5 * i, 'tile %d r tile barrier' % i), (9 + 5 * i, 'tile %d GEMM-2 issued' % i)]
names += [(26, 'tiles done')]
print('%-28s %8s %8s %8s  (us from this workgroup entering the iteration)' % ('stamp', 'min', 'median', 'max'))
for i, nme in names:
    c = col(i)
    print('%-28s %8.2f %8.2f %8.2f' % (nme, c.min(), np.median(c), c.max()))
