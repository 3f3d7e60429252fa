#!/usr/bin/env python3
"""Per-workgroup time stamps of the fp32 line search's one-launch-per-iteration kernel (csrc/bt_iter.hip) at BASELINE
config 3: the first tile of every workgroup in the last full launch of a solve.  Needs the debug build
  tools/build_variant.sh bti_t bt_iter.hip -DLASSO_BTI_TIMING=0      (=j: stamps of every workgroup's j-th tile)
Prints min / median / max of each phase over the 256 workgroups, in microseconds."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
from lasso_amd import _native as nat
LIB = os.path.join(ROOT, 'variants', sys.argv[1] if len(sys.argv) > 1 else 'liblasso_bti_t.so')
nat.use_library(LIB)
from lasso_amd.linear.solvers import ista
from recipes import recipe_xw

X, W = recipe_xw(16384, 256, 1024)
Xg, Wg = X.cuda(), W.cuda()
z0 = torch.zeros(16384, 1024, device='cuda')
for _ in range(4):
    ista(Xg, z0, Wg, 0.5, lr=1.0, maxiter=10, tol=0.0, backtrack=True)
torch.cuda.synchronize()
lib = C.CDLL(LIB)
buf = (C.c_uint64 * (1024 * 16))()
assert lib.lasso_debug_bti_stamps(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16)[:256].astype(np.float64) / 100.0   # 100 MHz -> us
t0 = t[:, 0].min()
names = ['tile start', 'accept done (stores issued)', 'barrier', 'GEMM-1 done', 'r tile + fragments', 'GEMM-2 done',
         'p in registers + barrier', 'trial0 candidate written', 'trial0 barrier', 'trial0 GEMM-1 done', 'trial0 end',
         'trials but the last done', 'tile end', 'kernel end', 'last GEMM-1 issued']
print('%-30s %8s %8s %8s   (us from the first workgroup entering the launch)' % ('stamp', 'min', 'median', 'max'))
for i, nme in enumerate(names):
    print('%-30s %8.2f %8.2f %8.2f' % (nme, t[:, i].min() - t0, np.median(t[:, i]) - t0, t[:, i].max() - t0))
print()
print('%-34s %7s %7s %7s' % ('phase (per workgroup, first tile)', 'min', 'median', 'max'))
for a, b, label in ((0, 1, 'accept: loads, math, stores'), (1, 2, 'barrier'), (2, 3, 'GEMM-1 (gradient)'),
                    (3, 4, 'r tile, barrier, fragments'), (4, 5, 'GEMM-2'), (5, 6, 'p -> registers, barrier'),
                    (6, 7, 'candidate 0 in registers'), (7, 9, 'trial0 cand. write, barrier, GEMM-1'),
                    (9, 10, 'candidate 1 + trial0 sums, barrier'), (6, 11, 'all trials but the last'), (11, 14, 'g stores, prefetch, cand. write, barrier, last GEMM-1'),
                    (14, 12, 'last sums, barrier'),
                    (0, 12, 'whole tile'), (0, 13, 'whole launch (4 tiles)')):
    dt = t[:, b] - t[:, a]
    print('%-34s %7.2f %7.2f %7.2f' % (label, dt.min(), np.median(dt), dt.max()))
