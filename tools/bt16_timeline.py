#!/usr/bin/env python3
"""Per-workgroup time stamps of one outer iteration of the single-launch bf16 line search
(csrc/bt16_persist.hip) at BASELINE config 3.  Needs the debug build
  tools/build_variant.sh bt16_t bt16_persist.hip -DLASSO_BT16_TIMING=5     (stamps of outer iteration 5)
Prints, over the 256 workgroups, min / median / max of each phase in microseconds and the spread of
the absolute stamps (who is late)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
from lasso_amd import _native as nat
LIB = os.path.join(ROOT, 'variants', sys.argv[1] if len(sys.argv) > 1 else 'liblasso_bt16_t.so')
nat.use_library(LIB)
from lasso_amd.linear.solvers import ista
from recipes import recipe_xw

X, W = recipe_xw(16384, 256, 1024)
Xg, Wg = X.cuda().bfloat16(), W.cuda().bfloat16()
z0 = torch.zeros(16384, 1024, device='cuda', dtype=torch.bfloat16)
for _ in range(5):
    ista(Xg, z0, Wg, 0.5, lr=1.0, maxiter=10, tol=0.0, backtrack=True)
torch.cuda.synchronize()
lib = C.CDLL(LIB)
buf = (C.c_uint64 * (1024 * 16))()
assert lib.lasso_debug_bt16_stamps(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16)[:256].astype(np.float64) / 100.0   # 100 MHz -> us
t -= t[:, 0].min()
names = ['iter start', 'gradient done', 'trial1 start', 'trial1 issue', 'trial1 decide begin', 'trial1 decide end',
         'trial1 passes end', 'trial1 published', 'trial2 start', 'accept start', 'accept end']
print('%-22s %8s %8s %8s   (absolute, us from the first workgroup entering the iteration)' % ('stamp', 'min', 'median', 'max'))
for i, nme in enumerate(names):
    print('%-22s %8.2f %8.2f %8.2f' % (nme, t[:, i].min(), np.median(t[:, i]), t[:, i].max()))
print()
def phase(a, b, label):
    dt = t[:, b] - t[:, a]
    print('%-34s %7.2f %7.2f %7.2f' % (label, dt.min(), np.median(dt), dt.max()))
print('%-34s %7s %7s %7s' % ('phase (per workgroup)', 'min', 'median', 'max'))
phase(0, 1, 'gradient (GEMM-1 + GEMM-2)')
phase(0, 11, '  GEMM-1')
phase(11, 12, '  residual -> LDS, barrier')
phase(12, 13, '  GEMM-2 up to the last g stores')
phase(13, 1, '  last g stores + barrier')
phase(2, 3, 'trial1: passes before the sweep loads')
phase(3, 4, 'trial1: double pass after issue')
phase(4, 5, 'trial1: decide (verdict of trial0)')
phase(5, 6, 'trial1: remaining passes')
phase(6, 7, 'trial1: publish')
phase(7, 8, 'trial1 end -> trial2 start')
phase(9, 10, 'accept')
late = np.argsort(-t[:, 7])[:8]
print('latest publishers of trial1 (workgroup, XCD = wg % 8):', [(int(w), int(w) % 8) for w in late])
xcd = np.array([t[np.arange(256) % 8 == x, 6].mean() - t[np.arange(256) % 8 == x, 5].mean() for x in range(8)])
print('mean "remaining passes" per XCD:', np.round(xcd, 2))
