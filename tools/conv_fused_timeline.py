#!/usr/bin/env python3
"""Phase timeline of the one-launch convolutional iteration (conv_fused.hip), wave 0 of every workgroup, first image.
Needs the debug build:  tools/build_variant.sh cf_t conv_fused.hip -DLASSO_CF_TIMING
usage: conv_fused_timeline.py [N C K ksize padding code_hw]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd")]
import numpy as np
import torch
from lasso_amd import _native as nat
LIB = os.path.join(ROOT, "variants", "liblasso_cf_t.so")
nat.use_library(LIB)
from lasso_amd.conv2d import ista_conv2d

N, Cc, K, ks, pd, Hz = [int(a) for a in sys.argv[1:7]] if len(sys.argv) > 6 else (256, 1, 64, 7, 0, 26)
g = torch.Generator().manual_seed(0)
w = torch.randn(K, Cc, ks, ks, generator=g) / ks
H = (Hz - 1) - 2 * pd + ks
x = torch.randn(N, Cc, H, H, generator=g)
lr = 0.5 / w.pow(2).sum().item()
xg, wg, zg = x.cuda(), w.cuda(), torch.zeros(N, K, Hz, Hz, device="cuda")
for _ in range(3):
    ista_conv2d(xg, zg, wg, 0.1, stride=1, padding=pd, maxiter=20, lr=lr, tol=0.0)
torch.cuda.synchronize()
lib = C.CDLL(LIB)
buf = (C.c_uint64 * (1024 * 64))()
assert lib.lasso_debug_cf_stamps(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 64).astype(np.float64) / 100.0
used = 256 if N < 256 else min(N, 256)
t = t[:used]
t0 = t[:, 1].min()
names = {1: "image starts", 26: "residual image written", 27: "gradient fragments loaded",
         29: "gradient blocks done (wave 0)", 30: "sum reduced"}
for b in range(12):
    names[32 + b] = "wave 0 starts its gradient block %d" % b
for w_ in range(8):
    names[48 + w_] = "wave %d leaves the gradient phase" % w_
for c in range(8):
    names[2 + 3 * c] = "chunk %d: first A loads issued" % c
    names[3 + 3 * c] = "chunk %d: COLS stored" % c
    names[4 + 3 * c] = "chunk %d: taps added" % c
print("N=%d C=%d K=%d %dx%d pad %d code %dx%d: us from the first workgroup's entry (min / median / max over %d workgroups), last iteration"
      % (N, Cc, K, ks, ks, pd, Hz, Hz, used))
prev = None
for slot in sorted(names, key=lambda k: (np.median(t[:, k]), k)):
    col = t[:, slot]
    if not (col > 0).all() or col.max() < t0:
        continue
    med = np.median(col) - t0
    print("%-34s %8.2f %8.2f %8.2f   %s" % (names[slot], col.min() - t0, med, col.max() - t0,
                                             "" if prev is None else "+%.2f" % (med - prev)))
    prev = med
