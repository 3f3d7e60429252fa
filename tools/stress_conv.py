#!/usr/bin/env python3
"""Seeded sweep over random geometries of the convolutional solver (image channels, atoms either side of the 64 /
128-atom tiles of the gradient kernel, kernel sizes, strides, paddings, ragged code grids), 6 FISTA and 6 ISTA
iterations each against the CPU oracle.  Not part of the test suite; prints the worst deviation per family of
kernels (which synthesis kernel the geometry takes; whether the gradient kernel may pick its 64- / 32-atom tiles).  usage: stress_conv.py [trials]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
import torch
from lasso_amd.conv2d import ista_conv2d
from oracle import lasso_oracle as orc

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 120
g = torch.Generator().manual_seed(2024)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
worst, count = {}, {}
t0 = time.time()
for trial in range(trials):
    C = [1, 1, 2, 3, 3, 4, 6, 8, 12, 16, 17][ri(0, 10)]
    K = [ri(1, 20), ri(21, 64), 64, ri(65, 128), ri(129, 260)][ri(0, 4)]
    ks = [1, 3, 3, 5, 5, 7, 4][ri(0, 6)]
    kh, kw = (ks, ks) if ri(0, 5) else (ks, max(1, ks - 2))
    st = 1 if ri(0, 3) else 2
    ph, pw = ri(0, kh - 1), ri(0, kw - 1)
    Hz, Wz = ri(1, 30), [ri(1, 30), ri(31, 70)][ri(0, 3) == 0]
    N = ri(1, 6)
    H, W_ = (Hz - 1) * st - 2 * ph + kh, (Wz - 1) * st - 2 * pw + kw
    if H < 1 or W_ < 1:
        continue
    w = torch.randn(K, C, kh, kw, generator=g) / (kh * kw) ** 0.5
    x = torch.randn(N, C, H, W_, generator=g)
    z0 = torch.randn(N, K, Hz, Wz, generator=g) * 0.05
    lr = 0.3 / max(w.pow(2).sum().item(), 1e-3)
    ckk = C * kh * kw
    fam = ("grad:" + ("explicit" if ckk > 192 else ("narrow-tiles-allowed" if K <= 64 else "tile128")) + " synth:" +
           ("synth" if (8 <= C <= 16 and st == 1 and kh == kw and kh in (3, 5, 7)) else
            "few" if (C < 8 and st == 1 and 4 <= K <= 128 and K % 4 == 0 and ckk <= 128) else "explicit"))
    for fast in (True, False):
        ref = orc.conv_fista(x, z0, w, 0.1, stride=st, padding=(ph, pw), fast=fast, maxiter=6, lr=lr, tol=0.0)
        got = ista_conv2d(x.cuda(), z0.cuda(), w.cuda(), 0.1, stride=st, padding=(ph, pw), fast=fast, maxiter=6,
                          lr=lr, tol=0.0)
        err = (got.cpu() - ref).abs().max().item()
        if err > worst.get(fam, (-1, ""))[0]:
            worst[fam] = (err, (N, C, K, kh, kw, st, (ph, pw), Hz, Wz))
    count[fam] = count.get(fam, 0) + 1
print("geometries", sum(count.values()), "seconds", round(time.time() - t0, 1))
for fam in sorted(worst):
    print("%-34s n=%-3d worst max|dz| %.3e at (N, C, K, kh, kw, stride, padding, Hz, Wz) = %s" %
          (fam, count[fam], worst[fam][0], worst[fam][1]))
bad = [f for f in worst if worst[f][0] > 5e-5]
print("ALL WITHIN 5e-5" if not bad else "ABOVE 5e-5: %s" % bad)
