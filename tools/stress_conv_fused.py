#!/usr/bin/env python3
"""Seeded sweep over random geometries of the whole-iterations-in-one-kernel form of the convolutional solver
(conv_fused.hip): whole images (N >= CUs) and banded ones (N < CUs), 1-7 channels, 4-128 atoms, kernel sizes 1-7 (square
and not), paddings, ragged code grids.  Each geometry: 7 FISTA and 7 ISTA iterations against the CPU oracle on the first
images, and BITWISE against the two-kernel form (LASSO_CONV_FUSED=0) on all of them.  Not part of the test suite.
usage: stress_conv_fused.py [trials]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
import torch
from lasso_amd import _native as nat
from lasso_amd.conv2d import ista_conv2d
from oracle import lasso_oracle as orc

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cus = torch.cuda.get_device_properties(0).multi_processor_count
g = torch.Generator().manual_seed(515)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
worst, count, notbit, skipped = {}, {}, [], 0
t0 = time.time()
for trial in range(trials):
    C = [1, 1, 1, 2, 3, 3, 4, 7][ri(0, 7)]
    K = 4 * [ri(1, 4), ri(5, 8), ri(9, 16), 16, ri(17, 32)][ri(0, 4)]
    ks = [1, 3, 3, 5, 5, 7, 4][ri(0, 6)]
    kh, kw = (ks, ks) if ri(0, 4) else (ks, max(1, ks - 2))
    ph, pw = ri(0, kh - 1), ri(0, kw - 1)
    banded = ri(0, 2) == 0
    Hz, Wz = (ri(8, 48), ri(4, 40)) if banded else (ri(1, 30), ri(1, 34)) if ri(0, 3) else (ri(20, 64), ri(20, 64))
    N = [cus // 2, cus // 3, cus // 4, cus - 1][ri(0, 3)] if banded else cus + ri(0, 40) if ri(0, 2) else cus - ri(1, 160)
    H, W_ = (Hz - 1) - 2 * ph + kh, (Wz - 1) - 2 * pw + kw
    if H < 1 or W_ < 1:
        continue
    name = nat.lib().lasso_conv_ista_kernel_name(N, C, H, W_, K, Hz, Wz, kh, kw, 1, 1, ph, pw)
    if b"conv_fused_kernel" not in name:
        skipped += 1
        continue
    fam = ("bands " if N < cus else "whole ") + name.decode().replace("lasso::", "")
    w = torch.randn(K, C, kh, kw, generator=g) / (kh * kw) ** 0.5
    x = torch.randn(N, C, H, W_, generator=g)
    z0 = torch.randn(N, K, Hz, Wz, generator=g) * 0.05
    lr = 0.3 / max(w.pow(2).sum().item(), 1e-3)
    nref = min(N, 6)
    xg, zg, wg = x.cuda(), z0.cuda(), w.cuda()
    for fast in (True, False):
        ref = orc.conv_fista(x[:nref], z0[:nref], w, 0.1, padding=(ph, pw), fast=fast, maxiter=7, lr=lr, tol=0.0)
        os.environ.pop("LASSO_CONV_FUSED", None)
        got = ista_conv2d(xg, zg, wg, 0.1, padding=(ph, pw), fast=fast, maxiter=7, lr=lr, tol=0.0)
        os.environ["LASSO_CONV_FUSED"] = "0"
        two = ista_conv2d(xg, zg, wg, 0.1, padding=(ph, pw), fast=fast, maxiter=7, lr=lr, tol=0.0)
        os.environ.pop("LASSO_CONV_FUSED", None)
        err = (got[:nref].cpu() - ref).abs().max().item()
        if err > worst.get(fam, (-1, ""))[0]:
            worst[fam] = (err, (N, C, K, kh, kw, (ph, pw), Hz, Wz))
        if not torch.equal(got, two):
            notbit.append((N, C, K, kh, kw, (ph, pw), Hz, Wz, fast, (got - two).abs().max().item()))
    count[fam] = count.get(fam, 0) + 1
print("geometries", sum(count.values()), "not covered by the kernel", skipped, "seconds", round(time.time() - t0, 1))
for fam in sorted(worst):
    print("%-44s n=%-3d worst max|dz| %.3e at (N, C, K, kh, kw, padding, Hz, Wz) = %s" % (fam, count[fam], worst[fam][0], worst[fam][1]))
bad = [f for f in worst if worst[f][0] > 5e-5]
print("ALL WITHIN 5e-5 OF THE ORACLE" if not bad else "ABOVE 5e-5: %s" % bad)
print("ALL BITWISE THE TWO-KERNEL FORM" if not notbit else "NOT BITWISE (%d): %s" % (len(notbit), notbit[:8]))
