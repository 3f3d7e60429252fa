#!/usr/bin/env python3
"""The convolutional solver with FEWER images than CUs: the two-kernel form (LASSO_CONV_FUSED=0), the dispatch (whole
images or bands of conv_fused.hip, chosen by its cost rule) and the dispatch without bands (LASSO_CONV_FUSED_BANDS=0):
us per iteration at 20 iterations and a hash of the codes (equal hashes: bitwise equal).  usage: ab_conv_small_batches.py"""
import os, sys, time, hashlib, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd")]
from lasso_amd.conv2d import ista_conv2d
def run(N, C, K, ks, pd, Hz):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(K, C, ks, ks, generator=g) / ks
    H = (Hz - 1) - 2 * pd + ks
    x = torch.randn(N, C, H, H, generator=g)
    lr = 0.5 / w.pow(2).sum().item()
    xg, wg, zg = x.cuda(), w.cuda(), torch.zeros(N, K, Hz, Hz, device="cuda")
    out = {}
    for tag, env in (("two-kernel", {"LASSO_CONV_FUSED": "0"}), ("dispatch", {}), ("no bands", {"LASSO_CONV_FUSED_BANDS": "0"})):
        for k in ("LASSO_CONV_FUSED", "LASSO_CONV_FUSED_BANDS"): os.environ.pop(k, None)
        os.environ.update(env)
        z = ista_conv2d(xg, zg, wg, 0.1, padding=pd, maxiter=20, lr=lr, tol=0.0)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): ista_conv2d(xg, zg, wg, 0.1, padding=pd, maxiter=20, lr=lr, tol=0.0)
        torch.cuda.synchronize()
        out[tag] = (round((time.perf_counter() - t) / 10 / 20 * 1e6, 1), hashlib.sha256(z.cpu().numpy().tobytes()).hexdigest()[:8])
    print((N, C, K, ks, pd, Hz), out)
for N in (255, 224, 192, 160, 144, 128, 96, 64):
    run(N, 1, 64, 7, 0, 26)
for N in (200, 128, 64):
    run(N, 3, 32, 5, 2, 32)
for N in (200, 128):
    run(N, 1, 16, 3, 1, 8)
