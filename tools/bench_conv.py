#!/usr/bin/env python3
"""Convolutional FISTA: HIP timing per iteration + CPU oracle beside it."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
import torch
from lasso_amd import _native as nat
if '--lib' in sys.argv:
    nat.use_library(sys.argv[sys.argv.index('--lib') + 1])
from lasso_amd.conv2d import ista_conv2d
out = []
cases = [(256, 1, 64, 7, 1, 0, 26), (64, 3, 128, 5, 1, 2, 64), (32, 16, 256, 3, 1, 1, 64)]
if '--last' in sys.argv:
    cases = cases[-1:]
if '--case' in sys.argv:                      # one case by index (kernel traces of a single geometry)
    i = int(sys.argv[sys.argv.index('--case') + 1])
    cases = cases[i:i + 1]
for (N, C, K, ks, st, pd, Hz) in cases:
    g = torch.Generator().manual_seed(0)
    w = torch.randn(K, C, ks, ks, generator=g) / ks
    H = (Hz - 1) * st - 2 * pd + ks
    x = torch.randn(N, C, H, H, generator=g)
    z0 = torch.zeros(N, K, Hz, Hz)
    lr = 0.5 / w.pow(2).sum().item()
    xg, wg, zg = x.cuda(), w.cuda(), z0.cuda()
    iters = 20
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        ista_conv2d(xg, zg, wg, 0.1, stride=st, padding=pd, maxiter=iters, lr=lr, tol=0.0)
    torch.cuda.synchronize()
    t = time.perf_counter(); reps = 5
    for _ in range(reps):
        ista_conv2d(xg, zg, wg, 0.1, stride=st, padding=pd, maxiter=iters, lr=lr, tol=0.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps / iters
    M, ckk = N * Hz * Hz, C * ks * ks
    rec = {"N": N, "C": C, "K": K, "ksize": ks, "stride": st, "padding": pd, "code_hw": Hz,
           "ms_per_iteration": dt * 1e3, "tflops": 4.0 * M * ckk * K / dt / 1e12,
           "patch_matrix_GBps": 4.0 * M * ckk * 4 / dt / 1e9}
    if "--no-cpu" not in sys.argv:
        from oracle import lasso_oracle as orc
        orc.conv_fista(x[:2], z0[:2], w, 0.1, stride=st, padding=pd, maxiter=2, lr=lr, tol=0.0)
        t = time.perf_counter()
        orc.conv_fista(x, z0, w, 0.1, stride=st, padding=pd, maxiter=5, lr=lr, tol=0.0)
        rec["cpu_ms_per_iteration"] = (time.perf_counter() - t) / 5 * 1e3
        rec["cpu_threads"] = torch.get_num_threads()
    out.append(rec)
print(json.dumps(out))
