#!/usr/bin/env python3
"""Unfused path (shapes beyond the fused tile kernel): FISTA iterations/s and TFLOP/s."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
import torch
if '--lib' in sys.argv:
    from lasso_amd import _native as _nat
    _nat.use_library(os.path.abspath(sys.argv[sys.argv.index('--lib') + 1]))
from lasso_amd.linear import sparse_encode
from recipes import recipe_xw
out = []
for n, d, k in ((4096, 256, 2048), (4096, 512, 2048), (4096, 784, 1024), (16384, 512, 4096)):
    X, W = recipe_xw(n, d, k)
    Xg, Wg = X.cuda(), W.cuda()
    iters = 50
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        sparse_encode(Xg, Wg, alpha=0.5, lr=0.05, maxiter=iters, tol=0.0)
    torch.cuda.synchronize()
    t = time.perf_counter()
    reps = 5
    for _ in range(reps):
        z = sparse_encode(Xg, Wg, alpha=0.5, lr=0.05, maxiter=iters, tol=0.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    out.append({"n": n, "d": d, "k": k, "ms_per_iteration": dt / iters * 1e3,
                "tflops": 4.0 * n * d * k * iters / dt / 1e12})
print(json.dumps(out))
