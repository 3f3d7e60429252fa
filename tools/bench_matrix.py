#!/usr/bin/env python3
"""Throughput of the fixed-step FISTA solve across shapes (fused tiles 16x256 / 32x128 / 64x64
and the unfused path), 100 iterations, tol=0.  One JSON list; evidence for DESIGN.md."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
import torch
from lasso_amd.linear import sparse_encode
from recipes import recipe_xw

def timed(fn, reps):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25:
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

out = []
shapes = [(4096, 256, 1024), (8192, 256, 1024), (65536, 256, 1024), (512, 256, 1024), (4096, 256, 512),
          (4096, 128, 512), (8192, 64, 256), (65536, 64, 256), (4096, 200, 1000), (4096, 100, 300),
          (4096, 512, 2048), (4096, 784, 1024), (16384, 512, 4096)]
for n, d, k in shapes:
    X, W = recipe_xw(n, d, k)
    Xg, Wg = X.cuda(), W.cuda()
    iters = 100 if n * d * k <= 4096 * 784 * 1024 * 4 else 20
    ms = timed(lambda: sparse_encode(Xg, Wg, alpha=0.5, lr=0.05, maxiter=iters, tol=0.0), 5)
    path = "fused" if d <= 256 and k <= 1024 else "unfused"
    out.append({"n": n, "d": d, "k": k, "path": path, "iterations": iters, "ms_per_solve": ms,
                "iterations_per_s": iters / ms * 1e3, "tflops_useful": 4.0 * n * d * k * iters / ms / 1e9})
print(json.dumps(out))
