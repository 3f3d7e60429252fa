#!/usr/bin/env python3
"""Fixed-step FISTA on rows with fewer features than the tile's padded width (d < 256 / 128): useful TFLOP/s per shape;
run once as is and once with LASSO_NO_DSTEPS=1 (the full-width kernels) for the A/B.  One JSON list."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
import torch
from lasso_amd.linear import sparse_encode
from recipes import recipe_xw


def timed(fn, reps):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = []
for n, d, k in [(4096, 64, 1024), (65536, 64, 1024), (8192, 64, 512), (65536, 64, 512), (4096, 96, 1024), (4096, 100, 300),
                (4096, 150, 1024), (4096, 192, 1024), (8192, 192, 512), (4096, 200, 1000), (4096, 224, 1024),
                (4096, 256, 1024)]:
    X, W = recipe_xw(n, d, k)
    Xg, Wg = X.cuda(), W.cuda()
    ms = timed(lambda: sparse_encode(Xg, Wg, alpha=0.5, lr=0.05, maxiter=100, tol=0.0), 5)
    out.append({"n": n, "d": d, "k": k, "ms_per_100_iterations": ms, "tflops_useful": 4.0 * n * d * k * 100 / ms / 1e9,
                "dsteps": "off" if os.environ.get("LASSO_NO_DSTEPS") else "on"})
print(json.dumps(out))
