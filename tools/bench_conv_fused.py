#!/usr/bin/env python3
"""Per-iteration time of the convolutional solver as the SLOPE between two solve lengths (20 and 60 iterations, tol = 0)
and the per-solve fixed cost (layout changes, W packing, launch) as the intercept, on the geometries of the
many-iterations-per-launch kernel (conv_fused.hip).  usage: bench_conv_fused.py [--lib other.so]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd")]
import torch
from lasso_amd import _native as nat
if "--lib" in sys.argv:
    nat.use_library(os.path.abspath(sys.argv[sys.argv.index("--lib") + 1]))
from lasso_amd.conv2d import ista_conv2d
CASES = [(256, 1, 64, 7, 0, 26), (256, 1, 64, 7, 3, 32), (256, 3, 32, 5, 2, 32), (512, 1, 32, 5, 0, 28), (1024, 1, 40, 7, 2, 16),
         (2048, 1, 64, 3, 1, 8), (2048, 1, 16, 3, 1, 8), (300, 2, 48, 3, 1, 20)]
out = []
if "--first" in sys.argv:
    CASES = CASES[:1]
for (N, C, K, ks, pd, Hz) in CASES:
    g = torch.Generator().manual_seed(0)
    w = torch.randn(K, C, ks, ks, generator=g) / ks
    H = (Hz - 1) - 2 * pd + ks
    x = torch.randn(N, C, H, H, generator=g)
    lr = 0.5 / w.pow(2).sum().item()
    xg, wg, zg = x.cuda(), w.cuda(), torch.zeros(N, K, Hz, Hz, device="cuda")
    t = {}
    for mi in (20, 60):
        best = 1e9
        for rep in range(3):
            ista_conv2d(xg, zg, wg, 0.1, stride=1, padding=pd, maxiter=mi, lr=lr, tol=0.0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                ista_conv2d(xg, zg, wg, 0.1, stride=1, padding=pd, maxiter=mi, lr=lr, tol=0.0)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 10)
        t[mi] = best
    per_it = (t[60] - t[20]) / 40
    flop = 4.0 * N * Hz * Hz * C * ks * ks * K
    out.append({"case": [N, C, K, ks, 1, pd, Hz], "us_per_iteration": round(per_it * 1e6, 2), "tflops": round(flop / per_it / 1e12, 1),
                "us_fixed_per_solve": round((t[20] - 20 * per_it) * 1e6, 1), "us_per_iteration_at_20": round(t[20] / 20 * 1e6, 2),
                "tflops_at_20": round(flop / (t[20] / 20) / 1e12, 1)})
    print(json.dumps(out[-1]))
