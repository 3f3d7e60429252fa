#!/usr/bin/env python3
"""Ragged batches (more tiles than CUs, last round partly filled): auto dispatch (full rounds on the tile kernel,
the tail on the split-k kernel) against the tile kernel alone.  d=256, k=1024 / 512, 100 iterations, tol=0."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
import torch
from lasso_amd.linear.solvers import ista
from recipes import recipe_xw


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for k in (1024, 512):
    for n in (4096, 4112, 4608, 5120, 6144, 7168, 8192, 8704, 10240, 12288 + 512):
        X, W = recipe_xw(n, 256, k)
        Xg, Wg = X.cuda(), W.cuda()
        z0 = torch.zeros(n, k, device="cuda")
        row = {"n": n, "d": 256, "k": k}
        for kern in ("tile", "auto"):
            ms = timed(lambda: ista(Xg, z0, Wg, 0.5, lr=0.05, maxiter=100, tol=0.0, kernel=kern))
            row[kern + "_us_per_iter"] = ms * 10.0
        row["auto_tflops_useful"] = 4.0 * n * 256 * k / (row["auto_us_per_iter"] * 1e-6) / 1e12
        print(json.dumps(row), flush=True)
