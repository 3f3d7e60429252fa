#!/usr/bin/env python3
"""Small-batch FISTA throughput: tile kernel vs split-k kernel vs auto dispatch, n = 16 .. 4096
(d=256, k=1024 and two smaller dictionaries), 100 iterations, tol=0.  One JSON list."""
import json, os, sys, time
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

out = []
SHAPES = ((256, 1024),) if "--k1024" in sys.argv else ((256, 1024), (256, 512), (200, 256))
for d, k in SHAPES:
    for n in (16, 128, 256, 512, 768, 1024, 1536, 2048, 2560, 3072, 4096):
        X, W = recipe_xw(n, d, k)
        Xg, Wg = X.cuda(), W.cuda()
        z0 = torch.zeros(n, k, device="cuda")
        row = {"n": n, "d": d, "k": k}
        for kern in ("tile", "splitk1", "splitk1s", "splitk2", "splitk4", "splitk2g", "splitk4g", "auto"):
            ms = timed(lambda: ista(Xg, z0, Wg, 0.5, lr=0.05, maxiter=100, tol=0.0, kernel=kern))
            row[kern + "_us_per_iter"] = ms * 10.0
            row[kern + "_iters_per_s"] = 100 / ms * 1e3
        row["auto_tflops_useful"] = 4.0 * n * d * k * row["auto_iters_per_s"] / 1e12
        out.append(row)
        print(json.dumps(row), flush=True)
# time-to-tol at the 8-GPU shard of BASELINE config 2
X, W = recipe_xw(512)
Xg, Wg = X.cuda(), W.cuda()
z0 = torch.zeros(512, 1024, device="cuda")
for kern in ("tile", "splitk"):
    ista(Xg, z0, Wg, 0.5, lr=1 / 8.877719052098003, maxiter=2000, tol=1e-5, kernel=kern)
    torch.cuda.synchronize(); t = time.perf_counter()
    _, info = ista(Xg, z0, Wg, 0.5, lr=1 / 8.877719052098003, maxiter=2000, tol=1e-5, kernel=kern, return_info=True)
    torch.cuda.synchronize()
    print(json.dumps({"time_to_tol_n512": kern, "ms": (time.perf_counter() - t) * 1e3, "iterations": info["iterations"]}))
