#!/usr/bin/env python3
"""Fixed cost of a tile in the fused FISTA kernel: solve time against the iteration count (intercept = launch + tile
prologue / epilogue) and against the number of tile rounds at a fixed iteration count."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
import torch
from lasso_amd.linear.solvers import ista
from recipes import recipe_xw, LAMBDA_MAX_C2


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


out = []
for n in (4096, 8192, 16384):
    X, W = recipe_xw(n)
    Xg, Wg = X.cuda(), W.cuda()
    z0 = torch.zeros(n, 1024, device="cuda")
    for M in (5, 10, 20, 40):
        us = timed(lambda: ista(Xg, z0, Wg, 0.5, lr=1.0 / LAMBDA_MAX_C2, maxiter=M, tol=0.0))
        out.append({"n": n, "iterations": M, "us": us, "us_per_iteration_and_round": us / M / (n // 4096)})
        print(out[-1], flush=True)
