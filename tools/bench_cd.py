#!/usr/bin/env python3
"""Greedy coordinate descent (algorithm='cd') at the BASELINE
config-2 shape: HIP solver timing (events, after a clock warm-up) and the CPU oracle
timed beside it on a bounded row sample.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402

from lasso_amd.linear.solvers import coord_descent  # noqa: E402
from recipes import recipe_xw  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--d", type=int, default=256)
    ap.add_argument("--k", type=int, default=1024)
    ap.add_argument("--maxiter", type=int, default=1000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cpu-rows", type=int, default=256)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    X, W = recipe_xw(a.n, a.d, a.k)
    Xg, Wg = X.cuda(), W.cuda()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:                         # clocks up
        coord_descent(Xg, Wg, None, 0.5, maxiter=a.maxiter)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        z = coord_descent(Xg, Wg, None, 0.5, maxiter=a.maxiter)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    z, info = coord_descent(Xg, Wg, None, 0.5, maxiter=a.maxiter, return_info=True)
    row_steps = a.n * a.maxiter                                   # no row converges at tol=1e-6*k here
    out = {"workload": "coord_descent n=%d d=%d k=%d alpha=0.5 maxiter=%d" % (a.n, a.d, a.k, a.maxiter),
           "ms_per_solve": ms, "row_steps_per_s": row_steps / (ms * 1e-3),
           "l2_read_GBps": row_steps * 4.0 * max(256, a.k) / (ms * 1e-3) / 1e9,
           "max_steps": info["max_steps"], "n_active": info["n_active"]}
    if not a.no_cpu:
        from oracle import lasso_oracle as orc
        Xs = X[:a.cpu_rows]
        orc.coordinate_descent(Xs[:8], W, None, 0.5, maxiter=10)
        t = time.perf_counter()
        orc.coordinate_descent(Xs, W, None, 0.5, maxiter=a.maxiter)
        dt = time.perf_counter() - t
        out["cpu_baseline"] = {"row_steps_per_s": a.cpu_rows * a.maxiter / dt, "cores": torch.get_num_threads(),
                               "kind": "port", "sample": "%d rows x %d steps, %.1f s" % (a.cpu_rows, a.maxiter, dt)}
        out["speedup_vs_cpu"] = out["row_steps_per_s"] / out["cpu_baseline"]["row_steps_per_s"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
