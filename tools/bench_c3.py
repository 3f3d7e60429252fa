#!/usr/bin/env python3
"""BASELINE config 3: FISTA with backtracking line search, n=16384 d=256 k=1024, bf16
tensors at the API (computed in fp32), lr0=1.0, 10 outer iterations.  One JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402

if '--lib' in sys.argv:                       # an A/B build (tools/build_variant.sh)
    from lasso_amd import _native as _nat
    _nat.use_library(os.path.abspath(sys.argv[sys.argv.index('--lib') + 1]))
from lasso_amd.linear.solvers import ista  # noqa: E402
from recipes import recipe_xw  # noqa: E402

TRIALS = [5, 3, 5, 4, 4, 4, 4, 3, 5, 5]      # fp32 trace of SURVEY 8d / fixture G3


def run(dtype, reps=10, kernel='auto'):
    X, W = recipe_xw(16384, 256, 1024)
    Xg, Wg = X.cuda().to(dtype), W.cuda().to(dtype)
    z0 = torch.zeros(16384, 1024, device="cuda", dtype=dtype)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        ista(Xg, z0, Wg, 0.5, lr=1.0, maxiter=10, tol=0.0, backtrack=True, kernel=kernel)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        z = ista(Xg, z0, Wg, 0.5, lr=1.0, maxiter=10, tol=0.0, backtrack=True, kernel=kernel)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / reps * 1e3
    _, info = ista(Xg, z0, Wg, 0.5, lr=1.0, maxiter=10, tol=0.0, backtrack=True, kernel=kernel, return_info=True)
    flop = sum(4 + 2 * t for t in info["trials"]) * 16384 * 256 * 1024
    zf = z.float()
    obj = ((0.5 * (zf @ Wg.float().T - Xg.float()).pow(2).sum() + 0.5 * zf.abs().sum()) / 16384).item()
    return {"dtype": str(dtype), "kernel": kernel, "ms_per_solve": ms, "tflops_algorithmic": flop / ms / 1e9,
            "objective": obj, "trials": info["trials"]}


if __name__ == "__main__":
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 10
    res = {"workload": "config 3: backtracking FISTA n=16384 d=256 k=1024, 10 outer iterations"}
    if "--bf16-only" not in sys.argv:
        res["fp32"] = run(torch.float32, reps)                         # one launch per outer iteration (bt_iter.hip)
        res["fp32_multilaunch"] = run(torch.float32, reps, kernel='splitk')   # round 4's gradient / trials / accept launches
    res["bf16_api"] = run(torch.bfloat16, reps)                       # persistent single-launch kernel
    res["bf16_multilaunch"] = run(torch.bfloat16, reps, kernel='tile')  # the grad / trial / decide / finish launches
    if "--fixed" in sys.argv:      # same data, fixed step 1/L, 10 iterations: fp32 fused kernel vs bf16 path
        from recipes import LAMBDA_MAX_C2
        X, W = recipe_xw(16384, 256, 1024)
        for name, dt in (("fixed_fp32", torch.float32), ("fixed_bf16", torch.bfloat16)):
            Xg, Wg = X.cuda().to(dt), W.cuda().to(dt)
            z0 = torch.zeros(16384, 1024, device="cuda", dtype=dt)
            f = lambda: ista(Xg, z0, Wg, 0.5, lr=1.0 / LAMBDA_MAX_C2, maxiter=10, tol=0.0)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.3:
                f()
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(reps):
                f()
            torch.cuda.synchronize()
            res[name + "_ms_per_10_iterations"] = (time.perf_counter() - t) / reps * 1e3
    print(json.dumps(res))
