#!/usr/bin/env python3
"""Time lasso_ridge_solve (k x k blocked Cholesky + substitutions) against torch.linalg on the same
Gram matrices.  usage: bench_ridge.py [k d n]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-lasso_amd')):
    sys.path.insert(0, p)
import torch
from lasso_amd.engine import HipEngine

k, d, n = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (1024, 256, 8192)
g = torch.Generator().manual_seed(0)
Z = (torch.randn(n, k, generator=g) * (torch.rand(n, k, generator=g) < 0.2)).cuda()
X = torch.randn(n, d, generator=g).cuda()
eng = HipEngine()
A, B = eng.gram(Z, X, torch.empty(k * k + k * d, device='cuda'))
lam = 1e-2 * n

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3

def torch_ridge():
    M = A.clone(); M.diagonal().add_(lam)
    return torch.cholesky_solve(B, torch.linalg.cholesky(M)).T

V = eng.ridge(A, B, lam, check=True)
M = A.double().clone(); M.diagonal().add_(lam)
ref = torch.cholesky_solve(B.double(), torch.linalg.cholesky(M)).T
print(json.dumps({"k": k, "d": d, "ridge_ms": timeit(lambda: eng.ridge(A, B, lam)),
                  "torch_linalg_ms": timeit(torch_ridge),
                  "max_err_vs_fp64": (V.double() - ref).abs().max().item(),
                  "torch_err_vs_fp64": (torch_ridge().double() - ref).abs().max().item()}))
