#!/usr/bin/env python3
"""The unfused FISTA path (d > 256 or k > 1024) with and without the stop rule: us per iteration of a solve whose rule
does not fire, against the same number of iterations with tol = 0.  usage: bench_stop_rule.py [--lib other.so]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd")]
import torch
from lasso_amd import _native as nat
if '--lib' in sys.argv:
    nat.use_library(os.path.abspath(sys.argv[sys.argv.index('--lib') + 1]))
from lasso_amd.linear.solvers.ista import ista
for (n, d, k) in ((4096, 784, 1024), (2048, 300, 2048), (512, 512, 1536)):
    g = torch.Generator().manual_seed(0)
    W = torch.nn.functional.normalize(torch.randn(d, k, generator=g), dim=0).cuda()
    X = torch.randn(n, d, generator=g).cuda()
    z0 = torch.zeros(n, k, device="cuda")
    lr = 1.0 / torch.linalg.eigvalsh((W.T @ W).double().cpu())[-1].item()
    rec = {"n": n, "d": d, "k": k}
    for mi in (10, 60):
        for name, tol in (("rule", 1e-12), ("no_rule", 0.0)):
            ts = []
            for rep in range(5):
                torch.cuda.synchronize(); t = time.perf_counter()
                ista(X, z0, W, 0.2, lr=lr, maxiter=mi, tol=tol)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
            rec["us_per_iteration_%s_maxiter%d" % (name, mi)] = round(min(ts) * 1e6 / mi, 1)
    print(json.dumps(rec), flush=True)
