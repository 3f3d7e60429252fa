#!/usr/bin/env python3
"""E-step timing at BASELINE config 5's shape (d=64, k=256) and a mid shape (d=128, k=512)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-lasso_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from lasso_amd.linear import sparse_encode
for (n, d, k) in [(65536, 64, 256), (65536, 128, 512), (16384, 64, 256)]:
    g = torch.Generator().manual_seed(0)
    W = torch.nn.functional.normalize(torch.randn(d, k, generator=g), dim=0).cuda()
    X = torch.randn(n, d, generator=g).cuda()
    lr = 1.0 / torch.linalg.eigvalsh((W.double() @ W.double().T))[-1].item()
    f = lambda: sparse_encode(X, W, 0.3, lr=lr, maxiter=100, tol=0.0)
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 5 * 1e3
    print(json.dumps({"n": n, "d": d, "k": k, "ms_per_100_iters": ms, "useful_TFLOPs": 4.0 * n * d * k * 100 / ms / 1e9}))
