#!/usr/bin/env python3
"""lambda_max of the product build against another library (--lib), bit for bit, on a few dictionaries; and its time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-lasso_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from lasso_amd import _native as nat
from lasso_amd.engine import HipEngine
def run(lib):
    if lib: nat.use_library(lib)
    eng = HipEngine()
    out = []
    for (d, k, seed) in [(64, 256, 7), (32, 100, 8), (48, 300, 9), (64, 64, 10), (256, 1024, 0), (256, 1024, 1), (128, 512, 2), (96, 768, 3), (200, 2048, 4), (256, 4096, 5), (192, 1000, 6)]:
        g = torch.Generator().manual_seed(seed)
        W = torch.nn.functional.normalize(torch.randn(d, k, generator=g), dim=0).cuda()
        L = eng.lipschitz(W)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): L = eng.lipschitz(W)
        torch.cuda.synchronize()
        ref = float(torch.linalg.eigvalsh((W.double() @ W.double().T))[-1])
        out.append((d, k, float(L).hex(), abs(float(L) - ref) / ref, (time.perf_counter() - t) / 20 * 1e6))
    return out
lib = sys.argv[sys.argv.index('--lib') + 1] if '--lib' in sys.argv else None
for r in run(lib): print("d=%d k=%d L=%s rel=%.2e  %.1f us" % r)
