#!/usr/bin/env python3
"""Time the atom sweep (lasso_dict_sweep) at k=1024, d=256.  usage: bench_sweep.py [--lib path]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-lasso_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from lasso_amd import _native as nat
if '--lib' in sys.argv:
    nat.use_library(sys.argv[sys.argv.index('--lib') + 1])
from recipes import recipe_xw, recipe_c4_init, LAMBDA_MAX_C4
from lasso_amd.engine import HipEngine
from lasso_amd.linear import sparse_encode
from lasso_amd.parallel import constrained_mstep

eng = HipEngine()
D0 = recipe_c4_init().cuda()
X, _ = recipe_xw(8192)
X = X.cuda()
Z = sparse_encode(X, D0, 0.5, lr=1.0 / LAMBDA_MAX_C4, maxiter=10, tol=0.0)
A, B = eng.gram(Z, X, torch.empty(1024 * 1024 + 1024 * 256, device='cuda'))
D = D0.clone()
constrained_mstep(eng, A, B, D); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20): constrained_mstep(eng, A, B, D)
torch.cuda.synchronize()
print(json.dumps({"lib": sys.argv[-1] if '--lib' in sys.argv else "default", "sweep_ms": (time.perf_counter() - t) / 20 * 1e3,
                  "checksum": float(D.double().sum())}))
