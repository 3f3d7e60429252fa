#!/usr/bin/env python3
"""Time the pieces of one EM step at BASELINE config 4's per-GPU shard (n=8192) and at the
full n=65536 on one GPU: E-step (10 FISTA its, lr given), objective, Gram, sweep, Lipschitz."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-lasso_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from recipes import recipe_xw, recipe_c4_init, LAMBDA_MAX_C4
if '--lib' in sys.argv:                       # an A/B build (tools/build_variant.sh)
    from lasso_amd import _native as _nat
    _nat.use_library(os.path.abspath(sys.argv[sys.argv.index('--lib') + 1]))
from lasso_amd.engine import HipEngine
from lasso_amd.linear import sparse_encode, dict_learning
from lasso_amd.parallel import constrained_mstep

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3

eng = HipEngine()
D0 = recipe_c4_init().cuda()
NS = (int(sys.argv[sys.argv.index('--n') + 1]),) if '--n' in sys.argv else (8192, 65536)
for n in NS:
    X, _ = recipe_xw(n)
    X = X.cuda()
    lr = 1.0 / LAMBDA_MAX_C4
    Z = sparse_encode(X, D0, 0.5, lr=lr, maxiter=10, tol=0.0)
    buf = torch.empty(1024 * 1024 + 1024 * 256, device='cuda')
    out = {"n": n}
    out["estep_10it_ms"] = timeit(lambda: sparse_encode(X, D0, 0.5, lr=lr, maxiter=10, tol=0.0))
    out["estep_default_ms(lr=auto,tol=1e-5)"] = timeit(lambda: sparse_encode(X, D0, 0.5))
    out["lipschitz_ms"] = timeit(lambda: eng.lipschitz(D0))
    out["objective_ms"] = timeit(lambda: eng.objective_sums(X, Z, D0, 0.5))
    out["gram_ms"] = timeit(lambda: eng.gram(Z, X, buf))
    A, B = eng.gram(Z, X, buf)
    D = D0.clone()
    out["sweep_ms"] = timeit(lambda: constrained_mstep(eng, A, B, D))
    out["ridge_ms"] = timeit(lambda: eng.ridge(A, B, 1e-2 * n))
    for steps in (10, 40):          # the first call also pays the one-off allocations (workspaces, pinned words)
        dict_learning(X, 1024, alpha=0.5, steps=2, algorithm='ista', progbar=False, device='cuda', init_weight=D0)
        torch.cuda.synchronize()
        t = time.perf_counter()
        dict_learning(X, 1024, alpha=0.5, steps=steps, algorithm='ista', progbar=False, device='cuda', init_weight=D0)
        torch.cuda.synchronize()
        out["em_step_ms(constrained, %d steps avg)" % steps] = (time.perf_counter() - t) / steps * 1e3
    print(json.dumps(out))
