#!/usr/bin/env python3
"""EM step and its pieces at the shape of BASELINE config 5 (8 x 8 patches: d=64, k=256, constrained dict_learning,
defaults), n = 8192 (the 8-GPU shard) and 65536.  One JSON line per n."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
import torch
from lasso_amd.linear import dict_learning, sparse_encode
from lasso_amd.engine import HipEngine
from lasso_amd.parallel import constrained_mstep

eng = HipEngine()
torch.manual_seed(0)


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3


for n in (8192, 65536):
    d, k = 64, 256
    X = torch.randn(n, d).cuda()
    D0 = torch.nn.functional.normalize(torch.randn(d, k), dim=0).cuda()
    out = {"n": n, "d": d, "k": k}
    out["estep_default_ms(lr=auto,tol=1e-5)"] = timeit(lambda: sparse_encode(X, D0, 0.5))
    out["lipschitz_ms"] = timeit(lambda: eng.lipschitz(D0))
    Z = sparse_encode(X, D0, 0.5)
    buf = torch.empty(k * k + k * d, device='cuda')
    out["objective_ms"] = timeit(lambda: eng.objective_sums(X, Z, D0, 0.5))
    out["gram_ms"] = timeit(lambda: eng.gram(Z, X, buf))
    A, B = eng.gram(Z, X, buf)
    D = D0.clone()
    out["sweep_ms"] = timeit(lambda: constrained_mstep(eng, A, B, D))
    for steps in (10, 40):
        dict_learning(X, k, alpha=0.5, steps=2, algorithm='ista', progbar=False, device='cuda', init_weight=D0)
        torch.cuda.synchronize(); t = time.perf_counter()
        dict_learning(X, k, alpha=0.5, steps=steps, algorithm='ista', progbar=False, device='cuda', init_weight=D0)
        torch.cuda.synchronize()
        out["em_step_ms(constrained, %d steps avg)" % steps] = (time.perf_counter() - t) / steps * 1e3
    print(json.dumps(out), flush=True)
