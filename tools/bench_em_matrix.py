#!/usr/bin/env python3
"""The pieces of an EM step (10 E-step iterations, objective, Gram, sweep, Lipschitz) over d in 64..256, k in 256..1024 at
n = 65536: a quick way to spot a shape that falls off (useful TFLOP/s per piece)."""
import sys, time, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'pytorch-lasso_amd'), os.path.join(ROOT, 'tests')]
from lasso_amd.linear import sparse_encode
from lasso_amd.engine import HipEngine
from lasso_amd.parallel import constrained_mstep
eng = HipEngine()
torch.manual_seed(0)
def timeit(fn, reps=8):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
for n in (65536,):
  for d in (64, 128, 192, 256):
    for k in (256, 512, 768, 1024):
        X = torch.randn(n, d).cuda()
        D0 = torch.nn.functional.normalize(torch.randn(d, k), dim=0).cuda()
        Z = sparse_encode(X, D0, 0.5)
        buf = torch.empty(k * k + k * d, device='cuda')
        e = timeit(lambda: sparse_encode(X, D0, 0.5, lr=0.05, tol=0.0, maxiter=10))
        o = timeit(lambda: eng.objective_sums(X, Z, D0, 0.5))
        g = timeit(lambda: eng.gram(Z, X, buf))
        A, B = eng.gram(Z, X, buf); D = D0.clone()
        s = timeit(lambda: constrained_mstep(eng, A, B, D))
        l = timeit(lambda: eng.lipschitz(D0))
        print('d %3d k %4d  estep10 %.3f ms %5.1f TF | obj %.3f ms %5.1f TF | gram %.3f ms %5.1f TF | sweep %.3f | lip %.3f' % (
            d, k, e, 40.0 * n * d * k / e / 1e9, o, 2.0 * n * d * k / o / 1e9, g, 2.0 * n * k * (k / 2 + d) / g / 1e9, s, l), flush=True)
