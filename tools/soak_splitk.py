#!/usr/bin/env python3
"""Soak of the small-batch kernels: many back-to-back solves at the strong-scaling shard sizes; every 50th result is
compared bitwise with the first (the kernels are deterministic).  Run under rocprofv3 --kernel-trace to count stand-by
activations (tools/_soak.sh)."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'pytorch-lasso_amd'), os.path.join(ROOT, 'tests')]
from lasso_amd.linear.solvers import ista
from recipes import recipe_xw
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for n in (512, 1024, 2048):
    X, W = recipe_xw(n, 256, 1024)
    Xg, Wg = X.cuda(), W.cuda()
    z0 = torch.zeros(n, 1024, device='cuda')
    ref = ista(Xg, z0, Wg, 0.5, lr=0.1, maxiter=100, tol=0.0).clone()
    bad = 0
    for i in range(reps):
        z = ista(Xg, z0, Wg, 0.5, lr=0.1, maxiter=100, tol=0.0)
        if i % 50 == 0 and not torch.equal(z, ref):
            bad += 1
    torch.cuda.synchronize()
    print('n', n, 'solves', reps, 'mismatches', bad)
