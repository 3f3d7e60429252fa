#!/usr/bin/env python3
"""Soak of the co-operative M-step launches on a quiet GPU: N x {atom sweep (k=1024, d=256), Lipschitz constant of the
result}, each compared bit for bit with the first run; the give-up flags of both launches (flags[1] of the sweep's and
of the squarings' workspace) are read after every run -- a quiet GPU must never need the stand-by.
usage: soak_mstep.py [N]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-lasso_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from lasso_amd import _native as nat
from lasso_amd.engine import HipEngine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
eng = HipEngine()
k, d, n = 1024, 256, 4096
g = torch.Generator().manual_seed(5)
Z = (torch.randn(n, k, generator=g) * (torch.rand(n, k, generator=g) < 0.2)).cuda()
X = torch.randn(n, d, generator=g).cuda()
D = torch.nn.functional.normalize(torch.randn(d, k, generator=g), dim=0).cuda()
A, B = eng.gram(Z, X, torch.empty(k * k + k * d, device='cuda'))
al = lambda x: (x + 255) // 256 * 256
nblk = k // 32
rows = nblk * 32 * 256
off_flags = 2 * al(k * 256 * 4) + al(32 * 256 * 4) + 256 + 3 * rows * 4       # sweep workspace: ... | DtN | dDg | Uw | flags
ref_D, ref_L, sweep_aborts, lip_aborts = None, None, 0, 0
t0 = time.time()
for it in range(N):
    D1 = D.clone()
    eng.sweep(A, B, D1, None, 1e-10, False)
    L = eng.lipschitz(D1)
    ws = eng._ws(0, "sweep")
    sweep_aborts += int(ws.view(torch.uint8)[off_flags + 4: off_flags + 8].view(torch.int32).item() != 0)
    lws = nat.workspace(D1.device, 0, tag='lip')
    lip_aborts += int(lws.view(torch.uint8)[16 * 8 + 4: 16 * 8 + 8].view(torch.int32).item() != 0)
    if ref_D is None:
        ref_D, ref_L = D1, L
    else:
        assert torch.equal(D1, ref_D), it
        assert L == ref_L, it
print("%d runs in %.1f s: dictionary and lambda_max bitwise reproducible (lambda = %s); stand-by needed: sweep %d, squarings %d"
      % (N, time.time() - t0, float(ref_L).hex(), sweep_aborts, lip_aborts))
