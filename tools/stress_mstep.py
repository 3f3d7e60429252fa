#!/usr/bin/env python3
"""Seeded sweep over ragged shapes of the round-2 M-step kernels (single-launch atom sweep, 128x128
Gram product, ridge solve) and the half-height tile / split-k dispatch, each against the CPU oracle.
Not part of the test suite (minutes of oracle time); prints the worst deviation per family."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
import torch
from lasso_amd.engine import HipEngine
from lasso_amd.linear import update_dict, update_dict_ridge, sparse_encode
from oracle import lasso_oracle as orc

eng = HipEngine()
g = torch.Generator().manual_seed(12345)
worst = {}
def note(fam, err, desc):
    if err > worst.get(fam, (0, ""))[0]: worst[fam] = (err, desc)

t0 = time.time()
for trial in range(40):
    k = int(torch.randint(8, 700, (1,), generator=g)); d = int(torch.randint(4, 257, (1,), generator=g))
    n = int(torch.randint(k // 2 + 2, 3 * k + 50, (1,), generator=g))
    Z = torch.randn(n, k, generator=g) * (torch.rand(n, k, generator=g) < 0.25)
    X = torch.randn(n, d, generator=g)
    D = torch.nn.functional.normalize(torch.randn(d, k, generator=g), dim=0)
    # Gram
    A, B = eng.gram(Z.cuda(), X.cuda(), torch.empty(k * k + k * d, device="cuda"))
    note("gram", max((A.cpu() - Z.T @ Z).abs().max().item(), (B.cpu() - Z.T @ X).abs().max().item()) /
         max(1.0, (Z.T @ Z).abs().max().item()), (n, d, k))
    # constrained M-step (skip problems with empty atoms: both sides then draw random replacements)
    if (Z.abs().sum(0) > 0).all():
        Dh, Zh = D.clone().cuda(), Z.clone().cuda()
        torch.manual_seed(3); update_dict(Dh, X.cuda(), Zh)
        Dr, Zr = D.clone(), Z.clone()
        torch.manual_seed(3); orc.update_dict(Dr, X, Zr)
        note("sweep", (Dh.cpu() - Dr).abs().max().item(), (n, d, k))
    # ridge
    V = update_dict_ridge(X.cuda(), Z.cuda(), lambd=1e-2)
    Vr = orc.update_dict_ridge(X.double(), Z.double(), lambd=1e-2).float()
    note("ridge", (V.cpu() - Vr).abs().max().item() / max(1.0, Vr.abs().max().item()), (n, d, k))
for trial in range(60):
    d = int(torch.randint(4, 257, (1,), generator=g)); k = int(torch.randint(8, 1025, (1,), generator=g))
    n = int(torch.randint(1, 5000, (1,), generator=g))
    W = torch.nn.functional.normalize(torch.randn(d, k, generator=g), dim=0)
    X = torch.randn(n, d, generator=g)
    lr = 0.9 / orc.lipschitz_constant(W, "exact")          # a stable step: divergent runs amplify rounding noise
    z = sparse_encode(X.cuda(), W.cuda(), alpha=0.2, lr=lr, maxiter=12, tol=0.0)
    zr = orc.sparse_encode(X, W, alpha=0.2, lr=lr, maxiter=12, tol=0.0)
    note("encode", (z.cpu() - zr).abs().max().item(), (n, d, k))
print("seconds", round(time.time() - t0, 1))
for fam, (err, desc) in worst.items():
    print("%-8s worst %.3e at (n, d, k) = %s" % (fam, err, desc))
