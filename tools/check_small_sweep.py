#!/usr/bin/env python3
"""Digest of the dictionaries the atom sweep returns on a few small shapes (d <= 64, k <= 256) -- run once with the
product build and once with --lib variants/liblasso_nosmall.so (the single-launch sweep with workers): same digests."""
import os, sys, hashlib, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-lasso_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from lasso_amd import _native as nat
if '--lib' in sys.argv:
    nat.use_library(sys.argv[sys.argv.index('--lib') + 1])
from lasso_amd.engine import HipEngine
eng = HipEngine()
for (n, d, k, seed, dead, positive) in [(2048, 64, 256, 1, 0, False), (2048, 64, 256, 2, 3, False), (1024, 48, 200, 3, 2, True),
                                        (512, 64, 32, 4, 0, False), (512, 20, 64, 5, 1, False), (1024, 64, 160, 6, 0, False)]:
    g = torch.Generator().manual_seed(seed)
    Z = torch.randn(n, k, generator=g) * (torch.rand(n, k, generator=g) < 0.2)
    if dead:
        Z[:, torch.randperm(k, generator=g)[:dead]] = 0
    X = torch.randn(n, d, generator=g)
    D = torch.nn.functional.normalize(torch.randn(d, k, generator=g), dim=0).cuda()
    pool = torch.randn(max(dead, 1), d, generator=g).cuda()
    A, B = eng.gram(Z.cuda(), X.cuda(), torch.empty(k * k + k * d, device="cuda"))
    D1 = D.clone()
    mask, ndeg = eng.sweep(A, B, D1, pool, 1e-10, positive)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20):
        D2 = D.clone(); eng.sweep(A, B, D2, pool, 1e-10, positive)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t) / 20 * 1e6
    h = hashlib.sha1(D1.cpu().numpy().tobytes() + mask.cpu().numpy().tobytes()).hexdigest()[:16]
    print("d=%d k=%d dead=%d pos=%d ndeg=%d digest=%s norm_err=%.1e" % (d, k, dead, positive, ndeg, h, (D1.norm(dim=0) - 1).abs().max().item()), file=sys.stdout)
    print("   %.0f us per sweep call" % us, file=sys.stderr)
