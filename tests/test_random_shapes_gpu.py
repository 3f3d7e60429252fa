"""Randomised-shape sweep (seeded, deterministic) over every solver entry point against the
CPU oracle: ragged n/d/k around the tile boundaries of the fused kernels (16/32/64-row
tiles, 256/512/1024 atoms), the unfused path, coordinate descent, the convolutional solver
and the M-step.  Cheap cases, many of them -- the parity bar is the same as in the
dedicated test files (max|dz| <= 5e-5, dictionary 1e-4)."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


def _problem(n, d, k, seed):
    g = torch.Generator().manual_seed(seed)
    W = torch.nn.functional.normalize(torch.randn(d, k, generator=g), dim=0)
    X = torch.randn(n, d, generator=g)
    return X, W


def _shapes(count, seed, dmax, kmax, nmax=150):
    rnd = random.Random(seed)
    edges_d = [1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257]
    edges_k = [2, 3, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025]
    out = []
    for i in range(count):
        d = rnd.choice(edges_d) if i % 2 else rnd.randint(1, dmax)
        k = rnd.choice(edges_k) if i % 3 == 0 else rnd.randint(2, kmax)
        out.append((rnd.randint(1, nmax), min(d, dmax), min(k, kmax)))
    return out


def test_fista_random_shapes():
    from lasso_amd.linear import sparse_encode
    from oracle import lasso_oracle as orc
    for i, (n, d, k) in enumerate(_shapes(36, 101, 300, 1300)):
        X, W = _problem(n, d, k, i)
        lr = 1.0 / max(orc.lipschitz_constant(W, "exact"), 1e-3)
        fast = bool(i % 2)
        ref = orc.sparse_encode(X, W, alpha=0.2, fast=fast, lr=lr, maxiter=11, tol=0.0)
        got = sparse_encode(X.cuda(), W.cuda(), alpha=0.2, fast=fast, lr=lr, maxiter=11, tol=0.0)
        assert (got.cpu() - ref).abs().max().item() <= 5e-5, (n, d, k, fast)


def test_stop_rule_random_shapes():
    """iterations-to-tolerance equal to the oracle's (exact global rule, in-kernel or chunked)."""
    from lasso_amd.linear.solvers import ista
    from oracle import lasso_oracle as orc
    for i, (n, d, k) in enumerate(_shapes(7, 202, 256, 1024, nmax=4000)):      # (the oracle's 300 iterations on the CPU set the time)
        X, W = _problem(n, d, k, 50 + i)
        lr = 1.0 / max(orc.lipschitz_constant(W, "exact"), 1e-3)
        z0 = torch.zeros(n, k)
        tr = orc.FistaTrace()
        ref = orc.fista(X, z0, W, 0.4, lr=lr, maxiter=300, tol=1e-4, trace=tr)
        got, info = ista(X.cuda(), z0.cuda(), W.cuda(), 0.4, lr=lr, maxiter=300, tol=1e-4, return_info=True)
        assert (got.cpu() - ref).abs().max().item() <= 5e-5, (n, d, k)
        assert info["iterations"] == tr.iterations, (n, d, k, info, tr.iterations)


def test_stop_rule_on_a_ragged_batch_against_the_oracle():
    """n in (4096, 8192) at d=256, k=1024 is the range where run_impl splits a batch into full rounds on
    the tile kernel and a ragged tail on the split-k kernel (another order of the per-iteration sums of |z - z_next|):
    the iteration at which the exact rule fires, and the codes, against the CPU oracle -- not against another kernel."""
    from lasso_amd.linear.solvers import ista
    from lasso_amd import _native as nat
    from oracle import lasso_oracle as orc
    n, d, k = 4900, 256, 1024
    assert b"split" in nat.lib().lasso_fista_kernel_name(n, d, k, nat.LASSO_F32, 0)      # the hybrid dispatch is what runs
    X, W = _problem(n, d, k, 77)
    lr = 1.0 / max(orc.lipschitz_constant(W, "exact"), 1e-3)
    z0 = torch.zeros(n, k)
    tr = orc.FistaTrace()
    ref = orc.fista(X, z0, W, 0.4, lr=lr, maxiter=60, tol=1e-3, trace=tr)        # (~42 iterations)
    assert 5 < tr.iterations < 60                       # the rule does fire inside the budget
    got, info = ista(X.cuda(), z0.cuda(), W.cuda(), 0.4, lr=lr, maxiter=60, tol=1e-3, return_info=True)
    assert info["iterations"] == tr.iterations, (info, tr.iterations)
    assert (got.cpu() - ref).abs().max().item() <= 5e-5


def test_cd_random_shapes():
    from lasso_amd.linear.solvers import coord_descent
    from oracle import lasso_oracle as orc
    for i, (n, d, k) in enumerate(_shapes(20, 303, 300, 3000, nmax=40)):
        X, W = _problem(n, d, k, 100 + i)
        ref = orc.coordinate_descent(X, W, None, 0.3, maxiter=15)
        got = coord_descent(X.cuda(), W.cuda(), None, 0.3, maxiter=15)
        assert (got.cpu() - ref).abs().max().item() <= 5e-5, (n, d, k)


def test_update_dict_random_shapes():
    from lasso_amd.linear import update_dict, lasso_loss
    from oracle import lasso_oracle as orc
    for i, (n, d, k) in enumerate(_shapes(12, 404, 700, 400, nmax=400)):
        n = max(n, 20)
        X, W = _problem(n, d, k, 200 + i)
        lr = 1.0 / max(orc.lipschitz_constant(W, "exact"), 1e-3)
        Z = orc.sparse_encode(X, W, 0.3, lr=lr, maxiter=6, tol=0.0)
        assert abs(lasso_loss(X.cuda(), Z.cuda(), W.cuda(), 0.3).item() - orc.lasso_objective(X, Z, W, 0.3).item()) \
            <= 3e-6 * abs(orc.lasso_objective(X, Z, W, 0.3).item())
        Dref, Zref = W.clone(), Z.clone()
        torch.manual_seed(9)
        orc.update_dict(Dref, X, Zref)
        D, Zc = W.clone().cuda(), Z.clone().cuda()
        torch.manual_seed(9)
        update_dict(D, X.cuda(), Zc)
        used = Z.abs().sum(0) > 0
        if used.any():
            assert (D.cpu() - Dref)[:, used].abs().max().item() <= 1e-4, (n, d, k)
        assert torch.equal(Zc.cpu() == 0, Zref == 0), (n, d, k)


def test_conv_random_shapes():
    from lasso_amd.conv2d import ista_conv2d
    from oracle import lasso_oracle as orc
    rnd = random.Random(505)
    for i in range(14):
        N, C, K = rnd.randint(1, 5), rnd.randint(1, 6), rnd.randint(1, 70)
        kh, kw = rnd.choice([1, 2, 3, 5]), rnd.choice([1, 3, 4, 7])
        sh, sw = rnd.randint(1, 3), rnd.randint(1, 3)
        ph, pw = rnd.randint(0, kh - 1), rnd.randint(0, kw - 1)
        lo_h, lo_w = max(1, (2 * ph) // sh + 1), max(1, (2 * pw) // sw + 1)
        Hz, Wz = rnd.randint(lo_h, lo_h + 8), rnd.randint(lo_w, lo_w + 8)
        H, Wd = (Hz - 1) * sh - 2 * ph + kh, (Wz - 1) * sw - 2 * pw + kw
        if H <= 0 or Wd <= 0:
            continue
        g = torch.Generator().manual_seed(300 + i)
        w = torch.randn(K, C, kh, kw, generator=g) / (kh * kw) ** 0.5
        x = torch.randn(N, C, H, Wd, generator=g)
        z0 = torch.randn(N, K, Hz, Wz, generator=g) * 0.05
        lr = 0.3 / max(w.pow(2).sum().item(), 1e-3)
        ref = orc.conv_fista(x, z0, w, 0.1, stride=(sh, sw), padding=(ph, pw), maxiter=7, lr=lr, tol=0.0)
        got = ista_conv2d(x.cuda(), z0.cuda(), w.cuda(), 0.1, stride=(sh, sw), padding=(ph, pw), maxiter=7, lr=lr,
                          tol=0.0)
        assert (got.cpu() - ref).abs().max().item() <= 5e-5, (N, C, K, kh, kw, sh, sw, ph, pw, Hz, Wz)
