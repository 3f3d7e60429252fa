"""Pin the plain-C fp64 oracle (oracle/lasso_oracle.c) against the golden outputs of the
reference and against the torch oracle (fp32 round-off apart)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def clib():
    so = os.path.join(ROOT, "oracle", "liblasso_oracle_c.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = C.CDLL(so)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    lib.oracle_fista.restype = C.c_int
    lib.oracle_fista.argtypes = [dp, dp, dp, dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                 C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, ip]
    lib.oracle_lasso_loss.restype = C.c_double
    lib.oracle_lasso_loss.argtypes = [dp, dp, dp, C.c_int, C.c_int, C.c_int, C.c_double]
    lib.oracle_update_dict.restype = C.c_int
    lib.oracle_update_dict.argtypes = [dp, dp, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, dp]
    return lib


def P(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def fista_c(lib, X, W, alpha, lr, fast=1, maxiter=25, tol=0.0, backtrack=0, eta=1.5):
    X = np.ascontiguousarray(X, np.float64)
    W = np.ascontiguousarray(W, np.float64)
    n, d = X.shape
    k = W.shape[1]
    z0 = np.zeros((n, k))
    z = np.empty((n, k))
    trials = (C.c_int * max(maxiter, 1))()
    its = lib.oracle_fista(P(X), P(W), P(z0), P(z), n, d, k, alpha, lr, fast, maxiter, tol, backtrack, eta, trials)
    return z, its, list(trials)[:its]


@pytest.mark.parametrize("tag", ["a", "c", "d"])
def test_c_oracle_against_reference_outputs(clib, golden, tag):
    g = golden("small_cases")
    X, W, lr = g[tag + "_X"], g[tag + "_W"], float(g[tag + "_lr"])
    z, its, _ = fista_c(clib, X, W, 0.3, lr)
    assert its == 25 and np.abs(z - g[tag + "_z_fista"]).max() < 2e-5
    z, _, _ = fista_c(clib, X, W, 0.3, lr, fast=0)
    assert np.abs(z - g[tag + "_z_ista"]).max() < 2e-5
    z, _, trials = fista_c(clib, X, W, 0.3, 1.0, maxiter=8, backtrack=1)
    assert np.abs(z - g[tag + "_z_bt"]).max() < 5e-5 and all(t >= 1 for t in trials)
    zf = np.ascontiguousarray(g[tag + "_z_fista"], np.float64)
    n, d = X.shape
    k = W.shape[1]
    loss = clib.oracle_lasso_loss(P(np.ascontiguousarray(X, np.float64)), P(zf),
                                  P(np.ascontiguousarray(W, np.float64)), n, d, k, 0.3)
    assert abs(loss - float(g[tag + "_loss"])) < 1e-5 * abs(loss)
    # atom sweep: replacement directions taken from the reference's own output
    D = np.ascontiguousarray(W, np.float64).copy()
    Z = zf.copy()
    fresh = np.ascontiguousarray(g[tag + "_D_bcd"].T, np.float64)      # [k][d]
    clib.oracle_update_dict(P(D), P(np.ascontiguousarray(X, np.float64)), P(Z), n, d, k, 0, 1e-10, P(fresh))
    assert np.abs(D - g[tag + "_D_bcd"]).max() < 5e-5
    assert np.array_equal(Z == 0, g[tag + "_Z_after_bcd"] == 0)


def test_c_oracle_stop_rule_matches_torch_oracle(clib):
    from oracle import lasso_oracle as orc
    g = torch.Generator().manual_seed(4)
    W = torch.nn.functional.normalize(torch.randn(12, 30, generator=g), dim=0)
    X = torch.randn(20, 12, generator=g)
    lr = 1.0 / orc.lipschitz_constant(W, "exact")
    tr = orc.FistaTrace()
    zt = orc.fista(X, X.new_zeros(20, 30), W, 0.2, lr=lr, maxiter=500, tol=1e-4, trace=tr)
    z, its, _ = fista_c(clib, X.numpy(), W.numpy(), 0.2, lr, maxiter=500, tol=1e-4)
    assert abs(its - tr.iterations) <= 1
    assert np.abs(z - zt.numpy()).max() < 1e-3
