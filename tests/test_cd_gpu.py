"""GPU parity tests for greedy coordinate descent: the HIP solver
(through the C ABI) against the golden fixtures generated from the reference's
coord_descent (tests/golden/cd_cases.npz) and against the CPU oracle.

Tolerances.  The solver commits ONE coordinate per row and step, chosen by an argmax, so
it is a discrete trajectory: as long as the HIP run and the CPU run pick the same
coordinates the codes agree to fp32 rounding (checked: max|dz| <= 5e-5, the FISTA bound,
measured <= 6e-6 up to 60 steps).  Near convergence the candidate moves shrink to the
size of the rounding differences in b = xW (MFMA vs MKL summation order), picks start to
differ and so do individual entries -- exactly as they do between the reference in fp32
and in fp64.  Long runs are therefore judged like bf16 in SURVEY 8d: on the per-row
objective (rtol 2e-4) and against the fp64 trajectory as the common yardstick.
"""
import numpy as np
import pytest
import torch

from recipes import recipe_xw

pytestmark = pytest.mark.gpu

Z_ATOL = 5e-5


def _mods():
    from lasso_amd.linear import sparse_encode
    from lasso_amd.linear.solvers import coord_descent
    from oracle import lasso_oracle as orc
    return sparse_encode, coord_descent, orc


def _row_objective(X, W, z, alpha):
    return 0.5 * (z @ W.T - X).pow(2).sum(1) + alpha * z.abs().sum(1)


def test_golden_short_runs(golden):
    _, coord_descent, _ = _mods()
    g = golden("cd_cases")
    for tag in "abcd":
        X, W, a = torch.from_numpy(g[tag + "_X"]), torch.from_numpy(g[tag + "_W"]), float(g[tag + "_alpha"])
        for mi in (1, 7, 60):
            got, info = coord_descent(X.cuda(), W.cuda(), None, a, maxiter=mi, return_info=True)
            ref = torch.from_numpy(g["%s_z_%d" % (tag, mi)])
            assert got.is_cuda and got.shape == ref.shape and got.dtype == ref.dtype
            assert (got.cpu() - ref).abs().max().item() <= Z_ATOL, (tag, mi)
            assert info["max_steps"] == mi


def test_golden_long_runs_objective(golden):
    _, coord_descent, _ = _mods()
    g = golden("cd_cases")
    for tag in "abcd":
        X, W, a = torch.from_numpy(g[tag + "_X"]), torch.from_numpy(g[tag + "_W"]), float(g[tag + "_alpha"])
        got = coord_descent(X.cuda(), W.cuda(), None, a, maxiter=1000).cpu()
        ref = torch.from_numpy(g[tag + "_z_1000"])
        o_got, o_ref = _row_objective(X, W, got, a), _row_objective(X, W, ref, a)
        assert ((o_got - o_ref).abs() / o_ref).max().item() <= 2e-4, tag
        # most rows still follow the very same trajectory
        assert ((got - ref).abs().max(1)[0] <= 1e-4).float().mean().item() >= 0.7, tag


def test_golden_warm_start_updates_z0_in_place(golden):
    """coordinate_descent.py:14,47 -- a caller-supplied z0 ends up holding the tracked z."""
    _, coord_descent, _ = _mods()
    g = golden("cd_cases")
    for tag in "abcd":
        X, W, a = torch.from_numpy(g[tag + "_X"]), torch.from_numpy(g[tag + "_W"]), float(g[tag + "_alpha"])
        z0 = torch.from_numpy(g[tag + "_z0"].copy()).cuda()
        got = coord_descent(X.cuda(), W.cuda(), z0, a, maxiter=40, tol=1e-4)
        assert (got.cpu() - torch.from_numpy(g[tag + "_z_warm"])).abs().max().item() <= Z_ATOL
        assert (z0.cpu() - torch.from_numpy(g[tag + "_z0_after"])).abs().max().item() <= Z_ATOL
        # a CPU z0 is updated in place as well (staged through the device)
        z0c = torch.from_numpy(g[tag + "_z0"].copy())
        coord_descent(X.cuda(), W.cuda(), z0c, a, maxiter=40, tol=1e-4)
        assert (z0c - torch.from_numpy(g[tag + "_z0_after"])).abs().max().item() <= Z_ATOL


def test_sparse_encode_cd_arm(golden):
    sparse_encode, _, _ = _mods()
    g = golden("cd_cases")
    for tag in "abcd":
        X, W, a = torch.from_numpy(g[tag + "_X"]), torch.from_numpy(g[tag + "_W"]), float(g[tag + "_alpha"])
        got = sparse_encode(X.cuda(), W.cuda(), alpha=a, algorithm="cd", maxiter=25)
        assert (got.cpu() - torch.from_numpy(g[tag + "_z_sparse_encode"])).abs().max().item() <= Z_ATOL


@pytest.mark.parametrize("n,d,k", [(1, 3, 2), (5, 7, 64), (33, 200, 513), (9, 300, 1500),
                                   (6, 40, 2049), (3, 20, 4096), (70, 100, 300)])
def test_shapes_match_oracle(n, d, k):
    _, coord_descent, orc = _mods()
    g = torch.Generator().manual_seed(n * 1000 + k)
    W = torch.nn.functional.normalize(torch.randn(d, k, generator=g), dim=0)
    X = torch.randn(n, d, generator=g)
    for mi in (1, 20):
        ref, rinfo = orc.coordinate_descent(X, W, None, 0.3, maxiter=mi, return_info=True)
        got, info = coord_descent(X.cuda(), W.cuda(), None, 0.3, maxiter=mi, return_info=True)
        assert (got.cpu() - ref).abs().max().item() <= Z_ATOL, (mi,)
        assert info["max_steps"] == int(rinfo["row_steps"].max())
        assert info["n_active"] == rinfo["n_active"]


def test_rows_stop_independently():
    """Per-row stop rule (:45-48): with a loose tol some rows finish early; the step count
    of every row and the active set must match the oracle."""
    _, coord_descent, orc = _mods()
    g = torch.Generator().manual_seed(5)
    W = torch.nn.functional.normalize(torch.randn(16, 40, generator=g), dim=0)
    X = torch.randn(64, 16, generator=g)
    ref, rinfo = orc.coordinate_descent(X, W, None, 0.4, maxiter=30, tol=2e-3, return_info=True)
    got, info = coord_descent(X.cuda(), W.cuda(), None, 0.4, maxiter=30, tol=2e-3, return_info=True)
    assert 0 < rinfo["n_active"] < 64 or int(rinfo["row_steps"].min()) < 30
    assert info["n_active"] == rinfo["n_active"]
    assert info["max_steps"] == int(rinfo["row_steps"].max())
    assert (got.cpu() - ref).abs().max().item() <= Z_ATOL


def test_fp64_yardstick():
    """Long run: the HIP fp32 result is as close to the fp64 trajectory's objective as the
    reference's own fp32 arithmetic is (the divergence of individual entries is the
    argmax chaos described in the module docstring, not an error of the kernel)."""
    _, coord_descent, orc = _mods()
    g = torch.Generator().manual_seed(11)
    W = torch.nn.functional.normalize(torch.randn(48, 200, generator=g), dim=0)
    X = torch.randn(100, 48, generator=g)
    z64 = orc.coordinate_descent(X.double(), W.double(), None, 0.3, maxiter=1000)
    z32 = orc.coordinate_descent(X, W, None, 0.3, maxiter=1000)
    zg = coord_descent(X.cuda(), W.cuda(), None, 0.3, maxiter=1000).cpu()
    o64 = _row_objective(X.double(), W.double(), z64, 0.3)
    e32 = ((_row_objective(X.double(), W.double(), z32.double(), 0.3) - o64).abs() / o64).max().item()
    eg = ((_row_objective(X.double(), W.double(), zg.double(), 0.3) - o64).abs() / o64).max().item()
    assert eg <= max(2.0 * e32, 2e-4), (eg, e32)


def test_c2_shape_statistics(golden):
    """BASELINE config-2 shape (first 512 rows of the recipe; rows are independent)."""
    _, coord_descent, _ = _mods()
    g = golden("cd_cases")
    X, W = recipe_xw(512, 256, 1024)
    for mi in (100, 1000):
        z = coord_descent(X.cuda(), W.cuda(), None, 0.5, maxiter=mi).cpu()
        obj = _row_objective(X, W, z, 0.5)
        ref = torch.from_numpy(g["c2_obj_rows_%d" % mi])
        assert ((obj - ref).abs() / ref).max().item() <= 2e-4
        assert abs(obj.mean().item() - ref.mean().item()) <= 1e-5 * ref.mean().item()
    z = coord_descent(X.cuda(), W.cuda(), None, 0.5, maxiter=100).cpu()
    assert (z[:64, :64] - torch.from_numpy(g["c2_corner_100"])).abs().max().item() <= Z_ATOL


def test_row_sharding_is_exact():
    """rows are independent, so any row shard reproduces the full batch bitwise."""
    _, coord_descent, _ = _mods()
    X, W = recipe_xw(300, 64, 256)
    Xg, Wg = X.cuda(), W.cuda()
    full = coord_descent(Xg, Wg, None, 0.2, maxiter=200)
    parts = torch.cat([coord_descent(Xg[i:i + 77].contiguous(), Wg, None, 0.2, maxiter=200)
                       for i in range(0, 300, 77)])
    assert torch.equal(full, parts)


def test_edge_cases_and_errors():
    sparse_encode, coord_descent, _ = _mods()
    X, W = recipe_xw(8, 16, 32)
    Xg, Wg = X.cuda(), W.cuda()
    # empty batch
    assert coord_descent(Xg[:0], Wg).shape == (0, 32)
    # maxiter=0: no step, z = S_alpha(xW) (:52)
    z = coord_descent(Xg, Wg, None, 0.3, maxiter=0)
    assert (z.cpu() - torch.nn.functional.softshrink(X @ W, 0.3)).abs().max().item() <= 1e-5
    with pytest.raises(AssertionError):
        coord_descent(Xg, Wg, torch.zeros(8, 31, device="cuda"))
    with pytest.raises(AssertionError):
        coord_descent(Xg[:, :15], Wg)
    with pytest.raises(NotImplementedError):
        coord_descent(Xg, torch.zeros(16, 5000, device="cuda"))
    with pytest.raises(TypeError):
        sparse_encode(Xg, Wg, algorithm="cd", lr=0.1)      # not a coord_descent kwarg


def test_verbose_prints_reference_format(capsys):
    _, coord_descent, orc = _mods()
    X, W = recipe_xw(8, 16, 32)
    z = coord_descent(X.cuda(), W.cuda(), None, 0.3, maxiter=3, verbose=True)
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 3 and out[0].startswith("iter 0 - loss: ")
    ref = orc.coordinate_descent(X, W, None, 0.3, maxiter=3)
    assert (z.cpu() - ref).abs().max().item() <= Z_ATOL
    loss = float(out[-1].split("loss: ")[1])
    assert abs(loss - _row_objective(X, W, ref, 0.3).sum().item()) <= 1e-3 * max(1.0, abs(loss))
