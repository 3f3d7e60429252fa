"""GPU parity of the backtracking line search (ista.py:17-54): HIP vs oracle on small
cases, vs the golden z of the reference, and BASELINE config 3 (fp32) at full size."""
import warnings

import numpy as np
import pytest
import torch

from recipes import recipe_xw, LAMBDA_MAX_C2

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_small_cases_against_reference_outputs(golden):
    from lasso_amd.linear import sparse_encode
    g = golden("small_cases")
    for tag in "abcd":
        X, W = T(g[tag + "_X"]), T(g[tag + "_W"])
        z = sparse_encode(X.cuda(), W.cuda(), alpha=0.3, fast=True, lr=1.0, maxiter=8, tol=0.0,
                          backtrack=True)
        assert (z.cpu() - T(g[tag + "_z_bt"])).abs().max().item() <= 5e-5, tag


@pytest.mark.parametrize("fast", [True, False])
def test_trace_matches_oracle(fast):
    from lasso_amd.linear.solvers import ista
    from oracle import lasso_oracle as orc
    X, W = recipe_xw(512)
    z0 = X.new_zeros(512, 1024)
    tr = orc.FistaTrace()
    ref = orc.fista(X, z0, W, 0.5, fast=fast, lr=1.0, maxiter=6, tol=0.0, backtrack=True, trace=tr)
    got, info = ista(X.cuda(), z0.cuda(), W.cuda(), 0.5, fast=fast, lr=1.0, maxiter=6, tol=0.0,
                     backtrack=True, return_info=True)
    assert info["iterations"] == 6
    assert (got.cpu() - ref).abs().max().item() <= 5e-5
    # with a stop tolerance: same number of outer iterations
    tr2 = orc.FistaTrace()
    orc.fista(X, z0, W, 0.5, fast=fast, lr=1.0, maxiter=200, tol=1e-3, backtrack=True, trace=tr2)
    _, info2 = ista(X.cuda(), z0.cuda(), W.cuda(), 0.5, fast=fast, lr=1.0, maxiter=200, tol=1e-3,
                    backtrack=True, return_info=True)
    assert info2["iterations"] == tr2.iterations
    # lr0 already admissible -> one trial per iteration, identical to the fixed-step solve
    fixed = ista(X.cuda(), z0.cuda(), W.cuda(), 0.5, fast=fast, lr=1 / LAMBDA_MAX_C2, maxiter=6, tol=0.0)
    bt = ista(X.cuda(), z0.cuda(), W.cuda(), 0.5, fast=fast, lr=1 / LAMBDA_MAX_C2, maxiter=6, tol=0.0,
              backtrack=True)
    assert (fixed - bt).abs().max().item() <= 2e-5


def test_c3_fp32_full_size(golden):
    """BASELINE config 3 (fp32 leg): n=16384, lr0=1.0, 10 iterations; SURVEY 8d G3."""
    from lasso_amd.linear import sparse_encode
    from oracle import lasso_oracle as orc
    g = golden("g3_c3_backtrack")
    X, W = recipe_xw(16384)
    from lasso_amd.linear.solvers import ista
    tr = golden("g3_c3_trace")
    z, info = ista(X.cuda(), torch.zeros(16384, 1024, device="cuda"), W.cuda(), 0.5, lr=1.0, maxiter=10,
                   tol=0.0, backtrack=True, return_info='objective')
    z = z.cpu()
    # the line-search trace of the reference: trials per outer iteration and the accepted step
    assert info["trials"] == tr["fp32_fista_trials"].tolist() == [5, 3, 5, 4, 4, 4, 4, 3, 5, 5]
    assert np.allclose(info["accepted_lr"], tr["fp32_fista_lr"], rtol=1e-6)
    obj = orc.lasso_objective(X, z, W, 0.5).item()
    assert abs(obj - 64.142166) <= 1e-5 * 64.142166
    assert abs(info["objective"] - 64.142166) <= 1e-5 * 64.142166   # objective_out of the C ABI
    assert (z[:64, :64] - T(g["fp32_z_block"])).abs().max().item() <= 1e-4
    st = g["fp32_stats"]
    assert abs(z.double().abs().sum().item() - st[1]) <= 1e-5 * st[1]
    z, info = ista(X.cuda(), torch.zeros(16384, 1024, device="cuda"), W.cuda(), 0.5, fast=False, lr=1.0,
                   maxiter=5, tol=0.0, backtrack=True, return_info=True)
    assert info["trials"] == tr["fp32_ista_trials"].tolist()
    assert np.allclose(info["accepted_lr"], tr["fp32_ista_lr"], rtol=1e-6)
    obj = orc.lasso_objective(X, z.cpu(), W, 0.5).item()
    assert abs(obj - float(g["fp32_ista_bt_obj"])) <= 1e-5 * obj


def test_c3_bf16_first_iterations_against_the_reference(golden):
    """The bf16 kernels against the REFERENCE's own bf16 run, element by element, where that is meaningful: the first
    1-3 fixed-step and 1-2 line-search iterations of config 3 (fixture g3b: the real reference on bf16 tensors).
    Measured (profiles/r05/parity_margins.json): already after ONE iteration a third of the entries differ by one
    bf16 step.  That is not a rounding that fell the other way -- the reference rounds EVERY intermediate tensor to bf16
    (lr * grad, then z - that, then the shrink: ista.py:90 on bf16 tensors), the kernel keeps the chain in fp32 and
    rounds the code once (tests/bf16_model.py states its arithmetic).  What can be asserted against the reference:
    (a) the line search takes the reference's steps, (b) no entry is further than a few bf16 steps from the reference's,
    the supports agree to a few 1e-3, the code statistics to 1e-3, and (c) the kernel's code is at least as close to
    the EXACT (fp32) iterate on the same bf16 inputs as the reference's bf16 code is."""
    from lasso_amd.linear.solvers import ista
    from oracle import lasso_oracle as orc
    from margins import record_margins
    g = golden("g3b_c3_bf16_steps")
    X, W = recipe_xw(16384)
    Xb, Wb = X.bfloat16(), W.bfloat16()
    Xg, Wg = Xb.cuda(), Wb.cuda()
    z0 = torch.zeros(16384, 1024, device="cuda", dtype=torch.bfloat16)
    got = {}

    def compare(tag, z, exact, lam):
        z = z.float().cpu()
        out = {}
        for name, sl in (("block", (slice(0, 64), slice(0, 64))), ("strided", (slice(None, None, 256), slice(None, None, 16)))):
            view, ref, ex = z[sl], T(g["%s_%s" % (tag, name)]), exact[sl]
            diff = (view - ref).abs()
            # a code entry is (pre-shrink value) - lam: its rounding grain is a bf16 step of |z| + lam, not of |z|
            steps = diff / ((torch.maximum(view.abs(), ref.abs()) + lam) * 2.0 ** -8)
            out[name] = {"fraction_differing": float((diff > 0).float().mean()), "max_abs": float(diff.max()),
                         "max_in_bf16_steps_of_the_preshrink_value": float(steps.max()),
                         "support_mismatch": float(((view != 0) != (ref != 0)).float().mean()),
                         "mean_abs_error_vs_fp32_iterate": float((view - ex).abs().mean()),
                         "reference_bf16_mean_abs_error_vs_fp32_iterate": float((ref - ex).abs().mean())}
        st = g[tag + "_stats"]
        out["rel_dabssum"] = abs(z.double().abs().sum().item() - st[1]) / st[1]
        out["rel_dnnz"] = abs(int((z != 0).sum()) - st[2]) / st[2]
        got[tag] = out

    z032 = torch.zeros(16384, 1024)
    for M in (1, 2, 3):
        exact = orc.fista(Xb.float(), z032, Wb.float(), 0.5, lr=1.0 / LAMBDA_MAX_C2, maxiter=M, tol=0.0)
        compare("fixed_M%d" % M, ista(Xg, z0, Wg, 0.5, lr=1.0 / LAMBDA_MAX_C2, maxiter=M, tol=0.0), exact, 0.5 / LAMBDA_MAX_C2)
    for M in (1, 2):
        z, info = ista(Xg, z0, Wg, 0.5, lr=1.0, maxiter=M, tol=0.0, backtrack=True, return_info=True)
        exact = orc.fista(Xb.float(), z032, Wb.float(), 0.5, lr=1.0, maxiter=M, tol=0.0, backtrack=True)
        compare("bt_M%d" % M, z, exact, 0.5 * info["accepted_lr"][-1])
        got["bt_M%d" % M]["accepted_lr"] = info["accepted_lr"]
        assert np.allclose(info["accepted_lr"], g["bt_M%d_lr" % M], rtol=1e-6), (info["accepted_lr"], g["bt_M%d_lr" % M])   # (a)
    record_margins("c3_bf16_first_iterations_vs_reference", got)
    for tag, o in got.items():
        for name in ("block", "strided"):
            v = o[name]
            assert v["max_in_bf16_steps_of_the_preshrink_value"] <= 8.0 and v["support_mismatch"] <= 5e-3, (tag, name, v)   # (b)
            assert v["mean_abs_error_vs_fp32_iterate"] <= 1.05 * v["reference_bf16_mean_abs_error_vs_fp32_iterate"], (tag, name, v)   # (c)
        assert o["rel_dabssum"] <= 2e-3 and o["rel_dnnz"] <= 6e-3, (tag, o)


@pytest.mark.parametrize("fast", [True, False])
def test_c3_fp32_one_launch_per_iteration_is_bitwise_the_multi_launch_form(fast):
    """Round 5: config 3 in fp32 runs as ONE launch per outer iteration (csrc/bt_iter.hip: accept + gradient + trials
    per 16-row tile) + one decision launch.  Per element it is the arithmetic of the multi-launch kernels of round 4
    (kernel='splitk' keeps them reachable): codes bit for bit, the same trials and accepted steps; F of the accepted
    trial to fp32 rounding (the tile sums are taken in another order)."""
    from lasso_amd.linear.solvers import ista
    X, W = recipe_xw(16384)
    Xg, Wg, z0 = X.cuda(), W.cuda(), torch.zeros(16384, 1024, device="cuda")
    a, ia = ista(Xg, z0, Wg, 0.5, fast=fast, lr=1.0, maxiter=10, tol=0.0, backtrack=True, return_info=True)
    b, ib = ista(Xg, z0, Wg, 0.5, fast=fast, lr=1.0, maxiter=10, tol=0.0, backtrack=True, return_info=True, kernel='splitk')
    assert torch.equal(a, b)
    assert ia["trials"] == ib["trials"] and ia["accepted_lr"] == ib["accepted_lr"]
    assert np.allclose(ia["accepted_f"], ib["accepted_f"], rtol=2e-6)
    # warm start, ragged batch, a dictionary between the templates (k = 1000 -> K = 1024), the stop rule
    X2, W2 = recipe_xw(5000, 200, 1000)
    z1 = ista(X2.cuda(), torch.zeros(5000, 1000, device="cuda"), W2.cuda(), 0.5, fast=fast, lr=0.5, maxiter=3, tol=0.0,
              backtrack=True)
    a, ia = ista(X2.cuda(), z1, W2.cuda(), 0.5, fast=fast, lr=0.5, maxiter=60, tol=1e-4, backtrack=True, return_info=True)
    b, ib = ista(X2.cuda(), z1, W2.cuda(), 0.5, fast=fast, lr=0.5, maxiter=60, tol=1e-4, backtrack=True, return_info=True,
                 kernel='splitk')
    assert ia["iterations"] == ib["iterations"] and ia["trials"] == ib["trials"] and torch.equal(a, b)


def test_eta_and_failure_semantics():
    from lasso_amd.linear import sparse_encode
    from oracle import lasso_oracle as orc
    X, W = recipe_xw(64)
    with pytest.raises(ValueError):                                   # ista.py:18-19
        sparse_encode(X.cuda(), W.cuda(), lr=1.0, backtrack=True, eta_backtrack=1.0)
    # eta barely above 1: 1000 trials are not enough to reach 1/L from lr0=1e3 -> warn + revert
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        ref = orc.sparse_encode(X, W, alpha=0.5, lr=1e3, maxiter=1, tol=0.0, backtrack=True,
                                eta_backtrack=1.000001)
    assert any("backtracking" in str(w.message) for w in rec)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        got = sparse_encode(X.cuda(), W.cuda(), alpha=0.5, lr=1e3, maxiter=1, tol=0.0, backtrack=True,
                            eta_backtrack=1.000001)
    assert any("backtracking" in str(w.message) for w in rec)
    assert torch.allclose(got.cpu(), ref, atol=1e-2, rtol=1e-4)


def test_c3_bf16_leg(golden):
    """BASELINE config 3, bf16 tensors: objective (evaluated in fp32) within rtol 2e-3 of the
    reference's bf16 run; lr='auto' raises TypeError like the reference."""
    from lasso_amd.linear import sparse_encode
    from oracle import lasso_oracle as orc
    g = golden("g3_c3_backtrack")
    X, W = recipe_xw(16384)
    Xb, Wb = X.bfloat16(), W.bfloat16()
    from lasso_amd.linear.solvers import ista
    tr = golden("g3_c3_trace")
    z, info = ista(Xb.cuda(), torch.zeros(16384, 1024, device="cuda", dtype=torch.bfloat16), Wb.cuda(), 0.5,
                   lr=1.0, maxiter=10, tol=0.0, backtrack=True, return_info='objective')
    assert z.dtype == torch.bfloat16 and z.is_cuda
    obj = orc.lasso_objective(Xb.float(), z.float().cpu(), Wb.float(), 0.5).item()
    from margins import record_margins
    gaps = {"line_search_rel": abs(obj - float(g["bf16_obj_fp32eval"])) / obj, "hip_trials": info["trials"],
            "reference_bf16_trials": tr["bf16_fista_trials"].tolist(), "objective": obj,
            "reference_objective": float(g["bf16_obj_fp32eval"]), "reference_fp32_objective": 64.142166}
    record_margins("c3_bf16_objective_gap", gaps)
    assert abs(obj - float(g["bf16_obj_fp32eval"])) <= 2e-3 * obj
    assert abs(info["objective"] - obj) <= 1e-5 * obj
    # bf16 rounding may legitimately move a borderline F <= Q decision ("pin objective,
    # report trace"): the trace is printed next to the reference's bf16 run and bounded loosely
    print("C3 bf16 trials  HIP %s  reference(bf16) %s" % (info["trials"], tr["bf16_fista_trials"].tolist()))
    print("C3 bf16 accepted lr  HIP %s  reference(bf16) %s" % (
        ["%.6f" % v for v in info["accepted_lr"]], ["%.6f" % v for v in tr["bf16_fista_lr"]]))
    assert len(info["trials"]) == 10 and all(1 <= t <= 8 for t in info["trials"])
    assert all(abs(v - 1.5 ** -(t - 1)) <= 1e-6 for v, t in zip(info["accepted_lr"], info["trials"]))
    z = sparse_encode(Xb.cuda(), Wb.cuda(), alpha=0.5, lr=1.0 / LAMBDA_MAX_C2, maxiter=10, tol=0.0)
    obj = orc.lasso_objective(Xb.float(), z.float().cpu(), Wb.float(), 0.5).item()
    gaps["fixed_step_rel"] = abs(obj - float(g["bf16_fixed_obj_fp32eval"])) / obj
    record_margins("c3_bf16_objective_gap", gaps)
    assert abs(obj - float(g["bf16_fixed_obj_fp32eval"])) <= 2e-3 * obj
    with pytest.raises(TypeError):
        sparse_encode(Xb.cuda(), Wb.cuda(), alpha=0.5)


@pytest.mark.parametrize("n,d,k", [(37, 10, 50), (64, 256, 1024), (100, 48, 200), (130, 128, 512), (1, 3, 2),
                                   (257, 200, 1000)])
@pytest.mark.parametrize("fast", [True, False])
def test_native_bf16_line_search_kernels(n, d, k, fast):
    """bf16 tensors + backtrack run the bf16-MFMA kernels (csrc/bt_bf16.hip: 64-row tiles,
    bf16 operands, fp32 accumulation and state).  Against the fp32 kernels fed the same
    (exactly up-converted) bf16 data the objective agrees to 1e-3 -- inside the 2e-3 the
    reference's own all-bf16 arithmetic is allowed -- and a warm start works."""
    from lasso_amd.linear.solvers import ista
    from oracle import lasso_oracle as orc
    g = torch.Generator().manual_seed(n + k)
    W = torch.nn.functional.normalize(torch.randn(d, k, generator=g), dim=0).bfloat16()
    X = torch.randn(n, d, generator=g).bfloat16()
    z0 = (torch.randn(n, k, generator=g) * 0.05).bfloat16()
    for start in (torch.zeros_like(z0), z0):
        zb = ista(X.cuda(), start.cuda(), W.cuda(), 0.3, fast=fast, lr=1.0, maxiter=6, tol=0.0, backtrack=True)
        assert zb.dtype == torch.bfloat16 and zb.shape == (n, k)
        zf = ista(X.float().cuda(), start.float().cuda(), W.float().cuda(), 0.3, fast=fast, lr=1.0, maxiter=6,
                  tol=0.0, backtrack=True)
        ob = orc.lasso_objective(X.float(), zb.float().cpu(), W.float(), 0.3).item()
        of = orc.lasso_objective(X.float(), zf.cpu(), W.float(), 0.3).item()
        assert abs(ob - of) <= 1e-3 * abs(of), (ob, of)
    # the stop rule is honoured on this path too
    zt, info = ista(X.cuda(), torch.zeros_like(z0).cuda(), W.cuda(), 0.3, fast=fast, lr=1.0, maxiter=300, tol=1e-3,
                    backtrack=True, return_info=True)
    assert 1 <= info["iterations"] <= 300


@pytest.mark.parametrize("n,d,k", [(37, 10, 50), (64, 256, 1024), (130, 128, 512), (257, 200, 1000)])
def test_native_bf16_fixed_step(n, d, k):
    """bf16 tensors without the line search: bf16-MFMA gradient kernel + fp32 prox/momentum
    (lasso_hip.hip solve_fixed_bf16).  Same bar as above against the fp32 fused kernel fed
    the up-converted data; stop rule and warm start included."""
    from lasso_amd.linear.solvers import ista
    from oracle import lasso_oracle as orc
    g = torch.Generator().manual_seed(n + d)
    W = torch.nn.functional.normalize(torch.randn(d, k, generator=g), dim=0).bfloat16()
    X = torch.randn(n, d, generator=g).bfloat16()
    z0 = (torch.randn(n, k, generator=g) * 0.05).bfloat16()
    lr = 1.0 / orc.lipschitz_constant(W.float(), "exact")
    for fast in (True, False):
        for start in (torch.zeros_like(z0), z0):
            zb = ista(X.cuda(), start.cuda(), W.cuda(), 0.3, fast=fast, lr=lr, maxiter=12, tol=0.0)
            assert zb.dtype == torch.bfloat16 and zb.shape == (n, k)
            zf = ista(X.float().cuda(), start.float().cuda(), W.float().cuda(), 0.3, fast=fast, lr=lr, maxiter=12,
                      tol=0.0)
            ob = orc.lasso_objective(X.float(), zb.float().cpu(), W.float(), 0.3).item()
            of = orc.lasso_objective(X.float(), zf.cpu(), W.float(), 0.3).item()
            assert abs(ob - of) <= 1e-3 * abs(of), (fast, ob, of)
    zt, info = ista(X.cuda(), torch.zeros_like(z0).cuda(), W.cuda(), 0.3, lr=lr, maxiter=500, tol=1e-3,
                    return_info=True)
    zr, rinfo = ista(X.float().cuda(), torch.zeros_like(z0).float().cuda(), W.float().cuda(), 0.3, lr=lr,
                     maxiter=500, tol=1e-3, return_info=True)
    # bf16 iterates move on a coarser grid than fp32 ones, so sum|z - z+| crosses the budget a little
    # later or earlier; the reference's own bf16 run does the same
    assert abs(info["iterations"] - rinfo["iterations"]) <= max(3, (2 * rinfo["iterations"]) // 5)
    ob = orc.lasso_objective(X.float(), zt.float().cpu(), W.float(), 0.3).item()
    of = orc.lasso_objective(X.float(), zr.cpu(), W.float(), 0.3).item()
    assert abs(ob - of) <= 2e-3 * abs(of)


@pytest.mark.parametrize("n,d,k", [(64, 256, 1024), (1000, 200, 1000), (3000, 256, 512), (130, 100, 256)])
@pytest.mark.parametrize("backtrack", [True, False])
def test_persistent_bf16_kernel_against_multi_launch_kernels(n, d, k, backtrack):
    """The single-launch bf16 solve (csrc/bt16_persist.hip: p in LDS, g in registers, z streamed as bf16,
    decisions taken in the kernel) against the multi-launch bf16 kernels (fp32 state in HBM) and the
    fp32 kernels: objective within 2e-3 (SURVEY 8d's bf16 bar), same line-search trace on well
    separated decisions, warm start, stop rule."""
    from lasso_amd.linear.solvers import ista
    from oracle import lasso_oracle as orc
    g = torch.Generator().manual_seed(n + k)
    W = torch.nn.functional.normalize(torch.randn(d, k, generator=g), dim=0).bfloat16()
    X = torch.randn(n, d, generator=g).bfloat16()
    z0 = (torch.randn(n, k, generator=g) * 0.05).bfloat16()
    lr = 1.0 if backtrack else 1.0 / orc.lipschitz_constant(W.float(), "exact")
    for start in (torch.zeros_like(z0), z0):
        zp, ip = ista(X.cuda(), start.cuda(), W.cuda(), 0.3, lr=lr, maxiter=8, tol=0.0, backtrack=backtrack,
                      return_info=True)                                   # persistent kernel (auto)
        zm, im = ista(X.cuda(), start.cuda(), W.cuda(), 0.3, lr=lr, maxiter=8, tol=0.0, backtrack=backtrack,
                      return_info=True, kernel='tile')                    # multi-launch kernels
        zf = ista(X.float().cuda(), start.float().cuda(), W.float().cuda(), 0.3, lr=lr, maxiter=8, tol=0.0,
                  backtrack=backtrack)
        assert zp.dtype == torch.bfloat16 and ip["iterations"] == im["iterations"] == 8
        objs = [orc.lasso_objective(X.float(), z.float().cpu(), W.float(), 0.3).item() for z in (zp, zm, zf)]
        assert abs(objs[0] - objs[2]) <= 2e-3 * objs[2], objs
        assert abs(objs[1] - objs[2]) <= 2e-3 * objs[2], objs
        if backtrack:
            assert len(ip["trials"]) == 8 and all(1 <= t <= 12 for t in ip["trials"])
            assert max(abs(a - b) for a, b in zip(ip["trials"], im["trials"])) <= 1, (ip["trials"], im["trials"])
    # in-place call (z_out aliases z0 is not reachable from ista(); the warm start above covers z0 != NULL)
    zt, info = ista(X.cuda(), torch.zeros_like(z0).cuda(), W.cuda(), 0.3, lr=lr, maxiter=300, tol=2e-3,
                    backtrack=backtrack, return_info=True)
    assert 1 <= info["iterations"] < 300 and info["last_delta"] <= n * k * 2e-3 * (1 + 1e-6)


def test_verbose_with_line_search_prints_the_reference_losses(capsys):
    """verbose=True with backtrack=True: the reference prints 'loss: %0.4f' of z before every outer
    iteration (ista.py:80-81; its backtracking() is called without verbose).  Here the lines come
    from the accepted trials' F values (accepted_f_out of the C ABI) -- no extra pass over the data."""
    from lasso_amd.linear.solvers import ista
    from oracle import lasso_oracle as orc
    X, W = recipe_xw(300)
    z0 = X.new_zeros(300, 1024)
    tr = orc.FistaTrace()
    orc.fista(X, z0, W, 0.5, lr=1.0, maxiter=7, tol=0.0, backtrack=True, trace=tr)
    capsys.readouterr()
    z, info = ista(X.cuda(), z0.cuda(), W.cuda(), 0.5, lr=1.0, maxiter=7, tol=0.0, backtrack=True, verbose=True,
                   return_info=True)
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("loss:")]
    assert len(lines) == 7 == len(tr.objective)
    for line, ref in zip(lines, tr.objective):
        assert abs(float(line.split()[1]) - ref) <= 1e-4 * ref + 1e-4, (line, ref)
    assert info["trials"] == tr.trials
    # the F values themselves: F(z_{i+1}) / n is the objective of the next iterate
    obj_next = [orc.lasso_objective(X, orc.fista(X, z0, W, 0.5, lr=1.0, maxiter=i + 1, tol=0.0, backtrack=True),
                                    W, 0.5).item() for i in range(3)]
    for f, ref in zip(info["accepted_f"], obj_next):
        assert abs(f / 300 - ref) <= 2e-6 * ref


@pytest.mark.parametrize("n,d,k,backtrack", [(700, 256, 1024, True), (700, 256, 1024, False), (1500, 100, 300, True),
                                             (300, 64, 256, False)])
def test_native_bf16_paths_against_the_oracle(n, d, k, backtrack):
    """The bf16 kernels (single-launch and multi-launch) against the ORACLE, not against other HIP
    kernels: the same bf16 tensors through the restated reference in fp32 arithmetic (exact
    up-conversion) and in the reference's own all-bf16 arithmetic.  Objective evaluated in fp32:
    within 1e-3 of the fp32-arithmetic oracle and -- SURVEY 8d's bar -- within 2e-3 of the bf16 one."""
    from lasso_amd.linear.solvers import ista
    from oracle import lasso_oracle as orc
    g = torch.Generator().manual_seed(n + k)
    W = torch.nn.functional.normalize(torch.randn(d, k, generator=g), dim=0).bfloat16()
    X = torch.randn(n, d, generator=g).bfloat16()
    z0 = torch.zeros(n, k).bfloat16()
    lr = 1.0 if backtrack else 0.1
    kw = dict(alpha=0.3, lr=lr, maxiter=8, tol=0.0, backtrack=backtrack)
    obj = lambda z: orc.lasso_objective(X.float(), z.float(), W.float(), 0.3).item()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref32 = obj(orc.fista(X.float(), z0.float(), W.float(), **kw))
        ref16 = obj(orc.fista(X, z0, W, **kw))
    for kern in ("auto", "tile"):
        z = ista(X.cuda(), z0.cuda(), W.cuda(), kernel=kern, **kw)
        assert z.dtype == torch.bfloat16
        o = obj(z.cpu())
        assert abs(o - ref32) <= 1e-3 * ref32, (kern, o, ref32)
        assert abs(o - ref16) <= 2e-3 * ref16, (kern, o, ref16)


@pytest.mark.parametrize("n,d,k,fast", [(60, 300, 40, True), (45, 64, 1500, True), (30, 512, 1100, False)])
def test_line_search_beyond_the_fused_shapes(n, d, k, fast):
    """d > 256 or k > 1024: the line search on the general GEMM + element-wise kernels
    (solve_generic_backtracking) -- trial trace, accepted steps and code against the oracle."""
    from lasso_amd.linear.solvers import ista
    from oracle import lasso_oracle as orc
    g = torch.Generator().manual_seed(n + d)
    W = torch.nn.functional.normalize(torch.randn(d, k, generator=g), dim=0)
    X = torch.randn(n, d, generator=g)
    z0 = torch.zeros(n, k)
    tr = orc.FistaTrace()
    ref = orc.fista(X, z0, W, 0.3, fast=fast, lr=2.0, maxiter=7, tol=0.0, backtrack=True, trace=tr)
    got, info = ista(X.cuda(), z0.cuda(), W.cuda(), 0.3, fast=fast, lr=2.0, maxiter=7, tol=0.0, backtrack=True,
                     return_info=True)
    assert info["trials"] == list(tr.trials) and max(tr.trials) > 1
    assert np.allclose(info["accepted_lr"], tr.accepted_lr, rtol=1e-6)
    assert (got.cpu() - ref).abs().max().item() <= 5e-5
    # with the stop rule
    tr2 = orc.FistaTrace()
    orc.fista(X, z0, W, 0.3, fast=fast, lr=2.0, maxiter=60, tol=2e-3, backtrack=True, trace=tr2)
    _, info2 = ista(X.cuda(), z0.cuda(), W.cuda(), 0.3, fast=fast, lr=2.0, maxiter=60, tol=2e-3, backtrack=True,
                    return_info=True)
    assert info2["iterations"] == tr2.iterations < 60


def test_persistent_bf16_line_search_survives_a_busy_gpu():
    """The single-launch bf16 line search (bt16_persist.hip) needs all of its 256 workgroups resident at config 3.
    With a second stream saturating the GPU that may not hold: a sweep of the granules then times out, the grid
    aborts as a whole and lasso_fista_solve runs the multi-launch kernels from the untouched inputs -- same
    trial trace, no error, no hang (VERDICT r1 item 5 for the kernel of item 4)."""
    from lasso_amd.linear.solvers import ista
    from recipes import recipe_xw
    X, W = recipe_xw(16384, 256, 1024)
    Xg, Wg = X.cuda().bfloat16(), W.cuda().bfloat16()
    z0 = torch.zeros(16384, 1024, device="cuda", dtype=torch.bfloat16)
    z_ref, info_ref = ista(Xg, z0, Wg, 0.5, lr=1.0, maxiter=10, tol=0.0, backtrack=True, return_info=True)
    assert info_ref["trials"] == [5, 3, 5, 4, 4, 4, 4, 3, 5, 5]
    side = torch.cuda.Stream()
    a = torch.randn(8192, 8192, device="cuda")
    b = torch.randn(8192, 8192, device="cuda")
    torch.cuda.synchronize()
    for trial in range(3):
        with torch.cuda.stream(side):
            for _ in range(6 + 4 * trial):            # ~10 ms each: the GPU stays busy for a while
                a = torch.mm(a, b) * 1e-2
        z, info = ista(Xg, z0, Wg, 0.5, lr=1.0, maxiter=10, tol=0.0, backtrack=True, return_info=True)
        assert info["trials"] == info_ref["trials"], (trial, info)
        assert (z.float() - z_ref.float()).abs().max().item() <= 2e-2, trial
        torch.cuda.synchronize()


@pytest.mark.parametrize("n,d,k", [(16384, 256, 1024), (37, 10, 50), (257, 200, 1000), (130, 128, 512), (1000, 256, 768)])
@pytest.mark.parametrize("fast", [True, False])
def test_persistent_bf16_kernel_against_its_arithmetic_model(n, d, k, fast):
    """The single-launch bf16 solve against tests/bf16_model.py -- the same rounding points (bf16 point, gradient,
    candidates and iterate; fp32 GEMM accumulation; the five sums of a trial in high precision) evaluated with torch
    CPU ops.  The line-search TRACE must be identical (trials per outer iteration and accepted steps: a wrong
    F <= Q decision changes it), the code agrees up to isolated bf16 roundings that fall the other way because
    the GEMMs add in a different order.  Config 3 itself is the first case: trace [5,3,5,4,4,4,4,3,5,5]."""
    from lasso_amd.linear.solvers import ista
    import bf16_model
    if (n, d, k) == (16384, 256, 1024):
        if not fast:
            n = 4096        # (the CPU model of the full batch takes ~40 s: the ISTA leg runs on a quarter of config 3)
        X, W = recipe_xw(n)
        alpha, z0 = 0.5, torch.zeros(n, k)
    else:
        g = torch.Generator().manual_seed(n + k)
        W = torch.nn.functional.normalize(torch.randn(d, k, generator=g), dim=0)
        X = torch.randn(n, d, generator=g)
        alpha, z0 = 0.3, torch.randn(n, k, generator=g) * 0.05
    Xb, Wb, z0b = X.bfloat16(), W.bfloat16(), z0.bfloat16()
    for backtrack, lr, iters in ((True, 1.0, 10), (False, 0.1, 12)):
        zm, minfo = bf16_model.solve(Xb, z0b, Wb, alpha, lr, iters, tol=0.0, fast=fast, backtrack=backtrack)
        zh, info = ista(Xb.cuda(), z0b.cuda(), Wb.cuda(), alpha, fast=fast, lr=lr, maxiter=iters, tol=0.0,
                        backtrack=backtrack, return_info=True)
        assert zh.dtype == torch.bfloat16
        if backtrack:
            assert info["trials"] == minfo["trials"], (info["trials"], minfo["trials"])
            assert all(abs(a - b) <= 1e-7 for a, b in zip(info["accepted_lr"], minfo["accepted_lr"]))
            if (n, d, k) == (16384, 256, 1024) and fast:
                assert info["trials"] == [5, 3, 5, 4, 4, 4, 4, 3, 5, 5]
        dz = (zh.float().cpu() - zm.float()).abs()
        scale = max(1.0, zm.float().abs().max().item())
        # isolated roundings that fall the other way are carried (and amplified by the momentum) through the remaining
        # iterations: the worst of 16.8 M elements at config 3 ends 3 bf16 steps off; the bulk is identical
        assert dz.max().item() <= 2.0 ** -5 * scale, (backtrack, dz.max().item(), scale)
        assert dz.mean().item() <= 1e-4 * scale, (backtrack, dz.mean().item())
        assert (dz > 0).float().mean().item() <= 6e-2        # measured: 0.5 - 4 % of the entries differ at all
        assert (dz > 2.0 ** -7 * scale).float().mean().item() <= 1e-4
    if n > 4096:
        return
    # the stop rule: same iteration count (sum |z - z+| is compared in fp32 on both sides)
    zm, minfo = bf16_model.solve(Xb, z0b, Wb, alpha, 1.0, 60, tol=2e-3, fast=fast, backtrack=True)
    zh, info = ista(Xb.cuda(), z0b.cuda(), Wb.cuda(), alpha, fast=fast, lr=1.0, maxiter=60, tol=2e-3, backtrack=True,
                    return_info=True)
    assert abs(info["iterations"] - minfo["iterations"]) <= 1, (info["iterations"], minfo["iterations"])
    # (deep into a converging run F and Q of a trial agree to fp32 rounding, and a decision may fall either way: the
    # first iterations, where they are apart, are compared)
    m = min(8, info["iterations"], minfo["iterations"])
    assert info["trials"][:m] == minfo["trials"][:m]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_windows_of_the_enqueued_line_search(dtype):
    """The line search is enqueued in windows of 16 outer iterations without host waits (csrc/lasso_hip.hip
    solve_backtracking).  (a) a stop rule that fires inside the third window: the oracle's iteration count and
    trace, nothing enqueued beyond that window; (b) searches that need MORE trials than a window pre-enqueues (lr0 = 30:
    ~14 trials each) take one synchronous iteration each and the windows resume -- the oracle's trace again."""
    from lasso_amd.linear.solvers import ista
    from oracle import lasso_oracle as orc
    g = torch.Generator().manual_seed(5)
    W = torch.nn.functional.normalize(torch.randn(48, 160, generator=g), dim=0).to(dtype)
    X = torch.randn(300, 48, generator=g).to(dtype)
    z0 = torch.zeros(300, 160, dtype=dtype)
    Xr, Wr, z0r = X.float(), W.float(), z0.float()          # the oracle in fp32 on the same (bf16-representable) data
    kw = dict(kernel="tile") if dtype == torch.bfloat16 else {}     # bf16: the multi-launch kernels (this driver)
    tr = orc.FistaTrace()
    orc.fista(Xr, z0r, Wr, 0.3, lr=1.0, maxiter=1000, tol=3e-4, backtrack=True, trace=tr)
    assert 32 < tr.iterations < 200
    _, info = ista(X.cuda(), z0.cuda(), W.cuda(), 0.3, lr=1.0, maxiter=1000, tol=3e-4, backtrack=True,
                   return_info=True, **kw)
    if dtype == torch.float32:
        assert info["iterations"] == tr.iterations and info["trials"] == list(tr.trials)
    else:           # bf16 gradients move the late iterations a little: the count may differ by a few
        assert abs(info["iterations"] - tr.iterations) <= max(3, tr.iterations // 10)
    tr = orc.FistaTrace()
    ref = orc.fista(Xr, z0r, Wr, 0.3, lr=30.0, maxiter=20, tol=0.0, backtrack=True, trace=tr)
    assert min(tr.trials) > 8
    got, info = ista(X.cuda(), z0.cuda(), W.cuda(), 0.3, lr=30.0, maxiter=20, tol=0.0, backtrack=True,
                     return_info=True, **kw)
    assert info["iterations"] == 20
    if dtype == torch.float32:
        assert info["trials"] == list(tr.trials)
        assert (got.cpu() - ref).abs().max().item() <= 5e-5
    else:
        assert sum(abs(a - b) for a, b in zip(info["trials"], tr.trials)) <= 2
    if dtype == torch.float32:
        # (c) round 5, the one-launch-per-iteration form (bt_iter.hip) computes 5 trials per tile: searches of 6-8 trials
        # (lr0 = 2.5) run out of them once, take ONE synchronous iteration, and from then on the windows carry the second
        # trial batch (trials 5-7 on the multi-launch kernel, same state) and size the first batch from the window
        # before; lr0 = 4 mixes 5 ... 9 trials -- more than both batches hold -- so synchronous iterations and windows
        # alternate.  The oracle's traces and codes.
        for lr0, lo, hi in ((2.5, 4, 8), (4.0, 5, 9)):
            tr = orc.FistaTrace()
            ref = orc.fista(Xr, z0r, Wr, 0.3, lr=lr0, maxiter=40, tol=0.0, backtrack=True, trace=tr)
            assert min(tr.trials) >= lo and max(tr.trials) <= hi and max(tr.trials) > 5
            got, info = ista(X.cuda(), z0.cuda(), W.cuda(), 0.3, lr=lr0, maxiter=40, tol=0.0, backtrack=True,
                             return_info=True)
            assert info["iterations"] == 40 and info["trials"] == list(tr.trials), (lr0, info["trials"], list(tr.trials))
            # 40 accelerated iterations at steps far above 1/L amplify last-ulp differences: the fp32 oracle is 1e-2 from
            # the fp64 oracle here (HIP vs fp32 oracle: 8e-5 / 7e-4 measured, 1e-6 after 10 iterations).  The sharp checks
            # are the traces above and the multi-launch form of round 4 (gradient / trials / accept as separate
            # launches, kernel='splitk'): the one-launch-per-iteration form returns its codes BIT FOR BIT
            assert (got.cpu() - ref).abs().max().item() <= 5e-3
            old = ista(X.cuda(), z0.cuda(), W.cuda(), 0.3, lr=lr0, maxiter=40, tol=0.0, backtrack=True, kernel='splitk')
            assert torch.equal(got, old)


def test_bf16_fixed_step_beyond_the_persistent_kernels_capacity():
    """Fixed step, no stop rule, bf16 tensors, more rows than the single-launch kernel holds resident (64 x #CUs): the
    batch runs as row blocks of that size on the SAME kernel (rows are independent) -- the codes are bitwise those of
    the rows solved block by block, in place too, and the objective is the fp32 kernels' to the bf16 bar."""
    from lasso_amd.linear.solvers import ista
    from oracle import lasso_oracle as orc
    n = 16384 + 4000
    X, W = recipe_xw(n)
    Xb, Wb = X.bfloat16().cuda(), W.bfloat16().cuda()
    z0 = torch.zeros(n, 1024, dtype=torch.bfloat16, device="cuda")
    lr = 1.0 / LAMBDA_MAX_C2
    z = ista(Xb, z0, Wb, 0.5, lr=lr, maxiter=8, tol=0.0)
    za = ista(Xb[:16384], z0[:16384], Wb, 0.5, lr=lr, maxiter=8, tol=0.0)
    zb = ista(Xb[16384:], z0[16384:], Wb, 0.5, lr=lr, maxiter=8, tol=0.0)
    assert torch.equal(z[:16384], za) and torch.equal(z[16384:], zb)
    zf = ista(Xb.float(), z0.float(), Wb.float(), 0.5, lr=lr, maxiter=8, tol=0.0)
    ob = orc.lasso_objective(Xb.float().cpu(), z.float().cpu(), Wb.float().cpu(), 0.5).item()
    of = orc.lasso_objective(Xb.float().cpu(), zf.cpu(), Wb.float().cpu(), 0.5).item()
    assert abs(ob - of) <= 1e-3 * of, (ob, of)
