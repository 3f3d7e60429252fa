"""GPU parity tests for convolutional ISTA/FISTA and the Toeplitz Lipschitz bound
 against the golden fixtures generated from the reference's
``ista_conv2d`` / ``lip_bound_conv2d`` (tests/golden/conv_cases.npz) and the CPU oracle.
fp32 tolerance as for the linear solver: max|dz| <= 5e-5, objective rtol <= 1e-6 with an
explicit step; lr='auto' goes through the bound (rtol 1e-5: sin/cos differ in the last
ulp between libm and the device) so those runs are compared at 2e-4."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

Z_ATOL = 5e-5
TAGS = "abcde"


def _mods():
    from lasso_amd.conv2d import ista_conv2d, lip_bound_conv2d, LipBoundConv2d
    from lasso_amd.conv2d.ista import conv_loss
    from oracle import lasso_oracle as orc
    return ista_conv2d, lip_bound_conv2d, LipBoundConv2d, conv_loss, orc


def _case(g, tag):
    N, C, K, ks, st, pd, Hz, Wz = [int(v) for v in g[tag + "_cfg"]]
    x, w = torch.from_numpy(g[tag + "_x"]), torch.from_numpy(g[tag + "_w"])
    return x, w, torch.zeros(N, K, Hz, Wz), float(g[tag + "_alpha"]), float(g[tag + "_lr"]), st, pd


def test_golden_fixed_step(golden):
    ista_conv2d, _, _, conv_loss, orc = _mods()
    g = golden("conv_cases")
    for tag in TAGS:
        x, w, z0, alpha, lr, st, pd = _case(g, tag)
        for fast in (True, False):
            for mi in (1, 12):
                ref = torch.from_numpy(g["%s_z_%s_%d" % (tag, "fista" if fast else "ista", mi)])
                got = ista_conv2d(x.cuda(), z0.cuda(), w.cuda(), alpha, stride=st, padding=pd, fast=fast,
                                  maxiter=mi, lr=lr, tol=0.0)
                assert got.is_cuda and got.shape == ref.shape and got.dtype == ref.dtype
                assert (got.cpu() - ref).abs().max().item() <= Z_ATOL, (tag, fast, mi)
                o_ref = orc.conv_objective(x, ref, w, alpha, st, pd).item()
                o_got = conv_loss(x.cuda(), got, w.cuda(), alpha, st, pd).item()
                assert abs(o_got - o_ref) <= 2e-6 * abs(o_ref), (tag, fast, mi)


def test_golden_warm_start_and_inputs_untouched(golden):
    ista_conv2d, _, _, _, _ = _mods()
    g = golden("conv_cases")
    for tag in TAGS:
        x, w, _, alpha, lr, st, pd = _case(g, tag)
        zw = torch.from_numpy(g[tag + "_z0_warm"]).cuda()
        keep = zw.clone()
        got = ista_conv2d(x.cuda(), zw, w.cuda(), alpha, stride=st, padding=pd, maxiter=6, lr=lr, tol=0.0)
        assert (got.cpu() - torch.from_numpy(g[tag + "_z_warm"])).abs().max().item() <= Z_ATOL, tag
        assert torch.equal(zw, keep) and got.data_ptr() != zw.data_ptr()


def test_lipschitz_bound(golden):
    _, lip_bound_conv2d, LipBoundConv2d, _, orc = _mods()
    g = golden("conv_cases")
    for tag in TAGS:
        x, w, _, alpha, lr, st, pd = _case(g, tag)
        if st != 1:
            with pytest.raises(NotImplementedError):
                lip_bound_conv2d(w.cuda(), pd, stride=st)
            continue
        ref = float(g[tag + "_lip"])
        got = lip_bound_conv2d(w.cuda(), pd)
        assert got.dim() == 0 and got.is_cuda and got.dtype == torch.float32
        assert abs(got.item() - ref) <= 1e-5 * ref, tag
        assert abs(lip_bound_conv2d(w.cuda(), pd, sqrt=True).item() - float(g[tag + "_lip_sqrt"])) <= 1e-5 * ref
        assert abs(LipBoundConv2d(tuple(w.shape), pd)(w.cuda()).item() - ref) <= 1e-5 * ref
        # more output than input channels: the reference transposes the kernel first (:106-107)
        wt = w.transpose(0, 1).contiguous()
        assert abs(lip_bound_conv2d(wt.cuda(), pd).item() - orc.conv_lipschitz_bound(wt, pd).item()) <= 1e-5 * ref
    with pytest.raises(ValueError):
        lip_bound_conv2d(torch.zeros(4, 1, 4, 4, device="cuda"), 0)
    with pytest.raises(ValueError):
        lip_bound_conv2d(torch.zeros(4, 1, 3, 5, device="cuda"), 0)


def test_auto_step_and_stop_rule(golden):
    ista_conv2d, _, _, _, orc = _mods()
    g = golden("conv_cases")
    for tag in TAGS:
        x, w, z0, alpha, lr, st, pd = _case(g, tag)
        if st != 1:
            with pytest.raises(NotImplementedError):
                ista_conv2d(x.cuda(), z0.cuda(), w.cuda(), alpha, stride=st, padding=pd)
            continue
        ref = torch.from_numpy(g[tag + "_z_auto_tol"])
        _, rinfo = orc.conv_fista(x, z0, w, alpha, stride=st, padding=pd, maxiter=200, tol=1e-4, return_info=True)
        got, info = ista_conv2d(x.cuda(), z0.cuda(), w.cuda(), alpha, stride=st, padding=pd, maxiter=200, tol=1e-4,
                                return_info=True)
        # The step comes from the bound, which the device evaluates with its own sin/cos
        # (1e-6-level difference).  How much that matters after up to 200 momentum steps
        # depends on the problem (the bound is a LOWER bound of L, so 1/bound sits at the edge
        # of stability): measure the oracle's own sensitivity to a 2e-6 change of the step.
        assert abs(info["iterations"] - rinfo["iterations"]) <= 1, (tag, info, rinfo)
        err = (got.cpu() - ref).abs().max().item()
        if err <= 2e-4:
            continue            # (the two extra 200-iteration oracle runs below only where they are needed: 60 s of CPU)
        L = orc.conv_lipschitz_bound(w, pd).item()
        sens = max((orc.conv_fista(x, z0, w, alpha, stride=st, padding=pd, maxiter=200, tol=1e-4,
                                   lr=float(np.float32(1.0 / (L * (1 + rel))))) - ref).abs().max().item()
                   for rel in (2e-6, -2e-6))
        assert err <= 2e-4 + 3 * sens, (tag, info, rinfo, sens)


def test_stop_rule_chunks_replay_the_stopping_iteration():
    """The rule of ista.py:44-46 is evaluated once per chunk of speculated iterations (lasso_conv_ista_solve); when it
    fires inside a chunk the solve is put back to the chunk's head and replayed: the codes must be bitwise those of
    exactly `iterations` iterations without a rule, the count the oracle's, for rules that fire at the first
    iteration, inside and at the end of later chunks, and never."""
    ista_conv2d, _, _, _, orc = _mods()
    g = torch.Generator().manual_seed(11)
    for (N, C, K, ks, pd, Hz) in ((3, 1, 12, 5, 2, 14), (2, 3, 72, 3, 1, 12), (2, 16, 20, 3, 0, 9)):
        w = torch.randn(K, C, ks, ks, generator=g) / ks
        H = (Hz - 1) - 2 * pd + ks
        x = torch.randn(N, C, H, H, generator=g)
        z0 = torch.zeros(N, K, Hz, Hz)
        lr = 0.5 / w.pow(2).sum().item()
        xg, wg, zg = x.cuda(), w.cuda(), z0.cuda()
        # the sum of every iteration of a long run (one solve per length: the sums are not returned as a vector)
        deltas = []
        for m in range(1, 91):
            _, info = ista_conv2d(xg, zg, wg, 0.1, padding=pd, maxiter=m, lr=lr, tol=1e-30, return_info=True)
            assert info["iterations"] == m
            deltas.append(info["last_delta"])
        # a rule can fire first at the iterations whose sum is below every earlier one: the first of them, some in
        # the middle (inside and at the end of speculated chunks) and the last
        lows = [m for m in range(1, 91) if all(deltas[m - 1] < d for d in deltas[:m - 1])]
        assert len(lows) >= 4, lows
        picks = sorted(set([lows[0], lows[1], lows[len(lows) // 3], lows[len(lows) // 2], lows[-2], lows[-1]]))
        for m in picks:
            tol = float(np.float32(deltas[m - 1]) * np.float32(1.0 + 1e-6)) / z0.numel()
            got, info = ista_conv2d(xg, zg, wg, 0.1, padding=pd, maxiter=200, lr=lr, tol=tol, return_info=True)
            assert info["iterations"] == m and info["last_delta"] == deltas[m - 1], (m, info, lows)
            plain = ista_conv2d(xg, zg, wg, 0.1, padding=pd, maxiter=m, lr=lr, tol=0.0)
            assert torch.equal(got, plain), (N, C, K, m, info)
        _, rinfo = orc.conv_fista(x, z0, w, 0.1, padding=pd, maxiter=200, lr=lr, tol=tol, return_info=True)
        assert abs(rinfo["iterations"] - picks[-1]) <= 1, (picks, rinfo)   # (a sum within 1e-6 of the budget may fall either side)
        # a rule that never fires: all maxiter iterations, the codes of the plain run
        got, info = ista_conv2d(xg, zg, wg, 0.1, padding=pd, maxiter=37, lr=lr, tol=1e-30, return_info=True)
        assert info["iterations"] == 37 and torch.equal(got, ista_conv2d(xg, zg, wg, 0.1, padding=pd, maxiter=37, lr=lr, tol=0.0))


@pytest.mark.parametrize("N,C,K,kh,kw,stride,padding,Hz,Wz", [
    (1, 1, 1, 1, 1, 1, 0, 1, 1), (2, 1, 3, 3, 5, (1, 2), (1, 0), 6, 5), (3, 2, 70, 3, 3, 1, 1, 10, 12),
    (2, 5, 4, 5, 5, 3, 2, 4, 6), (64, 1, 32, 7, 7, 1, 0, 22, 22), (0, 1, 4, 3, 3, 1, 0, 5, 5),
    # 8 <= C <= 16, stride 1, square 3/5/7 kernels: the implicit-GEMM synthesis kernel (conv_synth.hip), every
    # instantiation, image sizes that are not multiples of the 4 x 16 pixel tile, paddings 0 .. ks-1, K % 32 != 0
    (2, 16, 256, 3, 3, 1, 1, 21, 35), (3, 8, 20, 3, 3, 1, 0, 9, 30), (2, 12, 64, 3, 3, 1, 2, 17, 18),
    (2, 9, 100, 3, 3, 1, 1, 16, 16), (1, 16, 200, 3, 3, 1, 0, 7, 40), (2, 8, 32, 5, 5, 1, 2, 13, 19),
    (2, 16, 64, 5, 5, 1, 0, 9, 21), (1, 10, 128, 5, 5, 1, 4, 12, 17), (2, 8, 24, 7, 7, 1, 3, 14, 20),
    (1, 16, 64, 7, 7, 1, 0, 8, 25),
    # C < 8, stride 1, K <= 128 a multiple of 4, <= 128 taps: the few-channel synthesis kernel (conv_synth_few.hip) --
    # W in LDS (5 x 8, 8 x 8 column blocks x atom groups) and in registers, many bands per image with their halos,
    # K % 16 != 0, a non-square kernel, asymmetric padding, more work items than workgroups
    (3, 3, 128, 5, 5, 1, 2, 20, 24), (2, 1, 64, 7, 7, 1, 3, 13, 17), (2, 4, 20, 3, 3, 1, 1, 9, 11),
    (1, 7, 100, 3, 5, 1, (0, 2), 8, 9), (600, 1, 16, 3, 3, 1, 1, 8, 8), (2, 3, 36, 5, 5, 1, 0, 6, 7),
    # same channel counts where it does not apply (stride 2, non-square kernel, C > 16): explicit path
    (2, 16, 32, 3, 3, 2, 1, 8, 9), (2, 8, 16, 3, 5, 1, 1, 9, 9), (1, 17, 8, 3, 3, 1, 1, 6, 6),
    # the gradient kernel's 128-pixel x 64-atom tile (K <= 64) where no MFMA step is padding (63 and 96 taps: the
    # instantiations without the skip), on a code grid wider than 48 (2 x 64 pixel tiles) and with K <= 64 next to
    # a second block of atoms (K = 72: 128-atom tiles)
    (2, 7, 40, 3, 3, 1, 1, 9, 10), (2, 6, 24, 4, 4, 1, 1, 7, 9), (1, 1, 8, 3, 3, 1, 1, 5, 50), (1, 2, 72, 3, 3, 1, 1, 5, 50)])
def test_shapes_match_oracle(N, C, K, kh, kw, stride, padding, Hz, Wz):
    ista_conv2d, _, _, _, orc = _mods()
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    g = torch.Generator().manual_seed(N * 100 + K)
    w = torch.randn(K, C, kh, kw, generator=g) / (kh * kw) ** 0.5
    x = torch.randn(N, C, (Hz - 1) * sh - 2 * ph + kh, (Wz - 1) * sw - 2 * pw + kw, generator=g)
    z0 = torch.randn(N, K, Hz, Wz, generator=g) * 0.05
    lr = 0.3 / max(w.pow(2).sum().item(), 1e-3)
    for fast in (True, False):
        ref = orc.conv_fista(x, z0, w, 0.1, stride=stride, padding=padding, fast=fast, maxiter=9, lr=lr, tol=0.0)
        got = ista_conv2d(x.cuda(), z0.cuda(), w.cuda(), 0.1, stride=stride, padding=padding, fast=fast,
                          maxiter=9, lr=lr, tol=0.0)
        assert got.shape == ref.shape
        if N:
            assert (got.cpu() - ref).abs().max().item() <= Z_ATOL


def _geom_args(N, C, K, kh, kw, ph, pw, Hz, Wz):
    return (N, C, (Hz - 1) - 2 * ph + kh, (Wz - 1) - 2 * pw + kw, K, Hz, Wz, kh, kw, 1, 1, ph, pw)


@pytest.mark.parametrize("n_spec,C,K,kh,kw,padding,Hz,Wz", [
    # n_spec >= 0: N = CUs + n_spec images (a workgroup per image, up to 64 iterations per launch);
    # n_spec < -1: N = CUs // -n_spec images, each cut into bands of code rows (one launch per iteration)
    # the BASELINE-like geometry: 49 taps in four column blocks, 64 atoms, 26-pixel code rows -- rows 4, 9, 14, ... hold
    # a multiple of 128 code pixels (the two-kernel form's chunk cuts, whose order of the taps the kernel reproduces)
    (0, 1, 64, 7, 7, 0, 12, 26),
    # 75 taps (five column blocks), K % 16 != 0, padding, three channels; more than two outputs per thread
    (5, 3, 24, 5, 5, 2, 10, 12),
    # non-square kernel and code grid, asymmetric padding, K <= 16, more images than workgroups
    (300, 2, 12, 3, 5, (1, 0), 20, 9),
    # one tap column block, K = 4, an image smaller than one MFMA row block
    (1, 1, 4, 3, 3, 1, 3, 4),
    # kernel width outside {3, 5, 7} (the masked tap loop), 48 atoms
    (2, 1, 48, 4, 4, 1, 9, 11),
    # more than 64 atoms: contracted in two halves by the synthesis, split four ways over the gradient waves; K % 16 != 0
    (0, 3, 128, 5, 5, 2, 10, 12), (3, 1, 100, 3, 3, 1, 9, 9),
    # fewer images than CUs: four bands of ten code rows with one row of halo either side (first / last band: rows of
    # the convolution's zero padding), y through two buffers
    (-4, 2, 48, 3, 3, 1, 40, 24),
    # two bands of twenty rows, 7 x 7 taps: the two-kernel form's chunk cuts counted from the first code row of ITS bands
    (-2, 1, 32, 7, 7, 3, 40, 26),
    # more than 4096 residual values per image: sixteen outputs per thread (the two-kernel form cuts such images into
    # bands even at N = CUs)
    (0, 3, 24, 5, 5, 2, 40, 38),
    # n_spec -1: N = CUs - 37 -- fewer images than CUs but still a workgroup per image (bands would cost 3x the
    # synthesis); the two-kernel form cuts THESE images into two bands, whose chunk cuts the overlap-add follows per row
    (-1, 1, 64, 7, 7, 0, 12, 26)])
def test_many_iterations_per_launch_kernel(n_spec, C, K, kh, kw, padding, Hz, Wz):
    """conv_fused.hip: small few-channel images run whole iterations in one kernel, a workgroup per image (up to 64
    iterations per launch) or per band of an image (one per launch).  Against the oracle (the usual fp32 bound), BITWISE
    against the two-kernel form (LASSO_CONV_FUSED=0: same lane -> operand assignment in both GEMMs, same order of the
    overlap-add), across the 64-iteration launch boundary, from a warm start, with ISTA and FISTA, and under the stop rule
    (the two-kernel form adds the sums in another order: compared through the replayed plain run)."""
    import os
    from lasso_amd import _native as nat
    ista_conv2d, _, _, _, orc = _mods()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    N = cus + n_spec if n_spec >= 0 else cus - 37 if n_spec == -1 else cus // -n_spec
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    assert b"conv_fused_kernel" in nat.lib().lasso_conv_ista_kernel_name(*_geom_args(N, C, K, kh, kw, ph, pw, Hz, Wz))
    os.environ["LASSO_CONV_FUSED"] = "0"
    try:
        assert b"conv_fused_kernel" not in nat.lib().lasso_conv_ista_kernel_name(*_geom_args(N, C, K, kh, kw, ph, pw, Hz, Wz))
    finally:
        os.environ.pop("LASSO_CONV_FUSED", None)
    g = torch.Generator().manual_seed(K * 7 + Hz)
    w = torch.randn(K, C, kh, kw, generator=g) / (kh * kw) ** 0.5
    x = torch.randn(N, C, (Hz - 1) - 2 * ph + kh, (Wz - 1) - 2 * pw + kw, generator=g)
    z0 = torch.randn(N, K, Hz, Wz, generator=g) * 0.05
    lr = 0.3 / max(w.pow(2).sum().item(), 1e-3)
    xg, wg, zg = x.cuda(), w.cuda(), z0.cuda()

    def both(**kw_):
        os.environ.pop("LASSO_CONV_FUSED", None)
        a = ista_conv2d(xg, zg, wg, 0.1, padding=padding, lr=lr, **kw_)
        os.environ["LASSO_CONV_FUSED"] = "0"
        try:
            b = ista_conv2d(xg, zg, wg, 0.1, padding=padding, lr=lr, **kw_)
        finally:
            os.environ.pop("LASSO_CONV_FUSED", None)
        return a, b

    n_ref = min(N, 24)                                   # (the oracle on the first images: they are independent)
    for fast in (True, False):
        ref = orc.conv_fista(x[:n_ref], z0[:n_ref], w, 0.1, padding=padding, fast=fast, maxiter=9, lr=lr, tol=0.0)
        one, two = both(fast=fast, maxiter=9, tol=0.0)
        assert (one[:n_ref].cpu() - ref).abs().max().item() <= Z_ATOL
        assert torch.equal(one, two), (fast, (one - two).abs().max().item())
    one, two = both(maxiter=70, tol=0.0)                 # 64 + 6 iterations: two launches
    assert torch.equal(one, two)
    # the stop rule: a budget the sums cross inside a speculated chunk (just above the sum of iteration 23)
    _, probe = ista_conv2d(xg, zg, wg, 0.1, padding=padding, lr=lr, maxiter=23, tol=1e-30, return_info=True)
    tol = float(np.float32(probe["last_delta"]) * np.float32(1.0 + 1e-4)) / z0.numel()
    (one, info1), (two, info2) = both(maxiter=150, tol=tol, return_info=True)
    assert 1 < info1["iterations"] <= 23, info1
    assert abs(info1["iterations"] - info2["iterations"]) <= 1, (info1, info2)     # (sums added in another order)
    os.environ.pop("LASSO_CONV_FUSED", None)
    plain = ista_conv2d(xg, zg, wg, 0.1, padding=padding, lr=lr, maxiter=info1["iterations"], tol=0.0)
    assert torch.equal(one, plain)
    assert abs(info1["last_delta"] - info2["last_delta"]) <= 1e-5 * abs(info2["last_delta"]) or info1["iterations"] != info2["iterations"]


@pytest.mark.parametrize("form", ["fused", "bands", "two-kernel"])
def test_stop_rule_count_pinned_exactly_against_the_oracle(form):
    """ista.py:44-46 with an explicit step: the iteration at which the rule fires is the ORACLE's, exactly -- not +-1 --
    on the one-launch kernel (a workgroup per image), on its banded form and on the two-kernel form.  The budget is
    placed where the oracle's sums leave room: at least 1 % away from the sum of the iteration that fires and from
    every earlier one, far beyond the 1e-6-level differences of the summation orders (the three
    `<= 1` tolerances elsewhere in this file stay for lr='auto' and for budgets placed ON a sum)."""
    import math
    import os
    from lasso_amd import _native as nat
    ista_conv2d, _, _, _, orc = _mods()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    if form == "bands":
        N, C, K, ks, pd, Hz = max(cus // 4, 8), 1, 8, 3, 1, 32         # fewer images than a third of the CUs: bands of code rows
    else:
        N, C, K, ks, pd, Hz = cus, 1, 8, 3, 1, 12                      # an image per CU
    g = torch.Generator().manual_seed(3)
    w = torch.randn(K, C, ks, ks, generator=g) / ks
    H = (Hz - 1) - 2 * pd + ks
    x = torch.randn(N, C, H, H, generator=g)
    z0 = torch.zeros(N, K, Hz, Hz)
    lr = 0.5 / w.pow(2).sum().item()
    # the oracle's sum of every iteration (one run per length: the loop is deterministic)
    sums = [orc.conv_fista(x, z0, w, 0.1, padding=pd, maxiter=m, lr=lr, tol=0.0, return_info=True)[1]["last_delta"]
            for m in range(1, 31)]
    # an iteration m whose sum is the first below a budget with >= 1 % of room on both sides
    pick = None
    for m in range(8, 30):
        lo, hi = sums[m - 1], min(sums[:m - 1])
        if lo < hi and hi / lo >= 1.03:
            pick, budget = m, math.sqrt(lo * hi)
            break
    assert pick is not None, sums
    tol = budget / z0.numel()
    _, rinfo = orc.conv_fista(x, z0, w, 0.1, padding=pd, maxiter=200, lr=lr, tol=tol, return_info=True)
    assert rinfo["iterations"] == pick
    args = _geom_args(N, C, K, ks, ks, pd, pd, Hz, Hz)
    if form == "two-kernel":
        os.environ["LASSO_CONV_FUSED"] = "0"
    try:
        name = nat.lib().lasso_conv_ista_kernel_name(*args)
        assert (b"conv_fused_kernel" in name) == (form != "two-kernel"), name
        got, info = ista_conv2d(x.cuda(), z0.cuda(), w.cuda(), 0.1, padding=pd, maxiter=200, lr=lr, tol=tol,
                                return_info=True)
    finally:
        os.environ.pop("LASSO_CONV_FUSED", None)
    assert info["iterations"] == pick, (form, info, rinfo, sums[pick - 2:pick + 1])
    ref = orc.conv_fista(x, z0, w, 0.1, padding=pd, maxiter=pick, lr=lr, tol=0.0)
    assert (got.cpu() - ref).abs().max().item() <= Z_ATOL


def test_errors_and_edge_cases():
    ista_conv2d, _, _, _, _ = _mods()
    w = torch.randn(4, 2, 3, 3, device="cuda")
    x = torch.randn(2, 2, 10, 10, device="cuda")
    z0 = torch.zeros(2, 4, 8, 8, device="cuda")
    assert ista_conv2d(x, z0, w, maxiter=0, lr=0.1) is z0                      # ista.py:32,49
    with pytest.raises(RuntimeError):
        ista_conv2d(x, torch.zeros(2, 4, 7, 8, device="cuda"), w, lr=0.1)      # x_hat - x size mismatch
    with pytest.raises(RuntimeError):
        ista_conv2d(x, z0, torch.randn(4, 3, 3, 3, device="cuda"), lr=0.1)     # channel mismatch
    with pytest.raises(TypeError):
        ista_conv2d(x, z0, w, backtrack=True)                                  # not an ista_conv2d kwarg
    # CPU tensors are staged through the device, result comes back on the CPU
    out = ista_conv2d(x.cpu(), z0.cpu(), w.cpu(), 0.1, lr=0.05, maxiter=3)
    assert out.device.type == "cpu"


def test_verbose_prints_reference_format(capsys):
    ista_conv2d, _, _, _, orc = _mods()
    g = torch.Generator().manual_seed(3)
    w = torch.randn(4, 1, 3, 3, generator=g) * 0.3
    x = torch.randn(2, 1, 8, 8, generator=g)
    z0 = torch.zeros(2, 4, 6, 6)
    z = ista_conv2d(x.cuda(), z0.cuda(), w.cuda(), 0.1, lr=0.1, maxiter=3, tol=0.0, verbose=True)
    lines = capsys.readouterr().out.strip().splitlines()
    assert len(lines) == 3 and all(l.startswith("loss: ") for l in lines)
    assert abs(float(lines[0][6:]) - orc.conv_objective(x, z0, w, 0.1).item()) <= 1e-3
    ref = orc.conv_fista(x, z0, w, 0.1, lr=0.1, maxiter=3, tol=0.0)
    assert (z.cpu() - ref).abs().max().item() <= Z_ATOL
