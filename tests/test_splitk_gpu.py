"""The small-batch (split-k) FISTA kernel -- a 16-row tile shared by Kpad/128 workgroups,
csrc/fista_splitk.hip -- against the one-workgroup-per-tile kernel and the oracle: the two
kernels are the same arithmetic, so a row's code must be BITWISE the same whichever runs it
(and therefore independent of how the batch is sharded over GPUs); fixed iteration counts,
warm starts, chunked state hand-over, the in-kernel stop rule."""
import pytest
import torch

from recipes import recipe_xw, LAMBDA_MAX_C2

pytestmark = pytest.mark.gpu


def _case(n, d, k, seed=1):
    g = torch.Generator().manual_seed(seed)
    W = torch.nn.functional.normalize(torch.randn(d, k, generator=g), dim=0)
    X = torch.randn(n, d, generator=g)
    return X, W


@pytest.mark.parametrize("n,d,k", [(16, 256, 1024), (37, 256, 1024), (512, 256, 1024), (1000, 256, 1024),
                                   (2048, 256, 1024), (3000, 200, 1000), (130, 256, 512), (777, 150, 600),
                                   (90, 200, 256), (1, 129, 130), (4096, 256, 512), (2100, 200, 256)])
@pytest.mark.parametrize("fast", [True, False])
def test_split_kernel_is_bitwise_the_tile_kernel(n, d, k, fast):
    from lasso_amd.linear.solvers import ista
    from oracle import lasso_oracle as orc
    X, W = _case(n, d, k, seed=n + k)
    Xg, Wg = X.cuda(), W.cuda()
    lr = 1.0 / orc.lipschitz_constant(W, "exact")
    g = torch.Generator().manual_seed(3)
    warm = (torch.randn(n, k, generator=g) * (torch.rand(n, k, generator=g) < 0.1)).cuda()
    for z0 in (torch.zeros(n, k, device="cuda"), warm):
        for iters in (1, 2, 9, 40):
            zt = ista(Xg, z0, Wg, 0.3, fast=fast, lr=lr, maxiter=iters, tol=0.0, kernel='tile')
            # cost-model choice, then T = 1, 2, 4 tiles per group (T >= 2: partials streamed by LDS-DMA; '..g': gathered in registers)
            for kern in ('splitk', 'splitk1', 'splitk2', 'splitk4', 'splitk2g', 'splitk4g', 'splitk1s'):
                zs = ista(Xg, z0, Wg, 0.3, fast=fast, lr=lr, maxiter=iters, tol=0.0, kernel=kern)
                assert torch.equal(zt, zs), (kern, iters, (zt - zs).abs().max().item())
    ref = orc.fista(X, X.new_zeros(n, k), W, 0.3, fast=fast, lr=lr, maxiter=40, tol=0.0)
    assert (zs.cpu() - orc.fista(X, warm.cpu(), W, 0.3, fast=fast, lr=lr, maxiter=40, tol=0.0)).abs().max() <= 5e-5
    zc = ista(Xg, torch.zeros(n, k, device="cuda"), Wg, 0.3, fast=fast, lr=lr, maxiter=40, tol=0.0, kernel='splitk')
    assert (zc.cpu() - ref).abs().max().item() <= 5e-5


def test_split_kernel_state_handover_in_chunks():
    """lasso_fista_run chunks (the building block of the multi-GPU exact stop rule): (z, y) and the
    per-iteration deltas carried across launches, split-k and tile kernel alternating."""
    from lasso_amd.engine import HipEngine
    from lasso_amd import _native as nat
    X, W = recipe_xw(700)
    Xg, Wg = X.cuda(), W.cuda()
    eng = HipEngine()
    lr = 1.0 / LAMBDA_MAX_C2
    z_ref, y_ref, d_ref = eng.fista_run(Xg, Wg, None, None, 0.5, lr, True, 0, 30, True, kernel=nat.KERNEL_TILE)
    ws = eng.fista_workspace(700, 256, 1024, 30)
    z, y, deltas = None, None, []
    for it0, c, kern in ((0, 7, nat.KERNEL_SPLITK), (7, 13, nat.KERNEL_TILE), (20, 10, nat.KERNEL_SPLITK)):
        z, y, dl = eng.fista_run(Xg, Wg, z, y, 0.5, lr, True, it0, c, True, ws=ws, kernel=kern)
        deltas.append(dl)
    assert torch.equal(z, z_ref) and torch.equal(y, y_ref)
    got = torch.cat(deltas)
    # the per-iteration sums add the same numbers in a different (fixed) order
    assert torch.allclose(got, d_ref, rtol=2e-6, atol=0)
    # ... and are bitwise reproducible
    z2, y2, d2 = eng.fista_run(Xg, Wg, None, None, 0.5, lr, True, 0, 7, True, kernel=nat.KERNEL_SPLITK)
    assert torch.equal(d2, deltas[0])


@pytest.mark.parametrize("n", [512, 100, 16, 1024, 2048, 1500])
def test_split_kernel_in_kernel_stop_rule(n):
    """n <= 512 rows (the per-GPU shard of BASELINE's n=4096 on 8 GPUs): the stop rule is evaluated
    inside the split-k kernel; iteration count = the oracle's, code = the tile kernel's."""
    from lasso_amd.linear.solvers import ista
    from oracle import lasso_oracle as orc
    X, W = recipe_xw(n)
    Xg, Wg = X.cuda(), W.cuda()
    z0 = torch.zeros(n, 1024, device="cuda")
    lr = 1.0 / LAMBDA_MAX_C2
    for fast, tol in ((True, 1e-4), (False, 3e-4), (True, 1e-5))[:3 if n <= 1024 else 2]:   # (the 263-iteration CPU run: small batches only)
        tr = orc.FistaTrace()
        orc.fista(X, X.new_zeros(n, 1024), W, 0.5, fast=fast, lr=lr, maxiter=1000, tol=tol, trace=tr)
        zs, info_s = ista(Xg, z0, Wg, 0.5, fast=fast, lr=lr, maxiter=1000, tol=tol, return_info=True,
                          kernel='splitk')
        zt, info_t = ista(Xg, z0, Wg, 0.5, fast=fast, lr=lr, maxiter=1000, tol=tol, return_info=True,
                          kernel='tile')
        assert info_s["iterations"] == tr.iterations == info_t["iterations"], (info_s, info_t, tr.iterations)
        assert torch.equal(zs, zt)
        # maxiter below the stopping point
        z9, info9 = ista(Xg, z0, Wg, 0.5, fast=fast, lr=lr, maxiter=9, tol=tol, return_info=True, kernel='splitk')
        assert info9["iterations"] == 9
        assert torch.equal(z9, ista(Xg, z0, Wg, 0.5, fast=fast, lr=lr, maxiter=9, tol=0.0, kernel='tile'))


def test_auto_dispatch_names():
    from lasso_amd import _native as nat
    L = nat.lib()
    assert b"splitk" in L.lasso_fista_kernel_name(512, 256, 1024, nat.LASSO_F32, 0)
    assert b"fista_splitk_rs_kernel<2>" in L.lasso_fista_kernel_name(1024, 256, 1024, nat.LASSO_F32, 0)
    assert b"fista_tile_sp" in L.lasso_fista_kernel_name(4096, 256, 1024, nat.LASSO_F32, 0)
    assert b"fista_tile_sp" in L.lasso_fista_kernel_name(512, 64, 256, nat.LASSO_F32, 0)     # tall tiles: no split


@pytest.mark.parametrize("n", [1425, 600, 2500, 3300])
def test_chunked_stop_rule_through_multi_round_split_launches(n):
    """The chunked evaluation of the stop rule records per-iteration sums from whatever plan the
    cost model picks -- including split-k launches of several rounds whose last round leaves
    some groups without a tile: same iteration count as the oracle and the in-kernel rule."""
    from lasso_amd.linear.solvers import ista
    from oracle import lasso_oracle as orc
    X, W = recipe_xw(n)
    Xg, Wg = X.cuda(), W.cuda()
    z0 = torch.zeros(n, 1024, device="cuda")
    lr = 1.0 / LAMBDA_MAX_C2
    for fast, tol in ((False, 1e-3), (True, 3e-4)):
        tr = orc.FistaTrace()
        orc.fista(X, X.new_zeros(n, 1024), W, 0.5, fast=fast, lr=lr, maxiter=500, tol=tol, trace=tr)
        for kern in ('auto', 'splitk1', 'splitk2', 'splitk4', 'tile'):
            z, info = ista(Xg, z0, Wg, 0.5, fast=fast, lr=lr, maxiter=500, tol=tol, return_info=True,
                           stop_mode='chunked', kernel=kern)
            assert info["iterations"] == tr.iterations, (kern, info, tr.iterations)
            assert abs(info["last_delta"] - tr.delta[-1]) <= 2e-6 * tr.delta[-1]


@pytest.mark.parametrize("n,d,k", [(4096, 256, 768), (3500, 200, 700), (8192, 256, 640)])
def test_768_atom_tile_kernel_is_bitwise_the_1024_atom_kernels(n, d, k):
    """512 < k <= 768 on a large batch runs the 768-atom instantiation of the tile kernel (a quarter less work than
    the padding to 1024); the same rows in a small batch run the split-k kernel padded to 1024.  The canonical
    128-atom slice order makes a row's code independent of both (the padded slices add exact zeros)."""
    from lasso_amd.linear.solvers import ista
    from lasso_amd import _native as nat
    from oracle import lasso_oracle as orc
    assert b"fista_tile_sp_kernel<768" in nat.lib().lasso_fista_kernel_name(n, d, k, nat.LASSO_F32, 0)
    assert b"<768" not in nat.lib().lasso_fista_kernel_name(512, d, k, nat.LASSO_F32, 0)
    X, W = _case(n, d, k, seed=n + k)
    Xg, Wg = X.cuda(), W.cuda()
    lr = 1.0 / orc.lipschitz_constant(W, "exact")
    for iters in (1, 25):
        z = ista(Xg, torch.zeros(n, k, device="cuda"), Wg, 0.3, lr=lr, maxiter=iters, tol=0.0)
        zs = ista(Xg[:512], torch.zeros(512, k, device="cuda"), Wg, 0.3, lr=lr, maxiter=iters, tol=0.0)
        assert torch.equal(z[:512], zs), (iters, (z[:512] - zs).abs().max().item())
    ref = orc.fista(X[:300], X.new_zeros(300, k), W, 0.3, lr=lr, maxiter=25, tol=0.0)
    assert (z[:300].cpu() - ref).abs().max().item() <= 5e-5
    if n > 4096:
        return      # (the whole-batch CPU run to tolerance below: the two smaller batches carry it)
    # the stop rule (in-kernel at 4096 rows, chunked beyond) and lr='auto' on the 768-atom kernel
    tr = orc.FistaTrace()
    orc.fista(X, X.new_zeros(n, k), W, 0.3, lr=lr, maxiter=400, tol=1e-4, trace=tr)
    _, info = ista(Xg, torch.zeros(n, k, device="cuda"), Wg, 0.3, lr=lr, maxiter=400, tol=1e-4, return_info=True)
    assert info["iterations"] == tr.iterations
    za = ista(Xg, torch.zeros(n, k, device="cuda"), Wg, 0.3, maxiter=10)
    zb = ista(Xg, torch.zeros(n, k, device="cuda"), Wg, 0.3, lr=1.0 / orc.lipschitz_constant(W, "exact"), maxiter=10)
    assert (za - zb).abs().max().item() <= 1e-4


@pytest.mark.parametrize("n,d,k", [(4096, 100, 300), (700, 128, 384), (9000, 65, 257)])
def test_384_atom_tile_kernels(n, d, k):
    """256 < k <= 384 with d <= 128 runs the 384-atom instantiations of the 32 x 128 / 16 x 128 tile kernels (a quarter
    less padding than 512): against the oracle, the stop rule, lr='auto'; rows independent of the batch they sit in."""
    from lasso_amd.linear.solvers import ista
    from lasso_amd import _native as nat
    from oracle import lasso_oracle as orc
    assert b"fista_tile_sp_kernel<384" in nat.lib().lasso_fista_kernel_name(n, d, k, nat.LASSO_F32, 0)
    X, W = _case(n, d, k, seed=n + k)
    Xg, Wg = X.cuda(), W.cuda()
    lr = 1.0 / orc.lipschitz_constant(W, "exact")
    z = ista(Xg, torch.zeros(n, k, device="cuda"), Wg, 0.3, lr=lr, maxiter=25, tol=0.0)
    ref = orc.fista(X[:400], X.new_zeros(min(n, 400), k), W, 0.3, lr=lr, maxiter=25, tol=0.0)
    assert (z[:400].cpu() - ref).abs().max().item() <= 5e-5
    zs = ista(Xg[:100], torch.zeros(100, k, device="cuda"), Wg, 0.3, lr=lr, maxiter=25, tol=0.0)
    assert torch.equal(z[:100], zs)
    tr = orc.FistaTrace()
    orc.fista(X, X.new_zeros(n, k), W, 0.3, lr=lr, maxiter=400, tol=1e-4, trace=tr)
    _, info = ista(Xg, torch.zeros(n, k, device="cuda"), Wg, 0.3, lr=lr, maxiter=400, tol=1e-4, return_info=True)
    assert info["iterations"] == tr.iterations
    za = ista(Xg, torch.zeros(n, k, device="cuda"), Wg, 0.3, maxiter=10)
    zb = ista(Xg, torch.zeros(n, k, device="cuda"), Wg, 0.3, lr=lr, maxiter=10)
    assert (za - zb).abs().max().item() <= 1e-4


@pytest.mark.parametrize("n,d,k", [(4096, 64, 1024), (5000, 100, 1000), (4352, 128, 768), (4100, 33, 600)])
def test_narrow_tiles_for_short_rows_and_large_dictionaries(n, d, k):
    """d <= 128 with more than 512 atoms on a batch that fills the chip: 4-wave workgroups on 16 x 128 tiles instead of
    8 waves on 16 x 256 (half the padded work; d=64, k=1024: 34 -> 60 TFLOP/s useful).  Bitwise the code of the wide
    kernel and of the same rows in a small batch (split-k kernel), the oracle's code, the oracle's iteration count."""
    from lasso_amd.linear.solvers import ista
    from lasso_amd import _native as nat
    from oracle import lasso_oracle as orc
    name = nat.lib().lasso_fista_kernel_name(n, d, k, nat.LASSO_F32, 0)
    assert b"16, false, 4>" in name, name
    X, W = _case(n, d, k, seed=n + k)
    Xg, Wg = X.cuda(), W.cuda()
    lr = 1.0 / orc.lipschitz_constant(W, "exact")
    z = ista(Xg, torch.zeros(n, k, device="cuda"), Wg, 0.3, lr=lr, maxiter=25, tol=0.0)
    zs = ista(Xg[:512], torch.zeros(512, k, device="cuda"), Wg, 0.3, lr=lr, maxiter=25, tol=0.0)
    assert torch.equal(z[:512], zs)
    g = torch.Generator().manual_seed(5)
    warm = (torch.randn(n, k, generator=g) * (torch.rand(n, k, generator=g) < 0.1)).cuda()
    zw = ista(Xg, warm, Wg, 0.3, fast=False, lr=lr, maxiter=9, tol=0.0)
    assert torch.equal(zw[:300], ista(Xg[:300], warm[:300], Wg, 0.3, fast=False, lr=lr, maxiter=9, tol=0.0))
    ref = orc.fista(X[:300], X.new_zeros(300, k), W, 0.3, lr=lr, maxiter=25, tol=0.0)
    assert (z[:300].cpu() - ref).abs().max().item() <= 5e-5
    tr = orc.FistaTrace()
    orc.fista(X, X.new_zeros(n, k), W, 0.3, lr=lr, maxiter=400, tol=1e-4, trace=tr)
    for mode in ('global', 'chunked'):
        _, info = ista(Xg, torch.zeros(n, k, device="cuda"), Wg, 0.3, lr=lr, maxiter=400, tol=1e-4, return_info=True,
                       stop_mode=mode)
        assert info["iterations"] == tr.iterations, (mode, info, tr.iterations)


def _device_cus():
    import ctypes
    from lasso_amd import _native as nat
    c = ctypes.c_int(0)
    assert nat.lib().lasso_hip_device_cus(ctypes.byref(c)) == 0
    return c.value


# (rounds of 16-row tiles on the device's CUs, rows past the last full round, atoms): sized from the CU count, so the
# cases are ragged on any part (the sizes were written for 256 CUs)
@pytest.mark.parametrize("rounds,extra,k", [(1, 4, 1024), (1, 904, 1024), (1, 2104, 1024), (2, 8, 1024), (1, 404, 512),
                                            (2, 808, 300)])
def test_ragged_batches_run_their_last_round_on_the_split_kernel(rounds, extra, k):
    n = _device_cus() * 16 * rounds + extra
    _ragged_case(n, k, _device_cus() * 32)


def _ragged_case(n, k, n_whole):
    """More tiles than CUs, the last round partly filled: the full rounds run on the tile kernel, the tail's tiles on
    the split-k kernel (lasso_hip.hip run_impl).  Codes bitwise those of the tile kernel alone -- cold and warm start,
    FISTA and ISTA --; the per-iteration sums of |z - z_next| (tile rows + split-k rows) agree with it to fp32 rounding,
    through both reduction kernels; chained chunks (y state in and out) stay bitwise; the stop rule's iteration count
    is the tile kernel's."""
    from lasso_amd.linear.solvers import ista
    from lasso_amd.engine import HipEngine
    from lasso_amd import _native as nat
    name = nat.lib().lasso_fista_kernel_name(n, 256, k, nat.LASSO_F32, 0)
    assert b"split-k" in name or b"splitk" in name          # (tiny tails: the cost model may give the whole batch to split-k)
    assert b"split-k" not in nat.lib().lasso_fista_kernel_name(n_whole, 256, k, nat.LASSO_F32, 0)   # whole rounds: tile kernel only
    X, W = _case(n, 256, k, seed=n + k)
    Xg, Wg = X.cuda(), W.cuda()
    lr = 0.9 / ((k / 256) * (1.0 + (256 / k) ** 0.5) ** 2)
    z0 = torch.zeros(n, k, device="cuda")
    g = torch.Generator().manual_seed(7)
    warm = (torch.randn(n, k, generator=g) * (torch.rand(n, k, generator=g) < 0.1)).cuda()
    for fast, start, iters in ((True, z0, 12), (False, warm, 5)):
        za = ista(Xg, start, Wg, 0.4, fast=fast, lr=lr, maxiter=iters, tol=0.0)
        zt = ista(Xg, start, Wg, 0.4, fast=fast, lr=lr, maxiter=iters, tol=0.0, kernel='tile')
        assert torch.equal(za, zt), (fast, (za - zt).abs().max().item())
    eng = HipEngine()
    z1, y1, d1 = eng.fista_run(Xg, Wg, None, None, 0.4, lr, True, 0, 6, True)
    z2, y2, d2 = eng.fista_run(Xg, Wg, z1, y1, 0.4, lr, True, 6, 5, True)
    t1, u1, e1 = eng.fista_run(Xg, Wg, None, None, 0.4, lr, True, 0, 6, True, kernel=nat.KERNEL_TILE)
    t2, u2, e2 = eng.fista_run(Xg, Wg, t1, u1, 0.4, lr, True, 6, 5, True, kernel=nat.KERNEL_TILE)
    assert torch.equal(z2, t2) and torch.equal(y2, u2)
    assert (torch.cat([d1, d2]) / torch.cat([e1, e2]) - 1).abs().max().item() <= 2e-6
    _, ia = ista(Xg, z0, Wg, 0.4, lr=lr, maxiter=300, tol=2e-4, return_info=True)
    _, it = ista(Xg, z0, Wg, 0.4, lr=lr, maxiter=300, tol=2e-4, return_info=True, kernel='tile')
    assert ia["iterations"] == it["iterations"] and 5 < ia["iterations"] < 300
    assert abs(ia["last_delta"] - it["last_delta"]) <= 2e-6 * it["last_delta"]
