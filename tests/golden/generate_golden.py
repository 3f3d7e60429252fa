#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REAL
reference (rfeinman/pytorch-lasso, mounted read-only at /root/reference).

Runs only in the build container (the reference does not travel to the GPU
box); the .npz files it writes are data (inputs are re-creatable from the
seeded recipes in tests/recipes.py; expected outputs are stored).  The
reference ships no tests/golden vectors of its own, so
these files are what pins parity.

The reference does not import as-is under scipy 1.15 (iterative_ridge.py:5
imports a private scipy symbol); the 3-line shim below aliases that symbol
BEFORE importing -- no reference file is edited or copied.

Usage:  python tests/golden/generate_golden.py [g1 g2 g3 g3trace g4 g5 small inits cd conv]
"""
import os
import sys
import time

sys.dont_write_bytecode = True
import numpy as np
import scipy.optimize.optimize as _so
from scipy.optimize import _optimize as _o

_so._status_message = _o._status_message          # shim
sys.path.insert(0, "/root/reference")
import torch  # noqa: E402
import lasso  # noqa: E402,F401
from lasso.linear import dict_learning, sparse_encode  # noqa: E402

ref_ista_mod = sys.modules["lasso.linear.solvers.ista"]
ref_dl_mod = sys.modules["lasso.linear.dict_learning"]
ref_ista = ref_ista_mod.ista
ref_loss = ref_dl_mod.lasso_loss

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from recipes import (recipe_xw, recipe_c4_init, recipe_c5, LAMBDA_MAX_C2,  # noqa: E402
                     LAMBDA_MAX_C4)


def zstats(z):
    z64 = z.double()
    return dict(sum=z64.sum().item(), abssum=z64.abs().sum().item(),
                nnz=int((z != 0).sum().item()))


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print("wrote", path, os.path.getsize(path), "bytes")


def g1():
    """README plumbing (BASELINE config 1)."""
    torch.manual_seed(0)
    data = torch.randn(100, 10)
    D, losses = dict_learning(data, 50, alpha=0.5, algorithm="ista", progbar=False)
    z = sparse_encode(data, D, alpha=0.2, algorithm="ista")
    loss_z = ref_loss(data, z, D, 0.2)
    # explicit-lr variant: bitwise reproducible on CPU
    torch.manual_seed(0)
    data2 = torch.randn(100, 10)
    assert torch.equal(data, data2)
    # capture the RNG-dependent init so the HIP path can start from the same D0
    torch.manual_seed(0)
    _ = torch.randn(100, 10)
    D0 = torch.empty(10, 50)
    torch.nn.init.orthogonal_(D0)
    D0 = torch.nn.functional.normalize(D0, dim=0)
    torch.manual_seed(0)
    _ = torch.randn(100, 10)
    Dfix, losses_fix = dict_learning(data, 50, alpha=0.5, algorithm="ista",
                                     progbar=False, lr=0.05)
    zfix = sparse_encode(data, Dfix, alpha=0.2, algorithm="ista", lr=0.05)
    # unconstrained (ridge) variant
    torch.manual_seed(0)
    _ = torch.randn(100, 10)
    Dr, losses_r = dict_learning(data, 50, alpha=0.5, constrained=False,
                                 algorithm="ista", progbar=False, lr=0.05)
    # persist variant
    torch.manual_seed(0)
    _ = torch.randn(100, 10)
    Dp, losses_p = dict_learning(data, 50, alpha=0.5, persist=True,
                                 algorithm="ista", progbar=False, lr=0.05)
    save("g1_readme", data=data.numpy(), D0=D0.numpy(),
         D_auto=D.numpy(), losses_auto=losses.numpy(), z_auto=z.numpy(),
         loss_z_auto=loss_z.item(),
         D_fix=Dfix.numpy(), losses_fix=losses_fix.numpy(), z_fix=zfix.numpy(),
         D_ridge=Dr.numpy(), losses_ridge=losses_r.numpy(),
         D_persist=Dp.numpy(), losses_persist=losses_p.numpy())


def g2():
    """C2: FISTA/ISTA fixed step, n=4096 d=256 k=1024 fp32."""
    X, W = recipe_xw(4096, 256, 1024)
    lr = 1.0 / LAMBDA_MAX_C2
    z0 = X.new_zeros(4096, 1024)
    out = {}
    Ms = [1, 2, 5, 10, 50, 100, 263, 1000]
    obj, st = [], []
    for M in Ms:
        z = ref_ista(X, z0, W, 0.5, fast=True, lr=lr, maxiter=M, tol=0.0)
        obj.append(ref_loss(X, z, W, 0.5).item())
        s = zstats(z)
        st.append([s["sum"], s["abssum"], s["nnz"]])
        out["z_block_M%d" % M] = z[:64, :64].numpy().copy()
        out["z_strided_M%d" % M] = z[::64, ::16].numpy().copy()
        print("G2 M=%d obj=%.6f" % (M, obj[-1]), s)
    out["Ms"] = np.array(Ms)
    out["objective"] = np.array(obj, dtype=np.float64)
    out["stats"] = np.array(st, dtype=np.float64)
    # ISTA (fast=False)
    obj_i = []
    for M in [1, 10, 100]:
        z = ref_ista(X, z0, W, 0.5, fast=False, lr=lr, maxiter=M, tol=0.0)
        obj_i.append(ref_loss(X, z, W, 0.5).item())
        out["ista_z_block_M%d" % M] = z[:64, :64].numpy().copy()
    out["ista_Ms"] = np.array([1, 10, 100])
    out["ista_objective"] = np.array(obj_i)
    # to-tolerance runs (default tol=1e-5), explicit lr -> bitwise reproducible
    z = ref_ista(X, z0, W, 0.5, fast=True, lr=lr, maxiter=2000, tol=1e-5)
    s = zstats(z)
    out["tol_fista_obj"] = ref_loss(X, z, W, 0.5).item()
    out["tol_fista_stats"] = np.array([s["sum"], s["abssum"], s["nnz"]])
    out["tol_fista_z_block"] = z[:64, :64].numpy().copy()
    z = ref_ista(X, z0, W, 0.5, fast=False, lr=lr, maxiter=3000, tol=1e-5)
    s = zstats(z)
    out["tol_ista_obj"] = ref_loss(X, z, W, 0.5).item()
    out["tol_ista_stats"] = np.array([s["sum"], s["abssum"], s["nnz"]])
    # warm start: z0 = result of 5 iterations
    zw = ref_ista(X, z0, W, 0.5, fast=True, lr=lr, maxiter=5, tol=0.0)
    z = ref_ista(X, zw, W, 0.5, fast=True, lr=lr, maxiter=5, tol=0.0)
    out["warm_obj"] = ref_loss(X, z, W, 0.5).item()
    out["warm_z_block"] = z[:64, :64].numpy().copy()
    out["check_W"] = W[0, :3].numpy()
    out["check_X"] = X[0, :3].numpy()
    save("g2_c2_fista", **out)


def g2b():
    """C2 at alpha = 0.1 (iterations-to-tol FISTA 766 / ISTA 1963 with tol=1e-5): the reference's iteration
    counts, objectives and one z block with the explicit step 1/lambda_max (bitwise reproducible)."""
    X, W = recipe_xw(4096, 256, 1024)
    lr = 1.0 / LAMBDA_MAX_C2
    z0 = X.new_zeros(4096, 1024)
    out = {}
    for name, fast, cap in (("fista", True, 3000), ("ista", False, 6000)):
        # the reference returns no iteration count: count the evaluations of its stop test (ista.py:93) by watching
        # torch.Tensor.abs on [n, k] operands inside the call
        count = {"n": 0}
        orig_abs = torch.Tensor.abs

        def counting_abs(self, *a, **kw):
            if self.dim() == 2 and self.shape == (4096, 1024):
                count["n"] += 1
            return orig_abs(self, *a, **kw)
        torch.Tensor.abs = counting_abs
        try:
            t0 = time.time()
            z = ref_ista(X, z0, W, 0.1, fast=fast, lr=lr, maxiter=cap, tol=1e-5)
        finally:
            torch.Tensor.abs = orig_abs
        s = zstats(z)
        out[name + "_iterations"] = np.array(count["n"])
        out[name + "_obj"] = ref_loss(X, z, W, 0.1).item()
        out[name + "_stats"] = np.array([s["sum"], s["abssum"], s["nnz"]])
        out[name + "_z_block"] = z[:64, :64].numpy().copy()
        print("G2b alpha=0.1 %s: %d iterations, obj %.6f, %.1f s" % (name, count["n"], out[name + "_obj"], time.time() - t0), s)
    save("g2b_c2_alpha01", **out)


def g3():
    """C3: FISTA + backtracking, n=16384 d=256 k=1024."""
    X, W = recipe_xw(16384, 256, 1024)
    z0 = X.new_zeros(16384, 1024)
    out = {}
    z = ref_ista(X, z0, W, 0.5, fast=True, lr=1.0, maxiter=10, tol=0.0,
                 backtrack=True)
    out["fp32_obj"] = ref_loss(X, z, W, 0.5).item()
    out["fp32_z_block"] = z[:64, :64].numpy().copy()
    s = zstats(z)
    out["fp32_stats"] = np.array([s["sum"], s["abssum"], s["nnz"]])
    print("G3 fp32 obj", out["fp32_obj"])
    # lr0 = 1/L -> one trial per iteration
    z = ref_ista(X, z0, W, 0.5, fast=True, lr=1.0 / LAMBDA_MAX_C2, maxiter=10,
                 tol=0.0, backtrack=True)
    out["fp32_obj_lrL"] = ref_loss(X, z, W, 0.5).item()
    # ISTA + backtracking
    z = ref_ista(X, z0, W, 0.5, fast=False, lr=1.0, maxiter=5, tol=0.0,
                 backtrack=True)
    out["fp32_ista_bt_obj"] = ref_loss(X, z, W, 0.5).item()
    # bf16
    Xb, Wb = X.bfloat16(), W.bfloat16()
    zb = ref_ista(Xb, Xb.new_zeros(16384, 1024), Wb, 0.5, fast=True, lr=1.0,
                  maxiter=10, tol=0.0, backtrack=True)
    out["bf16_obj_fp32eval"] = ref_loss(Xb.float(), zb.float(), Wb.float(), 0.5).item()
    print("G3 bf16 obj", out["bf16_obj_fp32eval"])
    zb = ref_ista(Xb, Xb.new_zeros(16384, 1024), Wb, 0.5, fast=True,
                  lr=1.0 / LAMBDA_MAX_C2, maxiter=10, tol=0.0)
    out["bf16_fixed_obj_fp32eval"] = ref_loss(Xb.float(), zb.float(), Wb.float(), 0.5).item()
    save("g3_c3_backtrack", **out)


def g3b():
    """C3 on bf16 tensors, the FIRST iterations of the real reference (the bf16 kernels were anchored to
    the reference by one scalar).  After one or two iterations the reference's bf16 arithmetic (bf16 tensors, the
    GEMMs' results rounded to bf16) and a kernel that keeps g and z in bf16 can differ only by roundings that fall the
    other way -- there is no trajectory yet for a borderline decision to steer.  Stored: z blocks (as float32; they are
    bf16 values) and code statistics after 1, 2, 3 fixed-step iterations, and after 1 and 2 line-search iterations with
    the trials / accepted steps of those iterations."""
    X, W = recipe_xw(16384, 256, 1024)
    Xb, Wb = X.bfloat16(), W.bfloat16()
    z0 = Xb.new_zeros(16384, 1024)
    out = {}
    for M in (1, 2, 3):
        z = ref_ista(Xb, z0, Wb, 0.5, fast=True, lr=1.0 / LAMBDA_MAX_C2, maxiter=M, tol=0.0)
        assert z.dtype == torch.bfloat16
        s = zstats(z.float())
        out["fixed_M%d_block" % M] = z[:64, :64].float().numpy().copy()
        out["fixed_M%d_strided" % M] = z[::256, ::16].float().numpy().copy()
        out["fixed_M%d_stats" % M] = np.array([s["sum"], s["abssum"], s["nnz"]])
        out["fixed_M%d_obj" % M] = ref_loss(Xb.float(), z.float(), Wb.float(), 0.5).item()
        print("G3b fixed M=%d" % M, s, out["fixed_M%d_obj" % M])
    orig = ref_ista_mod.backtracking
    for M in (1, 2):
        log = []

        def wrapped(z, x_, weight, alpha, lr0, eta=1.5, maxiter=1000, verbose=False):
            z_next, lr = orig(z, x_, weight, alpha, lr0, eta, maxiter, verbose)
            log.append(lr)
            return z_next, lr
        ref_ista_mod.backtracking = wrapped
        try:
            z = ref_ista(Xb, z0, Wb, 0.5, fast=True, lr=1.0, maxiter=M, tol=0.0, backtrack=True)
        finally:
            ref_ista_mod.backtracking = orig
        s = zstats(z.float())
        out["bt_M%d_block" % M] = z[:64, :64].float().numpy().copy()
        out["bt_M%d_strided" % M] = z[::256, ::16].float().numpy().copy()
        out["bt_M%d_stats" % M] = np.array([s["sum"], s["abssum"], s["nnz"]])
        out["bt_M%d_lr" % M] = np.array(log, dtype=np.float64)
        out["bt_M%d_obj" % M] = ref_loss(Xb.float(), z.float(), Wb.float(), 0.5).item()
        print("G3b bt M=%d" % M, log, s, out["bt_M%d_obj" % M])
    save("g3b_c3_bf16_steps", **out)


def g3trace():
    """Line-search trace of the C3 recipe: trials per outer iteration and the
    accepted step, recorded by wrapping the reference's own `backtracking` at run time (ista()
    discards the returned step, ista.py:87), plus the per-iteration objective ista() prints
    with verbose=True (ista.py:80-81)."""
    import math
    X, W = recipe_xw(16384, 256, 1024)
    out = {}
    orig = ref_ista_mod.backtracking

    def run(tag, x, w, fast, iters):
        log = []

        def wrapped(z, x_, weight, alpha, lr0, eta=1.5, maxiter=1000, verbose=False):
            z_next, lr = orig(z, x_, weight, alpha, lr0, eta, maxiter, verbose)
            log.append((lr, int(round(math.log(lr0 / lr) / math.log(eta))) + 1))
            return z_next, lr
        ref_ista_mod.backtracking = wrapped
        try:
            ref_ista(x, x.new_zeros(x.shape[0], w.shape[1]), w, 0.5, fast=fast, lr=1.0, maxiter=iters,
                     tol=0.0, backtrack=True)
        finally:
            ref_ista_mod.backtracking = orig
        out[tag + "_lr"] = np.array([a for a, _ in log], dtype=np.float64)
        out[tag + "_trials"] = np.array([b for _, b in log], dtype=np.int32)
        print(tag, out[tag + "_trials"].tolist(), out[tag + "_lr"].tolist())

    run("fp32_fista", X, W, True, 10)
    run("fp32_ista", X, W, False, 5)
    run("bf16_fista", X.bfloat16(), W.bfloat16(), True, 10)
    save("g3_c3_trace", **out)


def g4():
    """C4: EM loop n=65536 d=256 k=1024."""
    X, _ = recipe_xw(65536, 256, 1024)
    D0 = recipe_c4_init()
    lr = 1.0 / LAMBDA_MAX_C4
    out = {"check_D0": D0[0, :3].numpy(), "check_X": X[0, :3].numpy()}

    # run the reference EM loop from a given D0: monkeypatch the init only
    def run(constrained, steps, **kw):
        orig = torch.nn.init.orthogonal_
        torch.nn.init.orthogonal_ = lambda w: w.copy_(D0)
        try:
            t = time.time()
            D, losses = dict_learning(X, 1024, alpha=0.5, constrained=constrained,
                                      steps=steps, progbar=False, **kw)
            print("G4 constrained=%s %d steps %.1fs" % (constrained, steps, time.time() - t),
                  losses.tolist())
        finally:
            torch.nn.init.orthogonal_ = orig
        return D, losses

    D, losses = run(True, 3, algorithm="ista")
    out["c_losses_auto"] = losses.numpy()
    out["c_D_cols_auto"] = D[:, :32].numpy().copy()
    D, losses = run(False, 3, algorithm="ista")
    out["r_losses_auto"] = losses.numpy()
    out["r_D_cols_auto"] = D[:, :32].numpy().copy()
    save("g4_c4_em", **out)


def g5():
    """C5 stand-in: synthetic centred 8x8 'patches' (SURVEY 8d G5)."""
    torch.manual_seed(0)
    X = recipe_c5(8192, reseed=False)
    st = torch.get_rng_state()
    D, losses = dict_learning(X, 256, alpha=0.1, constrained=True, steps=10,
                              progbar=False, algorithm="ista")
    # capture the init the reference drew
    torch.set_rng_state(st)
    D0 = torch.empty(64, 256)
    torch.nn.init.orthogonal_(D0)
    D0 = torch.nn.functional.normalize(D0, dim=0)
    print("G5 losses", losses.tolist())
    save("g5_c5_patches", losses=losses.numpy(), D=D.numpy(), D0=D0.numpy(),
         check_X=X[0, :3].numpy())


def small():
    """Small full-output cases (all shapes ragged on purpose)."""
    out = {}
    g = torch.Generator().manual_seed(1234)
    for tag, (n, d, k) in {"a": (37, 10, 50), "b": (64, 256, 1024),
                           "c": (100, 48, 200), "d": (16, 64, 256)}.items():
        W = torch.nn.functional.normalize(torch.randn(d, k, generator=g), dim=0)
        X = torch.randn(n, d, generator=g)
        L = torch.linalg.eigvalsh((W.double() @ W.double().T))[-1].item()
        lr = 1.0 / L
        z0 = X.new_zeros(n, k)
        zf = ref_ista(X, z0, W, 0.3, fast=True, lr=lr, maxiter=25, tol=0.0)
        zi = ref_ista(X, z0, W, 0.3, fast=False, lr=lr, maxiter=25, tol=0.0)
        zb = ref_ista(X, z0, W, 0.3, fast=True, lr=1.0, maxiter=8, tol=0.0,
                      backtrack=True)
        zt = ref_ista(X, z0, W, 0.3, fast=True, lr=lr, maxiter=500, tol=1e-4)
        out.update({tag + "_X": X.numpy(), tag + "_W": W.numpy(),
                    tag + "_lr": lr, tag + "_z_fista": zf.numpy(),
                    tag + "_z_ista": zi.numpy(), tag + "_z_bt": zb.numpy(),
                    tag + "_z_tol": zt.numpy()})
        # M-steps on the FISTA code
        D = W.clone()
        Zc = zf.clone()
        torch.manual_seed(7)
        Dn = ref_dl_mod.update_dict(D, X, Zc)
        out[tag + "_D_bcd"] = Dn.numpy().copy()
        out[tag + "_Z_after_bcd"] = Zc.numpy().copy()
        out[tag + "_D_ridge"] = ref_dl_mod.update_dict_ridge(X, zf, lambd=1e-2).numpy().copy()
        out[tag + "_loss"] = ref_loss(X, zf, W, 0.3).item()
    save("small_cases", **out)


def inits():
    """sparse_encode(init=...) for every init mode of sparse_encode.py:19-35."""
    from lasso.linear import initialize_code
    out = {}
    g = torch.Generator().manual_seed(77)
    for tag, (n, d, k) in {"under": (40, 12, 30), "over": (40, 30, 12)}.items():
        W = torch.nn.functional.normalize(torch.randn(d, k, generator=g), dim=0)
        X = torch.randn(n, d, generator=g)
        out[tag + "_X"], out[tag + "_W"] = X.numpy(), W.numpy()
        for mode in ("zero", "transpose", "lstsq", "ridge"):
            out["%s_z0_%s" % (tag, mode)] = initialize_code(X, W, 0.3, mode).numpy().copy()
            out["%s_z_%s" % (tag, mode)] = sparse_encode(X, W, alpha=0.3, algorithm="ista", init=mode,
                                                         lr=0.05, maxiter=20, tol=0.0).numpy().copy()
    save("init_modes", **out)


def cd():
    """Greedy coordinate descent (coordinate_descent.py:5-54):
    full outputs on small ragged shapes, statistics + a corner at the C2 shape."""
    from lasso.linear.solvers import coord_descent
    out = {}
    g = torch.Generator().manual_seed(4321)
    for tag, (n, d, k, alpha) in {"a": (37, 10, 50, 0.2), "b": (48, 256, 1024, 0.5),
                                  "c": (100, 48, 200, 0.3), "d": (16, 64, 256, 0.1)}.items():
        W = torch.nn.functional.normalize(torch.randn(d, k, generator=g), dim=0)
        X = torch.randn(n, d, generator=g)
        out[tag + "_X"], out[tag + "_W"], out[tag + "_alpha"] = X.numpy(), W.numpy(), alpha
        for mi in (1, 7, 60, 1000):
            out["%s_z_%d" % (tag, mi)] = coord_descent(X, W, None, alpha, maxiter=mi).numpy().copy()
        # warm start: z0 is updated in place by the reference (:14,47)
        z0 = torch.nn.functional.softshrink(X @ W, alpha)
        out[tag + "_z0"] = z0.numpy().copy()
        out[tag + "_z_warm"] = coord_descent(X, W, z0, alpha, maxiter=40, tol=1e-4).numpy().copy()
        out[tag + "_z0_after"] = z0.numpy().copy()
        out[tag + "_z_sparse_encode"] = sparse_encode(X, W, alpha=alpha, algorithm="cd",
                                                      maxiter=25).numpy().copy()
    # C2 shape (n reduced to 512 rows of the recipe: rows are independent)
    X, W = recipe_xw(512, 256, 1024)
    for mi in (100, 1000):
        t = time.time()
        z = coord_descent(X, W, None, 0.5, maxiter=mi)
        print("cd C2[:512] maxiter", mi, "%.1fs" % (time.time() - t))
        obj = (0.5 * (z @ W.T - X).pow(2).sum(1) + 0.5 * z.abs().sum(1))
        out["c2_obj_rows_%d" % mi] = obj.numpy().copy()
        out["c2_corner_%d" % mi] = z[:64, :64].numpy().copy()
        st = zstats(z)
        out["c2_stats_%d" % mi] = np.array([st["sum"], st["abssum"], st["nnz"]])
    save("cd_cases", **out)


def conv():
    """Convolutional ISTA/FISTA (lasso/conv2d/ista.py:7-49) and the Toeplitz Lipschitz
    bound (lasso/conv2d/lip_const.py:96-135), SURVEY 8f row f3."""
    from lasso.conv2d.ista import ista_conv2d
    from lasso.conv2d.lip_const import lip_bound_conv2d
    out = {}
    g = torch.Generator().manual_seed(2468)
    # tag: (N, C, K, ksize, stride, padding, Hz, Wz, alpha)
    cases = {"a": (4, 1, 8, 5, 1, 0, 12, 12, 0.1), "b": (3, 3, 6, 3, 2, 1, 7, 9, 0.05),
             "c": (2, 2, 5, 7, 1, 3, 9, 11, 0.1), "d": (5, 1, 16, 5, 1, 2, 28, 28, 0.2),
             "e": (2, 4, 40, 3, 1, 1, 16, 16, 0.02)}
    for tag, (N, C, K, ks, st, pd, Hz, Wz, alpha) in cases.items():
        W = torch.randn(K, C, ks, ks, generator=g) * (1.0 / ks)
        H, Wd = (Hz - 1) * st - 2 * pd + ks, (Wz - 1) * st - 2 * pd + ks
        x = torch.randn(N, C, H, Wd, generator=g)
        z0 = torch.zeros(N, K, Hz, Wz)
        out[tag + "_x"], out[tag + "_w"] = x.numpy(), W.numpy()
        out[tag + "_cfg"] = np.array([N, C, K, ks, st, pd, Hz, Wz], dtype=np.int64)
        out[tag + "_alpha"] = alpha
        lr = 0.5 / (W.pow(2).sum().item())          # a safe explicit step
        out[tag + "_lr"] = lr
        for fast in (True, False):
            for mi in (1, 12):
                z = ista_conv2d(x, z0, W, alpha, stride=st, padding=pd, fast=fast, maxiter=mi, lr=lr, tol=0.0)
                out["%s_z_%s_%d" % (tag, "fista" if fast else "ista", mi)] = z.numpy().copy()
        zw = torch.randn(N, K, Hz, Wz, generator=g) * 0.1
        out[tag + "_z0_warm"] = zw.numpy().copy()
        out[tag + "_z_warm"] = ista_conv2d(x, zw, W, alpha, stride=st, padding=pd, maxiter=6, lr=lr,
                                           tol=0.0).numpy().copy()
        if st == 1:
            out[tag + "_lip"] = lip_bound_conv2d(W, pd).item()
            out[tag + "_lip_sqrt"] = lip_bound_conv2d(W, pd, sqrt=True).item()
            z = ista_conv2d(x, z0, W, alpha, stride=st, padding=pd, maxiter=200, tol=1e-4)   # lr='auto', stop rule
            out[tag + "_z_auto_tol"] = z.numpy().copy()
    save("conv_cases", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g5", "small", "g4"]
    torch.set_num_threads(8)
    for w in which:
        t = time.time()
        globals()[w]()
        print(w, "done in %.1fs" % (time.time() - t))
