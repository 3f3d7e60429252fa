"""CPU oracle for the ISTA/FISTA sparse-encode + dictionary-learning hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pytorch-lasso_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / the timed CPU baseline.

This is a *restatement* (not a copy) of the algorithm of rfeinman/pytorch-lasso
for the path named by BASELINE.json, written from SURVEY.md section 8a.  Every
function cites the reference file:line whose arithmetic it follows.  The
arithmetic itself lives in third-party code (PyTorch ATen CPU kernels and
SciPy/ARPACK; the reference pins no versions -- this container has torch
2.10.0, scipy 1.15.3), so this restatement issues the *same ATen ops in the same
order*: with an explicit float ``lr`` it is bitwise identical to the reference
on CPU, which is how it is pinned (tests/golden/generate_golden.py ran the real
reference in the build container and stored its outputs; tests/test_oracle.py
checks this module against those files).  Parity status: PINNED against outputs
of the reference itself (the reference ships no tests or golden vectors).

Shapes: x [n,d], weight [d,k] (atoms are columns),
z [n,k]; all row-major.
"""
import math
import warnings

import torch

__all__ = [
    "soft_threshold", "lipschitz_constant", "momentum_schedule",
    "backtracking_step", "fista", "initial_code", "sparse_encode",
    "lasso_objective", "dict_evaluate", "update_dict", "update_dict_gram",
    "update_dict_ridge", "dict_learning", "FistaTrace", "coordinate_descent",
    "conv_lipschitz_bound", "conv_fista", "conv_objective",
]


# --------------------------------------------------------------------------
# elementwise pieces
# --------------------------------------------------------------------------
def soft_threshold(v, lam):
    """S_lam(v): the proximal map of lam*|.|_1.

    The reference calls ``F.softshrink`` (ista.py:40,52,90), whose ATen CPU
    kernel is ``v-lam if v>lam; v+lam if v<-lam; else 0``.  We call the same
    ATen op so that rounding is identical.
    """
    return torch.nn.functional.softshrink(v, lam)


def momentum_schedule(n_iter):
    """Nesterov coefficients (t_i - 1)/t_{i+1}, t_0 = 1, as python floats.

    Follows ista.py:78 (t starts at the int 1) and ista.py:98-101
    (t_next = (1+sqrt(1+4t^2))/2).  The schedule does not depend on the data,
    which is what lets the HIP engine precompute it on the host.
    """
    coeffs, t = [], 1
    for _ in range(n_iter):
        t_next = (1 + math.sqrt(1 + 4 * t ** 2)) / 2
        coeffs.append((t - 1) / t_next)
        t = t_next
    return coeffs


# --------------------------------------------------------------------------
# Lipschitz constant  (ista.py:8-14)
# --------------------------------------------------------------------------
def lipschitz_constant(weight, method="arpack"):
    """lambda_max(W^T W), the Lipschitz constant of the RSS gradient.

    method='arpack' follows ista.py:8-14 exactly: fp32 Gram [k,k] on the
    tensor's device, copied to host, ARPACK ``eigsh(k=1, which='LM')``.  That
    path is not run-to-run reproducible (~1e-6 relative jitter, SURVEY.md
    section 0).  method='exact' is the deterministic fp64 value (eigvalsh of
    the smaller of W W^T / W^T W) used as ground truth for the device-side
    estimator.
    """
    if method == "arpack":
        from scipy.sparse.linalg import eigsh
        gram = torch.matmul(weight.t(), weight)
        return eigsh(gram.detach().cpu().numpy(), k=1, which="LM",
                     return_eigenvectors=False).item()
    if method == "exact":
        w = weight.detach().to("cpu", torch.float64)
        gram = w @ w.t() if w.shape[0] <= w.shape[1] else w.t() @ w
        return torch.linalg.eigvalsh(gram)[-1].item()
    raise ValueError("unknown method %r" % (method,))


# --------------------------------------------------------------------------
# backtracking line search  (ista.py:17-54)
# --------------------------------------------------------------------------
def backtracking_step(p, x, weight, alpha, lr0, eta=1.5, max_trials=1000,
                      trace=None):
    """One Beck-Teboulle backtracking proximal step from the point ``p``.

    Follows ista.py:17-54.  All sums run over the WHOLE batch, i.e. one step
    size is chosen for all samples (ista.py:23,28,33-35).  Returns
    ``(z_next, lr_accepted)``.  On failure after ``max_trials`` trials it
    warns and reverts to lr0 (ista.py:48-52).
    """
    if eta <= 1:
        raise ValueError("eta must be > 1.")                      # ista.py:18-19

    resid0 = torch.matmul(p, weight.T) - x                        # ista.py:22
    f0 = 0.5 * resid0.pow(2).sum()                                # ista.py:23
    g0 = torch.matmul(resid0, weight)                             # ista.py:24

    lr, trials, accepted = lr0, 0, False
    z_next = None
    while trials < max_trials:
        z_next = soft_threshold(p - lr * g0, alpha * lr)          # ista.py:40
        resid1 = torch.matmul(z_next, weight.T) - x               # ista.py:27
        l1 = z_next.abs().sum()
        big_f = 0.5 * resid1.pow(2).sum() + alpha * l1            # ista.py:28
        dz = z_next - p                                           # ista.py:31
        big_q = (f0 + (dz * g0).sum()                             # ista.py:32-35
                 + (0.5 / lr) * dz.pow(2).sum()
                 + alpha * z_next.abs().sum())
        trials += 1
        if big_f <= big_q:                                        # ista.py:45
            accepted = True
            break
        lr = lr / eta                                             # ista.py:47
    if not accepted:
        warnings.warn("backtracking line search failed. Reverting to initial "
                      "step size")                                # ista.py:49-50
        lr = lr0
        z_next = soft_threshold(p - lr * g0, alpha * lr)          # ista.py:51-52
    if trace is not None:
        trace.trials.append(trials)
        trace.accepted_lr.append(lr)
    return z_next, lr


class FistaTrace:
    """Optional per-iteration record (not in the reference; test aid)."""

    def __init__(self):
        self.objective = []      # mean objective of z BEFORE each iteration
        self.delta = []          # sum|z - z_next| of each iteration
        self.trials = []         # line-search trials per iteration
        self.accepted_lr = []
        self.iterations = 0
        self.stopped = False


# --------------------------------------------------------------------------
# ISTA / FISTA  (ista.py:57-104)
# --------------------------------------------------------------------------
def fista(x, z0, weight, alpha=1.0, fast=True, lr="auto", maxiter=10,
          tol=1e-5, backtrack=False, eta_backtrack=1.5, trace=None):
    """min_z 0.5*||z W^T - x||^2 + alpha*||z||_1 by (accelerated) proximal
    gradient.  Follows ista.py:57-104 (same defaults, same op order).

    * lr='auto' -> 1/lipschitz_constant(weight)               (ista.py:59-63)
    * absolute stop budget = z0.numel()*tol                    (ista.py:64)
    * the stop test compares z (not y) with z_next and runs BEFORE the
      momentum update; on stop the returned code is z_next     (ista.py:93-95)
    * with backtracking the accepted step size is discarded, every outer
      iteration restarts from lr                               (ista.py:87)
    * maxiter=0 returns z0 itself                              (ista.py:76,104)
    """
    if lr == "auto":
        lr = 1 / lipschitz_constant(weight)
    budget = z0.numel() * tol

    def mean_objective(zk):                                       # ista.py:66-69
        resid = torch.matmul(zk, weight.T) - x
        return (0.5 * resid.pow(2).sum() + alpha * zk.abs().sum()) / x.size(0)

    z = z0
    y, t = z0, 1                                                  # ista.py:76-78
    for _ in range(maxiter):
        if trace is not None:
            trace.objective.append(float(mean_objective(z)))
        p = y if fast else z                                      # ista.py:84
        if backtrack:
            z_next, _ = backtracking_step(p, x, weight, alpha, lr,
                                          eta_backtrack, trace=trace)
        else:
            resid = torch.matmul(p, weight.T) - x                 # ista.py:72
            grad = torch.matmul(resid, weight)                    # ista.py:73
            z_next = soft_threshold(p - lr * grad, alpha * lr)    # ista.py:90
        delta = (z - z_next).abs().sum()                          # ista.py:93
        if trace is not None:
            trace.delta.append(float(delta))
            trace.iterations += 1
        if delta <= budget:
            z = z_next
            if trace is not None:
                trace.stopped = True
            break
        if fast:                                                  # ista.py:98-101
            t_next = (1 + math.sqrt(1 + 4 * t ** 2)) / 2
            y = z_next + ((t - 1) / t_next) * (z_next - z)
            t = t_next
        z = z_next                                                # ista.py:102
    return z


# --------------------------------------------------------------------------
# sparse_encode boundary  (sparse_encode.py:8-51, 62-63)
# --------------------------------------------------------------------------
_DEFAULT_INIT = {"ista": "zero"}                                  # sparse_encode.py:8-16


def initial_code(x, weight, alpha, mode):
    """z0 initialisation, sparse_encode.py:19-35: 'zero' (:22-23, the hot-path default),
    'unif' (:24-25), 'lstsq' (:26-27), 'ridge' (:28-29), 'transpose' (:30-31)
    (SURVEY.md section 8f row f1)."""
    n, k = x.size(0), weight.size(1)
    if mode == "zero":
        return x.new_zeros(n, k)
    if mode == "unif":
        return x.new(n, k).uniform_(-0.1, 0.1)
    if mode == "transpose":
        return torch.matmul(x, weight)
    if mode == "lstsq":                                           # :26-27 -> utils.py:13-25
        return least_squares_code(x, weight)
    if mode == "ridge":                                           # :28-29 -> utils.py:28-40
        return ridge_code(x, weight, alpha)
    raise ValueError("invalid init parameter '{}'.".format(mode))  # :33


def least_squares_code(x, weight):
    """z0 with W z0_i = x_i in the least-squares / least-norm sense via a reduced QR
    (lasso/linear/utils.py:13-25, called as lstsq(x.T, weight).T)."""
    d, k = weight.shape
    b = x.T                                                        # [d, n]
    if d < k:                                                      # under-determined: least norm
        Q, R = torch.linalg.qr(weight.T, mode="reduced")           # W^T = Q R,  Q [k,d], R [d,d]
        y = torch.linalg.solve_triangular(R.T, b, upper=False)     # R^T y = b
        sol = Q @ y
    else:                                                          # over-determined: least squares
        Q, R = torch.linalg.qr(weight, mode="reduced")
        sol = torch.linalg.solve_triangular(R, Q.T @ b, upper=True)
    return sol.T


def ridge_code(x, weight, alpha):
    """z0 = argmin ||W z - x||^2 + alpha ||z||^2 per sample via Cholesky of W^T W + alpha I
    (lasso/linear/utils.py:28-40, called as ridge(x.T, weight, alpha=alpha).T)."""
    gram = weight.T @ weight
    gram.diagonal().add_(alpha)
    chol, info = torch.linalg.cholesky_ex(gram)
    if info != 0:
        raise RuntimeError("The Gram matrix is not positive definite. Try increasing 'alpha'.")
    return torch.cholesky_solve(weight.T @ x.T, chol).T


def sparse_encode(x, weight, alpha=1.0, z0=None, algorithm="ista", init=None,
                  **kwargs):
    """sparse_encode.py:38-73 restricted to the 'ista' (:62-63) and 'cd' (:54-55) arms."""
    n, k = x.size(0), weight.size(1)
    if z0 is not None:
        assert z0.shape == (n, k)                                 # :44-45
    else:
        if init is None:
            init = _DEFAULT_INIT.get(algorithm, "zero")           # :47-48
        z0 = initial_code(x, weight, alpha, init)                 # :51
    if algorithm == "ista":
        return fista(x, z0, weight, alpha, **kwargs)              # :62-63
    if algorithm == "cd":
        return coordinate_descent(x, weight, z0, alpha, **kwargs)  # :54-55
    if algorithm in ("gpsr", "iter-ridge", "interior-point",
                     "split-bregman", "own"):
        raise NotImplementedError("algorithm=%r is outside the hot path"
                                  % algorithm)
    raise ValueError("invalid algorithm parameter '{}'.".format(algorithm))  # :71


# --------------------------------------------------------------------------
# greedy coordinate descent  (coordinate_descent.py:5-54)
# --------------------------------------------------------------------------
def coordinate_descent(x, weight, z0=None, alpha=1.0, maxiter=1000, tol=1e-6,
                       return_info=False):
    """Greedy (largest-update-first) lasso coordinate descent.

    coordinate_descent.py:5-54.  State per sample row: the correlation vector
    b (starts at x W regardless of z0, :19) and the tracked code z (starts at
    z0 or 0, :10-14).  One step per row (:31-39): propose S_alpha(b), pick the
    coordinate j whose proposal moved furthest from z (first index on ties --
    torch.argmax), commit z_j, and correct b by column j of S = I - W^T W
    (:22-23) times the committed change.  A row leaves the active set for good
    once its committed change is <= tol*k (:9,45,47); rows never interact, so
    the batch loop of the reference is n independent per-row loops.  Returns
    S_alpha(b) (:52), NOT the tracked z.  Like the reference (:14,47) a
    caller-supplied z0 is updated in place and ends up holding the tracked z.
    """
    d, k = weight.shape
    n = x.shape[0]
    assert x.shape[1] == d                                        # :8
    thresh = tol * k                                              # :9
    if z0 is None:
        z = x.new_zeros(n, k)                                     # :11
    else:
        assert z0.shape == (n, k)                                 # :13
        z = z0                                                    # :14 (aliases)
    b = torch.mm(x, weight)                                       # :19
    S = -torch.mm(weight.T, weight)                               # :22
    S.diagonal().add_(1.0)                                        # :23
    rows = torch.arange(n, device=weight.device)                  # :41
    row_steps = torch.zeros(n, dtype=torch.int64)
    for _ in range(maxiter):                                      # :42
        if rows.numel() == 0:                                     # :43
            break
        z_act, b_act = z[rows], b[rows]
        prop = soft_threshold(b_act, alpha)                       # :32
        move = prop - z_act                                       # :33
        j = move.abs().argmax(1)                                  # :34
        jj = j.unsqueeze(1)
        b[rows] = b_act + S[:, j].T * move.gather(1, jj)          # :36
        z_new = z_act.scatter(1, jj, prop.gather(1, jj))          # :37
        change = (z_new - z_act).abs().sum(1)                     # :46
        z[rows] = z_new                                           # :47
        row_steps[rows] += 1
        rows = rows[change > thresh]                              # :48
    out = soft_threshold(b, alpha)                                # :52
    if return_info:
        return out, dict(row_steps=row_steps, n_active=int(rows.numel()), z_track=z)
    return out



# --------------------------------------------------------------------------
# convolutional ISTA/FISTA  (lasso/conv2d/ista.py:7-49,
# lasso/conv2d/lip_const.py:96-135)
# --------------------------------------------------------------------------
def conv_lipschitz_bound(kernel, padding, stride=1, sample=50, sqrt=False):
    """Araujo et al. Toeplitz bound on lambda_max(W^T W) of a stride-1 conv2d,
    lip_const.py:96-135: evaluate the kernel's 2-D Fourier symbol on a
    sample x sample frequency grid, sum the squared magnitudes over the
    larger channel dimension, take the maximum over frequencies and sum over
    the smaller channel dimension."""
    assert kernel.dim() == 4                                       # :98
    if kernel.size(-1) != kernel.size(-2):
        raise ValueError("The last 2 dim of the kernel must be equal.")      # :99-100
    if kernel.size(-1) % 2 != 1:
        raise ValueError("The dimension of the kernel must be odd.")         # :101-102
    if stride != 1:
        raise NotImplementedError("LipBound not implemented for stride > 1.")  # :103-104
    ks = kernel.size(-1)
    if kernel.size(0) > kernel.size(1):
        kernel = kernel.transpose(0, 1)                            # :106-107
    freq = torch.linspace(0, 2 * math.pi, sample)                  # :110
    f0, f1 = torch.meshgrid(freq, freq, indexing="ij")             # :111
    pos = 1.0 + torch.arange(padding - ks, padding)                # :116
    h0, h1 = torch.meshgrid(pos, pos, indexing="ij")               # :117
    phase = (f0.reshape(-1, 1) * h0.reshape(1, -1) + f1.reshape(-1, 1) * h1.reshape(1, -1)).T  # :120
    taps = kernel.flatten(2)                                       # :125
    re = torch.matmul(taps, torch.cos(phase))                      # :126
    im = torch.matmul(taps, torch.sin(phase))                      # :127
    power = re.square().sum(1) + im.square().sum(1)                # :128-130
    bound = power.max(-1)[0].sum()                                 # :131
    return bound.sqrt() if sqrt else bound                         # :132-135


def conv_objective(x, z, weight, alpha, stride=1, padding=0):
    """ista.py:23-26: (0.5*||x - conv_transpose2d(z)||^2 + alpha*||z||_1) / batch."""
    F = torch.nn.functional
    x_hat = F.conv_transpose2d(z, weight, stride=stride, padding=padding)
    return (0.5 * (x - x_hat).pow(2).sum() + alpha * z.abs().sum()) / x.size(0)


def conv_fista(x, z0, weight, alpha=1.0, stride=1, padding=0, fast=True,
               maxiter=10, lr="auto", tol=1e-5, return_info=False):
    """ista_conv2d, lasso/conv2d/ista.py:7-49: the proximal-gradient loop of
    fista() with x_hat = conv_transpose2d(z, W) as the synthesis operator and
    conv2d(. , W) as its adjoint (:18-20).  The momentum update comes before
    the stop test here (:39-46) -- same iterates, same result."""
    F = torch.nn.functional
    if lr == "auto":
        if stride != 1:
            raise NotImplementedError("auto lr is only implemented for stride == 1.")  # :10-12
        lr = 1 / conv_lipschitz_bound(weight, padding)             # :14-15 (a 0-d tensor)
    budget = z0.numel() * tol                                      # :16

    def step(p):                                                   # :18-20, :28-29
        resid = F.conv_transpose2d(p, weight, stride=stride, padding=padding) - x
        return soft_threshold(p - lr * F.conv2d(resid, weight, stride=stride, padding=padding),
                              alpha * lr)

    z, y, t, done, last = z0, z0, 1, 0, float("nan")
    for _ in range(maxiter):                                       # :36
        z_next = step(y) if fast else step(z)                      # :39
        if fast:
            t_next = (1 + math.sqrt(1 + 4 * t ** 2)) / 2           # :41
            y = z_next + ((t - 1) / t_next) * (z_next - z)         # :42
            t = t_next
        done += 1
        last = (z - z_next).abs().sum().item()                     # :44
        z = z_next
        if last <= budget:                                         # :44-46
            break
    if return_info:
        return z, dict(iterations=done, last_delta=last)
    return z


# --------------------------------------------------------------------------
# objective + M-steps + EM driver  (dict_learning.py)
# --------------------------------------------------------------------------
def lasso_objective(X, Z, weight, alpha=1.0):
    """(0.5*||X - Z W^T||^2 + alpha*||Z||_1)/n, dict_learning.py:10-13."""
    recon = torch.matmul(Z, weight.T)
    return (0.5 * (X - recon).pow(2).sum() + alpha * Z.abs().sum()) / X.size(0)


def dict_evaluate(X, weight, alpha, **kwargs):
    """dict_learning.py:16-20."""
    X = X.to(weight.device)
    Z = sparse_encode(X, weight, alpha, **kwargs)
    return lasso_objective(X, Z, weight, alpha)


def update_dict(dictionary, X, Z, random_seed=None, positive=False, eps=1e-10):
    """Constrained M-step: Gauss-Seidel sweep over atoms with unit-norm
    projection, dict_learning.py:56-103.  Mutates ``dictionary`` AND ``Z`` in
    place (:86,98) and returns ``dictionary`` (:103)."""
    k = dictionary.size(1)
    if random_seed is not None:
        torch.manual_seed(random_seed)                            # :78-79
    resid = X - torch.matmul(Z, dictionary.T)                     # :82
    for j in range(k):
        code_j = Z[:, j]
        resid += torch.outer(code_j, dictionary[:, j])            # :85
        dictionary[:, j] = torch.matmul(code_j, resid)            # :86
        if positive:
            dictionary[:, j].clamp_(0, None)                      # :87-88
        nrm = dictionary[:, j].norm()                             # :91
        if nrm < eps:                                             # :92
            dictionary[:, j].normal_()                            # :93
            if positive:
                dictionary[:, j].clamp_(0, None)
            dictionary[:, j] /= dictionary[:, j].norm()           # :96
            Z[:, j].zero_()                                       # :98
        else:
            dictionary[:, j] /= nrm                               # :100
            resid -= torch.outer(code_j, dictionary[:, j])        # :101
    return dictionary


def update_dict_gram(dictionary, gram_zz, gram_zx, positive=False, eps=1e-10,
                     fresh_atom=None):
    """The same sweep in Gram form: with
    A = Z^T Z [k,k] and B = Z^T X [k,d],
        u_j = B_j - A_j . D^T + A_jj d_j      (using the CURRENT D),
    which equals ``Z[:,j] @ R`` of dict_learning.py:85-86 because
    R = X - Z D^T at every point of the sweep.  This is the form the HIP
    M-step implements (it needs one all-reduce of A,B instead of passes over
    the row-sharded residual).  Returns (dictionary, degenerate_mask); the
    caller zeroes Z[:, j] for degenerate atoms (dict_learning.py:98).
    ``fresh_atom(j) -> [d]`` supplies the replacement direction for a
    degenerate atom (the reference draws it from torch's global RNG, :93).
    A/B are modified in place the way zeroing Z[:,j] would modify them."""
    k = dictionary.size(1)
    degenerate = torch.zeros(k, dtype=torch.bool)
    for j in range(k):
        d_j = dictionary[:, j]
        u = gram_zx[j] - torch.mv(dictionary, gram_zz[j]) + gram_zz[j, j] * d_j
        if positive:
            u = u.clamp(0, None)
        nrm = u.norm()
        if nrm < eps:
            u = fresh_atom(j) if fresh_atom is not None else torch.randn_like(u)
            if positive:
                u = u.clamp(0, None)
            dictionary[:, j] = u / u.norm()
            gram_zz[j, :] = 0
            gram_zz[:, j] = 0
            gram_zx[j, :] = 0
            degenerate[j] = True
        else:
            dictionary[:, j] = u / nrm
    return dictionary, degenerate


def update_dict_ridge(x, z, lambd=1e-4):
    """Unconstrained M-step V = ((Z^T Z + lambd*n*I)^-1 Z^T X)^T via Cholesky,
    dict_learning.py:106-123."""
    rhs = torch.mm(z.T, x)                                        # :117
    gram = torch.mm(z.T, z)                                       # :118
    gram.diagonal().add_(lambd * x.size(0))                       # :119
    chol = torch.linalg.cholesky(gram)                            # :120
    return torch.cholesky_solve(rhs, chol).T                      # :121


def dict_learning(X, n_components, alpha=1.0, constrained=True, persist=False,
                  lambd=1e-2, steps=60, device="cpu", progbar=False,
                  init_weight=None, **solver_kwargs):
    """EM dictionary learning, dict_learning.py:23-53.

    Same order of operations: orthogonal init (+ column normalisation when
    constrained) :28-31; per step E-step :38, objective BEFORE the M-step :39,
    optional warm start that aliases Z :40-41, then the M-step :44-47.
    ``init_weight`` (extension, for parity runs) replaces the RNG-dependent
    initial dictionary.  The tqdm bar of :35,50-51 is omitted."""
    n, d = X.shape
    X = X.to(device)
    if init_weight is None:
        weight = torch.empty(d, n_components, device=device)      # :28
        torch.nn.init.orthogonal_(weight)                         # :29
        if constrained:
            weight = torch.nn.functional.normalize(weight, dim=0)  # :30-31
    else:
        weight = init_weight.clone().to(device)
    Z0 = None
    losses = torch.zeros(steps, device=device)                    # :34
    for i in range(steps):
        Z = sparse_encode(X, weight, alpha, Z0, **solver_kwargs)  # :38
        losses[i] = lasso_objective(X, Z, weight, alpha)          # :39
        if persist:
            Z0 = Z                                                # :40-41
        if constrained:
            weight = update_dict(weight, X, Z)                    # :44-45
        else:
            weight = update_dict_ridge(X, Z, lambd=lambd)         # :46-47
    return weight, losses
