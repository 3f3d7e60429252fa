"""Mirror of lasso/linear/sparse_encode.py:8-73 for the 'ista' arm."""
import torch

from .solvers import ista, coord_descent

_init_defaults = {'ista': 'zero'}                                   # sparse_encode.py:8-16

_OFF_PATH_ALGOS = ('gpsr', 'iter-ridge', 'interior-point', 'split-bregman', 'own')


def _engine_for(x, weight):
    from ..engine import HipEngine
    from .. import _native as nat
    nat.require_gpu()
    return HipEngine(x.device if x.is_cuda else (weight.device if weight.is_cuda else None))


def _native_ok(x, weight):
    # the library's Gram / Cholesky kernels: fp32, no autograd graph, systems of at most 4096 unknowns
    return (x.dtype == torch.float32 and weight.dtype == torch.float32 and x.size(0) > 0
            and not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad)))


def _lstsq_qr(x, weight):
    # least-norm (d < k) / least-squares (d >= k) code through a reduced QR of the dictionary
    # (utils.py:13-25, the reference's own route): torch.linalg on the tensors' device
    d, k = weight.shape
    rhs = x.T
    if d < k:
        q, r = torch.linalg.qr(weight.T, mode='reduced')
        return (q @ torch.linalg.solve_triangular(r.T, rhs, upper=False)).T
    q, r = torch.linalg.qr(weight, mode='reduced')
    return torch.linalg.solve_triangular(r, q.T @ rhs, upper=True).T


def _lstsq_init(x, weight):
    """init='lstsq' (sparse_encode.py:26-27, utils.py:13-25).  On the library's own kernels through the
    normal equations with ONE step of iterative refinement (the corrected semi-normal equations):
    d >= k: (W^T W) z = W^T x (Gram product + blocked Cholesky, csrc/mstep.hip and ridge.hip), then the
    same solve for the residual x - z W^T; d < k: the least-norm code z = W^T u, (W W^T) u = x, refined the
    same way with the residual taken through W.  The refinement step brings the normal equations'
    cond(W)^2 error back to the level of the reference's QR route as long as cond(W)^2 * 2^-24 < 1
    (measured against fp64: cond 3 -> 8e-7, cond 50 -> 3e-6, cond 560 -> 5e-6 of max|z|, QR: 8e-7 / 1e-6 / 3e-6);
    when the Cholesky factorisation meets a non-positive pivot -- a rank-deficient dictionary -- the
    reference's QR route runs instead (torch.linalg on the device)."""
    d, k = weight.shape
    if not _native_ok(x, weight) or min(d, k) > 4096:
        return _lstsq_qr(x, weight)
    eng = _engine_for(x, weight)
    xg, wg = eng.to_device(x), eng.to_device(weight)
    n = xg.shape[0]
    wt = wg.T.contiguous()                                       # [k,d]
    try:
        if d >= k:
            buf = torch.empty(k * k + k * n, dtype=torch.float32, device=eng.device)
            A, B = eng.gram(wg, xg.T.contiguous(), buf)          # A = W^T W [k,k], B = W^T x^T [k,n]
            z0 = eng.ridge(A, B, 0.0, check=True)                # [n,k]
            r = xg - eng.init_transpose(z0, wt)                  # x - z W^T  [n,d]
            A, B = eng.gram(wg, r.T.contiguous(), buf)
            z0 = z0 + eng.ridge(A, B, 0.0)
        else:
            buf = torch.empty(2 * d * d, dtype=torch.float32, device=eng.device)
            A, _ = eng.gram(wt, wt, buf)                         # A = W W^T [d,d]
            u = eng.ridge(A, xg.T.contiguous(), 0.0, check=True)  # ((W W^T)^-1 x^T)^T  [n,d]
            r = xg - eng.init_transpose(eng.init_transpose(u, wg), wt)      # x - (u W) W^T
            u = u + eng.ridge(A, r.T.contiguous(), 0.0)
            z0 = eng.init_transpose(u, wg)                       # u W  [n,k]
    except torch.linalg.LinAlgError:
        return _lstsq_qr(x, weight)
    return z0.to(x.device)


def _ridge_init(x, weight, alpha):
    """init='ridge' (sparse_encode.py:28-29, utils.py:28-40): (W^T W + alpha I) z = W^T x per sample.
    A = W^T W and B = W^T x^T are ONE launch of the Gram kernel (lasso_gram_accumulate with the
    dictionary in the role of the codes), the Cholesky factorisation and both triangular solves are
    lasso_ridge_solve (csrc/ridge.hip)."""
    d, k = weight.shape
    if not _native_ok(x, weight) or k > 4096:
        gram = weight.T @ weight
        gram.diagonal().add_(alpha)
        chol, info = torch.linalg.cholesky_ex(gram)
        if info != 0:
            raise RuntimeError("The Gram matrix is not positive definite. Try increasing 'alpha'.")
        return torch.cholesky_solve(weight.T @ x.T, chol).T
    eng = _engine_for(x, weight)
    xg, wg = eng.to_device(x), eng.to_device(weight)
    n = xg.shape[0]
    buf = torch.empty(k * k + k * n, dtype=torch.float32, device=eng.device)
    A, B = eng.gram(wg, xg.T.contiguous(), buf)
    try:
        z0 = eng.ridge(A, B, float(alpha), check=True)
    except torch.linalg.LinAlgError:
        raise RuntimeError("The Gram matrix is not positive definite. Try increasing 'alpha'.")   # utils.py:36-38
    return z0.to(x.device)


def initialize_code(x, weight, alpha, mode):
    """sparse_encode.py:19-35.  'zero' (:22-23) is the hot-path default; 'transpose' is a product on
    the library's own GEMM (lasso_init_transpose); 'ridge' and 'lstsq' run on its Gram and Cholesky
    kernels (lasso_gram_accumulate + lasso_ridge_solve); 'unif' is torch's RNG like the reference."""
    n_samples = x.size(0)
    n_components = weight.size(1)
    if mode == 'zero':
        z0 = x.new_zeros(n_samples, n_components)
    elif mode == 'unif':
        z0 = x.new(n_samples, n_components).uniform_(-0.1, 0.1)
    elif mode == 'transpose':                                        # :24-25
        if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
            # part of the caller's autograd graph like the reference's torch.matmul: dL/dz0 reaches x, W
            z0 = torch.matmul(x, weight)
        else:                                                        # on the library's NT GEMM
            eng = _engine_for(x, weight)
            z0 = eng.init_transpose(eng.to_device(x).float().contiguous(),
                                    eng.to_device(weight).float().contiguous()).to(device=x.device, dtype=x.dtype)
    elif mode == 'lstsq':                                            # :26-27 (utils.py:13-25)
        z0 = _lstsq_init(x, weight)
    elif mode == 'ridge':                                            # :28-29 (utils.py:28-40)
        z0 = _ridge_init(x, weight, alpha)
    else:
        raise ValueError("invalid init parameter '{}'.".format(mode))   # :33
    return z0


def sparse_encode(x, weight, alpha=1.0, z0=None, algorithm='ista', init=None,
                  **kwargs):
    """Same call surface as lasso.linear.sparse_encode (sparse_encode.py:38-73);
    ``algorithm='ista'`` and ``'cd'`` run on the HIP engine, the other solver names raise
    NotImplementedError (they are outside the accelerated path), anything else
    raises ValueError like the reference (:71)."""
    n_samples = x.size(0)
    n_components = weight.size(1)
    if z0 is not None:
        assert z0.shape == (n_samples, n_components)                 # :44-45
    else:
        if init is None:
            init = _init_defaults.get(algorithm, 'zero')             # :47-48
        if init == 'zero' and algorithm == 'ista' and kwargs.get('maxiter', 10) != 0 and n_samples * n_components > 1:
            from .solvers.ista import lazy_zeros
            z0 = lazy_zeros(x, n_samples, n_components)              # :22-23 without the n*k fill (and read)
        else:
            z0 = initialize_code(x, weight, alpha, mode=init)        # :51
    if algorithm == 'ista':
        z = ista(x, z0, weight, alpha, **kwargs)                     # :62-63
    elif algorithm == 'cd':
        z = coord_descent(x, weight, z0, alpha, **kwargs)            # :54-55
    elif algorithm in _OFF_PATH_ALGOS:
        raise NotImplementedError(
            "lasso_amd accelerates algorithm='ista' and 'cd'; %r is not on the HIP path" % algorithm)
    else:
        raise ValueError("invalid algorithm parameter '{}'.".format(algorithm))  # :71
    return z
