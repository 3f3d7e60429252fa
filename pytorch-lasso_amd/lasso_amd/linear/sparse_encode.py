"""Mirror of lasso/linear/sparse_encode.py:8-73 for the 'ista' arm."""
import torch

from .solvers import ista, coord_descent

_init_defaults = {'ista': 'zero'}                                   # sparse_encode.py:8-16

_OFF_PATH_ALGOS = ('gpsr', 'iter-ridge', 'interior-point', 'split-bregman', 'own')


def _lstsq_init(x, weight):
    # least-norm (d < k) / least-squares (d >= k) code through a reduced QR of the dictionary
    d, k = weight.shape
    rhs = x.T
    if d < k:
        q, r = torch.linalg.qr(weight.T, mode='reduced')
        return (q @ torch.linalg.solve_triangular(r.T, rhs, upper=False)).T
    q, r = torch.linalg.qr(weight, mode='reduced')
    return torch.linalg.solve_triangular(r, q.T @ rhs, upper=True).T


def _ridge_init(x, weight, alpha):
    # (W^T W + alpha I) z = W^T x per sample, Cholesky
    gram = weight.T @ weight
    gram.diagonal().add_(alpha)
    chol, info = torch.linalg.cholesky_ex(gram)
    if info != 0:
        raise RuntimeError("The Gram matrix is not positive definite. Try increasing 'alpha'.")
    return torch.cholesky_solve(weight.T @ x.T, chol).T


def initialize_code(x, weight, alpha, mode):
    """sparse_encode.py:19-35.  'zero' (:22-23) is the hot-path default; the other modes
    ('unif', 'lstsq', 'ridge') are one-off set-ups that run as torch / torch.linalg calls
    (rocSOLVER when the tensors are on the GPU) -- library plumbing, not part of the HIP hot
    path; 'transpose' is a product on the library's own GEMM (lasso_init_transpose)."""
    n_samples = x.size(0)
    n_components = weight.size(1)
    if mode == 'zero':
        z0 = x.new_zeros(n_samples, n_components)
    elif mode == 'unif':
        z0 = x.new(n_samples, n_components).uniform_(-0.1, 0.1)
    elif mode == 'transpose':                                        # :24-25, on the library's NT GEMM
        from ..engine import HipEngine
        from .. import _native as nat
        nat.require_gpu()
        eng = HipEngine(x.device if x.is_cuda else (weight.device if weight.is_cuda else None))
        z0 = eng.init_transpose(eng.to_device(x).float().contiguous(),
                                eng.to_device(weight).float().contiguous()).to(x.device)
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
