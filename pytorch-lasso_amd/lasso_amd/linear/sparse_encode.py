"""Mirror of lasso/linear/sparse_encode.py:8-73 for the 'ista' arm."""
import torch

from .solvers import ista

_init_defaults = {'ista': 'zero'}                                   # sparse_encode.py:8-16

_OFF_PATH_ALGOS = ('cd', 'gpsr', 'iter-ridge', 'interior-point', 'split-bregman', 'own')


def initialize_code(x, weight, alpha, mode):
    """sparse_encode.py:19-35.  'zero' (:22-23) is the hot-path default; 'unif'
    (:24-25) and 'transpose' (:30-31) are plain tensor ops; 'lstsq'/'ridge'
    (:26-29) belong to lasso/linear/utils.py and are not part of this engine."""
    n_samples = x.size(0)
    n_components = weight.size(1)
    if mode == 'zero':
        z0 = x.new_zeros(n_samples, n_components)
    elif mode == 'unif':
        z0 = x.new(n_samples, n_components).uniform_(-0.1, 0.1)
    elif mode == 'transpose':
        z0 = torch.matmul(x, weight)
    elif mode in ('lstsq', 'ridge'):
        raise NotImplementedError("lasso_amd: init=%r is outside the HIP hot path" % mode)
    else:
        raise ValueError("invalid init parameter '{}'.".format(mode))   # :33
    return z0


def sparse_encode(x, weight, alpha=1.0, z0=None, algorithm='ista', init=None,
                  **kwargs):
    """Same call surface as lasso.linear.sparse_encode (sparse_encode.py:38-73);
    ``algorithm='ista'`` runs on the HIP engine, the other solver names raise
    NotImplementedError (they are outside the accelerated path), anything else
    raises ValueError like the reference (:71)."""
    n_samples = x.size(0)
    n_components = weight.size(1)
    if z0 is not None:
        assert z0.shape == (n_samples, n_components)                 # :44-45
    else:
        if init is None:
            init = _init_defaults.get(algorithm, 'zero')             # :47-48
        z0 = initialize_code(x, weight, alpha, mode=init)            # :51
    if algorithm == 'ista':
        z = ista(x, z0, weight, alpha, **kwargs)                     # :62-63
    elif algorithm in _OFF_PATH_ALGOS:
        raise NotImplementedError(
            "lasso_amd accelerates algorithm='ista' only; %r is not on the HIP path" % algorithm)
    else:
        raise ValueError("invalid algorithm parameter '{}'.".format(algorithm))  # :71
    return z
