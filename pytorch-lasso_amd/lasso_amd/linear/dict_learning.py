"""HIP-backed mirror of lasso/linear/dict_learning.py (reference :10-123): same
function names, argument order, defaults and in-place semantics."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _native as nat
from ..engine import HipEngine
from ..parallel import em_loop, constrained_mstep
from .sparse_encode import sparse_encode


def _engine_for(*tensors, device=None):
    for t in tensors:
        if t is not None and t.is_cuda:
            return HipEngine(t.device)
    return HipEngine(device)


def lasso_loss(X, Z, weight, alpha=1.0):
    """(0.5*||X - Z W^T||^2 + alpha*||Z||_1)/n as a 0-d tensor (dict_learning.py:10-13)."""
    eng = _engine_for(X, Z, weight)
    out_device = X.device
    loss, _ = eng.objective_sums(eng.to_device(X), eng.to_device(Z), eng.to_device(weight), alpha)
    return loss.to(out_device)


def dict_evaluate(X, weight, alpha, **kwargs):
    """dict_learning.py:16-20."""
    X = X.to(weight.device)
    Z = sparse_encode(X, weight, alpha, **kwargs)
    return lasso_loss(X, Z, weight, alpha)


def update_dict(dictionary, X, Z, random_seed=None, positive=False, eps=1e-10):
    """Constrained M-step (dict_learning.py:56-103): Gauss-Seidel sweep over the atoms
    with unit-norm projection, computed in Gram form on the GPU.  Like the reference it
    updates ``dictionary`` AND ``Z`` in place (degenerate atoms get a fresh random
    direction drawn from torch's CPU generator and their codes are zeroed) and returns
    ``dictionary``."""
    if random_seed is not None:
        torch.manual_seed(random_seed)                               # :78-79
    eng = _engine_for(dictionary, X, Z)
    Dg, Xg, Zg = eng.to_device(dictionary), eng.to_device(X), eng.to_device(Z)
    d, k = Dg.shape
    buf = torch.empty(k * k + k * d, dtype=torch.float32, device=eng.device)
    A, B = eng.gram(Zg, Xg, buf)
    mask = constrained_mstep(eng, A, B, Dg, eps=eps, positive=positive)
    if mask is not None:
        eng.zero_columns(Zg, mask)
    if Dg.data_ptr() != dictionary.data_ptr():
        dictionary.copy_(Dg)
    if Zg.data_ptr() != Z.data_ptr():
        Z.copy_(Zg)
    return dictionary


def update_dict_ridge(x, z, lambd=1e-4):
    """Unconstrained M-step V = ((Z^T Z + lambd*n*I)^-1 Z^T X)^T (dict_learning.py:106-123)."""
    eng = _engine_for(x, z)
    out_device = x.device
    xg, zg = eng.to_device(x), eng.to_device(z)
    n, d = xg.shape
    k = zg.shape[1]
    buf = torch.empty(k * k + k * d, dtype=torch.float32, device=eng.device)
    A, B = eng.gram(zg, xg, buf)
    return eng.ridge(A, B, lambd * n, check=True).to(out_device)


def dict_learning(X, n_components, alpha=1.0, constrained=True, persist=False,
                  lambd=1e-2, steps=60, device='cpu', progbar=True,
                  init_weight=None, **solver_kwargs):
    """EM dictionary learning (dict_learning.py:23-53) on the HIP engine.

    ``device`` names where the RESULT lives (reference default 'cpu'); the arithmetic
    always runs on the current HIP device.  The initial dictionary is drawn exactly like
    the reference (orthogonal_ + column normalisation on ``device``, :28-31) unless
    ``init_weight`` (extension) is given.  Returns ``(weight [d,k], losses [steps])``."""
    nat.require_gpu()
    n_samples, n_features = X.shape
    out_device = torch.device(device)
    if init_weight is None:
        weight = torch.empty(n_features, n_components, device=out_device)   # :28
        nn.init.orthogonal_(weight)                                         # :29
        if constrained:
            weight = F.normalize(weight, dim=0)                             # :30-31
    else:
        weight = init_weight.detach().clone()
    eng = _engine_for(X if X.is_cuda else None,
                      device=out_device if out_device.type == 'cuda' else None)
    weight, losses = em_loop(eng, eng.to_device(X), eng.to_device(weight).clone(), alpha,
                             constrained=constrained, persist=persist, lambd=lambd, steps=steps,
                             progbar=progbar, solver_kwargs=solver_kwargs)
    return weight.to(out_device), losses.to(out_device)
