"""Differentiable encode: gradients of the HIP solve
(lasso_fista_backward) against torch.autograd through the CPU oracle's unrolled loop --
which is how the reference itself is differentiated (ista.py:57-104 is plain torch code).
Tolerance: 2e-4 of the gradient's max magnitude (fp32 GEMMs in a different summation order
on both the forward iterates and the backward products)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mods():
    from lasso_amd.linear import sparse_encode
    from lasso_amd.linear.solvers import ista
    from oracle import lasso_oracle as orc
    return sparse_encode, ista, orc


def _problem(n, d, k, seed=0):
    g = torch.Generator().manual_seed(seed)
    W = torch.nn.functional.normalize(torch.randn(d, k, generator=g), dim=0)
    X = torch.randn(n, d, generator=g)
    Z0 = torch.randn(n, k, generator=g) * 0.05
    G = torch.randn(n, k, generator=g)
    return X, W, Z0, G


def _grads(fn, X, W, Z0, G, dev):
    x = X.detach().clone().to(dev).requires_grad_(True)
    w = W.detach().clone().to(dev).requires_grad_(True)
    z0 = Z0.detach().clone().to(dev).requires_grad_(True)
    z = fn(x, z0, w)
    (z * G.to(dev)).sum().backward()
    return z.detach().cpu(), x.grad.cpu(), w.grad.cpu(), z0.grad.cpu()


@pytest.mark.parametrize("n,d,k", [(37, 10, 50), (64, 256, 1024), (130, 128, 512), (16, 64, 256),
                                   (40, 300, 70), (24, 96, 1300)])     # the last two: beyond the fused shapes
@pytest.mark.parametrize("fast", [True, False])
def test_gradients_match_autograd_through_the_oracle(n, d, k, fast):
    _, ista, orc = _mods()
    X, W, Z0, G = _problem(n, d, k)
    lr = 1.0 / orc.lipschitz_constant(W, "exact")
    for T in (1, 6):
        ref = _grads(lambda x, z0, w: orc.fista(x, z0, w, 0.3, fast=fast, lr=lr, maxiter=T, tol=0.0), X, W, Z0, G, "cpu")
        got = _grads(lambda x, z0, w: ista(x, z0, w, 0.3, fast=fast, lr=lr, maxiter=T, tol=0.0), X, W, Z0, G, "cuda")
        assert (got[0] - ref[0]).abs().max().item() <= 5e-5
        for name, a, b in zip(("dx", "dW", "dz0"), got[1:], ref[1:]):
            assert a.shape == b.shape
            assert (a - b).abs().max().item() <= 2e-4 * max(b.abs().max().item(), 1e-3), (name, T)


def test_early_stop_and_partial_requires_grad():
    sparse_encode, ista, orc = _mods()
    X, W, Z0, G = _problem(48, 32, 96, seed=3)
    lr = 1.0 / orc.lipschitz_constant(W, "exact")
    # the stop rule ends the loop early; the gradient is that of the executed iterations
    ref = _grads(lambda x, z0, w: orc.fista(x, z0, w, 0.5, lr=lr, maxiter=400, tol=1e-4), X, W, Z0, G, "cpu")
    got = _grads(lambda x, z0, w: ista(x, z0, w, 0.5, lr=lr, maxiter=400, tol=1e-4), X, W, Z0, G, "cuda")
    assert (got[0] - ref[0]).abs().max().item() <= 5e-5
    assert (got[2] - ref[2]).abs().max().item() <= 5e-4 * ref[2].abs().max().item()
    # only the dictionary needs a gradient (the dictionary-learning-by-backprop use case)
    w = W.cuda().requires_grad_(True)
    z = sparse_encode(X.cuda(), w, alpha=0.5, lr=lr, maxiter=5, tol=0.0)
    assert z.requires_grad
    (z * G.cuda()).sum().backward()
    wr = W.clone().requires_grad_(True)
    (orc.sparse_encode(X, wr, alpha=0.5, lr=lr, maxiter=5, tol=0.0) * G).sum().backward()
    assert (w.grad.cpu() - wr.grad).abs().max().item() <= 2e-4 * wr.grad.abs().max().item()
    # no grad requested -> the fused path, no graph
    with torch.no_grad():
        assert not sparse_encode(X.cuda(), w, alpha=0.5, lr=lr, maxiter=5, tol=0.0).requires_grad


@pytest.mark.parametrize("n,d,k", [(37, 10, 50), (64, 256, 1024), (40, 300, 70)])   # the last: beyond the fused shapes
@pytest.mark.parametrize("fast", [True, False])
def test_gradients_through_the_line_search(n, d, k, fast):
    """backtrack=True (ista.py:17-54): the accepted steps are python floats, i.e. constants of the reference's
    graph; the derivative is that of the iterations taken with those steps."""
    _, ista, orc = _mods()
    X, W, Z0, G = _problem(n, d, k, seed=5)
    trace = orc.FistaTrace()
    orc.fista(X, Z0, W, 0.3, fast=fast, lr=1.0, maxiter=5, tol=0.0, backtrack=True, trace=trace)
    assert max(trace.trials) > 1                      # the search really shrinks the step
    ref = _grads(lambda x, z0, w: orc.fista(x, z0, w, 0.3, fast=fast, lr=1.0, maxiter=5, tol=0.0, backtrack=True),
                 X, W, Z0, G, "cpu")
    got = _grads(lambda x, z0, w: ista(x, z0, w, 0.3, fast=fast, lr=1.0, maxiter=5, tol=0.0, backtrack=True),
                 X, W, Z0, G, "cuda")
    assert (got[0] - ref[0]).abs().max().item() <= 5e-5
    for name, a, b in zip(("dx", "dW", "dz0"), got[1:], ref[1:]):
        assert (a - b).abs().max().item() <= 2e-4 * max(b.abs().max().item(), 1e-3), name
    # return_info reports the search next to a result that carries the graph
    x = X.cuda().requires_grad_(True)
    z, info = ista(x, Z0.cuda(), W.cuda(), 0.3, fast=fast, lr=1.0, maxiter=5, tol=0.0, backtrack=True, return_info=True)
    assert z.requires_grad and info["trials"] == trace.trials
    assert max(abs(a - b) for a, b in zip(info["accepted_lr"], trace.accepted_lr)) <= 1e-6


def test_unsupported_combinations_fail_loudly():
    _, ista, _ = _mods()
    X, W, Z0, _ = _problem(8, 16, 32)
    with pytest.raises(NotImplementedError):
        ista(X, Z0, W.clone().requires_grad_(True), 0.3, lr=0.1, maxiter=3)      # CPU tensors
