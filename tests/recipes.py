"""Seeded synthetic-input recipes shared by the golden generator, the tests and
bench.py.  Inputs are always drawn on the CPU generator
(then copied to the device) so every box sees identical data."""
import torch
import torch.nn.functional as F

# lambda_max(W^T W) in fp64 for the C2/C3 recipe dictionary
LAMBDA_MAX_C2 = 8.877719052098003
# lambda_max for the C4 orthogonal-init dictionary
LAMBDA_MAX_C4 = 4.34106895730149


def recipe_xw(n, d=256, k=1024, seed=0):
    """W = normalize(randn(d,k)) drawn FIRST, then X = randn(n,d)."""
    g = torch.Generator().manual_seed(seed)
    W = F.normalize(torch.randn(d, k, generator=g), dim=0)
    X = torch.randn(n, d, generator=g)
    return X, W


def recipe_c4_init(d=256, k=1024):
    """Initial dictionary of the C4 EM recipe: torch.manual_seed(0);
    orthogonal_ on a [d,k] CPU tensor; column-normalised."""
    st = torch.get_rng_state()
    torch.manual_seed(0)
    W = torch.empty(d, k)
    torch.nn.init.orthogonal_(W)
    W = F.normalize(W, dim=0)
    torch.set_rng_state(st)
    return W


def recipe_c5(n, reseed=True):
    """Omniglot stand-in: centred uniform 8x8 'patches' (SURVEY 8d G5)."""
    if reseed:
        torch.manual_seed(0)
    X = torch.rand(n, 64)
    X -= X.mean(1, keepdim=True)
    return X
