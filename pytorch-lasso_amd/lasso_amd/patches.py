"""Patch front end around the dictionary-learning path: image ->
patches -> centre -> ``dict_learning`` -> reconstruct, on the GPU (csrc/conv.hip).

The reference's notebook that did this (examples/dict_learning_omniglot.ipynb, BASELINE
config 5) is absent from its checkout, so there is no reference code to mirror and parity
with it is UNPINNED; the layout is that of ``torch.nn.functional.unfold`` (rows = patch
positions in raster order, columns = (channel, row, col)), which the tests check.
"""
import torch

from . import _native as nat


def _pair(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


def extract_patches(images, patch_size, stride=1, center=True):
    """images [N,C,H,W] float32 -> (X [M, C*ph*pw], means [M] or None);  M = N*Ph*Pw.
    ``center`` subtracts every patch's own mean (the usual preprocessing for patch
    dictionaries)."""
    nat.require_gpu()
    if images.dim() != 4 or images.dtype != torch.float32:
        raise ValueError("extract_patches expects a float32 [N,C,H,W] tensor")
    ph, pw = _pair(patch_size)
    sh, sw = _pair(stride)
    N, C, H, W = images.shape
    if ph > H or pw > W:
        raise ValueError("patch larger than the image")
    dev = images.device if images.is_cuda else torch.device('cuda', torch.cuda.current_device())
    img = images.detach().to(dev).contiguous()
    M = N * ((H - ph) // sh + 1) * ((W - pw) // sw + 1)
    X = torch.empty((M, C * ph * pw), dtype=torch.float32, device=dev)
    means = torch.empty(M, dtype=torch.float32, device=dev) if center else None
    with torch.cuda.device(dev):
        nat.check(nat.lib().lasso_patches_extract(nat.ptr(img), nat.ptr(X), X.stride(0) if M else C * ph * pw,
                                                  nat.ptr(means), N, C, H, W, ph, pw, sh, sw, int(bool(center)),
                                                  nat.stream_ptr(dev)))
    return X, means


def reconstruct_from_patches(patches, image_shape, patch_size, stride=1, means=None):
    """Inverse of ``extract_patches`` by overlap-averaging: patches [M, C*ph*pw] (+ means [M])
    -> images ``image_shape`` = (N,C,H,W).  Exact when the patches are unmodified."""
    nat.require_gpu()
    N, C, H, W = (int(v) for v in image_shape)
    ph, pw = _pair(patch_size)
    sh, sw = _pair(stride)
    M = N * ((H - ph) // sh + 1) * ((W - pw) // sw + 1)
    if patches.dim() != 2 or patches.shape != (M, C * ph * pw) or patches.dtype != torch.float32:
        raise ValueError("patches must be float32 [%d, %d]" % (M, C * ph * pw))
    dev = patches.device if patches.is_cuda else torch.device('cuda', torch.cuda.current_device())
    P = patches.detach().to(dev).contiguous()
    mu = means.detach().to(dev).contiguous() if means is not None else None
    out = torch.empty((N, C, H, W), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        nat.check(nat.lib().lasso_patches_reconstruct(nat.ptr(P), P.stride(0) if M else C * ph * pw, nat.ptr(mu),
                                                      nat.ptr(out), N, C, H, W, ph, pw, sh, sw, nat.stream_ptr(dev)))
    return out
