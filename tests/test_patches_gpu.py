"""Patch front end against torch's unfold/fold on the CPU -- the
reference notebook is absent, so this pins the layout, not the notebook (parity with it
is unpinned)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,C,H,W,p,s", [(3, 1, 28, 28, 8, 1), (2, 3, 17, 21, (5, 3), (2, 3)), (1, 2, 8, 8, 8, 1),
                                         (4, 1, 105, 105, 8, 4)])
def test_extract_and_reconstruct(N, C, H, W, p, s):
    from lasso_amd.patches import extract_patches, reconstruct_from_patches
    g = torch.Generator().manual_seed(H)
    img = torch.rand(N, C, H, W, generator=g)
    F = torch.nn.functional
    ref = F.unfold(img, p, stride=s).transpose(1, 2).reshape(-1, F.unfold(img, p, stride=s).shape[1])
    X, means = extract_patches(img.cuda(), p, stride=s, center=False)
    assert means is None and torch.equal(X.cpu(), ref)
    Xc, mu = extract_patches(img.cuda(), p, stride=s, center=True)
    assert (mu.cpu() - ref.mean(1)).abs().max().item() <= 1e-6
    assert (Xc.cpu() - (ref - ref.mean(1, keepdim=True))).abs().max().item() <= 1e-6
    # overlap-average: fold(sum) / fold(ones)
    rec = reconstruct_from_patches(Xc, img.shape, p, stride=s, means=mu).cpu()
    cols = ref.reshape(N, -1, ref.shape[1]).transpose(1, 2)
    num = F.fold(cols, (H, W), p, stride=s)
    den = F.fold(torch.ones_like(cols), (H, W), p, stride=s)
    want = torch.where(den > 0, num / den.clamp(min=1), torch.zeros_like(num))
    assert (rec - want).abs().max().item() <= 1e-5
    covered = den > 0
    assert (rec - img)[covered].abs().max().item() <= 1e-5            # unmodified patches: exact inverse


def test_patch_pipeline_end_to_end():
    """image -> centred 8x8 patches -> dict_learning -> sparse codes -> reconstruction."""
    from lasso_amd.patches import extract_patches, reconstruct_from_patches
    from lasso_amd.linear import dict_learning, sparse_encode
    g = torch.Generator().manual_seed(0)
    img = torch.rand(6, 1, 32, 32, generator=g)
    X, mu = extract_patches(img.cuda(), 8, stride=2)
    torch.manual_seed(0)
    D, losses = dict_learning(X, 128, alpha=0.02, steps=8, progbar=False)
    assert losses[-1] < losses[0]
    Z = sparse_encode(X, D.cuda(), alpha=0.02, maxiter=50)
    rec = reconstruct_from_patches(Z @ D.cuda().T, img.shape, 8, stride=2, means=mu)
    err = (rec.cpu() - img).pow(2).mean().sqrt().item()
    assert err < 0.15, err      # rms of the images themselves is ~0.58
