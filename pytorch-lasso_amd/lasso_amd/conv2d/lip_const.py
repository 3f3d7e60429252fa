"""Toeplitz bound on the Lipschitz constant of a stride-1 conv2d: same call surface as
``lasso.conv2d.lip_const.lip_bound_conv2d`` / ``LipBoundConv2d`` (reference
lip_const.py:32-135), computed by ``lasso_conv_lip_bound`` on the GPU."""
import ctypes as C

import torch

from .. import _native as nat


def _check(kernel_size, stride):
    if not kernel_size[-1] == kernel_size[-2]:
        raise ValueError("The last 2 dim of the kernel must be equal.")     # :99-100
    if not kernel_size[-1] % 2 == 1:
        raise ValueError("The dimension of the kernel must be odd.")        # :101-102
    if not stride == 1:
        raise NotImplementedError("LipBound not implemented for stride > 1.")  # :103-104


def lip_bound_conv2d(kernel, padding, stride=1, sample=50, sqrt=False):
    """-> 0-d float32 tensor on the kernel's device, like the reference (:131-135)."""
    assert kernel.dim() == 4                                                # :98
    _check(kernel.shape, stride)
    nat.require_gpu()
    if kernel.dtype != torch.float32:
        raise NotImplementedError("lasso_amd: lip_bound_conv2d is implemented for float32 kernels")
    out_device = kernel.device
    dev = kernel.device if kernel.is_cuda else torch.device('cuda', torch.cuda.current_device())
    wg = kernel.detach().to(dev).contiguous()
    K, Cin, ks, _ = wg.shape
    L = nat.lib()
    val = C.c_double(0.0)
    with torch.cuda.device(dev):
        ws = nat.workspace(dev, L.lasso_conv_lip_workspace_bytes(K, Cin, ks, int(sample)), "convlip")
        nat.check(L.lasso_conv_lip_bound(nat.ptr(wg), K, Cin, ks, int(padding), int(sample), int(bool(sqrt)),
                                         C.byref(val), nat.ptr(ws), ws.numel(), nat.stream_ptr(dev)))
    return torch.tensor(val.value, dtype=torch.float32, device=out_device)


class LipBoundConv2d(torch.nn.Module):
    """Module form (lip_const.py:32-93): the geometry is fixed at construction, ``forward``
    takes the kernel."""

    def __init__(self, kernel_size, padding, stride=1, sample=50, sqrt=False):
        super().__init__()
        assert len(kernel_size) == 4                                        # :47
        _check(kernel_size, stride)
        self.ksize = kernel_size[-1]
        self.padding, self.sample, self.sqrt = padding, sample, sqrt

    def forward(self, kernel):
        assert kernel.dim() == 4                                            # :74
        assert kernel.size(2) == kernel.size(3) == self.ksize              # :75
        return lip_bound_conv2d(kernel, self.padding, sample=self.sample, sqrt=self.sqrt)
