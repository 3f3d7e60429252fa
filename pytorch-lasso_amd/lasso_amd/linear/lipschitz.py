"""lambda_max(W^T W): HIP replacement of ``_lipschitz_constant`` (reference
ista.py:8-14).  Deterministic fp64 on the device instead of a host ARPACK call."""
import ctypes as C

import torch

from .. import _native as nat


def lipschitz_constant(weight):
    """Returns a python float, like the reference.  ``weight`` [d,k] fp32."""
    nat.require_gpu()
    if weight.dtype != torch.float32:
        # the reference's bf16 + lr='auto' raises TypeError as well (ista.py:12)
        raise TypeError("lasso_amd: lr='auto' needs an fp32 dictionary, got %s" % weight.dtype)
    dev = weight.device if weight.is_cuda else torch.device('cuda', torch.cuda.current_device())
    w = weight.detach().to(dev).contiguous()
    d, k = w.shape
    L = nat.lib()
    with torch.cuda.device(dev):
        ws = nat.workspace(dev, L.lasso_lipschitz_workspace_bytes(d, k), tag='lip')
        out = C.c_double(0.0)
        nat.check(L.lasso_lipschitz(nat.ptr(w), w.stride(0), d, k, nat.LASSO_F32, C.byref(out),
                                    nat.ptr(ws), ws.numel(), nat.stream_ptr(dev)))
    return out.value
