"""lasso_amd -- MI355X-native ISTA/FISTA sparse-encode engine behind the
``lasso.linear`` call surface of rfeinman/pytorch-lasso.

    from lasso_amd.linear import sparse_encode, dict_learning

All arithmetic of the hot path runs in hand-written HIP kernels (gfx950) behind
the C ABI declared in ``include/lasso_hip.h``; PyTorch only provides device
memory, streams and ``torch.distributed``.  There is NO CPU fallback: without
the HIP extension and a visible GPU every entry point raises.
"""
from . import linear  # noqa: F401
from ._native import NativeError, lib_path  # noqa: F401

__version__ = "0.1.0"
