"""HIP-backed counterpart of ``lasso.conv2d`` (reference lasso/conv2d/ista.py,
lasso/conv2d/lip_const.py)."""
from .ista import ista_conv2d  # noqa: F401
from .lip_const import lip_bound_conv2d, LipBoundConv2d  # noqa: F401
