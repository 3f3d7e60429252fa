"""Mirror of lasso/linear/__init__.py:1-4 for the HIP hot path."""
from . import solvers  # noqa: F401
from .sparse_encode import sparse_encode, initialize_code  # noqa: F401
