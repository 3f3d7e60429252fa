"""Mirror of lasso/linear/__init__.py:1-4 for the HIP hot path."""
from . import solvers  # noqa: F401
from .dict_learning import (dict_learning, dict_evaluate, update_dict,  # noqa: F401
                            update_dict_ridge, lasso_loss)
from .sparse_encode import sparse_encode, initialize_code  # noqa: F401
