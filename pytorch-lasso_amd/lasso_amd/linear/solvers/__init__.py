"""Mirror of lasso/linear/solvers/__init__.py:1 -- only the 'ista' arm is on the
hot path."""
from .ista import ista  # noqa: F401
