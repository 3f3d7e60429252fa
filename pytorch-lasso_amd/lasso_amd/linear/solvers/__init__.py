"""Mirror of lasso/linear/solvers/__init__.py:1,5 -- the solvers on the HIP path:
'ista'  and greedy coordinate descent 'cd' (8f row f2)."""
from .ista import ista  # noqa: F401
from .coordinate_descent import coord_descent  # noqa: F401
