"""Mirror of /root/reference/adacharge/utils.py."""
from .acn import infrastructure_constraints_feasible  # noqa: F401  (utils.py:5-12)
