"""adacharge_amd: MI355X-native batched MPC solver behind adacharge's API.

Package root re-exports the same names as the reference's
``adacharge/__init__.py`` (star-imports of adacharge, postprocessing and
adaptive_charging_optimization).
"""
from .adaptive_charging_optimization import (  # noqa: F401
    AdaptiveChargingOptimization,
    InfeasibilityException,
    ObjectiveComponent,
    QuadObjective,
    Rates,
    demand_charge,
    equal_share,
    load_flattening,
    peak,
    quick_charge,
    total_energy,
    tou_energy_cost,
)
from .adacharge import (  # noqa: F401
    AdaptiveChargingAlgorithmOffline,
    AdaptiveSchedulingAlgorithm,
    get_active_sessions,
)
from .postprocessing import (  # noqa: F401
    ceil_to_set,
    diff_based_reallocation,
    floor_to_set,
    increment_in_set,
    index_based_reallocation,
    project_into_continuous_feasible_pilots,
    project_into_discrete_feasible_pilots,
)
from .utils import infrastructure_constraints_feasible  # noqa: F401

__version__ = "0.1.0"
