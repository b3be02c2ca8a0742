"""Host-side mirror of the reference's optimisation surface
(/root/reference/adacharge/adaptive_charging_optimization.py).

Same names, arguments and error behaviour as the reference:
``AdaptiveChargingOptimization(objective, interface, constraint_type="SOC",
enforce_energy_equality=False, solver="ECOS")`` and ``.solve(active_sessions,
infrastructure, peak_limit=None, prev_peak=0, verbose=False) -> (N, T)`` array
(aco.py:31-43, 286-321); ``ObjectiveComponent`` (aco.py:12-15);
``InfeasibilityException`` (aco.py:8-9); the objective library (aco.py:336-408).

What changes is what sits underneath: the cvxpy ``Problem.solve`` call
(aco.py:315-318) is replaced by the structured builder (builder.py) and the
HIP batched ADMM solver behind the C-ABI (include/acn_qp.h).  There is no CPU
fallback: if the HIP library is missing, ``solve`` raises.

Deviations that cannot be avoided without cvxpy:
objective functions return ``QuadObjective`` descriptors, not cvxpy
expressions, so user-defined objectives must be built from the helpers here;
``build_problem`` returns the structured ``ProblemBatch`` instead of cvxpy
objects.
"""
from __future__ import annotations

from collections import namedtuple
from typing import List, Optional, Sequence, Union

import numpy as np


class InfeasibilityException(Exception):
    pass


ObjectiveComponent = namedtuple("ObjectiveComponent", ["function", "coefficient", "kwargs"])
ObjectiveComponent.__new__.__defaults__ = (1, {})


# ---------------------------------------------------------------------------
# Descriptor algebra standing in for cvxpy expressions
# ---------------------------------------------------------------------------
class Rates:
    """Symbolic handle for the (N, T) rates variable (cp.Variable at aco.py:247)."""

    def __init__(self, shape):
        self.shape = tuple(shape)


class QuadObjective:
    """A concave objective term in the reference's *maximisation* convention:

        value(r) = <lin, r>  -  sq * sum(r^2)  -  flat * sum_t (v' r[:, t])^2  +  const

    with v = voltages / 1e3.  Closed under ``+`` and scalar ``*`` exactly like
    the cvxpy expressions it replaces (aco.py:210-218)."""

    __array_priority__ = 1000

    def __init__(self, lin, sq=0.0, flat=0.0, const=0.0, epigraph=None):
        self.lin = np.asarray(lin, float)
        self.sq = float(sq)
        self.flat = float(flat)
        self.const = float(const)
        self.epigraph = list(epigraph) if epigraph else []

    @classmethod
    def zero(cls, shape):
        return cls(np.zeros(shape))

    def __add__(self, other):
        if np.isscalar(other):
            return QuadObjective(self.lin, self.sq, self.flat, self.const + other, self.epigraph)
        return QuadObjective(
            self.lin + other.lin, self.sq + other.sq, self.flat + other.flat,
            self.const + other.const, self.epigraph + other.epigraph,
        )

    __radd__ = __add__

    def __mul__(self, c):
        c = float(c)
        return QuadObjective(
            c * self.lin, c * self.sq, c * self.flat, c * self.const,
            [(c * w, spec) for w, spec in self.epigraph],
        )

    __rmul__ = __mul__

    def __neg__(self):
        return self * -1.0

    def __sub__(self, other):
        return self + (-other)


# ---------------------------------------------------------------------------
#  Objective functions -- same signatures as aco.py:336-408.  All take rates as
#  first positional argument and swallow unknown keyword arguments.
# ---------------------------------------------------------------------------
def _volt_kw(infrastructure):
    return np.asarray(infrastructure.voltages, float) / 1e3


def quick_charge(rates, infrastructure, interface, **kwargs):
    """aco.py:363-371: sum_t ((T - t) / T) * sum_i r[i, t]."""
    T = rates.shape[1]
    c = np.array([(T - t) / T for t in range(T)])
    return QuadObjective(np.broadcast_to(c, rates.shape).copy())


def equal_share(rates, infrastructure, interface, **kwargs):
    """aco.py:374-375: -sum_squares(rates)."""
    return QuadObjective(np.zeros(rates.shape), sq=1.0)


def tou_energy_cost(rates, infrastructure, interface, **kwargs):
    """aco.py:378-380: -prices @ aggregate_period_energy."""
    prices = np.asarray(interface.get_prices(rates.shape[1]), float)
    v = _volt_kw(infrastructure)
    return QuadObjective(-(v[:, None] * (interface.period / 60)) * prices[None, :])


def total_energy(rates, infrastructure, interface, **kwargs):
    """aco.py:383-384: sum of per-period energy in kWh."""
    v = _volt_kw(infrastructure)
    return QuadObjective(np.repeat((v * (interface.period / 60))[:, None], rates.shape[1], axis=1))


def load_flattening(rates, infrastructure, interface, external_signal=None, **kwargs):
    """aco.py:403-408: -sum_t (v' r_t + ext_t)^2."""
    T = rates.shape[1]
    ext = np.zeros(T) if external_signal is None else np.asarray(external_signal, float)
    v = _volt_kw(infrastructure)
    return QuadObjective(-2.0 * v[:, None] * ext[None, :], flat=1.0, const=-float(ext @ ext))


def peak(rates, infrastructure, interface, baseline_peak=0, **kwargs):
    """aco.py:387-394: max(max_t aggregate_power_t, prev_peak kW[, baseline_peak]).
    Convex; only usable with a negative coefficient (as ``demand_charge`` does).
    Carried as an epigraph request for the builder."""
    prev_peak = interface.get_prev_peak() * infrastructure.voltages[0] / 1000
    floor = max(prev_peak, baseline_peak) if baseline_peak > 0 else prev_peak
    return QuadObjective(np.zeros(rates.shape), epigraph=[(1.0, {"floor": float(floor)})])


def demand_charge(rates, infrastructure, interface, baseline_peak=0, **kwargs):
    """aco.py:397-400: -demand_charge * peak."""
    p = peak(rates, infrastructure, interface, baseline_peak, **kwargs)
    dc = interface.get_demand_charge()
    return -dc * p


# ---------------------------------------------------------------------------
#  The optimiser (aco.py:18-321)
# ---------------------------------------------------------------------------
_HANDLE_CACHE = {}          # insertion-ordered: least recently used first
_HANDLE_CACHE_MAX = 64


def _site_handle(infrastructure, constraint_type, with_peak, device, with_flat=False, with_max=False):
    """One uploaded site per (infrastructure content, cone, peak, GPU).  The
    reference rebuilds every atom on every call (adacharge.py:152-158); here the
    site matrix is uploaded once and reused by all later MPC steps."""
    from . import backend
    from .builder import make_site

    cm = infrastructure.constraint_matrix
    key = (
        None if cm is None else (cm.shape, cm.tobytes()),
        None if infrastructure.constraint_limits is None else np.asarray(infrastructure.constraint_limits).tobytes(),
        None if infrastructure.phases is None else np.asarray(infrastructure.phases).tobytes(),
        np.asarray(infrastructure.voltages).tobytes(),
        constraint_type, bool(with_peak), int(device), bool(with_flat), bool(with_max),
    )
    ent = _HANDLE_CACHE.pop(key, None)
    if ent is None:
        site = make_site(infrastructure, constraint_type, with_peak=with_peak, with_flat=with_flat, with_max=with_max)
        ent = (site, backend.SiteHandle(site, device))
        while len(_HANDLE_CACHE) >= _HANDLE_CACHE_MAX:   # least recently used out; a caller that still holds the pair
            _HANDLE_CACHE.pop(next(iter(_HANDLE_CACHE)))  # keeps its handle alive (SiteHandle frees the device side in __del__)
    _HANDLE_CACHE[key] = ent                             # (re-)inserted last = most recently used
    return ent


def warn_inaccurate(status, pri_res, dua_res):
    """A schedule accepted as OPTIMAL_INACCURATE (aco.py:319 accepts it, and so does this path) is announced the way the
    reference's solver layer announces it -- cvxpy's ``Problem.solve`` emits a UserWarning "Solution may be inaccurate"
    for that status -- with the residuals the device reached, so the caller can judge the distance to the tolerances
    asked for (``last_result`` keeps them per problem).  Returns the number of such problems."""
    import warnings

    from . import backend

    status = np.asarray(status)
    bad = np.flatnonzero(status == backend.STATUS_SOLVED_INACCURATE)
    if bad.size:
        warnings.warn(
            f"Solution may be inaccurate: {bad.size} of {status.size} problems ended OPTIMAL_INACCURATE "
            f"(worst primal residual {float(np.max(np.asarray(pri_res)[bad])):.2e}, dual {float(np.max(np.asarray(dua_res)[bad])):.2e}); "
            "raise max_iter / retry_passes or inspect last_result", UserWarning, stacklevel=3)
    return int(bad.size)


class AdaptiveChargingOptimization:
    """Base class for all MPC based charging algorithms (aco.py:18-43).

    Args:
        objective (List[ObjectiveComponent]): components of the objective.
        interface: object providing ``period`` (and prices / demand charge /
            previous peak when the objective needs them).
        constraint_type (str): 'SOC' or 'LINEAR' (aco.py:35).
        enforce_energy_equality (bool): energy rows are equalities (aco.py:36).
        solver (str): accepted for source compatibility with the reference
            ("ECOS", "OSQP", None, cvxpy constants ...); every value runs the
            HIP ADMM backend.  Backend options go in ``solver_options``.
        solver_options (dict): overrides of ``acnqp_options`` fields
            (eps_abs, eps_rel, max_iter, rho, reg_rel, precision, ...), plus
            ``retry_stalled`` (default True; False = ``retry_passes=0``):
            problems the adaptive first pass leaves on a residual plateau
            are solved again from a cold start with a fixed penalty, inside
            the library (``acnqp_options.retry_passes``, include/acn_qp.h).
        device (int): HIP device ordinal.
    """

    def __init__(
        self,
        objective: List[ObjectiveComponent],
        interface,
        constraint_type="SOC",
        enforce_energy_equality=False,
        solver="ECOS",
        solver_options: Optional[dict] = None,
        device: int = 0,
    ):
        self.interface = interface
        self.constraint_type = constraint_type
        self.enforce_energy_equality = enforce_energy_equality
        self.solver = solver
        self.objective_configuration = objective
        self.solver_options = dict(solver_options or {})
        self.device = device
        self.last_result = None
        self._last_plan = None
        self._last_batch = None

    # the single-problem API asks for tighter residuals than the batch default so that the
    # reference's own test tolerances (1e-7 on the peak row, t_aco.py:257) hold
    _SINGLE_DEFAULTS = dict(eps_abs=1e-9, eps_rel=1e-9, max_iter=100000)

    # -- the reference's constraint builders (aco.py:45-198) -----------------------------------
    # They return cvxpy constraints there; here they return the numeric description of the same
    # constraint blocks (the arrays the C-ABI consumes), keyed exactly like the reference's dicts.
    @staticmethod
    def charging_rate_bounds(rates, active_sessions, evse_index):
        """aco.py:45-79: ``lb <= rates <= ub``; later sessions overwrite their own window and
        ``ub < lb`` is replaced by ``lb``."""
        lb, ub = np.zeros(rates.shape), np.zeros(rates.shape)
        for session in active_sessions:
            i = evse_index.index(session.station_id)
            w = slice(session.arrival_offset, session.arrival_offset + session.remaining_time)
            lb[i, w] = session.min_rates
            ub[i, w] = session.max_rates
        ub[ub < lb] = lb[ub < lb]
        return {"charging_rate_bounds.lb": lb, "charging_rate_bounds.ub": ub}

    @staticmethod
    def energy_constraints(rates, active_sessions, infrastructure, period, enforce_energy_equality=False):
        """aco.py:81-124: one row per session, ``coef * sum(rates[i, off:off+len]) <= (==) rhs``."""
        out = {}
        for session in active_sessions:
            i = infrastructure.get_station_index(session.station_id)
            out[f"energy_constraints.{session.session_id}"] = dict(
                evse=i, offset=session.arrival_offset, length=session.remaining_time,
                coefficient=infrastructure.voltages[i] * period / 1e3 / 60,
                rhs=session.remaining_demand, equality=bool(enforce_energy_equality),
            )
        return out

    @staticmethod
    def infrastructure_constraints(rates, infrastructure, constraint_type="SOC"):
        """aco.py:126-179: per constraint id, the row(s) applied to every period and the limit."""
        from .builder import _bad_constraint_type

        cm = infrastructure.constraint_matrix
        if cm is None or cm.shape == (0, 0):
            return {}
        out = {}
        if constraint_type == "SOC":
            if infrastructure.phases is None:
                raise ValueError("phases is required when using SOC infrastructure constraints.")
            ph = np.deg2rad(infrastructure.phases)
            for j, v in enumerate(cm):
                out[f"infrastructure_constraints.{infrastructure.constraint_ids[j]}"] = dict(
                    rows=np.stack([v * np.cos(ph), v * np.sin(ph)]), norm="l2", limit=infrastructure.constraint_limits[j])
        elif constraint_type == "LINEAR":
            for j, v in enumerate(cm):
                out[f"infrastructure_constraints.{infrastructure.constraint_ids[j]}"] = dict(
                    rows=np.abs(v)[None, :], norm="linear", limit=infrastructure.constraint_limits[j])
        else:
            _bad_constraint_type(constraint_type)
        return out

    @staticmethod
    def peak_constraint(rates, peak_limit):
        """aco.py:181-198: ``sum(rates, axis=0) <= peak_limit`` if a limit is given."""
        if peak_limit is not None:
            return {"peak_constraint": np.broadcast_to(np.asarray(peak_limit, float), (rates.shape[1],)).copy()}
        return {}

    def build_objective(self, rates, infrastructure, **kwargs):
        """aco.py:200-218: weighted sum of the objective components (a ``QuadObjective``)."""
        obj = QuadObjective.zero(rates.shape)
        for component in self.objective_configuration:
            merged = dict(kwargs)
            merged.update(component.kwargs)
            obj = obj + component.coefficient * component.function(rates, infrastructure, self.interface, **merged)
        return obj

    def build_problem(
        self,
        active_sessions,
        infrastructure,
        peak_limit: Optional[Union[float, List[float], np.ndarray]] = None,
        prev_peak: float = 0,
    ):
        """aco.py:220-284.  Returns the structured one-problem ``ProblemBatch``
        (the reference returns cvxpy objects; see module docstring)."""
        from .builder import build_batch

        return build_batch(
            [active_sessions], infrastructure, self.interface, self.objective_configuration,
            self.constraint_type, self.enforce_energy_equality,
            peak_limits=[peak_limit], prev_peak=prev_peak,
        )

    def solve(
        self,
        active_sessions,
        infrastructure,
        peak_limit: Union[float, List[float], np.ndarray] = None,
        prev_peak=0,
        verbose: bool = False,
        warm_start=None,
    ):
        """aco.py:286-321: (N, T) array of charging rates; raises
        ``InfeasibilityException`` unless the solve ends optimal.

        ``warm_start`` (extension; the reference starts every solve from nothing, adacharge.py:152-158):
        ``(x0, y0)`` -- a schedule (N, >= T) and the site-row multipliers ``last_multipliers`` (Mg, >= T) of an
        earlier, similar solve, already shifted to this problem's first period (the site must carry the same rows:
        same constraint type, peak / load-flattening / demand-charge rows).  Changes the iteration count, not the optimum."""
        if len(active_sessions) == 0:  # aco.py:310-311
            return np.zeros((infrastructure.num_stations, 1))
        rates, status = self.solve_batch(
            [active_sessions], infrastructure, peak_limits=[peak_limit], prev_peak=prev_peak,
            verbose=verbose, _defaults=self._SINGLE_DEFAULTS, warm_start=[warm_start],
        )
        from . import backend

        if status[0] not in backend.ACCEPTED_STATUSES:  # aco.py:319-320 (OPTIMAL or OPTIMAL_INACCURATE)
            raise InfeasibilityException(
                f"Solve failed with status {backend.STATUS_NAMES.get(int(status[0]), status[0])}"
            )
        return rates[0]

    def solve_table(self, table, infrastructure, peak_limits=None, prev_peak=0, _defaults: Optional[dict] = None,
                    warm_start=None):
        """Batched solve of a ``session_table.SessionTable`` (every snapshot non-empty): the array-native entry --
        no Python loop over sessions anywhere on the path, and (round 4) no dense (B, N, T) problem array on the host
        either: the sessions go to the library as they are (``acnqp_solve_table``) and ``charging_rate_bounds`` /
        ``energy_constraints`` / the linear cost (aco.py:45-124, 200-218) take their array form on the device.
        Returns ``(backend.BatchResult, builder.TablePlan)``; a warm start takes the dense entry (``plan.expand()``)."""
        from . import backend
        from .builder import _bad_constraint_type, _objective_needs_flat, _objective_needs_max, plan_from_table

        if self.constraint_type not in ("SOC", "LINEAR"):
            _bad_constraint_type(self.constraint_type)
        pl = [None] * table.B if peak_limits is None else list(peak_limits)
        site, handle = _site_handle(
            infrastructure, self.constraint_type, any(p is not None for p in pl), self.device,
            with_flat=_objective_needs_flat(self.objective_configuration),
            with_max=_objective_needs_max(self.objective_configuration),
        )
        plan = plan_from_table(
            table, infrastructure, self.interface, self.objective_configuration, self.constraint_type,
            self.enforce_energy_equality, peak_limits=pl, prev_peak=prev_peak, site=site,
        )
        opts = dict(_defaults or {})
        opts.update(self.solver_options)
        if not bool(opts.pop("retry_stalled", True)):   # shorthand kept from round 2: the passes live in the library now
            opts["retry_passes"] = 0
        self._last_plan, self._last_batch = plan, None
        if warm_start is not None and any(w is not None for w in warm_start):
            # per problem (x0, y0) or None; a problem without one starts from its own cold start's point only if all
            # are None -- mixed batches give the cold problems x0 = y0 = 0 (a valid, if plain, starting point)
            batch = self.last_batch
            wx = np.zeros((batch.B, batch.N, batch.Tm))
            wy = np.zeros((batch.B, site.Mg, batch.Tm))
            for b, w in enumerate(warm_start):
                if w is not None:
                    x0, y0 = np.asarray(w[0], float), np.asarray(w[1], float)
                    T = min(batch.Tm, x0.shape[1], y0.shape[1])
                    wx[b, :, :T], wy[b, :, :T] = x0[:, :T], y0[:, :T]
            res = handle.solve(batch, backend.default_options(**opts), warm=(wx, wy), want_y=True)
        else:
            res = handle.solve_table(plan, backend.default_options(**opts), want_y=True)
        self.last_result = res
        return res, plan

    @property
    def last_batch(self):
        """The structured problems behind ``last_result`` as dense arrays (diagnostics, tests): expanded on demand."""
        if getattr(self, "_last_batch", None) is None and getattr(self, "_last_plan", None) is not None:
            self._last_batch = self._last_plan.expand()
        return getattr(self, "_last_batch", None)

    @property
    def last_multipliers(self):
        """(B, Mg, Tm) site-row multipliers of the last solve: the ``y0`` of a later warm start."""
        return None if self.last_result is None else self.last_result.y

    def solve_batch(
        self,
        session_lists: Sequence[Sequence],
        infrastructure,
        peak_limits: Optional[Sequence] = None,
        prev_peak=0,
        verbose: bool = False,
        _defaults: Optional[dict] = None,
        warm_start: Optional[Sequence] = None,
    ):
        """Batched extension (not in the reference): one independent MPC
        problem per entry of ``session_lists`` (state snapshots, sites' days,
        demand scenarios), all solved in one kernel launch.  Returns
        ``(list of (N, T_b) arrays, status (B,) int32)``; never raises for
        solver outcomes -- inspect ``status``."""
        from . import backend
        from .builder import build_batch

        if self.constraint_type not in ("SOC", "LINEAR"):
            from .builder import _bad_constraint_type

            _bad_constraint_type(self.constraint_type)
        B = len(session_lists)
        nonempty = [k for k, sl in enumerate(session_lists) if len(sl) > 0]
        rates = [np.zeros((infrastructure.num_stations, 1)) for _ in range(B)]
        status = np.full(B, backend.STATUS_SOLVED, dtype=np.int32)
        if not nonempty:
            return rates, status
        pl = [None] * B if peak_limits is None else list(peak_limits)
        from .session_table import SessionTable

        table = SessionTable.from_sessions([session_lists[k] for k in nonempty], infrastructure)
        ws = None if warm_start is None else [warm_start[k] for k in nonempty]
        res, batch = self.solve_table(table, infrastructure, [pl[k] for k in nonempty], prev_peak, _defaults, ws)
        if verbose:
            for j, k in enumerate(nonempty):
                print(
                    f"[acnqp] problem {k}: status={backend.STATUS_NAMES.get(int(res.status[j]))} "
                    f"iters={int(res.iters[j])} pri_res={res.pri_res[j]:.3e} dua_res={res.dua_res[j]:.3e} "
                    f"obj={-(res.obj[j] + batch.const[j]):.9g} (maximised) kernel_ms={res.kernel_ms:.3f}"
                )
        for j, k in enumerate(nonempty):
            rates[k] = res.x[j, :, : int(batch.T[j])].copy()
            status[k] = res.status[j]
        warn_inaccurate(res.status, res.pri_res, res.dua_res)
        return rates, status
