"""Algorithm adapters -- mirror of /root/reference/adacharge/adacharge.py.

``AdaptiveSchedulingAlgorithm`` keeps the acnportal ``BaseAlgorithm`` plugin
contract (constructor kwargs ada.py:43-58, ``register_interface`` ada.py:116-133,
``schedule(active_sessions) -> {station_id: ndarray}`` ada.py:135-193); the body
is re-pointed at the HIP backend through ``AdaptiveChargingOptimization``.
"""
from __future__ import annotations

import warnings
from copy import deepcopy

import numpy as np

from .acn import (
    BaseAlgorithm,
    SessionInfo,
    apply_minimum_charging_rate,
    apply_upper_bound_estimate,
    enforce_pilot_limit,
)
from .adaptive_charging_optimization import *  # noqa: F401,F403  (ada.py:5 star-import)
from .adaptive_charging_optimization import AdaptiveChargingOptimization
from .postprocessing import (
    diff_based_reallocation,
    index_based_reallocation,
    project_into_continuous_feasible_pilots,
    project_into_discrete_feasible_pilots,
)


def get_active_sessions(active_evs, current_time):
    """ada.py:18-39: SessionInfo list for acnsim EV objects."""
    return [
        SessionInfo(
            ev.station_id, ev.session_id, ev.requested_energy, ev.energy_delivered,
            ev.arrival, ev.departure, current_time=current_time,
        )
        for ev in active_evs
    ]


class AdaptiveSchedulingAlgorithm(BaseAlgorithm):
    """Model Predictive Control based scheduling algorithm (ada.py:42-193)."""

    def __init__(
        self,
        objective,
        constraint_type="SOC",
        enforce_energy_equality=False,
        solver=None,
        peak_limit=None,
        estimate_max_rate=False,
        max_rate_estimator=None,
        uninterrupted_charging=False,
        quantize=False,
        reallocate=False,
        max_recompute=None,
        allow_overcharging=False,
        verbose=False,
        solver_options=None,
        device=0,
        warm_start=False,
    ):
        super().__init__()
        self.objective = objective
        self.constraint_type = constraint_type
        self.enforce_energy_equality = enforce_energy_equality
        self.solver = solver
        self.peak_limit = peak_limit
        # extension (off by default: the reference rebuilds everything each step and keeps no state, ada.py:152-158):
        # start each solve from the previous step's schedule and multipliers, shifted by the periods elapsed
        self.warm_start = warm_start
        self._warm = None   # (current_time, x (N, T), y (N, T)) of the previous solve
        self.estimate_max_rate = estimate_max_rate
        self.max_rate_estimator = max_rate_estimator
        self.uninterrupted_charging = uninterrupted_charging
        self.quantize = quantize
        self.reallocate = reallocate
        self.verbose = verbose
        self.solver_options = solver_options
        self.device = device
        if not self.quantize and self.reallocate:
            raise ValueError(
                "reallocate cannot be true without quantize. "
                "Otherwise there is nothing to reallocate :)."
            )
        if self.quantize:
            # ada.py:106-111 reads the BaseAlgorithm default before assigning (Appendix D.1)
            if self.max_recompute is not None:
                warnings.warn("Overriding max_recompute to 1 since quantization is on.")
            self.max_recompute = 1
        else:
            self.max_recompute = max_recompute
        self.allow_overcharging = allow_overcharging

    def register_interface(self, interface):
        """ada.py:116-133."""
        self._interface = interface
        if self.max_rate_estimator is not None:
            self.max_rate_estimator.register_interface(interface)

    def _preprocess(self, active_sessions, infrastructure):
        """ada.py:141-150."""
        active_sessions = enforce_pilot_limit(active_sessions, infrastructure)
        if self.estimate_max_rate:
            active_sessions = apply_upper_bound_estimate(self.max_rate_estimator, active_sessions)
        if self.uninterrupted_charging:
            active_sessions = apply_minimum_charging_rate(
                active_sessions, infrastructure, self.interface.period
            )
        return active_sessions

    def _trimmed_peak(self, active_sessions):
        """ada.py:160-167: scalar passes through, vector is sliced from the
        current simulation time."""
        if self.peak_limit is None or np.isscalar(self.peak_limit):
            return self.peak_limit
        t = self.interface.current_time
        horizon = max(s.arrival_offset + s.remaining_time for s in active_sessions)
        return self.peak_limit[t : t + horizon]

    def _postprocess(self, rates_matrix, active_sessions, infrastructure):
        """ada.py:176-189."""
        if self.quantize:
            if self.reallocate:
                rates_matrix = diff_based_reallocation(
                    rates_matrix, active_sessions, infrastructure, self.interface
                )
            else:
                rates_matrix = project_into_discrete_feasible_pilots(rates_matrix, infrastructure)
        else:
            rates_matrix = project_into_continuous_feasible_pilots(rates_matrix, infrastructure)
        return np.maximum(rates_matrix, 0)

    def _optimizer(self):
        return AdaptiveChargingOptimization(
            self.objective, self.interface, self.constraint_type, self.enforce_energy_equality,
            solver=self.solver, solver_options=self.solver_options, device=self.device,
        )

    def schedule(self, active_sessions):
        """See BaseAlgorithm (ada.py:135-193)."""
        if len(active_sessions) == 0:
            return {}
        infrastructure = self.interface.infrastructure_info()
        active_sessions = self._preprocess(active_sessions, infrastructure)
        optimizer = self._optimizer()
        warm = None
        if self.warm_start and self._warm is not None:
            t_prev, x_prev, y_prev = self._warm
            dt = self.interface.current_time - t_prev
            if 0 <= dt < x_prev.shape[1]:
                warm = (x_prev[:, dt:], y_prev[:, dt:])
        rates_matrix = optimizer.solve(
            active_sessions, infrastructure,
            peak_limit=self._trimmed_peak(active_sessions),
            prev_peak=self.interface.get_prev_peak(),
            verbose=self.verbose,
            warm_start=warm,
        )
        if self.warm_start:
            T = rates_matrix.shape[1]
            self._warm = (self.interface.current_time, rates_matrix.copy(), optimizer.last_multipliers[0][:, :T].copy())
        self.last_iterations = int(optimizer.last_result.iters[0])
        rates_matrix = self._postprocess(rates_matrix, active_sessions, infrastructure)
        return {
            station_id: rates_matrix[i, :] for i, station_id in enumerate(infrastructure.station_ids)
        }

    def schedule_batch(self, session_lists, peak_limits=None, as_arrays=False):
        """Batched extension: one schedule per state snapshot, every optimisation problem solved by one pipelined
        pass of the HIP library, and the steps either side of the solve -- the pre-processing of ada.py:141-150 and
        the post-processing of ada.py:176-189 -- done as array operations over the whole batch
        (session_table.py, postprocessing.*_batch).  ``session_lists``: a list of SessionInfo lists, or a
        ``SessionTable`` (no Python objects at all).  Returns one ``{station_id: ndarray}`` dict per snapshot
        (``{}`` for an empty one, ``None`` where the solve did not end optimal), or, with ``as_arrays=True``,
        ``(rates (B, N, Tmax), status (B,))``."""
        from . import session_table as st
        from .postprocessing import (
            diff_based_reallocation_batch,
            project_into_continuous_feasible_pilots_batch,
            project_into_discrete_feasible_pilots_batch,
        )

        infrastructure = self.interface.infrastructure_info()
        if isinstance(session_lists, st.SessionTable):
            table, B, nonempty, lists = session_lists, session_lists.B, np.arange(session_lists.B), None
        else:
            B = len(session_lists)
            nonempty = np.array([k for k, sl in enumerate(session_lists) if len(sl) > 0], dtype=np.int64)
            lists = [session_lists[k] for k in nonempty]
            table = st.SessionTable.from_sessions(lists, infrastructure) if len(nonempty) else None
        status = np.full(B, 1, dtype=np.int32)
        if table is None:
            return (np.zeros((B, infrastructure.num_stations, 1)), status) if as_arrays else [{} for _ in range(B)]
        table = st.enforce_pilot_limit(table, infrastructure)                       # ada.py:141
        if self.estimate_max_rate:                                                  # ada.py:143-146
            if lists is None:
                raise ValueError("estimate_max_rate needs SessionInfo lists (the estimator API takes sessions)")
            table = st.apply_upper_bound_estimate(table, [self.max_rate_estimator.get_maximum_rates(sl) for sl in lists])
        if self.uninterrupted_charging:                                             # ada.py:147-150
            table = st.apply_minimum_charging_rate(table, infrastructure, self.interface.period)
        if peak_limits is None:                                                     # ada.py:160-167
            if self.peak_limit is None or np.isscalar(self.peak_limit):
                pl = [self.peak_limit] * table.B
            else:
                end = np.zeros(table.B, dtype=np.int64)
                np.maximum.at(end, table.prob, table.off + table.rem)
                t = self.interface.current_time
                pl = [self.peak_limit[t : t + int(e)] for e in end]
        else:
            pl = [peak_limits[k] for k in nonempty]
        res, batch = self._optimizer().solve_table(table, infrastructure, pl, self.interface.get_prev_peak())
        from .adaptive_charging_optimization import warn_inaccurate

        warn_inaccurate(res.status, res.pri_res, res.dua_res)   # cvxpy's "Solution may be inaccurate" (aco.py:315-321)
        if self.quantize:                                                           # ada.py:176-184
            if self.reallocate:
                r = diff_based_reallocation_batch(res.x, table, infrastructure, self.interface)
            else:
                r = project_into_discrete_feasible_pilots_batch(res.x, infrastructure)
        else:                                                                       # ada.py:185-188
            r = project_into_continuous_feasible_pilots_batch(res.x, infrastructure)
        r = np.maximum(r, 0)                                                        # ada.py:189
        if as_arrays:
            rates = np.zeros((B,) + r.shape[1:])
            rates[nonempty] = r
            status[nonempty] = res.status
            return rates, status
        out = [{} for _ in range(B)]
        ids = infrastructure.station_ids
        for j, k in enumerate(nonempty):
            if res.status[j] not in (1, 5):   # OPTIMAL / OPTIMAL_INACCURATE, as aco.py:319
                out[k] = None
            else:
                rj = r[j, :, : int(batch.T[j])]
                out[k] = {sid: rj[i] for i, sid in enumerate(ids)}
        return out


class AdaptiveChargingAlgorithmOffline(BaseAlgorithm):
    """Offline optimisation with perfect future information (ada.py:196-294)."""

    def __init__(
        self,
        objective,
        constraint_type="SOC",
        enforce_energy_equality=False,
        solver=None,
        peak_limit=None,
        verbose=False,
        solver_options=None,
        device=0,
    ):
        super().__init__()
        self.max_recompute = 1
        self.objective = objective
        self.constraint_type = constraint_type
        self.enforce_energy_equality = enforce_energy_equality
        self.solver = solver
        self.peak_limit = peak_limit
        self.verbose = verbose
        self.solver_options = solver_options
        self.device = device
        self.sessions = None
        self.session_ids = None
        self.internal_schedule = None

    def register_events(self, events):
        """ada.py:234-247: only Plugin events are considered."""
        active_evs = [deepcopy(event[1].ev) for event in events.queue if event[1].event_type == "Plugin"]
        self.sessions = get_active_sessions(active_evs, 0)
        self.session_ids = set(s.session_id for s in self.sessions)

    def solve(self):
        """ada.py:249-276."""
        if self._interface is None:
            raise ValueError(
                "Error: self.interface is None. Please register interface before calling solve."
            )
        if self.sessions is None:
            raise ValueError(
                "No events registered. Please register an event queue before calling solve."
            )
        infrastructure = self.interface.infrastructure_info()
        self.sessions = enforce_pilot_limit(self.sessions, infrastructure)
        optimizer = AdaptiveChargingOptimization(
            self.objective, self.interface, self.constraint_type, self.enforce_energy_equality,
            solver=self.solver, solver_options=self.solver_options, device=self.device,
        )
        rates_matrix = optimizer.solve(self.sessions, infrastructure, self.peak_limit, verbose=self.verbose)
        rates_matrix = project_into_continuous_feasible_pilots(rates_matrix, infrastructure)
        self.internal_schedule = {
            station_id: rates_matrix[i, :] for i, station_id in enumerate(infrastructure.station_ids)
        }

    def schedule(self, active_evs):
        """ada.py:278-294."""
        if self.internal_schedule is None:
            raise ValueError(
                "No internal schedule found. Make sure to call solve before calling schedule or "
                "running a simulation."
            )
        for ev in active_evs:
            if ev.session_id not in self.session_ids:
                raise ValueError(f"Error: Session {ev.session_id} not included in offline solve.")
        current_time = self.interface.current_time
        return {ev.station_id: [self.internal_schedule[ev.station_id][current_time]] for ev in active_evs}
