"""Minimal stand-ins for the acnportal data contract the hot path touches.

acnportal is a third-party, un-pinned dependency of the reference
(/root/reference/setup.py:24) and is not installed here, so the attributes the
reference reads are restated from SURVEY.md Appendix B [recalled].  Only what
`adaptive_charging_optimization.py`, `adacharge.py`, `postprocessing.py` and
`utils.py` touch is provided; every attribute cites the reference line that
reads it.  If the real acnportal is importable its objects work unchanged --
everything downstream is duck-typed on these attribute names.
"""
from __future__ import annotations

from copy import copy as _shallow_copy, deepcopy
from typing import Dict, List, Optional

import numpy as np


class SessionInfo:
    """Charging-session description (acnportal.acnsim.interface.SessionInfo).

    Positional order follows the reference's own constructor call,
    adacharge.py:29-37.  Attributes read by the hot path:
    ``station_id`` (aco.py:63), ``session_id`` (aco.py:115), ``arrival_offset``
    / ``remaining_time`` (aco.py:64-73, 108-112, 244), ``remaining_demand``
    (aco.py:118-122), ``min_rates`` / ``max_rates`` (aco.py:68, 73).
    """

    def __init__(
        self,
        station_id,
        session_id,
        requested_energy,
        energy_delivered,
        arrival,
        departure,
        estimated_departure=None,
        current_time=0,
        min_rates=0,
        max_rates=float("inf"),
    ):
        self.station_id = station_id
        self.session_id = session_id
        self.requested_energy = requested_energy
        self.energy_delivered = energy_delivered
        self.arrival = arrival
        self.departure = departure
        self.estimated_departure = (
            estimated_departure if estimated_departure is not None else departure
        )
        self.current_time = current_time
        self.min_rates = self._expand(min_rates)
        self.max_rates = self._expand(max_rates)

    def _expand(self, rates):
        if np.isscalar(rates):
            return np.full(self.remaining_time, float(rates))
        return np.array(rates, dtype=float)

    @property
    def remaining_demand(self) -> float:
        return self.requested_energy - self.energy_delivered

    @property
    def arrival_offset(self) -> int:
        return max(self.arrival - self.current_time, 0)

    @property
    def remaining_time(self) -> int:
        remaining = min(
            self.departure - self.arrival, self.departure - self.current_time
        )
        return max(remaining, 0)


class InfrastructureInfo:
    """Electrical infrastructure description
    (acnportal.acnsim.interface.InfrastructureInfo).

    Read by the reference at aco.py:106,114,146-171,246; post.py:91-92,113-114,
    148,162,176; utils.py:6-10.
    """

    def __init__(
        self,
        constraint_matrix,
        constraint_limits,
        phases,
        voltages,
        constraint_ids=None,
        station_ids=None,
        max_pilot=None,
        min_pilot=None,
        allowable_pilots=None,
        is_continuous=None,
    ):
        self.constraint_matrix = (
            None if constraint_matrix is None else np.asarray(constraint_matrix, float)
        )
        self.constraint_limits = (
            None if constraint_limits is None else np.asarray(constraint_limits, float)
        )
        self.phases = None if phases is None else np.asarray(phases, float)
        self.voltages = np.asarray(voltages, float)
        n = len(self.voltages)
        m = 0 if self.constraint_matrix is None else self.constraint_matrix.shape[0]
        self.constraint_ids = (
            list(constraint_ids)
            if constraint_ids is not None
            else [f"_const_{j}" for j in range(m)]
        )
        self.station_ids = (
            list(station_ids)
            if station_ids is not None
            else [f"_station_{i}" for i in range(n)]
        )
        self._station_index = {s: i for i, s in enumerate(self.station_ids)}
        self.max_pilot = (
            np.asarray(max_pilot, float) if max_pilot is not None else np.full(n, np.inf)
        )
        self.min_pilot = (
            np.asarray(min_pilot, float) if min_pilot is not None else np.zeros(n)
        )
        self.allowable_pilots = (
            allowable_pilots if allowable_pilots is not None else [None] * n
        )
        self.is_continuous = (
            np.asarray(is_continuous, bool)
            if is_continuous is not None
            else np.ones(n, dtype=bool)
        )

    @property
    def num_stations(self) -> int:
        return len(self.station_ids)

    def get_station_index(self, station_id) -> int:
        return self._station_index[station_id]


class Interface:
    """Dict-backed interface: the attribute surface the reference reads from
    ``acnportal.acnsim.interface.Interface`` (``period`` aco.py:263,
    ``current_time`` ada.py:163, ``infrastructure_info()`` ada.py:139,
    ``get_prev_peak()`` ada.py:173 / aco.py:390, ``get_prices`` aco.py:379,
    ``get_demand_charge`` aco.py:399, ``remaining_amp_periods`` post.py:160).
    Mirrors acnportal's ``TestingInterface`` [recalled].
    """

    def __init__(self, data: Dict):
        self.data = data

    # -- state ---------------------------------------------------------------
    @property
    def period(self):
        return self.data["period"]

    @property
    def current_time(self):
        return self.data.get("current_time", 0)

    def active_sessions(self) -> List[SessionInfo]:
        sessions = self.data.get("active_sessions", [])
        out = []
        for s in sessions:
            if isinstance(s, SessionInfo):
                out.append(deepcopy(s))
            else:
                kw = dict(s)
                kw.setdefault("current_time", self.current_time)
                out.append(SessionInfo(**kw))
        return out

    def infrastructure_info(self) -> InfrastructureInfo:
        info = self.data["infrastructure_info"]
        if isinstance(info, InfrastructureInfo):
            return info
        return InfrastructureInfo(**info)

    # -- signals -------------------------------------------------------------
    def get_prev_peak(self):
        return self.data.get("prev_peak", 0)

    def get_prices(self, length, start=None):
        prices = np.asarray(self.data["prices"], float)
        t0 = self.current_time if start is None else start
        return prices[t0 : t0 + length]

    def get_demand_charge(self, start=None):
        return self.data["demand_charge"]

    def remaining_amp_periods(self, session: SessionInfo) -> float:
        infra = self.infrastructure_info()
        i = infra.get_station_index(session.station_id)
        amp_hours = session.remaining_demand * 1000 / infra.voltages[i]
        return amp_hours * 60 / self.period

    def is_feasible(self, load_currents, linear=False, violation_tolerance=1e-5):
        """acnportal ``Interface.is_feasible`` [recalled]: schedule dict ->
        bool, SOC norm (or LINEAR |C| sums) against the constraint limits."""
        infra = self.infrastructure_info()
        if infra.constraint_matrix is None or infra.constraint_matrix.size == 0:
            return True
        length = max((len(v) for v in load_currents.values()), default=0)
        rates = np.zeros((infra.num_stations, length))
        for sid, sched in load_currents.items():
            rates[infra.get_station_index(sid), : len(sched)] = sched
        cm = infra.constraint_matrix
        if linear:
            mag = np.abs(cm) @ rates
        else:
            ph = np.deg2rad(infra.phases)
            mag = np.hypot((cm * np.cos(ph)) @ rates, (cm * np.sin(ph)) @ rates)
        return bool(np.all(mag <= infra.constraint_limits[:, None] + violation_tolerance))


class BaseAlgorithm:
    """acnportal.algorithms.BaseAlgorithm [recalled, SURVEY.md section 8b]:
    ``_interface`` unset until ``register_interface``; ``max_recompute`` = 1;
    ``run()`` = ``schedule(interface.active_sessions())``."""

    def __init__(self):
        self._interface = None
        self.max_recompute = 1

    @property
    def interface(self):
        if self._interface is not None:
            return self._interface
        raise NotImplementedError(
            "No interface has been registered yet. Please call register_interface "
            "prior to using the algorithm."
        )

    def register_interface(self, interface) -> None:
        self._interface = interface

    def schedule(self, active_sessions):
        raise NotImplementedError

    def run(self):
        return self.schedule(self.interface.active_sessions())


# ---------------------------------------------------------------------------
# Session pre-processing (acnportal.algorithms.preprocessing) [recalled]; the
# reference calls them at adacharge.py:141-150.
# ---------------------------------------------------------------------------
def infrastructure_constraints_feasible(rates, infrastructure) -> bool:
    """Same formula and 1e-7 slack as the reference's utils.py:5-12."""
    phase_in_rad = np.deg2rad(infrastructure.phases)
    cm = infrastructure.constraint_matrix
    re = (cm * np.cos(phase_in_rad)) @ rates
    im = (cm * np.sin(phase_in_rad)) @ rates
    mag = np.hypot(re, im)
    lim = np.asarray(infrastructure.constraint_limits)
    if mag.ndim == 2:
        lim = lim[:, None]
    return bool(np.all(mag <= lim + 1e-7))


def reconcile_max_and_min(session: SessionInfo, choose_min: bool = True) -> SessionInfo:
    mask = session.max_rates < session.min_rates
    if choose_min:
        session.max_rates[mask] = session.min_rates[mask]
    else:
        session.min_rates[mask] = session.max_rates[mask]
    return session


def copy_sessions(active_sessions):
    """Independent copies of the sessions, as ``deepcopy(active_sessions)`` gives the reference's pre-processing steps
    -- written out, because a generic deepcopy of ~30 small objects was half of a single ``schedule()`` call's host time
    (0.75 of 1.5 ms, tools/profile_single_step.py): a shallow copy of each session whose arrays are copied and whose
    other mutable members (none in SessionInfo itself; a subclass may add some) still go through deepcopy."""
    out = []
    for s in active_sessions:
        if type(s) is SessionInfo:   # the usual case: scalars and the two rate arrays
            n = object.__new__(SessionInfo)
            n.__dict__.update(s.__dict__)
            n.min_rates = s.min_rates.copy()
            n.max_rates = s.max_rates.copy()
            out.append(n)
            continue
        n = _shallow_copy(s)
        for k, v in vars(n).items():
            if isinstance(v, np.ndarray):
                setattr(n, k, v.copy())
            elif not isinstance(v, _ATOMIC):
                setattr(n, k, deepcopy(v))
        out.append(n)
    return out


_ATOMIC = (int, float, str, bool, type(None), np.generic, tuple, frozenset, bytes, complex)


def enforce_pilot_limit(active_sessions, infrastructure):
    """Cap each session's max_rates at its EVSE's max_pilot (ada.py:141)."""
    new_sessions = copy_sessions(active_sessions)
    for session in new_sessions:
        i = infrastructure.get_station_index(session.station_id)
        session.max_rates = np.minimum(session.max_rates, infrastructure.max_pilot[i])
    return new_sessions


def apply_upper_bound_estimate(ub_estimator, active_sessions):
    """Cap max_rates with an estimator's per-session bound (ada.py:143-146)."""
    new_sessions = copy_sessions(active_sessions)
    upper_bounds = ub_estimator.get_maximum_rates(active_sessions)
    for session in new_sessions:
        session.max_rates = np.minimum(
            session.max_rates, upper_bounds.get(session.session_id, float("inf"))
        )
        reconcile_max_and_min(session)
    return new_sessions


def remaining_amp_periods(session, infrastructure, period) -> float:
    """acnportal.algorithms.utils.remaining_amp_periods [recalled]: A-periods still owed to a session."""
    i = infrastructure.get_station_index(session.station_id)
    amp_hours = session.remaining_demand * 1000 / infrastructure.voltages[i]
    return amp_hours * 60 / period


def apply_minimum_charging_rate(active_sessions, infrastructure, period, override=float("inf")):
    """acnportal.algorithms.preprocessing.apply_minimum_charging_rate [recalled], called at ada.py:147-150 as
    ``(active_sessions, infrastructure, self.interface.period)``: in arrival order, give each session the EVSE's
    minimum pilot (capped by ``override``) as ``min_rates[0]`` when it still needs that much charge and the
    network can carry it on top of the earlier arrivals' minimum pilots; otherwise pin its first period to zero."""
    session_queue = sorted(copy_sessions(active_sessions), key=lambda s: s.arrival)
    session_queue = [s for s in session_queue if s.remaining_time > 0]
    rates = np.zeros(len(infrastructure.station_ids))
    for session in session_queue:
        i = infrastructure.get_station_index(session.station_id)
        rates[i] = min(infrastructure.min_pilot[i], override)
        if remaining_amp_periods(session, infrastructure, period) >= rates[i] and infrastructure_constraints_feasible(
            rates, infrastructure
        ):
            session.min_rates[0] = max(rates[i], session.min_rates[0])
            reconcile_max_and_min(session)
        else:
            rates[i] = 0
            session.min_rates[0] = 0
            session.max_rates[0] = 0
    return session_queue


def earliest_deadline_first(sessions, interface):
    """acnportal sort function used by post.py:146 / t_post.py:14."""
    return sorted(sessions, key=lambda s: s.estimated_departure)


def least_laxity_first(sessions, interface):
    def laxity(s):
        lax = (s.estimated_departure - interface.current_time) - (
            interface.remaining_amp_periods(s) / np.max(s.max_rates)
        )
        return lax

    return sorted(sessions, key=laxity)
