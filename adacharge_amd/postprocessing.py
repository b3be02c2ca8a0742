"""Post-processing of a solved schedule -- mirror of
/root/reference/adacharge/postprocessing.py (same names and semantics, exact
known answers pinned by tests/test_postprocessing.py from the reference's own
t_post.py), written on numpy ``searchsorted`` instead of per-element bisect
loops so whole rows are rounded at once.
"""
from __future__ import annotations

from itertools import cycle
from typing import List

import numpy as np

from .acn import infrastructure_constraints_feasible


def _as_set(allowable_set):
    return np.asarray(allowable_set, dtype=float)


def floor_to_set(x, allowable_set, eps=0.05):
    """post.py:10-31: round x down into the set, but up when within eps of
    the next value; clip to the ends of the set.  Works on scalars and arrays."""
    s = _as_set(allowable_set)
    xa = np.asarray(x, dtype=float)
    pos = np.searchsorted(s, xa + eps, side="left")
    exact = (pos < len(s)) & (xa == s[np.minimum(pos, len(s) - 1)])
    out = s[np.clip(pos - 1, 0, len(s) - 1)]
    out = np.where(pos == 0, s[0], out)
    out = np.where(pos == len(s), s[-1], out)
    out = np.where(exact, xa, out)
    return out if out.ndim else out[()]


def ceil_to_set(x, allowable_set, eps=0.05):
    """post.py:34-55: round x up into the set, but down when within eps of the
    next lower value; clip to the ends of the set."""
    s = _as_set(allowable_set)
    xa = np.asarray(x, dtype=float)
    pos = np.searchsorted(s, xa - eps, side="right")
    exact = (pos > 0) & (xa == s[np.maximum(pos - 1, 0)])
    out = s[np.clip(pos, 0, len(s) - 1)]
    out = np.where(pos == 0, s[0], out)
    out = np.where(pos == len(s), s[-1], out)
    out = np.where(exact, xa, out)
    return out if out.ndim else out[()]


def increment_in_set(x, allowable_set):
    """post.py:58-74: next larger value of the set, clipped at its ends."""
    s = _as_set(allowable_set)
    xa = np.asarray(x, dtype=float)
    pos = np.searchsorted(s, xa, side="right")
    out = s[np.clip(pos, 0, len(s) - 1)]
    return out if out.ndim else out[()]


def project_into_continuous_feasible_pilots(rates: np.ndarray, infrastructure):
    """post.py:77-94: clip every row to [0, max_pilot_i]."""
    n = infrastructure.num_stations
    new_rates = np.array(rates, copy=True)
    cap = np.asarray(infrastructure.max_pilot)[:n]
    new_rates[:n] = np.minimum(new_rates[:n], cap.reshape((n,) + (1,) * (new_rates.ndim - 1)))
    return np.maximum(new_rates, 0)


def project_into_discrete_feasible_pilots(rates: np.ndarray, infrastructure):
    """post.py:97-118: floor every entry into its EVSE's allowable pilot set
    (eps = 0.05)."""
    new_rates = np.array(rates, copy=True)
    for i in range(infrastructure.num_stations):
        new_rates[i, :] = floor_to_set(rates[i, :], infrastructure.allowable_pilots[i], eps=0.05)
    return np.maximum(new_rates, 0)


def _first_period_caps(active_sessions, infrastructure, interface):
    """post.py:150-164 / 222-236: per-EVSE activity mask and upper bound for
    the first control period (sessions that have not arrived are ignored)."""
    n = infrastructure.num_stations
    active = np.zeros(n, dtype=bool)
    ub = np.zeros(n)
    for session in active_sessions:
        if session.arrival_offset == 0:
            i = infrastructure.station_ids.index(session.station_id)
            active[i] = True
            ub[i] = min(
                interface.remaining_amp_periods(session),
                session.max_rates[0],
                infrastructure.max_pilot[i],
            )
    return active, ub


def _round_robin_increment(column, order, active, ub, infrastructure, peak_limit):
    """Shared greedy loop of post.py:166-185 and 238-257: visit EVSEs in
    ``order`` cyclically; bump one to its next allowable pilot when the
    aggregate stays under ``peak_limit``, the EVSE under its cap and the network
    feasible (utils.py:5-12); otherwise retire it.  ``column`` is updated in place."""
    if len(order) == 0:
        return column
    for i in cycle(order):
        if not active.any():
            break
        if not active[i]:
            continue
        if column[i] >= ub[i]:
            active[i] = False
            continue
        trial = np.array(column, copy=True)
        trial[i] = increment_in_set(column[i], infrastructure.allowable_pilots[i])
        if (
            trial.sum() <= peak_limit
            and trial[i] <= ub[i]
            and infrastructure_constraints_feasible(trial, infrastructure)
        ):
            column[:] = trial
        else:
            active[i] = False
    return column


def index_based_reallocation(rates, active_sessions, infrastructure, peak_limit, sort_fn, interface):
    """post.py:121-186.  Mutates and returns ``rates`` like the reference
    (SURVEY.md Appendix D.5)."""
    order = [infrastructure.get_station_index(s.station_id) for s in sort_fn(active_sessions, interface)]
    active, ub = _first_period_caps(active_sessions, infrastructure, interface)
    col = rates[:, 0].copy()
    _round_robin_increment(col, order, active, ub, infrastructure, peak_limit)
    rates[:, 0] = col
    return rates


def diff_based_reallocation(rates, active_sessions, infrastructure, interface):
    """post.py:189-258: quantise, then hand the first period's rounding loss
    back, largest loss first."""
    init_rates = rates[:, 0]
    peak_limit = init_rates.sum()
    rounded = project_into_discrete_feasible_pilots(rates, infrastructure)

    def loss(session):
        i = infrastructure.get_station_index(session.station_id)
        return -(init_rates[i] - rounded[i, 0])

    order = [infrastructure.get_station_index(s.station_id) for s in sorted(active_sessions, key=loss)]
    active, ub = _first_period_caps(active_sessions, infrastructure, interface)
    col = rounded[:, 0].copy()
    _round_robin_increment(col, order, active, ub, infrastructure, peak_limit)
    rounded[:, 0] = col
    return rounded
