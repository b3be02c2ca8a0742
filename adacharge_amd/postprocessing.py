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


# ---------------------------------------------------------------------------------------------------------------
#  Whole-batch versions (ada.py:176-189 applied to every snapshot of a batched solve at once).  Same results as
#  the per-snapshot functions above (tests/test_postprocessing.py), written so that the Python interpreter sees
#  array operations over the batch instead of a loop over snapshots, EVSEs and increments.
# ---------------------------------------------------------------------------------------------------------------
def _pilot_table(infrastructure):
    """allowable_pilots as one (N, L) array padded with +inf (sets may differ per EVSE)."""
    sets = [np.asarray(a, float) for a in infrastructure.allowable_pilots]
    L = max(len(a) for a in sets)
    tab = np.full((len(sets), L), np.inf)
    for i, a in enumerate(sets):
        tab[i, : len(a)] = a
    return tab


def project_into_continuous_feasible_pilots_batch(rates: np.ndarray, infrastructure):
    """post.py:77-94 for rates (B, N, T)."""
    cap = np.asarray(infrastructure.max_pilot, float)[None, :, None]
    return np.maximum(np.minimum(rates, cap), 0)


def project_into_discrete_feasible_pilots_batch(rates: np.ndarray, infrastructure):
    """post.py:97-118 for rates (B, N, T): EVSEs that share an allowable set are floored in one call."""
    out = np.array(rates, copy=True)
    groups = {}
    for i, a in enumerate(infrastructure.allowable_pilots):
        groups.setdefault(tuple(np.asarray(a, float)), []).append(i)
    for aset, idx in groups.items():
        out[:, idx, :] = floor_to_set(rates[:, idx, :], np.array(aset), eps=0.05)
    return np.maximum(out, 0)


def diff_based_reallocation_batch(rates: np.ndarray, table, infrastructure, interface):
    """post.py:189-258 for every snapshot of a batch: ``rates`` (B, N, T) solved schedules (zero-padded beyond a
    snapshot's horizon), ``table`` the SessionTable the batch was built from.  The greedy round-robin is sequential
    inside a snapshot; here every step advances ALL snapshots by one visit (each has its own visiting order and
    position), so the loop length is the longest snapshot's, not the sum."""
    B, N = rates.shape[0], rates.shape[1]
    init = rates[:, :, 0]
    peak_limit = init.sum(axis=1)
    rounded = project_into_discrete_feasible_pilots_batch(rates, infrastructure)
    col = rounded[:, :, 0].copy()
    # visiting order: a snapshot's sessions sorted by rounding loss, largest first (stable, like sorted())
    loss = -(init[table.prob, table.evse] - col[table.prob, table.evse])
    order = np.lexsort((loss, table.prob))
    cnt = np.bincount(table.prob, minlength=B)
    start = np.r_[0, np.cumsum(cnt)[:-1]]
    visit = np.full((B, max(int(cnt.max()), 1)), -1, dtype=np.int64)
    pos_in = np.arange(len(order)) - np.repeat(start, cnt)
    visit[table.prob[order], pos_in] = table.evse[order]
    # first-period caps of the sessions that have arrived (post.py:222-236)
    active = np.zeros((B, N), dtype=bool)
    ub = np.zeros((B, N))
    here = np.flatnonzero((table.off == 0) & (np.diff(table.seg) > 0))
    volt = np.asarray(infrastructure.voltages, float)
    amp_periods = table.demand[here] * 1000.0 / volt[table.evse[here]] * 60.0 / interface.period
    cap = np.minimum(np.minimum(amp_periods, table.max_rates[table.seg[here]]), np.asarray(infrastructure.max_pilot, float)[table.evse[here]])
    active[table.prob[here], table.evse[here]] = True
    ub[table.prob[here], table.evse[here]] = cap        # one arrived session per EVSE (a later one overwrites, as the loop does)
    pilots = _pilot_table(infrastructure)
    cm = infrastructure.constraint_matrix
    ph = np.deg2rad(infrastructure.phases)
    Cre, Cim = (cm * np.cos(ph)).T, (cm * np.sin(ph)).T
    lim = np.asarray(infrastructure.constraint_limits)[None, :] + 1e-7
    ptr = np.zeros(B, dtype=np.int64)
    rows = np.arange(B)
    n_vis = np.maximum(cnt, 1)
    while True:
        run = active.any(axis=1) & (cnt > 0)            # `if not active.any(): break`
        if not run.any():
            break
        i = visit[rows, ptr % n_vis]
        i = np.where(run, i, 0)
        ptr += run
        act = run & active[rows, i]                     # `if not active[i]: continue`
        cur = col[rows, i]
        full = act & (cur >= ub[rows, i])               # `if column[i] >= ub[i]: active[i] = False`
        active[rows[full], i[full]] = False
        tr = act & ~full
        if not tr.any():
            continue
        r_, i_ = rows[tr], i[tr]
        nxt_pos = (pilots[i_] > cur[tr][:, None]).argmax(axis=1)          # increment_in_set: next larger value ...
        has = (pilots[i_] > cur[tr][:, None]).any(axis=1)
        last = np.isfinite(pilots[i_]).sum(axis=1) - 1
        nxt = np.where(has, pilots[i_, nxt_pos], pilots[i_, last])        # ... clipped at the end of the set
        trial = col[r_].copy()
        trial[np.arange(len(r_)), i_] = nxt
        ok = (trial.sum(axis=1) <= peak_limit[r_]) & (nxt <= ub[r_, i_])
        mag = np.hypot(trial @ Cre, trial @ Cim)
        ok &= np.all(mag <= lim, axis=1)
        col[r_[ok], i_[ok]] = nxt[ok]
        active[r_[~ok], i_[~ok]] = False
    rounded[:, :, 0] = col
    return rounded
