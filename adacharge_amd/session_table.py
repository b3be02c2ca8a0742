"""Struct-of-arrays description of the active sessions of a whole batch of MPC snapshots, and the reference's
session pre-processing (adacharge.py:141-150) as array operations on it.

The reference walks Python ``SessionInfo`` objects one at a time, deep-copying the list at every step
(acnportal.algorithms.preprocessing [recalled]).  With thousands of snapshots per solve that walk is the wall clock,
so the batch path keeps ONE table for all snapshots:

    prob[s], evse[s], off[s], rem[s], demand[s], arrival[s]      one entry per session
    min_rates / max_rates                                         ragged, concatenated; session s owns [seg[s], seg[s+1])

``SessionTable.from_sessions`` is the only place that touches Python objects (one pass); callers that already hold
arrays (recorded data, scenario generators: ``sites.snapshot_table``) never create the objects at all.  The three
pre-processing functions below give exactly the result of their per-session twins in ``acn.py`` (tests/test_session_table.py).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np


@dataclass
class SessionTable:
    B: int                      # snapshots (problems)
    N: int                      # EVSEs of the site
    prob: np.ndarray            # (S,) int64   snapshot index
    evse: np.ndarray            # (S,) int64   station index
    off: np.ndarray             # (S,) int64   arrival_offset
    rem: np.ndarray             # (S,) int64   remaining_time
    demand: np.ndarray          # (S,) float64 remaining_demand (kWh)
    arrival: np.ndarray         # (S,) int64   arrival period (orders apply_minimum_charging_rate)
    seg: np.ndarray             # (S + 1,) int64 offsets into min_rates / max_rates (lengths = len of the rate vectors)
    min_rates: np.ndarray       # (seg[-1],) float64
    max_rates: np.ndarray       # (seg[-1],) float64
    session_ids: Optional[list] = None   # for estimators keyed by session id

    @property
    def S(self) -> int:
        return len(self.prob)

    def copy(self) -> "SessionTable":
        return SessionTable(self.B, self.N, self.prob.copy(), self.evse.copy(), self.off.copy(), self.rem.copy(),
                            self.demand.copy(), self.arrival.copy(), self.seg.copy(), self.min_rates.copy(),
                            self.max_rates.copy(), None if self.session_ids is None else list(self.session_ids))

    @classmethod
    def from_sessions(cls, session_lists: Sequence[Sequence], infrastructure) -> "SessionTable":
        """One pass over the SessionInfo objects of every snapshot (the attributes aco.py:61-123 reads)."""
        index = {s: i for i, s in enumerate(infrastructure.station_ids)}
        prob, evse, off, rem, dem, arr, ids, mins, maxs = [], [], [], [], [], [], [], [], []
        for b, sl in enumerate(session_lists):
            for s in sl:
                prob.append(b)
                evse.append(index[s.station_id])
                off.append(s.arrival_offset)
                rem.append(s.remaining_time)
                dem.append(s.remaining_demand)
                arr.append(s.arrival)
                ids.append(s.session_id)
                mins.append(s.min_rates)
                maxs.append(s.max_rates)
        lens = np.fromiter((len(m) for m in mins), dtype=np.int64, count=len(mins))
        seg = np.zeros(len(mins) + 1, dtype=np.int64)
        np.cumsum(lens, out=seg[1:])
        cat = lambda parts: np.concatenate([np.asarray(p, float) for p in parts]) if parts else np.zeros(0)
        return cls(len(session_lists), len(infrastructure.station_ids), np.asarray(prob, np.int64), np.asarray(evse, np.int64),
                   np.asarray(off, np.int64), np.asarray(rem, np.int64), np.asarray(dem, float), np.asarray(arr, np.int64),
                   seg, cat(mins), cat(maxs), ids)

    def take(self, idx) -> "SessionTable":
        """The sessions ``idx`` (in that order) as a new table: a permutation and / or a selection."""
        idx = np.asarray(idx, np.int64)
        lens = np.diff(self.seg)[idx]
        seg = np.zeros(len(idx) + 1, dtype=np.int64)
        np.cumsum(lens, out=seg[1:])
        src = np.repeat(self.seg[:-1][idx] - seg[:-1], lens) + np.arange(seg[-1])   # gather of the ragged rate vectors
        return SessionTable(self.B, self.N, self.prob[idx], self.evse[idx], self.off[idx], self.rem[idx], self.demand[idx],
                            self.arrival[idx], seg, self.min_rates[src], self.max_rates[src],
                            None if self.session_ids is None else [self.session_ids[k] for k in idx])

    def owner(self) -> np.ndarray:
        """Session index of every entry of the ragged rate arrays."""
        return np.repeat(np.arange(self.S), np.diff(self.seg))


# ---------------------------------------------------------------------------------------------------------------
#  Pre-processing (adacharge.py:141-150) on the table; each returns a new table like its per-session twin returns
#  a new session list.
# ---------------------------------------------------------------------------------------------------------------
def _reconcile(table: SessionTable, sessions: Optional[np.ndarray] = None) -> None:
    """reconcile_max_and_min(choose_min=True): where max < min take min -- on all sessions or a subset."""
    bad = table.max_rates < table.min_rates
    if sessions is not None:
        pick = np.zeros(table.S, dtype=bool)
        pick[sessions] = True
        bad &= pick[table.owner()]
    table.max_rates[bad] = table.min_rates[bad]


def enforce_pilot_limit(table: SessionTable, infrastructure) -> SessionTable:
    """Cap every session's max_rates at its EVSE's max_pilot (adacharge.py:141)."""
    out = table.copy()
    cap = np.asarray(infrastructure.max_pilot, float)[out.evse]
    out.max_rates = np.minimum(out.max_rates, np.repeat(cap, np.diff(out.seg)))
    return out


def apply_upper_bound_estimate(table: SessionTable, upper_bounds: Sequence[dict]) -> SessionTable:
    """Cap max_rates with an estimator's per-session bound, then reconcile with the min rates (adacharge.py:143-146).
    ``upper_bounds[b]`` is the dict ``max_rate_estimator.get_maximum_rates(sessions of snapshot b)`` returned."""
    out = table.copy()
    ub = np.fromiter((upper_bounds[b].get(sid, np.inf) for b, sid in zip(out.prob, out.session_ids)), dtype=float, count=out.S)
    out.max_rates = np.minimum(out.max_rates, np.repeat(ub, np.diff(out.seg)))
    _reconcile(out)
    return out


def _network_feasible(rates: np.ndarray, infrastructure) -> np.ndarray:
    """utils.py:5-12 for a batch of first-period rate vectors (B, N): SOC magnitude <= limit + 1e-7 on every row."""
    cm = infrastructure.constraint_matrix
    if cm is None or cm.size == 0:
        return np.ones(len(rates), dtype=bool)
    ph = np.deg2rad(infrastructure.phases)
    mag = np.hypot(rates @ (cm * np.cos(ph)).T, rates @ (cm * np.sin(ph)).T)
    return np.all(mag <= np.asarray(infrastructure.constraint_limits)[None, :] + 1e-7, axis=1)


def apply_minimum_charging_rate(table: SessionTable, infrastructure, period, override=float("inf")) -> SessionTable:
    """acn.apply_minimum_charging_rate for every snapshot at once: in arrival order, a session gets its EVSE's
    minimum pilot as min_rates[0] when it still needs that much and the network carries it on top of the earlier
    arrivals; otherwise its first period is pinned to zero.  The greedy walk is sequential inside a snapshot, so the
    loop runs over the arrival RANK and every step is one array operation over all snapshots.
    Like its twin -- which returns ``sorted(sessions, key=arrival)`` (stable) without the sessions whose
    ``remaining_time <= 0`` -- the table comes back in (snapshot, arrival) order without those sessions: the order is
    what breaks ties between equal rounding losses in ``diff_based_reallocation`` (post.py:214-218)."""
    out = table.copy()
    alive = np.flatnonzero(out.rem > 0)
    perm = alive[np.lexsort((out.arrival[alive], out.prob[alive]))]     # stable: by snapshot, then arrival, then list order
    live = np.flatnonzero((out.rem > 0) & (np.diff(out.seg) > 0))
    if len(live) == 0:
        return out.take(perm)
    order = live[np.lexsort((out.arrival[live], out.prob[live]))]        # stable: by snapshot, then arrival
    p = out.prob[order]
    start = np.r_[0, np.flatnonzero(np.diff(p)) + 1]
    rank = np.arange(len(order)) - np.repeat(start, np.diff(np.r_[start, len(order)]))
    volt = np.asarray(infrastructure.voltages, float)
    min_pilot = np.minimum(np.asarray(infrastructure.min_pilot, float), override)
    rates = np.zeros((out.B, out.N))
    raised = []
    for r in range(int(rank.max()) + 1):
        s = order[rank == r]                      # at most one session per snapshot
        b, i = out.prob[s], out.evse[s]
        want = min_pilot[i]
        rates[b, i] = want
        need = out.demand[s] * 1000.0 / volt[i] * 60.0 / period          # remaining_amp_periods
        ok = (need >= want) & _network_feasible(rates[b], infrastructure)
        first = out.seg[s]
        good, bad = s[ok], s[~ok]
        out.min_rates[first[ok]] = np.maximum(want[ok], out.min_rates[first[ok]])
        raised.append(good)
        rates[out.prob[bad], out.evse[bad]] = 0.0
        out.min_rates[first[~ok]] = 0.0
        out.max_rates[first[~ok]] = 0.0
    _reconcile(out, np.concatenate(raised))
    return out.take(perm)
