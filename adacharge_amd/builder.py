"""Direct problem builder: sessions + infrastructure -> structured QP batch.

Replaces the reference's cvxpy expression-tree construction
(adaptive_charging_optimization.py:45-284) with plain numpy arrays that the
C-ABI (include/acn_qp.h) consumes.  Nothing here forms the explicit (P, q, A,
l, u): the constraint structure is kept symbolic --

  * per-variable bounds  lb <= r <= ub                       (aco.py:61-79)
  * per-session energy   sum_{t in window} r[i,t] <= cap     (aco.py:105-123)
        cap = remaining_demand / (V_i * period / 1e3 / 60)   [A-periods]
  * shared site rows     G r[:,t]  in  box / disc / peak     (aco.py:145-198)
  * objective            1/2 r'Pr + q'r,  P = pdiag I + lf (I_T (x) v v')
                                                             (aco.py:200-218, 336-408)

so the kernel can use a matrix-free ADMM whose only linear-algebra object is
the tiny site matrix G (M_g x N), shared by every problem of the batch.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

CONE_LINEAR = 0
CONE_SOC = 1


@dataclass
class SiteData:
    """Per-site data shared by every QP of a batch (uploaded once per handle).

    ``G`` rows: LINEAR -> |C| (M rows); SOC -> [C cos(phi); C sin(phi)] (2M
    rows, row j pairs with row j+M); plus a trailing all-ones row when the
    batch carries a peak limit (aco.py:196-198).  ``lam, Q`` is the
    eigen-decomposition of G G' and ``Ghat = Q' G``; with them
    (a I + rho G'G)^-1 = (I - Ghat' diag(rho / (a + rho lam)) Ghat) / a  for any
    per-problem (a, rho), so no factorisation ever happens on the device."""

    N: int
    M: int
    cone: int
    has_peak: bool
    has_flat: bool
    has_max: bool
    G: np.ndarray
    limits: np.ndarray
    lam: np.ndarray
    Q: np.ndarray
    Ghat: np.ndarray
    volt: np.ndarray

    @property
    def Mg(self) -> int:
        return self.G.shape[0]

    @property
    def flat_row(self) -> int:
        """Index of the aggregate-power row v = voltages/1e3 (load_flattening), if present."""
        return self.Mg - 1 - (1 if self.has_peak else 0) - (1 if self.has_max else 0)

    @property
    def max_row(self) -> int:
        """Index of the aggregate-power row used by demand_charge / peak, if present."""
        return self.Mg - 1 - (1 if self.has_peak else 0)


def make_site(infrastructure, constraint_type: str = "SOC", with_peak: bool = False, with_flat: bool = False,
              with_max: bool = False) -> SiteData:
    N = len(infrastructure.station_ids)
    cm = infrastructure.constraint_matrix
    if cm is None or cm.shape == (0, 0):  # aco.py:145-149
        rows = np.zeros((0, N))
        limits = np.zeros(0)
        M = 0
        cone = CONE_LINEAR if constraint_type != "SOC" else CONE_SOC
        if constraint_type not in ("SOC", "LINEAR"):
            _bad_constraint_type(constraint_type)
    elif constraint_type == "SOC":  # aco.py:151-164
        if infrastructure.phases is None:
            raise ValueError("phases is required when using SOC infrastructure constraints.")
        ph = np.deg2rad(np.asarray(infrastructure.phases, float))
        cm = np.asarray(cm, float)
        rows = np.vstack([cm * np.cos(ph), cm * np.sin(ph)])
        limits = np.asarray(infrastructure.constraint_limits, float).copy()
        M = cm.shape[0]
        cone = CONE_SOC
    elif constraint_type == "LINEAR":  # aco.py:165-172
        rows = np.abs(np.asarray(cm, float))
        limits = np.asarray(infrastructure.constraint_limits, float).copy()
        M = rows.shape[0]
        cone = CONE_LINEAR
    else:
        _bad_constraint_type(constraint_type)
    if with_flat:  # aggregate power in kW: charging_power / aggregate_power, aco.py:336-344
        rows = np.vstack([rows, np.asarray(infrastructure.voltages, float)[None, :] / 1e3])
    if with_max:  # same aggregate-power row, used by the demand-charge prox (aco.py:387-400)
        rows = np.vstack([rows, np.asarray(infrastructure.voltages, float)[None, :] / 1e3])
    if with_peak:
        rows = np.vstack([rows, np.ones((1, N))])
    G = np.ascontiguousarray(rows, dtype=np.float64)
    if G.shape[0]:
        lam, Q = np.linalg.eigh(G @ G.T)
        lam = np.maximum(lam, 0.0)
        lam[lam < 1e-12 * max(1.0, lam.max())] = 0.0
        Ghat = Q.T @ G
        Ghat[lam == 0.0] = 0.0
    else:
        lam, Q, Ghat = np.zeros(0), np.zeros((0, 0)), np.zeros((0, N))
    return SiteData(
        N, M, cone, bool(with_peak), bool(with_flat), bool(with_max), G, limits, lam,
        np.ascontiguousarray(Q), np.ascontiguousarray(Ghat),
        np.asarray(infrastructure.voltages, float).copy(),
    )


def _bad_constraint_type(constraint_type):
    # same text as the reference, including its "AFFINE" wording (aco.py:174-178)
    raise ValueError(
        "Invalid infrastructure constraint type: {0}. Valid options are SOC or AFFINE.".format(
            constraint_type
        )
    )


@dataclass
class ProblemBatch:
    """B structured QPs over one site, padded to a common horizon ``Tm``."""

    site: SiteData
    B: int
    Tm: int
    K: int                      # max sessions per EVSE in this batch
    T: np.ndarray               # (B,)  int32   own horizon (aco.py:243-245)
    lb: np.ndarray              # (B, N, Tm) f64, 0 outside windows and for t >= T[b]
    ub: np.ndarray              # (B, N, Tm) f64
    q: np.ndarray               # (B, N, Tm) f64  linear cost of the *minimisation*
    pdiag: np.ndarray           # (B,) f64   P = pdiag I          (equal_share)
    lf: np.ndarray              # (B,) f64   P_t += lf v v'       (load_flattening), v = volt/1e3
    s_off: np.ndarray           # (B, K, N) int32
    s_len: np.ndarray           # (B, K, N) int32  (0 = no session in this slot)
    s_cap: np.ndarray           # (B, K, N) f64    energy cap in A-periods
    s_eq: np.ndarray            # (B,) uint8       1 = energy rows are equalities
    peak: Optional[np.ndarray]  # (B, Tm) f64 or None; +inf = no limit in that period
    dc: np.ndarray = None       # (B,) f64   weight of max(max_t v'r_t, dfloor)   (demand_charge / peak)
    dfloor: np.ndarray = None   # (B,) f64   previous / baseline peak in kW
    const: np.ndarray = None    # (B,) f64 constant dropped from the objective
    presolve_status: np.ndarray = None  # (B,) int32, nonzero = infeasible before any solve

    @property
    def N(self) -> int:
        return self.site.N

    _PER_PROBLEM = ("T", "lb", "ub", "q", "pdiag", "lf", "s_off", "s_len", "s_cap", "s_eq", "peak", "dc", "dfloor",
                    "const", "presolve_status")

    def subset(self, index) -> "ProblemBatch":
        """The problems ``index`` (slice or integer array) of this batch as a batch of their own (views when sliced)."""
        import copy

        out = copy.copy(self)
        for name in self._PER_PROBLEM:
            a = getattr(self, name)
            setattr(out, name, None if a is None else a[index])
        out.B = len(out.T)
        return out

    @staticmethod
    def concatenate(batches: Sequence["ProblemBatch"]) -> "ProblemBatch":
        """Batches of one site and one shape (Tm, K) as a single batch."""
        import copy

        first = batches[0]
        if any(b.Tm != first.Tm or b.K != first.K or b.site is not first.site for b in batches):
            raise ValueError("concatenate needs batches of one site and one shape")
        out = copy.copy(first)
        for name in ProblemBatch._PER_PROBLEM:
            parts = [getattr(b, name) for b in batches]
            setattr(out, name, None if parts[0] is None else np.concatenate(parts))
        out.B = len(out.T)
        return out


def scenario_batch(base: ProblemBatch, demand_factor: np.ndarray, problem: int = 0) -> ProblemBatch:
    """Stochastic-MPC scenarios of ONE problem of ``base`` (BASELINE.json configs[3]): same session windows,
    bounds and objective, the remaining demand of every session scaled per scenario.  ``demand_factor`` is
    (S,) -- one factor per scenario -- or (S, K, N) -- one per session slot.  Pure array work (no Python loop
    over sessions): the energy caps are the only per-scenario data, everything else is broadcast."""
    f = np.asarray(demand_factor, float)
    S = f.shape[0]
    if f.ndim == 1:
        f = f[:, None, None]
    if f.shape[1:] not in ((1, 1), base.s_cap.shape[1:]):
        raise ValueError("demand_factor must have shape (S,) or (S, K, N)")
    p = slice(problem, problem + 1)
    rep = lambda a: None if a is None else np.ascontiguousarray(np.broadcast_to(a[p], (S,) + a.shape[1:]))
    return ProblemBatch(
        base.site, S, base.Tm, base.K, rep(base.T), rep(base.lb), rep(base.ub), rep(base.q), rep(base.pdiag),
        rep(base.lf), rep(base.s_off), rep(base.s_len), np.ascontiguousarray(base.s_cap[p] * f), rep(base.s_eq),
        rep(base.peak), rep(base.dc), rep(base.dfloor), rep(base.const), rep(base.presolve_status),
    )


def _objective_needs_max(objective) -> bool:
    from .adaptive_charging_optimization import demand_charge, peak

    return any(c.function in (demand_charge, peak) and c.coefficient != 0 for c in objective)


def _objective_needs_flat(objective) -> bool:
    from .adaptive_charging_optimization import load_flattening

    return any(c.function is load_flattening and c.coefficient != 0 for c in objective)


def objective_terms(objective, infrastructure, interface, N, T, prev_peak=0):
    """Evaluate the objective list on a symbolic rates handle; see
    adaptive_charging_optimization.py in this package for the descriptor
    algebra.  Returns (q (N,T), pdiag, lf, const)."""
    from .adaptive_charging_optimization import Rates, QuadObjective

    rates = Rates((N, T))
    total = QuadObjective.zero((N, T))
    for component in objective:
        kwargs = dict(prev_peak=prev_peak)
        kwargs.update(component.kwargs)
        term = component.function(rates, infrastructure, interface, **kwargs)
        total = total + component.coefficient * term
    if total.sq < 0 or total.flat < 0:
        raise ValueError(
            "Objective is not concave: the maximised objective has a positive "
            "quadratic coefficient (cvxpy would reject it as non-DCP)."
        )
    # epigraph terms: the maximised objective holds sum_e w_e * max(max_t agg_power_t, floor_e)
    dcw, dfloor = 0.0, 0.0
    if total.epigraph:
        dcw = -sum(w for w, _ in total.epigraph)
        if any(w > 0 for w, _ in total.epigraph):
            raise ValueError("peak enters the maximised objective with a positive weight: not concave")
        dfloor = max(spec["floor"] for _, spec in total.epigraph)
    # reference maximises `total`; we minimise its negation
    return -total.lin, 2.0 * total.sq, 2.0 * total.flat, -total.const, dcw, dfloor


def build_batch(
    session_lists: Sequence[Sequence],
    infrastructure,
    interface,
    objective,
    constraint_type: str = "SOC",
    enforce_energy_equality: bool = False,
    peak_limits: Optional[Sequence] = None,
    prev_peak=0,
    site: Optional[SiteData] = None,
    alloc=None,
) -> ProblemBatch:
    """One structured QP per entry of ``session_lists`` (all on one site).  ``alloc(shape, dtype)`` allocates the
    arrays the C ABI reads (default ``np.zeros``; ``backend.pinned_empty`` puts them in pinned host memory so the
    library copies them by DMA without a staging hop).  One pass over the session objects
    (``SessionTable.from_sessions``), then array work only (``build_batch_from_table``)."""
    from .session_table import SessionTable

    table = SessionTable.from_sessions(session_lists, infrastructure)
    return build_batch_from_table(table, infrastructure, interface, objective, constraint_type, enforce_energy_equality,
                                  peak_limits, prev_peak, site, alloc)


@dataclass
class TablePlan:
    """What ``acnqp_solve_table`` takes (include/acn_qp.h: acnqp_table): the sessions of a ``SessionTable`` grouped by
    problem with their EVSE slots and energy caps, the linear cost once per distinct horizon, the per-problem scalars --
    everything ``build_batch_from_table`` computes EXCEPT the dense (B, N, Tm) arrays, which the library forms on the
    device.  ``expand()`` forms them on the host (the ``ProblemBatch`` the other entry points take): the two paths
    describe the same problems bit for bit (tests/test_table_entry.py)."""
    site: SiteData
    B: int
    Tm: int
    K: int
    T: np.ndarray          # (B,) int32 horizon
    q_index: np.ndarray    # (B,) int32 row of q_table
    q_table: np.ndarray    # (H, N, Tm)
    pdiag: np.ndarray
    lf: np.ndarray
    dc: np.ndarray
    dfloor: np.ndarray
    const: np.ndarray
    s_eq: np.ndarray
    peak: Optional[np.ndarray]
    presolve_status: np.ndarray
    sess_seg: np.ndarray   # (B + 1,) int32
    s_evse: np.ndarray     # (S,) int32, sessions grouped by problem
    s_slot: np.ndarray
    s_off: np.ndarray
    s_len: np.ndarray
    s_cap: np.ndarray      # (S,) float64, A-periods
    rate_seg: np.ndarray   # (S + 1,) int32
    min_rates: np.ndarray
    max_rates: np.ndarray

    @property
    def N(self) -> int:
        return self.site.N

    @property
    def S(self) -> int:
        return len(self.s_evse)

    def expand(self, alloc=None) -> "ProblemBatch":
        """The dense ``ProblemBatch`` of this plan (numpy twin of acn_qp_api.hip::table_expand_kernel)."""
        if alloc is None:
            alloc = lambda shape, dtype=np.float64: np.zeros(shape, dtype)
        B, N, Tm, K = self.B, self.N, self.Tm, self.K
        lb, ub, q = alloc((B, N, Tm)), alloc((B, N, Tm)), alloc((B, N, Tm))
        prob = np.repeat(np.arange(B), np.diff(self.sess_seg))
        rlen = np.diff(self.rate_seg)
        owner = np.repeat(np.arange(self.S), rlen)
        j = np.arange(len(owner)) - np.repeat(self.rate_seg[:-1], rlen)
        flat = (prob[owner] * N + self.s_evse[owner]) * Tm + self.s_off[owner] + j
        lb.reshape(-1)[flat] = self.min_rates     # aco.py:62-73
        ub.reshape(-1)[flat] = self.max_rates
        np.maximum(ub, lb, out=ub)                # aco.py:75
        q[:] = self.q_table[self.q_index]
        s_off, s_len, s_cap = alloc((B, K, N), np.int32), alloc((B, K, N), np.int32), alloc((B, K, N))
        live = np.flatnonzero(self.s_len > 0)
        slot = (prob[live] * K + self.s_slot[live]) * N + self.s_evse[live]
        s_off.reshape(-1)[slot] = self.s_off[live]
        s_len.reshape(-1)[slot] = self.s_len[live]
        s_cap.reshape(-1)[slot] = self.s_cap[live]
        Ts, pdiag, lf, dc, dfl, s_eq = alloc(B, np.int32), alloc(B), alloc(B), alloc(B), alloc(B), alloc(B, np.uint8)
        Ts[:], pdiag[:], lf[:], dc[:], dfl[:], s_eq[:] = self.T, self.pdiag, self.lf, self.dc, self.dfloor, self.s_eq
        peak = None
        if self.peak is not None:
            peak = alloc((B, Tm))
            peak[:] = self.peak
        return ProblemBatch(self.site, B, Tm, K, Ts, lb, ub, q, pdiag, lf, s_off, s_len, s_cap, s_eq, peak, dc, dfl,
                            self.const.copy(), self.presolve_status.copy())


def plan_from_table(
    table,
    infrastructure,
    interface,
    objective,
    constraint_type: str = "SOC",
    enforce_energy_equality: bool = False,
    peak_limits: Optional[Sequence] = None,
    prev_peak=0,
    site: Optional[SiteData] = None,
) -> TablePlan:
    """Everything the reference's statement needs from a ``SessionTable`` (every snapshot needs at least one session)
    with no Python loop over sessions and no (B, N, Tm) array: the sessions grouped by snapshot, their slot = rank among
    their EVSE's sessions and energy cap in A-periods (aco.py:105-123), the objective once per distinct horizon
    (aco.py:200-218, 243-245)."""
    B, N = table.B, len(infrastructure.station_ids)
    if peak_limits is None:
        peak_limits = [None] * B
    any_peak = any(p is not None for p in peak_limits)
    need_flat = _objective_needs_flat(objective)
    need_max = _objective_needs_max(objective)
    if site is None:
        site = make_site(infrastructure, constraint_type, with_peak=any_peak, with_flat=need_flat, with_max=need_max)
    elif need_max and not site.has_max:
        raise ValueError("site was built without the demand-charge row but the objective uses demand_charge / peak")
    elif need_flat and not site.has_flat:
        raise ValueError("site was built without the aggregate-power row but the objective uses load_flattening")
    elif any_peak and not site.has_peak:
        raise ValueError("site was built without a peak row but a peak_limit was given")
    S = table.S
    per_snapshot = np.bincount(table.prob, minlength=B) if S and table.prob.min() >= 0 else np.zeros(0, np.int64)
    if S == 0 or len(per_snapshot) != B or (per_snapshot == 0).any():
        raise ValueError("every snapshot of a batch needs at least one session (aco.py:310-311 handles the empty case)")
    rlen_in = np.diff(table.seg)
    if np.any(rlen_in != np.maximum(table.rem, 0)):
        raise ValueError("min_rates / max_rates must have one entry per remaining period (aco.py:68, 73)")
    # ---- sessions grouped by snapshot (stable: the order inside a snapshot is the caller's) ----------------------------
    if np.any(np.diff(table.prob) < 0):
        table = table.take(np.argsort(table.prob, kind="stable"))
    prob, evse, off, rem = table.prob, table.evse, table.off, table.rem
    end = np.zeros(B, dtype=np.int64)
    np.maximum.at(end, prob, off + rem)            # aco.py:243-245
    Tm = int(end.max())
    volt = np.asarray(infrastructure.voltages, float)
    kwh_per_amp_period = volt * interface.period / 1e3 / 60   # aco.py:114
    presolve = np.zeros(B, dtype=np.int32)
    # ---- energy rows (aco.py:105-123): slot k = rank of the session among its EVSE's sessions of that snapshot ---
    live = np.flatnonzero(rem > 0)
    dead = np.flatnonzero(rem <= 0)
    if len(dead):   # empty window: 0 <= (==) remaining_demand
        bad = (table.demand[dead] < 0) | (enforce_energy_equality & (table.demand[dead] != 0))
        presolve[prob[dead[bad]]] = 1
    key = prob[live] * N + evse[live]
    s_slot = np.zeros(S, dtype=np.int32)
    if len(key) == 0 or np.bincount(key, minlength=B * N).max() <= 1:
        K = 1   # one session per (snapshot, EVSE) at most -- the usual case: every slot is 0, no sort (13 + 6 of 54 ms)
    else:
        order = np.argsort(key, kind="stable")
        ks = key[order]
        first = np.r_[0, np.flatnonzero(np.diff(ks)) + 1]
        counts = np.diff(np.r_[first, len(ks)])
        rank = np.arange(len(ks)) - np.repeat(first, counts)
        K = int(rank.max()) + 1
        # sessions sharing an EVSE must not overlap in time
        lo = live[order]
        o2 = np.lexsort((off[lo], ks))                 # by EVSE group, then window start
        so, sr, sk = off[lo][o2], rem[lo][o2], ks[o2]
        clash = (sk[1:] == sk[:-1]) & (so[1:] < so[:-1] + sr[:-1])
        if clash.any():
            w = lo[o2][1:][clash][0]
            raise ValueError(
                f"sessions on EVSE {infrastructure.station_ids[int(evse[w])]} overlap in time; the structured "
                "builder needs disjoint session windows per EVSE"
            )
        s_slot[live[order]] = rank
    s_cap = table.demand / kwh_per_amp_period[evse]
    sess_seg = np.zeros(B + 1, dtype=np.int32)
    sess_seg[1:] = np.cumsum(per_snapshot)
    # ---- objective: depends on the problem only through its horizon ----------------------------------------------
    horizons = np.unique(end)
    q_table = np.zeros((len(horizons), N, Tm))
    q_index = np.searchsorted(horizons, end).astype(np.int32)
    pdiag, lf, const, dc, dfloor = (np.zeros(B) for _ in range(5))
    for hidx, T in enumerate(horizons):
        T = int(T)
        qb, pd, lfc, c0, dcw, dfl = objective_terms(objective, infrastructure, interface, N, T, prev_peak)
        q_table[hidx, :, :T] = qb
        sel = q_index == hidx
        pdiag[sel], lf[sel], const[sel], dc[sel], dfloor[sel] = pd, lfc, c0, dcw, dfl
    peak = None
    if site.has_peak:
        peak = np.full((B, Tm), np.inf)
        for b in range(B):   # aco.py:196-198
            if peak_limits[b] is not None:
                peak[b, : end[b]] = np.broadcast_to(np.asarray(peak_limits[b], float), (int(end[b]),))
    s_eq = np.full(B, 1 if enforce_energy_equality else 0, dtype=np.uint8)
    i32 = lambda a: np.ascontiguousarray(a, np.int32)
    return TablePlan(site, B, Tm, K, i32(end), q_index, q_table, pdiag, lf, dc, dfloor, const, s_eq, peak, presolve,
                     sess_seg, i32(evse), s_slot, i32(off), i32(np.maximum(rem, 0)), np.ascontiguousarray(s_cap, np.float64),
                     i32(table.seg), np.ascontiguousarray(table.min_rates, np.float64), np.ascontiguousarray(table.max_rates, np.float64))


def build_batch_from_table(
    table,
    infrastructure,
    interface,
    objective,
    constraint_type: str = "SOC",
    enforce_energy_equality: bool = False,
    peak_limits: Optional[Sequence] = None,
    prev_peak=0,
    site: Optional[SiteData] = None,
    alloc=None,
) -> ProblemBatch:
    """The structured batch of a ``SessionTable`` (every snapshot needs at least one session) with no Python loop
    over sessions: ``plan_from_table`` (energy rows by group ranking, aco.py:105-123; the objective once per distinct
    horizon, aco.py:200-218, 243-245), then the dense arrays by ragged scatter (aco.py:61-79)."""
    return plan_from_table(table, infrastructure, interface, objective, constraint_type, enforce_energy_equality,
                           peak_limits, prev_peak, site).expand(alloc)
