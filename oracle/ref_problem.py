"""ORACLE -- test infrastructure only.  Nothing under adacharge_amd/ may import it.

CPU restatement (numpy) of how the reference *states* its optimisation problem,
following /root/reference/adacharge/adaptive_charging_optimization.py line by
line.  The reference hands this statement to cvxpy (aco.py:315-318); here it is
emitted as an explicit conic QP

    minimise   1/2 x' P x + q' x            x = vec(rates), row-major (N, T)
    subject to A_ub x <= b_ub,   A_eq x = b_eq,
               || F_k x ||_2 <= g_k         (one 2-row F_k per SOC constraint and period)

so that any independent solver (oracle/ipm.py, scipy HiGHS) can solve it.
Inputs are duck-typed on the acnportal attribute names.

PARITY STATUS: cvxpy, ECOS and acnportal are un-pinned third-party
dependencies (reference setup.py:24) absent from /root/reference and from this
image; the reference's own tests hold no numeric vectors for this path, only
invariants plus scenarios with derivable closed forms.
This restatement is pinned by those (tests/test_oracle_kats.py): KAT-1 closed
form, KAT-2 infeasible pair, KAT-3 TOU, KAT-4 aggregate; per-EVSE values on
degenerate LPs are "parity unpinned".
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp


@dataclass
class RefProblem:
    N: int
    T: int
    P: sp.csr_matrix
    q: np.ndarray
    A_ub: sp.csr_matrix
    b_ub: np.ndarray
    A_eq: sp.csr_matrix
    b_eq: np.ndarray
    soc: List[Tuple[sp.csr_matrix, float]] = field(default_factory=list)
    lb: Optional[np.ndarray] = None  # bounds of every variable, flat (rates first, then extras)
    ub: Optional[np.ndarray] = None
    n_extra: int = 0                 # epigraph variables appended after vec(rates) (demand_charge)
    extra_obj: Optional[object] = None   # callable(rates) -> value of the epigraph term at its optimum

    @property
    def n(self):
        return self.N * self.T + self.n_extra

    def rates_of(self, x):
        return np.asarray(x, float).reshape(-1)[: self.N * self.T].reshape(self.N, self.T)

    def objective(self, rates: np.ndarray) -> float:
        """Objective at a rates matrix (extras eliminated at their optimum) or at a full vector."""
        x = np.asarray(rates, float).reshape(-1)
        if len(x) == self.N * self.T and self.n_extra:
            nr = self.N * self.T
            return float(0.5 * x @ (self.P[:nr, :nr] @ x) + self.q[:nr] @ x + self.extra_obj(x.reshape(self.N, self.T)))
        return float(0.5 * x @ (self.P @ x) + self.q @ x)


def _vidx(i, t, T):
    return i * T + t


def charging_rate_bounds(sessions, station_ids, N, T):
    """aco.py:61-75 -- later sessions overwrite their own window; ub<lb -> lb."""
    lb, ub = np.zeros((N, T)), np.zeros((N, T))
    for s in sessions:
        i = station_ids.index(s.station_id)
        lb[i, s.arrival_offset : s.arrival_offset + s.remaining_time] = s.min_rates
        ub[i, s.arrival_offset : s.arrival_offset + s.remaining_time] = s.max_rates
    ub[ub < lb] = lb[ub < lb]
    return lb, ub


def energy_rows(sessions, infrastructure, period, T):
    """aco.py:105-123 -- sum(rates[i, off:off+rem]) * (V_i * period / 1e3 / 60)
    <= (or ==) remaining_demand.  Coefficient order of operations as aco.py:114."""
    rows, cols, vals, rhs = [], [], [], []
    for r, s in enumerate(sessions):
        i = infrastructure.get_station_index(s.station_id)
        k = infrastructure.voltages[i] * period / 1e3 / 60
        for t in range(s.arrival_offset, s.arrival_offset + s.remaining_time):
            rows.append(r)
            cols.append(_vidx(i, t, T))
            vals.append(k)
        rhs.append(s.remaining_demand)
    n = infrastructure.num_stations * T
    return sp.csr_matrix((vals, (rows, cols)), shape=(len(sessions), n)), np.array(rhs, float)


def infrastructure_rows(infrastructure, constraint_type, T):
    """aco.py:145-178.  Returns (A_lin, b_lin, soc_list)."""
    N = infrastructure.num_stations
    n = N * T
    cm = infrastructure.constraint_matrix
    if cm is None or cm.shape == (0, 0):
        return sp.csr_matrix((0, n)), np.zeros(0), []
    if constraint_type == "SOC":
        if infrastructure.phases is None:
            raise ValueError("phases is required when using SOC infrastructure constraints.")
        phase_in_rad = np.deg2rad(infrastructure.phases)
        soc = []
        for j, v in enumerate(cm):
            a = np.stack([v * np.cos(phase_in_rad), v * np.sin(phase_in_rad)])
            nz = np.nonzero(v)[0]
            for t in range(T):
                rows = np.repeat([0, 1], len(nz))
                cols = np.tile(nz * T + t, 2)
                vals = np.r_[a[0, nz], a[1, nz]]
                soc.append(
                    (sp.csr_matrix((vals, (rows, cols)), shape=(2, n)),
                     float(infrastructure.constraint_limits[j]))
                )
        return sp.csr_matrix((0, n)), np.zeros(0), soc
    if constraint_type == "LINEAR":
        rows, cols, vals, rhs = [], [], [], []
        r = 0
        for j, v in enumerate(cm):
            av = np.abs(v)
            nz = np.nonzero(av)[0]
            for t in range(T):
                rows += [r] * len(nz)
                cols += list(nz * T + t)
                vals += list(av[nz])
                rhs.append(infrastructure.constraint_limits[j])
                r += 1
        return sp.csr_matrix((vals, (rows, cols)), shape=(r, n)), np.array(rhs, float), []
    raise ValueError(
        "Invalid infrastructure constraint type: {0}. Valid options are SOC or AFFINE.".format(
            constraint_type
        )
    )


def peak_rows(peak_limit, N, T):
    """aco.py:196-198 -- sum over EVSEs per period <= peak_limit (scalar or (T,))."""
    n = N * T
    if peak_limit is None:
        return sp.csr_matrix((0, n)), np.zeros(0)
    rows = np.repeat(np.arange(T), N)
    cols = (np.arange(N)[None, :] * T + np.arange(T)[:, None]).reshape(-1)
    A = sp.csr_matrix((np.ones(N * T), (rows, cols)), shape=(T, n))
    b = np.broadcast_to(np.asarray(peak_limit, float), (T,)).copy()
    return A, b


def objective_terms(objective_spec, infrastructure, interface, N, T):
    """aco.py:200-218 + 336-408.  ``objective_spec`` is a list of
    ``(name, coefficient, kwargs)``.  The reference MAXIMISES sum(coef * f);
    returned (P, q) are for the equivalent minimisation."""
    n = N * T
    q = np.zeros((N, T))
    p_diag = 0.0
    p_blocks = []  # (coef, v) -> P_t += 2 coef v v'
    volt = np.asarray(infrastructure.voltages, float)
    for name, coef, kwargs in objective_spec:
        if name == "quick_charge":  # aco.py:363-371
            c = np.array([(T - t) / T for t in range(T)])
            q -= coef * c[None, :]
        elif name == "equal_share":  # aco.py:374-375
            p_diag += 2.0 * coef
        elif name == "tou_energy_cost":  # aco.py:378-380, 336-360
            prices = np.asarray(interface.get_prices(T), float)
            q += coef * prices[None, :] * (volt[:, None] / 1e3) * (interface.period / 60)
        elif name == "total_energy":  # aco.py:383-384
            q -= coef * (volt[:, None] / 1e3) * (interface.period / 60) * np.ones((1, T))
        elif name in ("demand_charge", "peak"):  # aco.py:387-400 -- epigraph, handled by the caller
            continue
        elif name == "load_flattening":  # aco.py:403-408
            ext = kwargs.get("external_signal")
            ext = np.zeros(T) if ext is None else np.asarray(ext, float)
            v = volt / 1e3
            p_blocks.append((coef, v))
            q += 2.0 * coef * ext[None, :] * v[:, None]
        else:
            raise NotImplementedError(name)
    P = sp.identity(n, format="csr") * p_diag
    for coef, v in p_blocks:
        vv = sp.csr_matrix(np.outer(v, v) * 2.0 * coef)
        # x is (i, t) row-major: block for period t picks indices i*T + t
        sel = sp.csr_matrix(
            (np.ones(n), (np.arange(n), (np.arange(n) % T) * N + np.arange(n) // T)), shape=(n, n)
        )  # sel[i*T + t, t*N + i] = 1: maps the (t, i)-ordered vector to the (i, t)-ordered one
        P = P + sel @ sp.kron(sp.identity(T), vv, format="csr") @ sel.T
    return sp.csr_matrix(P), q.reshape(-1)


def build_reference_problem(
    sessions,
    infrastructure,
    interface,
    objective_spec,
    constraint_type="SOC",
    enforce_energy_equality=False,
    peak_limit=None,
) -> RefProblem:
    """aco.py:220-284 (build_problem)."""
    T = max(s.arrival_offset + s.remaining_time for s in sessions)  # aco.py:243-245
    N = len(infrastructure.station_ids)  # aco.py:246
    n = N * T
    lb, ub = charging_rate_bounds(sessions, list(infrastructure.station_ids), N, T)
    I = sp.identity(n, format="csr")
    A_e, b_e = energy_rows(sessions, infrastructure, interface.period, T)
    A_i, b_i, soc = infrastructure_rows(infrastructure, constraint_type, T)
    A_p, b_p = peak_rows(peak_limit, N, T)
    ub_blocks = [-I, I, A_i, A_p]
    ub_rhs = [-lb.reshape(-1), ub.reshape(-1), b_i, b_p]
    if enforce_energy_equality:
        A_eq, b_eq = A_e, b_e
    else:
        ub_blocks.append(A_e)
        ub_rhs.append(b_e)
        A_eq, b_eq = sp.csr_matrix((0, n)), np.zeros(0)
    P, q = objective_terms(objective_spec, infrastructure, interface, N, T)
    A_ub = sp.vstack(ub_blocks, format="csr")
    b_ub = np.concatenate(ub_rhs)
    lbf, ubf = lb.reshape(-1), ub.reshape(-1)
    # demand_charge (aco.py:387-400): maximise -dc * max(max_t agg_power_t, prev_peak kW[, baseline_peak])
    # = minimise dc * p with p >= v' r_t for all t, p >= floor.  One epigraph variable p appended.
    dc_terms = [(coef, kw) for name, coef, kw in objective_spec if name in ("demand_charge", "peak")]
    if dc_terms:
        v = np.asarray(infrastructure.voltages, float) / 1e3
        w = 0.0
        floor = -np.inf
        for (coef, kw), name in zip(dc_terms, [nm for nm, _, _ in objective_spec if nm in ("demand_charge", "peak")]):
            prev = interface.get_prev_peak() * infrastructure.voltages[0] / 1000
            base = kw.get("baseline_peak", 0)
            floor = max(floor, max(prev, base) if base > 0 else prev)
            w += coef * (interface.get_demand_charge() if name == "demand_charge" else -1.0)
        if w < 0:
            raise ValueError("peak enters the maximised objective with a positive sign: not concave")
        big = 1e7
        # bounds block stays [-I; I] over ALL variables so that oracle/ipm.polish can fix coordinates
        A_rest = sp.hstack([A_ub[2 * n :], sp.csr_matrix((A_ub.shape[0] - 2 * n, 1))], format="csr")
        rows = np.repeat(np.arange(T), N)
        cols = (np.arange(N)[None, :] * T + np.arange(T)[:, None]).reshape(-1)
        Apk = sp.hstack([sp.csr_matrix((np.tile(v, T), (rows, cols)), shape=(T, n)), -np.ones((T, 1))], format="csr")
        I1 = sp.identity(n + 1, format="csr")
        lbf = np.r_[lbf, floor]
        ubf = np.r_[ubf, big]
        A_ub = sp.vstack([-I1, I1, A_rest, Apk], format="csr")
        b_ub = np.concatenate([-lbf, ubf, b_ub[2 * n :], np.zeros(T)])
        A_eq = sp.hstack([A_eq, sp.csr_matrix((A_eq.shape[0], 1))], format="csr")
        soc = [(sp.hstack([F, sp.csr_matrix((2, 1))], format="csr"), g) for F, g in soc]
        P = sp.block_diag([P, sp.csr_matrix((1, 1))], format="csr")
        q = np.r_[q, w]
        extra = lambda r, v=v, w=w, floor=floor: w * max(float((v @ r).max()), floor)
        return RefProblem(N, T, P, q, A_ub, b_ub, A_eq, b_eq, soc, lbf, ubf, 1, extra)
    return RefProblem(N, T, P, q, A_ub, b_ub, A_eq, b_eq, soc, lbf, ubf)
