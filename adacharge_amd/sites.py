"""Synthetic charging sites and MPC-snapshot generators.

These are *inputs*, not part of the reference: the reference ships no site
definitions (they live in acnportal, absent here).  ``caltech54`` follows the
shape of acnportal's ``caltech_acn`` as recalled in SURVEY.md section 8d and is
labelled "Caltech-shaped" everywhere it is reported.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from .acn import InfrastructureInfo, SessionInfo


def caltech54(voltage: float = 208.0, max_pilot: float = 32.0) -> InfrastructureInfo:
    """54 EVSEs on a 3-phase delta: 26 on AB (+30 deg), 14 on BC (-90 deg),
    14 on CA (+150 deg).  8 constraint rows: two 80 A pods of 8 EVSEs with
    mixed phases, three secondary (per-phase-pair sum) rows at 416.667 A and
    three primary rows 1/4 (I_x - I_y) at 180.505 A; 178 non-zeros."""
    n_ab, n_bc, n_ca = 26, 14, 14
    n = n_ab + n_bc + n_ca
    phases = np.array([30.0] * n_ab + [-90.0] * n_bc + [150.0] * n_ca)
    ab = np.arange(0, n_ab)
    bc = np.arange(n_ab, n_ab + n_bc)
    ca = np.arange(n_ab + n_bc, n)
    cm = np.zeros((8, n))
    av_pod = np.r_[ab[:3], bc[:3], ca[:2]]
    cc_pod = np.r_[ab[3:6], bc[3:5], ca[2:5]]
    cm[0, av_pod] = 1.0
    cm[1, cc_pod] = 1.0
    cm[2, ab] = 1.0
    cm[3, bc] = 1.0
    cm[4, ca] = 1.0
    cm[5, ab], cm[5, ca] = 0.25, -0.25
    cm[6, bc], cm[6, ab] = 0.25, -0.25
    cm[7, ca], cm[7, bc] = 0.25, -0.25
    sec = 150e3 / 3 / 120
    pri = 150e3 / 3 / 277
    limits = np.array([80.0, 80.0, sec, sec, sec, pri, pri, pri])
    ids = ["AV-Pod", "CC-Pod", "Sec-A", "Sec-B", "Sec-C", "Pri-A", "Pri-B", "Pri-C"]
    return InfrastructureInfo(
        cm,
        limits,
        phases,
        np.full(n, voltage),
        constraint_ids=ids,
        station_ids=[f"CT-{i:03d}" for i in range(n)],
        max_pilot=np.full(n, max_pilot),
        min_pilot=np.full(n, 8.0),
        allowable_pilots=[np.r_[0.0, np.arange(8.0, max_pilot + 1)] for _ in range(n)],
        is_continuous=np.zeros(n, dtype=bool),
    )


def balanced_three_phase(
    n_evse: int,
    pods: int,
    load_fraction: float = 1.0 / 3.0,
    voltage: float = 208.0,
    max_pilot: float = 32.0,
    name: str = "SY",
) -> InfrastructureInfo:
    """Synthetic three-phase site in the same style (used for "jpl52" and
    "synth512", both *synthetic*): ``pods`` contiguous pod rows, three
    per-phase-pair sums and three primary difference rows, every limit set to
    ``load_fraction`` of that row's full-load magnitude."""
    per = [n_evse // 3 + (1 if r < n_evse % 3 else 0) for r in range(3)]
    phases = np.repeat([30.0, -90.0, 150.0], per)
    idx = np.split(np.arange(n_evse), np.cumsum(per)[:-1])
    rows = []
    for members in np.array_split(np.random.default_rng(7).permutation(n_evse), pods):
        r = np.zeros(n_evse)
        r[members] = 1.0
        rows.append(r)
    for g in idx:
        r = np.zeros(n_evse)
        r[g] = 1.0
        rows.append(r)
    for a, b in ((0, 2), (1, 0), (2, 1)):
        r = np.zeros(n_evse)
        r[idx[a]] = 0.25
        r[idx[b]] = -0.25
        rows.append(r)
    cm = np.array(rows)
    ph = np.deg2rad(phases)
    full = np.hypot((cm * np.cos(ph)) @ np.full(n_evse, max_pilot),
                    (cm * np.sin(ph)) @ np.full(n_evse, max_pilot))
    limits = load_fraction * full
    return InfrastructureInfo(
        cm,
        limits,
        phases,
        np.full(n_evse, voltage),
        constraint_ids=[f"{name}-c{j}" for j in range(len(rows))],
        station_ids=[f"{name}-{i:04d}" for i in range(n_evse)],
        max_pilot=np.full(n_evse, max_pilot),
        min_pilot=np.full(n_evse, 8.0),
        allowable_pilots=[np.r_[0.0, np.arange(8.0, max_pilot + 1)] for _ in range(n_evse)],
        is_continuous=np.zeros(n_evse, dtype=bool),
    )


def jpl52() -> InfrastructureInfo:
    """Synthetic 52-EVSE site (the real JPL topology is not available)."""
    return balanced_three_phase(52, pods=4, load_fraction=0.45, name="JP")


def synth512() -> InfrastructureInfo:
    """Synthetic 512-EVSE site, M = 12 pods + 3 + 3 = 18 rows, limits at 1/3."""
    return balanced_three_phase(512, pods=12, load_fraction=1.0 / 3.0, name="SY")


def random_sessions(
    infra: InfrastructureInfo,
    horizon: int,
    rng: np.random.Generator,
    min_sessions: int = 10,
    max_rate: float = 32.0,
    demand_range=(0.5, 20.0),
    min_rate_fraction: float = 0.0,
) -> List[SessionInfo]:
    """One MPC state snapshot ("Instance"): S ~ U{min..N}
    distinct EVSEs, arrival_offset 0, remaining_time ~ U{1..T} with one forced
    to T, remaining_demand ~ U(lo, hi) kWh, min_rates 0, max_rates ``max_rate``;
    ``min_rate_fraction`` of the sessions get ``min_rates[0] = 8``."""
    n = infra.num_stations
    s = int(rng.integers(min(min_sessions, n), n + 1))
    evses = rng.choice(n, size=s, replace=False)
    rem = rng.integers(1, horizon + 1, size=s)
    rem[int(rng.integers(0, s))] = horizon
    demand = rng.uniform(demand_range[0], demand_range[1], size=s)
    sessions = []
    for k in range(s):
        mins = np.zeros(int(rem[k]))
        if min_rate_fraction > 0 and rng.random() < min_rate_fraction:
            mins[0] = 8.0
        sessions.append(
            SessionInfo(
                infra.station_ids[int(evses[k])],
                f"s{k}",
                float(demand[k]),
                0.0,
                0,
                int(rem[k]),
                current_time=0,
                min_rates=mins,
                max_rates=max_rate,
            )
        )
    return sessions


def random_sessions_general(infra, horizon, rng, two_per_evse=False, min_rates=False, demand_scale=1.0,
                            period=5) -> List[SessionInfo]:
    """A harder snapshot than ``random_sessions``: delayed arrivals, optionally two sessions with disjoint
    windows on 40 % of the EVSEs (aco.py:62-73 lets later sessions overwrite their own window, t_aco.py:194-208),
    optionally minimum rates over a prefix of the window (t_aco.py:211-229) and stepped maximum rates."""
    T = horizon
    sessions = []
    n = infra.num_stations
    for i in rng.choice(n, size=int(rng.integers(n // 3, n + 1)), replace=False):
        sid = infra.station_ids[int(i)]
        k = float(infra.voltages[int(i)]) * period / 60 / 1e3
        if two_per_evse and rng.random() < 0.4 and T >= 8:
            cut = int(rng.integers(3, T - 3))
            spans = [(0, cut), (cut + int(rng.integers(0, 2)), T)]
        else:
            a = int(rng.integers(0, max(1, T // 3)))
            spans = [(a, int(rng.integers(a + 1, T + 1)))]
        for j, (a, d) in enumerate(spans):
            if d <= a:
                continue
            L = d - a
            mins = np.zeros(L)
            if min_rates and rng.random() < 0.3:
                mins[: int(rng.integers(1, L + 1))] = 6.0
            maxs = np.full(L, 32.0)
            if rng.random() < 0.2:
                maxs[int(rng.integers(0, L)):] = 16.0
            dem = float(rng.uniform(0.2, 1.0) * demand_scale * 32 * L * k)
            dem = max(dem, mins.sum() * k + 0.01)
            sessions.append(SessionInfo(sid, f"{sid}-{j}", dem, 0.0, a, d, current_time=0, min_rates=mins, max_rates=maxs))
    return sessions


def offline_day(infra: InfrastructureInfo, rng: np.random.Generator, horizon: int = 288, n_sessions: int = 150,
                period: int = 5) -> List[SessionInfo]:
    """A day-shaped offline instance (adacharge.py:234-276, t_int.py:350-403 shape; synthetic -- the reference's own
    day comes from the ACN-Data web API): ``n_sessions`` sessions over ``horizon`` periods on the site's EVSEs,
    arrivals clustered in the morning, stays of 1-9 hours, several consecutive sessions per EVSE, never overlapping
    on one EVSE, demands a fraction of what the stay could deliver."""
    n = infra.num_stations
    free_from = np.zeros(n, dtype=int)           # first period each EVSE is free again
    sessions = []
    arrivals = np.sort(np.clip(rng.normal(0.42 * horizon, 0.27 * horizon, size=4 * n_sessions), 0, horizon - 13).astype(int))
    for a in arrivals:
        if len(sessions) >= n_sessions:
            break
        cand = np.flatnonzero(free_from <= a)
        if len(cand) == 0:
            continue
        i = int(rng.choice(cand))
        stay = int(rng.integers(12, 97))
        d = min(horizon, a + stay)
        k = float(infra.voltages[i]) * period / 60 / 1e3
        dem = float(rng.uniform(0.15, 0.7) * 32.0 * (d - a) * k)
        sessions.append(SessionInfo(infra.station_ids[i], f"day-{len(sessions)}", dem, 0.0, int(a), int(d), current_time=0,
                                    min_rates=0.0, max_rates=32.0))
        free_from[i] = d + int(rng.integers(1, 6))
    return sessions


def eight_sites() -> List[InfrastructureInfo]:
    """The 8 sites of BASELINE.json configs[3] ("1024 demand scenarios x 8 sites"): the Caltech-shaped and the
    synthetic 52-EVSE site plus six synthetic three-phase sites of 30-64 EVSEs (all *synthetic*)."""
    out = [caltech54(), jpl52()]
    for k, (n, pods, frac) in enumerate([(30, 2, 0.5), (36, 3, 0.4), (42, 3, 0.45), (48, 4, 0.35), (60, 5, 0.4), (64, 4, 0.3)]):
        out.append(balanced_three_phase(n, pods=pods, load_fraction=frac, name=f"S{k}"))
    return out


def wide128() -> InfrastructureInfo:
    """Synthetic 128-EVSE three-phase site (general-shape kernel: N > 64)."""
    return balanced_three_phase(128, pods=6, load_fraction=0.4, name="WD")


def wide192() -> InfrastructureInfo:
    """Synthetic 192-EVSE three-phase site (large-site kernel at horizon 48; the widest shape the IPM oracle certifies
    in about a minute: tests/golden/wide.npz::wide192_t48_soc)."""
    return balanced_three_phase(192, pods=8, load_fraction=0.35, name="W2")


def snapshot_batch(
    infra: InfrastructureInfo,
    horizon: int,
    batch: int,
    seed: int = 20240,
    **kw,
) -> List[List[SessionInfo]]:
    """``batch`` independent snapshots from one SeedSequence spawned B ways."""
    children = np.random.SeedSequence(seed).spawn(batch)
    return [random_sessions(infra, horizon, np.random.default_rng(c), **kw) for c in children]


def snapshot_table(infra: InfrastructureInfo, horizon: int, batch: int, seed: int = 20240, min_sessions: int = 10,
                   max_rate: float = 32.0, demand_range=(0.5, 20.0)):
    """``batch`` MPC snapshots of the same distribution as ``random_sessions`` as ONE ``SessionTable`` -- arrays from
    the start, no SessionInfo objects (the array-native input of ``schedule_batch`` / ``solve_table``)."""
    from .session_table import SessionTable

    rng = np.random.default_rng(seed)
    n = infra.num_stations
    cnt = rng.integers(min(min_sessions, n), n + 1, size=batch)
    # distinct EVSEs per snapshot: the first cnt[b] entries of a random permutation
    perm = np.argsort(rng.random((batch, n)), axis=1)
    keep = np.arange(n)[None, :] < cnt[:, None]
    prob = np.repeat(np.arange(batch), cnt)
    evse = perm[keep]
    S = len(prob)
    rem = rng.integers(1, horizon + 1, size=S)
    first = np.r_[0, np.cumsum(cnt)[:-1]]
    rem[first + rng.integers(0, cnt)] = horizon          # one session per snapshot spans the horizon
    demand = rng.uniform(demand_range[0], demand_range[1], size=S)
    seg = np.r_[0, np.cumsum(rem)]
    return SessionTable(batch, n, prob.astype(np.int64), evse.astype(np.int64), np.zeros(S, np.int64), rem.astype(np.int64),
                        demand, np.zeros(S, np.int64), seg.astype(np.int64), np.zeros(int(seg[-1])), np.full(int(seg[-1]), float(max_rate)),
                        [f"s{k}" for k in range(S)])


def demand_scenarios(
    sessions: Sequence[SessionInfo], n_scenarios: int, rng: np.random.Generator, sigma: float = 0.25
) -> List[List[SessionInfo]]:
    """Stochastic-MPC scenarios (config 4): same windows, remaining demand
    scaled by lognormal(0, sigma) per session per scenario."""
    out = []
    for _ in range(n_scenarios):
        scen = []
        for s in sessions:
            f = float(rng.lognormal(0.0, sigma))
            scen.append(
                SessionInfo(
                    s.station_id,
                    s.session_id,
                    s.energy_delivered + s.remaining_demand * f,
                    s.energy_delivered,
                    s.arrival,
                    s.departure,
                    current_time=s.current_time,
                    min_rates=s.min_rates.copy(),
                    max_rates=s.max_rates.copy(),
                )
            )
        out.append(scen)
    return out
