"""Shared test helpers: golden-fixture loading and the four invariants the
reference's own test base class asserts (t_aco.py:50-83)."""
import os

import numpy as np

from adacharge_amd import sites
from adacharge_amd.acn import Interface, SessionInfo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "caltech54_T12.npz")


def load_golden():
    return np.load(GOLDEN, allow_pickle=False)


def golden_case(g, key):
    """Rebuild the SessionInfo list of fixture ``key``; returns
    (sessions, meta dict, expected dict)."""
    st = g[f"{key}_station"]
    sessions = []
    for k in range(len(st)):
        rem = int(g[f"{key}_departure"][k] - g[f"{key}_arrival"][k])
        mins = np.zeros(rem)
        mins[0] = g[f"{key}_min0"][k]
        sessions.append(
            SessionInfo(
                str(st[k]), f"s{k}", float(g[f"{key}_demand"][k]), 0.0,
                int(g[f"{key}_arrival"][k]), int(g[f"{key}_departure"][k]),
                current_time=0, min_rates=mins, max_rates=float(g[f"{key}_maxr"][k]),
            )
        )
    m = g[f"{key}_meta"]
    meta = dict(seed=int(m[0]), ct="SOC" if m[1] else "LINEAR", es=float(m[2]), eq=bool(m[3]), T=int(m[4]))
    exp = {k: g[f"{key}_{k}"] for k in ("rates", "obj", "agg") if f"{key}_{k}" in g}
    return sessions, meta, exp


def caltech_interface():
    infra = sites.caltech54()
    return infra, Interface({"infrastructure_info": infra, "period": 5})


# ---- the reference's invariants (t_aco.py:50-83) ------------------------------------
def assert_rates_below_max(rates, max_rate, tol=1e-3):
    assert (rates <= max_rate + tol).all()


def energy_delivered(rates, sessions, infrastructure, period):
    out = []
    for s in sessions:
        i = infrastructure.station_ids.index(s.station_id)
        e = rates[i, s.arrival_offset : s.arrival_offset + s.remaining_time].sum()
        out.append(e * infrastructure.voltages[i] * period / 1e3 / 60)
    return np.array(out)


def assert_energy_demands_met(rates, sessions, infrastructure, period):
    delivered = energy_delivered(rates, sessions, infrastructure, period)
    expected = np.array([s.remaining_demand for s in sessions])
    assert np.allclose(delivered, expected, atol=1e-4, rtol=1e-4)


def assert_no_charging_when_unplugged(rates, sessions, infrastructure):
    mask = np.ones(rates.shape, dtype=bool)
    for s in sessions:
        i = infrastructure.station_ids.index(s.station_id)
        mask[i, s.arrival_offset : s.arrival_offset + s.remaining_time] = False
    assert np.allclose(rates[mask], 0)


def infrastructure_violation(rates, infrastructure):
    """max over rows/periods of |[v cos; v sin] @ rates| - limit (utils.py:5-12 formula)."""
    ph = np.deg2rad(infrastructure.phases)
    cm = infrastructure.constraint_matrix
    mag = np.hypot((cm * np.cos(ph)) @ rates, (cm * np.sin(ph)) @ rates)
    return float((mag - np.asarray(infrastructure.constraint_limits)[:, None]).max())


def assert_infrastructure_satisfied(rates, infrastructure, tol=1e-3):
    assert infrastructure_violation(rates, infrastructure) <= tol


# ---- second fixture file: kernel variants and constraint families beyond caltech54 / T = 12 / inequality ----------
WIDE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wide.npz")


def load_wide():
    return np.load(WIDE, allow_pickle=False)


STALLED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stalled.npz")


def load_stalled():
    """tools/make_golden_stalled.py: certified optima of instances that end the adaptive first pass on a plateau."""
    return np.load(STALLED, allow_pickle=False)


def wide_case(g, name):
    """Rebuild fixture ``name`` of wide.npz / stalled.npz (tools/make_golden_wide.py, make_golden_stalled.py): returns
    (sessions, infrastructure, interface, meta dict, peak_limit, expected dict)."""
    site_tag = str(g[f"{name}_site"])
    infra = sites.eight_sites()[int(site_tag.split(":")[1])] if site_tag.startswith("eight:") else getattr(sites, site_tag)()
    iface = Interface({"infrastructure_info": infra, "period": 5})
    st, arr, dep = g[f"{name}_station"], g[f"{name}_arrival"], g[f"{name}_departure"]
    minr, maxr = g[f"{name}_minr"], g[f"{name}_maxr"]
    sessions, o = [], 0
    for k in range(len(st)):
        L = int(dep[k] - arr[k])
        sessions.append(SessionInfo(infra.station_ids[int(st[k])], f"s{k}", float(g[f"{name}_demand"][k]), 0.0,
                                    int(arr[k]), int(dep[k]), current_time=0,
                                    min_rates=minr[o:o + L].copy(), max_rates=maxr[o:o + L].copy()))
        o += L
    m = g[f"{name}_meta"]
    meta = dict(T=int(m[0]), ct="SOC" if m[1] else "LINEAR", eq=bool(m[2]), es=float(m[3]), seed=int(m[4]))
    pk = g[f"{name}_peak"]
    peak = None if np.isnan(pk[0]) else (float(pk[0]) if len(pk) == 1 else pk.copy())
    exp = dict(rates=g[f"{name}_rates"], obj=float(g[f"{name}_obj"]))
    return sessions, infra, iface, meta, peak, exp


# ---- a minimal closed-loop plant (shape of the reference's integration tests, t_int.py:16-48, without acnsim) --------
def closed_loop_fleet(infra, rng, n_evs=45, t_span=24, stay=(8, 13), fill=(0.5, 0.95), period=5):
    """EV records for a congested closed loop: arrivals over ``t_span`` periods, stays of ``stay`` periods, requests a
    ``fill`` fraction of what the stay could deliver at 32 A."""
    evs = []
    k = infra.voltages[0] * period / 1e3 / 60
    for n, i in enumerate(rng.choice(infra.num_stations, size=n_evs, replace=False)):
        arr = int(rng.integers(0, t_span))
        dur = int(rng.integers(stay[0], stay[1]))
        evs.append(dict(station=infra.station_ids[int(i)], sid=f"ev{n}", arrival=arr, departure=arr + dur,
                        requested=float(rng.uniform(*fill) * 32 * dur * k), delivered=0.0))
    return evs


def closed_loop_sessions(evs, t):
    """SessionInfo list of the EVs plugged in at period ``t`` that still need energy."""
    return [SessionInfo(e["station"], e["sid"], e["requested"], e["delivered"], e["arrival"], e["departure"],
                        current_time=t, max_rates=32.0)
            for e in evs if e["arrival"] <= t < e["departure"] and e["requested"] - e["delivered"] > 1e-9]


def closed_loop_apply(evs, t, first_period_rates, infra, period=5):
    k = infra.voltages[0] * period / 1e3 / 60
    for e in evs:
        if e["arrival"] <= t < e["departure"]:
            r = float(first_period_rates[infra.get_station_index(e["station"])])
            e["delivered"] = min(e["requested"], e["delivered"] + r * k)


PROX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prox.npz")


def load_prox():
    """tools/make_golden_prox.py: certified optima of problems with a load_flattening / demand_charge term at the shapes
    the large-site and the long-horizon kernel serve."""
    return np.load(PROX, allow_pickle=False)


def prox_case(g, name):
    """Rebuild fixture ``name`` of prox.npz: (sessions, infrastructure, interface, objective list, spec for
    oracle/ref_problem.py, meta dict, expected dict)."""
    from adacharge_amd import ObjectiveComponent, demand_charge, equal_share, load_flattening, total_energy

    infra = getattr(sites, str(g[f"{name}_site"]))()
    m = g[f"{name}_meta"]
    meta = dict(T=int(m[0]), ct="SOC" if m[1] else "LINEAR", eq=bool(m[2]), lf=bool(m[3]), seed=int(m[4]),
                es=float(m[5]) if len(m) > 5 else (0.0 if m[3] else 1e-3))
    if meta["lf"]:
        ext = g[f"{name}_ext"].copy()
        obj = [ObjectiveComponent(load_flattening, 1.0, {"external_signal": ext})]
        spec = [("load_flattening", 1.0, {"external_signal": ext})]
        if meta["es"]:
            obj.append(ObjectiveComponent(equal_share, meta["es"]))
            spec.append(("equal_share", meta["es"], {}))
        iface = Interface({"infrastructure_info": infra, "period": 5})
    else:
        obj = [ObjectiveComponent(total_energy, 20.0), ObjectiveComponent(demand_charge), ObjectiveComponent(equal_share, 1e-3)]
        spec = [("total_energy", 20.0, {}), ("demand_charge", 1.0, {}), ("equal_share", 1e-3, {})]
        iface = Interface({"infrastructure_info": infra, "period": 5, "demand_charge": 15.0, "prev_peak": 50.0})
    st, arr, dep = g[f"{name}_station"], g[f"{name}_arrival"], g[f"{name}_departure"]
    minr, maxr = g[f"{name}_minr"], g[f"{name}_maxr"]
    sessions, o = [], 0
    for k in range(len(st)):
        L = int(dep[k] - arr[k])
        sessions.append(SessionInfo(infra.station_ids[int(st[k])], f"s{k}", float(g[f"{name}_demand"][k]), 0.0,
                                    int(arr[k]), int(dep[k]), current_time=0,
                                    min_rates=minr[o:o + L].copy(), max_rates=maxr[o:o + L].copy()))
        o += L
    exp = dict(rates=g[f"{name}_rates"], obj=float(g[f"{name}_obj"]), binding=int(g[f"{name}_binding"][0]))
    return sessions, infra, iface, obj, spec, meta, exp
