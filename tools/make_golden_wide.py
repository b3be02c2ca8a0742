"""Generates tests/golden/wide.npz: oracle-certified optima for the kernel variants and constraint families the
first fixture file does not reach -- energy equalities (aco.py:116-119), scalar and vector peak limits
(aco.py:196-198), two sessions per EVSE (t_aco.py:194-208), minimum rates, the synthetic 52-EVSE site, horizon
24 (two column tiles), horizon 40 and a 128-EVSE site (general-shape kernel).

As with tools/make_golden.py the expected outputs come from the independent oracle (oracle/ref_problem.py restates
the problem as the reference states it, oracle/ipm.py solves it and certifies the KKT conditions <= 1e-9 on the full
problem); the reference itself cannot run here.  Re-run: python tools/make_golden_wide.py
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from adacharge_amd import sites
from adacharge_amd.acn import Interface
from oracle.ref_problem import build_reference_problem
from oracle.ipm import solve_certified

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "wide.npz")

# name, site, T, constraint type, equality, two sessions per EVSE, min rates, peak ("none" | "scalar" | "vector"),
# equal_share coefficient, seed
CASES = [
    ("eq_soc",       "caltech54", 12, "SOC",    True,  False, False, "none",   1e-3, 201),
    ("eq_lin",       "caltech54", 12, "LINEAR", True,  False, False, "none",   1e-2, 202),
    ("peak_scalar",  "caltech54", 12, "SOC",    False, False, False, "scalar", 1e-3, 203),
    ("peak_vector",  "caltech54", 12, "SOC",    False, False, True,  "vector", 1e-3, 204),
    ("peak_lin",     "caltech54", 12, "LINEAR", False, False, False, "vector", 1e-2, 205),
    ("two_soc",      "caltech54", 16, "SOC",    False, True,  True,  "none",   1e-3, 206),
    ("two_lin",      "caltech54", 16, "LINEAR", False, True,  False, "scalar", 1e-2, 207),
    ("t24_soc",      "caltech54", 24, "SOC",    False, False, True,  "none",   1e-3, 208),
    ("t24_eq",       "caltech54", 24, "SOC",    True,  False, False, "none",   1e-3, 209),
    ("jpl_t24_soc",  "jpl52",     24, "SOC",    False, False, False, "none",   1e-3, 210),
    ("jpl_t24_lin",  "jpl52",     24, "LINEAR", False, True,  True,  "none",   1e-2, 211),
    ("jpl_t12_peak", "jpl52",     12, "SOC",    False, False, False, "scalar", 1e-3, 212),
    ("t40_soc",      "caltech54", 40, "SOC",    False, False, True,  "none",   1e-3, 213),
    ("t40_two",      "caltech54", 40, "LINEAR", False, True,  False, "vector", 1e-2, 214),
    ("wide128_soc",  "wide128",   12, "SOC",    False, False, False, "none",   1e-3, 215),
    ("wide128_lin",  "wide128",   16, "LINEAR", False, True,  False, "none",   1e-2, 217),
    # a day-shaped offline instance (adacharge.py:234-276; t_int.py:350-403 solves one with energy equalities):
    # 288 periods, 150 sessions, up to 6 consecutive sessions per EVSE -- sites.offline_day
    ("offline_day",  "caltech54", 288, "SOC",   True,  "day", False, "none",   1e-3, 218),
    # round 3: a wide site at horizon 48 (large-site kernel with Anderson acceleration), congested (demand_scale 0.6
    # against limits at 35 % of the full load), minimum rates on 30 % of the sessions: 9,216 variables, ~1 min of IPM
    ("wide192_t48_soc", "wide192", 48, "SOC",   False, False, True,  "none",   1e-3, 2192),
    # BASELINE.json configs[4]'s own shape: the synthetic 512-EVSE site at horizon 48 (24,576 variables), congested
    ("synth512_t48_soc", "synth512", 48, "SOC", False, False, False, "none",   1e-3, 2512),
    # the reference's stress shape (t_aco.py:286-313: 54 EVSE x 144 periods) with a strictly convex objective, so that the
    # per-EVSE schedule is pinned (KAT-4 pins only the aggregate of the LP): long-horizon kernel, r0 / zh in LDS
    ("stress144_soc", "caltech54", 144, "SOC", False, False, True,  "none",   1e-3, 2144),
]


def main():
    only = sys.argv[1:]   # `python tools/make_golden_wide.py name ...`: (re)generate these cases, keep the others
    store = {}
    if only and os.path.exists(OUT):
        store = dict(np.load(OUT, allow_pickle=False))
    store["names"] = np.array([c[0] for c in CASES])
    for name, site_name, T, ct, eq, two, mins, peak_kind, es, seed in CASES:
        if only and name not in only:
            assert f"{name}_rates" in store, f"{name} is not in {OUT}: regenerate everything"
            continue
        infra = getattr(sites, site_name)()
        iface = Interface({"infrastructure_info": infra, "period": 5})
        rng = np.random.default_rng(seed)
        if two == "day":
            sl = sites.offline_day(infra, rng, horizon=T)
        else:
            sl = sites.random_sessions_general(infra, T, rng, two, mins,
                                               demand_scale=0.6 if site_name in ("wide192", "synth512") else (0.5 if eq else 1.5))
        Tb = max(s.arrival_offset + s.remaining_time for s in sl)
        full = 32.0 * len(sl)
        peak = None
        if peak_kind == "scalar":
            peak = float(rng.uniform(0.25, 0.5) * full)
        elif peak_kind == "vector":
            peak = rng.uniform(0.25, 0.6, size=Tb) * full
        # the peak must leave the minimum rates feasible
        if peak is not None:
            lbsum = np.zeros(Tb)
            for s in sl:
                lbsum[s.arrival_offset:s.arrival_offset + s.remaining_time] += s.min_rates
            peak = np.maximum(peak, lbsum + 1.0) if peak_kind == "vector" else max(peak, float(lbsum.max()) + 1.0)
        spec = [("quick_charge", 1, {}), ("equal_share", es, {})]
        t0 = time.time()
        prob = build_reference_problem(sl, infra, iface, spec, ct, eq, peak_limit=peak)
        r, res, cert = solve_certified(prob)
        assert cert is not None and cert.worst < 1e-9, (name, cert)
        st = {
            "station": np.array([infra.station_ids.index(s.station_id) for s in sl], np.int32),
            "arrival": np.array([s.arrival for s in sl], np.int32),
            "departure": np.array([s.departure for s in sl], np.int32),
            "demand": np.array([s.remaining_demand for s in sl]),
            "minr": np.concatenate([s.min_rates for s in sl]),
            "maxr": np.concatenate([s.max_rates for s in sl]),
            "peak": np.array([np.nan]) if peak is None else np.atleast_1d(np.asarray(peak, float)),
            "meta": np.array([T, 1 if ct == "SOC" else 0, 1 if eq else 0, es, seed], float),
            "site": np.array(site_name),
            "rates": r,
            "obj": np.array(prob.objective(r)),
            "cert": np.array([cert.stationarity, cert.primal, cert.dual]),
        }
        for k, v in st.items():
            store[f"{name}_{k}"] = v
        K = max(np.bincount(st["station"]))
        print(f"{name:14s} {site_name:10s} T={Tb:3d} {ct:6s} eq={int(eq)} K={K} S={len(sl):3d} peak={peak_kind:6s} "
              f"obj {prob.objective(r):.9f} cert {cert.worst:.1e}  {time.time() - t0:.1f}s", flush=True)
    np.savez_compressed(OUT, **store)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
