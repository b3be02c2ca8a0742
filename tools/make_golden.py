"""Generates tests/golden/*.npz: inputs and oracle-certified schedules.

The reference (cvxpy/ECOS) cannot run here, so the
expected outputs come from the independent oracle (oracle/ref_problem.py +
oracle/ipm.py): every stored schedule carries a KKT certificate < 1e-9, i.e.
it is the exact optimum of the problem the reference states, whatever solver
produced it.  Re-run: python tools/make_golden.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from adacharge_amd import sites
from adacharge_amd.acn import Interface
from oracle.ref_problem import build_reference_problem
from oracle.ipm import solve_certified, solve_lp_highs

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def pack_sessions(sl):
    return dict(
        station=np.array([s.station_id for s in sl]),
        demand=np.array([s.remaining_demand for s in sl]),
        arrival=np.array([s.arrival for s in sl]),
        departure=np.array([s.departure for s in sl]),
        min0=np.array([s.min_rates[0] for s in sl]),
        maxr=np.array([s.max_rates[0] for s in sl]),
    )


def main():
    infra = sites.caltech54()
    iface = Interface({"infrastructure_info": infra, "period": 5})
    T = 12
    cases = []
    for seed in range(6):
        for ct in ("LINEAR", "SOC"):
            for es in (1e-3, 1e-2):
                for eq in (False,):
                    cases.append((seed, ct, es, eq, 0.0))
    cases += [(100 + s, "SOC", 1e-3, False, 0.1) for s in range(3)]   # with min rates
    store = {}
    for n, (seed, ct, es, eq, mf) in enumerate(cases):
        sl = sites.random_sessions(infra, T, np.random.default_rng(seed), min_rate_fraction=mf)
        spec = [("quick_charge", 1, {}), ("equal_share", es, {})]
        prob = build_reference_problem(sl, infra, iface, spec, ct, eq)
        r, res, cert = solve_certified(prob)
        assert cert is not None and cert.worst < 1e-9, (seed, ct, es, cert)
        key = f"c{n:02d}"
        for k, v in pack_sessions(sl).items():
            store[f"{key}_{k}"] = v
        store[f"{key}_meta"] = np.array([seed, 1 if ct == "SOC" else 0, es, 1 if eq else 0, T, mf])
        store[f"{key}_rates"] = r
        store[f"{key}_obj"] = np.array(prob.objective(r))
        store[f"{key}_cert"] = np.array([cert.stationarity, cert.primal, cert.dual])
        print(key, seed, ct, es, eq, "S", len(sl), "obj %.9f" % prob.objective(r), "cert %.1e" % cert.worst)
    # LP cases: objective and aggregate from HiGHS
    for n, seed in enumerate(range(20, 26)):
        sl = sites.random_sessions(infra, T, np.random.default_rng(seed))
        prob = build_reference_problem(sl, infra, iface, [("quick_charge", 1, {})], "LINEAR")
        h = solve_lp_highs(prob)
        assert h.status == 0
        key = f"lp{n:02d}"
        for k, v in pack_sessions(sl).items():
            store[f"{key}_{k}"] = v
        store[f"{key}_meta"] = np.array([seed, 0, 0.0, 0, T, 0.0])
        store[f"{key}_obj"] = np.array(h.fun)
        store[f"{key}_agg"] = h.x.reshape(infra.num_stations, T).sum(0)
        print(key, seed, "LP obj %.9f" % h.fun)
    np.savez_compressed(os.path.join(OUT, "caltech54_T12.npz"), **store)


if __name__ == "__main__":
    main()
