"""Generates tests/golden/stalled.npz: oracle-certified optima of instances that END THE ADAPTIVE FIRST PASS ON A
PLATEAU (status SOLVED_INACCURATE / MAX_ITER after the stall window) -- the congested demand
scenarios of BASELINE.json configs[3] (8 sites x 1024 scenarios, tools/run_config.py cfg4), other snapshots of the
site that produces them (the 36-EVSE synthetic site: of the eight sites, four other workloads and 20,480 random
scenario problems scanned, it is the only one where the twin's single pass ever stalls) at horizons 12 and 24.

How they are found: the C twin of the device algorithm (oracle/admm_port.c, same Anderson memory as the kernels) runs
every scenario with retry_passes = 0; what it leaves at status 2 / 5 is a stalled instance.  Expected outputs come from
the independent oracle (oracle/ref_problem.py + oracle/ipm.py, KKT certificate <= 1e-9 on the full problem), never from
an ADMM.  The GPU test (tests/test_gpu_parity.py::test_stalled_instances_reach_the_certified_optimum) asks the DEFAULT
surface for these instances and compares with the certificate whatever status comes back.

    python tools/make_golden_stalled.py [per_site]       (default: up to 8 per workload that has any, >= 8 in total)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.acn import Interface, SessionInfo
from adacharge_amd.builder import build_batch, scenario_batch
from oracle import admm_port
from oracle.ipm import solve_certified
from oracle.ref_problem import build_reference_problem

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "stalled.npz")
PER_SITE = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ES = 1e-3


def scenario_sessions(infra, sessions, factor):
    """The session list of one demand scenario: remaining_demand scaled as builder.scenario_batch scales the energy
    caps (slot 0 of the session's EVSE; the snapshots have one session per EVSE)."""
    return [SessionInfo(s.station_id, s.session_id, float(s.remaining_demand * factor[0, infra.station_ids.index(s.station_id)]), 0.0,
                        s.arrival, s.departure, current_time=0, min_rates=s.min_rates.copy(), max_rates=s.max_rates.copy())
            for s in sessions]


def main():
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, ES)]
    spec = [("quick_charge", 1, {}), ("equal_share", ES, {})]
    store, names = {}, []

    def certify(name, site_tag, infra, iface, sl, T, first_status, first_iters):
        t0 = time.time()
        prob = build_reference_problem(sl, infra, iface, spec, "SOC", False, peak_limit=None)
        r, res, cert = solve_certified(prob)
        if cert is None or not cert.worst < 1e-9:   # the oracle itself could not certify this one: not a fixture
            print(f"{name}: the IPM oracle did not reach a certificate ({cert}); skipped", flush=True)
            return False
        st = {
            "station": np.array([infra.station_ids.index(s.station_id) for s in sl], np.int32),
            "arrival": np.array([s.arrival for s in sl], np.int32),
            "departure": np.array([s.departure for s in sl], np.int32),
            "demand": np.array([s.remaining_demand for s in sl]),
            "minr": np.concatenate([s.min_rates for s in sl]),
            "maxr": np.concatenate([s.max_rates for s in sl]),
            "peak": np.array([np.nan]),
            "meta": np.array([T, 1, 0, ES, 0], float),
            "site": np.array(site_tag),
            "rates": r,
            "obj": np.array(prob.objective(r)),
            "cert": np.array([cert.stationarity, cert.primal, cert.dual]),
            "first_pass": np.array([first_status, first_iters], np.int32),   # what the twin's single pass made of it
        }
        for k, v in st.items():
            store[f"{name}_{k}"] = v
        names.append(name)
        print(f"{name:12s} {site_tag:10s} T={T} S={len(sl):3d} twin single pass: status {first_status} after {first_iters} it; "
              f"obj {prob.objective(r):.9f} cert {cert.worst:.1e}  {time.time() - t0:.1f}s", flush=True)
        return True

    # ---- configs[3]: 8 sites x 1024 scenarios (tools/run_config.py cfg4, tests ...::test_config4...) -----------------
    for k, infra in enumerate(sites.eight_sites()):
        iface = Interface({"infrastructure_info": infra, "period": 5})
        rng = np.random.default_rng(500 + k)
        sessions = sites.random_sessions(infra, 12, rng)
        base = build_batch([sessions], infra, iface, obj, "SOC")
        f = rng.lognormal(0.0, 0.25, size=(1024, base.K, base.N))
        batch = scenario_batch(base, f)
        t0 = time.time()
        out = admm_port.solve_batch(batch, threads=8, accel_mem=5, retry_passes=0)
        bad = np.flatnonzero(np.isin(out["status"], (2, 5)) & (out["iters"] >= 3000))
        print(f"site {k} ({infra.num_stations} EVSE): {bad.size} stalled of 1024 in the twin's single pass ({time.time() - t0:.0f} s)", flush=True)
        for j in bad[:6]:
            sl = scenario_sessions(infra, sessions, f[j])
            # the scenario as the builder states it must be the scenario as the session list states it
            chk = build_batch([sl], infra, iface, obj, "SOC")
            assert np.allclose(chk.s_cap[0], batch.s_cap[j], rtol=1e-13, atol=0), "scenario reconstruction"
            certify(f"cfg4_s{k}_{j}", f"eight:{k}", infra, iface, sl, 12, int(out["status"][j]), int(out["iters"][j]))

    # ---- the same site, other snapshots (40 random bases x 32 demand scenarios each) at horizons 12 and 24: one case
    # per base that stalls, so that the fixture is not 8 scenarios of one snapshot ------------------------------------
    k = 3
    infra = sites.eight_sites()[k]
    iface = Interface({"infrastructure_info": infra, "period": 5})
    for T, want in ((12, 4), (24, 3)):
        got = 0
        for seed in range(40):
            rng = np.random.default_rng(9000 + 100 * k + seed)
            sessions = sites.random_sessions(infra, T, rng)
            base = build_batch([sessions], infra, iface, obj, "SOC")
            f = rng.lognormal(0.0, 0.25, size=(32, base.K, base.N))
            batch = scenario_batch(base, f)
            out = admm_port.solve_batch(batch, threads=8, accel_mem=5, retry_passes=0)
            bad = np.flatnonzero(np.isin(out["status"], (2, 5)) & (out["iters"] >= 3000))
            if bad.size == 0 or got >= want:
                continue
            j = int(bad[0])
            got += certify(f"s3_T{T}_b{seed}_{j}", f"eight:{k}", infra, iface, scenario_sessions(infra, sessions, f[j]), T,
                           int(out["status"][j]), int(out["iters"][j]))
    # other congested workloads scanned without a single stall in the twin's single pass (kept as a record):
    # jpl52 x 24 seed 31 (4,096), caltech54 x 12 / x 24 and jpl52 x 12 under heavy demand (2,048 each)

    store["names"] = np.array(names)
    assert len(names) >= 8, names
    np.savez_compressed(OUT, **store)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(names), "cases")


if __name__ == "__main__":
    main()
