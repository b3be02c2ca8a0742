"""Generates tests/golden/prox.npz: oracle-certified optima of problems whose objective carries a PROX row --
load_flattening (aco.py:403-408, quadratic in the per-period aggregate power) or demand_charge (aco.py:387-400, the
horizon-wide max) -- at the shapes the large-site kernel (N > 64) and the long-horizon kernel (horizon > 32) serve.
tests/golden/wide.npz pins those two kernels with quick_charge + equal_share only (round-3 finding: their prox rows were
compared with the C twin alone), and BASELINE.json configs[4] names load_flattening on 512 EVSE x 48 as its workload.

Cases
  lf192_t48_eq   wide192   x 48  load_flattening(ext) with energy equalities, site rows binding   large-site kernel, flat row
  lf512_t48_eq   synth512  x 48  bench.py's configs[4] leg: ITS generator, ITS first snapshot      large-site kernel, flat row
  dc128_t40      wide128   x 40  total_energy + demand_charge + equal_share                         large-site kernel, max row
  dc54_t96_soc   caltech54 x 96  the same objective                                                 long-horizon kernel, max row
  dc54_t144_lin  caltech54 x 144 the same, LINEAR rows                                              long-horizon kernel, max row
  lf54_t144_eq   caltech54 x 144 load_flattening(ext) with energy equalities                        long-horizon kernel, flat row

Expected outputs come from the independent oracle (oracle/ref_problem.py restates the problem as the reference states
it, oracle/ipm.py solves it and certifies the KKT conditions on the full problem); the reference itself cannot run here
(SURVEY.md section 8c).  Every case is saved as soon as it is certified.

    python tools/make_golden_prox.py [name ...]         (re)generate these cases (default: all), keep the others
    python tools/make_golden_prox.py --probe [name ...]  C twin only: iterations, row utilisation (choosing the scales)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from adacharge_amd import sites
from adacharge_amd.acn import Interface

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "prox.npz")


def ext_profile(T, amp=100.0):
    """External load profile (kW); amp = 100 is bench.py's of the configs[4] leg, 150 + 100 cos.  Load flattening fills
    the profile's valley: a deep one (amp ~ a quarter of the site's full load) is what makes the site rows BIND there."""
    return amp * (1.5 + np.cos(np.arange(T) / T * 2 * np.pi))


def _gen_general(scale, mins=False):
    return lambda infra, T, rng: sites.random_sessions_general(infra, T, rng, False, mins, demand_scale=scale)


def _gen_snapshot(seed_kw):
    return lambda infra, T, rng: sites.random_sessions(infra, T, rng, **seed_kw)


def _gen_bench_cfg4(index):
    """bench.other_workloads()["cfg4_synth512_T48_b2048"]: snapshot `index` of its eight (one generator, drawn in order)"""
    def gen(infra, T, rng):
        return [sites.random_sessions_general(infra, T, rng, False, False, demand_scale=0.12) for _ in range(index + 1)][index]
    return gen


def _gen_dense(lo, hi):
    """every EVSE occupied: arrival in the first third, a stay of at least half the horizon, a demand of lo..hi of what
    the stay could deliver at 32 A (the pods' feeders carry 31 % of their EVSEs' full load on the Caltech-shaped site)"""
    from adacharge_amd.acn import SessionInfo

    def gen(infra, T, rng):
        out = []
        for i in range(infra.num_stations):
            a = int(rng.integers(0, T // 3))
            d = int(rng.integers(a + T // 2, T + 1))
            L, k = d - a, float(infra.voltages[i]) * 5 / 60 / 1e3
            out.append(SessionInfo(infra.station_ids[i], f"d{i}", float(rng.uniform(lo, hi) * 32 * L * k), 0.0, a, d,
                                   current_time=0, min_rates=np.zeros(L), max_rates=np.full(L, 32.0)))
        return out
    return gen


# name -> (site, T, constraint type, equality, kind ("lf": amplitude of the external profile | "dc"), seed, generator)
CASES = {
    "lf192_t48_eq":  ("wide192",   48,  "SOC",    True,  ("lf", 300.0), 5192, _gen_general(0.3)),
    # the bench leg's own workload: rng = default_rng(5), its sixth snapshot (the one of the eight whose site rows bind
    # at the demand the generator draws; the leg's 256 scenarios per snapshot scale it by lognormal(0, 0.05))
    "lf512_t48_eq":  ("synth512",  48,  "SOC",    True,  ("lf", 100.0), 5,    _gen_bench_cfg4(5)),
    "dc128_t40":     ("wide128",   40,  "SOC",    False, ("dc", 0.0),   47,   _gen_snapshot(dict(min_sessions=60, demand_range=(5.0, 30.0)))),
    "dc54_t96_soc":  ("caltech54", 96,  "SOC",    False, ("dc", 0.0),   103,  _gen_snapshot(dict(demand_range=(5.0, 90.0)))),
    "dc54_t144_lin": ("caltech54", 144, "LINEAR", False, ("dc", 0.0),   151,  _gen_snapshot(dict(demand_range=(5.0, 60.0)))),
    "lf54_t144_eq":  ("caltech54", 144, "SOC",    True,  ("lf", 1000.0), 5144, _gen_dense(0.18, 0.3)),
    # the same three with equal_share * 1e-3 beside the flattening term: load_flattening alone is quadratic in the
    # per-period AGGREGATE only (rank one per period), so the per-EVSE split of its optimum is not unique -- like the LP
    # of quick_charge.  The pure cases pin objective, aggregate power and feasibility; these pin the per-EVSE rates.
    "lf192_t48_eq_es": ("wide192",   48,  "SOC", True, ("lf", 300.0, 1e-3),  5192, _gen_general(0.3)),
    "lf512_t48_eq_es": ("synth512",  48,  "SOC", True, ("lf", 100.0, 1e-3),  5,    _gen_bench_cfg4(5)),
    "lf54_t144_eq_es": ("caltech54", 144, "SOC", True, ("lf", 1000.0, 1e-3), 5144, _gen_dense(0.18, 0.3)),
}


def objective_of(kind, T):
    """(spec for oracle/ref_problem.py, ObjectiveComponent list for the builder, Interface settings)"""
    from adacharge_amd import ObjectiveComponent, demand_charge, equal_share, load_flattening, total_energy

    if kind[0] == "lf":
        ext = ext_profile(T, kind[1])
        es = kind[2] if len(kind) > 2 else 0.0
        spec = [("load_flattening", 1.0, {"external_signal": ext})] + ([("equal_share", es, {})] if es else [])
        obj = [ObjectiveComponent(load_flattening, 1.0, {"external_signal": ext})] + ([ObjectiveComponent(equal_share, es)] if es else [])
        return spec, obj, {}
    return ([("total_energy", 20.0, {}), ("demand_charge", 1.0, {}), ("equal_share", 1e-3, {})],
            [ObjectiveComponent(total_energy, 20.0), ObjectiveComponent(demand_charge), ObjectiveComponent(equal_share, 1e-3)],
            {"demand_charge": 15.0, "prev_peak": 50.0})


def case_problem(name):
    site_name, T, ct, eq, kind, seed, gen = CASES[name]
    infra = getattr(sites, site_name)()
    sl = gen(infra, T, np.random.default_rng(seed))
    Tb = max(s.arrival_offset + s.remaining_time for s in sl)   # the problem's own horizon (aco.py:243-245)
    spec, obj, extra = objective_of(kind, Tb)
    iface = Interface({"infrastructure_info": infra, "period": 5, **extra})
    return infra, iface, sl, spec, obj, ct, eq, kind, seed, T, site_name


def utilisation(r, infra, ct):
    cm = infra.constraint_matrix
    if ct == "SOC":
        ph = np.deg2rad(infra.phases)
        mag = np.hypot((cm * np.cos(ph)) @ r, (cm * np.sin(ph)) @ r)
    else:
        mag = np.abs(cm) @ r
    return mag / infra.constraint_limits[:, None]


def probe(names):
    from adacharge_amd.builder import build_batch
    from oracle import admm_port

    for name in names:
        infra, iface, sl, spec, obj, ct, eq, kind, seed, T, site_name = case_problem(name)
        batch = build_batch([sl], infra, iface, obj, ct, eq)
        t0 = time.time()
        out = admm_port.solve_batch(batch, accel_mem=5)
        Tb = int(batch.T[0])
        u = utilisation(out["x"][0][:, :Tb], infra, ct)
        print(f"{name:14s} S={len(sl):3d} T={Tb} status {out['status'][0]} iters {out['iters'][0]} rows>0.999: {(u > 0.999).sum()} "
              f"max util {u.max():.4f}  {time.time() - t0:.1f}s", flush=True)


def main():
    args = sys.argv[1:]
    if args and args[0] == "--probe":
        return probe(args[1:] or list(CASES))
    from oracle.ipm import solve_certified
    from oracle.ref_problem import build_reference_problem

    only = args or list(CASES)
    for name in only:
        infra, iface, sl, spec, obj, ct, eq, kind, seed, T, site_name = case_problem(name)
        t0 = time.time()
        prob = build_reference_problem(sl, infra, iface, spec, ct, eq)
        how = "ipm"
        r = cert = None
        if kind[0] != "lf":
            r, res, cert = solve_certified(prob)
        if cert is None or not cert.worst < 1e-9:
            # The interior-point method stops short of 1e-9 on the load-flattening problems (costs of 1e6 with a rank-one
            # quadratic per period: its reduced-accuracy exit, then an active-set guess the polish cannot repair in its
            # rounds).  What makes a fixture an ORACLE value is the KKT certificate on the full problem the reference
            # states -- sufficient for a convex program wherever the candidate came from -- so the active-set Newton
            # (oracle.ipm.polish) is started from the C twin's answer instead, and the same certificate is demanded.
            from adacharge_amd.builder import build_batch
            from oracle import admm_port
            from oracle.ipm import polish

            batch = build_batch([sl], infra, iface, obj, ct, eq)
            tw = admm_port.solve_batch(batch, eps_abs=1e-10, eps_rel=1e-10, max_iter=200000, accel_mem=5)
            assert tw["status"][0] == 1, (name, tw["status"], tw["iters"])
            r, cert = polish(prob, tw["x"][0][:, :prob.T], duals=None, act_tol=1e-6, max_rounds=60)
            r = prob.rates_of(r)
            how = "twin+polish"
        assert cert is not None and cert.worst < 1e-9, (name, cert)
        Tb = prob.T
        u = utilisation(r, infra, ct)
        st = {
            "station": np.array([infra.station_ids.index(s.station_id) for s in sl], np.int32),
            "arrival": np.array([s.arrival for s in sl], np.int32),
            "departure": np.array([s.departure for s in sl], np.int32),
            "demand": np.array([s.remaining_demand for s in sl]),
            "minr": np.concatenate([s.min_rates for s in sl]),
            "maxr": np.concatenate([s.max_rates for s in sl]),
            "meta": np.array([T, 1 if ct == "SOC" else 0, 1 if eq else 0, 1 if kind[0] == "lf" else 0, seed,
                              kind[2] if kind[0] == "lf" and len(kind) > 2 else (1e-3 if kind[0] == "dc" else 0.0)], float),
            "ext": ext_profile(Tb, kind[1]) if kind[0] == "lf" else np.zeros(0),
            "site": np.array(site_name),
            "rates": r,
            "obj": np.array(prob.objective(r)),
            "cert": np.array([cert.stationarity, cert.primal, cert.dual]),
            "how": np.array(how),
            "binding": np.array([(u > 1 - 1e-6).sum(), u.max()]),
        }
        # (re-read the file right before it is rewritten: several generators may run side by side, one case each)
        store = dict(np.load(OUT, allow_pickle=False)) if os.path.exists(OUT) else {}
        for k, v in st.items():
            store[f"{name}_{k}"] = v
        store["names"] = np.array(sorted({k.rsplit("_", 1)[0] for k in store if k.endswith("_rates")}))
        np.savez_compressed(OUT, **store)
        print(f"{name:14s} {site_name:10s} T={Tb:3d} {ct:6s} eq={int(eq)} S={len(sl):3d} obj {prob.objective(r):.9f} "
              f"cert {cert.worst:.1e} ({how}) binding rows {int(st['binding'][0])} (max util {u.max():.6f})  {time.time() - t0:.1f}s", flush=True)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
