"""Cases for the wave-per-problem kernel (acn_qp_wave.hpp), each solved through the host entry acnqp_solve_batch.  Run
as a script it solves one case under the environment it was started with and saves the result:
  ACNQP_WAVE_MIN_BATCH=1   every launch of a fitting shape goes to the wave kernel (the default; pinned here)
  ACNQP_NO_WAVE=1          the register-resident tiled kernel instead (the round-1..3 path)
tests/test_wave_kernel.py compares the two -- the same algorithm in two data layouts -- and the C twin."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CASES = ("soc", "linear", "equality", "short", "site30", "peak", "infeasible", "empty_set", "warm", "general_windows", "stalled",
         "h24", "h18_linear", "h24_equality", "h20_windows", "h24_infeasible", "h24_warm",
         "mt2_site36", "mt2_site64", "mt2_short", "mt2_equality", "mt2_warm", "mt2_infeasible", "mt2_h24", "mt2_h17", "mt2_t4", "h13", "h48", "h36_linear", "h40_equality", "flat_linear", "flat_soc", "flat_h24", "dc_linear", "dc_soc", "dc_h24")


def build(name):
    """(batch, options keywords, solve keywords)"""
    from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
    from adacharge_amd.acn import Interface
    from adacharge_amd.builder import build_batch

    infra = sites.caltech54()
    iface = Interface({"infrastructure_info": infra, "period": 5})
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
    rng = np.random.default_rng(9100)
    if name.startswith("dc_"):     # demand_charge: the horizon-wide prox of the "max" row (its sums cross the waves of a group)
        from adacharge_amd import demand_charge, total_energy

        T = 24 if name == "dc_h24" else 12
        diface = Interface({"infrastructure_info": infra, "period": 5, "demand_charge": 15.0, "prev_peak": 50.0})
        dobj = [ObjectiveComponent(total_energy, 20.0), ObjectiveComponent(demand_charge), ObjectiveComponent(equal_share, 1e-3)]
        ct = "LINEAR" if name == "dc_linear" else "SOC"
        return build_batch(sites.snapshot_batch(infra, T, 96, seed=961 + T), infra, diface, dobj, ct), {}, {}
    if name.startswith("flat_"):   # load_flattening: the prox row of the aggregate power (one more site row; SOC: two row tiles)
        from adacharge_amd import load_flattening, total_energy

        T = 24 if name == "flat_h24" else 12
        ext = 40.0 + 30.0 * np.cos(np.arange(T) / T * 2 * np.pi)
        fobj = [ObjectiveComponent(load_flattening, 1.0, {"external_signal": ext}), ObjectiveComponent(total_energy, 1500.0),
                ObjectiveComponent(equal_share, 1e-3)]
        ct = "SOC" if name == "flat_soc" else "LINEAR"
        return build_batch(sites.snapshot_batch(infra, T, 96, seed=951 + T), infra, iface, fobj, ct), {}, {}
    if name == "soc":
        return build_batch(sites.snapshot_batch(infra, 12, 192, seed=911), infra, iface, obj, "SOC"), {}, {}
    if name == "linear":
        return build_batch(sites.snapshot_batch(infra, 12, 192, seed=912), infra, iface, obj, "LINEAR"), {}, {}
    if name == "equality":   # energy DELIVERED (s_eq = 1), small demands so that every instance is feasible
        snaps = sites.snapshot_batch(infra, 12, 96, seed=913, demand_range=(0.1, 0.5))   # (a one-period session takes at most 0.55 kWh)
        return build_batch(snaps, infra, iface, obj, "SOC", True), {}, {}
    if name == "short":      # horizon 7 (five dead period registers per lane), 40 % of the sessions with a minimum rate
        snaps = sites.snapshot_batch(infra, 7, 96, seed=914, min_rate_fraction=0.4)
        return build_batch(snaps, infra, iface, obj, "SOC"), {}, {}
    if name == "site30":     # 30 EVSEs: half a wave of dead lanes, two all-padding EVSE tiles
        infra30 = sites.eight_sites()[2]
        iface30 = Interface({"infrastructure_info": infra30, "period": 5})
        snaps = sites.snapshot_batch(infra30, 12, 96, seed=915)
        return build_batch(snaps, infra30, iface30, obj, "SOC"), {}, {}
    if name == "peak":       # LINEAR rows + a peak row (9 of 16 padded rows), the limit binding in about half the periods
        snaps = sites.snapshot_batch(infra, 12, 96, seed=916)
        peaks = [rng.uniform(150.0, 600.0, size=12) for _ in snaps]
        return build_batch(snaps, infra, iface, obj, "LINEAR", False, peaks), {}, {}
    if name == "infeasible":   # equalities the SITE cannot carry once enough EVSEs are busy (every session alone is feasible:
                               # whole-horizon windows, 3 ... 6 of the 6.6 kWh one EVSE can take): certificates (status 3)
        from adacharge_amd.sites import SessionInfo

        snaps = []
        for b in range(128):
            evses = rng.choice(infra.num_stations, size=int(rng.integers(6, 55)), replace=False)
            snaps.append([SessionInfo(infra.station_ids[int(e)], f"s{k}", float(rng.uniform(3.0, 6.0)), 0.0, 0, 12, current_time=0,
                                      min_rates=np.zeros(12), max_rates=32.0) for k, e in enumerate(evses)])
        return build_batch(snaps, infra, iface, obj, "LINEAR", True), dict(max_iter=30000), {}
    if name == "empty_set":   # demands no session can take inside its own window: status 4 before the first iteration
        snaps = sites.snapshot_batch(infra, 12, 64, seed=917, demand_range=(10.0, 45.0))
        return build_batch(snaps, infra, iface, obj, "LINEAR", True), {}, {}
    if name == "warm":       # warm start from a perturbed earlier answer, multipliers returned
        b = build_batch(sites.snapshot_batch(infra, 12, 64, seed=918), infra, iface, obj, "SOC")
        return b, {}, dict(warm="self", want_y=True)
    if name == "general_windows":   # delayed arrivals and minimum rates over a prefix: lb != 0, windows that do not start at 0
        snaps = [sites.random_sessions_general(infra, 12, rng, False, min_rates=True) for _ in range(96)]
        return build_batch(snaps, infra, iface, obj, "SOC"), {}, {}
    # ---- horizons 13 ... 24: two waves per problem, twelve periods each
    # ---- horizons 33 ... 48 on one row tile: four waves per problem, twelve periods each
    if name == "h48":
        return build_batch(sites.snapshot_batch(infra, 48, 96, seed=941), infra, iface, obj, "SOC"), {}, {}
    if name == "h36_linear":   # the fourth wave holds no live period
        return build_batch(sites.snapshot_batch(infra, 36, 96, seed=942, min_rate_fraction=0.2), infra, iface, obj, "LINEAR"), {}, {}
    if name == "h40_equality":
        snaps = sites.snapshot_batch(infra, 40, 64, seed=943, demand_range=(0.1, 0.5))
        return build_batch(snaps, infra, iface, obj, "SOC", True), {}, {}
    if name == "h13":          # one live period in the second wave
        return build_batch(sites.snapshot_batch(infra, 13, 96, seed=925), infra, iface, obj, "SOC"), {}, {}
    if name == "h24":
        return build_batch(sites.snapshot_batch(infra, 24, 192, seed=921), infra, iface, obj, "SOC"), {}, {}
    if name == "h18_linear":   # the second wave holds six live periods
        return build_batch(sites.snapshot_batch(infra, 18, 128, seed=922, min_rate_fraction=0.3), infra, iface, obj, "LINEAR"), {}, {}
    if name == "h24_equality":
        snaps = sites.snapshot_batch(infra, 24, 96, seed=923, demand_range=(0.1, 0.5))
        return build_batch(snaps, infra, iface, obj, "SOC", True), {}, {}
    if name == "h20_windows":   # windows that start late / end early, minimum rates over a prefix: sums that cross the halves
        snaps = [sites.random_sessions_general(infra, 20, rng, False, min_rates=True) for _ in range(96)]
        return build_batch(snaps, infra, iface, obj, "SOC"), {}, {}
    if name == "h24_infeasible":
        from adacharge_amd.sites import SessionInfo

        snaps = []
        for b in range(96):
            evses = rng.choice(infra.num_stations, size=int(rng.integers(6, 55)), replace=False)
            snaps.append([SessionInfo(infra.station_ids[int(e)], f"s{k}", float(rng.uniform(6.0, 12.0)), 0.0, 0, 24, current_time=0,
                                      min_rates=np.zeros(24), max_rates=32.0) for k, e in enumerate(evses)])
        return build_batch(snaps, infra, iface, obj, "LINEAR", True), dict(max_iter=30000), {}
    if name == "h24_warm":
        b = build_batch(sites.snapshot_batch(infra, 24, 64, seed=924), infra, iface, obj, "SOC")
        return b, {}, dict(warm="self", want_y=True)
    # ---- two row tiles (17 ... 32 site rows) at horizon <= 12: two waves per problem, six periods each
    if name.startswith("mt2_"):
        k = 7 if name == "mt2_site64" else 3          # eight_sites: 36 EVSEs / 18 rows, 64 EVSEs / 20 rows
        infra2 = sites.eight_sites()[k]
        iface2 = Interface({"infrastructure_info": infra2, "period": 5})
        if name in ("mt2_site36", "mt2_site64"):
            return build_batch(sites.snapshot_batch(infra2, 12, 128, seed=931 + k), infra2, iface2, obj, "SOC"), {}, {}
        if name == "mt2_h24":        # two row tiles AND horizon 13 ... 24: four waves per problem, six periods each
            return build_batch(sites.snapshot_batch(infra2, 24, 96, seed=936), infra2, iface2, obj, "SOC"), {}, {}
        if name == "mt2_h17":        # ... the fourth wave idle (periods 18 ... 23 do not exist), the third with five live periods
            return build_batch(sites.snapshot_batch(infra2, 17, 96, seed=937, min_rate_fraction=0.1), infra2, iface2, obj, "SOC"), {}, {}
        if name == "mt2_t4":         # horizon 4: the second wave of the pair holds no live period at all
            return build_batch(sites.snapshot_batch(infra2, 4, 64, seed=938, demand_range=(0.2, 2.0)), infra2, iface2, obj, "SOC"), {}, {}
        if name == "mt2_short":      # horizon 7: the second wave holds one live period; minimum rates
            return build_batch(sites.snapshot_batch(infra2, 7, 96, seed=933, min_rate_fraction=0.1), infra2, iface2, obj, "SOC"), {}, {}
        if name == "mt2_equality":
            snaps = sites.snapshot_batch(infra2, 12, 96, seed=934, demand_range=(0.05, 0.25))
            return build_batch(snaps, infra2, iface2, obj, "SOC", True), {}, {}
        if name == "mt2_warm":
            return build_batch(sites.snapshot_batch(infra2, 12, 64, seed=935), infra2, iface2, obj, "SOC"), {}, dict(warm="self", want_y=True)
        if name == "mt2_infeasible":
            from adacharge_amd.sites import SessionInfo

            snaps = []
            for b in range(96):
                evses = rng.choice(infra2.num_stations, size=int(rng.integers(2, 24)), replace=False)
                snaps.append([SessionInfo(infra2.station_ids[int(e)], f"s{j}", float(rng.uniform(3.0, 6.0)), 0.0, 0, 12, current_time=0,
                                          min_rates=np.zeros(12), max_rates=32.0) for j, e in enumerate(evses)])
            return build_batch(snaps, infra2, iface2, obj, "SOC", True), dict(max_iter=30000), {}
    if name == "stalled":    # the congested fixtures the polish exists for (hand-over at polish_iters, resume behind it)
        from tests import helpers as H

        g = H.load_stalled()
        names = [str(n) for n in g["names"] if int(g[f"{n}_meta"][0]) == 12]
        cases = [H.wide_case(g, n) for n in names]
        meta = cases[0][3]
        sobj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, meta["es"])]
        return build_batch([c[0] for c in cases], cases[0][1], cases[0][2], sobj, "SOC"), {}, {}
    raise KeyError(name)


def solve(name):
    from adacharge_amd.backend import SiteHandle, default_options

    batch, okw, skw = build(name)
    h = SiteHandle(batch.site, 0)
    opts = default_options(**okw)
    warm = None
    if skw.get("warm") == "self":
        first = h.solve(batch, opts, want_y=True)
        rng = np.random.default_rng(5)
        warm = (first.x * rng.uniform(0.9, 1.0, size=first.x.shape), first.y)
    res = h.solve(batch, opts, warm=warm, want_y=bool(skw.get("want_y")))
    out = dict(x=res.x, iters=res.iters, status=res.status, pri=res.pri_res, dua=res.dua_res, obj=res.obj)
    if res.y is not None:
        out["y"] = res.y
    h.close()
    return out


if __name__ == "__main__":
    np.savez(sys.argv[2], **solve(sys.argv[1]))
