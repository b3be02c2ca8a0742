"""End-to-end latency of ONE AdaptiveChargingOptimization.solve() call on the reference's stress shape (54 EVSEs x 144
periods, tests/test_adacharge_stress.py of the reference: single problems, one at a time)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from adacharge_amd import AdaptiveChargingOptimization, ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.acn import Interface
infra = sites.caltech54()
iface = Interface({"infrastructure_info": infra, "period": 5})
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
for T in (48, 144, 288):
    snaps = sites.snapshot_batch(infra, T, 24, seed=100 + T, demand_range=(5.0, 60.0))
    for ct in ("SOC", "LINEAR"):
        opt = AdaptiveChargingOptimization(obj, iface, constraint_type=ct)
        opt.solve(snaps[0], infra)
        lat, its = [], []
        for sl in snaps:
            t0 = time.perf_counter(); opt.solve(sl, infra); lat.append(time.perf_counter() - t0); its.append(int(opt.last_result.iters[0]))
        lat = np.array(lat) * 1e3
        print(f"54 x {T} {ct}: solve() median {np.median(lat):.1f} ms, p90 {np.percentile(lat, 90):.1f} ms, max {lat.max():.1f} ms  (iterations median {int(np.median(its))}, max {max(its)})", flush=True)
