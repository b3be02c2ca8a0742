"""cProfile of AdaptiveSchedulingAlgorithm.schedule() on one snapshot (bench.py host_inclusive.single_step): where a
single MPC step's host time goes.    python tools/profile_single_step.py [calls]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from adacharge_amd import AdaptiveSchedulingAlgorithm, ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.acn import Interface

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
infra = sites.caltech54()
iface = Interface({"infrastructure_info": infra, "period": 5})
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
snaps = sites.snapshot_batch(infra, 12, n + 16, seed=515151)
alg = AdaptiveSchedulingAlgorithm(obj, solver_options={})
alg.register_interface(iface)
for k in range(16):
    alg.schedule(snaps[k])
lat = []
pr = cProfile.Profile()
for k in range(16, 16 + n):
    t0 = time.perf_counter()
    pr.enable(); alg.schedule(snaps[k]); pr.disable()
    lat.append(time.perf_counter() - t0)
print("median %.3f ms (with the profiler on), min %.3f" % (1e3 * np.median(lat), 1e3 * min(lat)))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
