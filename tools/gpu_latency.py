"""Dev tool: end-to-end latency of one AdaptiveChargingOptimization.solve() call (the reference's own usage:
one MPC step, host buffers in and out) and of solve_batch() on 256 snapshots, split into builder / library."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from adacharge_amd import AdaptiveChargingOptimization, ObjectiveComponent, equal_share, quick_charge, sites
from tests import helpers as H
infra, iface = H.caltech_interface()
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
snaps = sites.snapshot_batch(infra, 12, 256, seed=20240)
opt = AdaptiveChargingOptimization(obj, iface, solver_options=dict(eps_abs=1e-8, eps_rel=1e-8))
opt.solve(snaps[0], infra)   # site upload, module load
lat = []
for sl in snaps[:64]:
    t0 = time.perf_counter(); opt.solve(sl, infra); lat.append(time.perf_counter() - t0)
lat = np.array(lat) * 1e3
print("solve(): median %.2f ms, p90 %.2f ms, min %.2f ms per MPC step (builder + H2D + kernel + D2H)" % (np.median(lat), np.percentile(lat, 90), lat.min()))
for label in ("first call (allocations, staging buffers)", "second call"):
    t0 = time.perf_counter(); rates, status = opt.solve_batch(snaps, infra); t1 = time.perf_counter()
    print("solve_batch(256), %s: %.1f ms wall = %.0f QP/s end to end (kernel %.2f ms)" % (label, (t1 - t0) * 1e3, 256 / (t1 - t0), opt.last_result.kernel_ms))
# the C-ABI host-buffer entry alone (no builder): H2D of the problem arrays + kernel + D2H of schedules and statuses
from adacharge_amd.backend import SiteHandle, default_options
from adacharge_amd.builder import build_batch
batch = build_batch(snaps, infra, iface, obj, "SOC")
h = SiteHandle(batch.site, 0)
o = default_options()
h.solve(batch, o)
ts = []
for _ in range(10):
    t0 = time.perf_counter(); r = h.solve(batch, o); ts.append(time.perf_counter() - t0)
print("acnqp_solve_batch(256) host buffers: median %.2f ms wall = %.0f QP/s (kernel %.2f ms)" % (np.median(ts) * 1e3, 256 / np.median(ts), r.kernel_ms))
