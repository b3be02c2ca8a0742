"""Dev tool: slowest problems of a batch on the GPU vs the C port (iteration counts)."""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.backend import SiteHandle, default_options
from adacharge_amd.builder import build_batch
from oracle import admm_port
from tests import helpers as H
infra, iface = H.caltech_interface()
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
ct = sys.argv[1] if len(sys.argv) > 1 else "LINEAR"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
batch = build_batch(sites.snapshot_batch(infra, 12, B, seed=20240), infra, iface, obj, ct)
h = SiteHandle(batch.site, 0)
o = default_options(accel_mem=int(os.environ.get('ACCEL', '10')))
m = h.accel_columns(batch.Tm, batch.K, o)
r = h.solve(batch, o)
ref = admm_port.solve_batch(batch, threads=16, accel_mem=m)
idx = np.argsort(r.iters)[-8:]
print("accel", m, "gpu mean %.0f max %d | port mean %.0f max %d" % (r.iters.mean(), r.iters.max(), ref['iters'].mean(), ref['iters'].max()))
print("slowest on gpu:", idx, r.iters[idx], "port:", ref['iters'][idx], "status", r.status[idx])
print("max |dx| %.2e, |iters diff| > 100: %d" % (np.abs(r.x - ref['x']).max(), (np.abs(r.iters - ref['iters']) > 100).sum()))
