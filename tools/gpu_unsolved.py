"""Dev tool: which problems of the synthetic JPL T = 24 batch stop at max_iter, and with what residuals."""
import sys
sys.path.insert(0, '.')
import numpy as np
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.acn import Interface
from adacharge_amd.backend import SiteHandle, default_options
from adacharge_amd.builder import build_batch
infra = sites.jpl52(); iface = Interface({"infrastructure_info": infra, "period": 5})
qc = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
batch = build_batch(sites.snapshot_batch(infra, 24, 4096, seed=3), infra, iface, qc, "SOC")
h = SiteHandle(batch.site, 0)
for kw in (dict(), dict(accel_mem=0), dict(max_iter=100000)):
    r = h.solve(batch, default_options(**kw))
    bad = np.flatnonzero(r.status != 1)
    print(kw, "unsolved", bad.tolist(), "status", r.status[bad].tolist(), "iters", r.iters[bad].tolist(),
          "pri", r.pri_res[bad].tolist(), "dua", r.dua_res[bad].tolist(), "mean iters %.0f" % r.iters.mean(), flush=True)
