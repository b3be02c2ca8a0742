import sys; sys.path.insert(0, '.')
import numpy as np
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.acn import Interface
from adacharge_amd.backend import SiteHandle, default_options
from adacharge_amd.builder import build_batch, scenario_batch
infra = sites.eight_sites()[3]
iface = Interface({"infrastructure_info": infra, "period": 5})
rng = np.random.default_rng(503)
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-3)]
base = build_batch([sites.random_sessions(infra, 12, rng)], infra, iface, obj, "SOC")
batch = scenario_batch(base, rng.lognormal(0.0, 0.25, size=(1024, base.K, base.N)))
h = SiteHandle(batch.site, 0)
res = h.solve(batch, default_options())
ph, cm = np.deg2rad(infra.phases), infra.constraint_matrix
mag = np.hypot(np.einsum("mn,bnt->bmt", cm * np.cos(ph), res.x), np.einsum("mn,bnt->bmt", cm * np.sin(ph), res.x))
viol = (mag - infra.constraint_limits[None, :, None]).max(axis=(1, 2))
for st in (1, 2, 5):
    m = res.status == st
    if m.any():
        print("status", st, "count", m.sum(), "max violation %.3e" % viol[m].max(), "median %.3e" % np.median(viol[m]), "iters", res.iters[m].min(), res.iters[m].max(),
              "pri max %.2e" % res.pri_res[m].max())
v5 = np.sort(viol[res.status == 5])[::-1]
print("status-5 violations sorted:", np.array2string(v5[:12], precision=2))
