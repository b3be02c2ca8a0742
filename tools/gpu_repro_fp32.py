"""fp32, horizon 24 (column tiles CT = 2) through the host-buffer entry: the shape of test_config3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.acn import Interface
from adacharge_amd.backend import SiteHandle, default_options
from adacharge_amd.builder import build_batch

infra = sites.caltech54()
iface = Interface({"infrastructure_info": infra, "period": 5})
T, B = 24, int(sys.argv[1]) if len(sys.argv) > 1 else 8
snaps = sites.snapshot_batch(infra, T, B, seed=31)
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-3)]
batch = build_batch(snaps, infra, iface, obj, "SOC")
h = SiteHandle(batch.site, 0)
print("fp64", flush=True)
r = h.solve(batch, default_options())
print(r.status[:8], r.iters[:8], flush=True)
print("fp32", flush=True)
r = h.solve(batch, default_options(eps_abs=5e-5, eps_rel=5e-5, precision=32))
print(r.status[:8], r.iters[:8], flush=True)
