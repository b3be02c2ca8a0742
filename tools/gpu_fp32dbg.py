import sys, os
os.environ["ACNQP_LIBRARY"] = os.path.abspath(sys.argv[1])
sys.path.insert(0, '.')
import numpy as np, torch
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.backend import SiteHandle, default_options, DeviceBatch
from adacharge_amd.builder import build_batch
from tests import helpers as H
infra, iface = H.caltech_interface()
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
snaps = sites.snapshot_batch(infra, 12, 256, seed=20240)
batch = build_batch(snaps, infra, iface, obj, "SOC")
h = SiteHandle(batch.site, 0)
for prec, eps in ((32, 0.0), (32, 5e-5), (64, 5e-5)):
    o = default_options(max_iter=1000 if eps == 0 else 20000, eps_abs=eps, eps_rel=eps, precision=prec)
    res = h.solve(batch, o); res = h.solve(batch, o)
    print(sys.argv[1][-12:], "prec", prec, "eps", eps, "kernel_ms %.3f"%res.kernel_ms, "iters mean %.0f max %d solved %d"%(res.iters.mean(), res.iters.max(), (res.status==1).sum()))
