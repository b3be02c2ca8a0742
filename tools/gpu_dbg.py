import sys, time
sys.path.insert(0, '.')
import numpy as np
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.backend import SiteHandle, default_options
from adacharge_amd.builder import build_batch
from tests import helpers as H
infra, iface = H.caltech_interface()
T=12
snaps = sites.snapshot_batch(infra, T, 256, seed=11)
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-3)]
for ct in ("LINEAR","SOC"):
    batch = build_batch(snaps, infra, iface, obj, ct)
    h = SiteHandle(batch.site, 0)
    for eps in (1e-6, 1e-7, 1e-8):
        res = h.solve(batch, default_options(eps_abs=eps, eps_rel=eps))
        ph = np.deg2rad(infra.phases); cm = infra.constraint_matrix
        re = np.einsum("mn,bnt->bmt", cm*np.cos(ph), res.x); im = np.einsum("mn,bnt->bmt", cm*np.sin(ph), res.x)
        viol = (np.hypot(re, im) - infra.constraint_limits[None,:,None])
        lin = (np.einsum("mn,bnt->bmt", np.abs(cm), res.x) - infra.constraint_limits[None,:,None])
        b = np.unravel_index(viol.argmax(), viol.shape)
        print(ct, "eps", eps, "solved", (res.status==1).sum(), "iters mean %.0f max %d"%(res.iters.mean(), res.iters.max()), "max SOC viol %.2e at %s, max LIN viol %.2e, pri_res max %.2e"%(viol.max(), b, lin.max(), res.pri_res.max()), "ms %.2f"%res.kernel_ms)
# fp32
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-2)]
snaps = sites.snapshot_batch(infra, T, 32, seed=9)
batch = build_batch(snaps, infra, iface, obj, "SOC")
h = SiteHandle(batch.site, 0)
r64 = h.solve(batch, default_options(eps_abs=1e-9, eps_rel=1e-9, reg_rel=0.0))
for eps in (1e-3, 2e-4, 5e-5, 1e-5, 2e-6):
    r32 = h.solve(batch, default_options(eps_abs=eps, eps_rel=eps, reg_rel=0.0, precision=32, max_iter=20000))
    r64e = h.solve(batch, default_options(eps_abs=eps, eps_rel=eps, reg_rel=0.0))
    print("fp32 eps", eps, "solved", (r32.status==1).sum(), "iters mean %.0f"%r32.iters.mean(), "max|x32-x64| %.2e"%np.abs(r32.x-r64.x).max(), " fp64 same eps: iters %.0f err %.2e"%(r64e.iters.mean(), np.abs(r64e.x-r64.x).max()), "ms32 %.2f ms64 %.2f"%(r32.kernel_ms, r64e.kernel_ms))
import torch
print(torch.cuda.is_available())
