import sys, time, os
sys.path.insert(0, '.')
import numpy as np
from adacharge_amd.acn import Interface
from adacharge_amd import sites
from adacharge_amd.builder import build_batch
from adacharge_amd.adaptive_charging_optimization import *
from adacharge_amd.backend import SiteHandle, default_options
from oracle.ref_problem import build_reference_problem
from oracle.ipm import solve_certified
from oracle.admm_ref import solve_one, AdmmOptions
infra = sites.caltech54()
iface = Interface({"infrastructure_info": infra, "period":5})
T=12
for ct in ("LINEAR","SOC"):
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-3)]
    spec = [("quick_charge",1,{}),("equal_share",1e-3,{})]
    sl = sites.snapshot_batch(infra, T, 8, seed=1)
    batch = build_batch(sl, infra, iface, obj, ct)
    h = SiteHandle(batch.site)
    o = default_options(eps_abs=1e-8, eps_rel=1e-8, reg_rel=0.0, rho=0.1)
    t=time.time(); res = h.solve(batch, o); dt=time.time()-t
    print(ct, "status", res.status, "iters", res.iters, "kernel_ms %.3f wall %.3f"%(res.kernel_ms, dt))
    for b in range(4):
        ref = solve_one(batch, b, AdmmOptions(eps_abs=1e-8, eps_rel=1e-8, rho=0.1, sigma=1e-6, check_every=10, adapt_every=50))
        prob = build_reference_problem(sl[b], infra, iface, spec, ct)
        r, ires, cert = solve_certified(prob)
        print("  b",b,"gpu it",res.iters[b],"ref it",ref['iters'],"|gpu-ref| %.2e"%np.abs(res.x[b]-ref['x']).max(), "|gpu-ipm| %.2e"%np.abs(res.x[b][:, :T]-r).max(), "cert %.1e"%cert.worst, "obj gpu %.8f ipm %.8f"%(res.obj[b], prob.objective(r)))
# throughput
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
for B in (256, 4096):
    sl = sites.snapshot_batch(infra, T, B, seed=2)
    batch = build_batch(sl, infra, iface, obj, "LINEAR")
    h = SiteHandle(batch.site)
    o = default_options()
    res = h.solve(batch, o)
    t=time.time(); res = h.solve(batch, o); dt=time.time()-t
    print("B",B,"solved",(res.status==1).sum(),"iters mean %.0f max %d"%(res.iters.mean(), res.iters.max()),"kernel_ms %.3f wall_ms %.3f -> %.0f QP/s (kernel)"%(res.kernel_ms, dt*1e3, B/res.kernel_ms*1e3))
