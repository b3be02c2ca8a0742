import sys, time
sys.path.insert(0, '.')
import numpy as np
from adacharge_amd.acn import Interface
from adacharge_amd import sites
from adacharge_amd.builder import build_batch
from adacharge_amd.adaptive_charging_optimization import *
from oracle.ref_problem import build_reference_problem
from oracle.ipm import solve_certified
from oracle.admm_ref import solve_one, AdmmOptions
infra = sites.caltech54()
iface = Interface({"infrastructure_info": infra, "period":5})
T=12
es = float(sys.argv[1]) if len(sys.argv)>1 else 1e-3
ct = sys.argv[2] if len(sys.argv)>2 else "LINEAR"
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, es)]
spec = [("quick_charge",1,{}),("equal_share",es,{})]
for seed in range(6):
    sess = sites.random_sessions(infra, T, np.random.default_rng(seed))
    prob = build_reference_problem(sess, infra, iface, spec, ct)
    r, res, cert = solve_certified(prob)
    batch = build_batch([sess], infra, iface, obj, ct)
    for eps in (1e-4, 1e-6, 1e-8):
        tr=[]
        t=time.time(); out = solve_one(batch, 0, AdmmOptions(eps_abs=eps, eps_rel=eps, max_iter=50000), trace=tr); dt=time.time()-t
        dx = np.abs(out['x'][:, :T]-r).max()
        agg = np.abs(out['x'][:, :T].sum(0)-r.sum(0)).max()
        print(seed, len(sess), "eps%.0e"%eps, "st",out['status'], "it",out['iters'], "rho %.3g"%out['rho'], "dx %.2e agg %.2e objgap %.2e"%(dx, agg, (out['obj']-prob.objective(r))/abs(prob.objective(r))), "cert %.1e"%cert.worst)
