import sys, time
sys.path.insert(0, '.')
import numpy as np
from adacharge_amd.acn import Interface
from adacharge_amd import sites
from adacharge_amd.builder import build_batch
from adacharge_amd.adaptive_charging_optimization import *
from oracle.ref_problem import build_reference_problem
from oracle.ipm import solve_lp_highs
from oracle.admm_ref import solve_one, AdmmOptions
infra = sites.caltech54()
iface = Interface({"infrastructure_info": infra, "period":5})
T=12
ct="LINEAR"
kw = eval("dict(%s)"%sys.argv[1]) if len(sys.argv)>1 else {}
es_eff = float(sys.argv[2]) if len(sys.argv)>2 else 0.0
tot=0
for seed in range(8):
    sess = sites.random_sessions(infra, T, np.random.default_rng(seed))
    prob = build_reference_problem(sess, infra, iface, [("quick_charge",1,{})], ct)
    h = solve_lp_highs(prob)
    batch = build_batch([sess], infra, iface, [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, es_eff)], ct)
    opts = AdmmOptions(max_iter=30000, **kw)
    out = solve_one(batch, 0, opts)
    x = out['x'][:, :T]
    lpobj = prob.objective(x)
    viol = max(0, (prob.A_ub@x.reshape(-1) - prob.b_ub).max())
    tot+=out['iters']
    print(seed, len(sess), "st",out['status'], "it",out['iters'], "rho %.3g"%out['rho'], "LP objgap %.2e viol %.1e agg %.2e"%((lpobj-h.fun)/abs(h.fun), viol, np.abs(x.sum(0)-h.x.reshape(54,T).sum(0)).max()))
print("total iters", tot)
