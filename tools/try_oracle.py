import sys, time
sys.path.insert(0, '.')
import numpy as np
from adacharge_amd.acn import Interface
from adacharge_amd import sites
from tests.acn_testing import *
from oracle.ref_problem import build_reference_problem
from oracle.ipm import solve_reference_problem, solve_lp_highs

# KAT-1
sd = session_generator(2,[0]*2,[12]*2,[3.3]*2,[3.3]*2,[32]*2)
infra = single_phase_single_constraint(2, 64)
iface = TestingInterface({"active_sessions": sd, "infrastructure_info": infra, "current_time":0, "period":5})
for ct in ("SOC","LINEAR"):
  for eq in (False, True):
    prob = build_reference_problem(iface.active_sessions(), iface.infrastructure_info(), iface, [("quick_charge",1,{})], ct, eq)
    t=time.time(); r, res = solve_reference_problem(prob); 
    print(ct, eq, res.status, res.iters, time.time()-t, r[0])
# caltech
infra = sites.caltech54()
iface = Interface({"infrastructure_info": infra, "period":5})
for seed in range(3):
    sess = sites.random_sessions(infra, 12, np.random.default_rng(seed))
    prob = build_reference_problem(sess, infra, iface, [("quick_charge",1,{})], "LINEAR")
    t=time.time(); r, res = solve_reference_problem(prob); dt=time.time()-t
    h = solve_lp_highs(prob)
    print(seed, len(sess), res.status, res.iters, dt, res.pcost, h.fun, abs(res.pcost-h.fun), np.abs(r.sum(0)-h.x.reshape(54,12).sum(0)).max())
    prob = build_reference_problem(sess, infra, iface, [("quick_charge",1,{}),("equal_share",1e-3,{})], "SOC")
    t=time.time(); r, res = solve_reference_problem(prob); dt=time.time()-t
    print("  SOC+es", res.status, res.iters, dt, res.pcost, res.gap, res.pres, res.dres)
