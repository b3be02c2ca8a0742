import sys, time
sys.path.insert(0, '.')
import numpy as np
from adacharge_amd.acn import Interface
from adacharge_amd import sites
from oracle.ref_problem import build_reference_problem
from oracle.ipm import solve_certified
infra = sites.caltech54()
iface = Interface({"infrastructure_info": infra, "period":5})
for seed in range(4):
    sess = sites.random_sessions(infra, 12, np.random.default_rng(seed))
    for ct, obj in (("LINEAR",[("quick_charge",1,{})]), ("LINEAR",[("quick_charge",1,{}),("equal_share",1e-3,{})]), ("SOC",[("quick_charge",1,{}),("equal_share",1e-3,{})])):
        prob = build_reference_problem(sess, infra, iface, obj, ct)
        t=time.time(); r, res, cert = solve_certified(prob); dt=time.time()-t
        from oracle.ipm import solve_reference_problem
        r0,_ = solve_reference_problem(prob)
        print(seed, ct, len(obj), res.status, res.iters, "%.2fs"%dt, cert, "moved %.2e"%np.abs(r-r0).max(), "obj %.10f"%prob.objective(r))
