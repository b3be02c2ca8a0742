import sys
sys.path.insert(0, '.')
import numpy as np
from tests.acn_testing import *
from oracle.ref_problem import build_reference_problem
from oracle.ipm import solve_reference_problem
sd = session_generator(2,[0]*2,[12]*2,[3.3]*2,[3.3]*2,[32]*2)
infra = single_phase_single_constraint(2, 64)
iface = TestingInterface({"active_sessions": sd, "infrastructure_info": infra, "current_time":0, "period":5})
prob = build_reference_problem(iface.active_sessions(), iface.infrastructure_info(), iface, [("quick_charge",1,{})], sys.argv[1], False)
r, res = solve_reference_problem(prob, verbose=True, max_iter=30)
print(res.status, r[0])
