import sys; sys.path.insert(0,'.')
import numpy as np
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge
from adacharge_amd.backend import SiteHandle, default_options
from adacharge_amd.builder import build_batch
from oracle import admm_port
from tests.acn_testing import TestingInterface, session_generator, single_phase_single_constraint, three_phase_balanced_network
N,T=54,144
obj=[ObjectiveComponent(quick_charge)]
for name, net in (('1ph', single_phase_single_constraint(54, 32*54/3)), ('3ph', three_phase_balanced_network(18, 32*54/3))):
    for ct in ('LINEAR','SOC'):
        sd = session_generator(N, [0]*N, [T]*N, [10]*N, [10]*N, [32]*N)
        iface = TestingInterface({"active_sessions": sd, "infrastructure_info": net, "current_time": 0, "period": 5})
        b = build_batch([iface.active_sessions()], iface.infrastructure_info(), iface, obj, ct)
        h = SiteHandle(b.site, 0)
        out=[]
        for m in (0,5):
            o = default_options(eps_abs=1e-9, eps_rel=1e-9, max_iter=100000, accel_mem=m)
            r = h.solve(b, o)
            ref = admm_port.solve_batch(b, threads=1, eps_abs=1e-9, eps_rel=1e-9, max_iter=100000, accel_mem=m)
            out.append('m%d gpu %d its %.0f ms st %d | port %d its st %d dx %.1e'%(m, r.iters[0], r.kernel_ms, r.status[0], ref['iters'][0], ref['status'][0], np.abs(r.x-ref['x']).max()))
        print(name, ct, ' || '.join(out), flush=True)
