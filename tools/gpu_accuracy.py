"""Dev tool: accuracy of the schedules against the oracle-certified golden optima as a function of the residual
tolerance (27 strictly convex Caltech-shaped cases), with the iteration counts that tolerance costs."""
import sys
sys.path.insert(0, '.')
import numpy as np
from adacharge_amd import AdaptiveChargingOptimization, ObjectiveComponent, equal_share, quick_charge
from tests import helpers as H
g = H.load_golden()
infra, iface = H.caltech_interface()
keys = sorted(k[:-6] for k in g.files if k.endswith("_rates"))
for eps in (1e-9, 1e-8, 1e-7, 1e-6, 1e-5):
    worst, its, viol = 0.0, [], 0.0
    for key in keys:
        sl, meta, exp = H.golden_case(g, key)
        obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, meta["es"])]
        opt = AdaptiveChargingOptimization(obj, iface, constraint_type=meta["ct"], enforce_energy_equality=meta["eq"],
                                           solver_options=dict(reg_rel=0.0, eps_abs=eps, eps_rel=eps))
        rates = opt.solve(sl, infra)
        worst = max(worst, float(np.abs(rates - exp["rates"]).max()))
        its.append(int(opt.last_result.iters[0]))
        ph = np.deg2rad(infra.phases); cm = infra.constraint_matrix
        mag = np.hypot((cm * np.cos(ph)) @ rates, (cm * np.sin(ph)) @ rates) if meta["ct"] == "SOC" else np.abs(cm) @ rates
        viol = max(viol, float((mag - infra.constraint_limits[:, None]).max()))
    print(f"eps {eps:.0e}: max |rate - oracle| {worst:.2e} A ({worst / 32:.1e} of 32 A), max row violation {viol:.1e} A, iterations mean {np.mean(its):.0f} max {max(its)}", flush=True)
