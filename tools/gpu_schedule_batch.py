"""Where the time of AdaptiveSchedulingAlgorithm.schedule_batch goes (256 snapshots, 54 EVSE x 12): the array path
(SessionTable in, arrays out) against the object path (SessionInfo lists in, dicts out) and the kernel alone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from adacharge_amd import AdaptiveSchedulingAlgorithm, ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.acn import Interface
from adacharge_amd.session_table import SessionTable

infra = sites.caltech54()
iface = Interface({"infrastructure_info": infra, "period": 5, "current_time": 0})
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lists = sites.snapshot_batch(infra, 12, B, seed=3)
table = sites.snapshot_table(infra, 12, B, seed=3)
for name, kw in (("continuous", {}), ("quantize", dict(quantize=True)), ("quantize+reallocate+uninterrupted", dict(quantize=True, reallocate=True, uninterrupted_charging=True))):
    alg = AdaptiveSchedulingAlgorithm(obj, **kw)
    alg.register_interface(iface)
    alg.schedule_batch(table, as_arrays=True)   # warm: handle, staging
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); r, st = alg.schedule_batch(table, as_arrays=True); t.append(time.perf_counter() - t0)
    k_ms = alg._optimizer and None
    from adacharge_amd import adaptive_charging_optimization as aco
    t2 = []
    for _ in range(3):
        t0 = time.perf_counter(); out = alg.schedule_batch(lists); t2.append(time.perf_counter() - t0)
    print(f"{name:36s} B={B}: table->arrays {1e3*min(t):7.2f} ms ({B/min(t):8.0f} schedules/s) | SessionInfo lists->dicts {1e3*min(t2):7.2f} ms | solved {(st==1).sum()}/{B}", flush=True)
# pieces of the array path
from adacharge_amd import session_table as stt
from adacharge_amd.builder import build_batch_from_table, make_site
from adacharge_amd.backend import SiteHandle, default_options
from adacharge_amd import postprocessing as pp
site = make_site(infra, "SOC")
h = SiteHandle(site, 0)
def best(f, n=5):
    v = []
    for _ in range(n):
        t0 = time.perf_counter(); r = f(); v.append(time.perf_counter() - t0)
    return 1e3 * min(v), r
t_tab, _ = best(lambda: SessionTable.from_sessions(lists, infra))
t_pre, tb = best(lambda: stt.apply_minimum_charging_rate(stt.enforce_pilot_limit(table, infra), infra, 5))
t_bld, batch = best(lambda: build_batch_from_table(tb, infra, iface, obj, "SOC", site=site))
t_slv, res = best(lambda: h.solve(batch, default_options()))
t_q, _ = best(lambda: pp.project_into_discrete_feasible_pilots_batch(res.x, infra))
t_r, _ = best(lambda: pp.diff_based_reallocation_batch(res.x, tb, infra, iface))
print(f"pieces (ms): SessionInfo->table {t_tab:.2f} | preprocess {t_pre:.2f} | build {t_bld:.2f} | solve (H2D+kernel+D2H) {t_slv:.2f} "
      f"[kernel {res.kernel_ms:.2f}] | quantise {t_q:.2f} | reallocate {t_r:.2f}")
