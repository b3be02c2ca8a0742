"""Where the time of ONE 256-snapshot acnqp_solve_batch call goes (bench.py's strict_batch256 leg):
total per call with pinned / plain result arrays, the marshalling alone, the kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.acn import Interface
from adacharge_amd.backend import SiteHandle, default_options, _check
from adacharge_amd.builder import build_batch
infra = sites.caltech54(); iface = Interface({"infrastructure_info": infra, "period": 5})
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
batch = build_batch(sites.snapshot_batch(infra, 12, 256, seed=20240), infra, iface, obj, "SOC")
h = SiteHandle(batch.site, 0); o = default_options()
def med(f, n=40):
    f(); t = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(t))
print("solve(pinned_results=True)  %.2f ms" % med(lambda: h.solve(batch, o, pinned_results=True)))
print("solve(pinned_results=False) %.2f ms" % med(lambda: h.solve(batch, o, pinned_results=False)))
print("_marshal(pinned) alone      %.2f ms" % med(lambda: h._marshal(batch, True)))
print("_marshal(plain) alone       %.2f ms" % med(lambda: h._marshal(batch, False)))
p, r, res, keep = h._marshal(batch, True)
print("C call alone (pinned results, marshalled once) %.2f ms" % med(lambda: _check(h._lib.acnqp_solve_batch(h._h, C.byref(p), C.byref(o), C.byref(r)), "x")))
print("kernel %.2f ms" % h.last_kernel_ms())
