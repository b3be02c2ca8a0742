"""Run-to-run determinism probe of every kernel family: the same batch solved three times must return the same bits
(a workgroup's result may not depend on timing).  ACNQP_NO_RZL=1 / ACNQP_NO_LONG=1 select the diagnostic routes.
    python tools/gpu_determinism.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from adacharge_amd import ObjectiveComponent, equal_share, load_flattening, quick_charge, sites
from adacharge_amd.acn import Interface
from adacharge_amd.backend import SiteHandle, default_options
from adacharge_amd.builder import build_batch

def probe(name, infra, T, B, obj, seed, eq=False, **kw):
    iface = Interface({"infrastructure_info": infra, "period": 5})
    batch = build_batch(sites.snapshot_batch(infra, T, B, seed=seed, **kw), infra, iface, obj, "SOC", eq)
    h = SiteHandle(batch.site, 0)
    runs = [h.solve(batch, default_options()) for _ in range(3)]
    h.close()
    same = all(np.array_equal(runs[0].x, r.x) and np.array_equal(runs[0].iters, r.iters) for r in runs[1:])
    dmax = max(float(np.abs(runs[0].x - r.x).max()) for r in runs[1:])
    print("%-34s %s  statuses %s  iters mean %.0f  max|dx| between runs %.1e" % (
        name, "same bits" if same else "DIFFERENT", np.unique(runs[0].status).tolist(), runs[0].iters.mean(), dmax), flush=True)
    return same

qc12 = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
qc3 = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-3)]
ok = True
ok &= probe("tiled 54x12 (headline) x2048", sites.caltech54(), 12, 2048, qc12, 1)
ok &= probe("tiled 54x24 x1024", sites.caltech54(), 24, 1024, qc3, 2)
ok &= probe("long LDS-resident jpl52x24 x512", sites.jpl52(), 24, 512, qc3, 3)
ok &= probe("long 54x96 x64", sites.caltech54(), 96, 64, qc12, 196, demand_range=(5.0, 60.0))
ok &= probe("long 54x144 x64", sites.caltech54(), 144, 64, qc12, 244, demand_range=(5.0, 60.0))
ok &= probe("long 54x288 x16", sites.caltech54(), 288, 16, qc12, 388, demand_range=(5.0, 60.0))
ok &= probe("stream wide128x12 x256", sites.wide128(), 12, 256, qc3, 5, min_sessions=40)
ok &= probe("stream synth512x48 x32", sites.synth512(), 48, 32, qc3, 6, min_sessions=200)
ok &= probe("general 54x320 x8", sites.caltech54(), 320, 8, qc12, 7, demand_range=(5.0, 60.0))
print("ALL SAME" if ok else "NONDETERMINISTIC ROUTES PRESENT")
raise SystemExit(0 if ok else 1)
