"""Large-site (stream) kernel against the C twin: plain ADMM on both sides, wide site N = 128 and the 512 x 48
load-flattening shape; prints iteration counts, max |x - twin| and the launch time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, load_flattening, total_energy, sites
from adacharge_amd.acn import Interface
from adacharge_amd.backend import SiteHandle, default_options
from adacharge_amd.builder import build_batch
from oracle import admm_port

which = sys.argv[1] if len(sys.argv) > 1 else "wide"
if which == "wide":
    infra = sites.wide128()
    iface = Interface({"infrastructure_info": infra, "period": 5})
    snaps = sites.snapshot_batch(infra, 12, 8, seed=77, min_sessions=40)
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-3)]
    opts = dict()
else:
    infra = sites.synth512()
    iface = Interface({"infrastructure_info": infra, "period": 5})
    T = 48
    ext = 150.0 + 100.0 * np.cos(np.arange(T) / T * 2 * np.pi)
    obj = [ObjectiveComponent(load_flattening, 1.0, {"external_signal": ext}), ObjectiveComponent(total_energy, 600.0),
           ObjectiveComponent(equal_share, 1e-3)]
    snaps = sites.snapshot_batch(infra, T, int(sys.argv[2]) if len(sys.argv) > 2 else 4, seed=512, min_sessions=200)
    opts = dict(eps_abs=1e-6, eps_rel=1e-6)
for ct in ("LINEAR", "SOC"):
    batch = build_batch(snaps, infra, iface, obj, ct)
    h = SiteHandle(batch.site, 0)
    print(which, ct, "N", batch.N, "Tm", batch.Tm, "MR rows", batch.site.Mg, "accel", h.accel_columns(batch.Tm, batch.K, default_options()), flush=True)
    res = h.solve(batch, default_options(**opts))
    print(" gpu status", np.unique(res.status, return_counts=True), "iters", res.iters[:8], "kernel_ms %.3f" % res.kernel_ms, flush=True)
    nref = min(batch.B, 4)
    import copy
    sb = copy.copy(batch); sb.B = nref
    for name in ("T", "lb", "ub", "q", "pdiag", "lf", "s_off", "s_len", "s_cap", "s_eq", "dc", "dfloor"):
        setattr(sb, name, getattr(batch, name)[:nref])
    ref = admm_port.solve_batch(sb, threads=4, accel_mem=0, **opts)
    print(" ref status", ref["status"], "iters", ref["iters"], "max|dx| %.3e" % np.abs(ref["x"] - res.x[:nref]).max(),
          "obj rel %.2e" % np.abs((ref["obj"] - res.obj[:nref] + 0) / ref["obj"]).max() if not batch.site.has_flat else "", flush=True)
    h.close()
