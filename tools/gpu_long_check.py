"""Long-horizon kernel against the C twin (same Anderson memory on both sides) and against the general-shape kernel
(ACNQP_NO_LONG=1 in a second process is not needed: the twin is the checker).  Shapes: caltech54 / jpl52 at horizons
48, 96, 144, 288; prints statuses, iteration counts, max |x - twin| and the launch time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.acn import Interface
from adacharge_amd.backend import SiteHandle, default_options
from adacharge_amd.builder import build_batch
from oracle import admm_port

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ACC = int(sys.argv[2]) if len(sys.argv) > 2 else 5      # Anderson columns on both sides (0 = plain ADMM)
for site_name, infra in (("caltech54", sites.caltech54()), ("jpl52", sites.jpl52())):
    iface = Interface({"infrastructure_info": infra, "period": 5})
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
    for T in (48, 96, 144, 288):
        for ct in ("SOC", "LINEAR"):
            snaps = sites.snapshot_batch(infra, T, B, seed=100 + T, demand_range=(5.0, 60.0))
            batch = build_batch(snaps, infra, iface, obj, ct)
            h = SiteHandle(batch.site, 0)
            opts = default_options(accel_mem=ACC)
            acc = h.accel_columns(batch.Tm, batch.K, opts)
            res = h.solve(batch, opts)
            t0 = time.time()
            ref = admm_port.solve_batch(batch, threads=16, accel_mem=acc)
            dt = time.time() - t0
            dx = np.abs(ref["x"] - res.x).max()
            di = np.abs(ref["iters"].astype(int) - res.iters.astype(int)).max()
            print(f"{site_name} T={T} {ct} K={batch.K} aa={acc} gpu st {np.unique(res.status)} its mean {res.iters.mean():.0f} max {res.iters.max()} "
                  f"kernel {res.kernel_ms:.1f} ms | twin st {np.unique(ref['status'])} its mean {ref['iters'].mean():.0f} ({dt:.1f}s on 16 threads) "
                  f"| max|dx| {dx:.2e} max|d iters| {di}", flush=True)
            h.close()
