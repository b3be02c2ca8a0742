"""Per-phase s_memtime shares of the long-horizon kernel; needs a diagnostic build:
hipcc ... -DACNQP_STAMPS -o adacharge_amd/lib/libacn_qp_hip_stamps.so"""
import sys, os, ctypes as C
sys.path.insert(0, '.')
os.environ["ACNQP_LIBRARY"] = os.path.abspath("adacharge_amd/lib/libacn_qp_hip_stamps.so")
import numpy as np
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.acn import Interface
from adacharge_amd.backend import SiteHandle, default_options, load_library
from adacharge_amd.builder import build_batch
SITE = os.environ.get("SITE", "caltech54")
infra = getattr(sites, SITE)()
iface = Interface({"infrastructure_info": infra, "period": 5})
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
names = ["1a site tiles", "barrier", "1b tiles", "AA event", "barrier", "rows: fill", "rows: y1/r0", "barrier", "check"]
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for T in [int(x) for x in os.environ.get("HORIZONS", "48,144,288").split(",")]:
    snaps = sites.snapshot_batch(infra, T, NB, seed=100 + T, demand_range=(5.0, 60.0)) if T > 32 else sites.snapshot_batch(infra, T, NB, seed=20240)
    batch = build_batch(snaps, infra, iface, obj, "SOC")
    h = SiteHandle(batch.site, 0)
    res = h.solve(batch, default_options(accel_mem=int(os.environ.get("AA", "5"))))
    lib = load_library()
    buf = (C.c_ulonglong * (1024 * 16 * 12))()
    lib.acnqp_debug_read_stamps_long(buf, 1024 * 16 * 12)
    nw = 8
    NS = min(NB, 1024)   # the stamp buffer covers the first 1024 workgroups
    st = np.array(buf, dtype=np.float64).reshape(1024, 16, 12)[:NS, :nw]
    per_iter = st / res.iters[:NS, None, None]
    tot = per_iter.sum(-1).mean()
    print("T", T, "kernel_ms %.2f" % res.kernel_ms, "iters max", res.iters.max(), "us/iter of slowest %.1f" % (1e3 * res.kernel_ms / res.iters.max()))
    for k, n in enumerate(names):
        print("   %-12s %8.1f ticks/iter (wave mean)  w0 %.1f w%d %.1f  %.1f%%" % (n, per_iter[:, :, k].mean(), per_iter[:, 0, k].mean(), nw - 1, per_iter[:, nw - 1, k].mean(), 100 * per_iter[:, :, k].mean() / tot))
    print("   total %.1f ticks/iter (s_memtime, 100 MHz)" % tot)
    h.close()
