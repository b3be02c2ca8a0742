import sys, os, ctypes as C
sys.path.insert(0, '.')
os.environ["ACNQP_LIBRARY"] = os.path.abspath("adacharge_amd/lib/libacn_qp_hip_stamps.so")
import numpy as np
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.backend import SiteHandle, default_options, load_library
from adacharge_amd.builder import build_batch
from tests import helpers as H
infra, iface = H.caltech_interface()
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
names = ["r0+P+wh(8mfma)","barrier","psum+eh+hh","xt mfma+clip","xpose back+y1","siterows","check","xpose fwd","newton-other","anderson event","aa:dots+reduce","aa:solve"]
for ct in ("SOC","LINEAR"):
    NB = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    snaps = sites.snapshot_batch(infra, 12, NB, seed=20240)
    batch = build_batch(snaps, infra, iface, obj, ct)
    h = SiteHandle(batch.site, 0)
    res = h.solve(batch, default_options())
    lib = load_library()
    buf = (C.c_ulonglong * (1024*16*12))()
    lib.acnqp_debug_read_stamps(buf, 1024*16*12)
    nb = min(NB, 1024)   # the diagnostic build keeps the counters of the first 1024 workgroups
    st = np.array(buf, dtype=np.float64).reshape(1024,16,12)[:nb, :4]
    per_iter = st / res.iters[:nb,None,None]
    print(ct, "kernel_ms %.2f"%res.kernel_ms, "iters max", res.iters.max())
    tot = per_iter.sum(-1).mean()
    for k,n in enumerate(names):
        print("   %-12s %8.0f cycles/iter (wave mean)  w0 %.0f w3 %.0f  %.1f%%"%(n, per_iter[:,:,k].mean(), per_iter[:,0,k].mean(), per_iter[:,3,k].mean(), 100*per_iter[:,:,k].mean()/tot))
    print("   total %.0f cycles/iter (s_memtime ticks)"%tot)
