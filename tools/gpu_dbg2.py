import sys, os, ctypes as C
os.environ['ACNQP_LIBRARY']=os.path.abspath('adacharge_amd/lib/libacn_qp_hip_stamps.so')
sys.path.insert(0, '.')
import numpy as np
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.backend import SiteHandle, default_options
from adacharge_amd.builder import build_batch
from oracle import admm_port
from tests import helpers as H
import copy
infra, iface = H.caltech_interface()
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
snaps = sites.snapshot_batch(infra, 12, 2048, seed=20240)
for ct in ("SOC",):
    batch = build_batch(snaps, infra, iface, obj, ct)
    h = SiteHandle(batch.site, 0)
    res = h.solve(batch, default_options())
    from adacharge_amd.backend import load_library
    buf = (C.c_ulonglong * (1024*16*8))(); load_library().acnqp_debug_read_stamps(buf, 1024*16*8)
    print("waterfill calls(wave-level)", buf[1024*16*8-3], "passes", buf[1024*16*8-2], "guard>80 events", buf[1024*16*8-1])
    bad = np.nonzero(res.status != 1)[0]
    print(ct, "unsolved", bad, "iters", res.iters[bad], "pri", res.pri_res[bad], "dua", res.dua_res[bad])
    top = np.argsort(-res.iters)[:6]
    sb = build_batch([snaps[k] for k in top], infra, iface, obj, ct, site=batch.site)
    ref = admm_port.solve_batch(sb, threads=8)
    print("top gpu iters", res.iters[top], "port iters", ref['iters'], "port status", ref['status'])
    for j,k in enumerate(top):
        print("  ", k, "max|dx| %.2e"%np.abs(res.x[k]-ref['x'][j]).max(), "obj gpu %.9f port %.9f"%(res.obj[k], ref['obj'][j]))
