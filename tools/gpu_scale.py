import sys, os
sys.path.insert(0, '.')
import numpy as np, torch
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.backend import SiteHandle, default_options, DeviceBatch
from adacharge_amd.builder import build_batch
from tests import helpers as H
infra, iface = H.caltech_interface()
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
st = torch.cuda.current_stream().cuda_stream
for prec in (64, 32):
  for B in (256, 512, 1024, 2048, 4096):
    snaps = sites.snapshot_batch(infra, 12, B, seed=20240)
    batch = build_batch(snaps, infra, iface, obj, "SOC")
    h = SiteHandle(batch.site, 0)
    dev = DeviceBatch(batch, "cuda:0")
    o = default_options(max_iter=1000, eps_abs=0.0, eps_rel=0.0, precision=prec)   # fixed 1000 iterations
    h.solve_device(dev, o, stream=st); torch.cuda.synchronize()
    ms=[]
    for _ in range(3):
        h.solve_device(dev, o, stream=st); ms.append(h.last_kernel_ms())
    print(f"fp{prec} B={B} kernel_ms={np.mean(ms):.3f}  us per (QP-iteration)={1e3*np.mean(ms)/1000/B*256:.3f} (x256 CUs)  QP-iters/s={B*1000/np.mean(ms)*1e3:.3e}", flush=True)
