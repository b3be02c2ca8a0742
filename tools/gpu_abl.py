import sys, os
lib = sys.argv[1]
os.environ["ACNQP_LIBRARY"] = os.path.abspath(lib)
sys.path.insert(0, '.')
import numpy as np, torch
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.backend import SiteHandle, default_options, DeviceBatch
from adacharge_amd.builder import build_batch
from tests import helpers as H
infra, iface = H.caltech_interface()
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
snaps = sites.snapshot_batch(infra, 12, 256, seed=20240)
batch = build_batch(snaps, infra, iface, obj, "SOC")
h = SiteHandle(batch.site, 0)
dev = DeviceBatch(batch, "cuda:0")
o = default_options(max_iter=2000, eps_abs=0.0, eps_rel=0.0)   # fixed 2000 iterations for every problem
st = torch.cuda.current_stream().cuda_stream
for _ in range(2): h.solve_device(dev, o, stream=st)
torch.cuda.synchronize()
ms=[]
for _ in range(3):
    h.solve_device(dev, o, stream=st); ms.append(h.last_kernel_ms())
print(lib, "kernel_ms %.3f -> %.3f us/iter"%(np.mean(ms), 1e3*np.mean(ms)/2000))
