"""Dev tool: do two batches launched on two HIP streams overlap?  Tries pairs of torch pool streams."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.backend import SiteHandle, default_options, DeviceBatch
from adacharge_amd.builder import build_batch
from tests import helpers as H
infra, iface = H.caltech_interface()
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
o = default_options()
slots = []
for s in range(2):
    b = build_batch(sites.snapshot_batch(infra, 12, 256, seed=20240 + 1000 * s), infra, iface, obj, "SOC")
    slots.append((SiteHandle(b.site, 0), DeviceBatch(b, "cuda:0")))
streams = [torch.cuda.Stream() for _ in range(8)]
print("stream handles", [hex(s.cuda_stream) for s in streams])
def run(pair, steps=40):
    for i in range(4):
        h, d = slots[i % 2]; h.solve_device(d, o, stream=pair[i % 2].cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        h, d = slots[i % 2]; h.solve_device(d, o, stream=pair[i % 2].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for j in range(1, 8):
    print("pair (0,%d): %.3f ms/step" % (j, run((streams[0], streams[j]))))
cur = torch.cuda.current_stream()
print("pair (default, 0): %.3f ms/step" % run((cur, streams[0])))
