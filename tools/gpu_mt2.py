"""Two row tiles on one column tile (caltech54 x 12 with a peak limit: 17 site rows -> MR = 32): the register-resident
kernel at two workgroups per CU (256 registers, spills) against one per CU (512 registers, none).
    python tools/gpu_mt2.py [batch]      (ACNQP_OCC1=1 forces the one-workgroup build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.acn import Interface
from adacharge_amd.backend import DeviceBatch, SiteHandle, default_options
from adacharge_amd.builder import build_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
infra = sites.caltech54(); iface = Interface({"infrastructure_info": infra, "period": 5})
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-3)]
snaps = sites.snapshot_batch(infra, 12, B, seed=99)
batch = build_batch(snaps, infra, iface, obj, "SOC", peak_limits=[600.0] * B)
h = SiteHandle(batch.site, 0); dev = DeviceBatch(batch, "cuda:0")
ms = []
for _ in range(4):
    h.solve_device(dev, default_options(), stream=torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize(); ms.append(h.last_kernel_ms())
it = dev.iters.cpu().numpy(); st = dev.status.cpu().numpy()
print("OCC1" if os.environ.get("ACNQP_OCC1") else "OCC2", "site rows", batch.site.Mg, "kernel ms", [round(m, 2) for m in ms], "iters mean %.0f max %d" % (it.mean(), it.max()), "solved", int((st == 1).sum()), "/", B)
