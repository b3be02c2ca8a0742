"""Long-horizon kernel at batch: caltech54 / jpl52 at horizons 48, 96 (x and r0 / zh in LDS), 144 (r0 / zh in LDS),
one device-resident launch each; ACNQP_NO_XSL=1 / ACNQP_NO_RZL=1 move the arrays back to the workspace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.acn import Interface
from adacharge_amd.backend import DeviceBatch, SiteHandle, default_options
from adacharge_amd.builder import build_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for site_name, T in (("caltech54", 48), ("caltech54", 96), ("jpl52", 96), ("caltech54", 144)):
    infra = getattr(sites, site_name)()
    iface = Interface({"infrastructure_info": infra, "period": 5})
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
    batch = build_batch(sites.snapshot_batch(infra, T, B, seed=100 + T, demand_range=(5.0, 60.0)), infra, iface, obj, "SOC")
    h = SiteHandle(batch.site, 0)
    dev = DeviceBatch(batch, "cuda:0")
    for _ in range(2):
        h.solve_device(dev, default_options(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    it = dev.iters.cpu().numpy(); st = dev.status.cpu().numpy()
    ms = h.last_kernel_ms()
    print(f"{site_name} T={T} B={B}: kernel {ms:8.2f} ms  {B / ms:6.2f} kQP/s  its mean {it.mean():.0f} max {it.max()}  solved {(st == 1).sum()}", flush=True)
    h.close()
