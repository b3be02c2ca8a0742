"""A/B of the LDS-resident variant of the long-horizon kernel against the register-resident tiled kernel on the shape
it serves (two column tiles x two row tiles: jpl52 at horizon 24; ACNQP_LDS_LONG=0 routes it back to the tiled kernel):
kernel time, iterations, parity of the schedules.  (The round-2 A/B over all four small shapes needed extra
instantiations that are no longer built: DESIGN.md section 3.3 keeps the numbers.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.acn import Interface
import torch
from adacharge_amd.backend import DeviceBatch, SiteHandle, default_options
from adacharge_amd.builder import build_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for site_name, T in (("jpl52", 24),):
    infra = getattr(sites, site_name)()
    iface = Interface({"infrastructure_info": infra, "period": 5})
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
    batch = build_batch(sites.snapshot_batch(infra, T, B, seed=20240), infra, iface, obj, "SOC")
    out = {}
    for mode in ("", "1"):   # "": tiled kernel (switch off), "1": LDS-resident long-horizon kernel (default routing)
        os.environ["ACNQP_LDS_LONG"] = "1" if mode else "0"
        h = SiteHandle(batch.site, 0)
        dev = DeviceBatch(batch, "cuda:0")          # one device-resident launch: its HIP-event time is the kernel time
        for _ in range(2):
            h.solve_device(dev, default_options(), stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        ms = h.last_kernel_ms()
        res = h.solve(batch, default_options())     # the same through the host entry, for statuses and schedules
        out[mode] = res
        print(f"{site_name} T={T} B={B} {'lds-long' if mode else 'tiled   '}: kernel {ms:8.2f} ms  {B / ms:8.1f} kQP/s  its mean {res.iters.mean():.0f} max {res.iters.max()}"
              f"  status {dict(zip(*np.unique(res.status, return_counts=True)))}", flush=True)
        h.close()
    both = (out[""].status == 1) & (out["1"].status == 1)
    print("   max |dx| on problems solved by both: %.2e" % np.abs(out[""].x[both] - out["1"].x[both]).max(), flush=True)
