"""Quick perf/correctness probe on the GPU (dev tool)."""
import os, sys, time, json
sys.path.insert(0, '.')
import numpy as np
import torch
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.backend import SiteHandle, default_options, DeviceBatch
from adacharge_amd.builder import build_batch
from oracle import admm_port
from tests import helpers as H
infra, iface = H.caltech_interface()
T=12
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
for ct in ("SOC","LINEAR"):
    for B in (256, 2048):
        snaps = sites.snapshot_batch(infra, T, B, seed=20240)
        batch = build_batch(snaps, infra, iface, obj, ct)
        h = SiteHandle(batch.site, 0)
        dev = DeviceBatch(batch, "cuda:0")
        o = default_options(accel_mem=int(os.environ.get('ACCEL', '10')))
        m_eff = h.accel_columns(batch.Tm, batch.K, o)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(2): h.solve_device(dev, o, stream=st)
        torch.cuda.synchronize()
        ms=[]
        for _ in range(5):
            h.solve_device(dev, o, stream=st); ms.append(h.last_kernel_ms())
        it = dev.iters.cpu().numpy(); stt = dev.status.cpu().numpy()
        line = f"{ct} B={B} accel={m_eff} kernel_ms={np.mean(ms):.3f} iters mean {it.mean():.0f} max {it.max()} solved {(stt==1).sum()} us/iter(max) {1e3*np.mean(ms)/it.max():.2f} QP/s {B/np.mean(ms)*1e3:.0f}"
        if B == 256:
            ref = admm_port.solve_batch(batch, threads=16, accel_mem=m_eff)
            x = dev.x.cpu().numpy()
            line += f" | vs port: max|dx| {np.abs(x-ref['x']).max():.2e} iters equal {(it==ref['iters']).mean():.3f}"
        print(line, flush=True)
