"""Wave-per-problem kernel (acn_qp_wave.hpp) against the C twin and against the tiled kernel (run once with
ACNQP_NO_WAVE=1, once without): headline shape caltech54 x 12, SOC and LINEAR.  Dev tool."""
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
T = 12
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
sizes = [int(s) for s in os.environ.get("SIZES", "256,4096,16384").split(",")]
tag = "tiled" if os.environ.get("ACNQP_NO_WAVE") else "wave"
out = {}
for ct in ("SOC", "LINEAR"):
    for B in sizes:
        snaps = sites.snapshot_batch(infra, T, B, seed=20240)
        batch = build_batch(snaps, infra, iface, obj, ct)
        h = SiteHandle(batch.site, 0)
        dev = DeviceBatch(batch, "cuda:0")
        o = default_options()
        st = torch.cuda.current_stream().cuda_stream
        h.solve_device(dev, o, stream=st)
        torch.cuda.synchronize()
        ms = []
        for _ in range(3):
            h.solve_device(dev, o, stream=st); torch.cuda.synchronize(); ms.append(h.last_kernel_ms())
        it = dev.iters.cpu().numpy(); stt = dev.status.cpu().numpy(); x = dev.x.cpu().numpy()
        line = f"{tag} {ct} B={B} kernel_ms={np.min(ms):.3f} iters mean {it.mean():.1f} max {it.max()} solved {(stt==1).sum()} status {np.bincount(stt, minlength=7).tolist()} QP/s {B/np.min(ms)*1e3:.0f}"
        if B == 256:
            o0 = default_options(polish_iters=0)
            h.solve_device(dev, o0, stream=st); torch.cuda.synchronize()
            it0 = dev.iters.cpu().numpy(); x0 = dev.x.cpu().numpy()
            ref = admm_port.solve_batch(batch, threads=16, accel_mem=5)
            line += f" | vs port: max|dx| {np.abs(x0-ref['x']).max():.2e} iters equal {(it0==ref['iters']).mean():.3f} mean {it0.mean():.1f} vs {ref['iters'].mean():.1f}"
        if B == 256: np.save(f"gpurun_out/wave_{tag}_{ct}_{B}.npy", x)
        print(line, flush=True)
