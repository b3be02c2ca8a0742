import sys, os
sys.path.insert(0, '.')
import numpy as np, torch
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.acn import Interface
from adacharge_amd.backend import SiteHandle, default_options, DeviceBatch
from adacharge_amd.builder import build_batch
for name, T in (("jpl52", 24), ("caltech54", 24), ("caltech54", 12)):
    infra = getattr(sites, name)(); iface = Interface({"infrastructure_info": infra, "period": 5})
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-3)]
    batch = build_batch(sites.snapshot_batch(infra, T, 4096, seed=31), infra, iface, obj, "SOC")
    h = SiteHandle(batch.site, 0); dev = DeviceBatch(batch, "cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    o = default_options(accel_mem=0, max_iter=400, eps_abs=0.0, eps_rel=0.0)   # fixed 400 plain iterations
    ms = []
    for _ in range(3):
        h.solve_device(dev, o, stream=st); torch.cuda.synchronize(); ms.append(h.last_kernel_ms())
    print(f"{name} T={T} FORCE_STREAM={os.environ.get('ACNQP_FORCE_STREAM')}: 4096 problems x 400 plain iterations: {min(ms):.2f} ms -> {4096*400/min(ms)/1e3:.3e} problem-iterations/s", flush=True)
