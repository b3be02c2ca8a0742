"""Dev tool: one-launch timings of the other BASELINE.json configurations on one GPU (device-resident inputs).
Not a bench line: orientation numbers for DESIGN.md."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, load_flattening, total_energy, sites
from adacharge_amd.acn import Interface
from adacharge_amd.backend import SiteHandle, default_options, DeviceBatch
from adacharge_amd.builder import build_batch

st = torch.cuda.current_stream().cuda_stream


def run(name, batch, opts, reps=3):
    h = SiteHandle(batch.site, 0)
    dev = DeviceBatch(batch, "cuda:0")
    h.solve_device(dev, opts, stream=st); torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        h.solve_device(dev, opts, stream=st); ms.append(h.last_kernel_ms())
    it = dev.iters.cpu().numpy(); s = dev.status.cpu().numpy()
    print(f"{name}: B={batch.B} kernel_ms={np.mean(ms):.2f} QP/s={batch.B / np.mean(ms) * 1e3:.0f} iters mean {it.mean():.0f} max {it.max()} "
          f"solved {(s == 1).sum()}/{batch.B} anderson={h.accel_columns(batch.Tm, batch.K, opts)}", flush=True)
    h.close()


qc = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
# configs[2]: Caltech + JPL, horizon 24, batch 4096, fp32
for site_name in ("caltech54", "jpl52"):
    infra = getattr(sites, site_name)()
    iface = Interface({"infrastructure_info": infra, "period": 5})
    batch = build_batch(sites.snapshot_batch(infra, 24, 4096, seed=3), infra, iface, qc, "SOC")
    run(f"cfg3 {site_name} T=24 fp64 eps 1e-8", batch, default_options())
# configs[3]: 1024 demand scenarios of one site snapshot (one GPU's share of 8 sites x 1024)
infra = sites.caltech54()
iface = Interface({"infrastructure_info": infra, "period": 5})
base = sites.snapshot_batch(infra, 12, 1, seed=11)[0]
scen = sites.demand_scenarios(base, 1024, np.random.default_rng(5))
run("cfg4 1024 demand scenarios, 54x12 fp64", build_batch(scen, infra, iface, qc, "SOC"), default_options())
# configs[4] shape: 512 EVSE x 48, load flattening (general kernel), a small batch
infra = sites.synth512()
iface = Interface({"infrastructure_info": infra, "period": 5})
T = 48
ext = 150.0 + 100.0 * np.cos(np.arange(T) / T * 2 * np.pi)
obj = [ObjectiveComponent(load_flattening, 1.0, {"external_signal": ext}), ObjectiveComponent(total_energy, 600.0),
       ObjectiveComponent(equal_share, 1e-3)]
batch = build_batch(sites.snapshot_batch(infra, T, 64, seed=512, min_sessions=200), infra, iface, obj, "SOC")
run("cfg5 shape 512x48 load_flattening fp64 eps 1e-6 (general kernel)", batch, default_options(eps_abs=1e-6, eps_rel=1e-6, reg_rel=0.0), reps=1)
