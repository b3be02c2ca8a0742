"""Large-site kernel (512 x 48, load flattening): kernel time of repeated launches at several batch sizes."""
import sys, os
sys.path.insert(0, '.')
import numpy as np, torch
from adacharge_amd import ObjectiveComponent, equal_share, load_flattening, total_energy, sites
from adacharge_amd.acn import Interface
from adacharge_amd.backend import SiteHandle, default_options, DeviceBatch
from adacharge_amd.builder import build_batch, scenario_batch, ProblemBatch
infra = sites.synth512(); iface = Interface({"infrastructure_info": infra, "period": 5})
T = 48; ext = 150.0 + 100.0 * np.cos(np.arange(T) / T * 2 * np.pi)
obj = [ObjectiveComponent(load_flattening, 1.0, {"external_signal": ext}), ObjectiveComponent(total_energy, 600.0), ObjectiveComponent(equal_share, 1e-3)]
base = build_batch(sites.snapshot_batch(infra, T, 8, seed=512, min_sessions=200), infra, iface, obj, "SOC")
rng = np.random.default_rng(0)
h = SiteHandle(base.site, 0)
st = torch.cuda.current_stream().cuda_stream
for B in [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048]:
    batch = ProblemBatch.concatenate([scenario_batch(base, rng.lognormal(0, 0.25, size=B // 8), problem=p) for p in range(8)])
    dev = DeviceBatch(batch, "cuda:0")
    o = default_options(eps_abs=1e-6, eps_rel=1e-6)
    ms = []
    for _ in range(4):
        h.solve_device(dev, o, stream=st); torch.cuda.synchronize(); ms.append(h.last_kernel_ms())
    it = dev.iters.cpu().numpy()
    print(f"B={B}: kernel ms {['%.2f' % m for m in ms]} iters {it.mean():.0f} -> {B / min(ms) * 1e3:.0f} QP/s, {min(ms) * 1e3 / it.mean() / max(1, B / 256):.1f} us per iteration per CU-slot", flush=True)
    del dev
