import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from adacharge_amd import ObjectiveComponent, equal_share, load_flattening, total_energy, sites
from adacharge_amd.acn import Interface
from adacharge_amd.backend import SiteHandle, default_options, DeviceBatch
from adacharge_amd.builder import build_batch, scenario_batch
infra = sites.synth512(); iface = Interface({"infrastructure_info": infra, "period": 5})
T=48; ext = 150.0 + 100.0*np.cos(np.arange(T)/T*2*np.pi)
obj = [ObjectiveComponent(load_flattening, 1.0, {"external_signal": ext}), ObjectiveComponent(total_energy, 600.0), ObjectiveComponent(equal_share, 1e-3)]
base = build_batch(sites.snapshot_batch(infra, T, 8, seed=512, min_sessions=200), infra, iface, obj, "SOC")
rng=np.random.default_rng(0)
for B in (256, 2048):
    parts=[scenario_batch(base, rng.lognormal(0,0.25,size=B//8), problem=p) for p in range(8)]
    import copy
    big=copy.copy(parts[0]); big.B=B
    for name in ("T","lb","ub","q","pdiag","lf","s_off","s_len","s_cap","s_eq","dc","dfloor","const","presolve_status"):
        v=[getattr(p,name) for p in parts]
        setattr(big,name,None if v[0] is None else np.concatenate(v))
    h=SiteHandle(big.site,0); dev=DeviceBatch(big,"cuda:0"); o=default_options(eps_abs=1e-6, eps_rel=1e-6, reg_rel=0.0)
    st=torch.cuda.current_stream().cuda_stream
    h.solve_device(dev,o,stream=st); torch.cuda.synchronize()
    h.solve_device(dev,o,stream=st); ms=h.last_kernel_ms()
    it=dev.iters.cpu().numpy(); s=dev.status.cpu().numpy()
    print(f"cfg5 shape B={B}: kernel {ms:.1f} ms -> {B/ms*1e3:.0f} QP/s, iters mean {it.mean():.0f} max {it.max()}, solved {(s==1).sum()}/{B}", flush=True)
    h.close(); del dev
