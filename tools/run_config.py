"""One device-resident launch of a BASELINE.json configuration other than the bench's (profiling target):

    python3 tools/run_config.py cfg2-caltech|cfg2-jpl|cfg3-site0|cfg3-site3|cfg5|stress144|cfg4 [scenarios]

Everything but `cfg4` is one of bench.py's `other_configs` legs (bench.other_workloads: one definition for the driver's
JSON line and for the profiles): cfg2 = horizon 24, batch 4096; cfg3-siteK = 1024 demand scenarios of one site of
configs[3]; cfg5 = configs[4] shape (synthetic 512 EVSE x 48, load flattening with energy equalities, large-site MFMA
kernel); stress144 = the reference's 54 x 144 stress shape (long-horizon kernel).  `cfg4` = all 8 sites of configs[3] one
after the other on this GPU.  Prints one JSON line with the kernel's HIP-event duration, QP/s, iterations, algorithmic
bytes / flops and the fractions of the roofs."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.acn import Interface
from adacharge_amd.backend import DeviceBatch, SiteHandle, default_options
from adacharge_amd.builder import build_batch, scenario_batch

which = sys.argv[1]
reps = 3
if which == "cfg4":
    # configs[3]: 1024 demand scenarios x 8 sites, horizon 12; on one GPU the 8 site shards run one after the other
    # (on a node: one site per rank).  One JSON line: per-site kernel time, statuses, and the total.
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-3)]
    st = torch.cuda.current_stream().cuda_stream
    rows, tot_ms, tot_n = [], 0.0, 0
    for k, infra in enumerate(sites.eight_sites()):
        iface = Interface({"infrastructure_info": infra, "period": 5})
        rng = np.random.default_rng(500 + k)
        base = build_batch([sites.random_sessions(infra, 12, rng)], infra, iface, obj, "SOC")
        batch = scenario_batch(base, rng.lognormal(0.0, 0.25, size=(S, base.K, base.N)))
        h = SiteHandle(batch.site, 0)
        dev = DeviceBatch(batch, "cuda:0")
        ms = []
        for _ in range(reps):
            h.solve_device(dev, default_options(), stream=st)
            torch.cuda.synchronize()
            ms.append(h.last_kernel_ms())
        it = dev.iters.cpu().numpy(); s_ = dev.status.cpu().numpy()
        rows.append(dict(site=infra.name if hasattr(infra, "name") else k, n_evse=batch.N, kernel_ms=min(ms), iters_mean=float(it.mean()),
                         iters_max=int(it.max()), solved=int((s_ == 1).sum()), inaccurate=int((s_ == 5).sum()), max_iter=int((s_ == 2).sum())))
        tot_ms += min(ms); tot_n += S
        h.close()
    print(json.dumps(dict(config="cfg4", scenarios_per_site=S, problems=tot_n, kernel_ms_total=tot_ms, qps=tot_n / tot_ms * 1e3, sites=rows)))
    raise SystemExit(0)
ALIASES = {"cfg2-caltech": "cfg2_caltech54_T24_b4096", "cfg2-jpl": "cfg2_jpl52_T24_b4096", "cfg5": "cfg4_synth512_T48_b2048",
           "stress144": "stress_caltech54_T144_b256", "cfg3-site3": "cfg3_site3_T12_b1024", "cfg3-site0": "cfg3_site0_T12_b1024"}
import bench   # the workloads are bench.py's `other_configs` legs (one definition for the driver's line and the profiles)
name = ALIASES.get(which, which)
if name not in bench.other_workloads():
    raise SystemExit(__doc__)
print(json.dumps({name: bench.other_configs_leg(torch.device("cuda", 0), only=[name])[name]}))
