"""One device-resident launch of a BASELINE.json configuration other than the bench's (profiling target):

    python3 tools/run_config.py cfg2-caltech|cfg2-jpl|cfg4|cfg5|stress144 [batch]

cfg2: horizon 24, batch 4096, fp64 (tiled kernel, two column tiles); cfg5: synthetic 512 EVSE x 48, load_flattening
(large-site MFMA kernel); stress144: the reference's N = 54 x T = 144 stress LP shape (general-shape kernel).
Prints one JSON line with the kernel's HIP-event duration, QP/s, iterations and the algorithmic bytes."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from adacharge_amd import ObjectiveComponent, equal_share, load_flattening, quick_charge, total_energy, sites
from adacharge_amd.acn import Interface
from adacharge_amd.backend import DeviceBatch, SiteHandle, default_options
from adacharge_amd.builder import ProblemBatch, build_batch, scenario_batch

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
if which.startswith("cfg2"):
    infra = sites.caltech54() if which.endswith("caltech") else sites.jpl52()
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    iface = Interface({"infrastructure_info": infra, "period": 5})
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-3)]
    batch = build_batch(sites.snapshot_batch(infra, 24, B, seed=31), infra, iface, obj, "SOC")
    opts = default_options()
elif which == "cfg5":
    infra = sites.synth512()
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    iface = Interface({"infrastructure_info": infra, "period": 5})
    T = 48
    ext = 150.0 + 100.0 * np.cos(np.arange(T) / T * 2 * np.pi)
    obj = [ObjectiveComponent(load_flattening, 1.0, {"external_signal": ext}), ObjectiveComponent(total_energy, 600.0),
           ObjectiveComponent(equal_share, 1e-3)]
    base = build_batch(sites.snapshot_batch(infra, T, 8, seed=512, min_sessions=200), infra, iface, obj, "SOC")
    rng = np.random.default_rng(0)
    batch = ProblemBatch.concatenate([scenario_batch(base, rng.lognormal(0, 0.25, size=B // 8), problem=p) for p in range(8)])
    opts = default_options(eps_abs=1e-6, eps_rel=1e-6)
elif which == "stress144":
    infra = sites.caltech54()
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    iface = Interface({"infrastructure_info": infra, "period": 5})
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
    batch = build_batch(sites.snapshot_batch(infra, 144, B, seed=144, demand_range=(5.0, 60.0)), infra, iface, obj, "SOC")
    opts = default_options()
else:
    raise SystemExit(__doc__)
h = SiteHandle(batch.site, 0)
dev = DeviceBatch(batch, "cuda:0")
st = torch.cuda.current_stream().cuda_stream
ms = []
for _ in range(reps):
    h.solve_device(dev, opts, stream=st)
    torch.cuda.synchronize()
    ms.append(h.last_kernel_ms())
it = dev.iters.cpu().numpy(); s = dev.status.cpu().numpy()
N, Tm, K = batch.N, batch.Tm, batch.K
n = N * Tm
per_qp_io = 8 * 4 * n + 16 * K * N + 45
out = dict(config=which, batch=batch.B, n_evse=N, horizon=Tm, site_rows=batch.site.Mg, kernel_ms=min(ms), qps=batch.B / min(ms) * 1e3,
           iters_mean=float(it.mean()), iters_max=int(it.max()), solved=int((s == 1).sum()), inaccurate=int((s == 5).sum()),
           anderson_columns=h.accel_columns(Tm, K, opts), io_bytes_per_qp=per_qp_io)
if which == "cfg5":   # streamed state: SURVEY 8d B_iter = w (3 n + 6 m), m = n + S + R + K rows
    S_rows = int((batch.s_len[0] > 0).sum()); R = 2 * batch.site.M * Tm
    b_iter = 8 * (3 * n + 6 * (n + S_rows + R))
    kernel_iter = 9 * 8 * (16 * ((N + 15) // 16)) * (16 * ((Tm + 15) // 16))   # what the kernel moves: 9 padded arrays
    alg = batch.B * (per_qp_io + float(it.mean()) * b_iter)
    out.update(b_iter_bytes=b_iter, kernel_stream_bytes_per_iter=kernel_iter, algorithmic_bytes_per_launch=alg,
               achieved_GBs=alg / (min(ms) * 1e-3) / 1e9, hbm_frac_of_8TBs=alg / (min(ms) * 1e-3) / 8e12)
print(json.dumps(out))
