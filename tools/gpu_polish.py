"""The device-side polish on the workloads where stragglers set the launch time: per leg, one
device-resident launch with the polish on (default), off (ACN round-3 behaviour: retry passes), and the polish's
counters.    python3 tools/gpu_polish.py [leg ...]   (legs of bench.other_workloads; default: the on-chip ones)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from adacharge_amd.backend import DeviceBatch, SiteHandle, default_options

legs = sys.argv[1:] or ["cfg3_site3_T12_b1024", "cfg3_site0_T12_b1024", "cfg2_caltech54_T24_b4096", "cfg2_jpl52_T24_b4096"]
dev = torch.device("cuda", 0)
def headline():
    from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
    from adacharge_amd.acn import Interface
    from adacharge_amd.builder import ProblemBatch, build_batch, make_site
    infra = sites.caltech54()
    iface = Interface({"infrastructure_info": infra, "period": 5})
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
    site = make_site(infra, "SOC")
    return ProblemBatch.concatenate([build_batch(sites.snapshot_batch(infra, 12, 256, seed=20240 + 104729 * g), infra, iface, obj, "SOC", site=site)
                                     for g in range(64)])


PS = [int(a[2:]) for a in sys.argv[1:] if a.startswith("P=")] or [800, 0, 1200]
legs = [a for a in legs if not a.startswith("P=")] or ["cfg3_site3_T12_b1024"]
for leg in legs:
    name, _, first = leg.partition("@")          # "<leg>@<n>": the first n problems of the leg
    batch = headline() if name == "headline" else bench.other_workloads()[name]()[0]
    if first:
        batch = batch.subset(slice(0, int(first)))
    out = {"leg": leg, "batch": batch.B}
    ref = None
    for name, o in [(f"P={P}", default_options(polish_iters=P)) for P in PS]:
        h = SiteHandle(batch.site, 0)
        db = DeviceBatch(batch, dev)
        ms = []
        for _ in range(3):
            h.solve_device(db, o, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            ms.append(h.last_kernel_ms())
        it, st, x = db.iters.cpu().numpy(), db.status.cpu().numpy(), db.x.cpu().numpy()
        stats = h.polish_stats()
        h.close()
        if ref is None:
            ref = x
        out[name] = {"ms": round(min(ms[1:]), 3), "iters_mean": round(float(it.mean()), 1), "iters_max": int(it.max()),
                     "solved": int((st == 1).sum()), "inaccurate": int((st == 5).sum()), "other": int((~np.isin(st, (1, 5))).sum()),
                     "polish": {k: (v // 3 if not isinstance(v, list) else [u // 3 for u in v]) for k, v in stats.items()}}
        out[name]["max_abs_diff_vs_first_A"] = float(np.abs(x - ref).max())
    print(json.dumps(out), flush=True)
