"""Device polish against its numpy specification, problem by problem, on the bench workload (LP-like, Tikhonov floor):
Newton rounds and distance, from the device's own hand-over point.   python3 tools/gpu_polish_vs_spec.py [batches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.acn import Interface
from adacharge_amd.backend import SiteHandle, default_options
from adacharge_amd.builder import ProblemBatch, build_batch, make_site
import oracle.polish_ref as PR

G = int(sys.argv[1]) if len(sys.argv) > 1 else 24
infra = sites.caltech54()
iface = Interface({"infrastructure_info": infra, "period": 5})
obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
site = make_site(infra, "SOC")
batch = ProblemBatch.concatenate([build_batch(sites.snapshot_batch(infra, 12, 256, seed=20240 + 104729 * g), infra, iface, obj, "SOC", site=site) for g in range(G)])
h = SiteHandle(batch.site, 0)
handed = h.solve(batch, default_options(max_iter=800, retry_passes=0, polish_iters=0), want_y=True)
res = h.solve(batch, default_options(retry_passes=0))
print(h.polish_stats())
h.close()
for b in np.flatnonzero(handed.status != 1):
    xs, info = PR.polish_batch_problem(batch, b, handed.x[b], handed.y[b])
    T = xs.shape[1]
    print(f"problem {b}: device status {res.status[b]} rounds {res.iters[b] - 800}; spec ok {info['ok']} rounds {info['rounds']} rows {info['rows']}; "
          f"|device - spec| {np.abs(xs - res.x[b][:, :T]).max():.2e}", flush=True)
