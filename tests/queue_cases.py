"""One launch of >= 768 problems per kernel family, solved through acnqp_solve_batch_device (ONE launch, so the queue
order of acn_qp_api.hip engages).  Run as a script it solves one case under the environment it was started with
(ACNQP_NO_ORDER=1: natural queue order; ACNQP_NO_QUEUE=1: the static one-workgroup-per-problem schedule) and saves the
result: tests/test_gpu_parity.py::test_work_queue_and_launch_order_do_not_change_results compares the bits."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# name -> (site, horizon, batch, equal_share weight, snapshot_batch keywords)
CASES = {
    "wave":     ("caltech54", 12, 1024, 1e-12, {}),    # the headline shape: one wave per problem (acn_qp_wave.hpp)
    "tiled":    ("caltech54", 16, 1024, 1e-12, {}),    # horizon 13 ... 16 stays with the register-resident tiled kernel
    "wave4":    ("jpl52", 24, 800, 1e-3, {}),          # two row tiles x horizon 24: four waves per problem (acn_qp_wave.hpp)
    "long-lds": ("jpl52", 28, 800, 1e-3, {}),          # horizon 25 ... 32 stays with the LDS-resident long-horizon kernel
    "long-96":  ("caltech54", 96, 800, 1e-12, dict(demand_range=(5.0, 60.0))),
    "stream":   ("wide128", 12, 800, 1e-3, dict(min_sessions=40)),
    "general":  ("caltech54", 300, 768, 1e-12, dict(demand_range=(5.0, 60.0))),
}


def build(name):
    from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
    from adacharge_amd.acn import Interface
    from adacharge_amd.builder import build_batch

    site, T, B, es, kw = CASES[name]
    infra = getattr(sites, site)()
    iface = Interface({"infrastructure_info": infra, "period": 5})
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, es)]
    return build_batch(sites.snapshot_batch(infra, T, B, seed=4242 + T, **kw), infra, iface, obj, "SOC")


def solve(name):
    """(x, iters, status, launches whose queue order was sorted)"""
    import torch

    from adacharge_amd.backend import DeviceBatch, SiteHandle, default_options

    batch = build(name)
    h = SiteHandle(batch.site, 0)
    db = DeviceBatch(batch, torch.device("cuda", 0))
    h.solve_device(db, default_options(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out = dict(x=db.x.cpu().numpy(), iters=db.iters.cpu().numpy(), status=db.status.cpu().numpy(),
               ordered=np.array(h.ordered_launches()), keys=(batch.s_len > 0).sum(axis=(1, 2)))
    h.close()
    return out


if __name__ == "__main__":
    np.savez(sys.argv[2], **solve(sys.argv[1]))
