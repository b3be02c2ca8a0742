"""One device-resident launch of a BASELINE.json configuration other than the bench's (profiling target):

    python3 tools/run_config.py cfg2-caltech|cfg2-jpl|cfg3-site0|cfg3-site3|cfg5|stress144|stress144-2k|cfg4 [scenarios] [--ranks R]

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

def cfg4_site_batches(S):
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-3)]
    out = []
    for k, infra in enumerate(sites.eight_sites()):
        iface = Interface({"infrastructure_info": infra, "period": 5})
        rng = np.random.default_rng(500 + k)
        base = build_batch([sites.random_sessions(infra, 12, rng)], infra, iface, obj, "SOC")
        out.append(scenario_batch(base, rng.lognormal(0.0, 0.25, size=(S, base.K, base.N))))
    return out


def cfg4_rank(rank, world, port, S, backend, q):
    """One rank of configs[3] as a sharded job (adacharge_amd.distributed.solve_sites_sharded): site-major order, one
    SiteHandle per site the rank owns, ONE all-gather of the padded schedules."""
    import time
    import torch.distributed as dist
    from adacharge_amd.distributed import solve_sites_sharded

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group(backend, rank=rank, world_size=world)
    batches = cfg4_site_batches(S)
    dev = 0 if torch.cuda.device_count() < world else rank   # one GPU box: every rank on cuda:0 (rehearsal)
    torch.cuda.set_device(dev)
    solve_sites_sharded(batches, local_device=dev, device=None if backend == "nccl" else "cpu")   # warm-up: handles, workspaces
    dist.barrier()
    t0 = time.perf_counter()
    x, st = solve_sites_sharded(batches, local_device=dev, device=None if backend == "nccl" else "cpu")
    dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        q.put(dict(config="cfg4-sharded", ranks=world, backend=backend, scenarios_per_site=S, problems=int(len(st)), wall_ms=1e3 * dt,
                   qps=len(st) / dt, solved=int((st == 1).sum()), shape=list(x.shape),
                   note="wall time of one sharded solve incl. upload, solve, all-gather and download on every rank; "
                        + ("all ranks share cuda:0 (rehearsal on a one-GPU box)" if torch.cuda.device_count() < world else "one GPU per rank")))
    dist.destroy_process_group()



def main():
    which = sys.argv[1]
    reps = 3
    if which == "cfg4" and "--ranks" in sys.argv:
        import torch.multiprocessing as mp

        R = int(sys.argv[sys.argv.index("--ranks") + 1])
        S = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 1024
        backend = "nccl" if torch.cuda.device_count() >= R else "gloo"
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=cfg4_rank, args=(r, R, 29700 + os.getpid() % 200, S, backend, q)) for r in range(R)]
        for p in procs:
            p.start()
        print(json.dumps(q.get(timeout=900)))
        for p in procs:
            p.join(timeout=120)
        raise SystemExit(max(p.exitcode or 0 for p in procs))
    if which == "cfg4":
        # configs[3]: 1024 demand scenarios x 8 sites, horizon 12; on one GPU the 8 site shards run one after the other
        # (`--ranks R`: as a sharded job, one process per rank).  One JSON line: per-site kernel time, statuses, and the total.
        S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
        st = torch.cuda.current_stream().cuda_stream
        rows, tot_ms, tot_n = [], 0.0, 0
        for k, batch in enumerate(cfg4_site_batches(S)):
            h = SiteHandle(batch.site, 0)
            dev = DeviceBatch(batch, "cuda:0")
            ms = []
            for _ in range(reps):
                h.solve_device(dev, default_options(), stream=st)
                torch.cuda.synchronize()
                ms.append(h.last_kernel_ms())
            it = dev.iters.cpu().numpy(); s_ = dev.status.cpu().numpy()
            rows.append(dict(site=k, n_evse=batch.N, kernel_ms=min(ms), iters_mean=float(it.mean()),
                             iters_max=int(it.max()), solved=int((s_ == 1).sum()), inaccurate=int((s_ == 5).sum()), max_iter=int((s_ == 2).sum())))
            tot_ms += min(ms); tot_n += S
            h.close()
        print(json.dumps(dict(config="cfg4", scenarios_per_site=S, problems=tot_n, kernel_ms_total=tot_ms, qps=tot_n / tot_ms * 1e3, sites=rows)))
        raise SystemExit(0)
    ALIASES = {"cfg2-caltech": "cfg2_caltech54_T24_b4096", "cfg2-jpl": "cfg2_jpl52_T24_b4096", "cfg5": "cfg4_synth512_T48_b2048",
               "stress144": "stress_caltech54_T144_b256", "stress144-2k": "stress_caltech54_T144_b2048", "cfg3-site3": "cfg3_site3_T12_b1024", "cfg3-site0": "cfg3_site0_T12_b1024"}
    import bench   # the workloads are bench.py's `other_configs` legs (one definition for the driver's line and the profiles)
    name = ALIASES.get(which, which)
    if name not in bench.other_workloads():
        raise SystemExit(__doc__)
    print(json.dumps({name: bench.other_configs_leg(torch.device("cuda", 0), only=[name])[name]}))


if __name__ == "__main__":
    main()
