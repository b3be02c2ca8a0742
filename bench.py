#!/usr/bin/env python3
"""Headline benchmark: MPC QP solves/sec, 54 EVSE x horizon 12 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (acnqp_solve_batch_device: one kernel
launch) over one batch of 256 independent MPC state snapshots per GPU, inputs
already resident in HBM.  Workload = BASELINE.json configs[1]: Caltech-shaped
54-EVSE network, horizon 12, quick_charge (+ equal_share*1e-12, the reference's
own integration-test objective, t_int.py:67-70), the reference's default SOC
constraints (aco.py:35), fp64 ADMM, batch 256 per GPU (weak scaling: every rank
gets its own 256 snapshots; for N > 1 each step ends with one RCCL all-gather of
the schedules so every rank holds the whole job's result).  `--pipeline D`
(default 8) keeps D such batches in flight per GPU, each on its own stream: a
launch lasts as long as its slowest problem, so the next batches' workgroups
move into the CUs the finished problems have left.

Prints ONE JSON line on rank 0 (see the contract in the task statement), with
`roofline` (HBM algorithmic bytes / measured kernel time) and `cpu_baseline`
(oracle/admm_port.c, the scalar C port of the same ADMM, on all host cores over
a bounded sample; N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# Batches are kept in flight on separate HIP streams (--pipeline); the runtime multiplexes streams onto
# 4 hardware queues by default, and two streams that share a queue run back to back.  Must be set before
# the HIP runtime initialises (i.e. before torch is imported).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TF = 78.6     # vector fp64 (vendor figure)


def new_stream(dev):
    """A fresh HIP stream (hipStreamCreateWithFlags, non-blocking) wrapped for torch.  torch's own stream
    pool is created 64 streams at a time and shares the hardware queues among them in an order the caller
    cannot see; streams created here, one per batch in flight and before that pool exists, each get a
    hardware queue of their own as long as GPU_MAX_HW_QUEUES allows."""
    import ctypes

    import torch

    try:
        hip = ctypes.CDLL("libamdhip64.so")
        handle = ctypes.c_void_p()
        rc = hip.hipStreamCreateWithFlags(ctypes.byref(handle), ctypes.c_uint(1))   # hipStreamNonBlocking
        if rc != 0 or not handle.value:
            raise RuntimeError(f"hipStreamCreateWithFlags failed ({rc})")
        return torch.cuda.ExternalStream(handle.value, device=dev)
    except Exception as exc:   # still correct, possibly less overlap: a stream of torch's pool
        print(f"[bench] own HIP stream unavailable ({exc}); using a torch pool stream", file=sys.stderr)
        return torch.cuda.Stream(device=dev)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=256, help="problems per GPU per step")
    ap.add_argument("--horizon", type=int, default=12)
    ap.add_argument("--constraint-type", default="SOC", choices=["SOC", "LINEAR"])
    ap.add_argument("--precision", type=int, default=64, choices=[64, 32])
    ap.add_argument("--pipeline", type=int, default=8,
                    help="batches kept in flight per GPU, one HIP stream each (1 = strictly one launch at a time)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --one-device rehearses the N > 1 path on a single-GPU box")
    ap.add_argument("--one-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--rehearse-collective", action="store_true",
                    help="rehearsal only: run the per-step all-gather (and its process group) even with one rank")
    return ap.parse_args()


def algorithmic_bytes(B, N, Tm, K, site):
    """HBM bytes one launch must move: per problem read lb, ub, q
    and write x (4*N*Tm doubles), the session table (K*N * (4+4+8) B), 1 double + 1 int + 1
    byte of scalars in, 3 doubles + 2 ints out; the site matrices once per launch."""
    per_qp = 8 * 4 * N * Tm + 16 * K * N + (8 + 4 + 1) + (3 * 8 + 2 * 4)
    if site.has_peak:
        per_qp += 8 * Tm
    site_bytes = 8 * (2 * site.Mg * N + site.Mg * site.Mg + site.Mg + site.M)
    return B * per_qp + site_bytes, per_qp, site_bytes


def flops_per_iteration(N, Tm, site):
    """ALGORITHMIC flops of one ADMM iteration."""
    Mg = site.Mg
    return 4 * Mg * N * Tm + 4 * Mg * Mg * Tm + 20 * N * Tm + 12 * Mg * Tm


def cpu_baseline_leg(batch, gpu_x, gpu_status, target_seconds, snaps, infra, iface, accel_mem=0):
    """The ONLY place bench.py touches oracle/: the scalar C port of the device ADMM timed on
    the host cores over a bounded sample, which also serves as the parity check of the sample."""
    from oracle import admm_port
    import copy

    cores = min(admm_port.max_threads(), os.cpu_count() or 1)

    def sub(k):
        sb = copy.copy(batch)
        sb.B = k
        for name in ("T", "lb", "ub", "q", "pdiag", "lf", "s_off", "s_len", "s_cap", "s_eq", "const", "presolve_status"):
            setattr(sb, name, getattr(batch, name)[:k])
        sb.peak = None if batch.peak is None else batch.peak[:k]
        return sb

    admm_port.solve_batch(sub(min(cores, batch.B)), threads=cores, accel_mem=accel_mem)  # warm the thread pool / caches
    t0 = time.perf_counter()
    out = admm_port.solve_batch(batch, threads=cores, accel_mem=accel_mem)   # one pass over the whole rank-0 batch: parity sample
    one_pass = max(time.perf_counter() - t0, 1e-4)
    reps = int(max(1, min(200, round(target_seconds / one_pass))))
    t0 = time.perf_counter()
    for _ in range(reps):   # bounded sample of the same workload: the batch, `reps` times
        admm_port.solve_batch(batch, threads=cores, accel_mem=accel_mem)
    dt = time.perf_counter() - t0
    n = batch.B
    n_timed = n * reps
    ok = (out["status"] == 1) & (gpu_status[:n] == 1)
    dx = float(np.abs(out["x"][ok] - gpu_x[:n][ok]).max()) if ok.any() else float("nan")
    res = {
        "value": n_timed / dt, "unit": "QP solves/s", "cores": int(cores), "kind": "port",
        "sample": f"the {batch.B} rank-0 problems x {reps} passes = {n_timed} solves, oracle/admm_port.c (scalar C "
                  f"port of the device ADMM, gcc -O3 -fopenmp, one problem per thread), {dt:.1f} s wall",
    }
    parity = {"port_sample": n, "max_abs_rate_diff_gpu_vs_port_A": dx,
              "status_mismatches_vs_port": int((out["status"] != gpu_status[:n]).sum())}
    # independent solvers on the problem the reference states (pure quick_charge): objective / aggregate
    # gap against scipy-HiGHS (LINEAR rows) or the certified IPM oracle (SOC rows), first problems only
    try:
        ctype = "SOC" if batch.site.cone == 1 else "LINEAR"
        from oracle.ipm import solve_lp_highs
        from oracle.ref_problem import build_reference_problem

        gaps, aggs, th = [], [], []
        for b in range(min(4, batch.B)):
            prob = build_reference_problem(snaps[b], infra, iface, [("quick_charge", 1, {})], ctype)
            t0 = time.perf_counter()
            if ctype == "LINEAR":
                h = solve_lp_highs(prob)
                ok_ref, fun, xr = h.status == 0, h.fun, h.x.reshape(prob.N, prob.T)
            else:
                from oracle.ipm import solve_reference_problem
                xr, r_ = solve_reference_problem(prob)
                ok_ref, fun = r_.status in ("optimal", "optimal_inaccurate"), prob.objective(xr)
            th.append(time.perf_counter() - t0)
            if ok_ref and gpu_status[b] == 1:
                T = int(batch.T[b])
                x = gpu_x[b][:, :T]
                gaps.append((prob.objective(x) - fun) / abs(fun))
                aggs.append(float(np.abs(x.sum(0) - xr.sum(0)).max()))
        parity["independent_solver"] = "scipy-HiGHS" if ctype == "LINEAR" else "oracle/ipm.py (NT-scaled IPM)"
        parity["lp_rel_objective_gap_max"] = float(np.max(np.abs(gaps))) if gaps else None
        parity["lp_aggregate_gap_max_A"] = float(np.max(aggs)) if aggs else None
        res["independent_solver_ms_per_solve_1thread"] = 1e3 * float(np.median(th))
    except Exception as exc:  # the LINEAR-only cross-check is informative, never fatal
        parity["highs_error"] = repr(exc)
    return res, parity


def main():
    # The contract is ONE JSON line on stdout.  Libraries print there too (RCCL's version banner, Gloo's rank
    # chatter): keep the real stdout for the JSON line and point file descriptor 1 at stderr for everything else.
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists by design)")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    collective = world > 1 or args.rehearse_collective
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29591")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
    from adacharge_amd.acn import Interface
    from adacharge_amd.backend import DeviceBatch, SiteHandle, default_options
    from adacharge_amd.builder import build_batch

    infra = sites.caltech54()
    iface = Interface({"infrastructure_info": infra, "period": 5})
    T, B = args.horizon, args.batch
    objective = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
    opts = default_options(precision=args.precision)
    if args.precision == 32:
        opts.eps_abs = opts.eps_rel = 5e-5
    gdev = dev if args.dist_backend == "nccl" else torch.device("cpu")
    # `--pipeline` independent batches per GPU, each with its own inputs, result tensors, HIP stream and
    # gather buffer: a step launches the next batch on the next stream, so that a batch's stragglers
    # (the launch lasts as long as its slowest problem) overlap with the following batch's solves.
    depth = max(1, args.pipeline)
    slots = []
    for s_ in range(depth):
        sn = sites.snapshot_batch(infra, T, B, seed=20240 + rank + 1000 * s_)
        bt = build_batch(sn, infra, iface, objective, args.constraint_type)
        slots.append(dict(
            snaps=sn, batch=bt, handle=SiteHandle(bt.site, local_rank), dbatch=DeviceBatch(bt, dev),
            stream=new_stream(dev) if depth > 1 else torch.cuda.current_stream(),
            gathered=torch.empty((world * B, bt.N, bt.Tm), dtype=torch.float64, device=gdev) if collective else None,
        ))
    snaps, batch, handle, dbatch = (slots[0][k] for k in ("snaps", "batch", "handle", "dbatch"))

    def step(i):
        sl = slots[i % depth]
        with torch.cuda.stream(sl["stream"]):
            sl["handle"].solve_device(sl["dbatch"], opts, stream=sl["stream"].cuda_stream)
            if collective:   # the one collective of the job: every rank ends up with all schedules
                if args.dist_backend == "nccl":
                    dist.all_gather_into_tensor(sl["gathered"], sl["dbatch"].x)
                else:
                    dist.all_gather(list(sl["gathered"].chunk(world)), sl["dbatch"].x.cpu())

    def fence():
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(depth):     # setup, not warm-up: every slot's first launch (module load, LDS attribute)
        step(i)
    fence()
    for i in range(args.warmup):
        step(i)
    fence()
    for sl in slots:
        sl["handle"].kernel_times()   # discard the warm-up launches
    kernel_ms = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
        if (i + 1) % (32 * depth) == 0:   # the event ring holds 64 launches per handle
            for sl in slots:
                kernel_ms += sl["handle"].kernel_times()
    fence()
    elapsed = time.perf_counter() - t0
    for sl in slots:
        kernel_ms += sl["handle"].kernel_times()   # HIP events on the launch streams, read after the fence
    gathered = slots[0]["gathered"]
    if collective:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=gdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    x = dbatch.x.cpu().numpy()
    status = dbatch.status.cpu().numpy()
    iters = dbatch.iters.cpu().numpy()
    used = slots
    solved = int(sum(int((sl["dbatch"].status == 1).sum().item()) for sl in used))
    iters_all = np.concatenate([sl["dbatch"].iters.cpu().numpy() for sl in used])
    if collective:
        cnt = torch.tensor([solved, B * len(used)], dtype=torch.int64, device=gdev)
        dist.all_reduce(cnt)
        solved_all, total_all = int(cnt[0]), int(cnt[1])
        if rank == 0:   # the gather really carries every rank's schedules
            assert torch.equal(gathered[:B].to(dev), dbatch.x)
    else:
        solved_all, total_all = solved, B * len(used)

    if rank == 0:
        k_avg_ms = float(np.mean(kernel_ms))
        abytes, per_qp, site_bytes = algorithmic_bytes(B, batch.N, batch.Tm, batch.K, batch.site)
        achieved = abytes / (k_avg_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tfile):   # committed result of the separate rocprofv3 --pmc passes
            try:
                traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        fl = flops_per_iteration(batch.N, batch.Tm, batch.site)
        # flops of the iterations actually run (mean over the batches in flight) per second of wall time
        valu_tf = float(iters_all.mean()) * B * args.steps * fl / elapsed / 1e12
        out = {
            "metric": "MPC QP solves/sec whole-node, 54 EVSE x horizon 12; max rate residual vs cvxpy",
            "value": world * B * args.steps / elapsed,
            "unit": "QP solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if args.precision == 64 else "f32",
            "data": "synthetic",
            "config": {
                "workload": f"caltech54 (Caltech-shaped, 54 EVSE) x horizon {T}, quick_charge + equal_share*1e-12, "
                            f"{args.constraint_type} constraints, batch {B} independent MPC snapshots per GPU "
                            f"(BASELINE.json configs[1])",
                "batch_per_gpu": B, "n_evse": batch.N, "horizon": T, "constraint_type": args.constraint_type,
                "batches_in_flight_per_gpu": depth,
                "parallelism": f"dp{world} (one batch shard per GPU" + (", RCCL all-gather of schedules per step)" if world > 1 else ")"),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": "acnqp::admm_tiled_kernel<%s, 4, 1, 1, 1, 2, 5>" % ("double" if args.precision == 64 else "float"),
                "kernel_avg_ms": k_avg_ms, "launches_timed": len(kernel_ms), "launches_in_flight": depth,
                "algorithmic_bytes_per_launch": abytes, "bytes_per_qp": per_qp,
                "note": "LDS-resident iterative solver: HBM is touched once per problem, so the HBM fraction is "
                        "small by construction (SURVEY.md H8); the VALU view is in `valu`",
            },
            "valu": {
                "achieved": valu_tf, "peak": FP64_VALU_PEAK_TF if args.precision == 64 else 157.3, "unit": "TFLOP/s",
                "frac": valu_tf / (FP64_VALU_PEAK_TF if args.precision == 64 else 157.3),
                "flops_per_iteration": fl, "iterations_mean": float(iters_all.mean()), "iterations_max": int(iters_all.max()),
            },
            "solver": {
                "solved": solved_all, "problems": total_all,
                "eps_abs": opts.eps_abs, "eps_rel": opts.eps_rel, "reg_rel": opts.reg_rel,
                "anderson_columns": handle.accel_columns(batch.Tm, batch.K, opts),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            cb, parity = cpu_baseline_leg(batch, x, status, args.cpu_seconds, snaps, infra, iface,
                                          accel_mem=handle.accel_columns(batch.Tm, batch.K, opts))
            out["cpu_baseline"] = cb
            out["parity"] = parity
        print(json.dumps(out), file=json_out, flush=True)
    if collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
