#!/usr/bin/env python3
"""Headline benchmark: MPC QP solves/sec, 54 EVSE x horizon 12 (BASELINE.json), END TO END.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload = BASELINE.json configs[1]: Caltech-shaped 54-EVSE network, horizon 12, quick_charge
(+ equal_share*1e-12, the reference's own integration-test objective, t_int.py:67-70), the reference's
default SOC constraints (aco.py:35), fp64 ADMM, batches of 256 independent MPC state snapshots.

One "step" = ONE call of the product entry point `acnqp_solve_batches` over `--batches` (default 64) such
batches per GPU, i.e. 16,384 problems per GPU per step, from pinned HOST buffers to pinned HOST buffers:
H2D of every problem array + kernels + D2H of schedules, statuses, iterations, residuals and objectives,
all inside the timed region (the metric SURVEY.md section 8d defines; `acnqp_create` -- the one-time site
upload -- is outside).  The library pipelines the call internally (chunks of 2,048 / 4,096 / 8,192 / 2,048 problems
on this shape -- small ones first and last, which shortens the exposed head and tail of the pipeline -- rotate over four streams with
their own device staging; the small per-problem arrays travel through pinned mirrors, one copy per chunk and
direction), so there is nothing for the bench to overlap by hand and
`value` is what any caller of the API gets.  Weak scaling: every rank owns its own 64 x 256 snapshots.
For N > 1 each step leaves the rank's schedules in HBM as well (acnqp_results.x_dev) and ends with the
job's single collective, one RCCL all-gather of the schedules over xGMI, overlapped with the next step's
solve; the timed region ends when every gather has landed.

Prints ONE JSON line on rank 0 with `roofline` (algorithmic HBM bytes of one launch / its HIP-event
duration), `cpu_baseline` (oracle/admm_port.c on the host cores, bounded sample, N = 1 only) and the
kernel-only rate of the same step (`kernel_only`).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TF = 78.6     # vector fp64 (vendor figure)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)   # (32 x ~19 ms: the timed region stays above half a second for the driver's clock)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="problems per batch (BASELINE.json configs[1]: 256)")
    ap.add_argument("--batches", type=int, default=64, help="batches per GPU per step (one acnqp_solve_batches call)")
    ap.add_argument("--horizon", type=int, default=12)
    ap.add_argument("--constraint-type", default="SOC", choices=["SOC", "LINEAR"])
    ap.add_argument("--precision", type=int, default=64, choices=[64])
    ap.add_argument("--pageable", action="store_true", help="problem / result arrays in pageable host memory")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the legs that time the other BASELINE.json configurations after the headline (N = 1 only)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --one-device rehearses the N > 1 path on a single-GPU box")
    ap.add_argument("--one-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--rehearse-collective", action="store_true",
                    help="rehearsal only: run the per-step all-gather (and its process group) even with one rank")
    return ap.parse_args(argv)


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start one fresh process per GPU (this parent never touches
    HIP, so no process that has initialised the GPU is replaced or forked) and return the worst exit code."""
    port = os.environ.get("MASTER_PORT", str(29500 + os.getpid() % 2000))
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    return max(p.wait() for p in procs)


def algorithmic_bytes(B, N, Tm, K, site):
    """HBM bytes one launch over B problems must move: per problem read lb, ub, q and write x
    (4*N*Tm doubles), the session table (K*N * (4+4+8) B), 1 double + 1 int + 1 byte of scalars in, 3 doubles +
    2 ints out; the site matrices once per launch."""
    per_qp = 8 * 4 * N * Tm + 16 * K * N + (8 + 4 + 1) + (3 * 8 + 2 * 4)
    if site.has_peak:
        per_qp += 8 * Tm
    site_bytes = 8 * (2 * site.Mg * N + site.Mg * site.Mg + site.Mg + site.M)
    return B * per_qp + site_bytes, per_qp, site_bytes


def flops_per_iteration(N, Tm, site):
    """Flops one ADMM iteration EXECUTES with G and Q dense (what the MFMA chains do), per problem."""
    Mg = site.Mg
    return 4 * Mg * N * Tm + 4 * Mg * Mg * Tm + 20 * N * Tm + 12 * Mg * Tm


def flops_per_iteration_sparse(batch):
    """ALGORITHMIC flops of one ADMM iteration: F_iter = 4 nnz(A) + 12 (n + m) + F_solve with
    nnz(A) = n + sum_s len_s + T nnz(G) (+ T N for a peak row), m = n + S + Mg T rows, and the matrix-free solve
    F_solve = 4 Mg^2 T (the two Mg x Mg products per period).  Mean over the batch's problems."""
    import numpy as np

    site = batch.site
    N, Tm = batch.N, batch.Tm
    T = np.asarray(batch.T, float)
    n = N * T
    nnzG = float(np.count_nonzero(site.G))
    sess = (batch.s_len > 0).reshape(batch.B, -1).sum(axis=1).astype(float)
    slen = batch.s_len.reshape(batch.B, -1).sum(axis=1).astype(float)
    nnzA = n + slen + T * nnzG
    m = n + sess + site.Mg * T
    return float(np.mean(4 * nnzA + 12 * (n + m) + 4 * site.Mg * site.Mg * T))


def streamed_bytes_per_iteration(batch):
    """SURVEY.md section 8d B_iter = w (3 n + 6 m) for kernels whose state is not on-chip (read x, q, z, y, l, u;
    write x, z, y), m = n + S + Mg T rows; mean over the batch."""
    import numpy as np

    T = np.asarray(batch.T, float)
    n = batch.N * T
    sess = (batch.s_len > 0).reshape(batch.B, -1).sum(axis=1).astype(float)
    m = n + sess + batch.site.Mg * T
    return float(np.mean(8 * (3 * n + 6 * m)))


# ---- the other BASELINE.json configurations (timed outside the headline region, reported in the same JSON line) -----
def other_workloads():
    """name -> builder returning (batch, options, streamed, note).  `streamed`: the kernel that serves the shape keeps its
    iterates in HBM / L2 (large-site and long-horizon kernels), so the roofline counts B_iter per iteration; otherwise
    the state is on chip and only the compulsory I/O counts."""
    import numpy as np

    from adacharge_amd import ObjectiveComponent, equal_share, load_flattening, quick_charge, sites
    from adacharge_amd.acn import Interface
    from adacharge_amd.backend import default_options
    from adacharge_amd.builder import ProblemBatch, build_batch, scenario_batch

    qc_es = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-3)]

    def cfg2(site_name):
        infra = getattr(sites, site_name)()
        iface = Interface({"infrastructure_info": infra, "period": 5})
        batch = build_batch(sites.snapshot_batch(infra, 24, 4096, seed=31), infra, iface, qc_es, "SOC")
        return batch, default_options(), False, "configs[2]: horizon 24, batch 4096, fp64 (fp32 is refused: DESIGN.md 3.5)"

    def cfg3_site(k):
        infra = sites.eight_sites()[k]
        iface = Interface({"infrastructure_info": infra, "period": 5})
        rng = np.random.default_rng(500 + k)
        base = build_batch([sites.random_sessions(infra, 12, rng)], infra, iface, qc_es, "SOC")
        batch = scenario_batch(base, rng.lognormal(0.0, 0.25, size=(1024, base.K, base.N)))
        return batch, default_options(), False, f"configs[3]: site {k} of the 8 ({infra.num_stations} EVSE), 1024 demand scenarios, horizon 12"

    def cfg4():
        # 512 EVSE x 48, load flattening of an external load profile with the sessions' energy DELIVERED (equalities,
        # t_int.py:350-403 style): the site rows bind (utilisation 0.96-1.0, every scenario feasible: C twin) and the
        # default tolerances need hundreds of iterations (round 2 timed a workload whose optimum was the all-zero
        # schedule: 20 iterations, nothing binding)
        infra = sites.synth512()
        iface = Interface({"infrastructure_info": infra, "period": 5})
        T = 48
        ext = 150.0 + 100.0 * np.cos(np.arange(T) / T * 2 * np.pi)
        obj = [ObjectiveComponent(load_flattening, 1.0, {"external_signal": ext})]
        rng = np.random.default_rng(5)
        snaps = [sites.random_sessions_general(infra, T, rng, False, False, demand_scale=0.12) for _ in range(8)]
        base = build_batch(snaps, infra, iface, obj, "SOC", True)
        batch = ProblemBatch.concatenate([scenario_batch(base, rng.lognormal(0.0, 0.05, size=256), problem=p) for p in range(8)])
        return batch, default_options(), True, "configs[4] shape: synthetic 512 EVSE x 48, load_flattening + energy equalities, 8 snapshots x 256 demand scenarios, default tolerances"

    def stress144(B=256):
        infra = sites.caltech54()
        iface = Interface({"infrastructure_info": infra, "period": 5})
        obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
        batch = build_batch(sites.snapshot_batch(infra, 144, B, seed=144, demand_range=(5.0, 60.0)), infra, iface, obj, "SOC")
        return batch, default_options(), True, f"the reference's stress shape (t_aco.py:286-466): 54 EVSE x 144 periods, batch {B}"

    return {
        "cfg2_caltech54_T24_b4096": lambda: cfg2("caltech54"),
        "cfg2_jpl52_T24_b4096": lambda: cfg2("jpl52"),
        "cfg3_site3_T12_b1024": lambda: cfg3_site(3),
        "cfg3_site0_T12_b1024": lambda: cfg3_site(0),
        "cfg4_synth512_T48_b2048": cfg4,
        "stress_caltech54_T144_b256": stress144,
        # eight workgroups' worth of problems per CU: the long-horizon kernel's bandwidth-bound regime (in the default
        # run since round 4, so that its HBM fraction is driver-timed)
        "stress_caltech54_T144_b2048": lambda: stress144(2048),
    }


# legs of `other_workloads` the default `python bench.py` run leaves out (profiling targets only)
NOT_IN_DEFAULT_RUN = ()


def time_device_launch(batch, opts, dev, reps=2):
    """One device-resident launch of `batch` (acnqp_solve_batch_device), HIP events around it; returns the leg's record."""
    import numpy as np
    import torch

    from adacharge_amd.backend import DeviceBatch, SiteHandle

    h = SiteHandle(batch.site, dev.index or 0)
    db = DeviceBatch(batch, dev)
    st = torch.cuda.current_stream().cuda_stream
    ms = []
    for _ in range(reps + 1):   # the first launch also allocates the kernel's workspace
        h.solve_device(db, opts, stream=st)
        torch.cuda.synchronize()
        ms.append(h.last_kernel_ms())
    ms = ms[1:]
    it = db.iters.cpu().numpy()
    stt = db.status.cpu().numpy()
    cols = h.accel_columns(batch.Tm, batch.K, opts)
    h.close()
    del db
    return min(ms), it, stt, cols


def other_configs_leg(dev, only=None):
    import numpy as np

    out = {}
    for name, build in other_workloads().items():
        if (only and name not in only) or (not only and name in NOT_IN_DEFAULT_RUN):
            continue
        t0 = time.perf_counter()
        batch, opts, streamed, note = build()
        t_build = time.perf_counter() - t0
        ms, it, stt, cols = time_device_launch(batch, opts, dev)
        N, Tm, K = batch.N, batch.Tm, batch.K
        _, per_qp, _ = algorithmic_bytes(1, N, Tm, K, batch.site)
        b_iter = streamed_bytes_per_iteration(batch) if streamed else 0.0
        abytes = batch.B * (per_qp + float(it.mean()) * b_iter)
        fl = flops_per_iteration_sparse(batch)
        tf = float(it.sum()) * fl / (ms * 1e-3) / 1e12
        out[name] = {
            "note": note, "batch": batch.B, "n_evse": N, "horizon": Tm, "site_rows": batch.site.Mg,
            "kernel_ms": ms, "qps": batch.B / (ms * 1e-3),
            "iters_mean": float(it.mean()), "iters_max": int(it.max()),
            "solved": int((stt == 1).sum()), "inaccurate": int((stt == 5).sum()), "failed": int(np.isin(stt, (2, 3, 4)).sum()),
            "anderson_columns": cols, "state": "streamed (HBM / L2)" if streamed else "on chip",
            "algorithmic_bytes_per_launch": abytes, "hbm_GBs": abytes / (ms * 1e-3) / 1e9,
            "hbm_frac": abytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "flops_per_iteration_sparse": fl, "fp64_TFs": tf, "fp64_frac": tf / FP64_VALU_PEAK_TF,
            "build_s": t_build,
        }
        if name.startswith("cfg3_"):
            # configs[3] is the scenario MPC: 1,024 problems per GPU, each with a wavefront of its own -- the launch lasts
            # as long as its slowest problem.  options.polish_stall (ABI v9, off by default) hands a stalled problem to the
            # polish early; reported beside the default, never instead of it
            from adacharge_amd.backend import default_options

            o2 = default_options(polish_stall=100)
            for f, _ in opts._fields_:
                if f != "polish_stall":
                    setattr(o2, f, getattr(opts, f))
            ms2, it2, st2, _ = time_device_launch(batch, o2, dev)
            out[name]["polish_stall_100"] = {"kernel_ms": ms2, "qps": batch.B / (ms2 * 1e-3), "iters_mean": float(it2.mean()),
                                             "iters_max": int(it2.max()), "solved": int((st2 == 1).sum())}
    return out


def strict_batch256_leg(handle, batch, opts, calls=40):
    """BASELINE.json configs[1] to the letter: ONE batch of 256 snapshots per acnqp_solve_batch call, host buffers in,
    host buffers out, nothing else in flight -- the latency-bound way to use the library."""
    import numpy as np

    # (the binding's defaults: plain numpy result arrays.  Pinned result arrays -- `pinned_results=True`, what the
    # pipelined entry wants -- cost a hipHostMalloc per call here: 1.5 ms of the 4.9 ms this leg reported before,
    # tools/gpu_strict256.py; the library call itself is kernel + 0.2 ms)
    handle.solve(batch, opts)
    t, k = [], []
    for _ in range(calls):
        t0 = time.perf_counter()
        r = handle.solve(batch, opts)
        t.append(time.perf_counter() - t0)
        k.append(r.kernel_ms)
    med = float(np.median(t))
    return {"note": "configs[1] strict: one acnqp_solve_batch call per 256-snapshot batch, end to end (H2D + kernel + D2H), "
                    "nothing overlapped; through the Python binding with its defaults (numpy arrays in, numpy arrays out)",
            "batch": batch.B, "calls": calls, "ms_per_call_median": 1e3 * med, "ms_per_call_min": 1e3 * float(np.min(t)),
            "qps": batch.B / med, "kernel_ms_median": float(np.median(k))}


def host_inclusive_leg(infra, iface, objective, site, handle, opts, n_snapshots=16384, horizon=12, reps=3):
    """From the caller's data to schedules, wall clock, ONE host thread: the statement -> arrays step
    is part of `north_star`'s path ("a direct (P, q, A, l, u) builder in Python").
      table:    a SessionTable of `n_snapshots` MPC snapshots (arrays from the start) -> builder.plan_from_table (numpy:
                slots, energy caps, one linear cost per horizon) -> acnqp_solve_table (H2D of the sessions, the dense
                problem arrays formed on the device, kernels, D2H of the schedules);
      sessions: lists of SessionInfo objects -> AdaptiveSchedulingAlgorithm.schedule_batch -> one dict per snapshot
                (reading the Python objects, pre- and post-processing included), on 2,048 snapshots."""
    import numpy as np

    from adacharge_amd import AdaptiveSchedulingAlgorithm, sites
    from adacharge_amd.builder import plan_from_table

    table = sites.snapshot_table(infra, horizon, n_snapshots, seed=424242)
    t_plan, t_solve, solved = [], [], 0
    res = None
    for _ in range(reps + 1):   # (the first pass allocates the pinned result arrays a periodic caller keeps: not timed)
        t0 = time.perf_counter()
        plan = plan_from_table(table, infra, iface, objective, "SOC", site=site)
        t1 = time.perf_counter()
        first = res is None
        res = handle.solve_table(plan, opts, pinned_results=True, out=res)
        t2 = time.perf_counter()
        if first:
            continue
        t_plan.append(t1 - t0); t_solve.append(t2 - t1)
        solved = int((res.status == 1).sum())
    k = int(np.argmin(np.add(t_plan, t_solve)))
    h2d = plan.S * (4 * 4 + 8) + (plan.S + 1) * 4 + plan.rate_seg[-1] * 16 + plan.B * (4 + 4 + 8 + 1 + 4) + plan.q_table.nbytes
    out = {"table": {
        "snapshots": n_snapshots, "sessions": int(plan.S), "qps": n_snapshots / (t_plan[k] + t_solve[k]),
        "plan_ms": 1e3 * t_plan[k], "solve_table_ms": 1e3 * t_solve[k], "solved": solved,
        "h2d_bytes_per_problem": float(h2d) / n_snapshots,
        "note": "SessionTable -> plan_from_table (numpy, one thread) -> acnqp_solve_table (sessions to the device, lb / ub / q formed "
                "there) -> schedules in (pinned, reused) host memory; best of %d" % reps}}
    n2 = 2048
    snaps = sites.snapshot_batch(infra, horizon, n2, seed=515151)
    alg = AdaptiveSchedulingAlgorithm(objective, solver_options={})
    alg.register_interface(iface)
    alg.schedule_batch(snaps[:64])   # site handle, module load
    t0 = time.perf_counter()
    outs = alg.schedule_batch(snaps)
    dt = time.perf_counter() - t0
    out["sessions"] = {"snapshots": n2, "qps": n2 / dt, "ms": 1e3 * dt, "solved": int(sum(o is not None for o in outs)),
                       "note": "lists of SessionInfo objects -> schedule_batch (pre-processing, table entry, post-processing) -> one "
                               "{station: rates} dict per snapshot; one host thread"}
    # configs[0]'s call pattern: ONE MPC step, AdaptiveSchedulingAlgorithm.schedule(active_sessions) -> {station: rates}
    # (ada.py:141-194; the reference solves it with cvxpy / ECOS on the CPU) -- the latency a simulator loop sees
    lat = []
    for k in range(200):
        t0 = time.perf_counter()
        o = alg.schedule(snaps[k])
        lat.append(time.perf_counter() - t0)
        assert o is not None
    lat = np.sort(np.asarray(lat[8:]))
    out["single_step"] = {"calls": int(lat.size), "ms_median": 1e3 * float(np.median(lat)), "ms_p95": 1e3 * float(lat[int(0.95 * lat.size)]),
                          "ms_min": 1e3 * float(lat[0]),
                          "note": "configs[0]: one schedule() call per MPC step on one snapshot (pre-processing, builder, H2D, "
                                  "kernel, D2H, post-processing; one host thread, nothing batched or overlapped)"}
    return out


def cpu_baseline_leg(batch, gpu_x, gpu_status, target_seconds, snaps, infra, iface, accel_mem=0, first=256):
    """The ONLY place bench.py touches oracle/: the scalar C port of the device ADMM timed on the host cores over a
    bounded sample, which also serves as the parity check of the sample.

    `batch`: >= 32 problems per thread (256 problems on 128 threads were two problems per thread -- a pass
    lasted as long as its slowest problem).  Reported: the all-core rate (`value`), the one-thread rate beside it, and
    scipy-HiGHS's time per solve on the LINEAR statement of the same instances (the only third-party solver in the
    image; cvxpy / ECOS are not installed)."""
    import numpy as np

    from oracle import admm_port

    cores = min(admm_port.max_threads(), os.cpu_count() or 1)
    t0 = time.perf_counter()
    out = admm_port.solve_batch(batch, threads=cores, accel_mem=accel_mem)   # one pass: warms the pool, the parity sample
    one_pass = max(time.perf_counter() - t0, 1e-4)
    reps = int(max(1, min(50, round(target_seconds / one_pass))))
    t0 = time.perf_counter()
    for _ in range(reps):   # bounded sample of the same workload
        admm_port.solve_batch(batch, threads=cores, accel_mem=accel_mem)
    dt = time.perf_counter() - t0
    n = batch.B
    # one thread: a slice of the same problems, sized to a few seconds
    n1 = int(max(8, min(n, round(3.0 / max(one_pass * cores / n, 1e-5)))))
    one = batch.subset(slice(0, n1))
    t0 = time.perf_counter()
    admm_port.solve_batch(one, threads=1, accel_mem=accel_mem)
    dt1 = max(time.perf_counter() - t0, 1e-6)
    ok = (out["status"] == 1) & (gpu_status[:n] == 1)
    dx = float(np.abs(out["x"][ok] - gpu_x[:n][ok]).max()) if ok.any() else float("nan")
    res = {
        "value": n * reps / dt, "unit": "QP solves/s", "cores": int(cores), "kind": "port",
        "one_thread_qps": n1 / dt1, "one_thread_ms_per_solve": 1e3 * dt1 / n1,
        "problems_per_thread_per_pass": n / cores,
        "sample": f"{n} problems of the timed workload (the first {n // first} batches) x {reps} passes = {n * reps} solves on "
                  f"{cores} threads ({n / cores:.0f} problems per thread and pass, dynamic schedule), oracle/admm_port.c (scalar C "
                  f"port of the device ADMM, gcc -O3 -fopenmp), {dt:.1f} s wall; one thread: {n1} problems in {dt1:.1f} s",
    }
    parity = {"port_sample": n, "max_abs_rate_diff_gpu_vs_port_A": dx,
              "status_mismatches_vs_port": int((out["status"] != gpu_status[:n]).sum())}
    # independent solvers on the problem the reference states (pure quick_charge): objective / aggregate gap
    # against scipy-HiGHS (LINEAR rows) or the certified IPM oracle (SOC rows), first problems only
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
        # scipy-HiGHS on the LINEAR statement of the same snapshots (aco.py:165-172 instead of the SOC rows): what a
        # compiled third-party LP solver needs per solve on one thread, whatever cone the timed workload uses
        tl = []
        for b in range(min(16, len(snaps))):
            prob = build_reference_problem(snaps[b], infra, iface, [("quick_charge", 1, {})], "LINEAR")
            t0 = time.perf_counter()
            solve_lp_highs(prob)
            tl.append(time.perf_counter() - t0)
        res["highs_linear_ms_per_solve_1thread"] = 1e3 * float(np.median(tl))
        res["highs_linear_note"] = ("scipy.optimize.linprog(method='highs') on the LINEAR statement of the first "
                                    f"{len(tl)} snapshots, median, scipy wrapper included; not the reference's cvxpy / ECOS path")
    except Exception as exc:  # the cross-check is informative, never fatal
        parity["independent_solver_error"] = repr(exc)
    return res, parity


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))   # before torch / HIP are touched in this process
    world = int(env_world or "1")
    if world != args.gpus and not args.one_device:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch one rank per GPU "
                         f"(torch.distributed.run --nproc-per-node {args.gpus}) or run without a launcher")
    # The contract is ONE JSON line on stdout.  Libraries print there too (RCCL's version banner, Gloo's rank
    # chatter): keep the real stdout for the JSON line and point file descriptor 1 at stderr for everything else.
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
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
    from adacharge_amd.backend import SiteHandle, default_options, pinned_empty
    from adacharge_amd.builder import ProblemBatch, build_batch, make_site

    infra = sites.caltech54()
    iface = Interface({"infrastructure_info": infra, "period": 5})
    T, B, G = args.horizon, args.batch, args.batches
    objective = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
    opts = default_options(precision=args.precision)
    if args.precision == 32:
        opts.eps_abs = opts.eps_rel = 5e-5
    alloc = None if args.pageable else pinned_empty
    site = make_site(infra, args.constraint_type)
    handle = SiteHandle(site, local_rank)
    # G batches of B distinct snapshots per rank, every array the ABI reads in (pinned) host memory
    snaps0 = None
    batches = []
    for g in range(G):
        sn = sites.snapshot_batch(infra, T, B, seed=20240 + 7919 * rank + 104729 * g)
        if g == 0:
            snaps0 = sn
        batches.append(build_batch(sn, infra, iface, objective, args.constraint_type, site=site, alloc=alloc))
    N, Tm, K = batches[0].N, batches[0].Tm, batches[0].K
    assert all(b.Tm == Tm and b.K == K for b in batches)
    per_step = G * B

    gdev = dev if args.dist_backend == "nccl" else torch.device("cpu")
    x_dev = gathered = None
    x_dev_ptrs = None
    if collective:   # the rank's schedules also stay in HBM (acnqp_results.x_dev); two gather buffers: the all-gather
        x_dev = [torch.zeros((per_step, N, Tm), dtype=torch.float64, device=dev) for _ in range(2)]
        gathered = [torch.empty((world * per_step, N, Tm), dtype=torch.float64, device=gdev) for _ in range(2)]
    runs = []
    for k in range(2 if collective else 1):
        ptrs = None
        if collective:
            ptrs = [x_dev[k].data_ptr() + g * B * N * Tm * 8 for g in range(G)]
        runs.append(handle.prepare_many(batches, pinned_results=not args.pageable, x_dev_ptrs=ptrs))
    pending = [None, None]
    gap_s = float(os.environ.get("BENCH_GAP_MS", "0")) * 1e-3

    def step(i):
        k = i % 2 if collective else 0
        if collective and pending[k] is not None:   # the gather issued two steps ago used these buffers
            pending[k].wait()
            pending[k] = None
        run, _ = runs[k]
        run(opts)   # acnqp_solve_batches: H2D + kernels + D2H (+ x_dev), synchronous
        if gap_s > 0:
            time.sleep(gap_s)   # diagnostic (BENCH_GAP_MS): an idle gap between steps, so that a trace shows them apart
        if collective:   # the one collective of the job, overlapped with the next step's solve
            if args.dist_backend == "nccl":
                pending[k] = dist.all_gather_into_tensor(gathered[k], x_dev[k], async_op=True)
            else:
                pending[k] = dist.all_gather(list(gathered[k].chunk(world)), x_dev[k].cpu(), async_op=True)

    def fence():
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    step(0)      # setup, not warm-up: module load, staging allocation, LDS attribute
    fence()
    for i in range(args.warmup):
        step(i)
    fence()
    handle.kernel_times()   # discard the warm-up launches
    kernel_ms = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
        kernel_ms += handle.kernel_times()   # HIP events of this step's launches (the call has synchronised)
    fence()
    elapsed = time.perf_counter() - t0
    if collective:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=gdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    results = runs[(args.steps - 1) % 2 if collective else 0][1]
    status_all = np.concatenate([r.status for r in results])
    iters_all = np.concatenate([r.iters for r in results])
    solved = int((status_all == 1).sum())
    if collective:
        cnt = torch.tensor([solved, per_step], dtype=torch.int64, device=gdev)
        dist.all_reduce(cnt)
        solved_all, total_all = int(cnt[0]), int(cnt[1])
        if rank == 0:   # the gather really carries this rank's schedules, and they are what came back to the host
            k = (args.steps - 1) % 2
            assert torch.equal(gathered[k][:per_step].to(dev), x_dev[k])
            assert np.array_equal(x_dev[k][:B].cpu().numpy(), results[0].x)
    else:
        solved_all, total_all = solved, per_step

    # kernel-only rate of the same step, outside the timed region: the step's problems resident in HBM, ONE launch of
    # the device-pointer entry, HIP events around it (what `value` would be with free copies)
    kernel_only = None
    if rank == 0:
        from adacharge_amd.backend import DeviceBatch

        dbig = DeviceBatch(ProblemBatch.concatenate(batches), dev)
        ms = []
        for _ in range(3):
            handle.solve_device(dbig, opts, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            ms.append(handle.last_kernel_ms())
        handle.kernel_times()
        kernel_only = {"value": per_step / (min(ms) * 1e-3), "unit": "QP solves/s per GPU", "launch_ms": min(ms),
                       "problems_per_launch": per_step,
                       "note": "device-resident inputs, one launch over the step's problems, HIP events; not the metric"}
        del dbig
    if rank == 0:
        batch = batches[0]
        # launches of one step differ in size (512, 1,024, then 2,048 problems: acn_qp_api.hip, run_pipeline): the
        # roofline is stated per MEAN launch -- mean problems per launch over mean launch duration
        launch_b = per_step * args.steps / max(len(kernel_ms), 1)
        k_avg_ms = float(np.mean(kernel_ms))
        _, per_qp, _ = algorithmic_bytes(launch_b, N, Tm, K, batch.site)
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tfile):   # committed result of the separate rocprofv3 --pmc passes (same command)
            try:
                tj = json.load(open(tfile))
                # launches differ in size: scale the measured bytes per problem to this run's mean launch
                traffic = tj["hbm_bytes_per_problem"] * launch_b if tj.get("hbm_bytes_per_problem") else tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        fl_dense = flops_per_iteration(N, Tm, batch.site)
        whole = ProblemBatch.concatenate(batches)
        fl_sparse = flops_per_iteration_sparse(whole)
        it_sum = float(iters_all.sum())
        # the dominant kernel is timed WITHOUT neighbours: one launch over the step's problems (kernel_only, HIP events on
        # the launch stream); the pipelined launches of the timed region overlap on four streams, so their durations
        # include the CUs they cede to each other and are reported beside it (pipelined_*)
        lone_ms = kernel_only["launch_ms"]
        abytes1, _, _ = algorithmic_bytes(per_step, N, Tm, K, batch.site)
        tf_sparse = it_sum * fl_sparse / (lone_ms * 1e-3) / 1e12
        tf_dense = it_sum * fl_dense / (lone_ms * 1e-3) / 1e12
        host_bytes = per_step * (per_qp - 0)   # H2D of the inputs + D2H of the results = the algorithmic bytes per QP
        out = {
            "metric": "MPC QP solves/sec whole-node, 54 EVSE x horizon 12; max rate residual vs cvxpy",
            "value": world * per_step * args.steps / elapsed,
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
                            f"{args.constraint_type} constraints, batches of {B} independent MPC snapshots "
                            f"(BASELINE.json configs[1]); one step = one acnqp_solve_batches call over {G} such batches "
                            f"per GPU ({per_step} problems), {'pageable' if args.pageable else 'pinned'} host buffers in, "
                            f"host buffers out: H2D + kernels + D2H inside the timed region",
                "batch": B, "batches_per_step_per_gpu": G, "problems_per_step_per_gpu": per_step,
                "n_evse": N, "horizon": T, "constraint_type": args.constraint_type,
                "parallelism": f"dp{world} (own snapshots per GPU" + (", one RCCL all-gather of schedules per step)" if world > 1 else ")"),
            },
            "roofline": {
                # register / LDS-resident iterative solver: HBM is touched once per problem, so the roof that
                # can bind is the fp64 arithmetic one (vector = matrix peak on MI355X); the HBM view is reported beside it
                "bound": "mfma", "binding_unit": "valu (fp64: the vector ALU and the matrix cores share the 78.6 TF peak; one wave per "
                                                 "SIMD issues a double-precision vector instruction every 8 cycles, the matrix pipe is "
                                                 "busy a quarter of the iteration: DESIGN.md 3.1)",
                "achieved": tf_sparse, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
                "frac": tf_sparse / FP64_VALU_PEAK_TF,
                "traffic": None if traffic is None else traffic / launch_b * per_step,
                "kernel": "acnqp::admm_wave_kernel<5, 1, 12, 1>",
                "launch_ms": lone_ms, "problems_per_launch": per_step,
                "flops_per_iteration_sparse": fl_sparse, "flops_per_iteration_dense": fl_dense,
                "achieved_dense_count": tf_dense, "frac_dense_count": tf_dense / FP64_VALU_PEAK_TF,
                "iterations_mean": float(iters_all.mean()), "iterations_max": int(iters_all.max()),
                "hbm": {"algorithmic_bytes_per_launch": abytes1, "bytes_per_qp": per_qp,
                        "achieved_GBs": abytes1 / (lone_ms * 1e-3) / 1e9, "peak_GBs": HBM_PEAK_GBS,
                        "frac": abytes1 / (lone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "traffic_bytes_per_launch": None if traffic is None else traffic / launch_b * per_step},
                "pipelined": {"launches_timed": len(kernel_ms), "problems_per_launch_mean": launch_b, "kernel_avg_ms": k_avg_ms,
                              "note": "launches of the timed region overlap on four streams: a launch's duration includes the "
                                      "CUs it cedes to its neighbours"},
                "note": "achieved = ALGORITHMIC flops (SURVEY.md 8d: 4 nnz(A) + 12 (n + m) + 4 Mg^2 T per iteration, sparse G) x "
                        "iterations run / the duration of ONE non-overlapped launch over the step's problems; the dense count "
                        "is what the MFMA chains execute (G and Q dense)",
            },
            "kernel_only": kernel_only,
            "pcie": {"host_bytes_per_step_per_gpu": int(host_bytes), "achieved_GBs_per_gpu": host_bytes * args.steps / elapsed / 1e9},
            "solver": {
                "solved": solved_all, "problems": total_all,
                "eps_abs": opts.eps_abs, "eps_rel": opts.eps_rel, "reg_rel": opts.reg_rel,
                "anderson_columns": handle.accel_columns(Tm, K, opts),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            # a CPU sample of >= 32 problems per thread: the first 16 batches (4,096 problems) of the timed workload
            ncb = max(1, min(G, 4096 // B))
            cb, parity = cpu_baseline_leg(ProblemBatch.concatenate(batches[:ncb]), np.concatenate([r.x for r in results[:ncb]]),
                                          np.concatenate([r.status for r in results[:ncb]]), args.cpu_seconds, snaps0, infra, iface,
                                          accel_mem=handle.accel_columns(Tm, K, opts), first=B)
            out["cpu_baseline"] = cb
            out["parity"] = parity
        if world == 1 and not args.no_other_configs:
            out["host_inclusive"] = host_inclusive_leg(infra, iface, objective, site, handle, opts, horizon=T)
            out["polish"] = handle.polish_stats()
            out["strict_batch256"] = strict_batch256_leg(handle, batches[0], opts)
            out["other_configs"] = other_configs_leg(dev)
        print(json.dumps(out), file=json_out, flush=True)
    if collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
