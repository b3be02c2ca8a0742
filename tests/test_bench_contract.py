"""The JSON line bench.py prints is a contract with the driver: check the committed record of the last GPU run
(profiles/r04_bench.json, written by `python bench.py` on an MI355X) and the bookkeeping helpers, on the CPU."""
import json
import os

import numpy as np
import pytest

import bench
from adacharge_amd import sites
from adacharge_amd.builder import make_site

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "QP solves/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1
    # value = problems of all timed steps / wall time; a step = one acnqp_solve_batches call over batches x batch problems
    cfg = d["config"]
    assert cfg["problems_per_step_per_gpu"] == cfg["batch"] * cfg["batches_per_step_per_gpu"] and cfg["batch"] == 256
    assert abs(d["value"] - d["n_gpus"] * cfg["problems_per_step_per_gpu"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert d["ms_per_step"] * d["steps"] >= 500.0          # the timed region is long enough for the driver's clock
    assert "H2D" in cfg["workload"] and "D2H" in cfg["workload"]   # the metric SURVEY.md section 8d defines
    assert d["solver"]["solved"] == d["solver"]["problems"]
    # roofline of the dominant kernel: ALGORITHMIC flops (sparse count of SURVEY 8d) x iterations run / the duration of
    # ONE non-overlapped launch over the step's problems; the HBM view beside it
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 78.6
    its = r["iterations_mean"] * r["problems_per_launch"]
    assert abs(r["achieved"] - its * r["flops_per_iteration_sparse"] / (r["launch_ms"] * 1e-3) / 1e12) <= 1e-9 * r["achieved"]
    assert r["flops_per_iteration_sparse"] < r["flops_per_iteration_dense"] and r["frac"] < r["frac_dense_count"]
    assert abs(r["launch_ms"] - d["kernel_only"]["launch_ms"]) < 1e-12
    h = r["hbm"]
    assert abs(h["achieved_GBs"] - h["algorithmic_bytes_per_launch"] / (r["launch_ms"] * 1e-3) / 1e9) <= 1e-9 * h["achieved_GBs"]
    assert r["traffic"] is None or r["traffic"] > h["algorithmic_bytes_per_launch"]
    # the other BASELINE.json configurations are in the driver's line
    legs = d["other_configs"]
    for name in ("cfg2_caltech54_T24_b4096", "cfg2_jpl52_T24_b4096", "cfg3_site3_T12_b1024", "cfg4_synth512_T48_b2048", "stress_caltech54_T144_b256",
                 "stress_caltech54_T144_b2048"):
        leg = legs[name]
        for k in ("kernel_ms", "iters_mean", "iters_max", "solved", "algorithmic_bytes_per_launch", "hbm_frac", "fp64_frac"):
            assert k in leg, (name, k)
        assert leg["solved"] == leg["batch"] and leg["failed"] == 0, name
    assert legs["cfg4_synth512_T48_b2048"]["iters_mean"] >= 100      # instances that need more than one residual check
    assert legs["cfg4_synth512_T48_b2048"]["anderson_columns"] == 5
    sb = d["strict_batch256"]
    assert sb["batch"] == 256 and sb["qps"] > 0 and sb["ms_per_call_median"] > sb["kernel_ms_median"]
    # round 4: the straggler tail is gone from the congested configs[3] site (<= 2,500 iterations, <= 15 ms)
    s3 = legs["cfg3_site3_T12_b1024"]
    assert s3["iters_max"] <= 2500 and s3["kernel_ms"] <= 15.0, s3
    # ... and the early hand-over (options.polish_stall, ABI v9) is reported beside the default for the scenario legs
    e3 = s3["polish_stall_100"]
    assert e3["solved"] == s3["batch"] and e3["iters_max"] <= s3["iters_max"] and e3["kernel_ms"] <= 1.05 * s3["kernel_ms"], (s3, e3)
    # ... the host side is in the driver's line (item 4: >= 200 k QP/s from a SessionTable, the builder inside the clock)
    hi = d["host_inclusive"]
    assert hi["table"]["solved"] == hi["table"]["snapshots"] == 16384 and hi["table"]["qps"] >= 200e3, hi["table"]
    assert hi["table"]["h2d_bytes_per_problem"] < 0.3 * r["hbm"]["bytes_per_qp"]
    assert hi["sessions"]["solved"] == hi["sessions"]["snapshots"]
    # configs[0]'s call pattern: one schedule() per MPC step; a step of the drop-in stays well under the 5-minute period
    # and under what one ECOS solve of this size takes on a CPU (tens of milliseconds)
    ss = hi["single_step"]
    assert ss["calls"] >= 100 and 0 < ss["ms_min"] <= ss["ms_median"] <= ss["ms_p95"] and ss["ms_median"] < 50.0, ss
    # ... the polish's counters, and a CPU baseline that is not tail-bound (item 7: >= 32 problems per thread)
    pol = d["polish"]
    assert pol["attempted"] > 0 and pol["solved"] >= 0.98 * pol["attempted"], pol
    assert c["problems_per_thread_per_pass"] >= 32 and c["one_thread_qps"] > 0 and c["highs_linear_ms_per_solve_1thread"] > 0


def test_bench_gpus_flag_needs_matching_world_size(monkeypatch):
    """`--gpus N` with a WORLD_SIZE that disagrees is refused before torch / HIP are touched."""
    import pytest
    import sys

    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert "WORLD_SIZE" in str(exc.value)


def test_algorithmic_bytes_and_flops_bookkeeping():
    infra = sites.caltech54()
    site = make_site(infra, "SOC", False, False, False)
    total, per_qp, site_bytes = bench.algorithmic_bytes(256, 54, 12, 1, site)
    assert per_qp == 8 * 4 * 54 * 12 + 16 * 1 * 54 + 13 + 32 == 21645
    assert total == 256 * per_qp + site_bytes == 5557184
    assert bench.flops_per_iteration(54, 12, site) == 69024
    # SURVEY 8d's sparse count on the headline workload: 4 nnz(A) + 12 (n + m) + 4 Mg^2 T ~ 51 k per iteration
    from adacharge_amd import ObjectiveComponent, equal_share, quick_charge
    from adacharge_amd.acn import Interface
    from adacharge_amd.builder import build_batch

    iface = Interface({"infrastructure_info": infra, "period": 5})
    b = build_batch(sites.snapshot_batch(infra, 12, 64, seed=20240), infra, iface,
                    [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)], "SOC")
    assert 49e3 < bench.flops_per_iteration_sparse(b) < 53e3
    assert 55e3 < bench.streamed_bytes_per_iteration(b) < 60e3
    assert np.isclose(bench.HBM_PEAK_GBS, 8000.0)


@pytest.mark.gpu
@pytest.mark.parametrize("flags,n_gpus", [
    (["--rehearse-collective"], 1),                                   # the nccl (= RCCL) branch with one rank
    (["--gpus", "2", "--dist-backend", "gloo", "--one-device"], 2),   # two ranks on the one GPU, gloo carries the gather
])
def test_bench_multi_rank_path_runs(flags, n_gpus):
    """the N > 1 code of bench.py (process group, x_dev, the per-step all-gather overlapped with the
    next solve, the max-over-ranks clock, the gather assert) is run on every GPU test pass, so it cannot rot unseen
    until an 8-GPU node appears.  bench.py is started as a CHILD process (for --gpus 2 it spawns its ranks itself,
    before anything in it touches HIP).  No scaling number comes out of this: both ranks share one GPU."""
    import subprocess
    import sys

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batches", "8",
           "--no-cpu-baseline", "--no-other-configs", *flags]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == n_gpus and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["solver"]["solved"] == d["solver"]["problems"] == n_gpus * 8 * 256
    assert "all-gather" in d["config"]["parallelism"] or n_gpus == 1
    assert d["value"] > 0 and d["roofline"]["frac"] > 0
