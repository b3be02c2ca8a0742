"""The JSON line bench.py prints is a contract with the driver: check the committed record of the last GPU run
(profiles/r02_bench.json, written by `python bench.py` on an MI355X) and the bookkeeping helpers, on the CPU."""
import json
import os

import numpy as np

import bench
from adacharge_amd import sites
from adacharge_amd.builder import make_site

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r02_bench.json")))
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
    # roofline of the dominant kernel: algorithmic bytes of one launch / its HIP-event duration
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_avg_ms"] * 1e-3) / 1e9) <= 1e-9 * r["achieved"]


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
    assert np.isclose(bench.HBM_PEAK_GBS, 8000.0)
