"""The wave-per-problem kernel (acn_qp_wave.hpp) against the register-resident tiled kernel it
replaces on the headline shape -- the same algorithm in another data layout -- and against the C twin.  Each case of
tests/wave_cases.py is solved in two child processes (ACNQP_WAVE_MIN_BATCH=1: the wave kernel whatever the launch size;
ACNQP_NO_WAVE=1: the tiled kernel); in this process the shape takes the wave kernel by default."""
import os
import subprocess
import sys

import numpy as np
import pytest

from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.builder import build_batch
from tests import helpers as H
from tests import wave_cases

RATE_TOL = 1e-4 * 32.0   # north_star: rates within 1e-4 of the maximum rate


def _both(tmp_path, name):
    out = {}
    for tag, env in (("wave", {"ACNQP_WAVE_MIN_BATCH": "1"}), ("tiled", {"ACNQP_NO_WAVE": "1"})):
        f = tmp_path / f"{name}_{tag}.npz"
        e = {k: v for k, v in os.environ.items() if k not in ("ACNQP_WAVE_MIN_BATCH", "ACNQP_NO_WAVE")}
        subprocess.run([sys.executable, wave_cases.__file__, name, str(f)], check=True, env=dict(e, **env), timeout=900)
        out[tag] = np.load(f)
    return out["wave"], out["tiled"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["soc", "linear", "equality", "short", "site30", "peak", "general_windows",
                                  "h24", "h18_linear", "h24_equality", "h20_windows",
                                  "mt2_site36", "mt2_site64", "mt2_short", "mt2_equality", "mt2_h24", "mt2_h17", "mt2_t4", "h13",
                                  "h48", "h36_linear", "h40_equality", "flat_linear", "flat_soc", "flat_h24",
                                  "dc_linear", "dc_soc", "dc_h24"])
def test_wave_kernel_agrees_with_the_tiled_kernel(tmp_path, name):
    """Same statuses, schedules within the rate tolerance, iteration counts that differ only where rounding moved a
    residual check (the two kernels sum in different orders): feasible shapes the wave kernel routes -- SOC and LINEAR
    rows, energy equalities, a short horizon with minimum rates, a 30-EVSE site, a peak row, windows that start late;
    and horizons 13 ... 24, where TWO waves share a problem (twelve periods each) and exchange the sums that cross the
    halves through an LDS mailbox; and sites of 17 ... 32 rows (two row tiles), where the two waves hold six periods each; and the prox rows of load_flattening and demand_charge."""
    w, t = _both(tmp_path, name)
    assert np.array_equal(w["status"], t["status"]), (w["status"], t["status"])
    assert (w["status"] == 1).all()
    assert np.abs(w["x"] - t["x"]).max() <= RATE_TOL
    same = (w["iters"] == t["iters"]).mean()
    assert same >= 0.85 and abs(w["iters"].mean() - t["iters"].mean()) <= 0.03 * t["iters"].mean(), (same, w["iters"].mean(), t["iters"].mean())
    assert np.allclose(w["obj"], t["obj"], rtol=1e-6, atol=1e-6 * np.abs(t["obj"]).max())


@pytest.mark.gpu
def test_wave_kernel_certifies_the_same_infeasible_problems(tmp_path):
    """Energy equalities the site cannot carry once enough EVSEs are busy: the same problems end INFEASIBLE (the
    certificate) on both kernels, the rest are solved to the same schedules; and a batch whose sessions cannot be served
    inside their own bounds ends EMPTY_SET with an all-zero schedule on both."""
    for case in ("infeasible", "h24_infeasible", "mt2_infeasible"):
        w, t = _both(tmp_path, case)
        assert np.array_equal(w["status"], t["status"]), case
        assert (w["status"] == 3).sum() >= 10 and np.isin(w["status"], (1, 3)).all(), (case, np.bincount(w["status"]))
        ok = w["status"] == 1
        assert ok.sum() >= 10 and np.abs(w["x"][ok] - t["x"][ok]).max() <= RATE_TOL, case
    w4, t4 = _both(tmp_path, "empty_set")
    assert (w4["status"] == 4).all() and (t4["status"] == 4).all() and (w4["x"] == 0).all() and (w4["iters"] == 0).all()


@pytest.mark.gpu
def test_wave_kernel_warm_start_and_multipliers(tmp_path):
    """A warm start (schedule + site-row multipliers of an earlier solve) is taken and shortens the solve on both kernels
    alike; the multipliers returned agree."""
    for case in ("warm", "h24_warm", "mt2_warm"):
        w, t = _both(tmp_path, case)
        assert (w["status"] == 1).all() and (t["status"] == 1).all(), case
        assert np.abs(w["x"] - t["x"]).max() <= RATE_TOL, case
        scale = max(1.0, np.abs(t["y"]).max())
        # (two row tiles: 18 rows of a small site, some of them redundant -- the multipliers are unique to 1e-4 only)
        assert np.abs(w["y"] - t["y"]).max() <= (1e-3 if case.startswith("mt2") else 1e-5) * scale, case
        assert abs(w["iters"].mean() - t["iters"].mean()) <= 0.1 * t["iters"].mean(), case


@pytest.mark.gpu
def test_wave_kernel_hands_stalled_problems_to_the_polish(tmp_path):
    """The congested horizon-12 fixtures of tests/golden/stalled.npz: the wave kernel leaves them to the polish kernel
    after polish_iters iterations like the tiled kernel does (three launches: solver, polish, resume), every case SOLVED
    within the rate tolerance of its certificate."""
    w, t = _both(tmp_path, "stalled")
    assert (w["status"] == 1).all() and (t["status"] == 1).all()
    assert w["iters"].max() <= 800 + 96 and t["iters"].max() <= 800 + 96
    g = H.load_stalled()
    names = [str(n) for n in g["names"] if int(g[f"{n}_meta"][0]) == 12]
    for b, n in enumerate(names):
        exp = g[f"{n}_rates"]
        assert np.abs(w["x"][b][:, :exp.shape[1]] - exp).max() <= RATE_TOL, n


@pytest.mark.gpu
def test_default_routing_is_by_shape_and_matches_the_twin():
    """In this process (no diagnostic variable set) the headline shape runs on the wave kernel whatever the launch size:
    256 problems solved alone give the bits they give inside a launch of 1,024 (a problem's result does not depend on what
    it is batched with), the launch follows the C twin iteration for iteration on most problems, and its duration is a
    wave's -- four problems in flight per CU."""
    import torch

    from adacharge_amd.backend import DeviceBatch, SiteHandle, default_options
    from oracle import admm_port

    assert not os.environ.get("ACNQP_NO_WAVE") and not os.environ.get("ACNQP_WAVE_MIN_BATCH")
    infra, iface = H.caltech_interface()
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
    batch = build_batch(sites.snapshot_batch(infra, 12, 1024, seed=20240), infra, iface, obj, "SOC")
    h = SiteHandle(batch.site, 0)
    opts = default_options(polish_iters=0)
    big = h.solve(batch, opts)
    small = h.solve(batch.subset(slice(0, 256)), opts)
    assert (big.status == 1).all()
    assert np.array_equal(big.x[:256], small.x) and np.array_equal(big.iters[:256], small.iters)
    ref = admm_port.solve_batch(batch.subset(slice(0, 256)), threads=8, accel_mem=5)
    assert (small.iters == ref["iters"]).mean() >= 0.9
    assert np.abs(small.x - ref["x"]).max() <= RATE_TOL
    dev = DeviceBatch(batch, "cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    h.solve_device(dev, opts, stream=st); torch.cuda.synchronize()
    h.solve_device(dev, opts, stream=st); torch.cuda.synchronize()
    ms_big = h.last_kernel_ms()
    assert np.array_equal(dev.x.cpu().numpy(), big.x)
    h.close()
    # 1,024 problems, one per wave, four per CU: the slowest problem's ~800 iterations at ~4.3 us (the tiled kernel, two
    # problems per CU, needed two rounds of them: 7.7 ms at 4,096)
    assert ms_big < 5.0, ms_big


@pytest.mark.gpu
def test_polish_stall_option_shortens_a_scenario_launch_and_keeps_the_optimum():
    """options.polish_stall (ABI v9): from polish_iters / 2 on, a problem whose residual score has stood still for that
    many iterations goes to the polish at once.  On the scenario MPC of BASELINE configs[3] -- 1,024 demand scenarios of
    one site, a wavefront each, so the launch lasts as long as its slowest problem -- the option cuts the longest ADMM run
    and the launch; every problem is still SOLVED, at the same (unique: equal_share 1e-3) optimum, and the polish is
    the one that did the extra work.  Off by default: the same launch at default options does not change."""
    import torch

    from adacharge_amd.acn import Interface
    from adacharge_amd.backend import DeviceBatch, SiteHandle, default_options
    from adacharge_amd.builder import scenario_batch

    infra = sites.eight_sites()[0]
    iface = Interface({"infrastructure_info": infra, "period": 5})
    rng = np.random.default_rng(500)
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-3)]
    base = build_batch([sites.random_sessions(infra, 12, rng)], infra, iface, obj, "SOC")
    batch = scenario_batch(base, rng.lognormal(0.0, 0.25, size=(1024, base.K, base.N)))
    h = SiteHandle(batch.site, 0)
    assert default_options().polish_stall == 0
    with pytest.raises(ValueError):
        h.solve(batch.subset(slice(0, 4)), default_options(polish_stall=-1))
    dev = DeviceBatch(batch, "cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    for tag, opts in (("default", default_options()), ("early", default_options(polish_stall=100))):
        before = h.polish_stats()
        h.solve_device(dev, opts, stream=st); torch.cuda.synchronize()
        tried = h.polish_stats()["attempted"] - before["attempted"]
        h.solve_device(dev, opts, stream=st); torch.cuda.synchronize()
        out[tag] = (dev.x.cpu().numpy().copy(), dev.iters.cpu().numpy().copy(), dev.status.cpu().numpy().copy(), h.last_kernel_ms(), tried)
    h.close()
    (xd, itd, std_, msd, trd), (xe, ite, ste, mse, tre) = out["default"], out["early"]
    assert (std_ == 1).all() and (ste == 1).all()
    assert np.abs(xd - xe).max() <= RATE_TOL
    assert tre > trd and ite.max() < itd.max(), (trd, tre, itd.max(), ite.max())
    assert mse < 0.95 * msd, (msd, mse)
