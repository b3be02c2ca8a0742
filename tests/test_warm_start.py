"""Warm start across MPC steps (extension; the reference keeps no state between calls, adacharge.py:152-158): the
previous step's schedule and site-row multipliers, shifted by one period, start the next solve.  It must change the iteration
count, never the answer.  CPU: the C twin.  GPU: the HIP path through the C ABI (acnqp_problems.warm_x / warm_y,
acnqp_results.y) and the adapter option ``AdaptiveSchedulingAlgorithm(warm_start=True)``."""
import numpy as np
import pytest

from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd.acn import Interface
from adacharge_amd.builder import build_batch
from tests import helpers as H

OBJ = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-3)]


def _loop(solve, steps=8, seed=11):
    """Run a congested closed loop twice over the same fleet history: cold and warm.  ``solve(batch, warm)`` ->
    (x (N, Tm), y (N, Tm), iters)."""
    infra = sites.caltech54()
    iface = Interface({"infrastructure_info": infra, "period": 5})
    out = {}
    for mode in ("cold", "warm"):
        evs = H.closed_loop_fleet(infra, np.random.default_rng(seed), n_evs=50, t_span=6)
        prev, iters, xs = None, [], []
        for t in range(4, 4 + steps):
            sl = H.closed_loop_sessions(evs, t)
            batch = build_batch([sl], infra, iface, OBJ, "SOC")
            warm = None
            if mode == "warm" and prev is not None:   # last step's schedule and site-row multipliers, one period later
                x0 = np.zeros((1, batch.N, batch.Tm)); y0 = np.zeros((1, batch.site.Mg, batch.Tm))
                T = min(batch.Tm, prev[0].shape[1] - 1)
                x0[0, :, :T], y0[0, :, :T] = prev[0][:, 1:1 + T], prev[1][:, 1:1 + T]
                warm = (x0, y0)
            x, y, it = solve(batch, warm)
            prev = (x, y)
            iters.append(it)
            xs.append(x[:, : int(batch.T[0])].copy())
            H.closed_loop_apply(evs, t, x[:, 0], infra)
        out[mode] = (iters, xs)
    return out


def _check(out):
    cold_it, cold_x = out["cold"]
    warm_it, warm_x = out["warm"]
    for a, b in zip(cold_x, warm_x):   # same fleet history (the applied first-period rates agree to solver tolerance)
        assert a.shape == b.shape and np.abs(a - b).max() <= 2e-3
    assert warm_it[0] == cold_it[0]                       # the first step has nothing to start from
    assert sum(warm_it[1:]) <= 0.8 * sum(cold_it[1:]), (cold_it, warm_it)
    return cold_it, warm_it


def test_c_twin_warm_start_cuts_iterations_not_the_answer():
    from oracle import admm_port

    def solve(batch, warm):
        kw = {} if warm is None else dict(warm_x=warm[0], warm_y=warm[1])
        o = admm_port.solve_batch(batch, accel_mem=5, **kw)
        assert o["status"][0] == 1
        return o["x"][0], o["y"][0], int(o["iters"][0])

    cold, warm = _check(_loop(solve))
    print("C twin closed loop: cold", cold, "warm", warm)


@pytest.mark.gpu
def test_hip_warm_start_matches_the_twin_and_cuts_iterations():
    from adacharge_amd.backend import SiteHandle, default_options
    from oracle import admm_port

    handles = {}

    def solve(batch, warm):
        h = handles.setdefault("h", SiteHandle(batch.site, 0))
        r = h.solve(batch, default_options(), warm=warm, want_y=True)
        assert r.status[0] == 1
        # the C twin from the same starting point: plain iteration near-bitwise, accelerated to solver tolerance
        kw = {} if warm is None else dict(warm_x=warm[0], warm_y=warm[1])
        plain = h.solve(batch, default_options(accel_mem=0), warm=warm, want_y=True)
        ref = admm_port.solve_batch(batch, accel_mem=0, **kw)
        assert ref["iters"][0] == plain.iters[0] and np.abs(ref["x"] - plain.x).max() <= 1e-6
        assert np.abs(ref["y"] - plain.y).max() <= 1e-6 * max(1.0, np.abs(ref["y"]).max())
        return r.x[0], r.y[0], int(r.iters[0])

    cold, warm = _check(_loop(solve))
    print("HIP closed loop: cold", cold, "warm", warm)


@pytest.mark.gpu
def test_adapter_warm_start_option():
    from adacharge_amd import AdaptiveSchedulingAlgorithm

    infra = sites.caltech54()
    iters = {}
    for ws in (False, True):
        evs = H.closed_loop_fleet(infra, np.random.default_rng(5), n_evs=50, t_span=6)
        iface = Interface({"infrastructure_info": infra, "period": 5, "current_time": 0, "active_sessions": []})
        alg = AdaptiveSchedulingAlgorithm(OBJ, warm_start=ws)
        alg.register_interface(iface)
        its = []
        for t in range(4, 12):
            iface.data["current_time"] = t
            iface.data["active_sessions"] = H.closed_loop_sessions(evs, t)
            sched = alg.run()
            its.append(alg.last_iterations)
            H.closed_loop_apply(evs, t, np.array([sched[s][0] for s in infra.station_ids]), infra)
            assert iface.is_feasible({k: v[:1] for k, v in sched.items()})
        iters[ws] = its
    print("adapter closed loop: cold", iters[False], "warm", iters[True])
    assert sum(iters[True][1:]) <= 0.9 * sum(iters[False][1:]), iters
