"""The numpy restatement of the device algorithm (oracle/admm_ref.py) against
the independent IPM oracle's golden vectors, and its projection against brute
force -- so a kernel/oracle disagreement can be attributed."""
import numpy as np
import pytest

from adacharge_amd import ObjectiveComponent, equal_share, quick_charge
from adacharge_amd.builder import build_batch
from oracle.admm_ref import AdmmOptions, project_window, solve_one
from tests.helpers import caltech_interface, golden_case, load_golden


@pytest.mark.parametrize("key", ["c01", "c03"])
def test_admm_restatement_matches_golden(key):
    g = load_golden()
    infra, iface = caltech_interface()
    sl, meta, exp = golden_case(g, key)
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, meta["es"])]
    batch = build_batch([sl], infra, iface, obj, meta["ct"], meta["eq"])
    out = solve_one(batch, 0, AdmmOptions(eps_abs=1e-8, eps_rel=1e-8, rho=0.01, max_iter=5000))
    assert out["status"] == 1
    assert np.abs(out["x"][:, : meta["T"]] - exp["rates"]).max() <= 1e-4 * 32


def test_project_window_is_the_euclidean_projection():
    rng = np.random.default_rng(0)
    from scipy.optimize import minimize

    for eq in (False, True):
        for _ in range(10):
            L = int(rng.integers(1, 7))
            v = rng.normal(10, 15, size=L)
            lb = rng.uniform(0, 4, size=L) * (rng.random(L) < 0.3)
            ub = lb + rng.uniform(1, 30, size=L)
            cap = rng.uniform(lb.sum(), ub.sum())
            z = project_window(v, lb, ub, cap, eq)
            assert (z >= lb - 1e-12).all() and (z <= ub + 1e-12).all()
            assert z.sum() <= cap + 1e-9 and (not eq or abs(z.sum() - cap) < 1e-9)
            cons = [{"type": "eq" if eq else "ineq", "fun": lambda x: cap - x.sum()}]
            ref = minimize(lambda x: 0.5 * ((x - v) ** 2).sum(), np.clip(v, lb, ub), jac=lambda x: x - v,
                           bounds=list(zip(lb, ub)), constraints=cons, method="SLSQP",
                           options=dict(ftol=1e-14, maxiter=200))
            assert np.abs(z - ref.x).max() < 1e-5


@pytest.mark.parametrize("ct,mem", [("SOC", 0), ("SOC", 5), ("LINEAR", 5), ("SOC", 10)])
def test_numpy_twin_and_c_port_run_the_same_algorithm(ct, mem):
    """Two independent restatements of the device algorithm (readable numpy, scalar C) must agree to
    rounding, with and without Anderson acceleration: same iteration counts, same schedules."""
    from adacharge_amd import sites
    from oracle import admm_port

    infra, iface = caltech_interface()
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-3)]
    batch = build_batch(sites.snapshot_batch(infra, 12, 3, seed=7), infra, iface, obj, ct)
    port = admm_port.solve_batch(batch, threads=3, accel_mem=mem)
    for b in range(batch.B):
        twin = solve_one(batch, b, AdmmOptions(eps_abs=1e-8, eps_rel=1e-8, reg_rel=0.06, accel_mem=mem))
        assert twin["status"] == 1 and port["status"][b] == 1
        assert abs(int(twin["iters"]) - int(port["iters"][b])) <= 20
        assert np.abs(twin["x"] - port["x"][b]).max() <= 1e-6


def test_anderson_acceleration_halves_the_iterations_and_keeps_the_answer():
    """On the headline problem class (54 EVSE x 12 periods, LP + tiny equal-share term) the accelerated
    iteration needs clearly fewer steps and ends at the same schedule to solver tolerance."""
    from adacharge_amd import sites
    from oracle import admm_port

    infra, iface = caltech_interface()
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
    batch = build_batch(sites.snapshot_batch(infra, 12, 48, seed=20240), infra, iface, obj, "SOC")
    plain = admm_port.solve_batch(batch, threads=8, accel_mem=0)
    fast = admm_port.solve_batch(batch, threads=8, accel_mem=5)
    assert (plain["status"] == 1).all() and (fast["status"] == 1).all()
    assert fast["iters"].mean() <= 0.7 * plain["iters"].mean()
    assert np.abs(fast["x"] - plain["x"]).max() <= 1e-4 * 32
    assert np.abs(fast["obj"] - plain["obj"]).max() <= 1e-7 * np.abs(plain["obj"]).max()


def test_c_port_certifies_infeasible_problems():
    """The reference's two infeasible scenarios (t_aco.py:119-175: a deadline too short, a network too small for
    an energy *equality*) end with the certificate status in the C port, like on the device, instead of max_iter."""
    from oracle import admm_port
    from tests.acn_testing import TestingInterface, session_generator, single_phase_single_constraint

    for kw in (dict(departures=[12, 4]), dict(limit=30)):
        N = 2
        dep = kw.get("departures", [12, 12])
        sd = session_generator(N, [0] * N, dep, [3.3] * N, [3.3] * N, [32] * N)
        net = single_phase_single_constraint(N, kw.get("limit", 64))
        iface = TestingInterface({"active_sessions": sd, "infrastructure_info": net, "current_time": 0, "period": 5})
        batch = build_batch([iface.active_sessions()], iface.infrastructure_info(), iface,
                            [ObjectiveComponent(quick_charge)], "SOC", True)
        if batch.presolve_status[0]:
            continue   # caught before iterating (a session's own bounds cannot meet its energy row)
        out = admm_port.solve_batch(batch, threads=1, accel_mem=5)
        # an empty session set (4) is caught before iterating, the network case by the certificate (3)
        assert out["status"][0] in (3, 4) and out["iters"][0] < 5000, (kw, out["status"], out["iters"])
        if "limit" in kw:
            assert out["status"][0] == 3
