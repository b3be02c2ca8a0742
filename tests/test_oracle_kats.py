"""Pins the oracle (oracle/ref_problem.py + oracle/ipm.py) on every
known-answer the reference's own tests allow without cvxpy:
KAT-1 closed form (t_aco.py:87-116), KAT-2 infeasible pair (t_aco.py:119-175),
KAT-3 TOU (t_aco.py:469-507), KAT-4 stress aggregate (t_aco.py:286-313), plus
scipy-HiGHS on LP instances and the committed golden fixtures."""
import numpy as np
import pytest

from adacharge_amd import sites
from oracle import ipm
from oracle.ref_problem import build_reference_problem
from tests.acn_testing import (
    TestingInterface, session_generator, single_phase_single_constraint, three_phase_balanced_network,
)
from tests.helpers import caltech_interface, golden_case, load_golden

QC = [("quick_charge", 1, {})]


def tiny(limit=64, departures=(12, 12), arrivals=(0, 0), prices=None, current_time=0):
    sd = session_generator(2, list(arrivals), list(departures), [3.3] * 2, [3.3] * 2, [32] * 2)
    infra = single_phase_single_constraint(2, limit)
    d = {"active_sessions": sd, "infrastructure_info": infra, "current_time": current_time, "period": 5}
    if prices is not None:
        d["prices"] = prices
    return TestingInterface(d)


KAT1_ROW = np.array([32.0] * 5 + [3.3 / (208 * 5 / 60 / 1e3) - 160] + [0.0] * 6)


@pytest.mark.parametrize("ct", ["SOC", "LINEAR"])
@pytest.mark.parametrize("eq", [False, True])
def test_kat1_closed_form(ct, eq):
    iface = tiny()
    prob = build_reference_problem(iface.active_sessions(), iface.infrastructure_info(), iface, QC, ct, eq)
    rates, res, cert = ipm.solve_certified(prob)
    assert cert.worst < 1e-9
    assert np.allclose(rates, np.stack([KAT1_ROW] * 2), atol=1e-7)
    assert abs(KAT1_ROW[5] - 30.384615384615387) < 1e-9


@pytest.mark.parametrize("kw", [dict(departures=(12, 4)), dict(limit=30)])
def test_kat2_infeasible(kw):
    iface = tiny(**kw)
    prob = build_reference_problem(iface.active_sessions(), iface.infrastructure_info(), iface, QC, "LINEAR", True)
    assert ipm.solve_lp_highs(prob).status == 2  # HiGHS: infeasible
    _, res = ipm.solve_reference_problem(prob, max_iter=60)
    assert res.status not in ("optimal", "optimal_inaccurate")


def test_kat3_tou_no_charging_in_expensive_periods():
    iface = tiny(prices=np.array([0.3] * 6 + [0.1] * 6))
    prob = build_reference_problem(
        iface.active_sessions(), iface.infrastructure_info(), iface, [("tou_energy_cost", 1, {})], "SOC", True
    )
    rates, res, cert = ipm.solve_certified(prob)
    assert np.allclose(rates[:, :6], 0, atol=1e-6)
    assert np.allclose(rates[:, 6:].sum(axis=1), 3.3 / (208 * 5 / 60 / 1e3), atol=1e-6)


def test_kat4_stress_aggregate():
    N, T = 54, 144
    sd = session_generator(N, [0] * N, [T] * N, [10] * N, [10] * N, [32] * N)
    infra = single_phase_single_constraint(N, 32 * N / 3)
    iface = TestingInterface({"active_sessions": sd, "infrastructure_info": infra, "current_time": 0, "period": 5})
    prob = build_reference_problem(iface.active_sessions(), iface.infrastructure_info(), iface, QC, "LINEAR")
    h = ipm.solve_lp_highs(prob)
    assert h.status == 0
    agg = h.x.reshape(N, T).sum(0)
    assert np.allclose(agg[:54], 576.0, atol=1e-6)
    assert abs(agg[54] - 49.846153846) < 1e-6
    assert np.allclose(agg[55:], 0, atol=1e-6)
    assert abs(h.fun - (-25411.153846)) < 1e-5


def test_ipm_matches_highs_on_lp():
    infra, iface = caltech_interface()
    for seed in (0, 1):
        sl = sites.random_sessions(infra, 12, np.random.default_rng(seed))
        prob = build_reference_problem(sl, infra, iface, QC, "LINEAR")
        h = ipm.solve_lp_highs(prob)
        rates, res = ipm.solve_reference_problem(prob)
        assert res.status == "optimal"
        assert abs(res.pcost - h.fun) <= 1e-7 * abs(h.fun)
        assert np.abs(rates.sum(0) - h.x.reshape(54, 12).sum(0)).max() < 1e-5


def test_golden_fixtures_reproduce():
    """The committed vectors are what the oracle produces today (two cases)."""
    g = load_golden()
    infra, iface = caltech_interface()
    for key in ("c02", "c13"):
        sl, meta, exp = golden_case(g, key)
        spec = [("quick_charge", 1, {}), ("equal_share", meta["es"], {})]
        prob = build_reference_problem(sl, infra, iface, spec, meta["ct"], meta["eq"])
        rates, res, cert = ipm.solve_certified(prob)
        assert cert.worst < 1e-9
        assert np.abs(rates - exp["rates"]).max() < 1e-7
        assert abs(prob.objective(rates) - float(exp["obj"])) < 1e-8


def test_three_phase_soc_vs_linear_differ():
    """SOC admits more current than LINEAR on mixed-phase rows."""
    N, T = 6, 4
    sd = session_generator(N, [0] * N, [T] * N, [10] * N, [10] * N, [32] * N)
    infra = three_phase_balanced_network(N // 3, 40)
    iface = TestingInterface({"active_sessions": sd, "infrastructure_info": infra, "current_time": 0, "period": 5})
    out = {}
    for ct in ("SOC", "LINEAR"):
        prob = build_reference_problem(
            iface.active_sessions(), iface.infrastructure_info(), iface,
            QC + [("equal_share", 1e-3, {})], ct,
        )
        rates, res, cert = ipm.solve_certified(prob)
        assert cert.worst < 1e-8
        out[ct] = rates.sum()
    assert out["SOC"] > out["LINEAR"] + 1.0


def test_cone_algebra():
    c = ipm._Cone(2, 3)
    rng = np.random.default_rng(0)

    def interior():
        u = rng.normal(size=c.m)
        u[:2] = abs(u[:2]) + 0.1
        q = u[2:].reshape(3, 3)
        q[:, 0] = np.linalg.norm(q[:, 1:], axis=1) + abs(rng.normal(size=3)) + 0.1
        return u

    s, z = interior(), interior()
    Winv, aW, aWi, lam = c.nt_scaling(s, z)
    assert np.abs(aW(z) - aWi(s)).max() < 1e-12
    assert np.abs(Winv @ aW(z) - z).max() < 1e-12
    u, v = interior(), rng.normal(size=c.m)
    assert np.abs(c.prod(u, c.div(u, v)) - v).max() < 1e-12
    d = rng.normal(size=c.m)
    a = c.max_step(u, d)
    assert abs(c.min_eig(u + a * d)) < 1e-10
