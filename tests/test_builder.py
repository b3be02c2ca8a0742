"""The structured builder (adacharge_amd/builder.py) against the oracle's
explicit restatement of aco.py (oracle/ref_problem.py): same objective value
and same feasible set on random points, plus the edge rules of aco.py."""
import numpy as np
import pytest

from adacharge_amd import (
    ObjectiveComponent, equal_share, load_flattening, quick_charge, total_energy, tou_energy_cost, sites,
)
from adacharge_amd.acn import Interface, SessionInfo
from adacharge_amd.builder import CONE_SOC, build_batch, make_site
from oracle.ref_problem import build_reference_problem
from tests.acn_testing import TestingInterface, session_generator, single_phase_single_constraint
from tests.helpers import caltech_interface


def structured_objective(batch, b, x):
    T = int(batch.T[b])
    xs = x[:, :T]
    return 0.5 * batch.pdiag[b] * (xs ** 2).sum() + (batch.q[b][:, :T] * xs).sum()


def structured_violation(batch, b, x, infra):
    """max violation of every structured constraint at x (N, T)."""
    T = int(batch.T[b])
    v = [0.0, (batch.lb[b][:, :T] - x).max(), (x - batch.ub[b][:, :T]).max()]
    for k in range(batch.K):
        for i in range(batch.N):
            L = batch.s_len[b, k, i]
            if L:
                o = batch.s_off[b, k, i]
                e = x[i, o : o + L].sum() - batch.s_cap[b, k, i]
                v.append(abs(e) if batch.s_eq[b] else e)
    site = batch.site
    Gx = site.G @ x
    M = site.M
    if site.cone == CONE_SOC:
        v.append((np.hypot(Gx[:M], Gx[M : 2 * M]) - site.limits[:, None]).max())
        r = 2 * M
    else:
        v.append((Gx[:M] - site.limits[:, None]).max())
        r = M
    if site.has_peak:
        v.append((Gx[r] - batch.peak[b][:T]).max())
    return max(v)


def reference_violation(prob, x):
    xv = x.reshape(-1)
    v = [0.0, (prob.A_ub @ xv - prob.b_ub).max()]
    if prob.A_eq.shape[0]:
        v.append(np.abs(prob.A_eq @ xv - prob.b_eq).max() * 0 + np.abs((prob.A_eq @ xv - prob.b_eq) / 0.017333333).max())
    for F, g in prob.soc:
        v.append(np.linalg.norm(F @ xv) - g)
    return max(v)


@pytest.mark.parametrize("ct", ["LINEAR", "SOC"])
@pytest.mark.parametrize("eq", [False, True])
@pytest.mark.parametrize("peak", [None, 300.0, "vec"])
def test_builder_matches_reference_statement(ct, eq, peak):
    infra, iface = caltech_interface()
    rng = np.random.default_rng(3)
    sl = sites.random_sessions(infra, 12, rng, min_rate_fraction=0.2)
    pk = np.linspace(200, 400, 12) if isinstance(peak, str) else peak
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 0.5e-2),
           ObjectiveComponent(total_energy, 0.3)]
    spec = [("quick_charge", 1, {}), ("equal_share", 0.5e-2, {}), ("total_energy", 0.3, {})]
    batch = build_batch([sl], infra, iface, obj, ct, eq, peak_limits=[pk])
    prob = build_reference_problem(sl, infra, iface, spec, ct, eq, pk)
    assert batch.Tm == prob.T and batch.N == prob.N
    assert np.array_equal(batch.lb[0].reshape(-1), prob.lb) and np.array_equal(batch.ub[0].reshape(-1), prob.ub)
    for _ in range(20):
        x = rng.uniform(-2, 34, size=(prob.N, prob.T)) * (prob.ub.reshape(prob.N, prob.T) > 0)
        assert abs(structured_objective(batch, 0, x) - prob.objective(x)) <= 1e-9 * (1 + abs(prob.objective(x)))
        sv, rv = structured_violation(batch, 0, x, infra), reference_violation(prob, x)
        # energy rows are scaled (A-periods vs kWh): compare feasibility verdicts and magnitudes loosely
        assert (sv <= 1e-9) == (rv <= 1e-9)
    # a feasible point for one is feasible for the other
    x = prob.lb.reshape(prob.N, prob.T).copy()
    assert (structured_violation(batch, 0, x, infra) <= 1e-9) == (reference_violation(prob, x) <= 1e-9)


def test_objective_library_matches_reference_formulas():
    infra, _ = caltech_interface()
    iface = Interface({"infrastructure_info": infra, "period": 5, "prices": np.linspace(0.1, 0.4, 12)})
    sl = sites.random_sessions(infra, 12, np.random.default_rng(5))
    ext = np.linspace(0, 20, 12)
    obj = [ObjectiveComponent(tou_energy_cost, 2.0), ObjectiveComponent(load_flattening, 0.01, {"external_signal": ext}),
           ObjectiveComponent(total_energy, 1.5)]
    spec = [("tou_energy_cost", 2.0, {}), ("load_flattening", 0.01, {"external_signal": ext}), ("total_energy", 1.5, {})]
    batch = build_batch([sl], infra, iface, obj, "LINEAR")
    prob = build_reference_problem(sl, infra, iface, spec, "LINEAR")
    v = infra.voltages / 1e3
    rng = np.random.default_rng(0)
    for _ in range(5):
        x = rng.uniform(0, 32, size=(54, 12))
        ours = (batch.q[0] * x).sum() + 0.5 * batch.pdiag[0] * (x ** 2).sum() + 0.5 * batch.lf[0] * ((v @ x) ** 2).sum()
        assert abs(ours - prob.objective(x)) < 1e-8 * abs(prob.objective(x))


def test_edge_rules():
    infra, iface = caltech_interface()
    # ub < lb is replaced by lb (aco.py:75)
    s = SessionInfo(infra.station_ids[0], "a", 5.0, 0.0, 0, 6, min_rates=8.0, max_rates=6.0)
    batch = build_batch([[s]], infra, iface, [ObjectiveComponent(quick_charge)], "LINEAR")
    assert np.all(batch.ub[0][0, :6] == 8.0) and np.all(batch.lb[0][0, :6] == 8.0)
    assert batch.Tm == 6 and np.all(batch.ub[0][1:] == 0)
    # two sessions on one EVSE use two slots (t_aco.py:194-208)
    a = SessionInfo(infra.station_ids[3], "a", 3.3, 0.0, 0, 12, max_rates=32)
    b = SessionInfo(infra.station_ids[3], "b", 3.3, 0.0, 12, 24, max_rates=32)
    batch = build_batch([[a, b]], infra, iface, [ObjectiveComponent(quick_charge)], "LINEAR")
    assert batch.K == 2 and batch.Tm == 24
    assert list(batch.s_off[0, :, 3]) == [0, 12] and list(batch.s_len[0, :, 3]) == [12, 12]
    # quick_charge weights use each problem's own horizon (aco.py:364-370) in a padded batch
    c = SessionInfo(infra.station_ids[1], "c", 3.3, 0.0, 0, 4, max_rates=32)
    batch = build_batch([[a], [c]], infra, iface, [ObjectiveComponent(quick_charge)], "LINEAR")
    assert batch.Tm == 12 and list(batch.T) == [12, 4]
    assert np.allclose(batch.q[1][0, :4], -np.array([1, 0.75, 0.5, 0.25])) and np.all(batch.q[1][:, 4:] == 0)
    # overlapping windows on one EVSE are rejected
    d = SessionInfo(infra.station_ids[3], "d", 3.3, 0.0, 6, 18, max_rates=32)
    with pytest.raises(ValueError):
        build_batch([[a, d]], infra, iface, [ObjectiveComponent(quick_charge)], "LINEAR")


def test_constraint_type_errors():
    infra, iface = caltech_interface()
    with pytest.raises(ValueError, match="SOC or AFFINE"):
        make_site(infra, "AFFINE")
    infra.phases = None
    with pytest.raises(ValueError, match="phases is required"):
        make_site(infra, "SOC")


def test_empty_constraint_matrix():
    infra = single_phase_single_constraint(2, 64)
    infra["constraint_matrix"] = np.zeros((0, 0))
    infra["constraint_limits"] = np.zeros(0)
    iface = TestingInterface({"active_sessions": [], "infrastructure_info": infra, "period": 5})
    site = make_site(iface.infrastructure_info(), "SOC")
    assert site.Mg == 0 and site.M == 0


def test_site_eigen_identity():
    infra, _ = caltech_interface()
    for ct in ("LINEAR", "SOC"):
        site = make_site(infra, ct, with_peak=True)
        G = site.G
        a, rho = 0.37, 0.8
        K = a * np.eye(site.N) + rho * G.T @ G
        D = rho / (a + rho * site.lam)
        Kinv = (np.eye(site.N) - site.Ghat.T @ (D[:, None] * site.Ghat)) / a
        assert np.abs(Kinv @ K - np.eye(site.N)).max() < 1e-10


def test_reference_constraint_builder_names():
    """The static builders keep the reference's names and dict keys (aco.py:45-218)."""
    from adacharge_amd import AdaptiveChargingOptimization, Rates, QuadObjective

    infra, iface = caltech_interface()
    sl = sites.random_sessions(infra, 12, np.random.default_rng(2), min_rate_fraction=0.3)
    rates = Rates((54, 12))
    b = AdaptiveChargingOptimization.charging_rate_bounds(rates, sl, infra.station_ids)
    batch = build_batch([sl], infra, iface, [ObjectiveComponent(quick_charge)], "SOC")
    assert np.array_equal(b["charging_rate_bounds.lb"], batch.lb[0]) and np.array_equal(b["charging_rate_bounds.ub"], batch.ub[0])
    e = AdaptiveChargingOptimization.energy_constraints(rates, sl, infra, 5, True)
    assert len(e) == len(sl) and all(k.startswith("energy_constraints.") for k in e)
    s0 = sl[0]
    row = e[f"energy_constraints.{s0.session_id}"]
    assert row["equality"] and abs(row["rhs"] / row["coefficient"] - batch.s_cap[0, 0, row["evse"]]) < 1e-9
    ic = AdaptiveChargingOptimization.infrastructure_constraints(rates, infra, "SOC")
    assert list(ic) == [f"infrastructure_constraints.{c}" for c in infra.constraint_ids]
    assert np.allclose(ic["infrastructure_constraints.Pri-A"]["rows"], np.stack([batch.site.G[5], batch.site.G[13]]))
    assert AdaptiveChargingOptimization.infrastructure_constraints(rates, infra, "LINEAR")["infrastructure_constraints.Pri-A"]["norm"] == "linear"
    with pytest.raises(ValueError, match="SOC or AFFINE"):
        AdaptiveChargingOptimization.infrastructure_constraints(rates, infra, "AFFINE")
    assert AdaptiveChargingOptimization.peak_constraint(rates, None) == {}
    assert AdaptiveChargingOptimization.peak_constraint(rates, 100.0)["peak_constraint"].shape == (12,)
    opt = AdaptiveChargingOptimization([ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 0.5)], iface)
    o = opt.build_objective(rates, infra, prev_peak=0)
    assert isinstance(o, QuadObjective) and o.sq == 0.5 and np.allclose(o.lin[0], [(12 - t) / 12 for t in range(12)])


def test_scenario_batch_equals_the_session_by_session_builder():
    """configs[3]: demand scenarios of one snapshot.  The vectorised path must produce exactly the arrays the
    general builder makes from the scaled SessionInfo lists."""
    from adacharge_amd import sites
    from adacharge_amd.builder import scenario_batch

    infra, iface = caltech_interface()
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-3)]
    base_sessions = sites.snapshot_batch(infra, 12, 1, seed=11)[0]
    scen = sites.demand_scenarios(base_sessions, 16, np.random.default_rng(5))
    slow = build_batch(scen, infra, iface, obj, "SOC")
    base = build_batch([base_sessions], infra, iface, obj, "SOC")
    with np.errstate(divide="ignore", invalid="ignore"):
        factor = np.where(base.s_len[0] > 0, slow.s_cap / base.s_cap[0], 1.0)
    fast = scenario_batch(base, factor)
    for name in ("T", "lb", "ub", "q", "pdiag", "lf", "s_off", "s_len", "s_eq"):
        assert np.array_equal(getattr(fast, name), getattr(slow, name)), name
    assert np.allclose(fast.s_cap, slow.s_cap, rtol=1e-13, atol=0)
    assert fast.B == 16 and fast.K == slow.K and fast.Tm == slow.Tm
    uniform = scenario_batch(base, np.array([0.5, 1.0, 2.0]))
    assert np.allclose(uniform.s_cap[2], 2.0 * base.s_cap[0]) and uniform.B == 3


def test_optimal_inaccurate_is_announced_like_cvxpy_does():
    """aco.py:319 accepts OPTIMAL_INACCURATE and so does the drop-in; cvxpy warns "Solution may be inaccurate" for that
    status and the drop-in's solve paths do the same, with the residuals reached."""
    import warnings

    import numpy as np
    import pytest

    from adacharge_amd.adaptive_charging_optimization import warn_inaccurate

    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert warn_inaccurate(np.array([1, 1, 3]), np.zeros(3), np.zeros(3)) == 0
    with pytest.warns(UserWarning, match="may be inaccurate: 2 of 4") as rec:
        assert warn_inaccurate(np.array([1, 5, 1, 5]), np.array([0, 2e-6, 0, 7e-6]), np.array([0, 1e-7, 0, 3e-6])) == 2
    assert "7.00e-06" in str(rec[0].message) and "3.00e-06" in str(rec[0].message)
