"""Oracle-certified fixtures for the PROX rows of the streaming kernels (tests/golden/prox.npz, generator
tools/make_golden_prox.py): load_flattening (aco.py:403-408) and demand_charge (aco.py:387-400) at
the shapes the large-site kernel (N > 64: 128 x 40, 192 x 48, BASELINE.json configs[4]'s own 512 x 48 with the bench
leg's generator) and the long-horizon kernel (54 x 96, 54 x 144) serve, site rows binding, energy equalities where the
bench leg has them.  Until round 4 those rows were compared with the C twin only.

load_flattening alone is quadratic in the per-period AGGREGATE power only (rank one per period): the per-EVSE split of its
optimum is not unique, like the LP of quick_charge.  The pure cases therefore pin what is unique -- the objective value,
the aggregate power per period, feasibility -- and the `_es` variants (equal_share * 1e-3 beside it) pin per-EVSE rates.

CPU: every fixture is feasible for the builder's statement, its objective matches, and the C twin reaches it.
GPU: the HIP path through the drop-in surface at default options."""
import numpy as np
import pytest

from adacharge_amd import AdaptiveChargingOptimization
from adacharge_amd.builder import build_batch
from tests import helpers as H

RATE_TOL = 1e-4 * 32.0
NAMES = [str(n) for n in H.load_prox()["names"]]
SLOW_ON_CPU = {"lf512_t48_eq", "lf512_t48_eq_es"}   # the twin needs minutes at 24,576 variables: GPU only


def _unique_rates(meta):
    return (not meta["lf"]) or meta["es"] > 0


def _check(rates, sl, infra, iface, obj, meta, exp, batch, res_obj):
    T = exp["rates"].shape[1]
    v = infra.voltages / 1e3
    agg, agg_ref = v @ rates[:, :T], v @ exp["rates"]
    assert np.abs(agg - agg_ref).max() <= 1e-4 * max(1.0, float(np.abs(agg_ref).max())), float(np.abs(agg - agg_ref).max())
    assert abs(res_obj - exp["obj"]) <= 1e-6 * abs(exp["obj"]), (res_obj, exp["obj"])
    if _unique_rates(meta):
        d = float(np.abs(rates[:, :T] - exp["rates"]).max())
        assert d <= RATE_TOL, d
    assert (rates[:, :T] >= batch.lb[0, :, :T] - 1e-6).all() and (rates[:, :T] <= np.maximum(batch.ub[0], batch.lb[0])[:, :T] + 1e-6).all()
    if meta["ct"] == "SOC":
        H.assert_infrastructure_satisfied(rates[:, :T], infra, tol=1e-4)
    else:
        assert (np.abs(infra.constraint_matrix) @ rates[:, :T] <= infra.constraint_limits[:, None] + 1e-4).all()
    for i in range(batch.N):
        for k in range(batch.K):
            L = int(batch.s_len[0, k, i])
            if L:
                o = int(batch.s_off[0, k, i])
                e = rates[i, o:o + L].sum()
                assert e <= batch.s_cap[0, k, i] + 1e-5 * max(1.0, batch.s_cap[0, k, i])
                if meta["eq"]:
                    assert abs(e - batch.s_cap[0, k, i]) <= 1e-5 * max(1.0, batch.s_cap[0, k, i])


def test_the_fixture_covers_both_prox_rows_on_both_streaming_kernels():
    g = H.load_prox()
    shapes = {n: (str(g[f"{n}_site"]), int(g[f"{n}_meta"][0]), bool(g[f"{n}_meta"][3])) for n in NAMES}
    wide = {k for k, (s, T, lf) in shapes.items() if s in ("wide128", "wide192", "synth512")}
    long_ = {k for k, (s, T, lf) in shapes.items() if s == "caltech54" and T > 32}
    for group in (wide, long_):
        assert any(shapes[k][2] for k in group) and any(not shapes[k][2] for k in group), group   # a flat row and a max row
    assert all(float(g[f"{n}_cert"].max()) < 1e-9 for n in NAMES)
    assert all(int(g[f"{n}_binding"][0]) >= 1 for n in NAMES)        # site rows bind in every case


@pytest.mark.parametrize("name", NAMES)
def test_fixture_is_consistent_with_the_builders_statement(name):
    g = H.load_prox()
    sl, infra, iface, obj, spec, meta, exp = H.prox_case(g, name)
    batch = build_batch([sl], infra, iface, obj, meta["ct"], meta["eq"])
    r = exp["rates"]
    T = r.shape[1]
    assert int(batch.T[0]) == T
    smooth = 0.5 * batch.pdiag[0] * (r ** 2).sum() + (batch.q[0, :, :T] * r).sum()
    full = smooth
    if batch.site.has_flat:
        vrow = batch.site.G[batch.site.flat_row]
        full = full + 0.5 * batch.lf[0] * ((vrow @ r) ** 2).sum()
    if batch.site.has_max:
        vrow = batch.site.G[batch.site.max_row]
        full = full + batch.dc[0] * max(float((vrow @ r).max()), float(batch.dfloor[0]))
    # (both sides leave out load_flattening's constant sum_t ext_t^2, aco.py:408: no solver sees it)
    assert abs(full - exp["obj"]) <= 1e-9 * abs(exp["obj"]), (full, exp["obj"])


@pytest.mark.parametrize("name", [n for n in NAMES if n not in SLOW_ON_CPU])
def test_c_twin_reaches_the_certified_optimum(name):
    from oracle import admm_port

    g = H.load_prox()
    sl, infra, iface, obj, spec, meta, exp = H.prox_case(g, name)
    batch = build_batch([sl], infra, iface, obj, meta["ct"], meta["eq"])
    out = admm_port.solve_batch(batch, eps_abs=1e-9, eps_rel=1e-9, max_iter=100000, accel_mem=5)
    assert out["status"][0] == 1, (name, out["status"], out["iters"])
    r = out["x"][0]
    T = exp["rates"].shape[1]
    v = infra.voltages / 1e3
    assert np.abs(v @ r[:, :T] - v @ exp["rates"]).max() <= 1e-4 * max(1.0, float(np.abs(v @ exp["rates"]).max()))
    if _unique_rates(meta):
        assert np.abs(r[:, :T] - exp["rates"]).max() <= RATE_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_path_reaches_the_certified_optimum(name):
    """The drop-in surface at DEFAULT options (large-site kernel: flat / max row with Anderson acceleration; long-horizon
    kernel: the same rows in its workspace layout) against the certificate."""
    g = H.load_prox()
    sl, infra, iface, obj, spec, meta, exp = H.prox_case(g, name)
    opt = AdaptiveChargingOptimization(obj, iface, constraint_type=meta["ct"], enforce_energy_equality=meta["eq"])
    rates = opt.solve(sl, infra)
    assert int(opt.last_result.status[0]) == 1
    _check(rates, sl, infra, iface, obj, meta, exp, opt.last_batch, float(opt.last_result.obj[0]))
