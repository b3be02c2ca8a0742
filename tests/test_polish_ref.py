"""oracle/polish_ref.py -- the numpy specification of the device-side polish (adacharge_amd/csrc/acn_qp_polish.hpp) --
against the IPM certificates of tests/golden/stalled.npz, on the CPU: started from the C twin's iterate after 800
iterations of its single adaptive pass (what the solver kernel hands over), the active-set Newton method reaches every
certified optimum of horizon 12 and verifies the KKT conditions; the Schur-complement step it is built on equals the
plain KKT solve.  The GPU twin of this test is tests/test_golden_stalled.py::test_polish_agrees_with_its_numpy_specification."""
import numpy as np
import pytest

from adacharge_amd import ObjectiveComponent, equal_share, quick_charge
from adacharge_amd.builder import build_batch
from oracle import polish_ref
from tests import helpers as H

NAMES = [str(n) for n in H.load_stalled()["names"]]


@pytest.mark.parametrize("name", [n for n in NAMES if "T24" not in n])
def test_spec_reaches_the_certificate_from_the_twins_iterate(name):
    from oracle import admm_port

    g = H.load_stalled()
    sl, infra, iface, meta, peak, exp = H.wide_case(g, name)
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, meta["es"])]
    batch = build_batch([sl], infra, iface, obj, meta["ct"], meta["eq"])
    one = admm_port.solve_batch(batch, accel_mem=5, retry_passes=0, max_iter=800)
    x, info = polish_ref.polish_batch_problem(batch, 0, one["x"][0], one["y"][0])
    assert info["ok"] and info["why"] == "kkt", info
    assert info["rounds"] <= 40 and info["rows"] <= polish_ref.MAX_ROWS
    assert np.abs(x - exp["rates"]).max() <= 1e-7, float(np.abs(x - exp["rates"]).max())
    H.assert_infrastructure_satisfied(x, infra, tol=1e-7)


def test_schur_step_equals_the_kkt_solve():
    """One Newton round by block elimination (session projector, then the SPD system of the site rows) against the plain
    solve of [pd I E' A' W'; E 0 0 0; A 0 0 0; W 0 0 -D] on random data with disjoint session supports."""
    rng = np.random.default_rng(0)
    n, nE, nA, nW, pd = 12, 3, 2, 2, 0.3
    g = rng.normal(size=n)
    E = np.zeros((nE, n)); E[0, :4] = 1; E[1, 4:7] = 1; E[2, 8:12] = 1
    A, W, D = rng.normal(size=(nA, n)), rng.normal(size=(nW, n)), np.array([0.7, 1.3])
    cE, cA = rng.normal(size=nE), rng.normal(size=nA)
    K = np.block([[pd * np.eye(n), E.T, A.T, W.T], [E, np.zeros((nE, nE + nA + nW))], [A, np.zeros((nA, nE + nA + nW))],
                  [W, np.zeros((nW, nE + nA)), -np.diag(D)]])
    ref = np.linalg.solve(K, np.r_[-g, cE, cA, np.zeros(nW)])
    nfree = E.sum(1)
    P = lambda v: v - E.T @ ((E @ v) / nfree)
    e = E.T @ (cE / nfree)
    R = np.vstack([A, W])
    S = R @ np.array([P(r) for r in R]).T + pd * np.diag(np.r_[np.zeros(nA), D])
    lam = polish_ref.cholesky_solve(S, R @ (-P(g) + pd * e) - pd * np.r_[cA, np.zeros(nW)], 0.0)
    dx = -P(g + R.T @ lam) / pd + e
    assert np.abs(dx - ref[:n]).max() <= 1e-12 and np.abs(lam - ref[n + nE:]).max() <= 1e-12


def test_spec_gives_up_cleanly():
    """More tight rows than the Schur system holds -> info['why'] == 'rows', x untouched; an inconsistent working set is
    reported, never returned as an answer."""
    g = H.load_stalled()
    name = next(n for n in NAMES if "T24" in n)
    sl, infra, iface, meta, peak, exp = H.wide_case(g, name)
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, meta["es"])]
    batch = build_batch([sl], infra, iface, obj, meta["ct"], meta["eq"])
    from oracle import admm_port

    one = admm_port.solve_batch(batch, accel_mem=5, retry_passes=0, max_iter=800)
    old = polish_ref.MAX_ROWS
    try:
        polish_ref.MAX_ROWS = 16
        x, info = polish_ref.polish_batch_problem(batch, 0, one["x"][0], one["y"][0])
    finally:
        polish_ref.MAX_ROWS = old
    assert not info["ok"] and info["why"] == "rows"
    assert np.array_equal(x, np.clip(one["x"][0][:, :x.shape[1]], batch.lb[0][:, :x.shape[1]], np.maximum(batch.ub[0], batch.lb[0])[:, :x.shape[1]]))
