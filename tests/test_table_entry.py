"""acnqp_solve_table (ABI v8): the statement -> arrays step below the ABI.  The caller hands over the
SESSION TABLE -- what charging_rate_bounds / energy_constraints loop over (aco.py:45-124) -- and one linear cost per
distinct horizon (aco.py:200-218, 243-245); lb, ub, q and the session slots are formed on the device.

CPU: ``TablePlan.expand()`` (the numpy twin of the device's expand kernel) gives the batch the dense builder always gave.
GPU: the table entry returns the same bits as the dense entry on the golden fixtures and on the bench workload; argument
checks; chunked calls."""
import numpy as np
import pytest

from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites, tou_energy_cost, total_energy
from adacharge_amd.acn import Interface
from adacharge_amd.builder import build_batch, plan_from_table
from adacharge_amd.session_table import SessionTable
from tests import helpers as H

QC = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-3)]


def _plan_and_batch(snaps, infra, iface, obj, ct="SOC", eq=False, peaks=None):
    table = SessionTable.from_sessions(snaps, infra)
    plan = plan_from_table(table, infra, iface, obj, ct, eq, peaks)
    return plan, plan.expand()


def test_plan_expands_to_the_dense_batch_of_the_per_session_builder():
    """The dense arrays of ``plan.expand()`` against a direct restatement of aco.py:61-123 with Python loops."""
    infra = sites.caltech54()
    iface = Interface({"infrastructure_info": infra, "period": 5})
    rng = np.random.default_rng(3)
    snaps = [sites.random_sessions_general(infra, 16, rng, two_per_evse=True, min_rates=True) for _ in range(6)]
    plan, batch = _plan_and_batch(snaps, infra, iface, QC)
    assert plan.K == batch.K >= 2 and plan.S == sum(len(s) for s in snaps)
    k = 208 * 5 / 1e3 / 60
    for b, sl in enumerate(snaps):
        lb, ub = np.zeros((54, batch.Tm)), np.zeros((54, batch.Tm))
        for s in sl:
            i = infra.get_station_index(s.station_id)
            lb[i, s.arrival_offset:s.arrival_offset + s.remaining_time] = s.min_rates
            ub[i, s.arrival_offset:s.arrival_offset + s.remaining_time] = s.max_rates
        ub[ub < lb] = lb[ub < lb]
        assert np.array_equal(lb, batch.lb[b]) and np.array_equal(ub, batch.ub[b])
        caps = sorted((infra.get_station_index(s.station_id), s.arrival_offset, s.remaining_time, s.remaining_demand / k) for s in sl)
        got = sorted((i, int(batch.s_off[b, kk, i]), int(batch.s_len[b, kk, i]), float(batch.s_cap[b, kk, i]))
                     for kk in range(batch.K) for i in range(54) if batch.s_len[b, kk, i] > 0)
        assert len(caps) == len(got) and all(a[:3] == g[:3] and abs(a[3] - g[3]) <= 1e-12 * abs(a[3]) for a, g in zip(caps, got))
        T = int(batch.T[b])
        assert np.array_equal(batch.q[b, :, :T], plan.q_table[plan.q_index[b], :, :T]) and not batch.q[b, :, T:].any()


def test_plan_groups_sessions_by_snapshot_whatever_the_tables_order():
    infra = sites.caltech54()
    iface = Interface({"infrastructure_info": infra, "period": 5})
    table = sites.snapshot_table(infra, 12, 32, seed=5)
    ref = plan_from_table(table, infra, iface, QC).expand()
    perm = np.random.default_rng(0).permutation(table.S)
    mixed = plan_from_table(table.take(perm), infra, iface, QC)
    assert np.all(np.diff(np.repeat(np.arange(mixed.B), np.diff(mixed.sess_seg))) >= 0)
    out = mixed.expand()
    for name in ("lb", "ub", "q", "s_off", "s_len", "s_cap", "T", "pdiag"):
        assert np.array_equal(getattr(ref, name), getattr(out, name)), name


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["golden12", "two_sessions_T16_linear_peak", "tou_T24", "jpl52_T24"])
def test_table_entry_returns_the_bits_of_the_dense_entry(case):
    from adacharge_amd.backend import SiteHandle, default_options

    peaks = None
    if case == "golden12":   # the SOC cases of tests/golden/caltech54_T12.npz that share the default equal_share weight
        g = H.load_golden()
        keys = sorted({k.rsplit("_", 1)[0] for k in g.files if k.endswith("_rates")})
        cases = [H.golden_case(g, k) for k in keys]
        cases = [c for c in cases if c[1]["ct"] == "SOC" and not c[1]["eq"] and c[1]["es"] == 1e-3]
        assert len(cases) >= 6
        infra, iface = H.caltech_interface()
        snaps, obj, ct = [c[0] for c in cases], QC, "SOC"
    elif case == "two_sessions_T16_linear_peak":
        infra = sites.caltech54()
        iface = Interface({"infrastructure_info": infra, "period": 5})
        rng = np.random.default_rng(11)
        snaps = [sites.random_sessions_general(infra, 16, rng, two_per_evse=True, min_rates=True, demand_scale=0.7) for _ in range(40)]
        obj, ct = QC, "LINEAR"
        peaks = [float(rng.uniform(300, 600)) if b % 2 else None for b in range(40)]
    elif case == "tou_T24":
        infra = sites.caltech54()
        iface = Interface({"infrastructure_info": infra, "period": 5, "prices": np.random.default_rng(2).uniform(0.05, 0.4, size=64)})
        snaps = sites.snapshot_batch(infra, 24, 96, seed=24)
        obj = [ObjectiveComponent(tou_energy_cost, 3.0), ObjectiveComponent(total_energy, 1.5), ObjectiveComponent(equal_share, 1e-3)]
        ct = "SOC"
    else:
        infra = sites.jpl52()
        iface = Interface({"infrastructure_info": infra, "period": 5})
        snaps, obj, ct = sites.snapshot_batch(infra, 24, 64, seed=52), QC, "SOC"
    plan, batch = _plan_and_batch(snaps, infra, iface, obj, ct, False, peaks)
    h = SiteHandle(batch.site, 0)
    dense = h.solve(batch, default_options(), want_y=True)
    table = h.solve_table(plan, default_options(), want_y=True)
    h.close()
    assert (dense.status == 1).all()
    for name in ("x", "status", "iters", "pri_res", "dua_res", "obj", "y"):
        assert np.array_equal(getattr(dense, name), getattr(table, name)), name


@pytest.mark.gpu
def test_table_entry_on_the_bench_workload_in_chunks_and_through_the_surface():
    """4,096 snapshots (several pipeline chunks, the ramp included) from a SessionTable: same bits as the dense entry;
    and AdaptiveSchedulingAlgorithm.schedule_batch -- which takes the table entry -- equals the per-snapshot schedule()."""
    from adacharge_amd import AdaptiveSchedulingAlgorithm
    from adacharge_amd.backend import SiteHandle, default_options

    infra = sites.caltech54()
    iface = Interface({"infrastructure_info": infra, "period": 5})
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-12)]
    table = sites.snapshot_table(infra, 12, 4096, seed=99)
    plan = plan_from_table(table, infra, iface, obj)
    batch = plan.expand()
    h = SiteHandle(batch.site, 0)
    a, b = h.solve(batch, default_options()), h.solve_table(plan, default_options())
    h.close()
    assert (a.status == 1).all() and np.array_equal(a.x, b.x) and np.array_equal(a.iters, b.iters)
    alg = AdaptiveSchedulingAlgorithm(obj, solver_options={})
    alg.register_interface(iface)
    small = sites.snapshot_batch(infra, 12, 6, seed=7)
    outs = alg.schedule_batch(small)
    for sl, out in zip(small, outs):
        one = alg.schedule(sl)
        assert out.keys() == one.keys()
        assert max(float(np.abs(out[k] - one[k]).max()) for k in out) <= 1e-4 * 32   # (schedule() asks for tighter residuals)


@pytest.mark.gpu
def test_table_entry_refuses_inconsistent_tables():
    from adacharge_amd.backend import SiteHandle, default_options

    infra = sites.caltech54()
    iface = Interface({"infrastructure_info": infra, "period": 5})
    plan = plan_from_table(sites.snapshot_table(infra, 12, 8, seed=1), infra, iface, QC)
    h = SiteHandle(plan.site, 0)
    import copy

    for field, value, msg in (("s_evse", 54, "EVSE or slot"), ("s_off", 12, "window outside"), ("q_index", 99, "q_index"),
                              ("s_len", 3, "one entry per remaining period")):
        bad = copy.copy(plan)
        arr = getattr(plan, field).copy()
        arr[0] = value if field != "s_len" else arr[0] + value
        setattr(bad, field, arr)
        if field == "s_len":   # keep the window inside the horizon so that the rate-count check is the one that fires
            bad.s_off = plan.s_off.copy(); bad.s_off[0] = 0; arr[0] = min(arr[0], 12)
        with pytest.raises(ValueError, match=msg):
            h.solve_table(bad, default_options())
    h.close()
