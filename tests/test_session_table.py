"""The batch pre-processing on a SessionTable (adacharge_amd/session_table.py) gives exactly what the per-session
functions of acn.py give (the reference's calls at adacharge.py:141-150), and the table-based builder gives the
batch the session-by-session statement gives.  CPU only."""
import numpy as np
import pytest

from adacharge_amd import ObjectiveComponent, equal_share, quick_charge, sites
from adacharge_amd import acn
from adacharge_amd import session_table as st
from adacharge_amd.acn import Interface
from adacharge_amd.builder import build_batch, build_batch_from_table


def _snapshots(infra, B, T, seed, two=False, mins=True):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(B):
        sl = sites.random_sessions_general(infra, T, rng, two, mins, demand_scale=float(rng.uniform(0.05, 1.5)))
        for s in sl:   # arrivals in the past as well, so that the arrival order is not the list order
            s.arrival = s.arrival - int(rng.integers(0, 4)) if s.arrival == 0 else s.arrival
        out.append(sl)
    return out


def _table_of(lists, infra):
    return st.SessionTable.from_sessions(lists, infra)


def _assert_same(table, lists, infra):
    """table (batch pre-processing) == SessionTable of the per-session result, session by session (matched by id)."""
    ref = _table_of(lists, infra)
    key_t = {(int(p), sid): k for k, (p, sid) in enumerate(zip(table.prob, table.session_ids))}
    assert len(key_t) == table.S
    for k, (p, sid) in enumerate(zip(ref.prob, ref.session_ids)):
        j = key_t[(int(p), sid)]
        a, b = slice(ref.seg[k], ref.seg[k + 1]), slice(table.seg[j], table.seg[j + 1])
        assert np.array_equal(ref.min_rates[a], table.min_rates[b]), (p, sid)
        assert np.array_equal(ref.max_rates[a], table.max_rates[b]), (p, sid)


def test_enforce_pilot_limit_matches_per_session():
    infra = sites.caltech54(max_pilot=20.0)
    lists = _snapshots(infra, 12, 12, 1)
    got = st.enforce_pilot_limit(_table_of(lists, infra), infra)
    _assert_same(got, [acn.enforce_pilot_limit(sl, infra) for sl in lists], infra)


class _Estimator:
    def get_maximum_rates(self, sessions):
        return {s.session_id: 5.0 + (hash(s.session_id) % 20) for k, s in enumerate(sessions) if k % 2 == 0}


def test_apply_upper_bound_estimate_matches_per_session():
    infra = sites.caltech54()
    lists = _snapshots(infra, 12, 12, 2)
    est = _Estimator()
    got = st.apply_upper_bound_estimate(_table_of(lists, infra), [est.get_maximum_rates(sl) for sl in lists])
    _assert_same(got, [acn.apply_upper_bound_estimate(est, sl) for sl in lists], infra)


@pytest.mark.parametrize("site_name,seed", [("caltech54", 3), ("jpl52", 4)])
def test_apply_minimum_charging_rate_matches_per_session(site_name, seed):
    infra = getattr(sites, site_name)()
    lists = _snapshots(infra, 16, 12, seed)
    got = st.apply_minimum_charging_rate(_table_of(lists, infra), infra, 5)
    want = [acn.apply_minimum_charging_rate(sl, infra, 5) for sl in lists]
    assert any((s.max_rates[0] == 0 and s.min_rates[0] == 0) for sl in want for s in sl)   # the network refusal path is exercised
    assert any(s.min_rates[0] == 8.0 for sl in want for s in sl)
    _assert_same(got, want, infra)
    # same ORDER as the per-session function (sorted by arrival, stable; remaining_time <= 0 dropped): the order breaks
    # ties in diff_based_reallocation
    ref = _table_of(want, infra)
    assert got.session_ids == ref.session_ids and np.array_equal(got.prob, ref.prob) and np.array_equal(got.seg, ref.seg)
    got = st.apply_minimum_charging_rate(_table_of(lists, infra), infra, 5, override=4.0)
    _assert_same(got, [acn.apply_minimum_charging_rate(sl, infra, 5, override=4.0) for sl in lists], infra)


def test_apply_minimum_charging_rate_drops_finished_sessions_and_sorts_by_arrival():
    infra = sites.caltech54()
    ids = infra.station_ids
    mk = lambda i, sid, arr, dep: acn.SessionInfo(ids[i], sid, 5.0, 0.0, arr, dep, current_time=3, max_rates=32.0)
    lists = [[mk(0, "late", 3, 9), mk(1, "gone", 0, 3), mk(2, "early", 1, 8), mk(3, "tie", 3, 7)]]
    got = st.apply_minimum_charging_rate(_table_of(lists, infra), infra, 5)
    want = acn.apply_minimum_charging_rate(lists[0], infra, 5)
    assert [s.session_id for s in want] == ["early", "late", "tie"]
    assert got.session_ids == ["early", "late", "tie"] and got.S == 3
    _assert_same(got, [want], infra)


@pytest.mark.parametrize("two", [False, True])
def test_table_builder_equals_session_builder(two):
    infra = sites.caltech54()
    iface = Interface({"infrastructure_info": infra, "period": 5})
    lists = _snapshots(infra, 10, 16, 7, two=two)
    obj = [ObjectiveComponent(quick_charge), ObjectiveComponent(equal_share, 1e-3)]
    peaks = [None if k % 3 else 300.0 + k for k in range(len(lists))]
    a = build_batch(lists, infra, iface, obj, "SOC", peak_limits=peaks)
    b = build_batch_from_table(_table_of(lists, infra), infra, iface, obj, "SOC", peak_limits=peaks)
    for name in ("T", "lb", "ub", "q", "pdiag", "s_off", "s_len", "s_cap", "s_eq", "peak"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    assert a.K == b.K == (2 if two else 1)


def test_overlapping_sessions_on_one_evse_are_refused():
    infra = sites.caltech54()
    iface = Interface({"infrastructure_info": infra, "period": 5})
    s = [acn.SessionInfo(infra.station_ids[0], "a", 5.0, 0.0, 0, 8, current_time=0, max_rates=32.0),
         acn.SessionInfo(infra.station_ids[0], "b", 5.0, 0.0, 6, 12, current_time=0, max_rates=32.0)]
    with pytest.raises(ValueError, match="overlap in time"):
        build_batch([s], infra, iface, [ObjectiveComponent(quick_charge)], "SOC")
