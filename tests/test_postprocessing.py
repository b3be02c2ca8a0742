"""Exact known answers for the post-processing row: the
scenario table of the reference's t_post.py:17-318, parametrised."""
from types import SimpleNamespace

import numpy as np
import pytest
from numpy import testing as nptest

from adacharge_amd import (
    ceil_to_set, diff_based_reallocation, floor_to_set, increment_in_set, index_based_reallocation,
    project_into_continuous_feasible_pilots, project_into_discrete_feasible_pilots,
)
from adacharge_amd.acn import earliest_deadline_first
from tests.acn_testing import (
    TestingInterface, session_generator, single_phase_single_constraint, three_phase_balanced_network,
)

SET = np.array([0, 5, 10])


@pytest.mark.parametrize("x,eps,want", [(5, 0.05, 5), (5, 0, 5), (4.9, 0.05, 0), (4.98, 0.05, 5), (-1, 0.05, 0), (15, 0.05, 10)])
def test_floor_to_set(x, eps, want):   # t_post.py:17-46
    assert floor_to_set(x, SET, eps=eps) == want


@pytest.mark.parametrize("x,eps,want", [(5, 0.05, 5), (5, 0, 5), (2.5, 0.05, 5), (5.02, 0.05, 5), (-1, 0.05, 0), (15, 0.05, 10)])
def test_ceil_to_set(x, eps, want):   # t_post.py:49-78
    assert ceil_to_set(x, SET, eps=eps) == want


@pytest.mark.parametrize("x,want", [(5, 10), (2.5, 5), (-1, 0), (15, 10)])
def test_increment_in_set(x, want):   # t_post.py:81-100
    assert increment_in_set(x, SET) == want


def mock_infra(allowable=None):
    return SimpleNamespace(
        max_pilot=np.full(5, 32), min_pilot=np.full(5, 0), num_stations=5,
        allowable_pilots=[allowable] * 5 if allowable is not None else None,
    )


@pytest.mark.parametrize("fill,want", [(16, 16), (33, 32), (-1, 0)])
def test_project_continuous(fill, want):   # t_post.py:103-123
    out = project_into_continuous_feasible_pilots(np.full((5, 20), fill), mock_infra())
    nptest.assert_equal(out, want)


@pytest.mark.parametrize("fill,want", [(16, 16), (18, 16), (15.98, 16), (33, 32), (-1, 0)])
def test_project_discrete(fill, want):   # t_post.py:126-157
    out = project_into_discrete_feasible_pilots(np.full((5, 20), fill, dtype=float), mock_infra([0, 8, 16, 24, 32]))
    nptest.assert_equal(out, want)


def _sessions(remaining=(3.3, 3.3, 3.3)):
    return session_generator(3, [0] * 3, [2, 3, 4], [3.3] * 3, list(remaining), [32] * 3, [0] * 3)


FINE = [np.array([0] + list(range(8, 33))) for _ in range(3)]
COARSE = [np.array([0, 8, 16, 24, 32]) for _ in range(3)]

REALLOC_CASES = [   # (infrastructure dict, sessions, peak, rows bumped to 17)   t_post.py:173-318
    (single_phase_single_constraint(3, 66, allowable_pilots=COARSE), _sessions(), 48, []),
    (single_phase_single_constraint(3, 66, allowable_pilots=FINE), _sessions(), 50, [0, 1]),
    (single_phase_single_constraint(3, 49, allowable_pilots=FINE), _sessions(), 60, [0]),
    (three_phase_balanced_network(1, 16.51 * np.sqrt(3), allowable_pilots=FINE), _sessions(), 60, [0]),
    (single_phase_single_constraint(3, 66, allowable_pilots=FINE), _sessions((0.277, 3.3, 3.3)), 50, [1, 2]),
]


@pytest.mark.parametrize("infra,sessions,peak,bumped", REALLOC_CASES)
def test_index_based_reallocation(infra, sessions, peak, bumped):
    iface = TestingInterface({"active_sessions": sessions, "infrastructure_info": infra, "current_time": 0, "period": 5})
    rates = np.full((3, 10), 16)   # integer array, mutated in place
    out = index_based_reallocation(
        rates, iface.active_sessions(), iface.infrastructure_info(), peak, earliest_deadline_first, iface
    )
    expected = np.full((3, 10), 16)
    expected[bumped, 0] = 17
    nptest.assert_equal(out, expected)
    assert out is rates


def test_diff_based_reallocation_returns_first_period_loss():
    infra = single_phase_single_constraint(3, 66, allowable_pilots=FINE)
    iface = TestingInterface({"active_sessions": _sessions(), "infrastructure_info": infra, "current_time": 0, "period": 5})
    rates = np.full((3, 10), 16.0)
    rates[:, 0] = [16.9, 16.5, 16.6]   # floors to 16 each; 2 A of rounding loss, biggest loser first
    out = diff_based_reallocation(rates, iface.active_sessions(), iface.infrastructure_info(), iface)
    nptest.assert_equal(out[:, 0], [17, 16, 17])
    nptest.assert_equal(out[:, 1:], 16)
    assert out[:, 0].sum() <= rates[:, 0].sum()


# ---- whole-batch versions (adacharge_amd/postprocessing.py, *_batch) == the per-snapshot functions ----------------
@pytest.mark.parametrize("site_name,seed", [("caltech54", 5), ("jpl52", 6)])
def test_batch_postprocessing_equals_per_snapshot(site_name, seed):
    from adacharge_amd import postprocessing as pp, sites
    from adacharge_amd.acn import Interface
    from adacharge_amd.session_table import SessionTable

    infra = getattr(sites, site_name)()
    iface = Interface({"infrastructure_info": infra, "period": 5})
    rng = np.random.default_rng(seed)
    B, T = 24, 12
    lists = [sites.random_sessions_general(infra, T, rng, two_per_evse=(b % 2 == 1), min_rates=False, demand_scale=1.0) for b in range(B)]
    table = SessionTable.from_sessions(lists, infra)
    rates = np.zeros((B, infra.num_stations, T))
    for b, sl in enumerate(lists):
        for s in sl:
            i = infra.station_ids.index(s.station_id)
            rates[b, i, s.arrival_offset : s.arrival_offset + s.remaining_time] = rng.uniform(0, 14, size=s.remaining_time)
    rates[0, :, 0] = 0.03   # within eps of a pilot value: the floor rounds UP (post.py:10-31)
    d = pp.project_into_discrete_feasible_pilots_batch(rates, infra)
    c = pp.project_into_continuous_feasible_pilots_batch(rates - 1.0, infra)
    got = pp.diff_based_reallocation_batch(rates, table, infra, iface)
    changed = 0
    for b in range(B):
        assert np.array_equal(d[b], pp.project_into_discrete_feasible_pilots(rates[b], infra))
        assert np.array_equal(c[b], pp.project_into_continuous_feasible_pilots(rates[b] - 1.0, infra))
        want = pp.diff_based_reallocation(rates[b].copy(), lists[b], infra, iface)
        assert np.array_equal(got[b], want), b
        changed += int((want[:, 0] != d[b][:, 0]).sum())
    assert changed > 0   # the reallocation loop did hand rounding loss back somewhere
