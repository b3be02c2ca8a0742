"""Session pre-processing on the way into the solve (reference adacharge.py:141-150): the three
acnportal functions the adapter calls, restated in adacharge_amd/acn.py, and the adapter's wiring of
``uninterrupted_charging`` / ``estimate_max_rate``.  CPU only."""
import numpy as np
import pytest

from adacharge_amd import AdaptiveSchedulingAlgorithm, ObjectiveComponent, quick_charge
from adacharge_amd.acn import (
    Interface,
    SessionInfo,
    apply_minimum_charging_rate,
    apply_upper_bound_estimate,
    enforce_pilot_limit,
)
from tests.acn_testing import single_phase_single_constraint


def _infra(n=3, limit=100.0, min_pilot=6, max_pilot=32):
    d = single_phase_single_constraint(n, limit, max_pilot=max_pilot, min_pilot=min_pilot)
    return Interface({"infrastructure_info": d, "period": 5}).infrastructure_info()


def _session(i, arrival=0, departure=6, energy=10.0, **kw):
    return SessionInfo(f"{i}", f"s{i}", energy, 0.0, arrival, departure, current_time=0, **kw)


def test_enforce_pilot_limit_caps_max_rates_and_copies():
    infra = _infra(max_pilot=16)
    s = [_session(0, max_rates=32.0), _session(1, max_rates=10.0)]
    out = enforce_pilot_limit(s, infra)
    assert np.all(out[0].max_rates == 16) and np.all(out[1].max_rates == 10)
    assert np.all(s[0].max_rates == 32)   # the input list is not modified


def test_apply_minimum_charging_rate_takes_period_as_third_argument():
    """ada.py:147-150 calls it as (sessions, infrastructure, interface.period): the period must not be
    mistaken for the `override` cap (min pilot 6 A became min(6, period))."""
    infra = _infra(min_pilot=6)
    out = apply_minimum_charging_rate([_session(0, max_rates=32.0)], infra, 5)
    assert out[0].min_rates[0] == 6
    out = apply_minimum_charging_rate([_session(0, max_rates=32.0)], infra, 1)
    assert out[0].min_rates[0] == 6
    out = apply_minimum_charging_rate([_session(0, max_rates=32.0)], infra, 5, override=4)
    assert out[0].min_rates[0] == 4


def test_apply_minimum_charging_rate_arrival_order_and_network_limit():
    # 3 EVSEs with min pilot 6 A under a 13 A limit: the two earliest arrivals get 6 A, the third is pinned to 0
    infra = _infra(n=3, limit=13.0, min_pilot=6)
    s = [_session(0, arrival=0, max_rates=32.0), _session(1, arrival=0, max_rates=32.0), _session(2, arrival=0, max_rates=32.0)]
    s[2].arrival = -3   # arrived first
    out = apply_minimum_charging_rate(s, infra, 5)
    by_id = {x.station_id: x for x in out}
    assert by_id["2"].min_rates[0] == 6 and by_id["0"].min_rates[0] == 6
    assert by_id["1"].min_rates[0] == 0 and by_id["1"].max_rates[0] == 0
    assert by_id["1"].max_rates[1] == 32   # only the first period is pinned


def test_apply_minimum_charging_rate_skips_sessions_that_need_less_than_the_minimum_pilot():
    infra = _infra(min_pilot=6)
    # 0.05 kWh at 208 V over 5-minute periods = 2.88 A-periods < 6 A
    out = apply_minimum_charging_rate([_session(0, energy=0.05, max_rates=32.0)], infra, 5)
    assert out[0].min_rates[0] == 0 and out[0].max_rates[0] == 0


def test_apply_minimum_charging_rate_reconciles_max_with_min():
    infra = _infra(min_pilot=8)
    out = apply_minimum_charging_rate([_session(0, max_rates=5.0)], infra, 5)
    assert out[0].min_rates[0] == 8 and out[0].max_rates[0] == 8


class _Estimator:
    def __init__(self, bounds):
        self.bounds = bounds
        self.interface = None

    def register_interface(self, interface):
        self.interface = interface

    def get_maximum_rates(self, sessions):
        return self.bounds


def test_apply_upper_bound_estimate_caps_and_reconciles():
    s = [_session(0, max_rates=32.0, min_rates=8.0), _session(1, max_rates=32.0)]
    out = apply_upper_bound_estimate(_Estimator({"s0": 6.0}), s)
    assert np.all(out[0].max_rates == 8)      # capped at 6, then raised to the session's min rate
    assert np.all(out[1].max_rates == 32)     # no estimate: unchanged


def test_adapter_wires_estimator_and_uninterrupted_charging():
    infra = _infra(n=2, min_pilot=6)
    iface = Interface({"infrastructure_info": infra, "period": 5})
    est = _Estimator({"s0": 12.0})
    alg = AdaptiveSchedulingAlgorithm([ObjectiveComponent(quick_charge)], estimate_max_rate=True,
                                      max_rate_estimator=est, uninterrupted_charging=True)
    alg.register_interface(iface)
    assert est.interface is iface   # ada.py:131-133
    pre = alg._preprocess([_session(0, max_rates=32.0), _session(1, max_rates=32.0)], infra)
    by_id = {x.session_id: x for x in pre}
    assert np.all(by_id["s0"].max_rates == 12) and np.all(by_id["s1"].max_rates == 32)
    assert by_id["s0"].min_rates[0] == 6 and by_id["s1"].min_rates[0] == 6


def test_copy_sessions_is_deepcopy_written_out():
    """The pre-processing steps copy the sessions before they cap rates (acnportal's preprocessing deep-copies); the
    written-out copy must give the same independent objects -- for SessionInfo itself and for a subclass with a mutable
    member of its own."""
    from copy import deepcopy

    import numpy as np

    from adacharge_amd import sites
    from adacharge_amd.acn import SessionInfo, copy_sessions, enforce_pilot_limit

    infra = sites.caltech54()
    sl = sites.snapshot_batch(infra, 12, 1, seed=7)[0]

    class Tagged(SessionInfo):
        pass

    t = deepcopy(sl[0]); t.__class__ = Tagged; t.tags = ["a", ["b"]]
    mixed = list(sl) + [t]
    got, want = copy_sessions(mixed), deepcopy(mixed)
    for g, w, src in zip(got, want, mixed):
        assert type(g) is type(src) and vars(g).keys() == vars(w).keys()
        for k, vw in vars(w).items():
            vg = getattr(g, k)
            assert np.array_equal(vg, vw) if isinstance(vw, np.ndarray) else vg == vw, k
            if isinstance(vw, (np.ndarray, list)):
                assert vg is not getattr(src, k), k
    assert got[-1].tags[1] is not t.tags[1]
    before = [s.max_rates.copy() for s in sl]
    capped = enforce_pilot_limit(sl, infra)
    capped[0].max_rates[:] = -1.0
    assert all(np.array_equal(s.max_rates, b) for s, b in zip(sl, before))   # the caller's sessions are untouched
