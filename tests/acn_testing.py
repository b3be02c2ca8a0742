"""Stand-ins for the acnportal test helpers the reference's tests import
(t_aco.py:4-5, t_post.py:7-12): ``session_generator``,
``single_phase_single_constraint``, ``three_phase_balanced_network`` and
``TestingInterface`` -- restated from SURVEY.md Appendix B [recalled; acnportal
is not installed].  They return plain dicts exactly like the originals so the
scenario tables below read like the reference's own tests."""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np

from adacharge_amd.acn import Interface


def session_generator(
    num_sessions,
    arrivals,
    departures,
    requested_energy,
    remaining_energy,
    max_rates,
    min_rates=None,
    station_ids=None,
    estimated_departures=None,
) -> List[Dict]:
    sessions = []
    for i in range(num_sessions):
        station_id = station_ids[i] if station_ids is not None else f"{i}"
        s = {
            "station_id": station_id,
            "session_id": f"{i}",
            "requested_energy": requested_energy[i],
            "energy_delivered": requested_energy[i] - remaining_energy[i],
            "arrival": arrivals[i],
            "departure": departures[i],
            "estimated_departure": (
                estimated_departures[i] if estimated_departures is not None else departures[i]
            ),
            "min_rates": min_rates[i] if min_rates is not None else 0,
            "max_rates": max_rates[i],
        }
        sessions.append(s)
    return sessions


def single_phase_single_constraint(
    num_evses, limit, max_pilot=32, min_pilot=8, allowable_pilots=None, is_continuous=None
) -> Dict:
    if allowable_pilots is None:
        allowable_pilots = [np.array([0.0] + list(range(min_pilot, max_pilot + 1)), float)] * num_evses
    if is_continuous is None:
        is_continuous = np.ones(num_evses, dtype=bool)
    return {
        "constraint_matrix": np.ones((1, num_evses)),
        "constraint_limits": np.array([limit], float),
        "phases": np.zeros(num_evses),
        "voltages": np.full(num_evses, 208.0),
        "constraint_ids": ["all"],
        "station_ids": [f"{i}" for i in range(num_evses)],
        "max_pilot": np.full(num_evses, float(max_pilot)),
        "min_pilot": np.full(num_evses, float(min_pilot)),
        "allowable_pilots": allowable_pilots,
        "is_continuous": is_continuous,
    }


def three_phase_balanced_network(
    evses_per_phase, limit, max_pilot=32, min_pilot=8, allowable_pilots=None, is_continuous=None
) -> Dict:
    n = 3 * evses_per_phase
    if allowable_pilots is None:
        allowable_pilots = [np.array([0.0] + list(range(min_pilot, max_pilot + 1)), float)] * n
    if is_continuous is None:
        is_continuous = np.ones(n, dtype=bool)
    k = evses_per_phase
    cm = np.array(
        [
            [1] * k + [-1] * k + [0] * k,
            [0] * k + [1] * k + [-1] * k,
            [-1] * k + [0] * k + [1] * k,
        ],
        float,
    )
    return {
        "constraint_matrix": cm,
        "constraint_limits": np.full(3, float(limit)),
        "phases": np.array([0.0] * k + [-120.0] * k + [120.0] * k),
        "voltages": np.full(n, 208.0),
        "constraint_ids": ["AB", "BC", "CA"],
        "station_ids": [f"{i}" for i in range(n)],
        "max_pilot": np.full(n, float(max_pilot)),
        "min_pilot": np.full(n, float(min_pilot)),
        "allowable_pilots": allowable_pilots,
        "is_continuous": is_continuous,
    }


class TestingInterface(Interface):
    """Dict-backed interface (acnportal.algorithms.tests.testing_interface)."""

    __test__ = False
