"""Host-side mirror of the reference's optimisation surface
(/root/reference/adacharge/adaptive_charging_optimization.py).

Same names, arguments and error behaviour as the reference:
``AdaptiveChargingOptimization(objective, interface, constraint_type="SOC",
enforce_energy_equality=False, solver="ECOS")`` and ``.solve(active_sessions,
infrastructure, peak_limit=None, prev_peak=0, verbose=False) -> (N, T)`` array
(aco.py:31-43, 286-321); ``ObjectiveComponent`` (aco.py:12-15);
``InfeasibilityException`` (aco.py:8-9); the objective library (aco.py:336-408).

What changes is what sits underneath: the cvxpy ``Problem.solve`` call
(aco.py:315-318) is replaced by the structured builder (builder.py) and the
HIP batched ADMM solver behind the C-ABI (include/acn_qp.h).  There is no CPU
fallback: if the HIP library is missing, ``solve`` raises.

Deviations that cannot be avoided without cvxpy:
objective functions return ``QuadObjective`` descriptors, not cvxpy
expressions, so user-defined objectives must be built from the helpers here;
``build_problem`` returns the structured ``ProblemBatch`` instead of cvxpy
objects.
"""
from __future__ import annotations

from collections import namedtuple
from typing import List, Optional, Sequence, Union

import numpy as np


class InfeasibilityException(Exception):
    pass


ObjectiveComponent = namedtuple("ObjectiveComponent", ["function", "coefficient", "kwargs"])
ObjectiveComponent.__new__.__defaults__ = (1, {})


# ---------------------------------------------------------------------------
# Descriptor algebra standing in for cvxpy expressions
# ---------------------------------------------------------------------------
class Rates:
    """Symbolic handle for the (N, T) rates variable (cp.Variable at aco.py:247)."""

    def __init__(self, shape):
        self.shape = tuple(shape)


class QuadObjective:
    """A concave objective term in the reference's *maximisation* convention:

        value(r) = <lin, r>  -  sq * sum(r^2)  -  flat * sum_t (v' r[:, t])^2  +  const

    with v = voltages / 1e3.  Closed under ``+`` and scalar ``*`` exactly like
    the cvxpy expressions it replaces (aco.py:210-218)."""

    __array_priority__ = 1000

    def __init__(self, lin, sq=0.0, flat=0.0, const=0.0, epigraph=None):
        self.lin = np.asarray(lin, float)
        self.sq = float(sq)
        self.flat = float(flat)
        self.const = float(const)
        self.epigraph = list(epigraph) if epigraph else []

    @classmethod
    def zero(cls, shape):
        return cls(np.zeros(shape))

    def __add__(self, other):
        if np.isscalar(other):
            return QuadObjective(self.lin, self.sq, self.flat, self.const + other, self.epigraph)
        return QuadObjective(
            self.lin + other.lin, self.sq + other.sq, self.flat + other.flat,
            self.const + other.const, self.epigraph + other.epigraph,
        )

    __radd__ = __add__

    def __mul__(self, c):
        c = float(c)
        return QuadObjective(
            c * self.lin, c * self.sq, c * self.flat, c * self.const,
            [(c * w, spec) for w, spec in self.epigraph],
        )

    __rmul__ = __mul__

    def __neg__(self):
        return self * -1.0

    def __sub__(self, other):
        return self + (-other)


# ---------------------------------------------------------------------------
#  Objective functions -- same signatures as aco.py:336-408.  All take rates as
#  first positional argument and swallow unknown keyword arguments.
# ---------------------------------------------------------------------------
def _volt_kw(infrastructure):
    return np.asarray(infrastructure.voltages, float) / 1e3


def quick_charge(rates, infrastructure, interface, **kwargs):
    """aco.py:363-371: sum_t ((T - t) / T) * sum_i r[i, t]."""
    T = rates.shape[1]
    c = np.array([(T - t) / T for t in range(T)])
    return QuadObjective(np.broadcast_to(c, rates.shape).copy())


def equal_share(rates, infrastructure, interface, **kwargs):
    """aco.py:374-375: -sum_squares(rates)."""
    return QuadObjective(np.zeros(rates.shape), sq=1.0)


def tou_energy_cost(rates, infrastructure, interface, **kwargs):
    """aco.py:378-380: -prices @ aggregate_period_energy."""
    prices = np.asarray(interface.get_prices(rates.shape[1]), float)
    v = _volt_kw(infrastructure)
    return QuadObjective(-(v[:, None] * (interface.period / 60)) * prices[None, :])


def total_energy(rates, infrastructure, interface, **kwargs):
    """aco.py:383-384: sum of per-period energy in kWh."""
    v = _volt_kw(infrastructure)
    return QuadObjective(np.repeat((v * (interface.period / 60))[:, None], rates.shape[1], axis=1))


def load_flattening(rates, infrastructure, interface, external_signal=None, **kwargs):
    """aco.py:403-408: -sum_t (v' r_t + ext_t)^2."""
    T = rates.shape[1]
    ext = np.zeros(T) if external_signal is None else np.asarray(external_signal, float)
    v = _volt_kw(infrastructure)
    return QuadObjective(-2.0 * v[:, None] * ext[None, :], flat=1.0, const=-float(ext @ ext))


def peak(rates, infrastructure, interface, baseline_peak=0, **kwargs):
    """aco.py:387-394: max(max_t aggregate_power_t, prev_peak kW[, baseline_peak]).
    Convex; only usable with a negative coefficient (as ``demand_charge`` does).
    Carried as an epigraph request for the builder."""
    prev_peak = interface.get_prev_peak() * infrastructure.voltages[0] / 1000
    floor = max(prev_peak, baseline_peak) if baseline_peak > 0 else prev_peak
    return QuadObjective(np.zeros(rates.shape), epigraph=[(1.0, {"floor": float(floor)})])


def demand_charge(rates, infrastructure, interface, baseline_peak=0, **kwargs):
    """aco.py:397-400: -demand_charge * peak."""
    p = peak(rates, infrastructure, interface, baseline_peak, **kwargs)
    dc = interface.get_demand_charge()
    return -dc * p
