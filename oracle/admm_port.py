"""ORACLE -- test infrastructure only.  ctypes wrapper of oracle/admm_port.c."""
import ctypes as C
import os

import numpy as np

from .build import build_oracle


class _Site(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("N", "Tm", "K", "Mg", "M", "cone", "has_peak", "has_flat", "has_max")] + [
        (k, C.c_void_p) for k in ("G", "Ghat", "Q", "lam", "limits")
    ]


class _Opts(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("eps_abs", "eps_rel", "rho", "sigma", "alpha", "adapt_tol", "reg_rel")] + [
        (k, C.c_int) for k in ("max_iter", "check_every", "adapt_every")
    ]


_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_oracle())
        _lib.admm_port_solve_batch.restype = C.c_int
        _lib.admm_port_max_threads.restype = C.c_int
    return _lib


def max_threads():
    return int(_load().admm_port_max_threads())


def solve_batch(batch, threads=1, eps_abs=1e-8, eps_rel=1e-8, rho=0.01, sigma=1e-6, alpha=1.4, adapt_tol=5.0,
                reg_rel=0.06, max_iter=20000, check_every=20, adapt_every=40):
    """Solve a builder.ProblemBatch-like object on the CPU; same defaults as
    acnqp_default_options.  Returns dict of arrays."""
    lib = _load()
    site = batch.site
    p = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
    keep = [np.ascontiguousarray(a, np.float64) for a in (site.G, site.Ghat, site.Q, site.lam, site.limits)]
    S = _Site(site.N, batch.Tm, batch.K, site.Mg, site.M, int(site.cone), int(site.has_peak), int(getattr(site, 'has_flat', False)), int(getattr(site, 'has_max', False)), *[p(a) for a in keep])
    O = _Opts(eps_abs, eps_rel, rho, sigma, alpha, adapt_tol, reg_rel, max_iter, check_every, adapt_every)
    B, N, Tm = batch.B, site.N, batch.Tm
    arrs = [np.ascontiguousarray(batch.T, np.int32), np.ascontiguousarray(batch.lb, np.float64), np.ascontiguousarray(batch.ub, np.float64),
            np.ascontiguousarray(batch.q, np.float64), np.ascontiguousarray(batch.pdiag, np.float64),
            np.ascontiguousarray(batch.lf, np.float64),
            np.ascontiguousarray(batch.dc if batch.dc is not None else np.zeros(batch.B), np.float64),
            np.ascontiguousarray(batch.dfloor if batch.dfloor is not None else np.zeros(batch.B), np.float64),
            np.ascontiguousarray(batch.s_off, np.int32), np.ascontiguousarray(batch.s_len, np.int32),
            np.ascontiguousarray(batch.s_cap, np.float64), np.ascontiguousarray(batch.s_eq, np.uint8)]
    peak = None if batch.peak is None else np.ascontiguousarray(batch.peak, np.float64)
    x = np.zeros((B, N, Tm)); status = np.zeros(B, np.int32); iters = np.zeros(B, np.int32)
    pri = np.zeros(B); dua = np.zeros(B); obj = np.zeros(B)
    lib.admm_port_solve_batch(
        C.byref(S), C.byref(O), C.c_int(B), *[p(a) for a in arrs], None if peak is None else p(peak),
        p(x), p(status), p(iters), p(pri), p(dua), p(obj), C.c_int(int(threads)),
    )
    return dict(x=x, status=status, iters=iters, pri_res=pri, dua_res=dua, obj=obj)
