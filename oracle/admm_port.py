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
        (k, C.c_int) for k in ("max_iter", "check_every", "adapt_every", "accel_mem", "stall_iters", "retry_passes", "retry_max_iter")
    ] + [(k, C.c_double) for k in ("retry_rho", "inaccurate_floor")]


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


def equilibrated(site):
    """Row equilibration exactly as libacn_qp_hip does it internally (acn_qp_api.hip, build_site_dev):
    every site row (a SOC pair = one row) scaled by 1/sqrt(|g|_2).  Returns
    (G, Ghat, Q, lam, limits, peak_scale, flat_scale, max_scale)."""
    G = np.array(site.G, dtype=np.float64, copy=True)
    lim = np.array(site.limits, dtype=np.float64, copy=True)
    M = site.M
    scale = np.ones(G.shape[0])
    for j in range(G.shape[0]):
        if site.cone == 1 and M <= j < 2 * M:
            continue
        n2 = float((G[j] ** 2).sum())
        if site.cone == 1 and j < M:
            n2 += float((G[j + M] ** 2).sum())
        sc = 1.0 / np.sqrt(np.sqrt(n2)) if n2 > 0 else 1.0
        scale[j] = sc
        if site.cone == 1 and j < M:
            scale[j + M] = sc
    G *= scale[:, None]
    lim *= scale[:M]
    if G.shape[0]:
        lam, Q = np.linalg.eigh(G @ G.T)
        lam = np.maximum(lam, 0.0)
        lam[lam < 1e-12 * max(1.0, lam.max())] = 0.0
        Gh = Q.T @ G
        Gh[lam == 0.0] = 0.0
    else:
        lam, Q, Gh = np.zeros(0), np.zeros((0, 0)), np.zeros((0, site.N))
    nrows = G.shape[0]
    pk = scale[nrows - 1] if site.has_peak else 1.0
    mx = scale[nrows - 1 - int(site.has_peak)] if getattr(site, "has_max", False) else 1.0
    fl = scale[nrows - 1 - int(site.has_peak) - int(getattr(site, "has_max", False))] if getattr(site, "has_flat", False) else 1.0
    equilibrated.last_scale = scale   # per row: y (caller's units) = scale * y (this solver's units)
    return G, np.ascontiguousarray(Gh), np.ascontiguousarray(Q), lam, lim, pk, fl, mx


def solve_batch(batch, threads=1, eps_abs=1e-8, eps_rel=1e-8, rho=0.02, sigma=1e-6, alpha=1.4, adapt_tol=3.0,
                reg_rel=0.06, max_iter=20000, check_every=20, adapt_every=20, accel_mem=0, warm_x=None, warm_y=None,
                stall_iters=3000, retry_passes=2, retry_max_iter=8000, retry_rho=0.5, inaccurate_floor=1e-5):
    """Solve a builder.ProblemBatch-like object on the CPU; same defaults as
    acnqp_default_options.  Returns dict of arrays."""
    lib = _load()
    site = batch.site
    p = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
    Ge, Ghe, Qe, lame, lime, pk_s, fl_s, mx_s = equilibrated(site)
    keep = [np.ascontiguousarray(a, np.float64) for a in (Ge, Ghe, Qe, lame, lime)]
    S = _Site(site.N, batch.Tm, batch.K, site.Mg, site.M, int(site.cone), int(site.has_peak), int(getattr(site, 'has_flat', False)), int(getattr(site, 'has_max', False)), *[p(a) for a in keep])
    O = _Opts(eps_abs, eps_rel, rho, sigma, alpha, adapt_tol, reg_rel, max_iter, check_every, adapt_every, int(accel_mem),
              int(stall_iters), int(retry_passes), int(retry_max_iter), float(retry_rho), float(inaccurate_floor))
    B, N, Tm = batch.B, site.N, batch.Tm
    arrs = [np.ascontiguousarray(batch.T, np.int32), np.ascontiguousarray(batch.lb, np.float64), np.ascontiguousarray(batch.ub, np.float64),
            np.ascontiguousarray(batch.q, np.float64), np.ascontiguousarray(batch.pdiag, np.float64),
            np.ascontiguousarray(batch.lf / (fl_s * fl_s), np.float64),
            np.ascontiguousarray((batch.dc if batch.dc is not None else np.zeros(batch.B)) / mx_s, np.float64),
            np.ascontiguousarray((batch.dfloor if batch.dfloor is not None else np.zeros(batch.B)) * mx_s, np.float64),
            np.ascontiguousarray(batch.s_off, np.int32), np.ascontiguousarray(batch.s_len, np.int32),
            np.ascontiguousarray(batch.s_cap, np.float64), np.ascontiguousarray(batch.s_eq, np.uint8)]
    peak = None if batch.peak is None else np.ascontiguousarray(batch.peak * pk_s, np.float64)
    x = np.zeros((B, N, Tm)); status = np.zeros(B, np.int32); iters = np.zeros(B, np.int32)
    pri = np.zeros(B); dua = np.zeros(B); obj = np.zeros(B); y = np.zeros((B, site.Mg, Tm))
    scale = equilibrated.last_scale
    wx = None if warm_x is None else np.ascontiguousarray(warm_x, np.float64)
    wy = None if warm_y is None else np.ascontiguousarray(np.asarray(warm_y, float) / scale[None, :, None], np.float64)
    lib.admm_port_solve_batch(
        C.byref(S), C.byref(O), C.c_int(B), *[p(a) for a in arrs], None if peak is None else p(peak),
        p(x), p(status), p(iters), p(pri), p(dua), p(obj), C.c_int(int(threads)),
        None if wx is None else p(wx), None if wy is None else p(wy), p(y),
    )
    return dict(x=x, status=status, iters=iters, pri_res=pri, dua_res=dua, obj=obj, y=y * scale[None, :, None])
