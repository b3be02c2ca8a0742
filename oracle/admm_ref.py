"""ORACLE -- test infrastructure only.  Nothing under adacharge_amd/ may import it.

Readable numpy restatement of the *device algorithm* (the ADMM the HIP kernel
runs, adacharge_amd/csrc/acn_qp_tiled.hpp), one problem at a time, fp64.  It
is not the reference's algorithm (the reference calls cvxpy/ECOS, aco.py:318);
it exists so that a kernel bug can be told apart from an algorithmic property:
the kernel must agree with this file to ~1e-9, and this file must agree with
the independent IPM oracle (oracle/ipm.py) to the solver tolerance.

Splitting (OSQP form, generalised from a box to closed convex sets/functions):

    minimise  1/2 pdiag |x|^2 + <q, x> + I_B(z1) + g(z2)
    s.t.      x = z1,   G x_t = z2_t  for every period t

  B  = { lb <= z <= ub,  sum_{t in window_s} z[i_s, t] <= (==) cap_s }   (lane-local)
  g  = indicator of the site rows' box / disc / peak sets, or the quadratic
       load-flattening penalty on a row (prox instead of projection)

x-update: (sigma + pdiag + rho) x~ + rho G'G x~ = rhs, solved in closed form
through the eigen-decomposition of the tiny G G' (see builder.SiteData).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

ST_SOLVED = 1
ST_MAX_ITER = 2
ST_PRIMAL_INFEASIBLE = 3
ST_PRESOLVE_INFEASIBLE = 4
ST_SOLVED_INACCURATE = 5

CONE_LINEAR = 0
CONE_SOC = 1


@dataclass
class AdmmOptions:
    eps_abs: float = 1e-6
    eps_rel: float = 1e-6
    max_iter: int = 20000
    rho: float = 0.02
    sigma: float = 1e-6
    alpha: float = 1.4
    check_every: int = 20
    adaptive_rho: bool = True
    adapt_every: int = 20
    adapt_tol: float = 3.0
    reg_rel: float = 0.0
    equilibrate: bool = True      # row scaling of the site matrix, as the library does internally
    accel_mem: int = 0            # Anderson-acceleration columns (0 = plain ADMM), see _Anderson


START_GAIN = 1e5
ADAPT_WIDEN = 8.0
STALL_GAIN, STALL_NEAR, STALL_ITERS = 0.9, 1.25, 3000   # the device kernels' stall rule (acn_qp_tiled.hpp)
AA_PERIOD, AA_REG, AA_SAFE, AA_DRIFT = 5, 1e-4, 1.2, 1e-3


class _Anderson:
    """Type-II Anderson acceleration of the ADMM fixed-point map u -> g(u), u = the stacked
    pre-projection points (zhat1, zhat2), exactly as oracle/admm_port.c and the tiled HIP kernel do it:
    an *event* every AA_PERIOD iterations; columns dF = f - f_prev, dG = g - g_prev (stored as
    float32) in a ring of ``mem`` slots; gamma = argmin |f - dF gamma|^2 + eta |gamma|^2 with
    eta = AA_REG * trace(dF'dF); u_next = g - dG gamma.  If the residual norm grew by more than
    AA_SAFE after an accelerated step the ring is cleared and acceleration pauses for
    1, 2, 4, ... 64 events.  No extrapolation while |dF_new| <= AA_DRIFT |f| (the map is drifting:
    its differences carry rounding noise only)."""

    def __init__(self, mem, dim):
        self.m, self.dim = mem, dim
        self.uprev = np.zeros(dim)
        self.pen = 1
        self.cool = 0
        self.fn_prev = 0.0
        self.was = False
        self.restart(None)

    def restart(self, u):
        self.dF = np.zeros((self.m, self.dim), np.float32)
        self.dG = np.zeros((self.m, self.dim), np.float32)
        self.valid = np.zeros(self.m, bool)
        self.H = np.zeros((self.m, self.m))
        self.b = np.zeros(self.m)
        self.head = 0
        self.fprev = None
        self.gprev = None
        self.was = False
        if u is not None:
            self.uprev = u.copy()

    def event(self, g, may_apply):
        f = g - self.uprev
        fn = float(np.sqrt(f @ f))
        if self.was and fn > AA_SAFE * self.fn_prev:
            u = self.uprev
            self.restart(None)
            self.uprev = u
            self.cool, self.pen = self.pen, min(64, 2 * self.pen)
        elif self.cool > 0:
            self.cool -= 1
        if self.fprev is not None:
            c = self.head
            self.dF[c] = (f - self.fprev).astype(np.float32)
            self.dG[c] = (g - self.gprev).astype(np.float32)
            self.valid[c] = True
            col = self.dF[c].astype(np.float64)
            for j in np.flatnonzero(self.valid):
                d = float(col @ self.dF[j].astype(np.float64))
                self.H[c, j] = self.H[j, c] = d
                if j != c:
                    self.b[j] += d            # dF_j . f_k = dF_j . f_(k-1) + dF_j . dF_c
            self.b[c] = float(col @ f)
            self.head = (c + 1) % self.m
        informative = self.fprev is not None and self.H[(self.head - 1) % self.m, (self.head - 1) % self.m] > (AA_DRIFT * fn) ** 2
        self.fprev, self.gprev, self.fn_prev, self.was = f, g.copy(), fn, False
        u = g
        if self.valid.any() and self.cool == 0 and may_apply and informative:
            v = np.flatnonzero(self.valid)
            Hv = self.H[np.ix_(v, v)]
            eta = AA_REG * np.trace(Hv) + 1e-300
            gam = np.linalg.solve(Hv + eta * np.eye(len(v)), self.b[v])
            u = g - gam @ self.dG[v].astype(np.float64)
            self.was = True
        self.uprev = u.copy()
        return u


def project_window(v, lb, ub, cap, eq):
    """Euclidean projection of v onto {lb <= z <= ub, sum z <= cap (== cap if eq)}.
    Exact, by locating the root of the piecewise-linear g(mu) = sum clip(v - mu)."""
    z = np.clip(v, lb, ub)
    s = z.sum()
    if (not eq and s <= cap) or (eq and s == cap):
        return z
    lo_sum, hi_sum = lb.sum(), ub.sum()
    if cap >= hi_sum:
        return ub.copy() if eq else z
    if cap <= lo_sum:
        return lb.copy()
    bp = np.unique(np.concatenate([v - ub, v - lb]))
    gv = np.array([np.clip(v - m, lb, ub).sum() for m in bp])  # non-increasing in mu
    # find segment [bp[k], bp[k+1]] with gv[k] >= cap >= gv[k+1]
    k = int(np.searchsorted(-gv, -cap, side="left"))
    k = min(max(k, 1), len(bp) - 1)
    m0, m1, g0, g1 = bp[k - 1], bp[k], gv[k - 1], gv[k]
    mu = m0 if g0 == g1 else m0 + (g0 - cap) * (m1 - m0) / (g0 - g1)
    return np.clip(v - mu, lb, ub)


def _project_B(v, lb, ub, s_off, s_len, s_cap, eq):
    z = np.clip(v, lb, ub)
    K, N = s_len.shape
    for k in range(K):
        for i in range(N):
            L = s_len[k, i]
            if L > 0:
                o = s_off[k, i]
                z[i, o : o + L] = project_window(
                    v[i, o : o + L], lb[i, o : o + L], ub[i, o : o + L], s_cap[k, i], eq
                )
    return z


def _prox_rows(zh, rho, site, limits, peak_b, T, lf, lf_ext):
    """z-update for the site rows.  zh is (Mg, Tm)."""
    z = zh.copy()
    M = site.M
    if site.cone == CONE_SOC:
        re, im = zh[:M], zh[M : 2 * M]
        nrm = np.hypot(re, im)
        scale = np.where(nrm > limits[:, None], limits[:, None] / np.maximum(nrm, 1e-300), 1.0)
        z[:M] = re * scale
        z[M : 2 * M] = im * scale
        r = 2 * M
    else:
        z[:M] = np.minimum(zh[:M], limits[:, None])
        r = M
    if getattr(site, "has_flat", False):   # prox of 1/2 lf z^2 on the aggregate-power row
        z[r] = zh[r] * (rho / (rho + lf))
        r += 1
    if site.has_peak:
        z[r] = np.minimum(zh[r], peak_b)
        r += 1
    z[:, T:] = 0.0
    return z


def solve_one(batch, b, opts: AdmmOptions = AdmmOptions(), trace=None):
    """Run the ADMM on problem ``b`` of a builder.ProblemBatch-like object.
    Returns dict(x (N,Tm), status, iters, pri_res, dua_res, obj, rho)."""
    site = batch.site
    N, Tm, T = site.N, batch.Tm, int(batch.T[b])
    if batch.presolve_status is not None and batch.presolve_status[b]:
        return dict(x=np.zeros((N, Tm)), status=ST_PRESOLVE_INFEASIBLE, iters=0,
                    pri_res=np.inf, dua_res=np.inf, obj=np.nan, rho=opts.rho)
    lb, ub, q = batch.lb[b], batch.ub[b], batch.q[b]
    pdiag = float(batch.pdiag[b])
    # Tikhonov floor: LP-like problems only (kRegResolve / effective_pdiag in acn_qp_tiled.hpp)
    has_prox = float(batch.lf[b]) > 0 or (getattr(batch, "dc", None) is not None and float(batch.dc[b]) > 0)
    if ub.max() > 0 and not has_prox and pdiag * ub.max() <= 1e-6 * np.abs(q).max():
        pdiag = max(pdiag, opts.reg_rel * np.abs(q).max() / (ub.max() * max(1, T)))
    eq = bool(batch.s_eq[b])
    G, Gh, lam, Q, limits = site.G, site.Ghat, site.lam, site.Q, site.limits
    peak_b = batch.peak[b] if batch.peak is not None else None
    lf_b = float(batch.lf[b])
    if opts.equilibrate:
        from .admm_port import equilibrated
        G, Gh, Q, lam, limits, pk_s, fl_s, _ = equilibrated(site)
        peak_b = None if peak_b is None else peak_b * pk_s
        lf_b /= fl_s * fl_s
    Mg = G.shape[0]
    sig, alpha = opts.sigma, opts.alpha
    rho = opts.rho
    # start: the schedule that ignores the site rows, z1 = Proj_B(-START_GAIN q), with the multiplier that
    # makes it stationary, y1 = -(q + pdiag z1); site rows at z2 = G z1, y2 = 0 (exact when no site row binds)
    z1 = _project_B(-START_GAIN * q, lb, ub, batch.s_off[b], batch.s_len[b], batch.s_cap[b], eq)
    x = z1.copy()
    y1 = -(q + pdiag * z1)
    z2 = G @ z1
    y2 = np.zeros((Mg, Tm))
    Gx = z2.copy()
    status = ST_MAX_ITER
    pri = dua = np.inf
    it = 0
    n_adapt = 0
    best_score, best_it = np.inf, 0
    aa = _Anderson(opts.accel_mem, N * Tm + Mg * Tm) if opts.accel_mem > 0 else None
    if aa is not None:
        aa.uprev = np.concatenate([(z1 + y1 / rho).ravel(), z2.ravel()])
    for it in range(1, opts.max_iter + 1):
        a = sig + pdiag + rho
        r0 = sig * x - q + rho * z1 - y1
        w = rho * z2 - y2
        wh = Q.T @ w
        gh0 = Gh @ r0
        gh = gh0 + lam[:, None] * wh
        ch = rho * gh / (a + rho * lam[:, None])
        eh = wh - ch
        xt = (r0 + Gh.T @ eh) / a
        zt2 = Q @ ((gh0 + lam[:, None] * eh) / a)
        x = alpha * xt + (1 - alpha) * x
        Gx = alpha * zt2 + (1 - alpha) * Gx
        zh1 = alpha * xt + (1 - alpha) * z1 + y1 / rho
        zh2 = alpha * zt2 + (1 - alpha) * z2 + y2 / rho
        check = it % opts.check_every == 0 or it == opts.max_iter
        if aa is not None and it % AA_PERIOD == 0:
            u = aa.event(np.concatenate([zh1.ravel(), zh2.ravel()]), not check)
            zh1, zh2 = u[: N * Tm].reshape(N, Tm), u[N * Tm :].reshape(Mg, Tm)
        z1 = _project_B(zh1, lb, ub, batch.s_off[b], batch.s_len[b], batch.s_cap[b], eq)
        y1 = rho * (zh1 - z1)
        z2 = _prox_rows(zh2, rho, site, limits, peak_b, T, lf_b, None)
        y2 = rho * (zh2 - z2)
        if check:
            Gty = G.T @ y2
            pri = max(np.abs(x - z1).max(), np.abs(Gx - z2).max() if Mg else 0.0)
            dua = np.abs(pdiag * x + q + y1 + Gty).max()
            npri = max(np.abs(x).max(), np.abs(z1).max(), np.abs(Gx).max() if Mg else 0.0, np.abs(z2).max() if Mg else 0.0)
            ndua = max(np.abs(pdiag * x).max(), np.abs(q).max(), np.abs(y1).max(), np.abs(Gty).max())
            if trace is not None:
                trace.append((it, pri, dua, rho))
            if pri <= opts.eps_abs + opts.eps_rel * npri and dua <= opts.eps_abs + opts.eps_rel * ndua:
                status = ST_SOLVED
                break
            score = max(pri / max(opts.eps_abs + opts.eps_rel * npri, 1e-300), dua / max(opts.eps_abs + opts.eps_rel * ndua, 1e-300))
            if score < STALL_GAIN * best_score:
                best_score, best_it = score, it
            ea, er = max(100 * opts.eps_abs, 1e-5), max(100 * opts.eps_rel, 1e-5)   # 100 x tolerance or cvxpy's OSQP default
            inacc = pri <= ea + er * npri and dua <= ea + er * ndua
            # the device kernels' stall rule (acn_qp_tiled.hpp)
            stalled = it - best_it >= STALL_ITERS and score <= STALL_NEAR * best_score
            if (it == opts.max_iter or stalled) and inacc:
                status = ST_SOLVED_INACCURATE
            if stalled:
                break
            if opts.adaptive_rho and it % opts.adapt_every == 0:
                ratio = np.sqrt((pri / max(npri, 1e-12)) / max(dua / max(ndua, 1e-12), 1e-30))
                tol_eff = opts.adapt_tol * (1.0 + n_adapt / ADAPT_WIDEN)   # the band widens: no limit cycles
                if ratio > tol_eff or ratio < 1.0 / tol_eff:
                    n_adapt += 1
                    rho = float(np.clip(rho * ratio, 1e-6, 1e6))
                    if aa is not None:   # the fixed-point map changed: restart the ring from (z, y)
                        aa.restart(np.concatenate([(z1 + y1 / rho).ravel(), (z2 + y2 / rho).ravel()]))
    # the feasible iterate is z1 (it satisfies bounds and energy rows exactly)
    xs = z1
    obj = 0.5 * pdiag * (xs * xs).sum() + (q * xs).sum()
    return dict(x=xs, xraw=x, status=status, iters=it, pri_res=pri, dua_res=dua, obj=obj, rho=rho)
