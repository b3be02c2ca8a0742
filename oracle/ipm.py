"""ORACLE -- test infrastructure only.  Nothing under adacharge_amd/ may import it.

Independent high-accuracy CPU solver for the problems oracle/ref_problem.py
states: a primal-dual interior-point method (Mehrotra predictor-corrector,
Nesterov-Todd scaling) over the cone  R_+^l x Q_3 x ... x Q_3.

The reference delegates this step to cvxpy -> ECOS (aco.py:318; both
third-party, un-pinned in reference setup.py:24, absent from /root/reference
and from this image).  ECOS's published algorithm (Domahidi, Chu, Boyd, "ECOS:
An SOCP solver for embedded systems", ECC 2013) is exactly this family: NT-scaled
Mehrotra predictor-corrector on the symmetric cones; it is restated here in the
CVXOPT ``coneqp`` form so the quadratic objective is handled directly instead
of through cvxpy's epigraph reformulation.  Like ECOS, a path-following method
converges to the analytic centre of the optimal face on degenerate LPs.

Cross-checks (tests/test_oracle_kats.py): closed-form KAT-1, scipy-HiGHS
objective/aggregate on LP instances.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp
import scipy.sparse.linalg as spla


@dataclass
class IPMResult:
    x: np.ndarray
    status: str
    iters: int
    pcost: float
    gap: float
    pres: float
    dres: float
    duals: object = None


class _Cone:
    """l non-negative orthant rows followed by nq second-order cones of dim 3."""

    def __init__(self, l, nq, dq=3):
        self.l, self.nq, self.dq = l, nq, dq
        self.m = l + nq * dq
        self.degree = l + nq

    def split(self, u):
        return u[: self.l], u[self.l :].reshape(self.nq, self.dq)

    def e(self):
        e = np.zeros(self.m)
        e[: self.l] = 1.0
        e[self.l :: self.dq][: self.nq] = 1.0
        return e

    def prod(self, u, v):
        ul, uq = self.split(u)
        vl, vq = self.split(v)
        out = np.empty(self.m)
        out[: self.l] = ul * vl
        o = out[self.l :].reshape(self.nq, self.dq)
        o[:, 0] = np.sum(uq * vq, axis=1)
        o[:, 1:] = uq[:, :1] * vq[:, 1:] + vq[:, :1] * uq[:, 1:]
        return out

    def div(self, lam, v):
        """u with lam o u = v."""
        ll, lq = self.split(lam)
        vl, vq = self.split(v)
        out = np.empty(self.m)
        out[: self.l] = vl / ll
        o = out[self.l :].reshape(self.nq, self.dq)
        l0, l1 = lq[:, 0], lq[:, 1:]
        det = l0 * l0 - np.sum(l1 * l1, axis=1)
        l1v1 = np.sum(l1 * vq[:, 1:], axis=1)
        o[:, 0] = (l0 * vq[:, 0] - l1v1) / det
        o[:, 1:] = (
            -l1 * vq[:, :1] + (det[:, None] * vq[:, 1:] + l1 * l1v1[:, None]) / l0[:, None]
        ) / det[:, None]
        return out

    def min_eig(self, u):
        """largest t with u - t e in K."""
        ul, uq = self.split(u)
        vals = []
        if self.l:
            vals.append(ul.min())
        if self.nq:
            vals.append((uq[:, 0] - np.linalg.norm(uq[:, 1:], axis=1)).min())
        return min(vals) if vals else np.inf

    def max_step(self, lam, d):
        """largest alpha >= 0 with lam + alpha d in K (inf if unbounded)."""
        ll, lq = self.split(lam)
        dl, dq = self.split(d)
        alpha = np.inf
        if self.l:
            neg = dl < 0
            if neg.any():
                alpha = min(alpha, float(np.min(-ll[neg] / dl[neg])))
        if self.nq:
            # map lam to e with the cone automorphism W_lam^-1 (W_lam e = lam):
            # lam + a d in K  <=>  e + a W_lam^-1 d in K  <=>  a (|dt_1| - dt_0) <= 1
            nl = np.linalg.norm(lq[:, 1:], axis=1)
            a = np.sqrt((lq[:, 0] - nl) * (lq[:, 0] + nl))
            lb = lq / a[:, None]
            w0, w1 = lb[:, 0], -lb[:, 1:]
            w1d1 = np.sum(w1 * dq[:, 1:], axis=1)
            dt0 = (w0 * dq[:, 0] + w1d1) / a
            dt1 = (w1 * dq[:, :1] + dq[:, 1:] + w1 * (w1d1 / (1.0 + w0))[:, None]) / a[:, None]
            r = np.linalg.norm(dt1, axis=1) - dt0
            rmax = float(r.max())
            if rmax > 0:
                alpha = min(alpha, 1.0 / rmax)
        return alpha

    def nt_scaling(self, s, z):
        """Returns (W^-1 as sparse block-diag, apply_W, apply_Winv, lam)."""
        sl, sq = self.split(s)
        zl, zq = self.split(z)
        wl = np.sqrt(sl / zl)
        blocks_diag = zl / sl  # W^-2 for the orthant
        if self.nq:
            J = np.ones(self.dq)
            J[1:] = -1.0
            ns, nz = np.linalg.norm(sq[:, 1:], axis=1), np.linalg.norm(zq[:, 1:], axis=1)
            sJs = (sq[:, 0] - ns) * (sq[:, 0] + ns)
            zJz = (zq[:, 0] - nz) * (zq[:, 0] + nz)
            sb = sq / np.sqrt(sJs)[:, None]
            zb = zq / np.sqrt(zJz)[:, None]
            gamma = np.sqrt((1.0 + np.sum(sb * zb, axis=1)) / 2.0)
            wb = (sb + zb * J[None, :]) / (2.0 * gamma)[:, None]
            eta = (sJs / zJz) ** 0.25
        else:
            wb = np.zeros((0, self.dq))
            eta = np.zeros(0)
            J = np.ones(self.dq)

        def _Wq(wvec, sign, u):  # (nq, dq) -> W (sign=+1) or W^-1 (sign=-1) applied
            w0, w1 = wvec[:, 0], wvec[:, 1:] * sign
            u0, u1 = u[:, 0], u[:, 1:]
            w1u1 = np.sum(w1 * u1, axis=1)
            out = np.empty_like(u)
            out[:, 0] = w0 * u0 + w1u1
            out[:, 1:] = w1 * u0[:, None] + u1 + w1 * (w1u1 / (1.0 + w0))[:, None]
            return out

        def apply_W(u):
            ul, uq = self.split(u)
            out = np.empty(self.m)
            out[: self.l] = wl * ul
            if self.nq:
                out[self.l :] = (_Wq(wb, +1.0, uq) * eta[:, None]).reshape(-1)
            return out

        def apply_Winv(u):
            ul, uq = self.split(u)
            out = np.empty(self.m)
            out[: self.l] = ul / wl
            if self.nq:
                out[self.l :] = (_Wq(wb, -1.0, uq) / eta[:, None]).reshape(-1)
            return out

        # W^-1 as an explicit sparse block-diagonal (symmetric); H = P + (W^-1 G)'(W^-1 G)
        # keeps H positive semidefinite by construction (no 2ww'-J cancellation).
        mats = [sp.diags(1.0 / wl)] if self.l else []
        if self.nq:
            Jw = wb * J[None, :]
            dq = self.dq
            blk = np.zeros((self.nq, dq, dq))
            blk[:, 0, 0] = wb[:, 0]
            blk[:, 0, 1:] = -wb[:, 1:]
            blk[:, 1:, 0] = -wb[:, 1:]
            blk[:, 1:, 1:] = np.eye(dq - 1)[None] + wb[:, 1:, None] * wb[:, None, 1:] / (1.0 + wb[:, 0])[:, None, None]
            blk /= eta[:, None, None]
            mats.append(sp.block_diag([b for b in blk], format="csr"))
        Winv = sp.block_diag(mats, format="csr") if mats else sp.csr_matrix((0, 0))
        lam = apply_W(z)
        return Winv, apply_W, apply_Winv, lam


class _KKT:
    """Solves  [H A'; A 0][dx; dy] = [r1; r2]  with H = P + G' W^-2 G."""

    def __init__(self, P, G, A, Winv):
        Gs = (Winv @ G).tocsr()
        H = (P + Gs.T @ Gs).tocsc()
        n = H.shape[0]
        self.H = H
        self.A = A
        self.p = A.shape[0]
        if n <= 2500:
            Hd = H.toarray()
            Hd[np.diag_indices(n)] *= 1.0 + 1e-14
            self._cho = sla.cho_factor(Hd, lower=True, check_finite=False)
            self._solve0 = lambda r: sla.cho_solve(self._cho, r, check_finite=False)
        else:
            H = H + sp.diags(H.diagonal() * 1e-14, format="csc")
            lu = spla.splu(H)
            self._solve0 = lu.solve
        if self.p:
            Ad = A.toarray()
            self._HinvAt = self._solveH(Ad.T)
            S = Ad @ self._HinvAt
            self._S = sla.cho_factor(S + 1e-14 * np.eye(self.p), lower=True)

    def _solveH(self, r):
        x = self._solve0(r)
        for _ in range(2):  # iterative refinement against the unregularised H
            x = x + self._solve0(r - self.H @ x)
        return x

    def solve(self, r1, r2):
        if not self.p:
            return self._solveH(r1), np.zeros(0)
        t = self._solveH(r1)
        dy = sla.cho_solve(self._S, self.A @ t - r2)
        dx = t - self._HinvAt @ dy
        return dx, dy


def solve_conic_qp(P, q, G, h, l, nq, A=None, b=None, tol=1e-9, max_iter=100, verbose=False):
    """min 1/2 x'Px + q'x  s.t.  Gx + s = h, s in R_+^l x Q_3^nq,  Ax = b."""
    n = len(q)
    cone = _Cone(l, nq)
    G = sp.csr_matrix(G)
    P = sp.csr_matrix(P)
    if A is None:
        A = sp.csr_matrix((0, n))
        b = np.zeros(0)
    A = sp.csr_matrix(A)
    e = cone.e()
    # starting point (CVXOPT coneqp): W = I
    kkt0 = _KKT(P, G, A, sp.identity(cone.m, format="csr"))
    x, y = kkt0.solve(-q + G.T @ h, b)
    zt = G @ x - h
    s = -zt
    ts = cone.min_eig(s)
    if ts <= 0:
        s = s + (1.0 - ts) * e
    z = zt.copy()
    tz = cone.min_eig(z)
    if tz <= 0:
        z = z + (1.0 - tz) * e
    nrm_q, nrm_b, nrm_h = max(1.0, np.linalg.norm(q)), max(1.0, np.linalg.norm(b)), max(1.0, np.linalg.norm(h))
    status = "max_iter"
    best = (np.inf, x, (y, z, s), 0.0, np.inf, np.inf, np.inf)
    stall = 0
    for it in range(max_iter):
        rx = P @ x + q + A.T @ y + G.T @ z
        ry = A @ x - b
        rz = G @ x + s - h
        gap = float(s @ z)
        pcost = float(0.5 * x @ (P @ x) + q @ x)
        dcost = pcost + float(y @ ry) + float(z @ rz) - gap
        pres = max(np.linalg.norm(ry) / nrm_b, np.linalg.norm(rz) / nrm_h)
        dres = np.linalg.norm(rx) / nrm_q
        relgap = gap / max(1.0, abs(pcost), abs(dcost))
        if verbose:
            print(f"{it:3d} pcost {pcost:+.10e} gap {gap:.2e} pres {pres:.2e} dres {dres:.2e}")
        merit = max(pres, dres, min(gap, relgap))
        if np.isfinite(merit) and merit < best[0]:
            best = (merit, x.copy(), (y.copy(), z.copy(), s.copy()), pcost, gap, pres, dres)
        if pres < tol and dres < tol and (gap < tol or relgap < tol):
            status = "optimal"
            break
        # infeasibility heuristics (certificates): h'z + b'y < 0 with G'z + A'y ~ 0
        hz = float(h @ z + b @ y)
        if hz < 0 and np.linalg.norm(G.T @ z + A.T @ y + P @ x * 0) / (-hz) < 1e-9 and pres > 1e-6 and it > 5:
            status = "primal_infeasible"
            break
        try:
            Winv, apply_W, apply_Winv, lam = cone.nt_scaling(s, z)
            kkt = _KKT(P, G, A, Winv)
        except (np.linalg.LinAlgError, ValueError):
            status = "numerical_failure"  # diverging iterates (typically an infeasible problem)
            break
        W2inv = Winv @ Winv

        def newton(ds_rhs):
            t = apply_W(cone.div(lam, ds_rhs))  # W'(lam <> ds_rhs), W symmetric
            r1 = -rx - G.T @ (W2inv @ (rz + t))
            dx, dy = kkt.solve(r1, -ry)
            dz = W2inv @ (G @ dx + rz + t)
            ds = -rz - G @ dx  # primal row of the Newton system, free of W cancellation
            return dx, dy, dz, ds

        lam2 = cone.prod(lam, lam)
        dx, dy, dz, ds = newton(-lam2)
        ds_s, dz_s = apply_Winv(ds), apply_W(dz)
        a_aff = min(1.0, cone.max_step(lam, ds_s), cone.max_step(lam, dz_s))
        sigma = (1.0 - a_aff) ** 3
        mu = gap / cone.degree
        dx, dy, dz, ds = newton(-lam2 - cone.prod(ds_s, dz_s) + sigma * mu * e)
        ds_s, dz_s = apply_Winv(ds), apply_W(dz)
        step = min(1.0, 0.99 * min(cone.max_step(lam, ds_s), cone.max_step(lam, dz_s)))
        x = x + step * dx
        y = y + step * dy
        z = z + step * dz
        s = s + step * ds
        stall = stall + 1 if step < 1e-4 else 0
        if not np.all(np.isfinite(x)) or not (cone.min_eig(s) > 0 and cone.min_eig(z) > 0) or stall >= 3:
            status = "stalled"
            break
    if status == "numerical_failure" and best[0] > 1e-6:
        return IPMResult(x, status, it + 1, pcost, gap, pres, dres), (y, z, s)
    if status != "optimal" and status != "primal_infeasible":
        # ECOS-like reduced-accuracy exit: keep the best iterate seen
        merit, x, (y, z, s), pcost, gap, pres, dres = best
        if merit < max(tol, 1e-9):
            status = "optimal"
        elif merit < 1e-6:
            status = "optimal_inaccurate"
    return IPMResult(x, status, it + 1, pcost, gap, pres, dres), (y, z, s)


def solve_reference_problem(prob, tol=1e-9, max_iter=100, verbose=False):
    """Solve a oracle.ref_problem.RefProblem; returns (rates (N,T), IPMResult)."""
    n = prob.n
    l = prob.A_ub.shape[0]
    Gs = [prob.A_ub]
    hs = [prob.b_ub]
    for F, g in prob.soc:
        Gs.append(sp.vstack([sp.csr_matrix((1, n)), -F], format="csr"))
        hs.append(np.array([g, 0.0, 0.0]))
    G = sp.vstack(Gs, format="csr")
    h = np.concatenate(hs)
    res, (y, z, s) = solve_conic_qp(
        prob.P, prob.q, G, h, l, len(prob.soc), prob.A_eq, prob.b_eq,
        tol=tol, max_iter=max_iter, verbose=verbose,
    )
    res.duals = (z, s)
    return prob.rates_of(res.x), res


def solve_lp_highs(prob):
    """scipy-HiGHS on an LP instance (P = 0, no SOC): independent cross-check."""
    from scipy.optimize import linprog

    assert prob.P.nnz == 0 or abs(prob.P).max() == 0
    assert not prob.soc
    r = linprog(
        prob.q,
        A_ub=prob.A_ub,
        b_ub=prob.b_ub,
        A_eq=prob.A_eq if prob.A_eq.shape[0] else None,
        b_eq=prob.b_eq if prob.A_eq.shape[0] else None,
        bounds=(None, None),
        method="highs",
    )
    return r


# ---------------------------------------------------------------------------
# Active-set polish with a KKT certificate
# ---------------------------------------------------------------------------
@dataclass
class KKTCertificate:
    stationarity: float   # max |P x + q + A' nu| over free coordinates
    primal: float         # max constraint violation
    dual: float           # most negative inequality multiplier (0 if none)
    n_active: int

    @property
    def worst(self):
        return max(self.stationarity, self.primal, self.dual)


def _initial_active_set(prob, x, duals, act_tol):
    n = prob.n
    lb, ub = prob.lb.reshape(-1), prob.ub.reshape(-1)
    A_lin, b_lin = prob.A_ub[2 * n :], prob.b_ub[2 * n :]
    nsoc = len(prob.soc)
    if duals is not None:
        z, s = duals
        l = prob.A_ub.shape[0]
        zl, sl = z[:l], s[:l]
        at_lb = (zl[:n] > sl[:n]) | (lb >= ub)
        at_ub = (zl[n : 2 * n] > sl[n : 2 * n]) | (lb >= ub)
        act_lin = zl[2 * n :] > sl[2 * n :]
        zq, sq = z[l:].reshape(nsoc, 3), s[l:].reshape(nsoc, 3)
        act_soc = zq[:, 0] > (sq[:, 0] - np.linalg.norm(sq[:, 1:], axis=1))
    else:
        scale = max(1.0, float(np.max(np.abs(ub))))
        at_lb = x <= lb + act_tol * scale
        at_ub = x >= ub - act_tol * scale
        act_lin = (b_lin - A_lin @ x) <= act_tol * np.maximum(1.0, np.abs(b_lin))
        soc_val = np.array([np.linalg.norm(F @ x) for F, _ in prob.soc]) if nsoc else np.zeros(0)
        soc_g = np.array([g for _, g in prob.soc]) if nsoc else np.zeros(0)
        act_soc = soc_val >= soc_g * (1.0 - act_tol)
    return at_lb, at_ub, act_lin, act_soc


def _polish_once(prob, x0, at_lb, at_ub, act_lin, act_soc, newton_iters=12):
    """Newton on the KKT equations of one active-set guess.  Returns
    (x, multipliers..., certificate pieces)."""
    n = prob.n
    x = x0.copy()
    lb, ub = prob.lb.reshape(-1), prob.ub.reshape(-1)
    x[at_lb] = lb[at_lb]
    x[at_ub & ~at_lb] = ub[at_ub & ~at_lb]
    fixed = at_lb | at_ub
    free = np.nonzero(~fixed)[0]
    nf = len(free)
    A_lin, b_lin = prob.A_ub[2 * n :], prob.b_ub[2 * n :]
    il = np.nonzero(act_lin)[0]
    ks = np.nonzero(act_soc)[0]
    neq = prob.A_eq.shape[0]
    C_lin = sp.vstack([A_lin[il], prob.A_eq], format="csr") if (len(il) + neq) else sp.csr_matrix((0, n))
    d_lin = np.concatenate([b_lin[il], prob.b_eq])
    Pd = prob.P.tocsr()
    Fs = [prob.soc[k][0] for k in ks]
    gs = np.array([prob.soc[k][1] for k in ks])
    nu_lin = np.zeros(C_lin.shape[0])
    nu_soc = np.zeros(len(ks))
    Pff = Pd[free][:, free].toarray()
    Cf = C_lin[:, free].toarray()
    for _ in range(newton_iters):
        grad = Pd @ x + prob.q + C_lin.T @ nu_lin
        Hs = Pff.copy()
        cols = []
        r_soc = np.zeros(len(ks))
        for k, F in enumerate(Fs):
            Fx = F @ x
            gk = F.T @ Fx / gs[k]
            grad = grad + nu_soc[k] * gk
            Ff = F[:, free].toarray()
            Hs += nu_soc[k] * (Ff.T @ Ff) / gs[k]
            cols.append(gk[free])
            r_soc[k] = (Fx @ Fx - gs[k] ** 2) / (2 * gs[k])
        Jc = np.vstack([Cf] + [c[None, :] for c in cols]) if (Cf.shape[0] + len(cols)) else np.zeros((0, nf))
        r_c = np.concatenate([C_lin @ x - d_lin, r_soc])
        mc = Jc.shape[0]
        K = np.block([[Hs, Jc.T], [Jc, np.zeros((mc, mc))]])
        rhs = -np.concatenate([grad[free], r_c])
        if rhs.size == 0 or np.max(np.abs(rhs)) < 1e-13:
            break
        sol = np.linalg.lstsq(K, rhs, rcond=1e-13)[0]
        x[free] += sol[:nf]
        nu_lin += sol[nf : nf + C_lin.shape[0]]
        nu_soc += sol[nf + C_lin.shape[0] :]
    grad = Pd @ x + prob.q + C_lin.T @ nu_lin
    for k, F in enumerate(Fs):
        grad = grad + nu_soc[k] * (F.T @ (F @ x)) / gs[k]
    return x, grad, nu_lin, nu_soc, free, il, ks


def polish(prob, rates, duals=None, act_tol=1e-5, max_rounds=25):
    """Primal-dual active-set refinement with a KKT certificate.

    Bounds are handled by fixing coordinates (the first 2n rows of prob.A_ub
    are -I, I by construction in ref_problem.build_reference_problem); the
    remaining active linear rows, the equality rows and the active SOC
    constraints ((|F x|^2 - g^2) / (2 g) = 0) are imposed as equalities on the
    free coordinates and solved by Newton.  The first guess comes from the
    IPM's (z, s) (``z_i > s_i``); after each solve, violated constraints join
    the set and constraints with a negative multiplier leave it.  The returned
    certificate is evaluated on the *full* problem, so a wrong final guess
    shows up as a large ``primal`` or ``dual`` entry, never as a silently
    wrong answer."""
    n = prob.n
    x = np.asarray(rates, float).reshape(-1).copy()
    lb, ub = prob.lb.reshape(-1), prob.ub.reshape(-1)
    A_lin, b_lin = prob.A_ub[2 * n :], prob.b_ub[2 * n :]
    soc_g = np.array([g for _, g in prob.soc]) if prob.soc else np.zeros(0)
    at_lb, at_ub, act_lin, act_soc = _initial_active_set(prob, x, duals, act_tol)
    eps = 1e-9
    best = None
    for rnd in range(max_rounds):
        xr, grad, nu_lin, nu_soc, free, il, ks = _polish_once(prob, x, at_lb, at_ub, act_lin, act_soc)
        stat = float(np.max(np.abs(grad[free]))) if len(free) else 0.0
        v_lb, v_ub = lb - xr, xr - ub
        v_lin = (A_lin @ xr - b_lin) if A_lin.shape[0] else np.zeros(0)
        v_soc = (np.array([np.linalg.norm(F @ xr) for F, _ in prob.soc]) - soc_g) if prob.soc else np.zeros(0)
        pr = max([0.0, v_lb.max(), v_ub.max()] + ([v_lin.max()] if v_lin.size else []) + ([v_soc.max()] if v_soc.size else []))
        if prob.A_eq.shape[0]:
            pr = max(pr, float(np.max(np.abs(prob.A_eq @ xr - prob.b_eq))))
        only_lb = at_lb & ~at_ub & (lb < ub)
        only_ub = at_ub & ~at_lb & (lb < ub)
        m_lb = np.where(only_lb, grad, np.inf)     # multiplier of x >= lb is +grad
        m_ub = np.where(only_ub, -grad, np.inf)    # multiplier of x <= ub is -grad
        m_lin = nu_lin[: len(il)]
        du = max([0.0, -m_lb.min(), -m_ub.min()] + ([-m_lin.min()] if m_lin.size else []) + ([-nu_soc.min()] if nu_soc.size else []))
        cert = KKTCertificate(stat, float(pr), float(du), int((at_lb | at_ub).sum() + len(il) + len(ks)))
        if best is None or cert.worst < best[1].worst:
            best = (xr.copy(), cert)
        if cert.worst < eps:
            break
        # primal violations join, negative multipliers leave
        changed = False
        add = v_lb > eps
        if add.any():
            at_lb = at_lb | add; changed = True
        add = v_ub > eps
        if add.any():
            at_ub = at_ub | add; changed = True
        if v_lin.size and (v_lin > eps).any():
            act_lin = act_lin | (v_lin > eps); changed = True
        if v_soc.size and (v_soc > eps).any():
            act_soc = act_soc | (v_soc > eps); changed = True
        if not changed:
            drop = m_lb < -eps
            if drop.any():
                at_lb = at_lb & ~drop; changed = True
            drop = m_ub < -eps
            if drop.any():
                at_ub = at_ub & ~drop; changed = True
            if m_lin.size and (m_lin < -eps).any():
                act_lin = act_lin.copy(); act_lin[il[m_lin < -eps]] = False; changed = True
            if nu_soc.size and (nu_soc < -eps).any():
                act_soc = act_soc.copy(); act_soc[ks[nu_soc < -eps]] = False; changed = True
        if not changed:
            break
        x = np.clip(xr, lb, ub)
    xr, cert = best
    return (xr if prob.n_extra else xr.reshape(prob.N, prob.T)), cert


def solve_certified(prob, tol=1e-9, verbose=False):
    """IPM followed by the polish; returns (rates, IPMResult, KKTCertificate).
    The polished point is kept only if its certificate is at least as good as
    1e-7 on every entry; otherwise the raw IPM point is returned together with
    the (bad) certificate so the caller can see it."""
    r0, res = solve_reference_problem(prob, tol=tol, verbose=verbose)
    if res.status not in ("optimal", "optimal_inaccurate"):
        return r0, res, None
    r1, cert = polish(prob, res.x, duals=getattr(res, "duals", None))
    if cert.worst < 1e-7:
        return prob.rates_of(r1), res, cert
    return r0, res, cert


def aggregate_range_on_optimal_face(prob, fstar, t, rel=1e-8):
    """ORACLE (test infrastructure).  quick_charge's cost depends only on the per-period aggregates, and its weights
    (T - t) / T are equally spaced: moving one ampere one period earlier gains 1 / T wherever it happens, so the optimal
    FACE of the LP often spans a range of aggregates in a given period (SURVEY.md H2 assumed the aggregate unique; it is
    only so on instances like KAT-4).  Returns (min, max) of sum_i x[i, t] over {x feasible, q'x <= fstar + rel |fstar|},
    two HiGHS solves: the yardstick for "which optimal point did the regularised solver pick"."""
    from scipy.optimize import linprog
    import scipy.sparse as sp

    assert not prob.soc
    N, T = prob.N, prob.T
    e = np.zeros((N, T))
    e[:, t] = 1.0
    A = sp.vstack([prob.A_ub, sp.csr_matrix(prob.q.reshape(1, -1))]).tocsr()
    b = np.concatenate([prob.b_ub, [fstar + rel * abs(fstar)]])
    out = []
    for sign in (1.0, -1.0):
        r = linprog(sign * e.reshape(-1), A_ub=A, b_ub=b, A_eq=prob.A_eq if prob.A_eq.shape[0] else None,
                    b_eq=prob.b_eq if prob.A_eq.shape[0] else None, bounds=(None, None), method="highs")
        assert r.status == 0, r.message
        out.append(sign * r.fun)
    return out[0], out[1]
