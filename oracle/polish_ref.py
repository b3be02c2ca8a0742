"""ORACLE -- test infrastructure only.  Nothing under adacharge_amd/ may import it.

numpy restatement of the DEVICE-SIDE POLISH (adacharge_amd/csrc/acn_qp_polish.hpp): the active-set Newton method that
takes over from the ADMM when a problem has not converged after ``options.polish_iters`` iterations.  The reference's
solver is an interior-point method (ECOS through cvxpy, aco.py:318): it never sits on the plateau a first-order method
reaches on the tangentially degenerate congested instances -- the polish is what gives the HIP
path that robustness.  Same arithmetic, same thresholds and the same order of decisions as the kernel, so that a
disagreement between the two is a device bug; the RESULT is checked against the IPM certificates of
tests/golden/stalled.npz, never against this file alone.

Problem (one MPC instance, acn_qp.h):  min 1/2 pd |x|^2 + <q, x>  over  lb <= x <= ub,  sum_window x <= (==) cap per
session,  per period t and site row j:  g_j' x_t <= lim_j (LINEAR) or |(g_j' x_t, g_{j+M}' x_t)| <= lim_j (SOC),
sum_i x_it <= peak_t.

Method.  Working set W = {variables at a bound} + {tight energy rows} + {tight site rows}, first guess from the ADMM
iterate (x on a bound; energy row tight; site-row multiplier non-zero).  A round solves the Newton (SQP) step of the
equality-constrained problem on the free variables,
    pd dx + E' mu + A' lam + W' sig = -(pd x + q),   E dx = c_E,   A dx = c_A,   W dx - D sig = 0,
E: tight energy rows, A: linearised tight site rows (a disc row: its outward normal), W / D: the discs' curvature
(tangent rows, D = |u| / nu), by block elimination: the energy rows have disjoint supports, so they reduce to the
projector P = I - sum_s 1_s 1_s' / n_s (n_s: free variables of the session's window), and what is left is the SPD system
    (R P R' + pd diag(0, D)) lam = R (-P g + pd e) - pd (c_A, 0),     R = [A; W],  e = E' (c_E / n)
of the size of the tight site rows -- solved without forming it (structured_solve: per-period blocks + Woodbury).  Then a ratio test against everything NOT in the
working set (step length alpha <= 1 to the first blocking constraint, which joins W), and -- once a full step has
converged -- the most negative multiplier leaves W.  KKT on the full problem is verified before the answer is accepted.
"""
from __future__ import annotations

import numpy as np

MAX_ROWS = 256          # rows of the Schur system at most (the kernel keeps it in LDS up to what fits, in L2 beyond)
MAX_ROUNDS = 96
TOL_BOUND = 1e-7        # |x - bound| below which the ADMM iterate counts as "on the bound"
TOL_ROW = 1e-9          # relative size of a site-row multiplier that counts as non-zero
TOL_STEP = 1e-7         # convergence of a round: |dx|_inf <= TOL_STEP max(1, |x|_inf) (the regularised Schur solve leaves
                        # ~4e-8 A of noise in dx on the degenerate instances: 1e-9 cost six idle rounds; the KKT check decides)
TOL_DUAL = 1e-9         # a multiplier below -TOL_DUAL max(1, |q|_inf) leaves the working set
TOL_PRIMAL = 1e-9       # accepted violation of a row, relative to max(1, limit)
STALL_ROUNDS = 4        # full steps without progress that count as the noise floor
REG_REL = 1e-9          # dual regularisation of the Schur system, relative to pd ...
REG_DIAG = 1e-12        # ... and relative to the largest |R_a|^2, whichever is larger (1e-11 already biases the answer
                        # beyond the KKT check; 1e-12 halves the rounds of the worst horizon-24 instance)
TANGENT_MIN = 1e-7      # a disc whose multiplier is below TANGENT_MIN max(1, |q|_inf) gets no curvature row


def effective_pdiag(pd_user, reg_rel, qnorm, ubmax, horizon, has_prox=False):
    """acn_qp_tiled.hpp::effective_pdiag (the Tikhonov floor of LP-like problems)"""
    if has_prox or not ubmax > 0 or pd_user * ubmax > 1e-6 * qnorm:
        return pd_user
    return max(pd_user, reg_rel * qnorm / (ubmax * max(horizon, 1)))


def cholesky_solve(S, rhs, reg):
    """The kernel's solve: S + reg I = L L' (right-looking, in place), two triangular solves.  Tight rows that depend on
    each other (the degenerate vertices these instances sit on) make S singular; the small diagonal `reg` = pd x 1e-9 --
    OSQP's polish does the same -- turns that into the least-norm multiplier instead of dropping rows: dropped rows
    would leave their residuals uncorrected and the iteration stalls short of the optimum (measured).  Returns None when
    a pivot is not positive (the working set is inconsistent: the polish gives up)."""
    m = len(rhs)
    L = S + reg * np.eye(m)
    for k in range(m):
        piv = L[k, k]
        if not piv > 0.0:
            return None
        d = np.sqrt(piv)
        L[k, k] = d
        L[k + 1:, k] /= d
        L[k + 1:, k + 1:] -= np.outer(L[k + 1:, k], L[k + 1:, k])
    y = rhs.copy()
    for k in range(m):          # forward
        y[k] /= L[k, k]
        y[k + 1:] -= L[k + 1:, k] * y[k]
    for k in range(m - 1, -1, -1):   # backward
        y[k] /= L[k, k]
        y[:k] -= L[k, :k] * y[k]
    return y


def ldl_factor(A):
    """In-place unit-lower L D L' of a symmetric matrix (no pivoting); returns (L with unit diagonal implied, D) or None
    when a pivot is not positive.  L[r][k] is stored UNSCALED as the kernel stores it: A[r][k] after the updates of the
    columns before k; the scaled factor is A[r][k] / D[k]."""
    n = len(A)
    A = A.copy()
    for k in range(n):
        piv = A[k, k]
        if not piv > 0.0:
            return None
        l = A[k + 1:, k] / piv
        A[k + 1:, k + 1:] -= np.outer(l, A[k + 1:, k])
    return A


def ldl_solve(F, b):
    n = len(b)
    D = np.diag(F).copy()
    z = b.copy()
    for k in range(n):                      # L z = b
        z[k + 1:] -= F[k + 1:, k] / D[k] * z[k]
    z = z / D
    for k in range(n - 1, -1, -1):          # L' x = D^-1 z
        z[:k] -= F[k, :k] / D[:k] * z[k]
    return z


def structured_solve(R, Rm, diag, rhs, reg, sessions, on, nfree, free, T):
    """(R P R' + diag + reg I) lam = rhs without forming the matrix (the kernel's phases 5-6).  With the rows ordered by
    period, R R' + diag + reg I = B is BLOCK DIAGONAL (a row lives on one period's variables: blocks of at most two rows
    per site row), and the session projector takes a low-rank term away:  R P R' = R R' - V V',  V[:, s] = R 1_s / sqrt(n_s)
    (one column per tight session with free variables; its entry in row a is R_a at the session's EVSE when the row's
    period lies in the window and that variable is free).  Woodbury:
        lam = y + B^-1 V w,   y = B^-1 rhs,   (I - V' B^-1 V) w = V' y
    -- per-period L D L' factors of <= 34 x 34 and one of the size of the tight sessions (<= 64), instead of a dense
    factorisation of the size of all tight site rows (<= 256).  Returns None on a non-positive pivot."""
    m = len(R)
    t_of = np.array([t for (_, t, _, _, _) in R])
    blocks = [np.flatnonzero(t_of == t) for t in range(T)]
    Fs = {}
    for t, idx in enumerate(blocks):
        if len(idx) == 0:
            continue
        Rt = Rm[idx][:, :, t]                                  # (m_t, N)
        F = ldl_factor(Rt @ Rt.T + np.diag(diag[idx] + reg))
        if F is None:
            return None
        Fs[t] = F

    def b_solve(v):                                            # B^-1 v, block by block
        out = np.zeros(m)
        for t, idx in enumerate(blocks):
            if len(idx):
                out[idx] = ldl_solve(Fs[t], v[idx])
        return out

    act = [s for s in range(len(sessions)) if on[s]]
    V = np.zeros((m, len(act)))
    for c, s in enumerate(act):
        i, o, L = sessions[s][0], sessions[s][1], sessions[s][2]
        for a in range(m):
            t = t_of[a]
            if o <= t < o + L and free[i, t]:
                V[a, c] = Rm[a, i, t] / np.sqrt(nfree[s])
    y = b_solve(rhs)
    if not act:
        return y
    BV = np.column_stack([b_solve(V[:, c]) for c in range(len(act))])
    C = np.eye(len(act)) - V.T @ BV
    FC = ldl_factor(C)
    if FC is None:
        return None
    w = ldl_solve(FC, V.T @ y)
    return y + BV @ w


def polish(lb, ub, q, pd, sessions, eq, G, M, cone_soc, limits, peak, x0, y0, verbose=False):
    """lb, ub, q, x0: (N, T); sessions: [(i, off, len, cap)]; G: (Mg, N) ABI rows ([C cos; C sin] (+ ones row) for SOC,
    |C| (+ ones row) for LINEAR); peak: (T,) or None (the all-ones row is then the last row of G); y0: (Mg, T) site-row
    multipliers of the ADMM iterate (ABI units).  Returns (x, info) -- info["ok"] False when the polish gives up."""
    N, T = x0.shape
    ub = np.maximum(ub, lb)
    x = np.minimum(np.maximum(x0, lb), ub)
    qn = max(1.0, float(np.abs(q).max()))
    has_peak = peak is not None
    # ---- site rows as (kind, j): 'box' g_j' x <= lim, 'disc' (j, j + M), 'peak'
    rows = [("disc", j) for j in range(M)] if cone_soc else [("box", j) for j in range(M)]
    if has_peak:
        rows.append(("peak", G.shape[0] - 1))
    nrow = len(rows)

    def row_value(kind, j, xt):   # (value, limit-independent pieces)
        if kind == "disc":
            u0, u1 = G[j] @ xt, G[j + M] @ xt
            return float(np.hypot(u0, u1)), (u0, u1)
        return float(G[j] @ xt), None

    def row_limit(kind, j, t):
        return float(peak[t]) if kind == "peak" else float(limits[j])

    # ---- first working set
    at_lb = x <= lb + TOL_BOUND
    at_ub = (x >= ub - TOL_BOUND) & ~at_lb
    fixed = ub - lb <= TOL_BOUND
    s_act = np.array([eq or x[i, o:o + L].sum() >= cap - TOL_BOUND * max(1.0, abs(cap)) for (i, o, L, cap, *_k) in sessions], bool)
    r_act = np.zeros((nrow, T), bool)
    nu = np.zeros((nrow, T))
    for r, (kind, j) in enumerate(rows):
        mag = np.hypot(y0[j], y0[j + M]) if kind == "disc" else y0[j]
        fin = np.array([np.isfinite(row_limit(kind, j, t)) for t in range(T)])
        r_act[r] = (mag > TOL_ROW * qn) & fin
        nu[r] = np.where(r_act[r], mag, 0.0)
    mu = np.zeros(len(sessions))
    info = dict(ok=False, rounds=0, rows=0, why="rounds")
    stall, best_step = 0, np.inf
    for rnd in range(MAX_ROUNDS):
        free = ~(at_lb | at_ub)
        # ---- rows of the Schur system, PERIOD-major: the normal row of a tight site row, then its tangent
        R = []       # (r, t, c0, c1, j) : R_a[i] = c0 G[j][i] + c1 G[j + M][i] on period t
        c_A, diag = [], []
        for t in range(T):
            for r, (kind, j) in enumerate(rows):
                if not r_act[r, t]:
                    continue
                val, u = row_value(kind, j, x[:, t])
                lim = row_limit(kind, j, t)
                if kind == "disc":
                    if not val > 1e-12:
                        continue
                    n0, n1 = u[0] / val, u[1] / val
                    R.append((r, t, n0, n1, j)); c_A.append(lim - val); diag.append(0.0)
                    if nu[r, t] > TANGENT_MIN * qn:
                        R.append((r, t, -n1, n0, j)); c_A.append(0.0); diag.append(pd * val / nu[r, t])
                else:
                    R.append((r, t, 1.0, 0.0, j)); c_A.append(lim - val); diag.append(0.0)
        m = len(R)
        if m > MAX_ROWS:
            info["why"] = "rows"
            return x, info
        Rm = np.zeros((m, N, T))
        for a, (r, t, c0, c1, j) in enumerate(R):
            Rm[a, :, t] = c0 * G[j] + (c1 * G[j + M] if c1 != 0.0 else 0.0)
        Rm *= free[None]
        # ---- session projector pieces
        g = np.where(free, pd * x + q, 0.0)
        nfree = np.array([free[i, o:o + L].sum() for (i, o, L, cap, *_k) in sessions])
        on = s_act & (nfree > 0)
        c_E = np.array([cap - x[i, o:o + L].sum() for (i, o, L, cap, *_k) in sessions])

        def project(v):   # P v on the free variables
            out = v.copy()
            for s, (i, o, L, cap, *_k) in enumerate(sessions):
                if on[s]:
                    w = free[i, o:o + L]
                    out[i, o:o + L] -= np.where(w, v[i, o:o + L][w].sum() / nfree[s], 0.0)
            return out

        e = np.zeros((N, T))
        for s, (i, o, L, cap, *_k) in enumerate(sessions):
            if on[s]:
                e[i, o:o + L] += np.where(free[i, o:o + L], c_E[s] / nfree[s], 0.0)
        vfree = -project(g) + pd * e
        rhs = Rm.reshape(m, -1) @ vfree.reshape(-1) - pd * np.array(c_A) if m else np.zeros(0)
        # (a PROXIMAL form, (S + reg I) lam = rhs + reg lam_prev, would carry no bias and allow a larger reg -- but it keeps the
        #  null-space component of the previous multipliers instead of the least-norm one, and on these degenerate vertices the
        #  wandering multipliers then trip the sign test: every stalled fixture failed with it)
        dmax = float(np.max(np.sum(Rm.reshape(m, -1) ** 2, axis=1))) if m else 0.0
        reg = max(REG_REL * pd, REG_DIAG * dmax)
        lam = structured_solve(R, Rm, np.array(diag), rhs, reg, sessions, on, nfree, free, T) if m else np.zeros(0)
        if lam is None:
            info["why"] = "pivot"
            return x, info
        v = g + np.tensordot(lam, Rm, axes=(0, 0)) if m else g.copy()
        dx = np.where(free, -project(v) / pd + e, 0.0)
        for s, (i, o, L, cap, *_k) in enumerate(sessions):
            mu[s] = -(pd * c_E[s] + v[i, o:o + L][free[i, o:o + L]].sum()) / nfree[s] if on[s] else 0.0
        nu_new = np.zeros_like(nu)
        for a, (r, t, c0, c1, j) in enumerate(R):
            if diag[a] == 0.0:
                nu_new[r, t] = lam[a]
        # ---- ratio test against everything outside the working set; ties: the smaller code (the kernel's reduction)
        def code(kind, p0, p1):
            return ({"lb": 0, "ub": 1, "s": 2, "r": 3}[kind] << 24) | (p0 << 8) | p1

        cands = []
        for i in range(N):
            for t in range(T):
                if not free[i, t]:
                    continue
                d = dx[i, t]
                if d < -1e-14:
                    cands.append((max((lb[i, t] - x[i, t]) / d, 0.0), code("lb", i, t), ("lb", i, t)))
                elif d > 1e-14:
                    cands.append((max((ub[i, t] - x[i, t]) / d, 0.0), code("ub", i, t), ("ub", i, t)))
        for s, (i, o, L, cap, *_k) in enumerate(sessions):
            if s_act[s]:
                continue
            de = dx[i, o:o + L].sum()
            if de > 1e-14:
                cands.append((max(c_E[s] / de, 0.0), code("s", i, _k[0] if _k else 0), ("s", s, 0)))
        for r, (kind, j) in enumerate(rows):
            for t in range(T):
                if r_act[r, t]:
                    continue
                lim = row_limit(kind, j, t)
                if not np.isfinite(lim):
                    continue
                if kind == "disc":
                    u = np.array([G[j] @ x[:, t], G[j + M] @ x[:, t]])
                    du = np.array([G[j] @ dx[:, t], G[j + M] @ dx[:, t]])
                    aa, bb, cc = du @ du, 2.0 * (u @ du), u @ u - lim * lim
                    if aa > 1e-28 and (bb > 0 or cc > 0):
                        dsc = bb * bb - 4 * aa * cc
                        if dsc >= 0:
                            cands.append((max((-bb + np.sqrt(dsc)) / (2 * aa), 0.0), code("r", r, t), ("r", r, t)))
                else:
                    du = G[j] @ dx[:, t]
                    if du > 1e-14:
                        cands.append((max((lim - G[j] @ x[:, t]) / du, 0.0), code("r", r, t), ("r", r, t)))
        step = float(np.abs(dx).max())
        alpha, block = 1.0, None
        cands = [c for c in cands if c[0] < 1.0]
        if cands:
            alpha, _, block = min(cands, key=lambda c: (c[0], c[1]))
        x = x + alpha * dx
        nu = nu_new
        changed = False
        if block is not None:
            kd, p0, p1 = block
            if kd == "lb":
                at_lb[p0, p1] = True; x[p0, p1] = lb[p0, p1]
            elif kd == "ub":
                at_ub[p0, p1] = True; x[p0, p1] = ub[p0, p1]
            elif kd == "s":
                s_act[p0] = True
            else:
                r_act[p0, p1] = True
            changed = True
        # noise floor: on the degenerate instances the regularised solve leaves 1e-5 A of noise in dx now and then; full
        # steps that have stopped shrinking (STALL_ROUNDS of them within 2x of the smallest so far on this working set) go to
        # the multiplier test and the KKT check like a converged one -- the check decides, and a failure there ends the polish
        # instead of burning the round limit
        if block is None:
            stall = stall + 1 if step >= 0.5 * best_step else 0
            best_step = min(best_step, step)
        else:
            stall, best_step = 0, np.inf
        floor_hit = block is None and stall >= STALL_ROUNDS and step <= 1e-3
        conv = block is None and (step <= TOL_STEP * max(1.0, float(np.abs(x).max())) or floor_hit)
        if verbose:
            print(f"  round {rnd:2d} free {int(free.sum()):4d} rows {m:3d} step {step:.2e} alpha {alpha:.3f} block {block}")
        if not changed and conv:
            # ---- multipliers of the whole working set; the most negative one leaves
            grad = pd * x + q
            for s, (i, o, L, cap, *_k) in enumerate(sessions):
                grad[i, o:o + L] += mu[s]
            for r, (kind, j) in enumerate(rows):
                for t in range(T):
                    if nu[r, t] == 0.0:
                        continue
                    if kind == "disc":
                        val, u = row_value(kind, j, x[:, t])
                        grad[:, t] += nu[r, t] * ((u[0] / val) * G[j] + (u[1] / val) * G[j + M])
                    else:
                        grad[:, t] += nu[r, t] * G[j]
            rel = []
            for i in range(N):
                for t in range(T):
                    if fixed[i, t]:
                        continue
                    if at_lb[i, t]:
                        rel.append((grad[i, t], code("lb", i, t), ("lb", i, t)))
                    if at_ub[i, t]:
                        rel.append((-grad[i, t], code("ub", i, t), ("ub", i, t)))
            if not eq:
                for s, (i, o, L, cap, *_k) in enumerate(sessions):
                    if s_act[s]:
                        rel.append((mu[s], code("s", i, _k[0] if _k else 0), ("s", s, 0)))
            for r in range(nrow):
                for t in range(T):
                    if r_act[r, t]:
                        rel.append((nu[r, t], code("r", r, t), ("r", r, t)))
            rel = [c for c in rel if c[0] < -TOL_DUAL * qn]
            who = min(rel, key=lambda c: (c[0], c[1]))[2] if rel else None
            if who is None:
                # ---- KKT on the full problem
                stat = float(np.abs(grad[~(at_lb | at_ub)]).max()) if (~(at_lb | at_ub)).any() else 0.0
                pv = 0.0
                for s, (i, o, L, cap, *_k) in enumerate(sessions):
                    d = x[i, o:o + L].sum() - cap
                    pv = max(pv, (abs(d) if eq else d) / max(1.0, abs(cap)))
                for r, (kind, j) in enumerate(rows):
                    for t in range(T):
                        lim = row_limit(kind, j, t)
                        if np.isfinite(lim):
                            pv = max(pv, (row_value(kind, j, x[:, t])[0] - lim) / max(1.0, lim))
                ok = stat <= 1e-8 * qn and pv <= 10 * TOL_PRIMAL
                info.update(ok=bool(ok), rounds=rnd + 1, rows=m, why="kkt" if ok else "verify", stat=stat, primal=pv)
                # multipliers in ABI units (a disc's pair: nu times its normal)
                y = np.zeros_like(y0)
                for r, (kind, j) in enumerate(rows):
                    for t in range(T):
                        if nu[r, t] == 0.0:
                            continue
                        if kind == "disc":
                            val, u = row_value(kind, j, x[:, t])
                            y[j, t], y[j + M, t] = nu[r, t] * u[0] / val, nu[r, t] * u[1] / val
                        else:
                            y[j, t] = nu[r, t]
                info["y"] = y
                return x, info
            stall, best_step = 0, np.inf
            kd, p0, p1 = who
            if kd == "lb":
                at_lb[p0, p1] = False
            elif kd == "ub":
                at_ub[p0, p1] = False
            elif kd == "s":
                s_act[p0] = False
            else:
                r_act[p0, p1] = False; nu[p0, p1] = 0.0
    info["rounds"] = MAX_ROUNDS
    return x, info


def polish_batch_problem(batch, b, x0, y0, reg_rel=0.06, verbose=False):
    """Problem ``b`` of a builder.ProblemBatch, from the ADMM iterate (x0 (N, Tm), y0 (Mg, Tm) in ABI units)."""
    site = batch.site
    T = int(batch.T[b])
    lb, ub, q = batch.lb[b][:, :T], np.maximum(batch.ub[b][:, :T], batch.lb[b][:, :T]), batch.q[b][:, :T]
    sessions = []
    for k in range(batch.K):
        for i in range(site.N):
            L = int(batch.s_len[b, k, i])
            if L:
                sessions.append((i, int(batch.s_off[b, k, i]), L, float(batch.s_cap[b, k, i]), k))
    pd = effective_pdiag(float(batch.pdiag[b]), reg_rel, float(np.abs(q).max()), float(ub.max()), T)
    peak = None if batch.peak is None else np.asarray(batch.peak[b][:T], float)
    x, info = polish(lb, ub, q, pd, sessions, bool(batch.s_eq[b]), np.asarray(site.G, float), site.M, site.cone == 1,
                     np.asarray(site.limits, float), peak, np.asarray(x0, float)[:, :T], np.asarray(y0, float)[:, :T], verbose)
    return x, info
