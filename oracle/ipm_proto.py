"""TEST INFRASTRUCTURE — numpy prototype of the stage-structured primal-dual interior-point solve.

Development/cross-check tool for the algorithm that oracle/nmpc_oracle.c (CPU) and
nmpc_amd/csrc (HIP) implement. Never imported by the product path.

Algorithm: primal-dual interior point on the NLP of oracle/nlp_ref.py (the reference's
`nlpsol('ipopt')` call, C6:345-346,432), following the published IPOPT scheme
(Waechter & Biegler 2006): slacks on every inequality, monotone barrier update,
fraction-to-boundary, inertia correction by diagonal regularisation, backtracking line
search; the KKT system is solved by a Riccati sweep over the horizon.
"""
from __future__ import annotations

import numpy as np

from .nlp_ref import NLPConfig


CKPT_EVERY, REFACTOR_BACK = 5, 2      # NMPC_CKPT_EVERY, NMPC_REFACTOR_BACK of include/nmpc_constants.h (tests/test_oracle_solver.py checks the header)


class Opts:
    tol = 1e-8
    max_iter = 200
    mu_init = 0.5
    mu_min = 1e-9
    kappa_eps = 10.0
    kappa_mu = 0.2
    theta_mu = 1.5
    tau_min = 0.99
    bound_push = 1e-2
    exact_hessian = True
    reg_where = 'u'       # 'all': delta*I on x and u (IPOPT); 'u': controls only; 'thu': headings and controls
    reg_mode = 'global'   # 'global' (IPOPT alg. IC) or 'stage' (per-stage shift of Quu only)
    partial_refactor = True   # a rejected pivot resumes the sweep at a saved stage a few stages above (include/nmpc_constants.h) instead of at N-1
    ls_max = 30
    eta = 1e-4
    verbose = False


def _stage_ineq(cfg: NLPConfig, k, x, u):
    """values and gradient rows of the inequalities h>=0 attached to stage k.

    returns hx (values), Jx (n x nx) for x-rows and hu, Ju for u-rows."""
    nx, nu, N = cfg.nx, cfg.nu, cfg.N
    hx, Jx, hu, Ju = [], [], [], []
    if k < N:
        lb = np.tile(np.array([-cfg.v_max, -cfg.w_max]), cfg.m)
        for c in range(nu):
            e = np.zeros(nu); e[c] = 1.0
            hu.append(u[c] - lb[c]); Ju.append(e)
            hu.append(-lb[c] - u[c]); Ju.append(-e)
    if k >= 1:
        bx = np.tile(np.array([cfg.xy_max, cfg.xy_max, cfg.th_max]), cfg.m)
        for c in range(nx):
            if np.isfinite(bx[c]):
                e = np.zeros(nx); e[c] = 1.0
                hx.append(x[c] + bx[c]); Jx.append(e)
                hx.append(bx[c] - x[c]); Jx.append(-e)
    if 1 <= k <= N - 1:
        for (i, j) in cfg.pairs():
            dx = x[3 * i] - x[3 * j]; dy = x[3 * i + 1] - x[3 * j + 1]
            g = np.zeros(nx)
            g[3 * i] = 2 * dx; g[3 * i + 1] = 2 * dy; g[3 * j] = -2 * dx; g[3 * j + 1] = -2 * dy
            hx.append(dx * dx + dy * dy - cfg.dmin ** 2); Jx.append(g)
        for i in range(cfg.m):
            for (ox, oy, orad) in cfg.obstacles:
                dx = x[3 * i] - ox; dy = x[3 * i + 1] - oy
                rr = np.sqrt(dx * dx + dy * dy)
                g = np.zeros(nx); g[3 * i] = dx / rr; g[3 * i + 1] = dy / rr
                hx.append(rr - cfg.rob_dim - orad - cfg.margin); Jx.append(g)
    return (np.array(hx), np.array(Jx).reshape(len(hx), nx), np.array(hu), np.array(Ju).reshape(len(hu), nu))


def _ineq_hess_x(cfg, k, x, zx_pairs_obs):
    """- sum z_i * hess h_i for the nonlinear x-rows of stage k (pairs, obstacles)."""
    nx = cfg.nx
    W = np.zeros((nx, nx))
    if not (1 <= k <= cfg.N - 1):
        return W
    o = 0
    for (i, j) in cfg.pairs():
        z = zx_pairs_obs[o]; o += 1
        for d in (0, 1):
            a, b = 3 * i + d, 3 * j + d
            W[a, a] -= 2 * z; W[b, b] -= 2 * z; W[a, b] += 2 * z; W[b, a] += 2 * z
    for i in range(cfg.m):
        for (ox, oy, orad) in cfg.obstacles:
            z = zx_pairs_obs[o]; o += 1
            dx = x[3 * i] - ox; dy = x[3 * i + 1] - oy
            rr = np.sqrt(dx * dx + dy * dy)
            n = np.array([dx, dy]) / rr
            W[3 * i: 3 * i + 2, 3 * i: 3 * i + 2] -= z * (np.eye(2) - np.outer(n, n)) / rr
    return W


def _dyn(cfg, x, u):
    nx, nu, T = cfg.nx, cfg.nu, cfg.T
    th = x[2::3]; v = u[0::2]; w = u[1::2]
    f = np.empty(nx)
    f[0::3] = v * np.cos(th); f[1::3] = v * np.sin(th); f[2::3] = w
    A = np.eye(nx); B = np.zeros((nx, nu))
    for i in range(cfg.m):
        A[3 * i, 3 * i + 2] = -T * v[i] * np.sin(th[i])
        A[3 * i + 1, 3 * i + 2] = T * v[i] * np.cos(th[i])
        B[3 * i, 2 * i] = T * np.cos(th[i]); B[3 * i + 1, 2 * i] = T * np.sin(th[i]); B[3 * i + 2, 2 * i + 1] = T
    return x + T * f, A, B


def solve(cfg: NLPConfig, p, w0, opts: Opts = None):
    """returns dict(x=w*, f=obj, status, iters, kkt, lam, hist)."""
    o = opts or Opts()
    nx, nu, N, T = cfg.nx, cfg.nu, cfg.N, cfg.T
    p = np.asarray(p, float).reshape(-1)
    x0p, xs = p[:nx], p[nx:]
    w0 = np.asarray(w0, float).reshape(-1)
    X = w0[: nx * (N + 1)].reshape(N + 1, nx).copy()
    U = w0[nx * (N + 1):].reshape(N, nu).copy()
    X[0] = x0p
    qd = np.tile(np.asarray(cfg.q, float), cfg.m)
    rd = np.tile(np.asarray(cfg.r, float), cfg.m)
    n_pairobs = cfg.M + cfg.m * cfg.K

    # stage-0 pair/obstacle rows act on the pinned state: feasibility pre-check only
    status = 0
    for (i, j) in cfg.pairs():
        if (x0p[3 * i] - x0p[3 * j]) ** 2 + (x0p[3 * i + 1] - x0p[3 * j + 1]) ** 2 < cfg.dmin ** 2:
            status = 3
    for i in range(cfg.m):
        for (ox, oy, orad) in cfg.obstacles:
            if np.hypot(x0p[3 * i] - ox, x0p[3 * i + 1] - oy) - cfg.rob_dim - orad < cfg.margin:
                status = 3
    if status == 3:
        return dict(x=np.concatenate([X.reshape(-1), U.reshape(-1)]), f=np.nan, status=3, iters=0, kkt=np.inf)

    # push the start into the interior of the simple bounds (IPOPT bound_push/bound_frac)
    lbu = np.tile(np.array([-cfg.v_max, -cfg.w_max]), cfg.m); ubu = -lbu
    pu = np.minimum(o.bound_push * np.maximum(1.0, np.abs(lbu)), o.bound_push * (ubu - lbu))
    U = np.clip(U, lbu + pu, ubu - pu)
    bx = np.tile(np.array([cfg.xy_max, cfg.xy_max, cfg.th_max]), cfg.m)
    fin = np.isfinite(bx)
    px = np.where(fin, np.minimum(o.bound_push * np.maximum(1.0, bx), o.bound_push * 2 * np.where(fin, bx, 1.0)), 0.0)
    for k in range(1, N + 1):
        X[k] = np.where(fin, np.clip(X[k], -bx + px, bx - px), X[k])

    mu = o.mu_init
    # slacks and duals
    Sx, Zx, Su, Zu = [], [], [], []
    for k in range(N + 1):
        hx, _, hu, _ = _stage_ineq(cfg, k, X[k], U[k] if k < N else None)
        sx = np.maximum(hx, o.bound_push); su = np.maximum(hu, 1e-12)
        Sx.append(sx); Su.append(su); Zx.append(mu / sx if sx.size else sx.copy()); Zu.append(mu / su if su.size else su.copy())
    lam = np.zeros((N + 1, nx))     # lam[k+1] pairs with defect c_k
    delta_last = 0.0
    need_shift = False
    n_tiny = 0
    n_restart = 0
    nu_pen = 1.0
    mhist = []; mh_key = None
    hist = []
    sweeps_total = [0]

    def eval_all(X, U):
        C = np.zeros((N, nx)); Hx = []; Hu = []
        for k in range(N):
            xn, _, _ = _dyn(cfg, X[k], U[k])
            C[k] = X[k + 1] - xn
        for k in range(N + 1):
            hx, _, hu, _ = _stage_ineq(cfg, k, X[k], U[k] if k < N else None)
            Hx.append(hx); Hu.append(hu)
        fval = float(np.sum(qd * (X[:N] - xs) ** 2) + np.sum(rd * U * U))
        return fval, C, Hx, Hu

    def barrier_phi(fval, Sx, Su, mu):
        return fval - mu * (sum(np.sum(np.log(s)) for s in Sx) + sum(np.sum(np.log(s)) for s in Su))

    def infeas(C, Hx, Hu, Sx, Su):
        t = np.sum(np.abs(C))
        for k in range(N + 1):
            t += np.sum(np.abs(Hx[k] - Sx[k])) + np.sum(np.abs(Hu[k] - Su[k]))
        return t

    it = 0
    kkt = np.inf
    while True:
        fval, C, Hx, Hu = eval_all(X, U)
        # --- stationarity residuals
        rdx = np.zeros((N + 1, nx)); rdu = np.zeros((N, nu))
        AB = [None] * N; JX = [None] * (N + 1); JU = [None] * (N + 1)
        for k in range(N + 1):
            _, Jx, _, Ju = _stage_ineq(cfg, k, X[k], U[k] if k < N else None)
            JX[k], JU[k] = Jx, Ju
            if k < N:
                _, A, B = _dyn(cfg, X[k], U[k]); AB[k] = (A, B)
        for k in range(1, N + 1):
            r = lam[k].copy()
            if k < N:
                r += 2 * qd * (X[k] - xs) - AB[k][0].T @ lam[k + 1]
            if Zx[k].size:
                r -= JX[k].T @ Zx[k]
            rdx[k] = r
        for k in range(N):
            rdu[k] = 2 * rd * U[k] - AB[k][1].T @ lam[k + 1] - JU[k].T @ Zu[k]
        e_d = max(np.max(np.abs(rdx)), np.max(np.abs(rdu)))
        e_c = np.max(np.abs(C))
        e_h = max([np.max(np.abs(Hx[k] - Sx[k])) if Sx[k].size else 0.0 for k in range(N + 1)] +
                  [np.max(np.abs(Hu[k] - Su[k])) if Su[k].size else 0.0 for k in range(N + 1)])
        def compl(mu_):
            return max([np.max(np.abs(Sx[k] * Zx[k] - mu_)) if Sx[k].size else 0.0 for k in range(N + 1)] +
                       [np.max(np.abs(Su[k] * Zu[k] - mu_)) if Su[k].size else 0.0 for k in range(N + 1)])
        zsum = sum(np.sum(z) for z in Zx) + sum(np.sum(z) for z in Zu)
        nz = sum(z.size for z in Zx) + sum(z.size for z in Zu)
        lsum = np.sum(np.abs(lam))
        smax = 100.0
        s_d = max(smax, (lsum + zsum) / (lam.size - nx + nz)) / smax
        s_c = max(smax, zsum / max(nz, 1)) / smax
        E0 = max(e_d / s_d, e_c, e_h, compl(0.0) / s_c)
        kkt = E0
        hist.append((it, fval, E0, mu, e_d, max(e_c, e_h)))
        if o.verbose:
            print("it %3d f %.8f E0 %.2e mu %.1e ed %.1e ec %.1e delta %.1e" % (it, fval, E0, mu, e_d, max(e_c, e_h), delta_last))
        if E0 <= o.tol:
            status = 0; break
        if it >= o.max_iter:
            status = 1; break
        while mu > o.mu_min and max(e_d / s_d, e_c, e_h, compl(mu) / s_c) <= o.kappa_eps * mu:
            mu = max(o.mu_min, min(o.kappa_mu * mu, mu ** o.theta_mu))
        tau = max(o.tau_min, 1.0 - mu)

        # --- assemble stage blocks of the condensed KKT system
        Hxx = [None] * (N + 1); Hux = [None] * N; Huu = [None] * N; gx = [None] * (N + 1); gu = [None] * N
        for k in range(N + 1):
            sig = Zx[k] / Sx[k] if Sx[k].size else Zx[k]
            H = np.zeros((nx, nx)); g = np.zeros(nx)
            if 1 <= k < N:
                H += np.diag(2 * qd); g += 2 * qd * (X[k] - xs)
            if k >= 1 and Sx[k].size:
                H += JX[k].T @ (sig[:, None] * JX[k])
                g -= JX[k].T @ (mu / Sx[k] - sig * (Hx[k] - Sx[k]))
                if o.exact_hessian and n_pairobs and 1 <= k <= N - 1:
                    H += _ineq_hess_x(cfg, k, X[k], Zx[k][-n_pairobs:])
            Hxx[k], gx[k] = H, g
            if k < N:
                sg = Zu[k] / Su[k]
                Hu_ = np.diag(2 * rd) + JU[k].T @ (sg[:, None] * JU[k])
                gu_ = 2 * rd * U[k] - JU[k].T @ (mu / Su[k] - sg * (Hu[k] - Su[k]))
                Hux_ = np.zeros((nu, nx))
                if o.exact_hessian:
                    for i in range(cfg.m):
                        th, v = X[k][3 * i + 2], U[k][2 * i]
                        lx, ly = lam[k + 1][3 * i], lam[k + 1][3 * i + 1]
                        if k >= 1:
                            Hxx[k][3 * i + 2, 3 * i + 2] += T * v * (lx * np.cos(th) + ly * np.sin(th))
                            Hux_[2 * i, 3 * i + 2] += T * (lx * np.sin(th) - ly * np.cos(th))
                Huu[k], gu[k], Hux[k] = Hu_, gu_, Hux_

        # --- Riccati with inertia correction
        # first trial: delta = 0, except right after an iteration that needed a shift (then a quarter of that shift directly)
        delta = max(1e-20, 0.25 * delta_last) if need_shift else 0.0
        ntry = 0
        nsweep = 0
        kst = N - 1; kf = 0; ckpt = {}; Ks_prev = ks_prev = None
        while True:
            nsweep += 1
            ok = True
            if o.reg_where == 'all': Dx = delta * np.eye(nx)
            elif o.reg_where == 'u': Dx = np.zeros((nx, nx))
            else: Dx = delta * np.diag(np.tile([0.0, 0.0, 1.0], cfg.m))
            P = Hxx[N] + Dx; pv = gx[N].copy()
            Ks = [None] * N; ks = [None] * N
            if o.partial_refactor and kst < N - 1:      # resume from the cost-to-go saved on entry to stage kst (stages above keep their gains)
                P, pv = ckpt[kst][0].copy(), ckpt[kst][1].copy()
                Ks[kst + 1:] = Ks_prev[kst + 1:]; ks[kst + 1:] = ks_prev[kst + 1:]
            for k in range(kst if o.partial_refactor else N - 1, -1, -1):
                if o.partial_refactor and (N - 1 - k) % CKPT_EVERY == 0:
                    ckpt[k] = (P.copy(), pv.copy())
                A, B = AB[k]
                b = -C[k]
                Pb = pv + P @ b
                Quu = Huu[k] + delta * np.eye(nu) + B.T @ P @ B
                Qux = Hux[k] + B.T @ P @ A
                qu = gu[k] + B.T @ Pb
                if o.reg_mode == 'stage':
                    dk = 0.0
                    while True:
                        try:
                            Qt = Quu + dk * np.eye(nu)
                            L = np.linalg.cholesky(Qt)
                            if np.any(np.diag(L) ** 2 < 1e-9 * np.abs(np.diag(Qt))):
                                raise np.linalg.LinAlgError
                            break
                        except np.linalg.LinAlgError:
                            dk = 1e-4 if dk == 0.0 else dk * 10.0
                            ntry += 1
                    Quu = Qt
                else:
                  try:
                    L = np.linalg.cholesky(Quu)
                    if np.any(np.diag(L) ** 2 < 1e-9 * np.abs(np.diag(Quu))):
                        raise np.linalg.LinAlgError
                  except np.linalg.LinAlgError:
                    ok = False; kf = k; break
                Kk = -np.linalg.solve(Quu, Qux); kk = -np.linalg.solve(Quu, qu)
                Ks[k], ks[k] = Kk, kk
                if k >= 1:
                    Qxx = Hxx[k] + Dx + A.T @ P @ A
                    qx = gx[k] + A.T @ Pb
                    P = Qxx + Qux.T @ Kk; P = 0.5 * (P + P.T)
                    pv = qx + Qux.T @ kk
            if ok:
                break
            ntry += 1
            if delta == 0.0:
                delta = 1e-4 if delta_last == 0.0 else max(1e-20, delta_last / 3.0)
            else:
                delta *= 100.0 if delta_last == 0.0 else 8.0
            if delta > 1e20:
                break
            # partial re-factorisation (include/nmpc_constants.h: NMPC_RESUME_STAGE): the nearest saved stage at or above kf + REFACTOR_BACK
            kst = min(kf + REFACTOR_BACK, N - 1); kst += (N - 1 - kst) % CKPT_EVERY
            Ks_prev, ks_prev = Ks, ks
        if not ok:
            status = 2; break
        if delta > 0:
            delta_last = delta
        need_shift = delta > 0.0 and (ntry > 0 or delta > 1e-6)
        sweeps_total[0] += nsweep
        # forward
        dX = np.zeros((N + 1, nx)); dU = np.zeros((N, nu))
        for k in range(N):
            dU[k] = Ks[k] @ dX[k] + ks[k]
            dX[k + 1] = AB[k][0] @ dX[k] + AB[k][1] @ dU[k] - C[k]
        # multipliers of the QP (= lam + dlam), matrix-free backward recursion
        lamn = np.zeros_like(lam)
        lamn[N] = -((Hxx[N] + Dx) @ dX[N] + gx[N])
        for k in range(N - 1, 0, -1):
            lamn[k] = AB[k][0].T @ lamn[k + 1] - ((Hxx[k] + Dx) @ dX[k] + Hux[k].T @ dU[k] + gx[k])
        # slack / dual steps
        dSx, dZx, dSu, dZu = [], [], [], []
        for k in range(N + 1):
            if Sx[k].size:
                ds = JX[k] @ dX[k] + (Hx[k] - Sx[k]); dz = (mu - Sx[k] * Zx[k] - Zx[k] * ds) / Sx[k]
            else:
                ds = np.zeros(0); dz = np.zeros(0)
            dSx.append(ds); dZx.append(dz)
            if Su[k].size:
                ds = JU[k] @ dU[k] + (Hu[k] - Su[k]); dz = (mu - Su[k] * Zu[k] - Zu[k] * ds) / Su[k]
            else:
                ds = np.zeros(0); dz = np.zeros(0)
            dSu.append(ds); dZu.append(dz)
        def maxstep(v, dv):
            a = 1.0
            for vk, dk in zip(v, dv):
                neg = dk < 0
                if np.any(neg):
                    a = min(a, np.min(-tau * vk[neg] / dk[neg]))
            return a
        a_p = min(maxstep(Sx, dSx), maxstep(Su, dSu))
        a_d = min(maxstep(Zx, dZx), maxstep(Zu, dZu))

        # --- l1 merit backtracking
        th0 = infeas(C, Hx, Hu, Sx, Su)
        phi0 = barrier_phi(fval, Sx, Su, mu)
        # directional derivative of the barrier objective
        dphi = 0.0
        for k in range(N + 1):
            if 1 <= k < N:
                dphi += np.dot(2 * qd * (X[k] - xs), dX[k])
            if k < N:
                dphi += np.dot(2 * rd * U[k], dU[k])
            dphi -= mu * (np.sum(dSx[k] / Sx[k]) + np.sum(dSu[k] / Su[k]))
        # penalty parameter (Nocedal-Wright 18.36 style with rho = 0.1)
        if th0 > 0:
            nu_trial = dphi / ((1 - 0.1) * th0)
            nu_pen = max(1.0, 0.5 * nu_pen)
            if nu_pen < nu_trial:
                nu_pen = nu_trial + 1.0
        D = dphi - nu_pen * th0
        if mh_key != (mu, nu_pen):
            mhist = []; mh_key = (mu, nu_pen)
        m0 = phi0 + nu_pen * th0
        mref = max([m0] + mhist[:3])
        mhist = [m0] + mhist[:2]
        alpha = a_p
        accepted = False
        for ls in range(o.ls_max):
            Xn = X + alpha * dX; Un = U + alpha * dU
            Sxn = [s + alpha * d for s, d in zip(Sx, dSx)]; Sun = [s + alpha * d for s, d in zip(Su, dSu)]
            fn, Cn, Hxn, Hun = eval_all(Xn, Un)
            thn = infeas(Cn, Hxn, Hun, Sxn, Sun)
            phin = barrier_phi(fn, Sxn, Sun, mu)
            if (phin + nu_pen * thn) <= mref + o.eta * alpha * D + 1e-13 * abs(phi0):
                accepted = True; break
            alpha *= 0.5
        if not accepted:
            if o.verbose:
                print("  line search failed; taking tiny step")
        if o.verbose:
            print("     alpha %.3g a_p %.3g a_d %.3g nu %.3g dphi %.3g th0 %.3g ntry %d" % (alpha, a_p, a_d, nu_pen, dphi, th0, ntry))
        X, U, Sx, Su = Xn, Un, Sxn, Sun
        a_dual = min(a_d, alpha)      # the duals never step further than the primal variables actually moved
        for k in range(N + 1):
            Zx[k] = Zx[k] + a_dual * dZx[k]; Zu[k] = Zu[k] + a_dual * dZu[k]
            # IPOPT eq. (16) safeguard
            if Sx[k].size:
                Zx[k] = np.clip(Zx[k], mu / (1e10 * Sx[k]), 1e10 * mu / Sx[k])
            if Su[k].size:
                Zu[k] = np.clip(Zu[k], mu / (1e10 * Su[k]), 1e10 * mu / Su[k])
        lam = lam + alpha * (lamn - lam)
        it += 1
        n_tiny = n_tiny + 1 if alpha < 1e-10 else 0
        if n_tiny >= 5:
            if n_restart >= 3:
                status = 4; break      # stalled: converging to an infeasible stationary point
            # barrier restart from the current primal point (restoration in miniature, see nmpc_oracle.c)
            n_restart += 1; n_tiny = 0
            mu = max(mu, o.mu_init)
            U = np.clip(U, lbu + pu, ubu - pu)      # back strictly inside the simple bounds, as at the start
            for k in range(1, N + 1):
                X[k] = np.where(fin, np.clip(X[k], -bx + px, bx - px), X[k])
            Sx, Zx, Su, Zu = [], [], [], []
            for k in range(N + 1):
                hx, _, hu, _ = _stage_ineq(cfg, k, X[k], U[k] if k < N else None)
                sx = np.maximum(hx, o.bound_push); su = np.maximum(hu, 1e-12)
                Sx.append(sx); Su.append(su); Zx.append(mu / sx if sx.size else sx.copy()); Zu.append(mu / su if su.size else su.copy())
            lam = np.zeros((N + 1, nx))
            delta_last = 0.0; nu_pen = 1.0; need_shift = False; mhist = []; mh_key = None

    wout = np.concatenate([X.reshape(-1), U.reshape(-1)])
    fval = float(np.sum(qd * (X[:N] - xs) ** 2) + np.sum(rd * U * U))
    return dict(x=wout, f=fval, status=status, iters=it, kkt=kkt, lam=lam, hist=hist, sweeps=sweeps_total[0])
