"""TEST INFRASTRUCTURE — numpy restatement of the reference's multiple-shooting NLP.

This file is part of the parity oracle. It is imported only by tests/, by
__graft_entry__.smoke() and by the golden-vector generator; the product path
(nmpc_amd) never imports it.

PARITY UNPINNED against CasADi/IPOPT: the reference's arithmetic lives in
`casadi` and IPOPT (both un-vendored, unpinned, not installed here; see
SURVEY.md §8c). What pins this file instead are the solver-independent
known answers of SURVEY.md §8c items 1-5 (tests/test_oracle_nlp.py) and the
scipy-SLSQP golden triples in tests/golden/.

Each function cites the block of the reference it restates.
  C1 = AllScripts/centralized_one_robots_implementation.py
  C2 = AllScripts/centralized_two_robots_implementation.py
  C6 = AllScripts/centralized_six_robots_implementation.py
  C10 = AllScripts/mpc_online_casadi_tb3_ten_multi_centralized_collision_avoidance.py
  O3 = AllScripts/third_scenario_mpc_obstacle_avoidance.py
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

INF = np.inf


@dataclass
class NLPConfig:
    """All literals of one reference script (C6:197-205, 252-266, 349-352)."""
    m: int = 2                       # robots
    N: int = 20                      # horizon
    T: float = 0.05                  # sample time [s]
    dmin: float = 0.15               # pair rows are bounded below by dmin**2 (C6:349)
    q: Tuple[float, float, float] = (1.0, 5.0, 0.1)    # Q diag per robot (C6:252-257)
    r: Tuple[float, float] = (0.5, 0.05)               # R diag per robot (C6:259-264)
    v_max: float = 0.22
    w_max: float = 2.84
    xy_max: float = 10.0             # |x|,|y| <= 10 (C6:351-352)
    th_max: float = INF              # theta unbounded in multi-robot files; 2*pi in O3:176
    obstacles: List[Tuple[float, float, float]] = field(default_factory=list)  # (ox, oy, obs_r)
    rob_dim: float = 0.2             # O3:58
    margin: float = 0.1              # lbg of obstacle rows (O3:175)
    pad_value: float = 3.5           # constant rows of the initial block (C6:278)
    pad_rows: bool = True            # C6:278 pads; the 1-robot files (C1:108, O3:122) do not
    pair_rows: bool = True           # False: multi-robot NLP without collision rows (AS/mpc_online_casadi_tb3_multi_centralized.py:115-148)

    @property
    def nx(self): return 3 * self.m
    @property
    def nu(self): return 2 * self.m
    @property
    def M(self): return self.m * (self.m - 1) // 2 if self.pair_rows else 0
    @property
    def K(self): return len(self.obstacles)
    @property
    def n_var(self): return self.nx * (self.N + 1) + self.nu * self.N
    @property
    def rows0(self):
        """rows of the initial-condition block (a5)."""
        return self.nx + (self.M if self.pad_rows else 0)
    @property
    def rows_k(self):
        """rows of each stage block: defect, pair rows, obstacle rows (a2,a4,a12)."""
        return self.nx + self.M + self.m * self.K
    @property
    def n_g(self): return self.rows0 + self.rows_k * self.N

    def pairs(self):
        """lexicographic i<j order 12,13,...,1m,23,... (C6:288-306)."""
        return [(i, j) for i in range(self.m) for j in range(i + 1, self.m)] if self.pair_rows else []


# ----------------------------------------------------------------------------
# a1: unicycle right-hand side (C6:207-237)
def rhs(cfg: NLPConfig, x: np.ndarray, u: np.ndarray) -> np.ndarray:
    out = np.empty(cfg.nx)
    th = x[2::3]
    v = u[0::2]
    w = u[1::2]
    out[0::3] = v * np.cos(th)
    out[1::3] = v * np.sin(th)
    out[2::3] = w
    return out


# a6: packing w = [vec(X); vec(U)], column-major reshape of n_x x (N+1) (C6:339)
def unpack(cfg: NLPConfig, w: np.ndarray):
    w = np.asarray(w, dtype=np.float64).reshape(-1)
    assert w.size == cfg.n_var
    X = w[: cfg.nx * (cfg.N + 1)].reshape(cfg.N + 1, cfg.nx)
    U = w[cfg.nx * (cfg.N + 1):].reshape(cfg.N, cfg.nu)
    return X, U


def pack(cfg: NLPConfig, X: np.ndarray, U: np.ndarray) -> np.ndarray:
    return np.concatenate([np.asarray(X, float).reshape(-1), np.asarray(U, float).reshape(-1)])


# a3: stage cost (C6:314); no 1/2, no terminal term, X_N absent
def objective(cfg: NLPConfig, w, p) -> float:
    X, U = unpack(cfg, w)
    xs = np.asarray(p, float).reshape(-1)[cfg.nx:]
    qd = np.tile(np.asarray(cfg.q, float), cfg.m)
    rd = np.tile(np.asarray(cfg.r, float), cfg.m)
    e = X[: cfg.N] - xs
    return float(np.sum(e * e * qd) + np.sum(U * U * rd))


def grad_objective(cfg: NLPConfig, w, p) -> np.ndarray:
    X, U = unpack(cfg, w)
    xs = np.asarray(p, float).reshape(-1)[cfg.nx:]
    qd = np.tile(np.asarray(cfg.q, float), cfg.m)
    rd = np.tile(np.asarray(cfg.r, float), cfg.m)
    gX = np.zeros_like(X)
    gX[: cfg.N] = 2.0 * qd * (X[: cfg.N] - xs)
    gU = 2.0 * rd * U
    return pack(cfg, gX, gU)


# a2, a4, a5, a12: constraint vector in the reference's row order
def constraints(cfg: NLPConfig, w, p) -> np.ndarray:
    X, U = unpack(cfg, w)
    p = np.asarray(p, float).reshape(-1)
    g = np.empty(cfg.n_g)
    g[: cfg.nx] = X[0] - p[: cfg.nx]                       # C6:278
    if cfg.pad_rows:
        g[cfg.nx: cfg.rows0] = cfg.pad_value
    pr = cfg.pairs()
    o = cfg.rows0
    for k in range(cfg.N):
        st = X[k]
        g[o: o + cfg.nx] = X[k + 1] - (st + cfg.T * rhs(cfg, st, U[k]))   # C6:318-323
        o += cfg.nx
        for (i, j) in pr:                                   # C6:288-306, evaluated at X_k
            g[o] = (st[3 * i] - st[3 * j]) ** 2 + (st[3 * i + 1] - st[3 * j + 1]) ** 2
            o += 1
        for i in range(cfg.m):                              # O3:145-150 (robot-major for m>1)
            for (ox, oy, orad) in cfg.obstacles:
                g[o] = np.sqrt((st[3 * i] - ox) ** 2 + (st[3 * i + 1] - oy) ** 2) - cfg.rob_dim - orad
                o += 1
    assert o == cfg.n_g
    return g


# a7: bounds (C6:349-352; O3:175-177)
def bounds(cfg: NLPConfig):
    lbx_s = np.tile(np.array([-cfg.xy_max, -cfg.xy_max, -cfg.th_max]), cfg.m)
    ubx_s = -lbx_s
    lbu = np.tile(np.array([-cfg.v_max, -cfg.w_max]), cfg.m)
    lbx = np.concatenate([np.tile(lbx_s, cfg.N + 1), np.tile(lbu, cfg.N)])
    ubx = np.concatenate([np.tile(ubx_s, cfg.N + 1), np.tile(-lbu, cfg.N)])
    lb0 = np.concatenate([np.zeros(cfg.nx), np.full(cfg.rows0 - cfg.nx, cfg.dmin ** 2)])
    ub0 = np.concatenate([np.zeros(cfg.nx), np.full(cfg.rows0 - cfg.nx, INF)])
    lbk = np.concatenate([np.zeros(cfg.nx), np.full(cfg.M, cfg.dmin ** 2), np.full(cfg.m * cfg.K, cfg.margin)])
    ubk = np.concatenate([np.zeros(cfg.nx), np.full(cfg.M + cfg.m * cfg.K, INF)])
    lbg = np.concatenate([lb0, np.tile(lbk, cfg.N)])
    ubg = np.concatenate([ub0, np.tile(ubk, cfg.N)])
    return lbx, ubx, lbg, ubg


def jacobian(cfg: NLPConfig, w, p) -> np.ndarray:
    """dense dg/dw (n_g x n_var), analytic."""
    X, U = unpack(cfg, w)
    nx, nu, N, T = cfg.nx, cfg.nu, cfg.N, cfg.T
    J = np.zeros((cfg.n_g, cfg.n_var))
    uo = nx * (N + 1)
    J[:nx, :nx] = np.eye(nx)
    pr = cfg.pairs()
    o = cfg.rows0
    for k in range(N):
        st, con = X[k], U[k]
        xc, xn, uc = k * nx, (k + 1) * nx, uo + k * nu
        for i in range(cfg.m):
            th, v = st[3 * i + 2], con[2 * i]
            r0 = o + 3 * i
            J[r0, xn + 3 * i] = 1.0; J[r0 + 1, xn + 3 * i + 1] = 1.0; J[r0 + 2, xn + 3 * i + 2] = 1.0
            J[r0, xc + 3 * i] = -1.0; J[r0 + 1, xc + 3 * i + 1] = -1.0; J[r0 + 2, xc + 3 * i + 2] = -1.0
            J[r0, xc + 3 * i + 2] = T * v * np.sin(th)
            J[r0 + 1, xc + 3 * i + 2] = -T * v * np.cos(th)
            J[r0, uc + 2 * i] = -T * np.cos(th)
            J[r0 + 1, uc + 2 * i] = -T * np.sin(th)
            J[r0 + 2, uc + 2 * i + 1] = -T
        o += nx
        for (i, j) in pr:
            dx = st[3 * i] - st[3 * j]; dy = st[3 * i + 1] - st[3 * j + 1]
            J[o, xc + 3 * i] = 2 * dx; J[o, xc + 3 * i + 1] = 2 * dy
            J[o, xc + 3 * j] = -2 * dx; J[o, xc + 3 * j + 1] = -2 * dy
            o += 1
        for i in range(cfg.m):
            for (ox, oy, orad) in cfg.obstacles:
                dx = st[3 * i] - ox; dy = st[3 * i + 1] - oy
                rr = np.sqrt(dx * dx + dy * dy)
                J[o, xc + 3 * i] = dx / rr; J[o, xc + 3 * i + 1] = dy / rr
                o += 1
    return J


def hess_lagrangian(cfg: NLPConfig, w, p, lam) -> np.ndarray:
    """dense Hessian of f + lam^T g (IPOPT sign convention), analytic."""
    X, U = unpack(cfg, w)
    nx, nu, N, T = cfg.nx, cfg.nu, cfg.N, cfg.T
    H = np.zeros((cfg.n_var, cfg.n_var))
    qd = np.tile(np.asarray(cfg.q, float), cfg.m)
    rd = np.tile(np.asarray(cfg.r, float), cfg.m)
    uo = nx * (N + 1)
    pr = cfg.pairs()
    o = cfg.rows0
    for k in range(N):
        st, con = X[k], U[k]
        xc, uc = k * nx, uo + k * nu
        H[xc: xc + nx, xc: xc + nx] += np.diag(2 * qd)
        H[uc: uc + nu, uc: uc + nu] += np.diag(2 * rd)
        for i in range(cfg.m):
            th, v = st[3 * i + 2], con[2 * i]
            lx, ly = lam[o + 3 * i], lam[o + 3 * i + 1]
            it, iv = xc + 3 * i + 2, uc + 2 * i
            H[it, it] += T * v * (lx * np.cos(th) + ly * np.sin(th))
            c = T * (lx * np.sin(th) - ly * np.cos(th))
            H[it, iv] += c; H[iv, it] += c
        o += nx
        for (i, j) in pr:
            l = lam[o]
            for d in (0, 1):
                a, b = xc + 3 * i + d, xc + 3 * j + d
                H[a, a] += 2 * l; H[b, b] += 2 * l; H[a, b] -= 2 * l; H[b, a] -= 2 * l
            o += 1
        for i in range(cfg.m):
            for (ox, oy, orad) in cfg.obstacles:
                l = lam[o]
                dx = st[3 * i] - ox; dy = st[3 * i + 1] - oy
                rr = np.sqrt(dx * dx + dy * dy)
                n = np.array([dx, dy]) / rr
                blk = (np.eye(2) - np.outer(n, n)) / rr
                a = xc + 3 * i
                H[a: a + 2, a: a + 2] += l * blk
                o += 1
    return H


# a11: shift + warm start (C6:160-169, 465); cold start (C6:398-400)
def shift(T: float, t0: float, u: np.ndarray):
    u = np.asarray(u)
    u0 = np.concatenate((u[1:], u[-1:]), axis=0)
    return t0 + T, u0


def shift_states(cfg: NLPConfig, Xsol: np.ndarray) -> np.ndarray:
    """X0 = [Xsol[1:]; Xsol[N-1]]: the appended row is row N-1 of the (N+1)-row array (C6:465)."""
    return np.concatenate((Xsol[1:], Xsol[cfg.N - 1: cfg.N]), axis=0)


def cold_start(cfg: NLPConfig, x0: np.ndarray) -> np.ndarray:
    return pack(cfg, np.tile(np.asarray(x0, float).reshape(1, -1), (cfg.N + 1, 1)), np.zeros((cfg.N, cfg.nu)))


# a13: offline plant step (casadi_test.py:17-26)
def plant_step(cfg: NLPConfig, x0: np.ndarray, u_first: np.ndarray) -> np.ndarray:
    x0 = np.asarray(x0, float).reshape(-1)
    return x0 + cfg.T * rhs(cfg, x0, np.asarray(u_first, float).reshape(-1))


# ----------------------------------------------------------------------------
# solver-independent KKT check of a candidate solution
def kkt_report(cfg: NLPConfig, w, p, tol_active: float = 1e-6):
    """Least-squares multipliers on the active set -> stationarity residual.

    Returns dict(stat, eq, ineq, bnd): inf-norms of the stationarity residual,
    equality violation, inequality violation, variable-bound violation."""
    from scipy.optimize import lsq_linear
    w = np.asarray(w, float).reshape(-1)
    lbx, ubx, lbg, ubg = bounds(cfg)
    g = constraints(cfg, w, p)
    J = jacobian(cfg, w, p)
    gf = grad_objective(cfg, w, p)
    eq = np.where(lbg == ubg)[0]
    ineq = np.where(lbg != ubg)[0]
    act_g = ineq[(g[ineq] - lbg[ineq]) <= tol_active]
    act_lb = np.where(w - lbx <= tol_active)[0]
    act_ub = np.where(ubx - w <= tol_active)[0]
    # stationarity: gf + J_eq^T l + J_act^T l_a(<=0) - e_lb z_lb(>=0) + e_ub z_ub(>=0) = 0
    cols = [J[eq].T, J[act_g].T]
    E = np.zeros((w.size, act_lb.size + act_ub.size))
    for c, i in enumerate(act_lb): E[i, c] = -1.0
    for c, i in enumerate(act_ub): E[i, act_lb.size + c] = 1.0
    A = np.concatenate(cols + [E], axis=1)
    lo = np.concatenate([np.full(eq.size, -INF), np.full(act_g.size, -INF), np.zeros(E.shape[1])])
    hi = np.concatenate([np.full(eq.size, INF), np.zeros(act_g.size), np.full(E.shape[1], INF)])
    res = lsq_linear(A, -gf, bounds=(lo, hi), tol=1e-14, max_iter=500)
    stat = float(np.max(np.abs(A @ res.x + gf))) if A.shape[1] else float(np.max(np.abs(gf)))
    return dict(stat=stat,
                eq=float(np.max(np.abs(g[eq] - lbg[eq]))),
                ineq=float(max(0.0, np.max(lbg[ineq] - g[ineq]))) if ineq.size else 0.0,
                bnd=float(max(0.0, np.max(lbx - w), np.max(w - ubx))))


# literal configurations of the reference files -------------------------------------------------
def cfg_one(N=20):   # C1:58-63
    return NLPConfig(m=1, N=N, T=0.05, dmin=0.0, v_max=0.22, w_max=2.84, pad_rows=False)

def cfg_two(N=20):   # C2:101-109
    return NLPConfig(m=2, N=N, T=0.05, dmin=0.15, v_max=0.22, w_max=2.84)

def cfg_six(N=20):   # C6:197-205
    return NLPConfig(m=6, N=N, T=0.3, dmin=0.4, v_max=0.15, w_max=1.5)

def cfg_two_nopairs(N=50):   # AS/mpc_online_casadi_tb3_multi_centralized.py:60-66,115-148: m=2, T=0.01, N=50, no pair rows, no padding rows
    return NLPConfig(m=2, N=N, T=0.01, dmin=0.0, v_max=0.22, w_max=2.84, pad_rows=False, pair_rows=False)

def cfg_ten(N=30):   # C10:169-177
    return NLPConfig(m=10, N=N, T=0.1, dmin=0.3, v_max=0.22, w_max=2.84)

def cfg_obs3(N=20):  # O3:56-63, 97-119, 175-177
    return NLPConfig(m=1, N=N, T=0.2, dmin=0.0, v_max=0.2, w_max=1.0, th_max=2 * np.pi, pad_rows=False,
                     rob_dim=0.2, margin=0.1,
                     obstacles=[(-0.6, 3.3, 0.2), (0.6, 3.3, 0.125), (0.0, 2.3, 0.15),
                                (1.0, 2.3, 0.15), (-0.6, 1.3, 0.2), (0.6, 1.3, 0.175)])

# literal start / goal sets
C2_START = np.array([-0.7112, -0.7112, 0.785, 0.7112, 0.7112, -2.356])     # C2:213-214
C2_GOAL = np.array([0.7112, 0.7112, 0.785, -0.7112, -0.7112, -2.356])       # C2:224
C6_START = np.array([0.7, 0.4, -2.618, 0.0, 0.8, -1.57, -0.7, 0.4, -0.523,
                     -0.7, -0.4, 0.523, 0.0, -0.8, 1.57, 0.7, -0.4, 2.618])  # C6:364-369
C6_GOAL = np.array([-0.7, -0.4, -2.618, 0.0, -0.8, -1.57, 0.7, -0.4, -0.523,
                    0.7, 0.4, 0.523, 0.0, 0.8, 1.57, -0.7, 0.4, 2.618])      # C6:386-388


def odom_to_global(odom, init, wrap_2pi=False):
    """Odometry callbacks of the scripts (C2:18-37): odom [n,4] = (x_r, y_r, q_z, q_w) in the robot's start frame,
    init [n,3] = (x_init, y_init, th_init) -> pose [n,3] in the global frame.  q_w is not used by the reference either.
    wrap_2pi: modify() of AllScripts/mpc_online_casadi.py:24-33 (th in [-pi, 0) -> th + 2 pi); identity in C2:62-68."""
    odom = np.asarray(odom, dtype=np.float64).reshape(-1, 4); init = np.asarray(init, dtype=np.float64).reshape(-1, 3)
    th = 2.0 * np.arcsin(odom[:, 2])
    if wrap_2pi:
        th = np.where((th >= -np.pi) & (th < 0.0), th + 2.0 * np.pi, th)
    c, s = np.cos(init[:, 2]), np.sin(init[:, 2])
    x = (c * odom[:, 0] - s * odom[:, 1]) + init[:, 0]
    y = (s * odom[:, 0] + c * odom[:, 1]) + init[:, 1]
    return np.stack([x, y, th + init[:, 2]], axis=1)
