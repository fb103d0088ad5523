"""TEST INFRASTRUCTURE — numpy restatement of the LIDAR-ray distance-state NMPC of the reference.

Imported only by tests/, by the golden-vector generator and by bench.py's cpu_baseline leg; the product never imports it.
PARITY UNPINNED against CasADi/IPOPT (see oracle/nlp_ref.py); pinned by the solver-independent known answers in
tests/test_oracle_lidar.py and scipy-SLSQP golden triples.

  V4 = AllScripts/obs_avoid_static_first_scenario_v4.py      V3 = AllScripts/obs_avoid_static_first_scenario_v3.py

The NLP (V4:59-151):
  states per stage  [x, y, theta, d_1 .. d_R]   (13 for R = numRays = 10, V4:78-87), X is n_states x (N+1)
  controls          U is 2 x Nc;  stage k uses U[:, min(k, Nc-1)]  (move blocking, V4:128-131;  V3: Nc = N)
  parameters        P = [x0 (3); xs (3); D0 (R); ray angles B (R)]                                    (V4:101)
  lidar points      pObs_m = Rz(theta_0) d_{m,0} e(B_m) + p_0, from the STAGE-0 decision variables    (V4:114-118)
  cost              sum_{k<N} (x_k - xs)' Q (x_k - xs) + con_k' R con_k + 0.1 sum_m 1 / d_{m,k}^2     (V4:121-123,135-136;  V3: no 1/d^2)
  constraints       g = [gx; gd], all equalities (V4:151,158-159):
      gx = [x_0 - P[0:3];  x_{k+1} - (x_k + T f(x_k, con_k)), k = 0..N-1]                               (V4:110,137-140)
      gd = [d_0 - P[6:6+R]; d_{m,k+1} - ||p_{k+1} - pObs_m||_1,  k = 0..N-1, m = 0..R-1]                 (V4:111,142-148)
  packing           w = [vec(X); vec(U)], column-major reshape: stage-major, 13 entries per stage      (V4:155)
  bounds            lbx = [repmat([x_min, y_min, theta_min], N+1); repmat(d_min, R (N+1)); repmat([v_min, w_min], Nc)]   (V4:161-176)
                    — built as "all pose bounds, then all distance bounds", which does NOT line up with the stage-major packing of
                    w: entry j < 3(N+1) of the X part gets the pose pattern by j mod 3, every later entry gets [d_min, d_max].
                    Restated as built (reference behaviour is the contract; SURVEY.md 7 "reproduce, do not fix").
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Tuple

import numpy as np

INF = np.inf


@dataclass
class LidarConfig:
    """literals of V4:56-75 (defaults) / V3:54-70 (lidar_v3())."""
    N: int = 100
    Nc: int = 50
    R: int = 10                      # numRays
    T: float = 0.075
    q: Tuple[float, float, float] = (1.0, 5.0, 0.1)     # V4:119
    r: Tuple[float, float] = (0.5, 0.05)                # V4:120
    lw: float = 0.1                  # L = 0.1 I (V4:121); 0 reproduces V3 (no 1/d^2 term)
    v_max: float = 0.15
    w_max: float = 1.5
    xy_max: float = 10.0
    th_max: float = INF
    d_min: float = 0.15              # robot_radius (V4:66-68)
    d_max: float = 10.0              # V3: inf
    aligned_bounds: bool = False     # False: lbx / ubx exactly as the script builds them (see the module docstring)

    @property
    def ns(self): return 3 + self.R
    @property
    def n_var(self): return self.ns * (self.N + 1) + 2 * self.Nc
    @property
    def n_g(self): return self.ns * (self.N + 1)
    @property
    def n_p(self): return 6 + 2 * self.R


def lidar_v4(N=100, Nc=50):
    return LidarConfig(N=N, Nc=Nc)


def lidar_v3(N=125):     # V3:54-70: N = 125, no move blocking, no barrier cost, robot_radius 0.2, omega_max 2, d_max inf
    return LidarConfig(N=N, Nc=N, lw=0.0, w_max=2.0, d_min=0.2, d_max=INF)


def ray_angles(R):       # B0 of V4:203-205
    return np.arange(R) * (2.0 * np.pi) / R


def unpack(cfg, w):
    w = np.asarray(w, dtype=np.float64).reshape(-1)
    assert w.size == cfg.n_var
    X = w[: cfg.ns * (cfg.N + 1)].reshape(cfg.N + 1, cfg.ns)
    U = w[cfg.ns * (cfg.N + 1):].reshape(cfg.Nc, 2)
    return X, U


def pack(cfg, X, U):
    return np.concatenate([np.asarray(X, float).reshape(-1), np.asarray(U, float).reshape(-1)])


def pobs(cfg, X0, p):
    """V4:114-118: lidar points from the stage-0 variables (pinned to P by the first rows of gx / gd)."""
    B = np.asarray(p, float).reshape(-1)[6 + cfg.R: 6 + 2 * cfg.R]
    th = X0[2]
    return np.stack([X0[0] + X0[3:] * np.cos(th + B), X0[1] + X0[3:] * np.sin(th + B)], axis=1)      # [R, 2]


def objective(cfg, w, p):
    X, U = unpack(cfg, w)
    xs = np.asarray(p, float).reshape(-1)[3:6]
    f = 0.0
    for k in range(cfg.N):
        con = U[min(k, cfg.Nc - 1)]
        e = X[k, :3] - xs
        f += float(np.dot(cfg.q, e * e) + np.dot(cfg.r, con * con))
        if cfg.lw:
            f += cfg.lw * float(np.sum(1.0 / X[k, 3:] ** 2))
    return f


def constraints(cfg, w, p):
    X, U = unpack(cfg, w)
    p = np.asarray(p, float).reshape(-1)
    N, R = cfg.N, cfg.R
    po = pobs(cfg, X[0], p)
    gx = np.empty(3 * (N + 1)); gd = np.empty(R * (N + 1))
    gx[:3] = X[0, :3] - p[:3]
    gd[:R] = X[0, 3:] - p[6: 6 + R]
    for k in range(N):
        con = U[min(k, cfg.Nc - 1)]
        th = X[k, 2]
        nxt = X[k, :3] + cfg.T * np.array([con[0] * np.cos(th), con[0] * np.sin(th), con[1]])
        gx[3 * (k + 1): 3 * (k + 2)] = X[k + 1, :3] - nxt
        gd[R * (k + 1): R * (k + 2)] = X[k + 1, 3:] - (np.abs(X[k + 1, 0] - po[:, 0]) + np.abs(X[k + 1, 1] - po[:, 1]))
    return np.concatenate([gx, gd])      # V4:151: all gx rows, then all gd rows


def bounds(cfg):
    N, R = cfg.N, cfg.R
    pose_lb = np.array([-cfg.xy_max, -cfg.xy_max, -cfg.th_max])
    if cfg.aligned_bounds:
        lbs = np.concatenate([pose_lb, np.full(R, cfg.d_min)]); ubs = np.concatenate([-pose_lb, np.full(R, cfg.d_max)])
        lbX = np.tile(lbs, N + 1); ubX = np.tile(ubs, N + 1)
    else:       # V4:161-172 verbatim: pose pattern for the first 3(N+1) entries, then the distance bounds
        lbX = np.concatenate([np.tile(pose_lb, N + 1), np.full(R * (N + 1), cfg.d_min)])
        ubX = np.concatenate([np.tile(-pose_lb, N + 1), np.full(R * (N + 1), cfg.d_max)])
    lbU = np.tile(np.array([-cfg.v_max, -cfg.w_max]), cfg.Nc)
    return np.concatenate([lbX, lbU]), np.concatenate([ubX, -lbU]), np.zeros(cfg.n_g), np.zeros(cfg.n_g)


def cold_start(cfg, x0_full):
    """V4:184-196: X0 = repmat(x0) (pose and scan), u0 = 0."""
    return pack(cfg, np.tile(np.asarray(x0_full, float).reshape(1, -1), (cfg.N + 1, 1)), np.zeros((cfg.Nc, 2)))


def make_p(cfg, x0_pose, xs, scan):
    """V4:230-236: p = [x0; xs; scan; ray angles]."""
    return np.concatenate([np.asarray(x0_pose, float), np.asarray(xs, float), np.asarray(scan, float), ray_angles(cfg.R)])


def shift_guess(cfg, w):
    """V4:258-270: u0 = [u[1:]; u[-1]];  X0 = [X[1:]; X[N-1]] (row N-1 appended, as in the other scripts)."""
    X, U = unpack(cfg, w)
    return pack(cfg, np.concatenate([X[1:], X[cfg.N - 1: cfg.N]]), np.concatenate([U[1:], U[-1:]]))


def grad_objective(cfg, w, p):
    X, U = unpack(cfg, w)
    xs = np.asarray(p, float).reshape(-1)[3:6]
    gX = np.zeros_like(X); gU = np.zeros_like(U)
    for k in range(cfg.N):
        j = min(k, cfg.Nc - 1)
        gX[k, :3] = 2.0 * np.asarray(cfg.q) * (X[k, :3] - xs)
        gU[j] += 2.0 * np.asarray(cfg.r) * U[j]
        if cfg.lw:
            gX[k, 3:] = -2.0 * cfg.lw / X[k, 3:] ** 3
    return pack(cfg, gX, gU)


def jacobian(cfg, w, p):
    """dense dg/dw, analytic; d|a|/da = sign(a) (CasADi's norm_1 derivative)."""
    X, U = unpack(cfg, w)
    p = np.asarray(p, float).reshape(-1)
    N, R, ns, T = cfg.N, cfg.R, cfg.ns, cfg.T
    B = p[6 + R: 6 + 2 * R]
    J = np.zeros((cfg.n_g, cfg.n_var))
    uo = ns * (N + 1)
    po = pobs(cfg, X[0], p)
    th0 = X[0, 2]
    # d pObs / d (x0, y0, th0, d_m0)
    J[:3, :3] = np.eye(3)
    go = 3 * (N + 1)
    for m in range(R):
        J[go + m, 3 + m] = 1.0
    for k in range(N):
        j = min(k, cfg.Nc - 1)
        th, v = X[k, 2], U[j, 0]
        r0 = 3 * (k + 1)
        xc, xn, uc = k * ns, (k + 1) * ns, uo + 2 * j
        for c in range(3):
            J[r0 + c, xn + c] = 1.0; J[r0 + c, xc + c] = -1.0
        J[r0, xc + 2] += T * v * np.sin(th); J[r0 + 1, xc + 2] += -T * v * np.cos(th)
        J[r0, uc] = -T * np.cos(th); J[r0 + 1, uc] = -T * np.sin(th); J[r0 + 2, uc + 1] = -T
        for m in range(R):
            row = go + R * (k + 1) + m
            sx = np.sign(X[k + 1, 0] - po[m, 0]); sy = np.sign(X[k + 1, 1] - po[m, 1])
            J[row, xn + 3 + m] = 1.0
            J[row, xn] += -sx; J[row, xn + 1] += -sy
            # through pObs(X_0): + sx * d pox + sy * d poy
            J[row, 0] += sx; J[row, 1] += sy
            J[row, 2] += sx * (-X[0, 3 + m] * np.sin(th0 + B[m])) + sy * (X[0, 3 + m] * np.cos(th0 + B[m]))
            J[row, 3 + m] += sx * np.cos(th0 + B[m]) + sy * np.sin(th0 + B[m])
    return J


def kkt_report(cfg, w, p, tol_active=1e-6):
    """least-squares multipliers on the active set -> stationarity residual of a candidate solution (solver-independent)."""
    from scipy.optimize import lsq_linear
    w = np.asarray(w, float).reshape(-1)
    lbx, ubx, lbg, ubg = bounds(cfg)
    g = constraints(cfg, w, p); J = jacobian(cfg, w, p); gf = grad_objective(cfg, w, p)
    act_lb = np.where(w - lbx <= tol_active)[0]; act_ub = np.where(ubx - w <= tol_active)[0]
    E = np.zeros((w.size, act_lb.size + act_ub.size))
    for c, i in enumerate(act_lb): E[i, c] = -1.0
    for c, i in enumerate(act_ub): E[i, act_lb.size + c] = 1.0
    A = np.concatenate([J.T, E], axis=1)
    lo = np.concatenate([np.full(J.shape[0], -INF), np.zeros(E.shape[1])]); hi = np.full(A.shape[1], INF)
    res = lsq_linear(A, -gf, bounds=(lo, hi), tol=1e-14, max_iter=500)
    return dict(stat=float(np.max(np.abs(A @ res.x + gf))), eq=float(np.max(np.abs(g))),
                bnd=float(max(0.0, np.max(lbx - w), np.max(w - ubx))))


def scan_of_world(pose, obstacles, R, scan_max=3.5):
    """synthetic LaserScan (V4:28-36 clips inf to 3.5): range along each ray to the nearest circular obstacle (ox, oy, r)."""
    x, y, th = pose
    out = np.full(R, scan_max)
    for m, b in enumerate(ray_angles(R)):
        dx, dy = np.cos(th + b), np.sin(th + b)
        for (ox, oy, r) in obstacles:
            fx, fy = ox - x, oy - y
            t = fx * dx + fy * dy
            h2 = fx * fx + fy * fy - t * t
            if fx * fx + fy * fy <= r * r:      # a pose inside (or touching) an obstacle reads range 0 on every ray
                out[m] = 0.0
            elif t > 0 and h2 < r * r:
                out[m] = min(out[m], max(t - np.sqrt(r * r - h2), 0.0))
    return out
