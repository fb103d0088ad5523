"""Host-side mirror of the reference's call surface for the LIDAR-ray distance-state NMPC
(V4 = AllScripts/obs_avoid_static_first_scenario_v4.py, V3 = ..._v3.py):

    solver = lidar_nlpsol('solver', 'ipopt', cfg, opts)            # V4:156-157 (lbx / ubx of V4:174-176 are taken here)
    sol    = solver(x0=, p=, lbx=, ubx=, lbg=, ubg=)               # V4:245 ; sol['x'] ((13 (N+1) + 2 Nc) x 1)

All arithmetic happens in libnmpc_hip.so (nmpc_lidar_* of include/nmpc_lidar.h); torch is device memory and streams."""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np

from . import _lib


@dataclass
class LidarProblemConfig:
    """literals of V4:56-75 (defaults); lidar_v3() gives V3:54-70."""
    N: int = 100
    Nc: int = 50
    R: int = 10
    T: float = 0.075
    q: Tuple[float, float, float] = (1.0, 5.0, 0.1)
    r: Tuple[float, float] = (0.5, 0.05)
    lw: float = 0.1                       # L = 0.1 I (V4:121)
    v_max: float = 0.15
    w_max: float = 1.5
    xy_max: float = 10.0
    th_max: float = math.inf
    d_min: float = 0.15
    d_max: float = 10.0
    tol: float = 1e-8
    mu_init: float = 0.5
    max_iter: int = 2000

    @property
    def ns(self): return 3 + self.R
    @property
    def n_var(self): return self.ns * (self.N + 1) + 2 * self.Nc
    @property
    def n_g(self): return self.ns * (self.N + 1)
    @property
    def n_p(self): return 6 + 2 * self.R

    def to_c(self) -> "_lib.CLidarConfig":
        c = _lib.CLidarConfig()
        c.N, c.Nc, c.R, c.max_iter, c.T, c.lw, c.tol, c.mu_init = self.N, self.Nc, self.R, self.max_iter, self.T, self.lw, self.tol, self.mu_init
        c.q[:] = self.q; c.r[:] = self.r
        return c

    def bounds(self):
        """lbx, ubx, lbg, ubg exactly as the script builds them (V4:158-176): the pose bounds for the first 3(N+1) entries of
        the state part, then the distance bounds for the remaining R(N+1) — NOT aligned with the stage-major packing of w
        (13 entries per stage); reproduced as built."""
        N, R = self.N, self.R
        pose_lb = np.array([-self.xy_max, -self.xy_max, -self.th_max])
        lbX = np.concatenate([np.tile(pose_lb, N + 1), np.full(R * (N + 1), self.d_min)])
        ubX = np.concatenate([np.tile(-pose_lb, N + 1), np.full(R * (N + 1), self.d_max)])
        lbU = np.tile(np.array([-self.v_max, -self.w_max]), self.Nc)
        return np.concatenate([lbX, lbU]), np.concatenate([ubX, -lbU]), np.zeros(self.n_g), np.zeros(self.n_g)

    def aligned_bounds(self):
        """the bounds the script presumably meant: [pose; distances] per stage."""
        lbs = np.concatenate([[-self.xy_max, -self.xy_max, -self.th_max], np.full(self.R, self.d_min)])
        ubs = np.concatenate([[self.xy_max, self.xy_max, self.th_max], np.full(self.R, self.d_max)])
        lbU = np.tile(np.array([-self.v_max, -self.w_max]), self.Nc)
        return np.concatenate([np.tile(lbs, self.N + 1), lbU]), np.concatenate([np.tile(ubs, self.N + 1), -lbU]), np.zeros(self.n_g), np.zeros(self.n_g)


def lidar_v4(N: int = 100, Nc: int = 50) -> LidarProblemConfig:       # V4:56-75
    return LidarProblemConfig(N=N, Nc=Nc)


def lidar_v3(N: int = 125) -> LidarProblemConfig:                    # V3:54-70
    return LidarProblemConfig(N=N, Nc=N, lw=0.0, w_max=2.0, d_min=0.2, d_max=math.inf)


def lidar_cold_start(cfg: LidarProblemConfig, x0_full) -> np.ndarray:
    """V4:184-196: X0 = repmat(x0) (pose and scan), u0 = 0, packed [vec(X); vec(U)]."""
    return np.concatenate([np.tile(np.asarray(x0_full, dtype=np.float64).reshape(-1), cfg.N + 1), np.zeros(2 * cfg.Nc)])


def lidar_params(cfg: LidarProblemConfig, pose, xs, scan) -> np.ndarray:
    """V4:230-236: p = [x0 pose; xs; scan; B0], B0[m] = m 2 pi / R (V4:203-205)."""
    return np.concatenate([np.asarray(pose, float).reshape(-1), np.asarray(xs, float).reshape(-1), np.asarray(scan, float).reshape(-1),
                           np.arange(cfg.R) * (2.0 * math.pi) / cfg.R])


class LidarSolver:
    """The object lidar_nlpsol() returns.  Owns the device workspace: one contiguous block per instance (the solve kernel runs one
    wavefront per instance and spreads the stages of the horizon over its lanes)."""

    def __init__(self, cfg: LidarProblemConfig, lbx=None, ubx=None, max_batch: int = 1, device: Optional[int] = None):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: the NMPC solve has no CPU fallback")
        self.torch = torch
        self.cfg = cfg
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        b = cfg.bounds()
        self._lbx = np.ascontiguousarray(b[0] if lbx is None else np.asarray(lbx, dtype=np.float64).reshape(-1))
        self._ubx = np.ascontiguousarray(b[1] if ubx is None else np.asarray(ubx, dtype=np.float64).reshape(-1))
        if self._lbx.size != cfg.n_var or self._ubx.size != cfg.n_var:
            raise ValueError(f"lbx / ubx must have {cfg.n_var} entries")
        self._ccfg = cfg.to_c()
        self._h = C.c_void_p()
        self.max_batch = int(max_batch)
        dp = C.POINTER(C.c_double)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nmpc_lidar_create(C.byref(self._ccfg), self._lbx.ctypes.data_as(dp), self._ubx.ctypes.data_as(dp), self.max_batch, C.byref(self._h)),
                       "nmpc_lidar_create")
        self.n_var, self.n_g, self.n_p = cfg.n_var, cfg.n_g, cfg.n_p
        assert self.n_var == self.lib.nmpc_lidar_n_var(C.byref(self._ccfg)) and self.n_g == self.lib.nmpc_lidar_n_g(C.byref(self._ccfg))
        self._stats: Dict = {}

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self.lib.nmpc_lidar_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def _dev(self, a, shape):
        return self.torch.as_tensor(a, dtype=self.torch.float64, device=self.device).reshape(shape).contiguous()

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def solve_batch(self, p, w0, want_fg: bool = False):
        torch = self.torch
        p = self._dev(p, (-1, self.n_p)); B = p.shape[0]
        w0 = self._dev(w0, (B, self.n_var))
        if B > self.max_batch:
            raise ValueError(f"batch {B} exceeds max_batch {self.max_batch}")
        w = torch.empty((B, self.n_var), dtype=torch.float64, device=self.device)
        obj = torch.empty(B, dtype=torch.float64, device=self.device); kkt = torch.empty(B, dtype=torch.float64, device=self.device)
        status = torch.empty(B, dtype=torch.int32, device=self.device); iters = torch.empty(B, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nmpc_lidar_solve_batch(self._h, B, p.data_ptr(), w0.data_ptr(), w.data_ptr(), obj.data_ptr(), status.data_ptr(),
                                                       iters.data_ptr(), kkt.data_ptr(), self._stream()), "nmpc_lidar_solve_batch")
        out = dict(x=w, f=obj, status=status, iters=iters, kkt=kkt)
        if want_fg:
            out["f"], out["g"] = self.eval_batch(p, w)
        return out

    def eval_batch(self, p, w):
        torch = self.torch
        p = self._dev(p, (-1, self.n_p)); B = p.shape[0]
        w = self._dev(w, (B, self.n_var))
        f = torch.empty(B, dtype=torch.float64, device=self.device); g = torch.empty((B, self.n_g), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nmpc_lidar_eval_batch(self._h, B, p.data_ptr(), w.data_ptr(), f.data_ptr(), g.data_ptr(), self._stream()), "nmpc_lidar_eval_batch")
        return f, g

    def shift_batch(self, w):
        """V4:258-270 on device: U rows drop first / repeat last; X rows [X_1..X_N; X_{N-1}]."""
        torch = self.torch
        w = self._dev(w, (-1, self.n_var))
        wn = torch.empty_like(w)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nmpc_lidar_shift_batch(self._h, w.shape[0], w.data_ptr(), wn.data_ptr(), self._stream()), "nmpc_lidar_shift_batch")
        return wn

    def scan_batch(self, pose, world, scan_max: float = 3.5):
        """synthetic LaserScan of callback_lidar (V4:29-36) on the device: pose [B,3], world [B,K,3] = (ox, oy, radius) -> scan [B,R]."""
        torch = self.torch
        pose = self._dev(pose, (-1, 3)); B = pose.shape[0]
        world = self._dev(world, (B, -1, 3)); K = world.shape[1]
        scan = torch.empty((B, self.cfg.R), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nmpc_lidar_scan_batch(B, self.cfg.R, K, pose.data_ptr(), world.data_ptr() if K else None, float(scan_max), scan.data_ptr(), self._stream()),
                       "nmpc_lidar_scan_batch")
        return scan

    def plant_batch(self, p, w_sol):
        """pose + T f(pose, u_0) on the device (the robot of V4 replaced by the model of its own NLP): p [B,n_p], w_sol [B,n_var] -> [B,3]."""
        torch = self.torch
        p = self._dev(p, (-1, self.n_p)); B = p.shape[0]
        w_sol = self._dev(w_sol, (B, self.n_var))
        out = torch.empty((B, 3), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nmpc_lidar_plant_batch(self._h, B, p.data_ptr(), w_sol.data_ptr(), out.data_ptr(), 0, self._stream()), "nmpc_lidar_plant_batch")
        return out

    def __call__(self, x0=None, p=None, lbx=None, ubx=None, lbg=None, ubg=None, **kw):
        """the reference's keyword call (V4:245); bounds must be the ones the solver was built with (they never change in the script)."""
        if kw:
            raise TypeError(f"unexpected arguments {sorted(kw)}")
        if p is None or x0 is None:
            raise ValueError("x0 and p are required")
        p = np.asarray(p, dtype=np.float64).reshape(-1); w0 = np.asarray(x0, dtype=np.float64).reshape(-1)
        if p.size != self.n_p:
            raise ValueError(f"p has {p.size} entries, expected {self.n_p}")
        if w0.size != self.n_var:
            raise ValueError(f"x0 has {w0.size} entries, expected {self.n_var}")
        for name, given, want in (("lbx", lbx, self._lbx), ("ubx", ubx, self._ubx), ("lbg", lbg, np.zeros(self.n_g)), ("ubg", ubg, np.zeros(self.n_g))):
            if given is None:
                continue
            g = np.asarray(given, dtype=np.float64).reshape(-1)
            if g.size != want.size:
                raise ValueError(f"{name} has {g.size} entries, expected {want.size}")
            if not np.array_equal(g, want):
                raise ValueError(f"{name} differs from the bounds this solver was built with; build a new solver instead")
        r = self.solve_batch(p[None], w0[None], want_fg=True)
        self.torch.cuda.synchronize(self.device)
        st = int(r["status"][0])
        self._stats = dict(return_status=_lib.STATUS_NAMES.get(st, str(st)), success=(st == 0), iter_count=int(r["iters"][0]), kkt_error=float(r["kkt"][0]), status_code=st)
        return {"x": r["x"][0].cpu().numpy().reshape(-1, 1), "f": float(r["f"][0]), "g": r["g"][0].cpu().numpy().reshape(-1, 1)}

    def stats(self) -> Dict:
        return dict(self._stats)


def lidar_nlpsol(name: str, plugin: str, problem: LidarProblemConfig, opts: Optional[dict] = None, lbx=None, ubx=None, max_batch: int = 1) -> LidarSolver:
    """Drop-in for casadi.nlpsol(name,'ipopt',nlp_prob,opts) of V4:156-157 with `problem` a LidarProblemConfig."""
    if plugin != "ipopt":
        raise ValueError("only the 'ipopt' call surface of the reference is mirrored")
    if not isinstance(problem, LidarProblemConfig):
        raise TypeError("problem must be a LidarProblemConfig")
    ip = dict((opts or {}).get("ipopt", {}))
    cfg = LidarProblemConfig(**{**problem.__dict__})
    if "max_iter" in ip: cfg.max_iter = int(ip["max_iter"])
    if "tol" in ip: cfg.tol = float(ip["tol"])
    elif "acceptable_tol" in ip: cfg.tol = float(ip["acceptable_tol"])
    if "mu_init" in ip: cfg.mu_init = float(ip["mu_init"])
    return LidarSolver(cfg, lbx=lbx, ubx=ubx, max_batch=max_batch)
