"""TEST INFRASTRUCTURE — ctypes loader for oracle/libnmpc_oracle.so (the C fp64 CPU restatement).

Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Config(C.Structure):
    """mirror of nmpc_config_t (include/nmpc.h)."""
    _fields_ = [("m", C.c_int32), ("N", C.c_int32), ("n_obs", C.c_int32), ("pad_rows", C.c_int32),
                ("T", C.c_double), ("dmin", C.c_double), ("q", C.c_double * 3), ("r", C.c_double * 2),
                ("v_max", C.c_double), ("w_max", C.c_double), ("xy_max", C.c_double), ("th_max", C.c_double),
                ("rob_dim", C.c_double), ("margin", C.c_double), ("pad_value", C.c_double),
                ("obs", C.c_double * 24), ("tol", C.c_double), ("mu_init", C.c_double),
                ("max_iter", C.c_int32), ("pair_rows", C.c_int32)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libnmpc_oracle.so")
    src = os.path.join(_HERE, "nmpc_oracle.c")
    hdrs = [os.path.join(_HERE, "..", "include", h) for h in ("nmpc.h", "nmpc_constants.h")]
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(so) < max(os.path.getmtime(f) for f in [src] + hdrs)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libnmpc_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        dp = C.POINTER(C.c_double); ip = C.POINTER(C.c_int32)
        _LIB.nmpc_oracle_solve_batch.argtypes = [C.POINTER(Config), C.c_int32, dp, dp, dp, dp, ip, ip, dp, C.c_int32]
        _LIB.nmpc_oracle_solve_batch.restype = C.c_int32
        _LIB.nmpc_oracle_eval_batch.argtypes = [C.POINTER(Config), C.c_int32, dp, dp, dp, dp]
        _LIB.nmpc_oracle_shift_batch.argtypes = [C.POINTER(Config), C.c_int32, dp, dp, dp, dp]
        _LIB.nmpc_oracle_max_threads.restype = C.c_int32
        for f in ("nmpc_oracle_n_var", "nmpc_oracle_n_g", "nmpc_oracle_n_p"):
            getattr(_LIB, f).argtypes = [C.POINTER(Config)]
            getattr(_LIB, f).restype = C.c_int32
    return _LIB


def make_config(nlp_cfg, tol=1e-8, mu_init=0.5, max_iter=2000) -> Config:
    """oracle.nlp_ref.NLPConfig -> nmpc_config_t."""
    c = Config()
    c.m, c.N, c.n_obs, c.pad_rows = nlp_cfg.m, nlp_cfg.N, len(nlp_cfg.obstacles), int(nlp_cfg.pad_rows)
    c.T, c.dmin = nlp_cfg.T, nlp_cfg.dmin
    c.q[:] = nlp_cfg.q; c.r[:] = nlp_cfg.r
    c.v_max, c.w_max, c.xy_max, c.th_max = nlp_cfg.v_max, nlp_cfg.w_max, nlp_cfg.xy_max, nlp_cfg.th_max
    c.rob_dim, c.margin, c.pad_value = nlp_cfg.rob_dim, nlp_cfg.margin, nlp_cfg.pad_value
    for i, (ox, oy, orad) in enumerate(nlp_cfg.obstacles):
        c.obs[3 * i], c.obs[3 * i + 1], c.obs[3 * i + 2] = ox, oy, orad
    c.tol, c.mu_init, c.max_iter = tol, mu_init, max_iter
    c.pair_rows = int(getattr(nlp_cfg, "pair_rows", True))
    return c


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def solve_batch(cfg: Config, p: np.ndarray, w0: np.ndarray, nthreads: int = 0):
    L = lib()
    p = np.ascontiguousarray(p, dtype=np.float64); w0 = np.ascontiguousarray(w0, dtype=np.float64)
    B = p.shape[0]
    nv = L.nmpc_oracle_n_var(C.byref(cfg))
    assert p.shape == (B, 6 * cfg.m) and w0.shape == (B, nv)
    w = np.empty((B, nv)); obj = np.empty(B); kkt = np.empty(B)
    st = np.empty(B, dtype=np.int32); it = np.empty(B, dtype=np.int32)
    rc = L.nmpc_oracle_solve_batch(C.byref(cfg), B, _dp(p), _dp(w0), _dp(w), _dp(obj),
                                   st.ctypes.data_as(C.POINTER(C.c_int32)), it.ctypes.data_as(C.POINTER(C.c_int32)), _dp(kkt), nthreads)
    assert rc == 0
    return dict(x=w, f=obj, status=st, iters=it, kkt=kkt)


def eval_batch(cfg: Config, p: np.ndarray, w: np.ndarray):
    L = lib()
    p = np.ascontiguousarray(p, dtype=np.float64); w = np.ascontiguousarray(w, dtype=np.float64)
    B = p.shape[0]
    ng = L.nmpc_oracle_n_g(C.byref(cfg))
    f = np.empty(B); g = np.empty((B, ng))
    L.nmpc_oracle_eval_batch(C.byref(cfg), B, _dp(p), _dp(w), _dp(f), _dp(g))
    return f, g


def shift_batch(cfg: Config, p: np.ndarray, w: np.ndarray, plant: bool = True):
    L = lib()
    p = np.ascontiguousarray(p, dtype=np.float64); w = np.ascontiguousarray(w, dtype=np.float64)
    B = p.shape[0]
    wn = np.empty_like(w); x0n = np.empty((B, 3 * cfg.m))
    L.nmpc_oracle_shift_batch(C.byref(cfg), B, _dp(p), _dp(w), _dp(wn), _dp(x0n) if plant else None)
    return wn, (x0n if plant else None)


def max_threads() -> int:
    return int(lib().nmpc_oracle_max_threads())


# ---- LIDAR-ray distance-state NMPC (oracle/lidar_oracle.c; V4 = AllScripts/obs_avoid_static_first_scenario_v4.py) ------------
class LidarCConfig(C.Structure):
    """mirror of nmpc_lidar_config_t (include/nmpc_lidar.h)."""
    _fields_ = [("N", C.c_int32), ("Nc", C.c_int32), ("R", C.c_int32), ("max_iter", C.c_int32), ("T", C.c_double),
                ("q", C.c_double * 3), ("r", C.c_double * 2), ("lw", C.c_double), ("tol", C.c_double), ("mu_init", C.c_double)]


_LLIB = None


def lidar_lib():
    global _LLIB
    if _LLIB is None:
        so = os.path.join(_HERE, "liblidar_oracle.so")
        src = os.path.join(_HERE, "lidar_oracle.c")
        hdrs = [os.path.join(_HERE, "..", "include", h) for h in ("nmpc_lidar.h", "nmpc_constants.h")]
        if not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(so) < max(os.path.getmtime(f) for f in [src] + hdrs)):
            subprocess.check_call(["make", "-C", _HERE, "-B", "liblidar_oracle.so"], stdout=subprocess.DEVNULL)
        _LLIB = C.CDLL(so)
        dp = C.POINTER(C.c_double); ip = C.POINTER(C.c_int32)
        _LLIB.nmpc_lidar_oracle_solve_batch.argtypes = [C.POINTER(LidarCConfig), dp, dp, C.c_int32, dp, dp, dp, dp, ip, ip, dp, C.c_int32]
        _LLIB.nmpc_lidar_oracle_solve_batch.restype = C.c_int32
        _LLIB.nmpc_lidar_oracle_eval_batch.argtypes = [C.POINTER(LidarCConfig), C.c_int32, dp, dp, dp, dp]
    return _LLIB


def lidar_config(cfg, tol=1e-8, mu_init=0.5, max_iter=2000) -> LidarCConfig:
    """oracle.lidar_ref.LidarConfig -> nmpc_lidar_config_t."""
    c = LidarCConfig()
    c.N, c.Nc, c.R, c.max_iter, c.T, c.lw, c.tol, c.mu_init = cfg.N, cfg.Nc, cfg.R, max_iter, cfg.T, cfg.lw, tol, mu_init
    c.q[:] = cfg.q; c.r[:] = cfg.r
    return c


def lidar_solve_batch(cfg, p, w0, max_iter=2000, nthreads=0, lbx=None, ubx=None):
    from . import lidar_ref as LR
    L = lidar_lib()
    cc = lidar_config(cfg, max_iter=max_iter)
    if lbx is None:
        lbx, ubx, _, _ = LR.bounds(cfg)
    lbx = np.ascontiguousarray(lbx, dtype=np.float64); ubx = np.ascontiguousarray(ubx, dtype=np.float64)
    p = np.ascontiguousarray(p, dtype=np.float64); w0 = np.ascontiguousarray(w0, dtype=np.float64)
    B = p.shape[0]
    assert p.shape == (B, cfg.n_p) and w0.shape == (B, cfg.n_var) and lbx.shape == (cfg.n_var,)
    w = np.empty((B, cfg.n_var)); obj = np.empty(B); kkt = np.empty(B)
    st = np.empty(B, dtype=np.int32); it = np.empty(B, dtype=np.int32)
    rc = L.nmpc_lidar_oracle_solve_batch(C.byref(cc), _dp(lbx), _dp(ubx), B, _dp(p), _dp(w0), _dp(w), _dp(obj),
                                         st.ctypes.data_as(C.POINTER(C.c_int32)), it.ctypes.data_as(C.POINTER(C.c_int32)), _dp(kkt), nthreads)
    assert rc == 0, rc
    return dict(x=w, f=obj, status=st, iters=it, kkt=kkt)


def lidar_eval_batch(cfg, p, w):
    L = lidar_lib()
    cc = lidar_config(cfg)
    p = np.ascontiguousarray(p, dtype=np.float64); w = np.ascontiguousarray(w, dtype=np.float64)
    B = p.shape[0]
    f = np.empty(B); g = np.empty((B, cfg.n_g))
    L.nmpc_lidar_oracle_eval_batch(C.byref(cc), B, _dp(p), _dp(w), _dp(f), _dp(g))
    return f, g
