"""ctypes binding of the C ABI in include/nmpc.h (libnmpc_hip.so).  No CPU fallback exists:
if the library is missing or no GPU is visible the product path raises."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
# the library lives at the short in-tree path <repo>/lib/libnmpc_hip.so (the package directory's own name is ~115 characters long;
# tools that record which shared objects a process mapped truncate such paths).  NMPC_SO: development (A/B builds in one GPU session)
SO_PATH = os.environ.get("NMPC_SO") or os.path.join(ROOT, "lib", "libnmpc_hip.so")

NMPC_MAX_ROBOTS = 10
NMPC_MAX_OBSTACLES = 8

STATUS_NAMES = {0: "Solve_Succeeded", 1: "Maximum_Iterations_Exceeded", 2: "Error_In_Step_Computation",
                3: "Infeasible_Problem_Detected", 4: "Restoration_Failed"}
ERRORS = {-1: "NMPC_E_ARG", -2: "NMPC_E_UNSUPPORTED", -3: "NMPC_E_HIP", -4: "NMPC_E_NOMEM"}

LIDAR_EXPORTS = ["nmpc_lidar_n_var", "nmpc_lidar_n_g", "nmpc_lidar_n_p", "nmpc_lidar_create", "nmpc_lidar_destroy", "nmpc_lidar_solve_batch",
                 "nmpc_lidar_eval_batch", "nmpc_lidar_shift_batch", "nmpc_lidar_scan_batch", "nmpc_lidar_plant_batch"]
DEBUG_EXPORTS = ["nmpc_debug_profile", "nmpc_debug_trace", "nmpc_debug_trace2", "nmpc_debug_workspace"]      # include/nmpc_debug.h
QUERY_KERNEL_FOR_BATCH, QUERY_WORKSPACE_BYTES, QUERY_LDS_BYTES, QUERY_MAX_BATCH, QUERY_KERNEL_FOR_ORDERED_BATCH = 1, 2, 3, 4, 5      # NMPC_QUERY_* of include/nmpc.h
EXPORTS = ["nmpc_n_var", "nmpc_n_g", "nmpc_n_p", "nmpc_config_default", "nmpc_create", "nmpc_create_opts", "nmpc_query", "nmpc_destroy",
           "nmpc_workspace_bytes", "nmpc_solve_batch", "nmpc_solve_batch_ordered", "nmpc_step_batch", "nmpc_eval_batch", "nmpc_shift_batch", "nmpc_odometry_batch", "nmpc_version"]


class CConfig(C.Structure):
    """nmpc_config_t"""
    _fields_ = [("m", C.c_int32), ("N", C.c_int32), ("n_obs", C.c_int32), ("pad_rows", C.c_int32),
                ("T", C.c_double), ("dmin", C.c_double), ("q", C.c_double * 3), ("r", C.c_double * 2),
                ("v_max", C.c_double), ("w_max", C.c_double), ("xy_max", C.c_double), ("th_max", C.c_double),
                ("rob_dim", C.c_double), ("margin", C.c_double), ("pad_value", C.c_double),
                ("obs", C.c_double * (3 * NMPC_MAX_OBSTACLES)), ("tol", C.c_double), ("mu_init", C.c_double),
                ("max_iter", C.c_int32), ("pair_rows", C.c_int32)]


class COptions(C.Structure):
    """nmpc_options_t"""
    _fields_ = [("kernel", C.c_int32), ("trace_instance", C.c_int32)]


class CLidarConfig(C.Structure):
    """nmpc_lidar_config_t (include/nmpc_lidar.h)"""
    _fields_ = [("N", C.c_int32), ("Nc", C.c_int32), ("R", C.c_int32), ("max_iter", C.c_int32), ("T", C.c_double),
                ("q", C.c_double * 3), ("r", C.c_double * 2), ("lw", C.c_double), ("tol", C.c_double), ("mu_init", C.c_double)]


_lib = None


def load():
    """Loads libnmpc_hip.so; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(f"{SO_PATH} is missing: build it with __graft_entry__.build() "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    # Load order matters in a process that also holds PyTorch: torch ships its own libamdhip64 (soname
    # libamdhip64.so.7).  Importing torch first makes this library's DT_NEEDED libamdhip64.so.7 bind to that
    # already-loaded runtime; the other order would put two HIP runtimes in one process and the second fails
    # to initialise (hipGetDeviceCount -> error).
    import torch  # noqa: F401
    L = C.CDLL(SO_PATH)
    vp, i32 = C.c_void_p, C.c_int32
    cp = C.POINTER(CConfig)
    for f in ("nmpc_n_var", "nmpc_n_g", "nmpc_n_p"):
        getattr(L, f).argtypes = [cp]; getattr(L, f).restype = i32
    L.nmpc_config_default.argtypes = [cp, i32, i32]; L.nmpc_config_default.restype = None
    L.nmpc_create.argtypes = [cp, i32, C.POINTER(vp)]; L.nmpc_create.restype = i32
    L.nmpc_create_opts.argtypes = [cp, i32, C.POINTER(COptions), C.POINTER(vp)]; L.nmpc_create_opts.restype = i32
    L.nmpc_query.argtypes = [vp, i32, C.c_int64]; L.nmpc_query.restype = C.c_int64
    L.nmpc_debug_profile.argtypes = [vp, C.POINTER(C.c_int64), i32]; L.nmpc_debug_profile.restype = i32
    L.nmpc_debug_trace.argtypes = [vp, vp, i32]; L.nmpc_debug_trace.restype = i32
    L.nmpc_debug_trace2.argtypes = [vp, vp, i32]; L.nmpc_debug_trace2.restype = i32
    L.nmpc_debug_workspace.argtypes = [vp, i32, vp, C.c_int64, vp]; L.nmpc_debug_workspace.restype = C.c_int64
    L.nmpc_destroy.argtypes = [vp]; L.nmpc_destroy.restype = i32
    L.nmpc_workspace_bytes.argtypes = [vp]; L.nmpc_workspace_bytes.restype = C.c_int64
    L.nmpc_solve_batch.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]; L.nmpc_solve_batch.restype = i32
    L.nmpc_solve_batch_ordered.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]; L.nmpc_solve_batch_ordered.restype = i32
    L.nmpc_step_batch.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]; L.nmpc_step_batch.restype = i32
    L.nmpc_eval_batch.argtypes = [vp, i32, vp, vp, vp, vp, vp]; L.nmpc_eval_batch.restype = i32
    L.nmpc_shift_batch.argtypes = [vp, i32, vp, vp, vp, vp, vp]; L.nmpc_shift_batch.restype = i32
    L.nmpc_odometry_batch.argtypes = [C.c_int64, vp, vp, vp, i32, vp]; L.nmpc_odometry_batch.restype = i32
    L.nmpc_version.argtypes = []; L.nmpc_version.restype = C.c_char_p
    lp = C.POINTER(CLidarConfig)
    for f in ("nmpc_lidar_n_var", "nmpc_lidar_n_g", "nmpc_lidar_n_p"):
        getattr(L, f).argtypes = [lp]; getattr(L, f).restype = i32
    L.nmpc_lidar_create.argtypes = [lp, C.POINTER(C.c_double), C.POINTER(C.c_double), i32, C.POINTER(vp)]; L.nmpc_lidar_create.restype = i32
    L.nmpc_lidar_destroy.argtypes = [vp]; L.nmpc_lidar_destroy.restype = i32
    L.nmpc_lidar_solve_batch.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]; L.nmpc_lidar_solve_batch.restype = i32
    L.nmpc_lidar_eval_batch.argtypes = [vp, i32, vp, vp, vp, vp, vp]; L.nmpc_lidar_eval_batch.restype = i32
    L.nmpc_lidar_shift_batch.argtypes = [vp, i32, vp, vp, vp]; L.nmpc_lidar_shift_batch.restype = i32
    L.nmpc_lidar_scan_batch.argtypes = [C.c_int64, i32, i32, vp, vp, C.c_double, vp, vp]; L.nmpc_lidar_scan_batch.restype = i32
    L.nmpc_lidar_plant_batch.argtypes = [vp, i32, vp, vp, vp, i32, vp]; L.nmpc_lidar_plant_batch.restype = i32
    _lib = L
    return L


def describe() -> str:
    """version string of the loaded HIP library and the line of /proc/self/maps that shows it mapped into this process
    (printed by smoke() and the GPU test session so the run's record shows WHICH native code executed)."""
    L = load()
    line = ""
    try:
        with open("/proc/self/maps") as f:
            for ln in f:
                if "libnmpc_hip.so" in ln:
                    line = ln.strip()
                    break
    except OSError:
        pass
    return "%s | mapped: %s" % (L.nmpc_version().decode(), line or "<not found in /proc/self/maps>")


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {ERRORS.get(rc, rc)}")
