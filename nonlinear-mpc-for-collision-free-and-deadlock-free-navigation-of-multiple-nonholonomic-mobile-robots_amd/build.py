"""Builds libnmpc_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libnmpc_hip.so")
SOURCES = ["nmpc_kernels.hip", "nmpc_solve_lds.hip", "nmpc_api.cpp"]
DEPS = SOURCES + ["nmpc_device.h", os.path.join("..", "..", "include", "nmpc.h")]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libnmpc_hip.so cannot be built")


def needs_build() -> bool:
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.exists(os.path.join(CSRC, d)) and os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return SO
    cmd = [_hipcc(), "--offload-arch=gfx950", os.environ.get("NMPC_OPT", "-O3"), "-std=c++17", "-fPIC", "-shared", "-o", SO] + SOURCES
    if os.environ.get("NMPC_PROFILE"):
        cmd.insert(1, "-DNMPC_PROFILE")
    if os.environ.get("NMPC_POISON"):          # debug: uninitialised-read hunt, e.g. NMPC_POISON='__builtin_nan("")' or 1e30
        cmd.insert(1, "-DNMPC_POISON=" + os.environ["NMPC_POISON"])
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
