"""Builds libnmpc_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.environ.get("NMPC_LIBDIR") or os.path.join(ROOT, "lib")      # short in-tree path (see _lib.SO_PATH); NMPC_LIBDIR: build somewhere else (the forced-build test: the shared in-tree artefact is not replaced under running processes)
SO = os.path.join(LIBDIR, "libnmpc_hip.so")
INFO = os.path.join(LIBDIR, "build_info.json")    # what the last build() call did: compiled or reused, and the source hash
SOURCES = ["nmpc_kernels.hip", "nmpc_solve_lds.hip", "nmpc_solve_col.hip", "nmpc_lidar.hip", "nmpc_api.cpp"]
# per-source code generation switches.  nmpc_lidar.hip: machine-level loop-invariant code motion hoists the 64-bit literals of log() / sincos()
# out of the stage loops into vector registers, the allocator then spills them and reloads each one from scratch, with a full vmcnt wait, at
# every use (496 scratch loads in the kernel, 6 per log); without the pass they are re-materialised where used (114).  nmpc_solve_col.hip:
# same pass, measured A/B in one session (round 3): two robots +2.3 %, six +3.7 % (B = 16384: +5.2 %), composite +2.3 %, ten +-0.
FILE_FLAGS = {"nmpc_lidar.hip": os.environ.get("NMPC_LIDAR_FLAGS", "-mllvm -disable-machine-licm").split(),
              "nmpc_solve_col.hip": os.environ.get("NMPC_COL_FLAGS", "-mllvm -disable-machine-licm").split()}
# compile units: (source, object name, extra flags).  The column-per-lane kernel is compiled in three parts (team sizes 1..5, 6..8, 9..10:
# ~120 s each instead of ~350 s in one unit; see NMPC_COL_PART in the source); with NMPC_COL_ONLY_M (development) in one.
UNITS = [(s, s.rsplit(".", 1)[0], []) for s in SOURCES if s != "nmpc_solve_col.hip"] \
    + [("nmpc_solve_col.hip", "nmpc_solve_col_p%d" % k, ["-DNMPC_COL_PART=%d" % k]) for k in (1, 2, 3)]
DEPS = SOURCES + ["nmpc_device.h", "nmpc_solve_common.h"] + [os.path.join("..", "..", "include", h) for h in ("nmpc.h", "nmpc_lidar.h", "nmpc_constants.h", "nmpc_debug.h")]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libnmpc_hip.so cannot be built")


def source_hash() -> str:
    """sha256 (first 16 hex digits) over the sources and headers the library is built from, plus the build flags that change
    the code; it is compiled into the library (nmpc_version()) so a stale or foreign binary is recognisable."""
    import hashlib
    h = hashlib.sha256()
    for d in DEPS:
        with open(os.path.join(CSRC, d), "rb") as f:
            h.update(d.encode()); h.update(f.read())
    for k in ("NMPC_OPT", "NMPC_PROFILE", "NMPC_POISON", "NMPC_COL_ONLY_M", "NMPC_SHIFT_ESCALATION", "NMPC_EXTRA_DEFS"):
        h.update((k + "=" + os.environ.get(k, "")).encode())
    h.update(repr(sorted(FILE_FLAGS.items())).encode())
    h.update(repr(UNITS).encode())      # per-source code generation switches (defaults and their overrides)
    return h.hexdigest()[:16]


def built_hash() -> str:
    """the src= tag inside the existing library ('' when absent); read from the file, the library is not loaded."""
    if not os.path.exists(SO):
        return ""
    import re
    with open(SO, "rb") as f:
        m = re.search(rb"one wavefront per swarm instance\) src=([0-9a-f]{16})", f.read())
    return m.group(1).decode() if m else ""


def needs_build() -> bool:
    """True when the library is missing or was built from other sources.  Where the sources are not available (they always
    travel with the repository) the existing library is used as it is."""
    if not os.path.exists(SO):
        return True
    if not all(os.path.exists(os.path.join(CSRC, d)) for d in DEPS):
        return False
    return built_hash() != source_hash()


def _record(mode: str, seconds: float) -> None:
    import json
    import time
    os.makedirs(LIBDIR, exist_ok=True)
    info = {"mode": mode, "src_hash": source_hash(), "library": SO, "seconds": round(seconds, 2), "unix_time": time.time(),
            "hipcc": _hipcc() if mode == "compiled" else None}
    with open(INFO, "w") as f:
        json.dump(info, f)
    print("[nmpc build] %s libnmpc_hip.so src=%s (%.1f s)" % (mode, info["src_hash"], seconds), flush=True)


def last_build_info() -> dict:
    """what the last build() call in this tree did ({} if none was recorded)"""
    import json
    try:
        with open(INFO) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def build(force: bool = False, verbose: bool = False) -> str:
    import time
    t0 = time.time()
    if not force and not needs_build():
        _record("reused", time.time() - t0)
        return SO
    os.makedirs(LIBDIR, exist_ok=True)
    flags = ["--offload-arch=gfx950", os.environ.get("NMPC_OPT", "-O3"), "-std=c++17", "-fPIC", '-DNMPC_SRC_HASH="%s"' % source_hash()]
    if os.environ.get("NMPC_PROFILE"):
        flags.append("-DNMPC_PROFILE")
    flags += os.environ.get("NMPC_EXTRA_DEFS", "").split()      # development: extra -D switches of experiments (A/B)
    if os.environ.get("NMPC_POISON"):          # debug: uninitialised-read hunt, e.g. NMPC_POISON='__builtin_nan("")' or 1e30
        flags.append("-DNMPC_POISON=" + os.environ["NMPC_POISON"])
    if os.environ.get("NMPC_SHIFT_ESCALATION"):       # development (A/B)
        flags.append("-DNMPC_SHIFT_ESCALATION=" + os.environ["NMPC_SHIFT_ESCALATION"])
    only = os.environ.get("NMPC_COL_ONLY_M")   # development: instantiate the column kernel for one team size only (fast rebuilds)
    if only:
        flags.append("-DNMPC_COL_ONLY_M=" + only)
    # one hipcc -c per source, in parallel (the solve kernels dominate the build time), then link
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.environ.get("NMPC_OBJDIR") or os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    skip = set(filter(None, os.environ.get("NMPC_REUSE_OBJ", "").split(",")))   # development: keep the objects of unchanged sources

    units = [(s, s.rsplit(".", 1)[0], []) for s in SOURCES] if only else UNITS

    def cc(unit):
        src, name, extra = unit
        obj = os.path.join(objdir, name + ".o")
        if src in skip and os.path.exists(obj):
            return obj
        cmd = [_hipcc()] + flags + FILE_FLAGS.get(src, []) + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
        return obj
    # longest units first; no more compilers at once than cores
    order = sorted(units, key=lambda u: {"nmpc_solve_col.hip": 0, "nmpc_solve_lds.hip": 1, "nmpc_kernels.hip": 2}.get(u[0], 3))
    with ThreadPoolExecutor(max_workers=max(1, min(len(order), os.cpu_count() or 1))) as ex:
        objs = list(ex.map(cc, order))
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs, cwd=CSRC)
    _record("compiled", time.time() - t0)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
