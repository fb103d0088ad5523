"""Development aid (CPU only): stage-local inertia correction on the CPU oracle.

    NMPC_ORACLE_STAGE_REG=0|1|2|3 [NMPC_ORACLE_SREG_BETA=..] python tools/sreg_experiment.py [B] [workloads...]

0 = global shift, whole backward sweep retried (round 1-3); 1 = shift delta_k I on the failing stage only, the stage redone from
P_{k+1}; 2 = rejected pivot d replaced by max(|d|, beta max(1, |d0|)) in place; 3 = rejected pivot d + delta with IPOPT's escalation
schedule run on the scalar pivot, delta remembered per (stage, control).  Prints per workload: stage factorisations per iteration in
sweep equivalents, mean / p99 / max iterations, converged fraction, the literal start/goal set's count (instance 0), the fraction of
instances that end at the point of variant 0 (1e-4) and the objective comparison where they do not.  Results: DESIGN.md.
"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(B, names):
    from tests import helpers as Hh
    from oracle import oracle_lib as O
    L = O.lib()
    L.nmpc_oracle_stats.argtypes = [C.POINTER(C.c_double), C.c_int]
    out = {}
    for name in names:
        nb = {"six": B, "two": B, "ten20": max(64, B // 4), "ten": max(64, B // 4), "composite": max(64, B // 2)}[name]
        ocfg, _, P, W0 = Hh.bench_batch(name, nb)
        st4 = (C.c_double * 4)()
        L.nmpc_oracle_stats(st4, 1)
        t = time.time(); r = O.solve_batch(O.make_config(ocfg, max_iter=2000), P, W0); dt = time.time() - t
        L.nmpc_oracle_stats(st4, 1)
        it = r["iters"]; st = r["status"]
        print("SREG=%s %-9s B=%4d: sweeps/iter %.3f repaired-stage frac %.3f | iters mean %.2f p50 %.0f p99 %.0f max %d inst0 %d | converged %.4f %s [%.1fs]" % (
            os.environ.get("NMPC_ORACLE_STAGE_REG", "shipped"), name, nb, st4[0] / max(st4[3], 1), st4[2] / max(st4[3], 1), it.mean(), np.percentile(it, 50), np.percentile(it, 99),
            it.max(), it[0], (st == 0).mean(), dict(zip(*[a.tolist() for a in np.unique(st, return_counts=True)])), dt), flush=True)
        out[name + "_x"] = r["x"]; out[name + "_f"] = r["f"]; out[name + "_it"] = it; out[name + "_st"] = st
    np.savez("/tmp/sreg_%s.npz" % os.environ.get("NMPC_ORACLE_STAGE_REG", "shipped"), **out)


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    names = sys.argv[2:] or ["six", "two", "ten20", "ten", "composite"]
    child(B, names)
    mode = os.environ.get("NMPC_ORACLE_STAGE_REG", "shipped")
    if mode != "0" and os.path.exists("/tmp/sreg_0.npz"):
        a = np.load("/tmp/sreg_0.npz"); b = np.load("/tmp/sreg_%s.npz" % mode)
        for name in names:
            if name + "_x" not in a or a[name + "_x"].shape != b[name + "_x"].shape:
                continue
            both = (a[name + "_st"] == 0) & (b[name + "_st"] == 0)
            same = np.max(np.abs(a[name + "_x"] - b[name + "_x"]), axis=1) <= 1e-4
            d = both & ~same
            rel = (b[name + "_f"][d] - a[name + "_f"][d]) / np.maximum(1.0, np.abs(a[name + "_f"][d]))
            print("   vs variant 0 %-9s: same point %.4f; elsewhere: variant %s lower on %.3f, median |rel gap| %.2e" % (
                name, same[both].mean(), mode, (rel < 0).mean() if d.any() else float("nan"), np.median(np.abs(rel)) if d.any() else float("nan")))
