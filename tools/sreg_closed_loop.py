"""Development aid (CPU only): warm closed-loop statistics of the CPU oracle under an inertia-correction variant (see tools/sreg_experiment.py).
    NMPC_ORACLE_STAGE_REG=.. python tools/sreg_closed_loop.py [B] [periods] [workloads...]
"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import helpers as Hh
from oracle import oracle_lib as O
L = O.lib(); L.nmpc_oracle_stats.argtypes = [C.POINTER(C.c_double), C.c_int]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for name in (sys.argv[3:] or ["six", "composite", "ten20", "two"]):
    ocfg, _, P, W = Hh.bench_batch(name, B)
    oc = O.make_config(ocfg, max_iter=2000)
    st4 = (C.c_double * 4)(); L.nmpc_oracle_stats(st4, 1)
    its = []; fails = 0; t = time.time()
    for k in range(T):
        r = O.solve_batch(oc, P, W)
        its.append(r["iters"].copy()); fails += int((r["status"] != 0).sum())
        W, xn = O.shift_batch(oc, P, r["x"])
        P = np.concatenate([xn, P[:, ocfg.nx:]], axis=1)
    L.nmpc_oracle_stats(st4, 1)
    its = np.array(its)
    print("SREG=%s %-9s B=%d x %d periods: sweeps/iter %.3f | warm iters mean %.2f p99 %.0f max %d | sum of per-period max %d | failed %d [%.1fs]" % (
        os.environ.get("NMPC_ORACLE_STAGE_REG", "shipped"), name, B, T, st4[0] / max(st4[3], 1), its[1:].mean(), np.percentile(its[1:], 99), its[1:].max(), its.max(axis=1).sum(), fails, time.time() - t), flush=True)
