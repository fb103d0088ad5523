"""Development aid (GPU box): launch time of small-team configurations on library variants (NMPC_SO), cold start, throughput shape.
    NMPC_SO=variants/libnmpc_x.so python tools/ab_small.py [B]      ->  one / three robots, three robots + 3 obstacles (N=20), two robots N=60"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nmpc_amd
from oracle import nlp_ref as R
from tests import helpers as Hh
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
three = R.NLPConfig(m=3, N=20, T=0.3, dmin=0.4, v_max=0.15, w_max=1.5)
four = R.NLPConfig(m=4, N=20, T=0.3, dmin=0.4, v_max=0.15, w_max=1.5)
four_thb = R.NLPConfig(m=4, N=20, T=0.3, dmin=0.4, v_max=0.15, w_max=1.5, th_max=2 * np.pi)
cases = [("one", R.cfg_one(20)), ("three", three), ("obs3", R.cfg_obs3(20)), ("two N=60", R.cfg_two(60)), ("four", four), ("four thb", four_thb), ("four N=40", R.NLPConfig(m=4, N=40, T=0.3, dmin=0.4, v_max=0.15, w_max=1.5))]
if os.environ.get("AB_CASES"):
    cases = [c for c in cases if c[0] in os.environ["AB_CASES"].split(",")]
for name, oc in cases:
    P, W0 = Hh.batch(oc, B, 7)
    s = nmpc_amd.NmpcSolver(Hh.to_product_cfg(oc, max_iter=2000), max_batch=B, kernel=3)
    Pd = torch.as_tensor(P, device="cuda"); Wd = torch.as_tensor(W0, device="cuda")
    r = s.solve_batch(Pd, Wd); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t = time.perf_counter(); r = s.solve_batch(Pd, Wd); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    it = r["iters"].cpu().numpy(); f = r["f"].cpu().numpy()
    print("AB %-9s %s B=%d: %.3f ms, %.0f solves/s, iters mean %.2f max %d, converged %.4f, sum f %.9e, lds %d" %
          (name, os.path.basename(os.environ.get("NMPC_SO", "lib")), B, 1e3 * min(ts), B / min(ts), it.mean(), it.max(), (r["status"].cpu().numpy() == 0).mean(), f[np.isfinite(f)].sum(),
           s.lib.nmpc_query(s._h, 3, B) if hasattr(s.lib, "nmpc_query") else -1), flush=True)
