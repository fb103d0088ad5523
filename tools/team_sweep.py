"""Development aid (GPU box): cold-start throughput of every team size 1..10 (N=20, the six-robot script's parameters, B=4096 and 16384) — a sanity
sweep: the rate should fall monotonically with the team size (round 4: the four-robot kernels had slipped to one wave per SIMD).   python tools/team_sweep.py [B ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nmpc_amd
from oracle import nlp_ref as R
from tests import helpers as Hh
Bs = [int(a) for a in sys.argv[1:]] or [4096]
for B in Bs:
    for m in range(1, 11):
        oc = R.NLPConfig(m=m, N=20, T=0.3, dmin=0.4, v_max=0.15, w_max=1.5, pad_rows=(m > 1))
        P, W0 = Hh.batch(oc, B, 9)
        s = nmpc_amd.NmpcSolver(Hh.to_product_cfg(oc, max_iter=2000), max_batch=B)
        Pd = torch.as_tensor(P, device="cuda"); Wd = torch.as_tensor(W0, device="cuda")
        r = s.solve_batch(Pd, Wd); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t = time.perf_counter(); r = s.solve_batch(Pd, Wd); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        it = r["iters"].cpu().numpy()
        print("m=%2d B=%5d kernel %d: %8.3f ms %9.0f solves/s  %7.1f us per iteration of the mean solve x B/1024  iters mean %.1f max %d  converged %.4f" %
              (m, B, s.kernel_for_batch(B), 1e3 * min(ts), B / min(ts), 1e6 * min(ts) / (it.mean() * B / 1024), it.mean(), it.max(), (r["status"].cpu().numpy() == 0).mean()), flush=True)
        del s
