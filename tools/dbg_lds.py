import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nmpc_amd
from oracle import nlp_ref as R, oracle_lib as O
from tests import helpers as Hh
for m, N in [(10, 8), (10, 12), (10, 14), (10, 20), (8, 20), (8, 30), (6, 35), (6, 60)]:
    ocfg = R.NLPConfig(m=m, N=N, T=0.1, dmin=0.3, v_max=0.22, w_max=2.84)
    P, W0 = Hh.batch(ocfg, 8, 3)
    s = nmpc_amd.NmpcSolver(Hh.to_product_cfg(ocfg, max_iter=300), max_batch=8)
    r = {k: v.cpu().numpy() for k, v in s.solve_batch(P, W0).items()}
    ref = O.solve_batch(O.make_config(ocfg, max_iter=300), P, W0)
    nx, nu, NP = 3 * m, 2 * m, m * (m - 1) // 2
    lds = 8 * (4 * (N + 1) * nx + 6 * N * nu + 2 * (N + 1) * NP + 4 * (N + 1) * 2 * m + 2 * N * m)
    print(m, N, "approx LDS KB %.0f" % (lds / 1024 + 30), "status", r["status"], "iters", r["iters"], "oracle", ref["iters"])
