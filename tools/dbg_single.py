import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nmpc_amd
from oracle import nlp_ref as R
from tests import helpers as Hh
ocfg = R.NLPConfig(m=6, N=35, T=0.1, dmin=0.3, v_max=0.22, w_max=2.84)
P, W0 = Hh.batch(ocfg, 8, 3)
s = nmpc_amd.NmpcSolver(Hh.to_product_cfg(ocfg, max_iter=300), max_batch=8)
r = s.solve_batch(P, W0); print("batch of 8:", r["status"].cpu().numpy(), r["iters"].cpu().numpy())
for i in range(8):
    r = s.solve_batch(P[i:i+1], W0[i:i+1]); print("alone", i, r["status"].cpu().numpy(), r["iters"].cpu().numpy(), "%.2e" % float(r["kkt"][0]))
r = s.solve_batch(P[::-1].copy(), W0[::-1].copy()); print("reversed:", r["status"].cpu().numpy(), r["iters"].cpu().numpy())
