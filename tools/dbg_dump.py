"""dev aid: iterates after k interior-point iterations (max_iter=k) for 256 six-robot instances, GPU vs the C oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nmpc_amd
from oracle import nlp_ref as R, oracle_lib as O
from tests import helpers as Hh
tag = sys.argv[1]
ocfg = R.cfg_six(20)
P, W0 = Hh.batch(ocfg, 256, 2)
out = {}
for mi in (1, 2, 4, 5, 6, 7, 8, 12, 16):
    s = nmpc_amd.NmpcSolver(Hh.to_product_cfg(ocfg, max_iter=mi), max_batch=256)
    x = s.solve_batch(P, W0)["x"].cpu().numpy()
    out["x%d" % mi] = x
    ref = O.solve_batch(O.make_config(ocfg, max_iter=mi), P, W0)["x"]
    d = np.abs(x - ref).max(axis=1)
    print(tag, "iters", mi, "max |gpu-oracle|", d.max(), "n>1e-9:", int((d > 1e-9).sum()), "worst", np.argsort(d)[-4:], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/dump_%s.npz" % tag, **out)
