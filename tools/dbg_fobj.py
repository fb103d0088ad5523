"""dev aid: reported objective vs objective evaluated at the returned iterate (nmpc_eval_batch), per instance."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nmpc_amd
from oracle import nlp_ref as R
from tests import helpers as Hh
ocfg = R.cfg_six(20); B = 64
P, W0 = Hh.batch(ocfg, B, 2)
for mi in (0, 1, 2, 400):
    s = nmpc_amd.NmpcSolver(Hh.to_product_cfg(ocfg, max_iter=mi), max_batch=B)
    r = s.solve_batch(P, W0)
    f, g = s.eval_batch(P, r["x"])
    d = (r["f"] - f).cpu().numpy()
    x0 = P[:, :18].reshape(B, 6, 3); xs = P[:, 18:].reshape(B, 6, 3)
    st0 = ((x0 - xs) ** 2 * np.array([1, 5, 0.1])).sum(axis=(1, 2))
    print("max_iter", mi, "f_reported - f_eval:", d[:6].round(6), " stage-0 cost:", st0[:6].round(6), " ratio", (d[:6] / st0[:6]).round(4))
s = nmpc_amd.NmpcSolver(Hh.to_product_cfg(ocfg, max_iter=1), max_batch=B)
r = s.solve_batch(P, W0)
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/fobj.npz", x=r["x"].cpu().numpy(), f=r["f"].cpu().numpy(), P=P)
