"""Development aid (GPU box): re-solve captured inputs (tools/soak_capture.py) with every kernel pin; oracle alongside.
   python tools/dbg_capture.py gpurun_out/soak_capture_composite.npz composite"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nmpc_amd
from oracle import nlp_ref as R
from oracle import oracle_lib as O
from tests import helpers as Hh
z = np.load(sys.argv[1]); name = sys.argv[2] if len(sys.argv) > 2 else "composite"
def _composite():
    rng = np.random.default_rng(7)
    c = R.cfg_six(25); c.rob_dim = 0.2; c.margin = 0.1
    c.obstacles = [(float(x), float(y), float(r)) for x, y, r in zip(rng.uniform(-1.5, 1.5, 8), rng.uniform(-1.5, 1.5, 8), rng.uniform(0.125, 0.2, 8))]
    return c
ocfg = {"six": R.cfg_six(20), "two": R.cfg_two(20), "ten": R.cfg_ten(20), "composite": _composite()}[name]
p, w = z["p"], z["w"]
ref = O.solve_batch(O.make_config(ocfg, max_iter=2000), p, w)
print("oracle", ref["status"], ref["iters"], ref["kkt"])
for kern in (None, "3", "4", "5", "2"):
    try:
        s = nmpc_amd.NmpcSolver(Hh.to_product_cfg(ocfg, max_iter=2000), max_batch=max(1, p.shape[0]), kernel=kern)
        r = s.solve_batch(p, w); torch.cuda.synchronize()
        print("kernel", kern, "status", r["status"].cpu().numpy(), "iters", r["iters"].cpu().numpy(), "kkt", r["kkt"].cpu().numpy(), "max|x - x_oracle|", float(np.abs(r["x"].cpu().numpy() - ref["x"]).max()))
    except Exception as e:
        print("kernel", kern, "failed:", e)
