"""Development aid (GPU box): first iteration at which the HIP solve of a captured input (tools/soak_capture.py) leaves the oracle's path:
state after exactly k iterations (max_iter = k) on both sides.   python tools/dbg_capture_iterk.py <npz> <workload> k0 k1 [step]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nmpc_amd
from oracle import nlp_ref as R
from oracle import oracle_lib as O
from tests import helpers as Hh
z = np.load(sys.argv[1]); name = sys.argv[2]; k0, k1 = int(sys.argv[3]), int(sys.argv[4]); st = int(sys.argv[5]) if len(sys.argv) > 5 else 1
def _composite():
    rng = np.random.default_rng(7)
    c = R.cfg_six(25); c.rob_dim = 0.2; c.margin = 0.1
    c.obstacles = [(float(x), float(y), float(r)) for x, y, r in zip(rng.uniform(-1.5, 1.5, 8), rng.uniform(-1.5, 1.5, 8), rng.uniform(0.125, 0.2, 8))]
    return c
ocfg = {"six": R.cfg_six(20), "two": R.cfg_two(20), "ten": R.cfg_ten(20), "composite": _composite()}[name]
p, w = z["p"][:1], z["w"][:1]
for k in range(k0, k1 + 1, st):
    s = nmpc_amd.NmpcSolver(Hh.to_product_cfg(ocfg, max_iter=k), max_batch=1, kernel=os.environ.get("NMPC_KERNEL"))
    r = {a: v.cpu().numpy() for a, v in s.solve_batch(p, w).items()}
    ref = O.solve_batch(O.make_config(ocfg, max_iter=k), p, w)
    print("k=%4d  |dx| %.2e  kkt hip %.3e ora %.3e  it %d/%d st %d/%d  f hip %.9g ora %.9g" % (k, np.abs(r["x"][0] - ref["x"][0]).max(), r["kkt"][0], ref["kkt"][0], r["iters"][0], ref["iters"][0],
          r["status"][0], ref["status"][0], r["f"][0], ref["f"][0]), flush=True)
