"""Development aid: the six-robot N=88 case of test_longest_lds_horizon_in_every_launch_shape on every kernel against the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nmpc_amd
from oracle import nlp_ref as R, oracle_lib as O
from tests import helpers as Hh
N = int(sys.argv[1]) if len(sys.argv) > 1 else 88
ocfg = R.cfg_six(N)
P, W0 = Hh.batch(ocfg, 2, 2)
ref = O.solve_batch(O.make_config(ocfg, max_iter=800), P, W0)
print("oracle iters", ref["iters"], "f", ref["f"], "status", ref["status"])
for kern in ("1", "2", "3", "4"):
    os.environ["NMPC_KERNEL"] = kern
    try:
        s = nmpc_amd.NmpcSolver(Hh.to_product_cfg(ocfg, max_iter=800), max_batch=2)
        r = {k: v.cpu().numpy() for k, v in s.solve_batch(P, W0).items()}
        print("kernel", kern, "->", s.kernel_for_batch(2), "iters", r["iters"], "f", r["f"], "status", r["status"], "max|dw|", np.abs(r["x"] - ref["x"]).max(axis=1))
    except Exception as e:
        print("kernel", kern, "failed:", e)
for mi in (5, 10, 15, 20, 25, 30, 40, 50):
    refk = O.solve_batch(O.make_config(ocfg, max_iter=mi), P, W0)
    for kern in ("2", "1"):
        os.environ["NMPC_KERNEL"] = kern
        s = nmpc_amd.NmpcSolver(Hh.to_product_cfg(ocfg, max_iter=mi), max_batch=2)
        r = {k: v.cpu().numpy() for k, v in s.solve_batch(P, W0).items()}
        print("max_iter", mi, "kernel", kern, "max|dw|", np.abs(r["x"] - refk["x"]).max(axis=1), "kkt hip", r["kkt"], "oracle", refk["kkt"])
