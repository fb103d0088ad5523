"""dev aid: GPU vs oracle state after exactly k iterations (max_iter=k) for chosen instances."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nmpc_amd
from oracle import nlp_ref as R, oracle_lib as O
from tests import helpers as Hh
B = int(sys.argv[1]); insts = [int(a) for a in sys.argv[2].split(",")]; k0, k1 = int(sys.argv[3]), int(sys.argv[4])
ocfg = R.cfg_six(20)
P, W0 = Hh.batch(ocfg, B, 2)
for mi in range(k0, k1 + 1):
    s = nmpc_amd.NmpcSolver(Hh.to_product_cfg(ocfg, max_iter=mi), max_batch=B)
    r = {k: v.cpu().numpy() for k, v in s.solve_batch(P, W0).items()}
    ref = O.solve_batch(O.make_config(ocfg, max_iter=mi), P[insts], W0[insts])
    for n, i in enumerate(insts):
        print("k=%3d inst %4d  |dx| %.2e  kkt gpu %.3e ora %.3e  it %d/%d st %d/%d  obj diff %.2e" % (
            mi, i, np.abs(r["x"][i] - ref["x"][n]).max(), r["kkt"][i], ref["kkt"][n], r["iters"][i], ref["iters"][n],
            r["status"][i], ref["status"][n], r["f"][i] - ref["f"][n]), flush=True)
