"""Development aid: compare the two HIP kernels (NMPC_KERNEL=1 vs 2) after a fixed number of iterations."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nmpc_amd
from oracle import nlp_ref as R
from tests import helpers as Hh
name = sys.argv[1] if len(sys.argv) > 1 else "six"
B = 64
ocfg = {"six": R.cfg_six(20), "two": R.cfg_two(20), "ten": R.cfg_ten(20), "one": R.cfg_one(20), "three": R.NLPConfig(m=3, N=20, T=0.3, dmin=0.4, v_max=0.15, w_max=1.5), "eight": R.NLPConfig(m=8, N=8, T=0.1, dmin=0.3, v_max=0.22, w_max=2.84), "five": R.NLPConfig(m=5, N=8, T=0.1, dmin=0.3, v_max=0.22, w_max=2.84)}[name]
P, W0 = Hh.batch(ocfg, B, 2)
for mi in [int(a) for a in (sys.argv[2].split(",") if len(sys.argv) > 2 else "1,2,3,5,8,12,20".split(","))]:
    out = []
    for kern in ("1", "2"):
        os.environ["NMPC_KERNEL"] = kern
        s = nmpc_amd.NmpcSolver(Hh.to_product_cfg(ocfg, max_iter=mi), max_batch=B)
        out.append(s.solve_batch(P, W0)["x"].cpu().numpy())
    d = np.max(np.abs(out[0] - out[1]), axis=1)
    print("max_iter %2d: max|x1-x2| per instance: median %.2e max %.2e  argmax %d" % (mi, np.median(d), d.max(), d.argmax()))
