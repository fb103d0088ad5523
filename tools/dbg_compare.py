"""Development aid: instance-by-instance comparison of the HIP solve with the CPU oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nmpc_amd
from oracle import nlp_ref as R, oracle_lib as O
from tests import helpers as Hh
name = sys.argv[1] if len(sys.argv) > 1 else "six"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
mi = int(sys.argv[3]) if len(sys.argv) > 3 else 300
ocfg = {"six": R.cfg_six(20), "two": R.cfg_two(20), "ten": R.cfg_ten(20), "one": R.cfg_one(20)}[name]
P, W0 = Hh.batch(ocfg, B, 2)
s = nmpc_amd.NmpcSolver(Hh.to_product_cfg(ocfg, max_iter=mi), max_batch=B)
r = {k: v.cpu().numpy() for k, v in s.solve_batch(P, W0).items()}
ref = O.solve_batch(O.make_config(ocfg, max_iter=mi), P, W0)
dw = np.max(np.abs(r["x"] - ref["x"]), axis=1)
print("status hip", np.bincount(r["status"], minlength=4), "oracle", np.bincount(ref["status"], minlength=4))
print("same basin frac", (dw <= 1e-6).mean(), " iters equal frac", (r["iters"] == ref["iters"]).mean())
print("iters hip mean %.2f oracle mean %.2f" % (r["iters"].mean(), ref["iters"].mean()))
bad = np.where(r["status"] != 0)[0]
print("non-converged:", bad[:10], "kkt", r["kkt"][bad[:10]], "oracle iters", ref["iters"][bad[:10]])
d = np.where(r["iters"] != ref["iters"])[0]
print("first differing:", d[:10], r["iters"][d[:10]], ref["iters"][d[:10]])
same = dw <= 1e-6
df = np.abs(r["f"] - ref["f"]) / np.maximum(1.0, np.abs(ref["f"]))
print("objective: max rel diff on same-basin instances %.2e, n > 1e-6: %d" % (df[same].max() if same.any() else 0.0, int((df[same] > 1e-6).sum())))
