"""dev aid: closed loop on the GPU for a named configuration; histogram of solve statuses / iteration tail per step, and the
inputs of the first non-converged solves (gpurun_out/soak_bad_<name>.npz)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nmpc_amd
from oracle import nlp_ref as R
from tests import helpers as Hh
import importlib.util
spec = importlib.util.spec_from_file_location("soak", os.path.join(os.path.dirname(__file__), "soak_closed_loop.py"))
name = sys.argv[1]; B = int(sys.argv[2]); steps = int(sys.argv[3])
def _composite():
    rng = np.random.default_rng(7)
    c = R.cfg_six(25); c.rob_dim = 0.2; c.margin = 0.1
    c.obstacles = [(float(x), float(y), float(r)) for x, y, r in zip(rng.uniform(-1.5, 1.5, 8), rng.uniform(-1.5, 1.5, 8), rng.uniform(0.125, 0.2, 8))]
    return c
ocfg = {"six": R.cfg_six(20), "ten": R.cfg_ten(20), "composite": _composite()}[name]
P, W = Hh.batch(ocfg, B, 2)
s = nmpc_amd.NmpcSolver(Hh.to_product_cfg(ocfg, max_iter=2000), max_batch=B)
p = torch.as_tensor(P, device="cuda"); w = torch.as_tensor(W, device="cuda")
hist = np.zeros(5, dtype=int); bad_p, bad_w, info = [], [], []
for step in range(steps):
    r = s.solve_batch(p, w)
    st = r["status"].cpu().numpy(); it = r["iters"].cpu().numpy()
    hist += np.bincount(st, minlength=5)
    for b in np.where(st != 0)[0][:3]:
        if len(bad_p) < 16:
            bad_p.append(p[b].cpu().numpy()); bad_w.append(w[b].cpu().numpy()); info.append((step, int(b), int(st[b]), int(it[b]), float(r["kkt"][b])))
    if step % 10 == 0:
        print("step", step, "status hist", np.bincount(st, minlength=5), "iters mean %.1f p99 %d max %d" % (it.mean(), np.percentile(it, 99), it.max()), flush=True)
    w, x0n = s.shift_batch(p, r["x"], plant=True)
    p = torch.cat([x0n, p[:, ocfg.nx:]], dim=1)
print("total status histogram [conv, max_iter, numeric, infeasible_x0, stalled]:", hist); print(info)
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/soak_bad_%s.npz" % name, P=np.array(bad_p), W=np.array(bad_w), info=np.array(info))
