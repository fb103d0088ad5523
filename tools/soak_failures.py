"""dev aid: closed loop on the GPU, saving the inputs (p, guess) of every solve that does not converge."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nmpc_amd
from oracle import nlp_ref as R
from tests import helpers as Hh
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
ocfg = R.cfg_six(20)
P, W = Hh.batch(ocfg, B, 2)
s = nmpc_amd.NmpcSolver(Hh.to_product_cfg(ocfg, max_iter=2000), max_batch=B)
p = torch.as_tensor(P, device="cuda"); w = torch.as_tensor(W, device="cuda")
bad_p, bad_w, bad_info = [], [], []
for step in range(steps):
    r = s.solve_batch(p, w)
    st = r["status"].cpu().numpy()
    for b in np.where(st != 0)[0]:
        bad_p.append(p[b].cpu().numpy()); bad_w.append(w[b].cpu().numpy()); bad_info.append((step, int(b), int(st[b]), int(r["iters"][b]), float(r["kkt"][b])))
    w, x0n = s.shift_batch(p, r["x"], plant=True)
    p = torch.cat([x0n, p[:, ocfg.nx:]], dim=1)
print(bad_info)
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/soak_bad.npz", P=np.array(bad_p), W=np.array(bad_w), info=np.array(bad_info))
