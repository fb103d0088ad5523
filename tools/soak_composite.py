"""Development aid (GPU box): closed-loop soak of the six-robot + eight-obstacle composite (BASELINE config 5): B swarms x steps control
periods on the HIP path; reports failed solves by status.   python tools/soak_composite.py [B] [steps] [seed index, default 4]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nmpc_amd
from oracle import nlp_ref as R
from tests import helpers as Hh
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rng = np.random.default_rng(7)
c = R.cfg_six(25); c.rob_dim = 0.2; c.margin = 0.1
c.obstacles = [(float(x), float(y), float(r)) for x, y, r in zip(rng.uniform(-1.5, 1.5, 8), rng.uniform(-1.5, 1.5, 8), rng.uniform(0.125, 0.2, 8))]
P, W = Hh.batch(c, B, seed)
s = nmpc_amd.NmpcSolver(Hh.to_product_cfg(c, max_iter=2000), max_batch=B)
p = torch.as_tensor(P, device="cuda"); w = torch.as_tensor(W, device="cuda")
fails = {}; tot = 0; t = time.perf_counter(); mx = 0; bad_p, bad_w, bad_i = [], [], []
for step in range(steps):
    r = s.solve_batch(p, w)
    st = r["status"].cpu().numpy(); tot += B; mx = max(mx, int(r["iters"].max()))
    for k in np.unique(st[st != 0]): fails[int(k)] = fails.get(int(k), 0) + int((st == k).sum())
    for b in np.where((st != 0) & (st != 3))[0]:
        bad_p.append(p[b].cpu().numpy()); bad_w.append(w[b].cpu().numpy()); bad_i.append((step, int(b), int(st[b]), int(r['iters'][b])))
    w, x0n = s.shift_batch(p, r["x"], plant=True)
    p = torch.cat([x0n, p[:, c.nx:]], dim=1)
torch.cuda.synchronize()
print(f"SOAK composite {B} swarms x {steps} periods (seed index {seed}): {tot} solves, failed by status {fails}, max iterations {mx}, {tot/(time.perf_counter()-t):.0f} solves/s")
if bad_p:
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez("gpurun_out/soak_comp_bad_%d.npz" % seed, p=np.array(bad_p), w=np.array(bad_w), info=np.array(bad_i))
    print("SOAK failing inputs:", bad_i)
