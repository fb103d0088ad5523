"""Development aid (GPU box): closed loop of a soak workload with plain solve + shift calls, capturing the inputs of every solve that does not
converge and re-solving them with the oracle on the host (is the failure the algorithm's or the kernel's?).
   python tools/soak_capture.py [six|two|ten|composite] [B] [steps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nmpc_amd
from oracle import nlp_ref as R
from oracle import oracle_lib as O
from tests import helpers as Hh
name = sys.argv[1] if len(sys.argv) > 1 else "composite"; B = int(sys.argv[2]) if len(sys.argv) > 2 else 512; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 120
def _composite():
    rng = np.random.default_rng(7)
    c = R.cfg_six(25); c.rob_dim = 0.2; c.margin = 0.1
    c.obstacles = [(float(x), float(y), float(r)) for x, y, r in zip(rng.uniform(-1.5, 1.5, 8), rng.uniform(-1.5, 1.5, 8), rng.uniform(0.125, 0.2, 8))]
    return c
ocfg = {"six": R.cfg_six(20), "two": R.cfg_two(20), "ten": R.cfg_ten(20), "composite": _composite()}[name]
P, W = Hh.batch(ocfg, B, 2)
s = nmpc_amd.NmpcSolver(Hh.to_product_cfg(ocfg, max_iter=2000), max_batch=B)
p = torch.as_tensor(P, device="cuda"); w = torch.as_tensor(W, device="cuda")
bad = []
OC = O.make_config(ocfg, max_iter=2000)
for step in range(steps):
    r = s.solve_batch(p, w)
    st = r["status"].cpu().numpy()
    for b in np.where(st != 0)[0]:
        bad.append((p[b].cpu().numpy(), w[b].cpu().numpy(), step, int(b), int(st[b]), int(r["iters"][b]), float(r["kkt"][b])))
    w, x0n = s.shift_batch(p, r["x"], plant=True)
    p = torch.cat([x0n, p[:, ocfg.nx:]], dim=1)
print(f"{name} B={B} steps={steps}: {len(bad)} solves did not converge of {B * steps}")
for pb, wb, step, b, st, it, kkt in bad:
    ref = O.solve_batch(OC, pb[None], wb[None])
    print(f"  period {step} swarm {b}: HIP status {st} after {it} iterations (kkt {kkt:.2e}); oracle on the same input: status {int(ref['status'][0])} after {int(ref['iters'][0])} iterations (kkt {float(ref['kkt'][0]):.2e})")
if bad:
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez("gpurun_out/soak_capture_%s.npz" % name, p=np.array([x[0] for x in bad]), w=np.array([x[1] for x in bad]), info=np.array([x[2:] for x in bad]))
