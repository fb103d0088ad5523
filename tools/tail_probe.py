"""Development aid (GPU box): how much of the six-robot launch is the single longest solve?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nmpc_amd, bench
from tests import helpers as Hh
from tests import helpers as Hh
ocfg, B, P, W0 = Hh.bench_batch("six")
cfg = Hh.to_product_cfg(ocfg)
os.environ["NMPC_KERNEL"] = os.environ.get("TAIL_KERNEL", "3")
s = nmpc_amd.NmpcSolver(cfg, max_batch=B)
def run(P, W0, tag):
    dP = torch.as_tensor(P, device="cuda"); dW = torch.as_tensor(W0, device="cuda")
    s.solve_batch(dP, dW); torch.cuda.synchronize()
    t = time.perf_counter(); r = s.solve_batch(dP, dW); torch.cuda.synchronize(); dt = time.perf_counter() - t
    it = r["iters"].cpu().numpy()
    print(f"{tag:40s} B={len(P):5d} {dt*1e3:7.2f} ms  {len(P)/dt:9.0f} solves/s  max iters {it.max()}  mean {it.mean():.1f}", flush=True)
    return it
it = run(P, W0, "bench batch (instance 0 = literal swap)")
run(P[:1], W0[:1], "literal swap alone")
o = np.argsort(-it)
run(P[o], W0[o], "sorted longest first")
keep = it <= 70
run(P[keep], W0[keep], "without the 44 solves above 70 iterations")
keep = it <= 50
run(P[keep], W0[keep], "without solves above 50 iterations")
