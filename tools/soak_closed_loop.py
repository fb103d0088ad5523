"""dev aid: long closed-loop soak on the GPU — failed solves, iteration tail, arrival / collision / deadlock statistics."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nmpc_amd
from oracle import nlp_ref as R
from tests import helpers as Hh
name = sys.argv[1] if len(sys.argv) > 1 else "six"; B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 120
def _composite():
    rng = np.random.default_rng(7)
    c = R.cfg_six(25); c.rob_dim = 0.2; c.margin = 0.1
    c.obstacles = [(float(x), float(y), float(r)) for x, y, r in zip(rng.uniform(-1.5, 1.5, 8), rng.uniform(-1.5, 1.5, 8), rng.uniform(0.125, 0.2, 8))]
    return c
ocfg = {"six": R.cfg_six(20), "two": R.cfg_two(20), "ten": R.cfg_ten(20), "obs3": R.cfg_obs3(20), "composite": _composite()}[name]
P, _ = Hh.batch(ocfg, B, 2)
s = nmpc_amd.NmpcSolver(Hh.to_product_cfg(ocfg, max_iter=2000), max_batch=B)
t = time.time()
ep = nmpc_amd.simulate_closed_loop(s, P[:, : ocfg.nx], P[:, ocfg.nx:], max_steps=steps, stop_tol=0.3, on_failure=os.environ.get("ON_FAILURE", "apply"), order_hint=os.environ.get("ORDER_HINT", "1") == "1")
dt = time.time() - t
print(f"{name} B={B} steps={ep.steps}: {ep.total_solves / dt:.0f} solves/s, failed solves {ep.failed_solves}/{ep.total_solves}, "
      f"arrived {ep.arrived.mean():.3f}, collision-free {ep.collision_free.mean():.4f}, deadlocked {ep.deadlocked.mean():.3f}, "
      f"min pair distance {ep.min_pair_distance.min():.6f} (dmin {ocfg.dmin}), mean iters first/last step {ep.mean_iters_by_step[0]:.1f}/{ep.mean_iters_by_step[-1]:.1f}")
