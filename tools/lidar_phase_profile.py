"""Development aid (GPU box): cycles per phase of the LIDAR solve kernel.  Needs a library built with NMPC_EXTRA_DEFS=-DNMPC_LIDAR_PROFILE
(the kernel then returns its clock64() totals in the first entries of w_out — results of such a build are NOT solutions).
  NMPC_EXTRA_DEFS=-DNMPC_LIDAR_PROFILE python tools/lidar_phase_profile.py [B ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nmpc_amd
from oracle import lidar_ref as LR
NAMES = ["(iteration start)", "A optimality error", "B0 condensed blocks", "B Riccati", "C forward recursion", "C' pose/ray step, eta+", "adjoint recursion",
         "D step lengths", "E line search", "G accept"]
lc = LR.lidar_v4()
lbx, ubx, _, _ = LR.bounds(lc)
for B in [int(a) for a in sys.argv[1:]] or [64, 2048]:
    rng = np.random.Generator(np.random.PCG64(20210146))
    Pl, Wl = [], []
    for _ in range(B):
        pose = np.array([rng.uniform(0.0, 0.15), rng.uniform(0.0, 0.15), rng.uniform(0.4, 1.1)])
        world = [(float(rng.uniform(0.8, 2.6)), float(rng.uniform(0.3, 2.4)), float(rng.uniform(0.15, 0.3))) for _ in range(3)]
        scan = LR.scan_of_world(pose, world, lc.R)
        Pl.append(LR.make_p(lc, pose, np.array([3.0, 2.5, 0.0]) + rng.uniform(-0.3, 0.3, 3), scan)); Wl.append(LR.cold_start(lc, np.concatenate([pose, scan])))
    Pl = torch.as_tensor(np.stack(Pl), device="cuda"); Wl = torch.as_tensor(np.stack(Wl), device="cuda")
    s = nmpc_amd.LidarSolver(nmpc_amd.lidar_v4(), lbx=lbx, ubx=ubx, max_batch=B)
    r = s.solve_batch(Pl, Wl); torch.cuda.synchronize()
    it = r["iters"].cpu().numpy().astype(float)
    prof = r["x"][:, :12].cpu().numpy()
    per_it = prof.sum(0) / it.sum()
    print(f"B={B}: mean iters {it.mean():.1f}; cycles per iteration {per_it.sum():.0f}")
    for n, v in zip(NAMES, per_it):
        print(f"   {n:28s} {v:10.0f}  {100 * v / per_it.sum():5.1f} %")
