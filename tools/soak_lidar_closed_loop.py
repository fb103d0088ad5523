"""Development aid (GPU box): LIDAR closed loop (V4:209-300) over a batch of simulated robots — arrival, minimum obstacle clearance, failed solves.
    python tools/soak_lidar_closed_loop.py [B=1024] [periods=160]
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nmpc_amd
from oracle import lidar_ref as LR
from tests import helpers as Hh
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 160
cfg = LR.LidarConfig(N=25, Nc=12, R=10, T=0.3, aligned_bounds=True)
lbx, ubx, _, _ = LR.bounds(cfg)
pose0, goals, world = Hh.lidar_episode_batch(20210141 + 67, B)
pc = nmpc_amd.LidarProblemConfig(N=cfg.N, Nc=cfg.Nc, R=cfg.R, T=cfg.T, max_iter=2000)
s = nmpc_amd.LidarSolver(pc, lbx=lbx, ubx=ubx, max_batch=B)
t = time.perf_counter()
r = nmpc_amd.simulate_lidar_closed_loop(s, pose0, goals, world, max_steps=steps)
torch.cuda.synchronize(); dt = time.perf_counter() - t
print(json.dumps({"episodes": B, "periods": r.steps, "solves": r.total_solves, "solves_per_s": r.total_solves / dt, "failed_solves": r.failed_solves,
                  "arrived_frac": float(r.arrived.mean()), "goals_reached_hist": np.bincount(r.goals_reached, minlength=3).tolist(),
                  "median_arrival_period": float(np.median(r.arrival_step[r.arrived])) if r.arrived.any() else None,
                  "min_clearance_m": float(r.min_clearance.min()), "robots_with_clearance_below_0.15": int((r.min_clearance < 0.15).sum()),
                  "mean_iters_first_period": float(r.mean_iters_by_step[0]), "mean_iters_later_periods": float(r.mean_iters_by_step[1:].mean()),
                  "config": "V4 NLP, N=25, Nc=12, R=10, T=0.3 s, aligned bounds, ng=2 goals, 3 circular obstacles 0.3-0.55 m off the legs"}))
