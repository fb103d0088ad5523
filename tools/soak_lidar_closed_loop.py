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
mode = sys.argv[3] if len(sys.argv) > 3 else "short"
if mode == "short":
    cfg = LR.LidarConfig(N=25, Nc=12, R=10, T=0.3, aligned_bounds=True)
    pose0, goals, world = Hh.lidar_episode_batch(20210141 + 67, B)
    label = "V4 NLP, N=25, Nc=12, R=10, T=0.3 s, aligned bounds, ng=2 goals, 3 circular obstacles 0.3-0.55 m off the legs"
else:
    # the script itself: V4's literals (N=100, Nc=50, T=0.075, R=10), its first two goals (V4:212-215, ng = 2), start at the origin;
    # mode "v4": bounds exactly as the script builds them (V4:161-176, misaligned); "v4_aligned": bounds aligned with the packing
    cfg = LR.LidarConfig(aligned_bounds=(mode == "v4_aligned"))
    rng = np.random.Generator(np.random.PCG64(20210141 + 68))
    g = np.array([[3.0, 2.5, 0.0], [0.0, 2.5, -0.785]])
    pose0 = np.stack([np.array([rng.uniform(0.16, 0.3), rng.uniform(0.16, 0.3), rng.uniform(0.2, 0.9)]) for _ in range(B)])
    goals = np.tile(g[None], (B, 1, 1))

    def seg_dist(c, a, b):
        d = b - a; t = np.clip(np.dot(c - a, d) / np.dot(d, d), 0.0, 1.0)
        return np.linalg.norm(c - (a + t * d))
    world = []
    for b in range(B):
        obs = []
        while len(obs) < 3:
            c = np.array([rng.uniform(-0.5, 3.5), rng.uniform(-0.5, 3.2)])
            d = min(seg_dist(c, pose0[b, :2], g[0, :2]), seg_dist(c, g[0, :2], g[1, :2]))
            if 0.35 <= d <= 0.7 and all(np.linalg.norm(c - np.array(o[:2])) >= 0.4 for o in obs):
                obs.append((c[0], c[1], rng.uniform(0.08, 0.15)))
        world.append(np.array(obs))
    world = np.stack(world)
    label = "V4 literal: N=100, Nc=50, R=10, T=0.075 s, goals (3, 2.5, 0) then (0, 2.5, -0.785), %s bounds, 3 circular obstacles 0.35-0.7 m off the legs" % ("aligned" if mode == "v4_aligned" else "the script's misaligned")
lbx, ubx, _, _ = LR.bounds(cfg)
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
                  "final_error_median": float(np.median(r.final_error)), "config": label}))
