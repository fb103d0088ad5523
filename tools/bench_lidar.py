"""Development aid (GPU box): LIDAR kernel throughput (one wavefront per instance)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nmpc_amd
from oracle import lidar_ref as LR
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lc = LR.lidar_v4()
rng = np.random.Generator(np.random.PCG64(20210146))
Pl, Wl = [], []
for _ in range(B):
    pose = np.array([rng.uniform(0.0, 0.15), rng.uniform(0.0, 0.15), rng.uniform(0.4, 1.1)])
    world = [(float(rng.uniform(0.8, 2.6)), float(rng.uniform(0.3, 2.4)), float(rng.uniform(0.15, 0.3))) for _ in range(3)]
    scan = LR.scan_of_world(pose, world, lc.R)
    Pl.append(LR.make_p(lc, pose, np.array([3.0, 2.5, 0.0]) + rng.uniform(-0.3, 0.3, 3), scan)); Wl.append(LR.cold_start(lc, np.concatenate([pose, scan])))
Pl = torch.as_tensor(np.stack(Pl), device="cuda"); Wl = torch.as_tensor(np.stack(Wl), device="cuda")
lbx, ubx, _, _ = LR.bounds(lc)
s = nmpc_amd.LidarSolver(nmpc_amd.lidar_v4(), lbx=lbx, ubx=ubx, max_batch=B)
for lanes in [64]:
    r = s.solve_batch(Pl, Wl); torch.cuda.synchronize()
    t = time.perf_counter(); r = s.solve_batch(Pl, Wl); torch.cuda.synchronize(); dt = time.perf_counter() - t
    it = r["iters"].cpu().numpy()
    print(f"B={B}: {dt*1e3:8.1f} ms  {B/dt:9.0f} solves/s  mean iters {it.mean():.1f} max {it.max()} converged {(r['status'].cpu().numpy()==0).mean():.3f}", file=sys.stderr, flush=True)
    import json
    # one JSON line in the shape tools/summarize_profiles.py reads (tools/collect_profiles.sh <tag> "tools/bench_lidar.py 4096")
    print(json.dumps({"metric": "LIDAR-state NMPC solves/sec (V4: N=100, Nc=50)", "value": B / dt, "unit": "solves/s", "ms_per_step": dt * 1e3,
                      "config": {"workload": "lidar_v4: 1 robot, 13 states (pose + 10 ray distances), N=100, Nc=50, cold start", "batch_per_gpu": B},
                      "library": nmpc_amd._lib.load().nmpc_version().decode(),
                      "solve_stats": {"mean_iters": float(it.mean()), "max_iters": float(it.max()), "converged_frac": float((r["status"].cpu().numpy() == 0).mean())}}), flush=True)
