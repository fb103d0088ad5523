"""Development aid (GPU box): differential fuzz of the LIDAR solve kernel against the C oracle over random configurations (horizon, control horizon,
ray count, sample time, weights, limits, aligned / script-built bounds) and random instances; one configuration in five runs a batch beyond one
instance per SIMD (the two-waves-per-SIMD build for ten rays).   python tools/fuzz_lidar.py [seed] [configurations]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nmpc_amd
from oracle import lidar_ref as LR, oracle_lib as O
from tests.test_gpu_lidar import _product, _world
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1; ncfg = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.Generator(np.random.PCG64(seed))
bad = 0
for t in range(ncfg):
    N = int(rng.integers(4, 61)); Nc = int(rng.integers(1, N + 1)); R = int(rng.choice([3, 4, 10, 10, 10, 12]))
    cfg = LR.LidarConfig(N=N, Nc=Nc, R=R, T=float(rng.uniform(0.05, 0.3)), q=tuple(rng.uniform(0.1, 5.0, 3)), r=tuple(rng.uniform(0.02, 1.0, 2)),
                         lw=float(rng.choice([0.0, 0.1, 0.3])), v_max=float(rng.uniform(0.1, 0.4)), w_max=float(rng.uniform(1.0, 3.0)),
                         th_max=float(rng.choice([np.inf, 2 * np.pi])), d_min=float(rng.uniform(0.1, 0.2)), d_max=float(rng.choice([10.0, np.inf])),
                         aligned_bounds=bool(rng.integers(0, 2)))
    B = 1100 if t % 5 == 4 else 24
    P, W0 = [], []
    for _ in range(B):
        pose = np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(-0.5, 1.2)]) if cfg.aligned_bounds else np.array([rng.uniform(0.0, 0.15), rng.uniform(0.0, 0.15), rng.uniform(0.4, 1.1)])
        scan = LR.scan_of_world(pose, _world(rng), cfg.R)
        P.append(LR.make_p(cfg, pose, np.array([3.0, 2.5, 0.0]) + rng.uniform(-0.3, 0.3, 3), scan)); W0.append(LR.cold_start(cfg, np.concatenate([pose, scan])))
    P = np.stack(P); W0 = np.stack(W0)
    lbx, ubx, _, _ = LR.bounds(cfg)
    ref = O.lidar_solve_batch(cfg, P, W0, max_iter=800, lbx=lbx, ubx=ubx)
    s = nmpc_amd.LidarSolver(_product(cfg, max_iter=800), lbx=lbx, ubx=ubx, max_batch=B)
    r = {k: v.cpu().numpy() for k, v in s.solve_batch(P, W0).items()}
    dw = np.max(np.abs(r["x"] - ref["x"]), axis=1)
    st_eq = (r["status"] == ref["status"]).mean(); same = (dw <= 1e-6)[(r["status"] == 0) & (ref["status"] == 0)]
    it_eq = (r["iters"] == ref["iters"]).mean()
    ok = st_eq >= 0.95 and (same.size == 0 or same.mean() >= 0.9)
    bad += not ok
    print("cfg %2d: N=%2d Nc=%2d R=%2d aligned=%d B=%4d  status-eq %.3f same-point %.3f iters-eq %.3f  iters hip %.1f ora %.1f  conv hip %.3f ora %.3f%s" %
          (t, N, Nc, R, cfg.aligned_bounds, B, st_eq, same.mean() if same.size else 1.0, it_eq, r["iters"].mean(), ref["iters"].mean(), (r["status"] == 0).mean(), (ref["status"] == 0).mean(),
           "" if ok else "   <-- MISMATCH"), flush=True)
print("mismatching configs:", bad)
