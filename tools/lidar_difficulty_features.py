"""Development aid (CPU only): is the iteration count of a cold LIDAR solve predictable from its inputs?  (A launch of 4096 instances lasts as
long as its longest solve plus the time at which that solve started: a dispatch order "long solves first" would be worth up to 1.5x.)
Iteration counts of the bench batch from the C oracle against cheap features of p = [pose; goal; scan; ray angles].

    python tools/lidar_difficulty_features.py [B]

Result (round 3, B=4096: mean 28.6, max 175): Spearman rank correlation with the iteration count — nearest scan return -0.08, sum 1/d^2 0.08,
goal distance 0.00, heading error -0.15, nearest lidar point to the start-goal segment -0.11, points within 0.4 m of it 0.10.  The 175-iteration
instance sees no obstacle at all.  No usable predictor: the tail is a property of the interior-point path, not of the geometry.
"""
import os
import sys

import numpy as np
from scipy.stats import spearmanr

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import lidar_ref as LR
from oracle import oracle_lib as OL
from tests import helpers as Hh

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lc = LR.lidar_v4()
rng = np.random.Generator(np.random.PCG64(Hh.SEED0 + 5))      # the LIDAR batch of bench.py
P, W = [], []
for _ in range(B):
    pose = np.array([rng.uniform(0.0, 0.15), rng.uniform(0.0, 0.15), rng.uniform(0.4, 1.1)])
    world = [(float(rng.uniform(0.8, 2.6)), float(rng.uniform(0.3, 2.4)), float(rng.uniform(0.15, 0.3))) for _ in range(3)]
    scan = LR.scan_of_world(pose, world, lc.R)
    P.append(LR.make_p(lc, pose, np.array([3.0, 2.5, 0.0]) + rng.uniform(-0.3, 0.3, 3), scan))
    W.append(LR.cold_start(lc, np.concatenate([pose, scan])))
P = np.stack(P); W = np.stack(W)
lbx, ubx, _, _ = LR.bounds(lc)
it = OL.lidar_solve_batch(lc, P, W, lbx=lbx, ubx=ubx)["iters"]
print("B=%d: mean %.1f, max %d iterations; longest at %s" % (B, it.mean(), it.max(), np.argsort(-it)[:6]))
pose, xs, scan, ang = P[:, 0:3], P[:, 3:6], P[:, 6:6 + lc.R], P[:, 6 + lc.R:6 + 2 * lc.R]
px = pose[:, None, 0] + scan * np.cos(ang + pose[:, None, 2]); py = pose[:, None, 1] + scan * np.sin(ang + pose[:, None, 2])
a, b, q = pose[:, None, :2], xs[:, None, :2], np.stack([px, py], -1)
ab = b - a
tt = np.clip(((q - a) * ab).sum(-1) / (ab * ab).sum(-1), 0, 1)
dseg = np.where(scan < 3.4, np.linalg.norm(q - (a + tt[..., None] * ab), axis=-1), 9.0)
feats = {"nearest scan return": scan.min(1), "sum 1/d^2": (1 / scan ** 2).sum(1), "goal distance": np.linalg.norm(xs[:, :2] - pose[:, :2], axis=1),
         "heading error": np.abs(np.arctan2(xs[:, 1] - pose[:, 1], xs[:, 0] - pose[:, 0]) - pose[:, 2]),
         "nearest lidar point to the start-goal segment": dseg.min(1), "lidar points within 0.4 m of it": (dseg < 0.4).sum(1)}
for k, v in feats.items():
    print("  %-48s Spearman %+.3f" % (k, spearmanr(v, it)[0]))
