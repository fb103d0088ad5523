"""Fixture generator: the one V4 LIDAR instance of a seeded batch of 16,384 (the generator of tools/bench_lidar.py, seed 20210146)
that crawls for 2074 iterations at mu_init = 0.5 (status 1 at max_iter = 2000 without the cold-start retry; 23-42 iterations from any
other initial barrier parameter) — found with the HIP path on the whole batch, index 15025.  Data only: inputs p and w0.

    python tests/golden/gen_lidar_cold_retry_case.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import lidar_ref as LR

INDEX, BATCH = 15025, 16384
lc = LR.lidar_v4()
rng = np.random.Generator(np.random.PCG64(20210146))
p = w0 = None
for b in range(BATCH):
    pose = np.array([rng.uniform(0.0, 0.15), rng.uniform(0.0, 0.15), rng.uniform(0.4, 1.1)])
    world = [(float(rng.uniform(0.8, 2.6)), float(rng.uniform(0.3, 2.4)), float(rng.uniform(0.15, 0.3))) for _ in range(3)]
    scan = LR.scan_of_world(pose, world, lc.R)
    goal = np.array([3.0, 2.5, 0.0]) + rng.uniform(-0.3, 0.3, 3)
    if b == INDEX:
        p = LR.make_p(lc, pose, goal, scan); w0 = LR.cold_start(lc, np.concatenate([pose, scan]))
        break
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lidar_cold_retry_case.npz"), p=p[None], w0=w0[None], index=INDEX)
print("wrote lidar_cold_retry_case.npz", p.shape, w0.shape)
