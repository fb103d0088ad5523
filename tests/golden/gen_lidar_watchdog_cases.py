"""Fixture generator: the V4 LIDAR instances of bench.py's batch (4096 instances, seed 20210141 + 5) that need more than 100 iterations
without the line-search watchdog (NMPC_LIDAR_ORACLE_WD=0: 175, 154 and 145 iterations; the l1 merit function rejects full steps across the kinks
of the 1-norm distance rows) plus the five longest solves that remain with it.  Data only: inputs p and w0, and the iteration counts of the
oracle without / with the watchdog at the time of writing.

    python tests/golden/gen_lidar_watchdog_cases.py
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lidar_ref as LR
from oracle import oracle_lib as OL
from tests import helpers as Hh

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lidar_watchdog_cases.npz")
B = 4096


def batch():
    lc = LR.lidar_v4()
    rng = np.random.Generator(np.random.PCG64(Hh.SEED0 + 5))      # the LIDAR batch of bench.py
    P, W = [], []
    for _ in range(B):
        pose = np.array([rng.uniform(0.0, 0.15), rng.uniform(0.0, 0.15), rng.uniform(0.4, 1.1)])
        world = [(float(rng.uniform(0.8, 2.6)), float(rng.uniform(0.3, 2.4)), float(rng.uniform(0.15, 0.3))) for _ in range(3)]
        scan = LR.scan_of_world(pose, world, lc.R)
        P.append(LR.make_p(lc, pose, np.array([3.0, 2.5, 0.0]) + rng.uniform(-0.3, 0.3, 3), scan))
        W.append(LR.cold_start(lc, np.concatenate([pose, scan])))
    return lc, np.stack(P), np.stack(W)


if __name__ == "__main__":
    lc, P, W = batch()
    lbx, ubx, _, _ = LR.bounds(lc)
    if len(sys.argv) > 1:       # child: iteration counts under the environment of the parent's choice
        np.save(sys.argv[1], OL.lidar_solve_batch(lc, P, W, lbx=lbx, ubx=ubx)["iters"])
        sys.exit(0)
    its = {}
    for tag, env in (("off", {"NMPC_LIDAR_ORACLE_WD": "0"}), ("on", {})):
        f = "/tmp/_lidar_wd_%s.npy" % tag
        e = {k: v for k, v in os.environ.items() if k != "NMPC_LIDAR_ORACLE_WD"}
        subprocess.check_call([sys.executable, os.path.abspath(__file__), f], env=dict(e, **env))
        its[tag] = np.load(f)
    idx = sorted(set(np.nonzero(its["off"] > 100)[0].tolist()) | set(np.argsort(-its["on"])[:5].tolist()))
    np.savez(OUT, p=P[idx], w0=W[idx], index=np.array(idx), iters_without=its["off"][idx], iters_with=its["on"][idx],
             batch_mean_max_without=np.array([its["off"].mean(), its["off"].max()]), batch_mean_max_with=np.array([its["on"].mean(), its["on"].max()]))
    print("wrote", OUT, idx, its["off"][idx], its["on"][idx], its["off"].mean(), its["on"].mean())
