"""Generator of tests/golden/restart_cases.npz: composite (six robots + eight obstacles, N = 25) closed-loop solves that stall
at an infeasible stationary point WITHOUT the barrier restart (NMPC_ORACLE_MAX_RESTARTS=0) — found by running the closed
loop with the restart-free oracle.  Inputs only; the expected behaviour is asserted by the tests with the shipping oracle."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from oracle import nlp_ref as R, oracle_lib as O
    from tests import helpers as Hh
    rng = np.random.default_rng(7)
    c = R.cfg_six(25); c.rob_dim = 0.2; c.margin = 0.1
    c.obstacles = [(float(x), float(y), float(r)) for x, y, r in zip(rng.uniform(-1.5, 1.5, 8), rng.uniform(-1.5, 1.5, 8), rng.uniform(0.125, 0.2, 8))]
    oc = O.make_config(c, max_iter=2000)
    P, W = Hh.batch(c, 512, 2)
    bp, bw = [], []
    for step in range(12):
        r = O.solve_batch(oc, P, W)
        for b in np.where(r["status"] == 4)[0]:
            bp.append(P[b].copy()); bw.append(W[b].copy())
        W, x0n = O.shift_batch(oc, P, r["x"])
        P = P.copy(); P[:, : c.nx] = x0n
    np.savez(os.path.join(ROOT, "tests", "golden", "restart_cases.npz"), p=np.array(bp[:12]), w=np.array(bw[:12]))
    print("stalled cases found:", len(bp))
else:
    env = dict(os.environ, NMPC_ORACLE_MAX_RESTARTS="0")
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "child"], env=env)
