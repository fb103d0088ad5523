"""Generator of tests/golden/cold_retry_cases.npz: warm-started solves of the six-robot + eight-obstacle composite (BASELINE config 5,
N = 25) captured from a CPU closed-loop soak (256 swarms x 40 periods = 10,240 solves) that FAIL without the cold-start retry
(NMPC_ORACLE_NO_COLD_RETRY=1): stall at an infeasible stationary point after three barrier restarts (status 4) or cycling until
max_iter (status 1).  Inputs only; the tests assert that the shipping oracle and the HIP path converge on all of them."""
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
    P, W = Hh.batch(c, 256, 4)
    bp, bw = [], []
    for step in range(40):
        r = O.solve_batch(oc, P, W)
        for b in np.where((r["status"] != 0) & (r["status"] != 3))[0]:      # status 3 = infeasible x0, the consequence of an earlier failure
            bp.append(P[b].copy()); bw.append(W[b].copy())
        W, x0n = O.shift_batch(oc, P, r["x"])
        P = P.copy(); P[:, : c.nx] = x0n
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cold_retry_cases.npz"), p=np.array(bp), w=np.array(bw))
    print("failing solves captured:", len(bp))
else:
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, NMPC_ORACLE_NO_COLD_RETRY="1"))
