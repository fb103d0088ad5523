"""Generator of tests/golden/cold_retry_cases.npz, cold_retry_cases2.npz and stall_case.npz (regenerated in round 4 for the partial
re-factorisation of the backward sweep: the captured failures are failures of the algorithm as shipped).

Warm-started solves of the six-robot + eight-obstacle composite (BASELINE config 5, N = 25) captured from CPU closed-loop soaks:
  cold_retry_cases.npz   256 swarms x 40 periods = 10,240 solves WITHOUT the cold-start retry (NMPC_ORACLE_NO_COLD_RETRY=1): the solves that
                         stall at an infeasible stationary point after three barrier restarts (status 4), cycle until max_iter (status 1) or
                         fail numerically (status 2);
  stall_case.npz         the first status-4 solve of that soak (the single-instance fixture of the stall tests);
  cold_retry_cases2.npz  512 swarms x 60 periods = 30,720 solves with ONE retry only (NMPC_ORACLE_MAX_COLD=1): the root failures the first
                         retry (cold start, mu_init) does not rescue; the second one (cold start, 10 mu_init) must.
Inputs only; the tests assert that the shipping oracle and the HIP path converge on all of them."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def soak(B, T):
    from oracle import nlp_ref as R, oracle_lib as O
    from tests import helpers as Hh
    rng = np.random.default_rng(7)
    c = R.cfg_six(25); c.rob_dim = 0.2; c.margin = 0.1
    c.obstacles = [(float(x), float(y), float(r)) for x, y, r in zip(rng.uniform(-1.5, 1.5, 8), rng.uniform(-1.5, 1.5, 8), rng.uniform(0.125, 0.2, 8))]
    oc = O.make_config(c, max_iter=2000)
    P, W = Hh.batch(c, B, 4)
    bp, bw, bs = [], [], []
    for step in range(T):
        r = O.solve_batch(oc, P, W)
        for b in np.where((r["status"] != 0) & (r["status"] != 3))[0]:      # status 3 = infeasible x0, the consequence of an earlier failure
            bp.append(P[b].copy()); bw.append(W[b].copy()); bs.append(int(r["status"][b]))
        W, x0n = O.shift_batch(oc, P, r["x"])
        P = P.copy(); P[:, : c.nx] = x0n
    return np.array(bp), np.array(bw), np.array(bs)


if len(sys.argv) > 1 and sys.argv[1] == "child1":
    p, w, s = soak(256, 40)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cold_retry_cases.npz"), p=p, w=w, status_without_retry=s)
    i = int(np.where(s == 4)[0][0])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "stall_case.npz"), p=p[i], w=w[i])
    print("failing solves captured without the retry:", len(p), "statuses", s.tolist(), "stall case = #%d" % i)
elif len(sys.argv) > 1 and sys.argv[1] == "child2":
    p, w, s = soak(512, 60)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cold_retry_cases2.npz"), p=p, w=w, status_with_one_retry=s)
    print("root failures left by the first retry:", len(p), "statuses", s.tolist())
else:
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "child1"], env=dict(os.environ, NMPC_ORACLE_NO_COLD_RETRY="1"))
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "child2"], env=dict(os.environ, NMPC_ORACLE_MAX_COLD="1"))
