"""dev aid: differential fuzz of the HIP path against the C oracle over random problem configurations
(robots, horizon, sample time, bounds, weights, obstacles, heading bound) and random instances."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nmpc_amd
from oracle import nlp_ref as R, oracle_lib as O
from tests import helpers as Hh
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1; ncfg = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.Generator(np.random.PCG64(seed))
bad = 0
for t in range(ncfg):
    m = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10]))
    N = int(rng.integers(2, 41)) if m <= 6 else int(rng.integers(2, 25))
    K = int(rng.integers(0, 9)) if m <= 3 else int(rng.integers(0, 3))
    cfg = R.NLPConfig(m=m, N=N, T=float(rng.uniform(0.05, 0.3)), dmin=float(rng.uniform(0.15, 0.4)),
                      q=tuple(rng.uniform(0.1, 5.0, 3)), r=tuple(rng.uniform(0.02, 1.0, 2)),
                      v_max=float(rng.uniform(0.1, 0.5)), w_max=float(rng.uniform(1.0, 3.0)), xy_max=float(rng.uniform(4.0, 10.0)),
                      th_max=float(rng.choice([np.inf, 2 * np.pi, 4.0])), rob_dim=0.2, margin=float(rng.uniform(0.05, 0.1)),
                      obstacles=[(float(x), float(y), float(r_)) for x, y, r_ in zip(rng.uniform(-1.5, 1.5, K), rng.uniform(-1.5, 1.5, K), rng.uniform(0.1, 0.2, K))],
                      pad_rows=bool(m > 1))
    B = 24 if m <= 6 else 8
    P = np.stack([Hh.instance(rng, cfg) for _ in range(B)])
    W0 = np.stack([R.cold_start(cfg, p[: cfg.nx]) for p in P])
    ref = O.solve_batch(O.make_config(cfg, max_iter=800), P, W0)
    s = nmpc_amd.NmpcSolver(Hh.to_product_cfg(cfg, max_iter=800), max_batch=B)
    r = {k: v.cpu().numpy() for k, v in s.solve_batch(P, W0).items()}
    dw = np.max(np.abs(r["x"] - ref["x"]), axis=1)
    same = dw <= 1e-6
    st_eq = (r["status"] == ref["status"])
    df = np.abs(r["f"] - ref["f"]) / np.maximum(1, np.abs(ref["f"]))
    ok = st_eq.mean() >= 0.9 and same.mean() >= 0.8 and (df[same] <= 1e-6).all() and (r["kkt"][r["status"] == 0] <= 1e-8).all()
    flag = "" if ok else "   <<<<<< MISMATCH"
    bad += not ok
    print(f"cfg {t:2d}: m={m} N={N} K={K} thb={np.isfinite(cfg.th_max)} status-eq {st_eq.mean():.2f} same-basin {same.mean():.2f} "
          f"iters hip {r['iters'].mean():.1f} ora {ref['iters'].mean():.1f} conv hip {(r['status'] == 0).mean():.2f} ora {(ref['status'] == 0).mean():.2f}{flag}", flush=True)
print("mismatching configs:", bad)
