"""TEST INFRASTRUCTURE — the restated NLP built through the CasADi API and solved by nlpsol('ipopt'), IF `import casadi` succeeds.

SURVEY.md 8(d) / BASELINE.md 3 allow exactly this as the one possible pin against the real CasADi/IPOPT solve: the build's OWN generator
(oracle/nlp_ref.py's NLPConfig: rows, order and bounds as restated there, each citing the script block) constructs the identical NLP with
CasADi symbols and hands it to nlpsol with the scripts' options (C6:345: max_iter 2000, print_level 0, acceptable_tol 1e-8,
acceptable_obj_change_tol 1e-6, print_time 0).  No reference file is read, imported or shipped.  casadi is NOT installed in the build
container nor (as far as any round has seen) on the GPU box: available() says so, tests skip, bench.py reports "casadi": "not importable".

Only tests/ and bench.py's baseline leg import this module.
"""
from __future__ import annotations

import time

import numpy as np

from . import nlp_ref as R

IPOPT_OPTS = {"print_time": 0, "ipopt": {"max_iter": 2000, "print_level": 0, "acceptable_tol": 1e-8, "acceptable_obj_change_tol": 1e-6}}      # C6:345


def available():
    """(True, version) when casadi imports and carries the ipopt plugin, else (False, reason)."""
    try:
        import casadi as ca
    except Exception as e:      # ModuleNotFoundError here; anything else is reported as the reason
        return False, "%s: %s" % (type(e).__name__, e)
    try:
        if not ca.has_nlpsol("ipopt"):
            return False, "casadi %s without the ipopt plugin" % ca.__version__
    except Exception as e:
        return False, "casadi %s: %s" % (getattr(ca, "__version__", "?"), e)
    return True, "casadi %s" % ca.__version__


def build_solver(cfg: R.NLPConfig, opts=None):
    """nlpsol('solver', 'ipopt', {'f','x','g','p'}, opts) of the NLP of oracle/nlp_ref.py (a1-a7, a12): w = [vec X; vec U] (C6:339),
    p = [x0; xs] (C6:241), g in the row order of nlp_ref.constraints()."""
    import casadi as ca
    nx, nu, N, m, T = cfg.nx, cfg.nu, cfg.N, cfg.m, cfg.T
    X = ca.SX.sym("X", nx, N + 1); U = ca.SX.sym("U", nu, N); P = ca.SX.sym("P", 2 * nx)
    qd = np.tile(np.asarray(cfg.q, float), m); rd = np.tile(np.asarray(cfg.r, float), m)
    f = 0
    g = [X[:, 0] - P[:nx]]                                              # a5 (C6:278)
    if cfg.pad_rows:
        g.append(ca.DM(np.full(cfg.rows0 - nx, cfg.pad_value)))
    for k in range(N):
        st, con = X[:, k], U[:, k]
        e = st - P[nx:]
        f = f + ca.dot(e * qd, e) + ca.dot(con * rd, con)                # a3 (C6:314): no 1/2, no terminal term
        rhs = ca.vertcat(*[ca.vertcat(con[2 * i] * ca.cos(st[3 * i + 2]), con[2 * i] * ca.sin(st[3 * i + 2]), con[2 * i + 1]) for i in range(m)])      # a1
        g.append(X[:, k + 1] - (st + T * rhs))                          # a2 (C6:318-323)
        for (i, j) in cfg.pairs():                                      # a4 (C6:288-306), at X_k
            g.append((st[3 * i] - st[3 * j]) ** 2 + (st[3 * i + 1] - st[3 * j + 1]) ** 2)
        for i in range(m):                                              # a12 (O3:145-150)
            for (ox, oy, orad) in cfg.obstacles:
                g.append(ca.sqrt((st[3 * i] - ox) ** 2 + (st[3 * i + 1] - oy) ** 2) - cfg.rob_dim - orad)
    w = ca.vertcat(ca.reshape(X, nx * (N + 1), 1), ca.reshape(U, nu * N, 1))      # column-major reshape = stage-major packing (a6)
    nlp = {"f": f, "x": w, "g": ca.vertcat(*g), "p": P}
    return ca.nlpsol("solver", "ipopt", nlp, dict(IPOPT_OPTS if opts is None else opts))


def solve(cfg: R.NLPConfig, p: np.ndarray, w0: np.ndarray, solver=None):
    """one solve per row of p / w0, single-threaded, exactly the keyword call of C6:432; returns x, f, seconds per solve and the
    solver's return_status strings."""
    solver = solver or build_solver(cfg)
    lbx, ubx, lbg, ubg = R.bounds(cfg)
    p = np.atleast_2d(p); w0 = np.atleast_2d(w0)
    xs, fs, ts, st = [], [], [], []
    for b in range(p.shape[0]):
        t = time.perf_counter()
        sol = solver(x0=w0[b], p=p[b], lbx=lbx, ubx=ubx, lbg=lbg, ubg=ubg)
        ts.append(time.perf_counter() - t)
        xs.append(np.asarray(sol["x"]).reshape(-1)); fs.append(float(sol["f"])); st.append(solver.stats().get("return_status", "?"))
    return {"x": np.array(xs), "f": np.array(fs), "seconds": np.array(ts), "return_status": st}
