"""Synthetic instance generator of SURVEY.md §8(d): PCG64(seed = 20210141 + config_index)."""
import numpy as np

from oracle import nlp_ref as R

SEED0 = 20210141


def sample_points(rng, m, dsep, lim=2.0, obstacles=(), clear=0.0):
    pts = []
    while len(pts) < m:
        c = rng.uniform(-lim, lim, 2)
        if all(np.hypot(*(c - q)) >= dsep for q in pts) and all(np.hypot(c[0] - ox, c[1] - oy) >= clear + orad for (ox, oy, orad) in obstacles):
            pts.append(c)
    return np.array(pts)


def instance(rng, cfg):
    """p = [x0; xs]: starts/goals uniform in [-2,2]^2 with pairwise distance >= dmin + 0.1, theta uniform."""
    m = cfg.m
    dsep = cfg.dmin + 0.1
    clear = cfg.rob_dim + cfg.margin + 0.05
    s = sample_points(rng, m, dsep, obstacles=cfg.obstacles, clear=clear)
    g = sample_points(rng, m, dsep, obstacles=cfg.obstacles, clear=clear)
    x0 = np.concatenate([s, rng.uniform(-np.pi, np.pi, (m, 1))], axis=1).reshape(-1)
    xs = np.concatenate([g, rng.uniform(-np.pi, np.pi, (m, 1))], axis=1).reshape(-1)
    return np.concatenate([x0, xs])


def batch(cfg, B, config_index):
    rng = np.random.Generator(np.random.PCG64(SEED0 + config_index))
    P = np.stack([instance(rng, cfg) for _ in range(B)])
    W0 = np.stack([R.cold_start(cfg, p[: cfg.nx]) for p in P])
    return P, W0


def to_product_cfg(ocfg, **kw):
    """oracle NLPConfig -> product ProblemConfig (two independent definitions of the same literals)."""
    import nmpc_amd
    return nmpc_amd.ProblemConfig(m=ocfg.m, N=ocfg.N, T=ocfg.T, dmin=ocfg.dmin, q=tuple(ocfg.q), r=tuple(ocfg.r), v_max=ocfg.v_max,
                                  w_max=ocfg.w_max, xy_max=ocfg.xy_max, th_max=ocfg.th_max, obstacles=list(ocfg.obstacles),
                                  rob_dim=ocfg.rob_dim, margin=ocfg.margin, pad_value=ocfg.pad_value, pad_rows=ocfg.pad_rows, **kw)


def closed_loop_oracle(ocfg, x0, goals, max_steps, stop_tol=5e-2, max_iter=2000):
    """The reference's main loop (casadi_test.py:143-183; goal sequencing of centralized_one_robots_implementation.py:176-239)
    driven by the CPU oracle — checker for nmpc_amd.simulate_closed_loop."""
    from oracle import oracle_lib as O
    oc = O.make_config(ocfg, max_iter=max_iter)
    x = np.array(x0, dtype=np.float64); B, nx = x.shape
    g = np.array(goals, dtype=np.float64)
    if g.ndim == 2:
        g = g[:, None, :]
    G = g.shape[1]
    gi = np.zeros(B, dtype=int); ar = np.arange(B)
    w = np.stack([R.cold_start(ocfg, xi) for xi in x])
    arrived = np.zeros(B, dtype=bool); arrival = np.full(B, -1)
    states = [x.copy()]
    steps = 0
    for step in range(max_steps):
        err = np.linalg.norm(x - g[ar, gi], axis=1)
        hit = (err <= stop_tol) & ~arrived
        last = gi == G - 1
        arrival[hit & last] = step; arrived |= hit & last
        gi = np.where(hit & ~last, gi + 1, gi)
        if arrived.all():
            break
        p = np.concatenate([x, g[ar, gi]], axis=1)
        r = O.solve_batch(oc, p, w)
        w, xn = O.shift_batch(oc, p, r["x"])
        x = np.where(arrived[:, None], x, xn)
        states.append(x.copy()); steps += 1
    err = np.linalg.norm(x - g[ar, gi], axis=1)
    late = (err <= stop_tol) & (gi == G - 1) & ~arrived
    arrival[late] = steps; arrived |= late
    return dict(steps=steps, arrived=arrived, arrival_step=arrival, final_error=err, states=np.stack(states))


def to_oracle_cfg(pcfg):
    """product ProblemConfig -> oracle NLPConfig (for the script presets, whose literals tests/test_abi_host.py checks
    against its own table)."""
    return R.NLPConfig(m=pcfg.m, N=pcfg.N, T=pcfg.T, dmin=pcfg.dmin, q=tuple(pcfg.q), r=tuple(pcfg.r), v_max=pcfg.v_max, w_max=pcfg.w_max,
                       xy_max=pcfg.xy_max, th_max=pcfg.th_max, obstacles=list(pcfg.obstacles), rob_dim=pcfg.rob_dim, margin=pcfg.margin,
                       pad_value=pcfg.pad_value, pad_rows=pcfg.pad_rows, pair_rows=pcfg.pair_rows)
