"""Synthetic instance generator of SURVEY.md §8(d): PCG64(seed = 20210141 + config_index)."""
import numpy as np

from oracle import nlp_ref as R

SEED0 = 20210141


def cfg_mix3(N=10):
    """the two row types of the synthetic composite (BASELINE config 5) at a size SLSQP solves: three robots' pair rows (C6:288-306) plus two
    circular obstacles (third_scenario_mpc_obstacle_avoidance.py:145-150), literals of C6 / the composite"""
    return R.NLPConfig(m=3, N=N, T=0.3, dmin=0.4, v_max=0.15, w_max=1.5, rob_dim=0.2, margin=0.1, obstacles=[(0.45, 0.1, 0.15), (-0.3, 0.5, 0.125)])


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


def to_oracle_cfg(pcfg):
    """product ProblemConfig -> oracle NLPConfig (the checker's own type; bench.make_batch returns the product's)."""
    return R.NLPConfig(m=pcfg.m, N=pcfg.N, T=pcfg.T, dmin=pcfg.dmin, q=tuple(pcfg.q), r=tuple(pcfg.r), v_max=pcfg.v_max, w_max=pcfg.w_max,
                       xy_max=pcfg.xy_max, th_max=pcfg.th_max, obstacles=list(pcfg.obstacles), rob_dim=pcfg.rob_dim, margin=pcfg.margin,
                       pad_value=pcfg.pad_value, pad_rows=pcfg.pad_rows, pair_rows=pcfg.pair_rows)


def bench_batch(name, B=0, rank=0, shard=None):
    """bench.make_batch with the oracle's config type in front: (oracle NLPConfig, B, P, W0)."""
    import bench
    pcfg, B, P, W0 = bench.make_batch(name, rank, B, shard=shard)
    return to_oracle_cfg(pcfg), B, P, W0


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


def lidar_closed_loop_oracle(cfg, pose0, goals, world, max_steps, arrive_tol=0.2, scan_max=3.5, lbx=None, ubx=None, max_iter=2000):
    """The main loop of obs_avoid_static_first_scenario_v4.py (V4:209-300) driven by the CPU oracle on a synthetic world of circular
    obstacles — checker for nmpc_amd.simulate_lidar_closed_loop: scan (callback_lidar V4:29-36 -> lidar_ref.scan_of_world), p (V4:230-236),
    solve (V4:245), shift (V4:258-270), robot = Euler model of the NLP, goal sequencing ne < 0.2 (V4:281-291), x0 <- [Xr; Scan] (V4:296-297)."""
    from oracle import lidar_ref as LR, oracle_lib as O
    pose = np.array(pose0, dtype=np.float64); B = pose.shape[0]
    g = np.array(goals, dtype=np.float64)
    if g.ndim == 2:
        g = g[:, None, :]
    G = g.shape[1]
    world = np.array(world, dtype=np.float64).reshape(B, -1, 3)
    if lbx is None:
        lbx, ubx, _, _ = LR.bounds(cfg)
    gi = np.zeros(B, dtype=int); ar = np.arange(B)
    arrived = np.zeros(B, dtype=bool); arrival = np.full(B, -1)
    scan = np.stack([LR.scan_of_world(pose[b], [tuple(o) for o in world[b]], cfg.R, scan_max) for b in range(B)])
    w = np.stack([LR.cold_start(cfg, np.concatenate([pose[b], scan[b]])) for b in range(B)])
    clear = np.full(B, np.inf)

    def track():
        if world.shape[1]:
            np.minimum(clear, (np.linalg.norm(pose[:, None, :2] - world[:, :, :2], axis=2) - world[:, :, 2]).min(axis=1), out=clear)
    track()
    poses = [pose.copy()]
    failed = steps = 0
    for step in range(max_steps):
        if arrived.all():
            break
        xs = g[ar, gi]
        p = np.stack([LR.make_p(cfg, pose[b], xs[b], scan[b]) for b in range(B)])
        r = O.lidar_solve_batch(cfg, p, w, max_iter=max_iter, lbx=lbx, ubx=ubx)
        failed += int((r["status"] != 0).sum())
        w = np.stack([LR.shift_guess(cfg, x) for x in r["x"]])
        u0 = r["x"][:, cfg.ns * (cfg.N + 1): cfg.ns * (cfg.N + 1) + 2]
        pn = pose + cfg.T * np.stack([u0[:, 0] * np.cos(pose[:, 2]), u0[:, 0] * np.sin(pose[:, 2]), u0[:, 1]], axis=1)
        pose = np.where(arrived[:, None], pose, pn)
        track()
        hit = (np.linalg.norm(pose - xs, axis=1) < arrive_tol) & ~arrived
        last = gi == G - 1
        arrival[hit & last] = step + 1; arrived |= hit & last
        gi = np.where(hit & ~last, gi + 1, gi)
        scan = np.stack([LR.scan_of_world(pose[b], [tuple(o) for o in world[b]], cfg.R, scan_max) for b in range(B)])
        poses.append(pose.copy()); steps += 1
    return dict(steps=steps, arrived=arrived, arrival_step=arrival, goals_reached=gi + arrived, min_clearance=clear, failed_solves=failed,
                final_error=np.linalg.norm(pose - g[ar, gi], axis=1), poses=np.stack(poses))


def lidar_episode_batch(seed, B):
    """synthetic LIDAR closed-loop episodes: start near the origin, ng = 2 goals (as V4:66-67) about a metre apart, three small circular
    obstacles between 0.3 and 0.55 m from the two straight legs start -> goal 1 -> goal 2 (rejection-sampled: close enough that the rays see
    them and the 1/d^2 cost bends the path, never on a leg).  Returns pose0 [B,3], goals [B,2,3], world [B,3,3]."""
    rng = np.random.Generator(np.random.PCG64(seed))

    def seg_dist(c, a, b):
        d = b - a
        t = np.clip(np.dot(c - a, d) / np.dot(d, d), 0.0, 1.0)
        return np.linalg.norm(c - (a + t * d))
    pose0, goals, world = [], [], []
    for _ in range(B):
        p0 = np.array([rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), rng.uniform(0.2, 1.0)])
        g1 = np.array([rng.uniform(0.8, 1.1), rng.uniform(0.5, 0.8), 0.0]); g2 = np.array([rng.uniform(0.0, 0.3), rng.uniform(1.1, 1.4), 0.785 * rng.uniform(1.5, 2.5)])
        obs = []
        while len(obs) < 3:
            c = np.array([rng.uniform(-0.5, 1.6), rng.uniform(-0.5, 1.9)])
            d = min(seg_dist(c, p0[:2], g1[:2]), seg_dist(c, g1[:2], g2[:2]))
            if 0.3 <= d <= 0.55 and all(np.linalg.norm(c - np.array(o[:2])) >= 0.3 for o in obs):
                obs.append((c[0], c[1], rng.uniform(0.06, 0.1)))
        world.append(np.array(obs)); pose0.append(p0); goals.append(np.stack([g1, g2]))
    return np.stack(pose0), np.stack(goals), np.stack(world)
