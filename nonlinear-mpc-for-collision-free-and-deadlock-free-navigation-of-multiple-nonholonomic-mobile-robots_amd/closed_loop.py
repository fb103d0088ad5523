"""Batched receding-horizon episodes: the reference's main loop run for B swarms at once.

Mirrors, per swarm and per control period (file:line of the reference):
  p = [x0; xs]                                   AllScripts/casadi_test.py:145            (C6:419)
  sol = solver(x0=guess, p=p, ...)               casadi_test.py:153                       (C6:432)
  x0 <- x0 + T f(x0, u[0]) ; guess <- shift      casadi_test.py:17-26,170,180             (C6:450,465)
  stop when ||x0 - xs|| <= tol                   casadi_test.py:143 (5e-2)
  goal sequencing: next goal when ||x - xs|| < tol   AllScripts/centralized_one_robots_implementation.py:176-188,236-239 (0.075)

Every arithmetic step runs in libnmpc_hip.so (nmpc_solve_batch, nmpc_shift_batch); the bookkeeping around it (norms,
goal indices, distance statistics) is a handful of torch reductions on device memory.  As in the reference, the control of a
non-converged solve is applied as returned; such solves are counted in `failed_solves`.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from .solver import NmpcSolver


@dataclass
class EpisodeResult:
    steps: int                    # control periods executed
    arrived: "np.ndarray"         # [B] bool: last goal reached within tol
    arrival_step: "np.ndarray"    # [B] int: period at which it happened (-1: never)
    collision_free: "np.ndarray"  # [B] bool: all pair distances >= dmin - coll_tol (and obstacle clearances) at every period
    min_pair_distance: "np.ndarray"   # [B] smallest centre distance seen (inf for one robot)
    deadlocked: "np.ndarray"      # [B] bool: not arrived and the swarm moved less than move_tol over the last `window` periods
    final_error: "np.ndarray"     # [B] ||x - xs|| at the end
    failed_solves: int            # solves that did not return status 0
    total_solves: int
    mean_iters_by_step: "np.ndarray"   # [steps]
    states: Optional["np.ndarray"] = None   # [steps+1, B, n_x] when keep_states


def simulate_closed_loop(solver: NmpcSolver, x0, goals, max_steps: int, stop_tol: float = 5e-2, coll_tol: float = 1e-6,
                         window: int = 10, move_tol: float = 1e-3, keep_states: bool = False,
                         on_failure: str = "apply", order_hint: bool = True) -> EpisodeResult:
    """Run B closed-loop episodes.  x0: [B, n_x]; goals: [B, n_x] or [B, G, n_x] (visited in order).

    on_failure: what a swarm does in a period whose solve did not converge (status != 0; ~1 in 1e5 warm solves stall at an
    infeasible stationary point): "apply" = the reference's behaviour, the returned iterate's first control is applied;
    "previous_plan" = the next control of the previous period's plan (the shifted guess) is applied instead."""
    if on_failure not in ("apply", "previous_plan"):
        raise ValueError("on_failure must be 'apply' or 'previous_plan'")
    torch = solver.torch
    cfg = solver.cfg
    nx, m = cfg.nx, cfg.m
    dev = solver.device
    x = solver._dev(x0, (-1, nx)).clone()
    B = x.shape[0]
    g = torch.as_tensor(np.asarray(goals, dtype=np.float64) if not torch.is_tensor(goals) else goals, dtype=torch.float64, device=dev)
    if g.dim() == 2:
        g = g[:, None, :]
    if g.shape[0] != B or g.shape[2] != nx:
        raise ValueError(f"goals must be [B, n_x] or [B, G, n_x] with B={B}, n_x={nx}; got {tuple(g.shape)}")
    G = g.shape[1]
    gi = torch.zeros(B, dtype=torch.long, device=dev)
    ar = torch.arange(B, device=dev)
    # cold start of C6:398-400: states replicated, controls zero
    w = torch.cat([x.repeat(1, cfg.N + 1), torch.zeros((B, cfg.N * cfg.nu), dtype=torch.float64, device=dev)], dim=1)
    arrived = torch.zeros(B, dtype=torch.bool, device=dev)
    arrival = torch.full((B,), -1, dtype=torch.long, device=dev)
    mind = torch.full((B,), float("inf"), dtype=torch.float64, device=dev)
    clear = torch.full((B,), float("inf"), dtype=torch.float64, device=dev)
    iu = torch.triu_indices(m, m, 1, device=dev)
    obs = torch.tensor(cfg.obstacles, dtype=torch.float64, device=dev).reshape(-1, 3) if len(cfg.obstacles) else None
    hist, its, states = [], [], []
    prev_iters = None
    failed = total = 0
    # on_failure == "apply" (the reference's behaviour) runs one nmpc_step_batch per period: solve, guess shift, plant step and the next
    # period's dispatch order on the device, p and w updated in place; "previous_plan" needs the status between solve and shift and
    # keeps the separate calls
    fused = on_failure == "apply"
    w = w.contiguous()
    pbuf = torch.empty((B, 2 * nx), dtype=torch.float64, device=dev)
    order = torch.arange(B, dtype=torch.int32, device=dev) if order_hint else None

    def track(xc):
        nonlocal mind, clear
        xy = xc.reshape(B, m, 3)[:, :, :2]
        if m > 1:
            d = (xy[:, iu[0]] - xy[:, iu[1]]).norm(dim=2).min(dim=1).values
            mind = torch.minimum(mind, d)
        if obs is not None:
            c = ((xy[:, :, None, :] - obs[None, None, :, :2]).norm(dim=3) - cfg.rob_dim - obs[None, None, :, 2]).amin(dim=(1, 2))
            clear = torch.minimum(clear, c)

    track(x)
    if keep_states:
        states.append(x.clone())
    steps = 0
    for step in range(max_steps):
        xs = g[ar, gi]
        err = (x - xs).norm(dim=1)
        hit = (err <= stop_tol) & ~arrived
        last = gi == G - 1
        newly = hit & last
        arrival = torch.where(newly, torch.full_like(arrival, step), arrival)
        arrived |= newly
        gi = torch.where(hit & ~last, gi + 1, gi)
        if bool(arrived.all()):
            break
        xs = g[ar, gi]
        # dispatch-order hint: the swarms that needed the most iterations in the previous period go first (rank correlation of
        # consecutive periods' iteration counts 0.6-0.7; -12 % launch time on the six-robot batch)
        if fused:
            pbuf[:, :nx] = x; pbuf[:, nx:] = xs
            r = solver.step_batch(pbuf, w, order)
            xn = pbuf[:, :nx].clone()
        else:
            p = torch.cat([x, xs], dim=1)
            r = solver.solve_batch(p, w, order=None if (prev_iters is None or not order_hint) else torch.argsort(prev_iters, descending=True))
            prev_iters = r["iters"]
            plan = torch.where((r["status"] == 0)[:, None], r["x"], w)
            w, xn = solver.shift_batch(p, plan, plant=True)
        total += B
        failed += int((r["status"] != 0).sum())
        its.append(r["iters"].double().mean())
        # a swarm that has arrived keeps solving (its problem is the fixed point) but stays where it is
        x = torch.where(arrived[:, None], x, xn)
        track(x)
        hist.append(x.clone())
        if len(hist) > window:
            hist.pop(0)
        if keep_states:
            states.append(x.clone())
        steps += 1
    xs = g[ar, gi]
    err = (x - xs).norm(dim=1)
    late = (err <= stop_tol) & (gi == G - 1) & ~arrived
    arrival = torch.where(late, torch.full_like(arrival, steps), arrival)
    arrived |= late
    moved = (hist[-1] - hist[0]).norm(dim=1) if len(hist) > 1 else torch.full((B,), float("inf"), dtype=torch.float64, device=dev)
    dead = ~arrived & (moved < move_tol)
    ok = torch.ones(B, dtype=torch.bool, device=dev)
    if m > 1:
        ok &= mind >= cfg.dmin - coll_tol
    if obs is not None:
        ok &= clear >= cfg.margin - coll_tol
    return EpisodeResult(steps=steps, arrived=arrived.cpu().numpy(), arrival_step=arrival.cpu().numpy(), collision_free=ok.cpu().numpy(),
                         min_pair_distance=mind.cpu().numpy(), deadlocked=dead.cpu().numpy(), final_error=err.cpu().numpy(),
                         failed_solves=failed, total_solves=total,
                         mean_iters_by_step=torch.stack(its).cpu().numpy() if its else np.zeros(0),
                         states=torch.stack(states).cpu().numpy() if keep_states else None)


def simulate_closed_loop_fleets(cfg, x0, goals, max_steps: int, fleets: int = 4, **kw) -> EpisodeResult:
    """simulate_closed_loop for B swarms run as `fleets` independent fleets of ~B / fleets swarms: one handle, one HIP stream and one host thread per
    fleet.  A control period of one fleet lasts as long as its longest solve; the periods of the other fleets run on the SIMDs that tail leaves idle
    (six robots, 4096 swarms: 355 k -> 430-440 k solves/s as four fleets; which streams run concurrently is the runtime's choice).
    Every swarm's episode is what simulate_closed_loop computes for it: swarms are independent, only the dispatch-order hint acts per fleet.
    cfg: ProblemConfig; the other arguments as in simulate_closed_loop."""
    import threading
    from .distributed import shard_range
    x0 = np.asarray(x0.cpu() if hasattr(x0, "cpu") else x0, dtype=np.float64).reshape(-1, cfg.nx)
    goals = np.asarray(goals.cpu() if hasattr(goals, "cpu") else goals, dtype=np.float64)
    B = x0.shape[0]
    fleets = max(1, min(int(fleets), B))
    parts = [shard_range(B, f, fleets) for f in range(fleets)]
    solvers = [NmpcSolver(cfg, max_batch=hi - lo) for lo, hi in parts]
    torch = solvers[0].torch
    streams = [torch.cuda.Stream(device=solvers[0].device) for _ in parts]
    res, err = [None] * fleets, [None] * fleets

    def run(f):
        lo, hi = parts[f]
        try:
            with torch.cuda.device(solvers[f].device), torch.cuda.stream(streams[f]):
                res[f] = simulate_closed_loop(solvers[f], x0[lo:hi], goals[lo:hi], max_steps, **kw)
        except Exception as e:      # re-raised in the caller's thread
            err[f] = e
    th = [threading.Thread(target=run, args=(f,)) for f in range(fleets)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in err:
        if e is not None:
            raise e
    steps = max(r.steps for r in res)
    its = np.zeros(steps); wts = np.zeros(steps)
    for r, (lo, hi) in zip(res, parts):
        its[:r.steps] += r.mean_iters_by_step * (hi - lo); wts[:r.steps] += hi - lo
    cat = lambda k: np.concatenate([getattr(r, k) for r in res])
    states = None
    if res[0].states is not None:      # fleets that arrived early stop early: their last state is held
        states = np.concatenate([np.concatenate([r.states, np.repeat(r.states[-1:], steps + 1 - r.states.shape[0], axis=0)]) for r in res], axis=1)
    return EpisodeResult(steps=steps, arrived=cat("arrived"), arrival_step=cat("arrival_step"), collision_free=cat("collision_free"),
                         min_pair_distance=cat("min_pair_distance"), deadlocked=cat("deadlocked"), final_error=cat("final_error"),
                         failed_solves=sum(r.failed_solves for r in res), total_solves=sum(r.total_solves for r in res),
                         mean_iters_by_step=its / np.maximum(wts, 1), states=states)


@dataclass
class LidarEpisodeResult:
    steps: int                      # control periods executed
    arrived: "np.ndarray"           # [B] bool: the last goal was reached (V4:284-291: goal_idx > ng)
    arrival_step: "np.ndarray"      # [B] int: period at which that happened (-1: never)
    goals_reached: "np.ndarray"     # [B] int: goals reached so far
    min_clearance: "np.ndarray"     # [B] smallest distance from the robot's centre to an obstacle SURFACE over the episode (inf: no obstacle)
    final_error: "np.ndarray"       # [B] ||pose - xs|| at the end
    failed_solves: int              # solves that did not return status 0
    total_solves: int
    mean_iters_by_step: "np.ndarray"
    poses: Optional["np.ndarray"] = None     # [steps+1, B, 3] when keep_poses


def simulate_lidar_closed_loop(solver, pose0, goals, world, max_steps: int, arrive_tol: float = 0.2, scan_max: float = 3.5,
                               keep_poses: bool = False) -> LidarEpisodeResult:
    """The main loop of AllScripts/obs_avoid_static_first_scenario_v4.py (V4:209-300) for B simulated robots at once.

    Per robot and control period (file:line of the reference):
      scan <- LaserScan of the world around the robot            callback_lidar V4:29-36 (here: nmpc_lidar_scan_batch on a synthetic world)
      p = [x0 pose; xs; scan; B0]                                V4:230-236
      sol = solver(x0=guess, p=p, ...)                           V4:245
      guess <- shift(sol)                                        V4:258-270 (row N-1 appended, controls drop first / repeat last)
      robot moves with u[0]                                      V4:272-279 (here: the Euler model of the NLP itself, nmpc_lidar_plant_batch)
      ne = ||Xr - xs|| < 0.2 -> next goal; past the last: arrived   V4:281-291 (goal_idx, ng)
      x0[3:] = Scan ; x0[:3] = Xr                                V4:296-297
    pose0 [B,3]; goals [B,G,3] (or [B,3]); world [B,K,3] circular obstacles (ox, oy, radius).  As in the script a non-converged solve's control is
    applied as returned (counted in failed_solves); a robot that has arrived stays where it is and keeps solving its fixed point."""
    torch = solver.torch
    cfg = solver.cfg
    dev = solver.device
    pose = solver._dev(pose0, (-1, 3)).clone()
    B = pose.shape[0]
    g = torch.as_tensor(np.asarray(goals, dtype=np.float64) if not torch.is_tensor(goals) else goals, dtype=torch.float64, device=dev)
    if g.dim() == 2:
        g = g[:, None, :]
    if g.shape[0] != B or g.shape[2] != 3:
        raise ValueError(f"goals must be [B, 3] or [B, G, 3] with B={B}; got {tuple(g.shape)}")
    G = g.shape[1]
    wd = solver._dev(world, (B, -1, 3))
    K = wd.shape[1]
    ang = torch.arange(cfg.R, dtype=torch.float64, device=dev) * (2.0 * np.pi / cfg.R)      # B0 (V4:203-205)
    gi = torch.zeros(B, dtype=torch.long, device=dev)
    ar = torch.arange(B, device=dev)
    arrived = torch.zeros(B, dtype=torch.bool, device=dev)
    arrival = torch.full((B,), -1, dtype=torch.long, device=dev)
    clear = torch.full((B,), float("inf"), dtype=torch.float64, device=dev)

    def track(pc):
        nonlocal clear
        if K:
            clear = torch.minimum(clear, ((pc[:, None, :2] - wd[:, :, :2]).norm(dim=2) - wd[:, :, 2]).amin(dim=1))

    scan = solver.scan_batch(pose, wd, scan_max)
    # cold start of V4:184-196: X0 = repmat([pose; scan]), u0 = 0
    w = torch.cat([torch.cat([pose, scan], dim=1).repeat(1, cfg.N + 1), torch.zeros((B, 2 * cfg.Nc), dtype=torch.float64, device=dev)], dim=1)
    track(pose)
    poses = [pose.clone()] if keep_poses else None
    its = []
    failed = total = steps = 0
    for step in range(max_steps):
        if bool(arrived.all()):
            break
        xs = g[ar, gi]
        p = torch.cat([pose, xs, scan, ang[None, :].expand(B, -1)], dim=1).contiguous()
        r = solver.solve_batch(p, w)
        total += B
        failed += int((r["status"] != 0).sum())
        its.append(r["iters"].double().mean())
        w = solver.shift_batch(r["x"])
        pn = solver.plant_batch(p, r["x"])
        pose = torch.where(arrived[:, None], pose, pn)
        track(pose)
        # V4:281-291: ne = ||Xr - xs|| < 0.2 -> goal_idx + 1; beyond ng the robot has arrived
        hit = ((pose - xs).norm(dim=1) < arrive_tol) & ~arrived
        last = gi == G - 1
        newly = hit & last
        arrival = torch.where(newly, torch.full_like(arrival, step + 1), arrival)
        arrived |= newly
        gi = torch.where(hit & ~last, gi + 1, gi)
        scan = solver.scan_batch(pose, wd, scan_max)          # V4:296: x0[3:] = Scan
        if keep_poses:
            poses.append(pose.clone())
        steps += 1
    err = (pose - g[ar, gi]).norm(dim=1)
    return LidarEpisodeResult(steps=steps, arrived=arrived.cpu().numpy(), arrival_step=arrival.cpu().numpy(), goals_reached=(gi + arrived.long()).cpu().numpy(),
                              min_clearance=clear.cpu().numpy(), final_error=err.cpu().numpy(), failed_solves=failed, total_solves=total,
                              mean_iters_by_step=torch.stack(its).cpu().numpy() if its else np.zeros(0),
                              poses=torch.stack(poses).cpu().numpy() if keep_poses else None)
