"""Development aid (GPU box): latency of the reference's own call, sol = solver(x0=, p=, lbx=, ubx=, lbg=, ubg=) for ONE swarm (C6:432), through nmpc_amd.nlpsol —
warm closed loop of the six-robot script's literal start / goal set, 40 periods.   python tools/call_latency.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nmpc_amd
import bench
for name, lit in (("six", bench.C6_LITERAL), ("two", bench.C2_LITERAL)):
    cfg = bench.workload(name)[0]
    solver = nmpc_amd.nlpsol("solver", "ipopt", cfg, {"ipopt": {"max_iter": 2000, "print_level": 0, "acceptable_tol": 1e-8, "acceptable_obj_change_tol": 1e-6}, "print_time": 0})
    lbx, ubx, lbg, ubg = cfg.bounds()
    x0 = np.array(lit[0], float); xs = np.array(lit[1], float)
    w = nmpc_amd.cold_start(cfg, x0)
    ts, kt, its = [], [], []
    for k in range(40):
        p = np.concatenate([x0, xs])
        torch.cuda.synchronize(); t = time.perf_counter()
        sol = solver(x0=w, p=p, lbx=lbx, ubx=ubx, lbg=lbg, ubg=ubg)
        ts.append(time.perf_counter() - t)
        its.append(solver.stats()["iter_count"])
        # the kernel alone: the same solve through the batched device API
        pd = torch.as_tensor(p[None], device="cuda"); wd = torch.as_tensor(np.asarray(w).reshape(1, -1), device="cuda")
        torch.cuda.synchronize(); t = time.perf_counter(); r = solver.solve_batch(pd, wd); torch.cuda.synchronize(); kt.append(time.perf_counter() - t)
        X = sol["x"].reshape(-1)
        N, nx, nu = cfg.N, cfg.nx, cfg.nu
        u = X[nx * (N + 1):].reshape(N, nu)
        Xs = X[:nx * (N + 1)].reshape(N + 1, nx)
        # plant step + shift (C6:450-465)
        th = x0[2::3]
        x0 = x0 + cfg.T * np.stack([u[0, 0::2] * np.cos(th), u[0, 0::2] * np.sin(th), u[0, 1::2]], 1).reshape(-1)
        w = np.concatenate([np.vstack([Xs[1:], Xs[N - 1:N]]).reshape(-1), np.vstack([u[1:], u[-1:]]).reshape(-1)])
    ts = np.array(ts[2:]) * 1e3; kt = np.array(kt[2:]) * 1e3; its = np.array(its[2:])
    print("%s robots, one swarm per call: solver(...) %.2f ms median (%.2f..%.2f), device solve alone %.2f ms, iterations median %d (%.3f ms per iteration of the call)" %
          (name, np.median(ts), ts.min(), ts.max(), np.median(kt), np.median(its), np.median(ts / np.maximum(its, 1))), flush=True)
