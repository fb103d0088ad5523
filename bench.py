#!/usr/bin/env python3
"""bench.py — NMPC solves/sec (batched swarms) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one nmpc_solve_batch over one batch of synthetic swarm instances (cold start,
inputs already resident in HBM, result write-back to HBM included).  Workload at every N:
BASELINE.json configs[2] — 6 robots, horizon N=20, 15 pair rows per stage, batch 4096 per GPU
(weak scaling: instances are independent, each rank solves its own 4096; no data-path collective,
one all_gather of the per-rank timings/counters at the end).

Prints ONE JSON line on rank 0 (see README of the task for the contract) including
  roofline     — the solve kernel against the fp64 matrix/vector peak, algorithmic flops of
                 SURVEY.md §8(d): iters * (F_kkt + F_asm) per solve, duration from HIP events;
  cpu_baseline — the C oracle (oracle/nmpc_oracle.c, OpenMP, one instance per thread) timed on
                 this box's host cores on a bounded sample of the same workload ("port").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 vector = matrix peak
HBM_PEAK_GBS = 8000.0


def _composite(R):
    """six robots (C6 literals) + 8 circular obstacles in [-1.5, 1.5]^2, radii U[0.125, 0.2], rob_dim 0.2, margin 0.1, N = 25."""
    rng = np.random.default_rng(7)
    c = R.cfg_six(25); c.rob_dim = 0.2; c.margin = 0.1
    c.obstacles = [(float(x), float(y), float(r)) for x, y, r in zip(rng.uniform(-1.5, 1.5, 8), rng.uniform(-1.5, 1.5, 8), rng.uniform(0.125, 0.2, 8))]
    return c


def workload(name: str):
    """(oracle NLPConfig, batch per GPU, config index) — literals from the reference scripts."""
    from oracle import nlp_ref as R
    table = {
        "two": (R.cfg_two(20), 1024, 1),
        "six": (R.cfg_six(20), 4096, 2),
        "ten": (R.cfg_ten(30), 512, 3),
        "composite": (_composite(R), 1024, 4),      # BASELINE.json configs[4]: synthetic, no reference script (mismatch 2)
    }
    return table[name]


def algorithmic_flops_per_iter(cfg) -> float:
    """F_kkt + F_asm per interior-point iteration (dense Riccati count)."""
    nx, nu, N, m, M, K = cfg.nx, cfg.nu, cfg.N, cfg.m, cfg.M, cfg.K
    f_kkt = N * (7.0 / 3.0 * nx ** 3 + 4.0 * nx ** 2 * nu + 2.0 * nx * nu ** 2 + nu ** 3 / 3.0)
    f_asm = N * (22.0 * m + 14.0 * M + 16.0 * m * K)
    return f_kkt + f_asm


def algorithmic_bytes_per_solve(cfg) -> float:
    """minimal fp64 I/O of one solve."""
    return 8.0 * (2 * cfg.nx + 2 * cfg.n_var) + 16.0 + 8.0 * 3 * cfg.K


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="six", choices=["two", "six", "ten", "composite"])
    ap.add_argument("--batch", type=int, default=0, help="instances per GPU (default: the workload's)")
    ap.add_argument("--max-iter", type=int, default=2000)
    ap.add_argument("--closed-loop", type=int, default=20, help="warm closed-loop steps reported as an extra (0 = skip)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="instances for the CPU baseline (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import nmpc_amd
    from tests import helpers as Hh

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the solve has no CPU fallback")
    # NMPC_BENCH_REHEARSAL=1: every rank on device 0 with the gloo backend — rehearses the N > 1 code path on a one-GPU box
    rehearsal = bool(os.environ.get("NMPC_BENCH_REHEARSAL"))
    torch.cuda.set_device(0 if rehearsal else local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    ocfg, B, cidx = workload(args.workload)
    if args.batch:
        B = args.batch
    cfg = Hh.to_product_cfg(ocfg, max_iter=args.max_iter)
    # synthetic instances of SURVEY.md §8(d); each rank draws its own shard (seed + rank stream)
    rng = np.random.Generator(np.random.PCG64([Hh.SEED0 + cidx, rank]))
    P = np.stack([Hh.instance(rng, ocfg) for _ in range(B)])
    # instance 0 of every batch is the reference's literal start/goal set (C6:364-388, C2:213-224); the literal
    # x0 of the ten-robot script has coincident robots (status 3 by construction), so that workload keeps its drawn instance
    from oracle import nlp_ref as R
    lit = {"six": (R.C6_START, R.C6_GOAL), "two": (R.C2_START, R.C2_GOAL)}.get(args.workload)
    if lit is not None:
        P[0] = np.concatenate(lit)
    W0 = np.stack([nmpc_amd.cold_start(cfg, p[: cfg.nx]) for p in P])
    solver = nmpc_amd.NmpcSolver(cfg, max_batch=B)
    dP = torch.as_tensor(P, device="cuda"); dW0 = torch.as_tensor(W0, device="cuda")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        r = solver.solve_batch(dP, dW0)
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        r = solver.solve_batch(dP, dW0)
        ev[k][1].record()
    barrier()
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    iters = r["iters"].cpu().numpy(); status = r["status"].cpu().numpy(); kkt = r["kkt"].cpu().numpy()
    stats = torch.tensor([dt, float(iters.sum()), float((status == 0).sum()), float(iters.max()), float(kkt[status == 0].max() if (status == 0).any() else 0.0), kern_ms],
                         dtype=torch.float64, device="cuda")
    if world > 1:
        if rehearsal:
            stats = stats.cpu()
        allst = [torch.empty_like(stats) for _ in range(world)]
        dist.all_gather(allst, stats)
        allst = torch.stack(allst).cpu().numpy()
    else:
        allst = stats.cpu().numpy()[None]
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    t_max = float(allst[:, 0].max())
    total_solves = float(B) * world * args.steps
    value = total_solves / t_max
    sum_iters = float(allst[:, 1].sum())
    fl_iter = algorithmic_flops_per_iter(ocfg)
    # roofline of the dominant kernel (solve_kernel), per launch on rank 0
    flops_launch = float(allst[0, 1]) * fl_iter
    achieved = flops_launch / (allst[0, 5] * 1e-3) / 1e12
    out = {
        "metric": "NMPC solves/sec (batched swarms), N_robots=%d, N=%d" % (ocfg.m, ocfg.N),
        "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * t_max / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "centralized_%s_robots: m=%d, N=%d, %d pair rows/stage, %d obstacles, batch=%d per GPU, cold start" %
                   (args.workload, ocfg.m, ocfg.N, ocfg.M, ocfg.K, B), "batch_per_gpu": B, "max_iter": args.max_iter, "tol": cfg.tol},
        "solve_stats": {"mean_iters": sum_iters / (B * world), "max_iters": float(allst[:, 3].max()),
                        "converged_frac": float(allst[:, 2].sum()) / (B * world), "max_kkt_converged": float(allst[:, 4].max())},
        "roofline": {"bound": "mfma", "kernel": "nmpc::solve_lds_kernel<%d,...>" % ocfg.m, "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP64_PEAK_TFLOPS, "traffic": None,
                     "flops_per_launch": flops_launch, "kernel_ms": float(allst[0, 5]),
                     "algorithmic_bytes_per_launch": algorithmic_bytes_per_solve(ocfg) * B,
                     "note": "fp64 MFMA/VALU peak 78.6 TFLOP/s; algorithmic flops = iters*(F_kkt+F_asm) of SURVEY.md 8(d)"},
    }
    # HBM traffic of the solve kernel: FETCH_SIZE / WRITE_SIZE of the committed rocprofv3 --pmc passes (profiles/current,
    # collected with tools/collect_profiles.sh on this same command, corrected as MI355X_MICROARCH.md prescribes), stored per
    # interior-point iteration and scaled by the iterations of THIS launch; null when no pass exists for this workload
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "current", "hbm_traffic.json")))
        if tj["workload"]["workload"] == out["config"]["workload"]:
            out["roofline"]["traffic"] = tj["hbm_bytes_per_iteration"] * float(allst[0, 1])
            out["roofline"]["traffic_note"] = "bytes/launch = PMC bytes per iteration (profiles/current/hbm_traffic.json: 2*FETCH_SIZE + WRITE_SIZE, KB units) x iterations of this launch"
            out["roofline"]["traffic_GBps"] = out["roofline"]["traffic"] / (allst[0, 5] * 1e-3) / 1e9
    except (OSError, KeyError, ValueError):
        pass
    # warm closed loop of SURVEY.md 8(d): 20 receding-horizon steps, each = solve + device shift/plant step (a13 + a11);
    # and the same cold solve with HOST buffers at the boundary (H2D of p, w0 and D2H of w included).  Both are extras:
    # `value` above is the device-resident cold solve.
    if world == 1 and args.closed_loop > 0:
        nx = cfg.nx
        Pc, Wc = dP.clone(), dW0.clone()
        its, conv = [], []
        prev_it = None      # dispatch-order hint of nmpc_solve_batch_ordered: previous period's iteration counts, longest first
        torch.cuda.synchronize(); t2 = time.perf_counter()
        for _ in range(args.closed_loop):
            rr = solver.solve_batch(Pc, Wc, order=None if prev_it is None else torch.argsort(prev_it, descending=True))
            prev_it = rr["iters"]
            Wc, x0n = solver.shift_batch(Pc, rr["x"], plant=True)
            Pc = torch.cat([x0n, Pc[:, nx:]], dim=1)
            its.append(rr["iters"]); conv.append(rr["status"])
        torch.cuda.synchronize(); t_cl = time.perf_counter() - t2
        its = torch.stack(its).double().cpu().numpy(); conv = torch.stack(conv).cpu().numpy()
        out["closed_loop"] = {"steps": args.closed_loop, "solves_per_s": B * args.closed_loop / t_cl, "ms_per_step": 1e3 * t_cl / args.closed_loop,
                              "mean_iters_first_step": float(its[0].mean()), "mean_iters_later_steps": float(its[1:].mean()) if args.closed_loop > 1 else None,
                              "max_iters_later_steps": float(its[1:].max()) if args.closed_loop > 1 else None,
                              "converged_frac": float((conv == 0).mean()),
                              "note": "warm-started receding horizon: solve (dispatch order = previous period's iteration counts, longest first), then nmpc_shift_batch (plant step x0+T f(x0,u0) and guess shift) on device"}
        torch.cuda.synchronize(); t3 = time.perf_counter()
        rh = solver.solve_batch(P, W0)
        xh = rh["x"].cpu().numpy(); _ = rh["status"].cpu().numpy()
        t_h = time.perf_counter() - t3
        out["host_buffers"] = {"solves_per_s": B / t_h, "ms_per_step": 1e3 * t_h,
                               "note": "same cold batch with pageable host numpy buffers at the boundary: H2D of p and w0, solve, D2H of w/status"}
    # (measured last: the OpenMP team of the oracle keeps the host cores spinning for a while after it returns)
    # CPU baseline: the C oracle on this box's host cores, bounded sample of the same workload
    if world == 1 and args.cpu_sample != 0:
        from oracle import oracle_lib as O
        cores = O.max_threads()
        n = args.cpu_sample if args.cpu_sample > 0 else min(B, 16 * cores)
        oc = O.make_config(ocfg, max_iter=args.max_iter)
        O.solve_batch(oc, P[:cores], W0[:cores])                  # warm the threads / page in
        t1 = time.perf_counter()
        ref = O.solve_batch(oc, P[:n], W0[:n])
        t_cpu = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": n / t_cpu, "unit": "solves/s", "cores": cores, "kind": "port",
                               "sample": "first %d instances of the same batch, OpenMP one instance per thread, %.2f s; "
                                         "CPU restatement (oracle/nmpc_oracle.c), not CasADi/IPOPT" % (n, t_cpu),
                               "mean_iters": float(ref["iters"].mean())}
        # the GPU results of those instances agree with the oracle (same-basin fraction reported, not asserted here)
        dw = np.max(np.abs(r["x"][:n].cpu().numpy() - ref["x"]), axis=1)
        out["cpu_baseline"]["same_basin_frac_vs_gpu"] = float((dw <= 1e-6).mean())
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
