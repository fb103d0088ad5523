#!/usr/bin/env python3
"""bench.py — NMPC solves/sec (batched swarms) on MI355X.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no launcher environment (WORLD_SIZE unset) this process starts N ranks itself, as a child
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...`, before it
has made any GPU call, and exits with the child's code; under a launcher (RANK / LOCAL_RANK / WORLD_SIZE set) it is one rank.

A "step" is one nmpc_solve_batch over one batch of synthetic swarm instances (cold start, inputs already resident in HBM,
result write-back to HBM included).  Workload at every N: BASELINE.json configs[2] — 6 robots, horizon N=20, 15 pair rows per
stage, batch 4096 per GPU (weak scaling: instances are independent, each rank solves its own 4096; no data-path collective).
The one collective the path has, the result gather (RCCL all_gather of w_out / status / iters), is timed
separately after the solve steps and reported under "gather".

--scaling strong [--batch-total T]: ONE global batch of T instances (default: the batch BASELINE.json quotes for the workload — 4096
for six / ten robots, 8192 for the composite) is split into contiguous shards (nmpc_amd.shard_range), one per rank: what BASELINE
configs[3] / [4] describe (4096 over 8 GPUs = 512 per GPU, 8192 over 8 = 1024 per GPU).  The driver's default is weak scaling.

Prints ONE JSON line on rank 0 including
  roofline     — the solve kernel against the fp64 vector peak (the kernel issues no MFMA), algorithmic flops of
                 SURVEY.md §8(d): iters * (F_kkt + F_asm) per solve, duration from HIP events on the launch stream;
  cpu_baseline — the C oracle (oracle/nmpc_oracle.c, OpenMP, one instance per thread) timed on this box's host cores on a
                 bounded sample of the same workload ("port");
  sweep        — (N = 1 only) the other north-star shapes, each with its own roofline: m=2 (B=4096 and BASELINE configs[1]'s own 1024) and
                 m=10 at N=20, B=4096; m=10 at N=30, B=512 (one GPU's shard of BASELINE configs[3]) and B=4096 (the whole batch on one GPU); the six-robot +
                 eight-obstacle composite; six robots at B=16384 (the launch outgrows its longest solve); the LIDAR-state NLP with
                 its own flop roofline and CPU baseline;
  two_streams  — (N = 1 only) the same batch through two handles on two HIP streams, launches alternating: sustained rate when a second launch may
                 run on the SIMDs the first one's tail leaves idle (an extra; `value` is one launch at a time).

--streams S (default 1): the K timed steps alternate over S handles, each on its own HIP stream — S launches in flight; `value` is then the
rate of that pipeline (six robots, B = 4096: 278 k at S = 1, 476 k at S = 2, 497 k at S = 3) and `config.streams` says so.  The default stays one
launch at a time, as every round has reported it; the roofline block always prices ONE launch (its own duration, contended when S > 1).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 vector = matrix peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


# ---- synthetic workload of SURVEY.md 8(d), built from the PRODUCT side (nmpc_amd presets): nothing here imports oracle/ or tests/
SEED0 = 20210141
# the scripts' literal start / goal sets, instance 0 of their batches (C6:364-388, C2:213-224); tests/test_abi_host.py checks them against the oracle's table
C2_LITERAL = ([-0.7112, -0.7112, 0.785, 0.7112, 0.7112, -2.356], [0.7112, 0.7112, 0.785, -0.7112, -0.7112, -2.356])
C6_LITERAL = ([0.7, 0.4, -2.618, 0.0, 0.8, -1.57, -0.7, 0.4, -0.523, -0.7, -0.4, 0.523, 0.0, -0.8, 1.57, 0.7, -0.4, 2.618],
              [-0.7, -0.4, -2.618, 0.0, -0.8, -1.57, 0.7, -0.4, -0.523, 0.7, 0.4, 0.523, 0.0, 0.8, 1.57, -0.7, 0.4, 2.618])


def composite_obstacles():
    """8 circular obstacles in [-1.5, 1.5]^2, radii U[0.125, 0.2] (BASELINE.json configs[4]: synthetic, no reference script)."""
    rng = np.random.default_rng(7)
    return [(float(x), float(y), float(r)) for x, y, r in zip(rng.uniform(-1.5, 1.5, 8), rng.uniform(-1.5, 1.5, 8), rng.uniform(0.125, 0.2, 8))]


def workload(name: str):
    """(product ProblemConfig, batch per GPU, config index) — literals from the reference scripts (nmpc_amd presets cite the lines)."""
    import nmpc_amd
    table = {
        "two": (lambda: nmpc_amd.centralized_two_robots(N=20), 4096, 1),            # north_star: N_robots=2, N=20, batch 4096 (BASELINE configs[1] quotes 1024)
        "six": (lambda: nmpc_amd.centralized_six_robots(N=20), 4096, 2),            # BASELINE configs[2], the headline
        "ten20": (lambda: nmpc_amd.ten_robots_collision_avoidance(N=20), 4096, 3),  # north_star: N_robots=10, N=20, batch 4096 (the file's own horizon)
        "ten": (lambda: nmpc_amd.ten_robots_collision_avoidance(N=30), 512, 3),     # BASELINE configs[3]: N=30, 512 per GPU
        "composite": (lambda: nmpc_amd.six_robots_eight_obstacles(N=25, obstacles=composite_obstacles()), 1024, 4),   # BASELINE configs[4] (mismatch 2)
    }
    mk, B, cidx = table[name]
    return mk(), B, cidx


def _sample_points(rng, m, dsep, lim=2.0, obstacles=(), clear=0.0):
    pts = []
    while len(pts) < m:
        c = rng.uniform(-lim, lim, 2)
        if all(np.hypot(*(c - q)) >= dsep for q in pts) and all(np.hypot(c[0] - ox, c[1] - oy) >= clear + orad for (ox, oy, orad) in obstacles):
            pts.append(c)
    return np.array(pts)


def instance(rng, cfg):
    """p = [x0; xs] of SURVEY.md 8(d): starts / goals uniform in [-2, 2]^2 with pairwise distance >= dmin + 0.1 (and clear of every obstacle), theta uniform."""
    m = cfg.m
    dsep = cfg.dmin + 0.1
    clear = cfg.rob_dim + cfg.margin + 0.05
    s = _sample_points(rng, m, dsep, obstacles=cfg.obstacles, clear=clear)
    g = _sample_points(rng, m, dsep, obstacles=cfg.obstacles, clear=clear)
    x0 = np.concatenate([s, rng.uniform(-np.pi, np.pi, (m, 1))], axis=1).reshape(-1)
    xs = np.concatenate([g, rng.uniform(-np.pi, np.pi, (m, 1))], axis=1).reshape(-1)
    return np.concatenate([x0, xs])


# global batch of --scaling strong: what BASELINE.json quotes for the workload (configs[1..4])
STRONG_TOTAL = {"two": 1024, "six": 4096, "ten20": 4096, "ten": 4096, "composite": 8192}


def algorithmic_flops_per_iter(cfg) -> float:
    """F_kkt + F_asm per interior-point iteration (dense Riccati count)."""
    nx, nu, N, m, M, K = cfg.nx, cfg.nu, cfg.N, cfg.m, cfg.M, len(cfg.obstacles)
    f_kkt = N * (7.0 / 3.0 * nx ** 3 + 4.0 * nx ** 2 * nu + 2.0 * nx * nu ** 2 + nu ** 3 / 3.0)
    f_asm = N * (22.0 * m + 14.0 * M + 16.0 * m * K)
    return f_kkt + f_asm


def algorithmic_bytes_per_solve(cfg) -> float:
    """minimal fp64 I/O of one solve."""
    return 8.0 * (2 * cfg.nx + 2 * cfg.n_var) + 16.0 + 8.0 * 3 * len(cfg.obstacles)


def _free_port() -> int:
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def self_launch(args) -> int:
    """--gpus N > 1 without a launcher: start N ranks as a child torch.distributed.run (this process has not touched the GPU and
    never will; it only relays the child's output and exit code)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def make_batch(name, rank, batch=0, shard=None, max_iter=2000):
    """(product config, B, P, W0).  shard = (lo, hi): the instances [lo, hi) of ONE global batch of `batch` instances drawn from the rank-0
    stream (strong scaling); otherwise every rank draws its own `batch` instances (weak scaling)."""
    import nmpc_amd
    cfg, B, cidx = workload(name)
    cfg.max_iter = max_iter
    if batch:
        B = batch
    # synthetic instances of SURVEY.md §8(d); each rank draws its own shard (seed + rank stream)
    rng = np.random.Generator(np.random.PCG64([SEED0 + cidx, 0 if shard else rank]))
    P = np.stack([instance(rng, cfg) for _ in range(B)])
    # instance 0 of every batch is the reference's literal start/goal set (C6:364-388, C2:213-224); the literal
    # x0 of the ten-robot script has coincident robots (status 3 by construction), so that workload keeps its drawn instance
    lit = {"six": C6_LITERAL, "two": C2_LITERAL}.get(name)
    if lit is not None:
        P[0] = np.concatenate([np.array(lit[0]), np.array(lit[1])])
    W0 = np.stack([nmpc_amd.cold_start(cfg, p[: cfg.nx]) for p in P])
    if shard:
        lo, hi = shard
        return cfg, hi - lo, P[lo:hi], W0[lo:hi]
    return cfg, B, P, W0


def timed_solves(solver, dP, dW0, steps, warmup, barrier, more=()):
    """W untimed + K timed nmpc_solve_batch launches; returns (wall seconds between the barriers, mean kernel ms from HIP
    events recorded on the launch stream, result of the last step).  more: further handles (--streams S > 1): the K launches alternate over
    1 + len(more) handles, each on its own HIP stream (S launches in flight)."""
    import torch
    if more:
        pool = [solver] + list(more)
        sts = [torch.cuda.Stream() for _ in pool]
        r = None
        for k in range(max(warmup, len(pool))):          # every handle and stream once before the clock starts
            with torch.cuda.stream(sts[k % len(pool)]):
                r = pool[k % len(pool)].solve_batch(dP, dW0)
        barrier()
        ev = []
        t0 = time.perf_counter()
        for k in range(steps):
            st = sts[k % len(pool)]
            with torch.cuda.stream(st):
                a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
                a.record(st)
                r = pool[k % len(pool)].solve_batch(dP, dW0)
                b.record(st)
                ev.append((a, b))
        barrier()
        dt = time.perf_counter() - t0
        return dt, float(np.mean([a.elapsed_time(b) for a, b in ev])), r
    r = None
    for _ in range(warmup):
        r = solver.solve_batch(dP, dW0)
    barrier()
    st = torch.cuda.current_stream()      # the stream nmpc_solve_batch launches on (NmpcSolver._stream)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for k in range(steps):
        ev[k][0].record(st)
        r = solver.solve_batch(dP, dW0)
        ev[k][1].record(st)
    barrier()
    dt = time.perf_counter() - t0
    return dt, float(np.mean([a.elapsed_time(b) for a, b in ev])), r


def two_stream_rate(solvers, dP, dW0, ref, launches=8):
    """the same cold batch through TWO handles (two workspaces) on two HIP streams, launches alternating: (solves/s, ms per launch, results equal
    to `ref`).  A launch lasts as long as its longest solve; the other stream's wavefronts run on the SIMDs that tail leaves idle."""
    import torch
    sts = [torch.cuda.Stream(), torch.cuda.Stream()]
    for k in range(2):
        with torch.cuda.stream(sts[k]):
            solvers[k].solve_batch(dP, dW0)
    torch.cuda.synchronize()
    t = time.perf_counter()
    res = []
    for k in range(launches):
        with torch.cuda.stream(sts[k % 2]):
            res.append(solvers[k % 2].solve_batch(dP, dW0))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    same = all(torch.equal(rk["iters"], ref["iters"]) and torch.equal(rk["status"], ref["status"]) for rk in res)
    return dP.shape[0] * launches / dt, 1e3 * dt / launches, bool(same)


FP64_SUSTAINED_TFLOPS = 47.7   # measured on this pool with every SIMD busy (tools/valu_probe.hip: the box clocks ~1.45 GHz under sustained fp64 load, not the 2.4 GHz of the peak)
ROOFLINE_NOTE = ("roofline blocks: compute/latency-bound fp64 kernels with zero MFMA instructions, priced against the fp64 VECTOR peak 78.6 TFLOP/s (= the fp64 matrix "
                 "peak on MI355X); achieved = algorithmic flops iters*(F_kkt+F_asm) of SURVEY.md 8(d) / kernel time from HIP events on the launch stream; frac_sustained = "
                 "achieved / 47.7 TFLOP/s (what tools/valu_probe.hip sustains on this pool); traffic = PMC bytes per iteration of the committed profile "
                 "(2*FETCH_SIZE + WRITE_SIZE, KB units, separate passes) x iterations of this launch, null unless the profile was taken from this very build, batch and kernel shape")


def profile_dir(wname, B):
    """committed rocprofv3 summaries (tools/collect_profiles.sh + tools/summarize_profiles.py) per workload AND batch: profiles/<dir>/hbm_traffic.json"""
    base = {"six": "current", "two": "current_two", "ten20": "current_ten20", "ten": "current_ten", "composite": "current_composite", "lidar": "current_lidar"}[wname]
    default_b = {"six": 4096, "two": 4096, "ten20": 4096, "ten": 512, "composite": 1024, "lidar": 4096}[wname]
    return base if B == default_b else "%s_b%d" % (base, B)


def roofline_block(cfg, B, sum_iters, kern_ms, lib_version, kernel_id, wname="six"):
    fl_iter = algorithmic_flops_per_iter(cfg)
    flops_launch = float(sum_iters) * fl_iter
    achieved = flops_launch / (kern_ms * 1e-3) / 1e12
    # the kernel nmpc_solve_batch picked for this team size and batch, as the library reports it (nmpc_query):
    # 3 column-per-lane, throughput shape (one wavefront per instance); 4 the same kernel's latency shape (two wavefronts per instance, four
    # where the library's rule says so: the composite); 2 element-per-lane; 1 HBM-resident fallback
    kname = {3: "nmpc::solve_col_kernel<%d,...,64>", 4: "nmpc::solve_col_kernel<%d,...,128|256> (latency shape)",
             2: "nmpc::solve_lds_kernel<%d,...>", 1: "nmpc::solve_kernel<%d,...>"}.get(kernel_id, "nmpc::solve_?_kernel<%d>") % cfg.m
    rl = {"bound": "fp64-valu", "kernel": kname, "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
          "frac": achieved / FP64_PEAK_TFLOPS, "frac_sustained": achieved / FP64_SUSTAINED_TFLOPS, "traffic": None, "flops_per_launch": flops_launch, "kernel_ms": kern_ms,
          "algorithmic_bytes_per_launch": algorithmic_bytes_per_solve(cfg) * B}
    # HBM traffic of the solve kernel from the committed PMC passes; bytes per iteration and instance DO depend on the batch (cache residency
    # of the per-instance workspaces) and on the kernel shape (the latency shape keeps the slacks and duals in LDS): only a profile taken from
    # this build, at this batch and in this shape is used
    try:
        pdir = profile_dir(wname, B)
        tj = json.load(open(os.path.join(ROOT, "profiles", pdir, "hbm_traffic.json")))
        same_lib = tj.get("library_src_hash") and ("src=" + tj["library_src_hash"]) in lib_version
        wl = tj["workload"]
        if same_lib and wl.get("m") == cfg.m and wl.get("N") == cfg.N and wl.get("batch_per_gpu") == B and wl.get("kernel_id", kernel_id) == kernel_id:
            rl["traffic"] = tj["hbm_bytes_per_iteration"] * float(sum_iters)
            rl["traffic_GBps"] = rl["traffic"] / (kern_ms * 1e-3) / 1e9
            rl["traffic_source"] = "profiles/%s@src=%s" % (pdir, tj["library_src_hash"])
        else:
            rl["traffic_source"] = "null: profiles/%s is of another build, batch or kernel shape (src=%s)" % (pdir, tj.get("library_src_hash"))
    except (OSError, KeyError, ValueError):
        rl["traffic_source"] = "null: no committed profile for this workload and batch"
    return rl


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="six", choices=["two", "six", "ten", "ten20", "composite"])
    ap.add_argument("--batch", type=int, default=0, help="instances per GPU (default: the workload's)")
    ap.add_argument("--max-iter", type=int, default=2000)
    ap.add_argument("--closed-loop", type=int, default=20, help="warm closed-loop steps reported as an extra (0 = skip)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="instances for the CPU baseline (0 = skip)")
    ap.add_argument("--sweep", type=int, default=-1, help="north-star sweep entries (default: on for --gpus 1, 0 = skip)")
    ap.add_argument("--streams", type=int, default=1, help="launches in flight per GPU: the K timed steps alternate over this many handles, each on its own HIP stream "
                    "(default 1: one launch at a time, as every round has reported `value`; 2 fills the SIMDs a launch's tail leaves idle)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="weak: every rank its own batch (default); strong: one global batch, sharded")
    ap.add_argument("--batch-total", type=int, default=0, help="--scaling strong: instances of the global batch (default: the workload's BASELINE batch)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))        # before any GPU call

    import torch
    import torch.distributed as dist
    import nmpc_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the solve has no CPU fallback")
    # NMPC_BENCH_REHEARSAL=1: every rank on device 0 with the gloo backend — rehearses the N > 1 code path on a one-GPU box
    rehearsal = bool(os.environ.get("NMPC_BENCH_REHEARSAL"))
    if not rehearsal and world > torch.cuda.device_count():
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} devices are visible")
    torch.cuda.set_device(0 if rehearsal else local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, expected {args.gpus}")
    lib_version = nmpc_amd._lib.load().nmpc_version().decode()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.scaling == "strong":
        total = args.batch_total or STRONG_TOTAL[args.workload]
        cfg, B, P, W0 = make_batch(args.workload, rank, total, shard=nmpc_amd.shard_range(total, rank, world), max_iter=args.max_iter)
    else:
        total = None
        cfg, B, P, W0 = make_batch(args.workload, rank, args.batch, max_iter=args.max_iter)
    nM, nK = cfg.M, len(cfg.obstacles)
    solver = nmpc_amd.NmpcSolver(cfg, max_batch=B)
    kernel_id = int(solver.kernel_for_batch(B))
    dP = torch.as_tensor(P, device="cuda"); dW0 = torch.as_tensor(W0, device="cuda")
    more = [nmpc_amd.NmpcSolver(cfg, max_batch=B) for _ in range(max(1, args.streams) - 1)]      # --streams S: S handles, S launches in flight (default 1)
    dt, kern_ms, r = timed_solves(solver, dP, dW0, args.steps, args.warmup, barrier, more)
    del more

    iters = r["iters"].cpu().numpy(); status = r["status"].cpu().numpy(); kkt = r["kkt"].cpu().numpy()
    # ---- the result gather of SURVEY.md 8(e): all_gather of w_out / status / iters of every rank's shard, timed on its own
    gather_ms = 0.0
    if world > 1:
        Bg = total if total is not None else B * world      # strong: the shards of the global batch (sizes may differ by one)
        torch.cuda.synchronize()
        for rep in range(4):          # first repetition warms the communicator up
            barrier()
            tg = time.perf_counter()
            gw = nmpc_amd.gather_results(r["x"] if not rehearsal else r["x"].cpu(), Bg)
            gs = nmpc_amd.gather_results((r["status"] if not rehearsal else r["status"].cpu())[:, None], Bg)
            gi = nmpc_amd.gather_results((r["iters"] if not rehearsal else r["iters"].cpu())[:, None], Bg)
            torch.cuda.synchronize()
            if rep:
                gather_ms += (time.perf_counter() - tg) * 1e3 / 3.0
        assert gw.shape == (Bg, cfg.n_var) and gs.shape == (Bg, 1) and gi.shape == (Bg, 1)
        lo = nmpc_amd.shard_range(Bg, rank, world)[0]
        assert torch.equal(gw[lo: lo + B].to(r["x"].device), r["x"]), "gathered shard differs from the local result"
    stats = torch.tensor([dt, float(iters.sum()), float((status == 0).sum()), float(iters.max()), float(kkt[status == 0].max() if (status == 0).any() else 0.0), kern_ms, gather_ms, float(B)],
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
    n_inst = float(allst[:, 7].sum())                 # instances of all ranks (weak: B per rank; strong: the shards of one global batch)
    total_solves = n_inst * args.steps
    value = total_solves / t_max
    sum_iters = float(allst[:, 1].sum())
    out = {
        "metric": "NMPC solves/sec (batched swarms), N_robots=%d, N=%d" % (cfg.m, cfg.N),
        "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * t_max / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "centralized_%s_robots: m=%d, N=%d, %d pair rows/stage, %d obstacles, batch=%d per GPU, cold start" %
                   (args.workload, cfg.m, cfg.N, nM, nK, B), "m": cfg.m, "N": cfg.N, "batch_per_gpu": B, "batch_total": int(n_inst), "max_iter": args.max_iter, "tol": cfg.tol,
                   "kernel_id": kernel_id, "streams": max(1, args.streams), "ranks": world, "backend": (dist.get_backend() if world > 1 else "none")},
        "library": lib_version,
        "solve_stats": {"mean_iters": sum_iters / n_inst, "max_iters": float(allst[:, 3].max()),
                        "converged_frac": float(allst[:, 2].sum()) / n_inst, "max_kkt_converged": float(allst[:, 4].max())},
        "roofline": roofline_block(cfg, B, float(allst[0, 1]), float(allst[0, 5]), lib_version, kernel_id, args.workload),
        "note": ROOFLINE_NOTE,
    }
    if world > 1:
        per_rank_bytes = B * (cfg.n_var * 8 + 8)
        gms = float(allst[:, 6].max())
        out["gather"] = {"collective": "all_gather of w_out [B, n_var] f64 + status + iters (nmpc_amd.gather_results)", "backend": dist.get_backend(),
                         "ms": gms, "bytes_per_rank_shard": per_rank_bytes, "recv_GBps_per_rank": per_rank_bytes * (world - 1) / (gms * 1e-3) / 1e9 if gms > 0 else None,
                         "note": "timed separately from the solve steps (mean of 3 after a warm-up); not part of `value`, which is the device-resident solve"}
    # warm closed loop of SURVEY.md 8(d): 20 receding-horizon steps, each = solve + device shift/plant step (a13 + a11);
    # and the same cold solve with HOST buffers at the boundary (H2D of p, w0 and D2H of w included).  Both are extras:
    # `value` above is the device-resident cold solve.
    if world == 1 and args.closed_loop > 0:
        try:
            nx = cfg.nx
            Pc, Wc = dP.clone(), dW0.clone()
            its, conv = [], []
            # nmpc_step_batch: solve + shift + plant step + the next period's dispatch order (longest solves first), all on the device;
            # the host only enqueues one call per period
            order = torch.arange(B, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize(); t2 = time.perf_counter()
            for _ in range(args.closed_loop):
                rr = solver.step_batch(Pc, Wc, order)
                its.append(rr["iters"]); conv.append(rr["status"])
            torch.cuda.synchronize(); t_cl = time.perf_counter() - t2
            its = torch.stack(its).double().cpu().numpy(); conv = torch.stack(conv).cpu().numpy()
            out["closed_loop"] = {"steps": args.closed_loop, "solves_per_s": B * args.closed_loop / t_cl, "ms_per_step": 1e3 * t_cl / args.closed_loop,
                                  "mean_iters_first_step": float(its[0].mean()), "mean_iters_later_steps": float(its[1:].mean()) if args.closed_loop > 1 else None,
                                  "max_iters_later_steps": float(its[1:].max()) if args.closed_loop > 1 else None,
                                  "converged_frac": float((conv == 0).mean()),
                                  "note": "warm receding horizon, one nmpc_step_batch per period (solve in longest-first order, shift, plant step, next order: all on the device)"}
            # the same closed loop as FOUR independent fleets of B/4 swarms, one handle and one HIP stream each: a period of one fleet overlaps the tails of the
            # others.  The runtime maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and streams that share a queue serialise: which
            # four streams run concurrently is the runtime's choice (measured: the first four streams of the pool 280-300 k, the next four 430-440 k), so two
            # sets of four are timed and both rates reported
            nfl = 4
            Bf = B // nfl
            fsol = [nmpc_amd.NmpcSolver(cfg, max_batch=Bf) for _ in range(nfl)]
            pool = [torch.cuda.Stream() for _ in range(2 * nfl)]
            rates, fmean = [], 0.0
            for fst in (pool[:nfl], pool[nfl:]):
                fP = [dP[f * Bf:(f + 1) * Bf].clone() for f in range(nfl)]; fW = [dW0[f * Bf:(f + 1) * Bf].clone() for f in range(nfl)]
                ford = [torch.arange(Bf, dtype=torch.int32, device="cuda") for _ in range(nfl)]
                for f in range(nfl):          # one untimed period per fleet: first launch of this kernel shape on this stream
                    with torch.cuda.stream(fst[f]):
                        fsol[f].step_batch(fP[f].clone(), fW[f].clone(), ford[f].clone())
                fits = []
                torch.cuda.synchronize(); t2 = time.perf_counter()
                for _ in range(args.closed_loop):
                    for f in range(nfl):
                        with torch.cuda.stream(fst[f]):
                            fits.append(fsol[f].step_batch(fP[f], fW[f], ford[f])["iters"])
                torch.cuda.synchronize(); rates.append(Bf * nfl * args.closed_loop / (time.perf_counter() - t2))
                fmean = float(torch.stack([i.double().mean() for i in fits]).mean().item())
            out["closed_loop"]["four_fleets"] = {"solves_per_s": max(rates), "solves_per_s_by_stream_set": rates, "ms_per_period": 1e3 * Bf * nfl / max(rates), "mean_iters": fmean,
                                                 "note": "the same swarms as four fleets of %d, one handle and one HIP stream each; two sets of four streams timed (streams that the runtime maps "
                                                         "to one hardware queue serialise), the better one reported" % Bf}
            del fsol, fP, fW, fits
            torch.cuda.synchronize(); t3 = time.perf_counter()
            rh = solver.solve_batch(P, W0)
            xh = rh["x"].cpu().numpy(); _ = rh["status"].cpu().numpy()
            t_h = time.perf_counter() - t3
            out["host_buffers"] = {"solves_per_s": B / t_h, "ms_per_step": 1e3 * t_h,
                                   "note": "same cold batch, pageable host buffers at the boundary (H2D p, w0; D2H w, status): PCIe-inclusive, never `value`"}
        except Exception as e_:      # an extra must never cost the record its headline: note the failure and go on
            out.setdefault("extras_failed", []).append("closed_loop: %r" % (e_,))
    # two launches in flight: the same cold batch solved through TWO handles (two workspaces) on two HIP streams, launches alternating.  A launch of
    # 4096 lasts as long as its longest solve; the second stream's wavefronts run on the SIMDs the first launch's tail leaves idle.  An extra:
    # `value` above stays one launch at a time on one stream.
    if world == 1 and args.closed_loop > 0:
        try:
            s2 = nmpc_amd.NmpcSolver(cfg, max_batch=B)
            v2, ms2, same = two_stream_rate([solver, s2], dP, dW0, r)
            out["two_streams"] = {"solves_per_s": v2, "ms_per_launch": ms2, "launches": 8, "same_iterations_as_value_run": same,
                                  "note": "two handles on two HIP streams, launches alternating (the second launch fills the SIMDs the first one's tail leaves idle); an extra, never `value`"}
            del s2
        except Exception as e_:      # an extra must never cost the record its headline: note the failure and go on
            out.setdefault("extras_failed", []).append("closed_loop: %r" % (e_,))
    # north-star sweep: N_robots in {2, 6, 10}, N=20, batch 4096 (+ BASELINE configs[3] and [4]); one warm-up + two timed launches each
    do_sweep = args.sweep if args.sweep >= 0 else (1 if (world == 1 and args.workload == "six" and not args.batch) else 0)
    if world == 1 and do_sweep:
        try:
            del solver
            out["sweep"] = []
            # (workload, batch; 0 = the workload's own): the four north-star shapes, then ten robots at N=30 with the whole BASELINE batch on one
            # GPU (what strong sharding to 512 per GPU is compared with) and six robots at B=16384, where the launch outgrows its longest solve
            for name, bsz in (("two", 0), ("two", 1024), ("ten20", 0), ("ten", 0), ("composite", 0), ("ten", 4096), ("six", 16384)):      # ("two", 1024): BASELINE configs[1]'s own batch
                c2, B2, P2, W2 = make_batch(name, 0, bsz, max_iter=args.max_iter)
                s2 = nmpc_amd.NmpcSolver(c2, max_batch=B2)
                d2, k2, r2 = timed_solves(s2, torch.as_tensor(P2, device="cuda"), torch.as_tensor(W2, device="cuda"), 2, 1, barrier)
                it2 = r2["iters"].cpu().numpy(); st2 = r2["status"].cpu().numpy()
                out["sweep"].append({"workload": "%s: m=%d, N=%d, K=%d, batch=%d, cold" % (name, c2.m, c2.N, len(c2.obstacles), B2),
                                     "m": c2.m, "N": c2.N, "batch": B2, "value": B2 * 2 / d2, "unit": "solves/s", "ms_per_step": 1e3 * d2 / 2,
                                     "mean_iters": float(it2.mean()), "max_iters": float(it2.max()), "converged_frac": float((st2 == 0).mean()),
                                     "status_counts": {str(k): int((st2 == k).sum()) for k in np.unique(st2)},
                                     "roofline": roofline_block(c2, B2, float(it2.sum()), k2, lib_version, int(s2.kernel_for_batch(B2)), name)})
                if B2 <= 4096:        # two launches in flight (see two_stream_rate): what a second stream recovers of this shape's tail
                    s3 = nmpc_amd.NmpcSolver(c2, max_batch=B2)
                    v3, ms3, same3 = two_stream_rate([s2, s3], torch.as_tensor(P2, device="cuda"), torch.as_tensor(W2, device="cuda"), r2, 4)
                    out["sweep"][-1]["two_streams"] = {"solves_per_s": v3, "ms_per_launch": ms3, "same_iterations": same3}
                    del s3
                del s2
                torch.cuda.empty_cache()
        except Exception as e_:      # an extra must never cost the record its headline: note the failure and go on
            out.setdefault("extras_failed", []).append("sweep: %r" % (e_,))
    # LIDAR-ray distance-state NMPC (the file BASELINE configs[4] names, AllScripts/obs_avoid_static_first_scenario_v4.py: one robot,
    # 13 states, N=100, Nc=50; SURVEY.md 0 mismatch 2): one wavefront per instance, the instance's workspace lives in HBM/L2
    if world == 1 and do_sweep:
        try:
            lc = nmpc_amd.lidar_v4()
            Bl = 4096
            rngl = np.random.Generator(np.random.PCG64(SEED0 + 5))
            poses, worlds, goals = [], [], []
            for _ in range(Bl):
                poses.append([rngl.uniform(0.0, 0.15), rngl.uniform(0.0, 0.15), rngl.uniform(0.4, 1.1)])
                worlds.append([(float(rngl.uniform(0.8, 2.6)), float(rngl.uniform(0.3, 2.4)), float(rngl.uniform(0.15, 0.3))) for _ in range(3)])
                goals.append(np.array([3.0, 2.5, 0.0]) + rngl.uniform(-0.3, 0.3, 3))
            poses = np.array(poses); worlds = np.array(worlds)
            lbx, ubx = lc.bounds()[:2]
            ls = nmpc_amd.LidarSolver(lc, lbx=lbx, ubx=ubx, max_batch=Bl)
            scans = ls.scan_batch(poses, worlds).cpu().numpy()          # the synthetic LaserScan of V4:29-36, on the device (nmpc_lidar_scan_batch)
            Pl = np.stack([nmpc_amd.lidar_params(lc, poses[b], goals[b], scans[b]) for b in range(Bl)])
            Wl = np.stack([nmpc_amd.lidar_cold_start(lc, np.concatenate([poses[b], scans[b]])) for b in range(Bl)])
            dl, kl, rl_ = timed_solves(ls, torch.as_tensor(Pl, device="cuda"), torch.as_tensor(Wl, device="cuda"), 2, 1, barrier)
            itl = rl_["iters"].cpu().numpy(); stl = rl_["status"].cpu().numpy()
            alg_bytes = (8.0 * (lc.n_p + 2 * lc.n_var) + 16.0) * Bl
            # Algorithmic flops per interior-point iteration of the LIDAR-state NLP, the same dense-Riccati count as SURVEY.md
            # 8(d) applied to what the kernel factors: the 3-state / 2-control pose recursion (the R distance states of a stage are eliminated
            # through their own linearised rows): F_ric = N (7/3 3^3 + 4 3^2 2 + 2 3 2^2 + 2^3/3); folding the distance rows into the pose block
            # and recovering their steps: F_fold = N R (2 (3 3 + 3) + 2 3); evaluation / assembly: F_asm = N (22 + 14 R)
            fl_iter_l = lc.N * ((7.0 / 3.0) * 27 + 4 * 9 * 2 + 2 * 3 * 4 + 8.0 / 3.0) + lc.N * lc.R * 30.0 + lc.N * (22.0 + 14.0 * lc.R)
            ach = fl_iter_l * float(itl.sum()) / (kl * 1e-3) / 1e12
            out["sweep"].append({"workload": "lidar_v4: 1 robot, 13 states (pose + 10 ray distances), N=100, Nc=50, batch=%d, cold start" % Bl,
                                 "m": 1, "N": lc.N, "batch": Bl, "value": Bl * 2 / dl, "unit": "solves/s", "ms_per_step": 1e3 * dl / 2,
                                 "mean_iters": float(itl.mean()), "max_iters": float(itl.max()), "converged_frac": float((stl == 0).mean()),
                                 "status_counts": {str(k): int((stl == k).sum()) for k in np.unique(stl)},
                                 "roofline": {"bound": "fp64-valu", "kernel": "nmpc_lidar::lidar_solve_kernel", "kernel_ms": kl,
                                              "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS, "traffic": None,
                                              "flops_per_iteration": fl_iter_l, "flops_per_launch": fl_iter_l * float(itl.sum()),
                                              "algorithmic_bytes_per_launch": alg_bytes,
                                              "hbm_frac_of_algorithmic_bytes": alg_bytes / (kl * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                              "frac_sustained": ach / FP64_SUSTAINED_TFLOPS,
                                              "note": "flops = iters * (F_ric + F_fold + F_asm) of the reduced 3-state recursion (DESIGN.md 4.5); latency-bound: one wavefront per instance, serial recursions over 100 stages"}})
            ls2 = nmpc_amd.LidarSolver(lc, lbx=lbx, ubx=ubx, max_batch=Bl)
            v3, ms3, same3 = two_stream_rate([ls, ls2], torch.as_tensor(Pl, device="cuda"), torch.as_tensor(Wl, device="cuda"), rl_, 4)
            out["sweep"][-1]["two_streams"] = {"solves_per_s": v3, "ms_per_launch": ms3, "same_iterations": same3}
            del ls2
            # PMC traffic of the LIDAR kernel (profiles/current_lidar, same stamp rule as the main kernel): bytes per iteration x iterations of this launch
            try:
                pdl = profile_dir("lidar", Bl)
                tjl = json.load(open(os.path.join(ROOT, "profiles", pdl, "hbm_traffic.json")))
                rll = out["sweep"][-1]["roofline"]
                if tjl.get("library_src_hash") and ("src=" + tjl["library_src_hash"]) in lib_version and tjl["workload"].get("batch_per_gpu") == Bl:
                    rll["traffic"] = tjl["hbm_bytes_per_iteration"] * float(itl.sum())
                    rll["traffic_GBps"] = rll["traffic"] / (kl * 1e-3) / 1e9
                    rll["traffic_source"] = "profiles/%s@src=%s" % (pdl, tjl["library_src_hash"])
                else:
                    rll["traffic_source"] = "null: profiles/%s is of another build or batch (src=%s)" % (pdl, tjl.get("library_src_hash"))
            except (OSError, KeyError, ValueError):
                pass
            # CPU baseline of the LIDAR workload: the C oracle (oracle/lidar_oracle.c, OpenMP one instance per thread) on a bounded sample
            if args.cpu_sample != 0:
                from oracle import oracle_lib as OL, lidar_ref as LR
                lc = LR.lidar_v4()          # the checker's own definition of the same literals
                coresl = OL.max_threads()
                nl = min(Bl, 4 * coresl)
                OL.lidar_solve_batch(lc, Pl[:coresl], Wl[:coresl], lbx=lbx, ubx=ubx)
                t1 = time.perf_counter()
                refl = OL.lidar_solve_batch(lc, Pl[:nl], Wl[:nl], lbx=lbx, ubx=ubx)
                t_cpul = time.perf_counter() - t1
                dwl = np.max(np.abs(rl_["x"][:nl].cpu().numpy() - refl["x"]), axis=1)
                out["sweep"][-1]["cpu_baseline"] = {"value": nl / t_cpul, "unit": "solves/s", "cores": coresl, "kind": "port",
                                                    "sample": "first %d instances of the same batch, OpenMP one instance per thread, %.2f s; CPU restatement "
                                                              "(oracle/lidar_oracle.c), not CasADi/IPOPT" % (nl, t_cpul),
                                                    "mean_iters": float(refl["iters"].mean()), "same_point_frac_vs_gpu": float((dwl <= 1e-6).mean())}
            del ls
            torch.cuda.empty_cache()
        except Exception as e_:      # an extra must never cost the record its headline: note the failure and go on
            out.setdefault("extras_failed", []).append("sweep: %r" % (e_,))
    # (measured last: the OpenMP team of the oracle keeps the host cores spinning for a while after it returns)
    # CPU baseline: the C oracle on this box's host cores, bounded sample of the same workload
    if world == 1 and args.cpu_sample != 0:
        try:
            from oracle import oracle_lib as O
            from tests import helpers as Hh
            ocfg = Hh.to_oracle_cfg(cfg)
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
        except Exception as e_:      # an extra must never cost the record its headline: note the failure and go on
            out.setdefault("extras_failed", []).append("cpu: %r" % (e_,))
    # the extra CPU row of BASELINE.md 3 / SURVEY.md 8(d): nlpsol('ipopt') on the identical NLP built through the CasADi API by the build's own
    # generator (oracle/casadi_probe.py) — only if casadi happens to be importable on this box; the record says which
    if world == 1 and args.cpu_sample != 0:
        try:
            from oracle import casadi_probe as CP
            from tests import helpers as Hh
            ok_ca, what_ca = CP.available()
            if not ok_ca:
                out["casadi"] = "not importable on this box (%s): no CasADi/IPOPT row, parity stays unpinned" % what_ca
            else:
                n_ca = min(B, 8)
                ref_ca = CP.solve(Hh.to_oracle_cfg(cfg), P[:n_ca], W0[:n_ca])
                dca = np.max(np.abs(r["x"][:n_ca].cpu().numpy() - ref_ca["x"]), axis=1)
                out["casadi"] = {"version": what_ca, "value": 1.0 / float(ref_ca["seconds"].mean()), "unit": "solves/s", "cores": 1, "kind": "casadi-ipopt (own generator, C6:345 options)",
                                 "sample": "first %d instances, one thread" % n_ca, "same_point_frac_vs_gpu": float((dca <= 1e-6).mean()), "return_status": sorted(set(ref_ca["return_status"]))}
        except Exception as e_:      # an extra must never cost the record its headline: note the failure and go on
            out.setdefault("extras_failed", []).append("cpu: %r" % (e_,))
    # LAST key: a digest of everything above in a few hundred characters — a record that keeps only the tail of this line still holds every
    # sweep entry (solves/s, roofline fraction, traffic), the closed-loop and host-buffer rates, the CPU baseline and the casadi probe
    dg = {"six_B%d" % B: [round(value), round(out["roofline"]["frac"], 4), round(out["roofline"].get("traffic_GBps") or 0)]}
    for s_ in out.get("sweep", []):
        dg["%s_B%d" % (s_["workload"].split(":")[0], s_["batch"])] = [round(s_["value"]), round(s_["roofline"]["frac"], 4), round(s_["roofline"].get("traffic_GBps") or 0), int(s_["max_iters"]), round(s_.get("two_streams", {}).get("solves_per_s", 0))]
    for k_ in ("closed_loop", "host_buffers", "two_streams"):
        if k_ in out:
            dg[k_] = round(out[k_]["solves_per_s"])
    if "cpu_baseline" in out:
        dg["cpu_baseline"] = [round(out["cpu_baseline"]["value"]), out["cpu_baseline"]["cores"]]
    dg["casadi"] = "not importable" if isinstance(out.get("casadi"), str) else (out.get("casadi") or "not probed")
    dg["legend"] = "workload_batch: [solves/s, frac of 78.6 TFLOP/s fp64, HBM-side GB/s (0 = no matching profile), max iterations, solves/s with two launches in flight on two streams (0 = not measured)]"
    out["digest"] = dg
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
