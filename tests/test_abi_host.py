"""CPU: the C-ABI library loads and exports every symbol include/nmpc.h declares (no compute calls without a GPU);
the Python host mirrors the reference call surface; sharding helpers; gloo world_size-2 gather."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from oracle import nlp_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported(built):
    import nmpc_amd
    hdr = open(os.path.join(ROOT, "include", "nmpc.h")).read()
    lnames = set(re.findall(r"\b(nmpc_lidar_[a-z_0-9]+)\s*\(", open(os.path.join(ROOT, "include", "nmpc_lidar.h")).read()))
    Ll = nmpc_amd._lib.load()
    assert lnames == set(nmpc_amd._lib.LIDAR_EXPORTS) and all(hasattr(Ll, n) for n in lnames)
    names = set(re.findall(r"\b(nmpc_[a-z_0-9]+)\s*\(", hdr))
    assert {"nmpc_create", "nmpc_create_opts", "nmpc_query", "nmpc_destroy", "nmpc_solve_batch", "nmpc_eval_batch", "nmpc_shift_batch"} <= names
    # development aids are declared too (include/nmpc_debug.h), and nothing else is exported under an nmpc_ name
    dnames = set(re.findall(r"\b(nmpc_debug_[a-z_0-9]+)\s*\(", open(os.path.join(ROOT, "include", "nmpc_debug.h")).read()))
    assert dnames == set(nmpc_amd._lib.DEBUG_EXPORTS) and all(hasattr(Ll, n) for n in dnames)
    out = subprocess.check_output(["nm", "-D", "--defined-only", nmpc_amd._lib.SO_PATH], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("nmpc_")}
    assert exported == names | lnames | dnames, exported ^ (names | lnames | dnames)
    # "no global state": the library reads no environment variable (the Python host does, for development)
    pkg = os.path.dirname(os.path.dirname(nmpc_amd._lib.__file__))
    for src in ("nmpc_api.cpp", "nmpc_lidar.hip", "nmpc_solve_col.hip", "nmpc_solve_lds.hip", "nmpc_kernels.hip"):
        assert "getenv" not in open(os.path.join(os.path.dirname(nmpc_amd._lib.__file__), "csrc", src)).read(), src
    L = nmpc_amd._lib.load()
    for n in names:
        assert hasattr(L, n), n
    assert set(nmpc_amd._lib.EXPORTS) == names
    assert b"gfx950" in L.nmpc_version()
    # the library carries the hash of the sources it was built from; build() rebuilds on any mismatch (not on mtimes)
    import importlib
    bld = importlib.import_module("nmpc_amd.build")
    assert ("src=" + bld.source_hash()).encode() in L.nmpc_version() and bld.built_hash() == bld.source_hash() and not bld.needs_build()
    assert "libnmpc_hip.so" in nmpc_amd._lib.describe()
    # the library sits at the short in-tree path lib/libnmpc_hip.so, and the last build() call recorded what it did
    assert nmpc_amd._lib.SO_PATH == os.path.join(ROOT, "lib", "libnmpc_hip.so") or os.environ.get("NMPC_SO")
    info = bld.last_build_info()
    assert info.get("mode") in ("compiled", "reused") and info.get("src_hash") == bld.source_hash(), info


def test_forced_build_compiles_every_source(tmp_path):
    """NMPC_FORCE_BUILD=1: __graft_entry__.build() must run hipcc on every source (not reuse a binary) and say so — the check that the tree
    builds from scratch here, independent of whatever library travelled with the snapshot.  Runs in a child process and builds into a
    temporary directory (NMPC_LIBDIR / NMPC_OBJDIR / NMPC_SO): the in-tree lib/libnmpc_hip.so that this session and any
    other process have mapped is not replaced; the child loads the fresh library and checks every declared export."""
    env = dict(os.environ, NMPC_FORCE_BUILD="1", NMPC_LIBDIR=str(tmp_path), NMPC_OBJDIR=str(tmp_path / "obj"), NMPC_SO=str(tmp_path / "libnmpc_hip.so"))
    out = subprocess.check_output([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT, env=env, text=True, stderr=subprocess.STDOUT)
    assert "[nmpc build] compiled libnmpc_hip.so" in out and "build(): compiled" in out and str(tmp_path) in out, out
    import importlib, json
    bld = importlib.import_module("nmpc_amd.build")
    info = json.load(open(tmp_path / "build_info.json"))
    assert info["mode"] == "compiled" and info["src_hash"] == bld.source_hash() and info["seconds"] > 5.0 and info["hipcc"], info
    assert ("src=" + bld.source_hash()).encode() in open(tmp_path / "libnmpc_hip.so", "rb").read()


def test_sizes_defaults_and_config_mirror(built):
    import nmpc_amd
    L = nmpc_amd._lib.load()
    for cfg, oc in ((nmpc_amd.centralized_one_robot(20), R.cfg_one(20)), (nmpc_amd.centralized_two_robots(20), R.cfg_two(20)),
                    (nmpc_amd.centralized_six_robots(20), R.cfg_six(20)), (nmpc_amd.ten_robots_collision_avoidance(30), R.cfg_ten(30)),
                    (nmpc_amd.third_scenario_obstacles(20), R.cfg_obs3(20))):
        cc = cfg.to_c()
        assert L.nmpc_n_var(C.byref(cc)) == oc.n_var == cfg.n_var
        assert L.nmpc_n_g(C.byref(cc)) == oc.n_g == cfg.n_g
        assert L.nmpc_n_p(C.byref(cc)) == 2 * oc.nx
        for a, b in zip(cfg.bounds(), R.bounds(oc)):
            np.testing.assert_array_equal(a, b)
    # file-own horizons of the scripts
    assert nmpc_amd.centralized_one_robot().N == 100 and nmpc_amd.centralized_two_robots().N == 70
    assert nmpc_amd.centralized_six_robots().N == 35 and nmpc_amd.ten_robots_collision_avoidance().N == 20
    d = nmpc_amd._lib.CConfig()
    L.nmpc_config_default(C.byref(d), 6, 20)
    assert (d.m, d.N, d.pad_rows, d.max_iter) == (6, 20, 1, 2000) and list(d.q) == [1.0, 5.0, 0.1] and list(d.r) == [0.5, 0.05]
    assert d.tol == 1e-8 and d.xy_max == 10.0 and d.pad_value == 3.5 and np.isinf(d.th_max)


def test_create_fails_loudly_without_gpu(built):
    """no CPU fallback: on a box without a HIP device the handle cannot be created and the host class raises."""
    import torch
    import nmpc_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = nmpc_amd._lib.load()
    cc = nmpc_amd.centralized_two_robots(20).to_c()
    h = C.c_void_p()
    assert L.nmpc_create(C.byref(cc), 4, C.byref(h)) == -3          # NMPC_E_HIP
    bad = nmpc_amd.centralized_two_robots(20); bad.m = 11
    assert L.nmpc_create(C.byref(bad.to_c()), 4, C.byref(h)) in (-2, -3)
    with pytest.raises(RuntimeError):
        nmpc_amd.nlpsol("solver", "ipopt", nmpc_amd.centralized_two_robots(20), {})
    with pytest.raises(ValueError):
        nmpc_amd.nlpsol("solver", "sqpmethod", nmpc_amd.centralized_two_robots(20), {})


def test_product_never_imports_oracle():
    """the oracle is the checker only: nothing in the package (host or native) references it."""
    pkg = [d for d in os.listdir(ROOT) if d.endswith("_amd") and os.path.isdir(os.path.join(ROOT, d))][0]
    banned = ("import oracle", "from oracle", "oracle_lib", "libnmpc_oracle", "nmpc_oracle_", "nlp_ref", "ipm_proto")
    for dirpath, _, files in os.walk(os.path.join(ROOT, pkg)):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                for b in banned:
                    assert b not in txt, (f, b)
    assert "oracle" not in open(os.path.join(ROOT, "nmpc_amd.py")).read()


def test_oracle_libraries_export_no_product_symbol(built):
    """the checker's shared objects carry their own names only (nmpc_oracle_* / nmpc_lidar_oracle_*): even loaded RTLD_GLOBAL they could not
    stand in for an entry point of include/nmpc.h or include/nmpc_lidar.h."""
    import nmpc_amd
    from oracle import oracle_lib as O
    O.build(); O.lidar_lib()
    product = set(nmpc_amd._lib.EXPORTS) | set(nmpc_amd._lib.LIDAR_EXPORTS) | set(getattr(nmpc_amd._lib, "DEBUG_EXPORTS", []))
    for so in ("libnmpc_oracle.so", "liblidar_oracle.so"):
        out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "oracle", so)], text=True)
        names = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
        assert names and not (names & product), (so, names & product)
        assert all(n.startswith(("nmpc_oracle_", "nmpc_lidar_oracle_")) for n in names), names


def test_host_shift_and_cold_start(built):
    import nmpc_amd
    t0, u0 = nmpc_amd.shift(0.05, 1.0, np.array([[1, 2], [3, 4], [5, 6]]))
    assert t0 == 1.05 and u0.tolist() == [[3, 4], [5, 6], [5, 6]]
    X = np.arange(12.0).reshape(4, 3)
    np.testing.assert_array_equal(nmpc_amd.shift_states(X), np.vstack([X[1:], X[2:3]]))
    cfg = nmpc_amd.centralized_two_robots(20)
    np.testing.assert_array_equal(nmpc_amd.cold_start(cfg, R.C2_START), R.cold_start(R.cfg_two(20), R.C2_START))


def test_shard_range_partitions_contiguously(built):
    import nmpc_amd
    for B in (1, 7, 4096, 8192, 513):
        for world in (1, 2, 3, 8):
            r = [nmpc_amd.shard_range(B, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1


def test_strong_scaling_shards_partition_one_global_batch():
    """bench.py --scaling strong: every rank draws the SAME global batch (rank-independent stream) and keeps its contiguous shard — the shards
    of all ranks concatenate to the batch one rank draws alone (BASELINE configs 4 / 5: 4096 or 8192 instances over 8 GPUs)."""
    import importlib.util
    import nmpc_amd
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    total, world = 203, 8
    ocfg, B, P, W0 = bench.make_batch("two", 0, total)
    assert B == total and P.shape[0] == total
    parts = []
    for rank in range(world):
        lo, hi = nmpc_amd.shard_range(total, rank, world)
        oc, b, p, w = bench.make_batch("two", rank, total, shard=(lo, hi))
        assert b == hi - lo and p.shape == (b, 2 * ocfg.nx) and w.shape == (b, ocfg.n_var)
        parts.append((p, w))
    np.testing.assert_array_equal(np.concatenate([p for p, _ in parts]), P)
    np.testing.assert_array_equal(np.concatenate([w for _, w in parts]), W0)
    # weak scaling: every rank its own instances
    assert not np.array_equal(bench.make_batch("two", 1, 16)[2], bench.make_batch("two", 0, 16)[2])
    assert bench.STRONG_TOTAL["ten"] == 4096 and bench.STRONG_TOTAL["composite"] == 8192


def test_gather_results_gloo_world2(tmp_path):
    """the only collective of the multi-GPU path (result gather) rehearsed with gloo on CPU, world_size 2."""
    script = tmp_path / "w.py"
    script.write_text(
        "import os, sys, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "import nmpc_amd\n"
        "dist.init_process_group('gloo')\n"
        "r, w = dist.get_rank(), dist.get_world_size()\n"
        "B = 7\n"
        "lo, hi = nmpc_amd.shard_range(B, r, w)\n"
        "full = torch.arange(B * 3, dtype=torch.float64).reshape(B, 3)\n"
        "out = nmpc_amd.gather_results(full[lo:hi].clone(), B)\n"
        "st = nmpc_amd.gather_results(torch.arange(lo, hi, dtype=torch.int32), B)\n"
        "assert torch.equal(out, full), out\n"
        "assert st.tolist() == list(range(B))\n"
        "dist.destroy_process_group()\n" % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]


def test_independent_robot_mode_packing(built):
    """B swarms of m robots <-> B*m single-robot problems; without pair rows the centralized objective
    is the sum of the single-robot objectives and the packing round-trips."""
    import nmpc_amd
    m, N, B = 3, 5, 4
    rng = np.random.default_rng(3)
    p = rng.normal(size=(B, 6 * m))
    p1 = nmpc_amd.split_swarm(p, m)
    assert p1.shape == (B * m, 6)
    assert np.array_equal(p1[1], np.concatenate([p[0, 3:6], p[0, 3 * m + 3: 3 * m + 6]]))
    w1 = rng.normal(size=(B * m, 3 * (N + 1) + 2 * N))
    w = nmpc_amd.merge_swarm(w1, m, N)
    c1 = R.cfg_one(N); cm = R.cfg_one(N); cm.m = m
    for b in range(B):
        f_sum = sum(R.objective(c1, w1[b * m + i], p1[b * m + i]) for i in range(m))
        assert abs(R.objective(cm, w[b], p[b]) - f_sum) <= 1e-12 * max(1.0, abs(f_sum))


def test_bench_roofline_inputs_match_survey():
    """algorithmic bytes / flops per solve that bench.py prices the roofline with."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    want = {"one": (R.cfg_one(20), 1712, 3.23e3), "two": (R.cfg_two(20), 3408, 25.9e3), "six": (R.cfg_six(20), 10192, 698.4e3),
            "ten30": (R.cfg_ten(30), 24976, 4.85e6), "ten20": (R.cfg_ten(20), 16976, 3.23e6)}
    for name, (cfg, nbytes, fkkt) in want.items():
        assert bench.algorithmic_bytes_per_solve(cfg) == nbytes, name
        f_asm = cfg.N * (22.0 * cfg.m + 14.0 * cfg.M + 16.0 * cfg.m * cfg.K)
        assert abs(bench.algorithmic_flops_per_iter(cfg) - f_asm - fkkt) <= 0.005 * fkkt, (name, bench.algorithmic_flops_per_iter(cfg) - f_asm)
    # every workload of the bench resolves to a configuration and a batch size
    for wname, (m, N, B) in {"two": (2, 20, 4096), "six": (6, 20, 4096), "ten20": (10, 20, 4096), "ten": (10, 30, 512), "composite": (6, 25, 1024)}.items():
        cfg, b, _ = bench.workload(wname)
        assert (cfg.m, cfg.N, b) == (m, N, B)


def test_bench_self_launches_ranks_and_never_reports_one_gpu_for_n(built):
    """ADVICE r1 / VERDICT r1: `python bench.py --gpus 2` without a launcher must start 2 ranks itself (before any GPU call) and must
    never print an n_gpus=1 line for it.  Without GPUs the ranks exit non-zero ("needs a GPU") and so does the parent; with
    NMPC_BENCH_REHEARSAL the multi-rank path (gloo, shared device) is exercised on a GPU box by tools/, not here."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the driver's scaling run")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--cpu-sample", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "needs a GPU" in (r.stderr + r.stdout)            # raised inside the child ranks
    assert '"n_gpus": 1' not in r.stdout and '"n_gpus"' not in r.stdout
    # a launcher environment that disagrees with --gpus is an error, not a silent one-rank run
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env2, timeout=120)
    assert r2.returncode != 0 and "WORLD_SIZE=1" in (r2.stderr + r2.stdout)


# (m, N, T, dmin, v_max, w_max, xy_max, K obstacles, pad rows) typed here from the scripts, independently of problem.py's table
_SCRIPT_LITERALS = {
    "first_scenario": (1, 100, 0.05, 0.0, 0.22, 2.84, 10.0, 0, False), "second_scenario": (2, 50, 0.1, 0.25, 0.22, 2.84, 10.0, 0, True),
    "third_scenario": (3, 50, 0.05, 0.3, 0.22, 2.84, 10.0, 0, True), "fourth_scenario": (4, 50, 0.1, 0.3, 0.22, 2.84, 10.0, 0, True),
    "fifth_scenario": (5, 35, 0.1, 0.3, 0.22, 2.84, 10.0, 0, True), "sixth_scenario": (6, 35, 0.3, 0.3, 0.22, 2.84, 10.0, 0, True),
    "decentralized_first_scenario": (1, 200, 0.05, 0.0, 0.22, 2.84, 2.0, 0, False),
    "decentralized_two_robots": (2, 50, 0.1, 0.25, 0.22, 2.84, 10.0, 0, True),
    "centralized_one_robot": (1, 100, 0.05, 0.0, 0.22, 2.84, 10.0, 0, False), "centralized_two_robots": (2, 70, 0.05, 0.15, 0.22, 2.84, 10.0, 0, True),
    "centralized_three_robots": (3, 60, 0.05, 0.15, 0.22, 2.84, 10.0, 0, True), "centralized_four_robots": (4, 45, 0.1, 0.4, 0.22, 2.84, 10.0, 0, True),
    "centralized_five_robots": (5, 40, 0.1, 0.4, 0.22, 2.84, 10.0, 0, True), "centralized_six_robots": (6, 35, 0.3, 0.4, 0.15, 1.5, 10.0, 0, True),
    "tb3_two_collision_free": (2, 100, 0.02, 0.25, 0.22, 2.84, 10.0, 0, True), "tb3_five_collision_free": (5, 70, 0.02, 0.3, 0.22, 2.84, 10.0, 0, True),
    "tb3_six_collision_free": (6, 35, 0.2, 0.3, 0.22, 2.84, 10.0, 0, True), "tb3_eight_collision_free": (8, 5, 0.02, 0.25, 0.22, 2.84, 10.0, 0, True),
    "ten_robots_collision_avoidance": (10, 20, 0.1, 0.3, 0.22, 2.84, 10.0, 0, True),
    "two_robots_no_collision_rows": (2, 50, 0.01, 0.0, 0.22, 2.84, 10.0, 0, False),
    "mpc_online_casadi": (1, 50, 0.01, 0.0, 0.22, 2.84, 10.0, 0, False), "mpc_online_casadi_tb3_1": (1, 200, 0.01, 0.0, 0.22, 2.84, 10.0, 0, False),
    "casadi_test": (1, 25, 0.25, 0.0, 0.22, 2.84, 10.0, 0, False), "casadi_test_mpc": (1, 50, 0.02, 0.0, 0.22, 2.84, 10.0, 0, False),
    "first_scenario_obstacle": (1, 100, 0.1, 0.0, 0.2, np.pi / 4, 10.0, 1, False),
    "second_scenario_obstacles": (1, 100, 0.1, 0.0, 0.2, np.pi / 4, 10.0, 4, False),
    "third_scenario_obstacles": (1, 100, 0.2, 0.0, 0.2, 1.0, 10.0, 6, False),
}


def test_script_presets_hold_the_scripts_literals(built):
    """every reference script that builds the NLP of SURVEY 8 is a parameter set of the one generic solver (SURVEY 2:
    'scenario variants as extra parameter sets'); sizes follow the scripts' own `args` shapes."""
    import nmpc_amd
    from tests import helpers as Hh
    L = nmpc_amd._lib.load()
    assert set(nmpc_amd.script_names()) == set(_SCRIPT_LITERALS)
    for name, (m, N, T, dmin, v, w, xy, K, pad) in _SCRIPT_LITERALS.items():
        c = nmpc_amd.script_preset(name)
        assert (c.m, c.N, c.T, c.dmin, c.v_max, c.xy_max, len(c.obstacles), c.pad_rows) == (m, N, T, dmin, v, xy, K, pad), name
        assert abs(c.w_max - w) < 1e-15 and c.q == (1.0, 5.0, 0.1) and c.r == (0.5, 0.05), name
        M = m * (m - 1) // 2 if c.pair_rows else 0
        # the scripts' lbg: (N+1) blocks of (3m + M) rows (second_scenario.py:181), or 3 + N*(3+K) for the obstacle files
        assert c.n_g == ((3 * m + M) * (N + 1) if K == 0 else 3 + N * (3 + K)), name
        assert c.n_var == 3 * m * (N + 1) + 2 * m * N
        cc = c.to_c()
        assert L.nmpc_n_var(C.byref(cc)) == c.n_var and L.nmpc_n_g(C.byref(cc)) == c.n_g
        oc = Hh.to_oracle_cfg(c)
        for a, b in zip(c.bounds(), R.bounds(oc)):
            np.testing.assert_array_equal(a, b)
        assert nmpc_amd.script_preset(name, N=7).N == 7
    ob = nmpc_amd.script_preset("first_scenario_obstacle")
    assert (ob.rob_dim, ob.margin, ob.th_max) == (0.15, 0.05, 2 * np.pi) and ob.obstacles == [(0.4, 1.1, 0.15)]
    assert nmpc_amd.script_preset("second_scenario_obstacles").obstacles[1] == (-0.75, 0.0, 0.125)
    with pytest.raises(KeyError):
        nmpc_amd.script_preset("no_such_script")


def test_row_paired_layout_emulation_matches_the_one_row_layout():
    """tools/rp_emu.py: the row-paired backward sweep of csrc/nmpc_solve_col.hip written out lane by lane in numpy (64 lanes, two rows per
    register, row_newbcast / permlane16_swap / permlane32_swap / bpermute / readlane as index arithmetic) against tools/lane_emu.py's emulation
    of the one-row-per-register sweep on random stage packs, two to six robots (odd teams leave the upper half's last robot slot empty): the
    pivot rows the forward sweep reads and the cost-to-go handed to the next stage agree to rounding."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import rp_emu
    from lane_emu import lane_sweep
    rng = np.random.default_rng(3)
    for m in (2, 3, 4, 5, 6):
        N = 3
        pk, off = rp_emu.random_packs(m, N, rng)
        ref = lane_sweep(pk, m, N, 3 * m, 2 * m, off)
        got, Pk, pv = rp_emu.rp_sweep(pk, m, N, off)
        err = max(np.abs(ref[k] - got[k]).max() / max(1.0, np.abs(ref[k]).max()) for k in range(N))
        assert err < 1e-13 and np.abs(Pk - Pk.T).max() < 1e-12, (m, err)


_ASM = {}      # team size -> assembly file of this session (one hipcc run per team size, the three of the hazard test in parallel)


def _col_kernel_asm(tmp_path, m, also=()):
    """gfx950 assembly of the column kernel instantiated for one team size (hipcc cross-compiles here); `also`: team sizes compiled
    alongside, in parallel, for later calls"""
    import importlib
    import shutil
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    bld = importlib.import_module("nmpc_amd.build")
    hipcc = next((c for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")) if c and os.path.exists(c)), None)
    if hipcc is None:
        pytest.skip("no hipcc")
    src = os.path.join(bld.CSRC, "nmpc_solve_col.hip")

    def cc(mm):
        d = tempfile.mkdtemp(prefix="nmpc_asm_m%d_" % mm)
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", '-DNMPC_SRC_HASH="t"', "-DNMPC_COL_ONLY_M=%d" % mm]
                              + bld.FILE_FLAGS.get("nmpc_solve_col.hip", [])          # the code generation switches of the shipped build
                              + ["-I" + os.path.join(ROOT, "include"), "-c", src, "-o", os.path.join(d, "col.o"), "--save-temps=obj"], cwd=bld.CSRC)
        return mm, os.path.join(d, "nmpc_solve_col-hip-amdgcn-amd-amdhsa-gfx950.s")
    todo = [mm for mm in dict.fromkeys((m,) + tuple(also)) if mm not in _ASM]
    if todo:
        with ThreadPoolExecutor(max_workers=len(todo)) as ex:
            _ASM.update(dict(ex.map(cc, todo)))
    return _ASM[m]


def test_hot_loops_of_the_column_kernel_hold_no_spills(built, tmp_path):
    """Build check (hipcc cross-compiles here): the two hot loops of the throughput kernel — the backward Riccati stage and the
    forward sweep, the loops that contain the DPP multiply-adds — hold no scratch (spill) instruction.  A spill reload inside
    a loop that prefetches waits (vmcnt is in order) for every prefetch in flight; the property is fragile under edits anywhere
    in the kernel, so it is pinned here for the headline team size."""
    asm = _col_kernel_asm(tmp_path, 6, also=(2, 10))      # the hazard test below reads all three
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "asm_loops.py"), asm, "_ZN4nmpc16solve_col_kernelILi6ELi0", "--all"], text=True)
    loops = []
    for line in out.splitlines():
        m = re.search(r"loop\s+(\d+)-\s*(\d+):\s+(\d+) instr, dpp\s+(\d+), .*scratch ld\s+(\d+) st\s+(\d+), vmcnt\(0\)\s+(\d+)", line)
        if m:
            loops.append(tuple(int(g) for g in m.groups()))
    # innermost loop that holds all 147 elimination multiply-adds of a row-paired stage (two rows per register: 282 single-row ones
    # before), and the forward loop (the 29 of one stage, unrolled twice or not)
    back = min((l for l in loops if l[3] == 147), key=lambda l: l[2])
    fwd = min((l for l in loops if l[3] in (29, 58)), key=lambda l: l[2])
    print("backward stage loop", back, "forward loop", fwd)
    assert back[4] == 0 and back[5] == 0, out
    assert fwd[4] == 0 and fwd[5] == 0, out
    assert back[2] < 1100, back          # ~1000 instructions per stage (1190 with one row per register)
    # end of round 3: no scratch at all, and still within the 256 registers that let two waves share a SIMD — for every instantiation
    # of the headline team size (throughput shape, both latency shapes; DESIGN.md 4.1: occupancy request, no machine LICM)
    kernels = _kernel_resources(asm, "solve_col_kernelILi6")
    assert len(kernels) >= 6, kernels
    for name, res in kernels.items():
        assert res["private_segment_fixed_size"] == 0 and res["vgpr_spill_count"] == 0 and res["vgpr_count"] <= 256, (name, res)


def _kernel_resources(asm_path, name_part):
    """{kernel symbol: {.private_segment_fixed_size, .vgpr_count, .vgpr_spill_count, ...}} from the metadata of a compiled .s file"""
    out, cur = {}, None
    for line in open(asm_path):
        m = re.match(r"\s*-?\s*\.name:\s+(\S+)", line)
        if m:
            cur = m.group(1) if name_part in m.group(1) and not m.group(1).endswith(".kd") else None
            if cur:
                out[cur] = {}
            continue
        m = re.match(r"\s*\.(private_segment_fixed_size|vgpr_count|vgpr_spill_count|sgpr_spill_count|agpr_count):\s+(\d+)", line)
        if m and cur:
            out[cur][m.group(1)] = int(m.group(2))
    return {k: v for k, v in out.items() if "vgpr_count" in v}


def test_throughput_kernels_keep_two_waves_per_simd(built, tmp_path):
    """Build check on lib/libnmpc_hip.so itself (ELF notes of its gfx950 code objects): every throughput-shape instantiation of the column
    kernel for up to six robots — what a batch of thousands runs on — fits two wavefronts per SIMD, i.e. vgpr_count <= 256 of the unified
    512-entry file.  The four-robot kernels had slipped to 257..267 registers (one wave per SIMD: -10 % at B = 4096, -30 % at 16384)
    when the elastic phase went in; nothing else would have noticed."""
    import importlib
    bld = importlib.import_module("nmpc_amd.build")
    tools = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists(os.path.join(tools, "llvm-objcopy")) and os.path.exists(os.path.join(tools, "llvm-readelf"))):
        pytest.skip("no llvm-objcopy / llvm-readelf")
    fat = str(tmp_path / "fat.bin")
    subprocess.check_call([os.path.join(tools, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, bld.SO, str(tmp_path / "copy.so")])
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(b"\x7fELF", data)]
    regs = {}
    for n, a in enumerate(starts):
        p = str(tmp_path / ("co%d.elf" % n))
        with open(p, "wb") as f:
            f.write(data[a:starts[n + 1] if n + 1 < len(starts) else len(data)])
        notes = subprocess.run([os.path.join(tools, "llvm-readelf"), "--notes", p], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:      # one block per kernel: its keys are sorted, .agpr_count comes first
            name = re.search(r"\n\s+\.name:\s+(\S+)", blk)
            if not name:
                continue
            d = {"agpr_count": int(blk.split()[0])}
            for key in ("vgpr_count", "private_segment_fixed_size"):
                m = re.search(r"\n\s+\.%s:\s+(\d+)" % key, blk)
                if m:
                    d[key] = int(m.group(1))
            regs[name.group(1)] = d
    col = {k: v for k, v in regs.items() if "solve_col_kernel" in k}
    assert len(col) >= 50, len(col)
    seen = 0
    for k, v in col.items():
        m = re.search(r"solve_col_kernelILi(\d+)ELi(\d)ELi(\d)ELi(\d+)E", k)
        team, tpb = int(m.group(1)), int(m.group(4))
        if team <= 6 and tpb == 64:
            seen += 1
            assert v["vgpr_count"] <= 256, (k, v)      # .vgpr_count is the unified allocation (architectural + accumulation registers)
        if team <= 6:
            assert v.get("private_segment_fixed_size", 0) <= 32, (k, v)      # four robots, duals in LDS: 4 spilled dwords outside the hot loops
    assert seen >= 20, seen
    lid = {k: v for k, v in regs.items() if "lidar_solve_kernel" in k}
    assert len(lid) == 3 and all(v.get("private_segment_fixed_size", 0) == 0 for k, v in lid.items() if "ILi10ELi2E" not in k), lid
    assert all(v["vgpr_count"] <= 256 for k, v in lid.items() if "ILi10ELi2E" in k), lid


def test_lidar_kernel_holds_no_spills(built, tmp_path):
    """Build check: the LIDAR solve kernel (both instantiations) compiles without scratch.  Round 2's kernel reloaded spilled literals of
    log() inside its hot loops — a full vmcnt wait per reload, ~1,500 cycles per logarithm; the property depends on the
    per-source code generation switches of build.py and on the occupancy request, so it is pinned here."""
    import importlib
    import shutil
    bld = importlib.import_module("nmpc_amd.build")
    hipcc = next((c for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")) if c and os.path.exists(c)), None)
    if hipcc is None:
        pytest.skip("no hipcc")
    out = tmp_path / "lidar.s"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-S"] + bld.FILE_FLAGS.get("nmpc_lidar.hip", [])
                          + [os.path.join(bld.CSRC, "nmpc_lidar.hip"), "-o", str(out)], cwd=bld.CSRC)
    kernels = _kernel_resources(str(out), "lidar_solve_kernel")
    assert len(kernels) == 3, kernels          # ray count of the scripts (10) at one and at two waves per SIMD, and the run-time count
    for name, res in kernels.items():
        if "ILi10ELi2E" in name:               # the 256-register instantiation for batches beyond one instance per SIMD: spills outside the recursions, by choice
            assert res["vgpr_count"] <= 256 and res["private_segment_fixed_size"] <= 320, (name, res)
        else:
            assert res["private_segment_fixed_size"] == 0 and res["vgpr_spill_count"] == 0, (name, res)
    text = open(out).read()
    assert text.count("v_fmac_f64_dpp") >= 40          # the column-per-lane Riccati stage is what got compiled (NMPC_LIDAR_DPP=1)


def test_dpp_reads_of_the_column_kernel_keep_their_wait_states(built, tmp_path):
    """Build check: every v_fmac_f64_dpp of the column kernel (inline asm: the compiler's hazard recognizer does not see it) has 2 wait
    states between a VALU / permlane-swap write of its DPP source register and the read — the kernel's own s_nop markers provide them,
    provided the register allocator inserts no copy in between.  Checked on the compiled code of two, six and ten robots (row-paired
    and one-row-per-register sweeps, one / two / four rows of 16 lanes) with tools/asm_hazards.py."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import asm_hazards
    for m, least in ((2, 20), (6, 150), (10, 600)):
        n, hz = asm_hazards.check(_col_kernel_asm(tmp_path, m, also=(2, 6, 10)), "_ZN4nmpc16solve_col_kernel")
        print("m=%d: %d DPP multiply-adds, %d hazards" % (m, n, len(hz)))
        assert n >= least and not hz, (m, n, hz[:5])


def test_bench_workload_is_built_from_the_product_side():
    """bench.py builds its workload from nmpc_amd presets and its own generator; oracle/ and tests/ are imported only in
    the cpu_baseline / casadi legs (inside `if ... args.cpu_sample != 0` blocks).  The literal start/goal sets it carries equal the oracle's."""
    import importlib.util, re
    path = os.path.join(ROOT, "bench.py")
    spec = importlib.util.spec_from_file_location("bench", path)
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    assert np.array_equal(np.concatenate(bench.C6_LITERAL), np.concatenate([R.C6_START, R.C6_GOAL]))
    assert np.array_equal(np.concatenate(bench.C2_LITERAL), np.concatenate([R.C2_START, R.C2_GOAL]))
    src = open(path).read().split("\n")
    for i, line in enumerate(src):
        if re.search(r"^\s*(from|import) (oracle|tests)\b", line):
            # the nearest enclosing `if` at a smaller indent must be a cpu_sample guard
            ind = len(line) - len(line.lstrip())
            j = i - 1
            while j >= 0 and not (src[j].strip().startswith("if ") and len(src[j]) - len(src[j].lstrip()) < ind):
                j -= 1
            assert j >= 0 and "cpu_sample != 0" in src[j], (i + 1, line, src[j] if j >= 0 else None)
    # the product config of every workload equals the oracle's own definition of the same script literals
    from tests import helpers as Hh
    for name, oc in (("two", R.cfg_two(20)), ("six", R.cfg_six(20)), ("ten20", R.cfg_ten(20)), ("ten", R.cfg_ten(30))):
        assert Hh.to_oracle_cfg(bench.workload(name)[0]) == oc, name


def test_committed_bench_record_and_its_table():
    """profiles/current/bench.json is a complete line of bench.py (the contract's keys, roofline and cpu_baseline blocks, the digest as the last
    key) taken from the library whose profiles sit next to it, and tools/design_table.py — the generator of DESIGN.md 7 — still reads it."""
    import json
    rec = os.path.join(ROOT, "profiles", "current", "bench.json")
    d = json.load(open(rec))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert list(d)[-1] == "digest" and d["unit"] == "solves/s" and d["dtype"] == "f64" and d["vs_baseline"] is None and d["n_gpus"] == 1
    assert d["config"]["workload"].startswith("centralized_six_robots") and d["config"]["batch_per_gpu"] == 4096
    rl = d["roofline"]
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-12 and rl["traffic"] and rl["traffic_source"].startswith("profiles/current@src=")
    assert rl["traffic_source"].split("src=")[1] in d["library"]          # the traffic figure comes from THIS build's profile
    tj = json.load(open(os.path.join(ROOT, "profiles", "current", "hbm_traffic.json")))
    assert tj["library_src_hash"] in d["library"]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and "extras_failed" not in d
    assert d["two_streams"]["same_iterations_as_value_run"] is True and d["two_streams"]["solves_per_s"] > d["value"]
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "design_table.py"), rec], text=True)
    rows = [l for l in out.splitlines() if l.startswith("| ")]
    assert len(rows) == 1 + 9 and "**headline**" in rows[1] and "LIDAR" in rows[-1], out[:400]      # header + nine shapes (the separator line starts with "|-")
