"""CPU: the C interior-point oracle pinned by independent evidence — scipy-SLSQP golden triples
(tests/golden/, generator committed), the solver-independent KKT report and the numpy prototype."""
import os

import numpy as np
import pytest

from oracle import nlp_ref as R, oracle_lib as O, ipm_proto as I
from tests import helpers as Hh

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = {"one": R.cfg_one(20), "two": R.cfg_two(20), "obs3": R.cfg_obs3(20),
         "three": R.NLPConfig(m=3, N=10, T=0.3, dmin=0.4, v_max=0.15, w_max=1.5), "mix3": Hh.cfg_mix3(10),
         "six": R.cfg_six(20), "ten": R.cfg_ten(20)}      # headline configuration incl. the literal C6:364-388 swap; two ten-robot instances


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_slsqp_golden(name):
    """'scipy-SLSQP oracle, not CasADi/IPOPT'.  SLSQP's own accuracy is ~1e-5 in w (its polish run moves by up to
    5e-5), the barrier offset of the interior point is ~ n_active * mu = 1e-7 in f."""
    cfg = CASES[name]
    if not os.path.exists(os.path.join(GOLD, "slsqp_%s.npz" % name)):
        pytest.skip("tests/golden/slsqp_%s.npz not generated (gen_golden.py %s)" % (name, name))
    z = np.load(os.path.join(GOLD, "slsqp_%s.npz" % name))
    r = O.solve_batch(O.make_config(cfg, max_iter=500), z["p"], z["w0"])
    assert (r["status"] == 0).all() and (r["kkt"] <= 1e-8).all()
    df = np.abs(r["f"] - z["f_pol"]) / np.maximum(1.0, np.abs(z["f_pol"]))
    dw = np.max(np.abs(r["x"] - z["w_pol"]), axis=1)
    same = df < 1e-6
    big = name in ("six", "ten")      # 618 / 1030 variables: SLSQP stops at a stationarity of 2-4e-5 (w within ~3e-4), and on 2 of the 5 six-robot
    # instances (the literal antipodal swap among them) its cold start ends in a basin with a HIGHER objective than ours
    # non-convex: a cold-start SLSQP may pick another basin.  Measured agreement: one 6/6, two 5/6, obs3 4/4, three 3/3, mix3 3/4, six 3/5, ten 2/2
    assert same.sum() >= {"one": 6, "two": 5, "obs3": 4, "three": 3, "mix3": 3, "six": 3, "ten": 2}[name], (df, dw)
    assert (dw[same] < (5e-4 if big else 2e-4)).all(), dw
    # where the basin differs both must be KKT points, and ours is re-checked independently
    for b in np.where(~same)[0]:
        k = R.kkt_report(cfg, r["x"][b], z["p"][b], tol_active=1e-3)
        assert k["stat"] < 1e-5 and k["eq"] < 1e-8 and k["ineq"] < 1e-8, k


def test_oracle_solutions_are_kkt_points_six_robots():
    from tests import helpers as Hh
    cfg = R.cfg_six(20)
    P, W0 = Hh.batch(cfg, 12, 2)
    r = O.solve_batch(O.make_config(cfg, max_iter=500), P, W0)
    assert (r["status"] == 0).all()
    lbx, ubx, lbg, ubg = R.bounds(cfg)
    for b in range(12):
        k = R.kkt_report(cfg, r["x"][b], P[b], tol_active=1e-3)
        assert k["stat"] < 1e-5 and k["eq"] < 1e-8 and k["ineq"] < 1e-8 and k["bnd"] == 0.0, (b, k)
        assert r["f"][b] == pytest.approx(R.objective(cfg, r["x"][b], P[b]), rel=1e-12)
        assert np.array_equal(r["x"][b][: cfg.nx], P[b][: cfg.nx])      # X_0 pinned to x0


def test_literal_scenarios_converge():
    """the scripts' own start/goal sets (C2:213-224, C6:364-388); the antipodal six-robot swap is the hard one."""
    for cfg, s, g in ((R.cfg_two(20), R.C2_START, R.C2_GOAL), (R.cfg_six(20), R.C6_START, R.C6_GOAL)):
        p = np.concatenate([s, g])[None]
        r = O.solve_batch(O.make_config(cfg, max_iter=1000), p, R.cold_start(cfg, s)[None])
        assert r["status"][0] == 0, (r["iters"], r["kkt"])
        k = R.kkt_report(cfg, r["x"][0], p[0], tol_active=1e-3)
        assert k["stat"] < 1e-5 and k["eq"] < 1e-8 and k["ineq"] < 1e-8


def test_fixed_point_and_infeasible_start():
    cfg = R.cfg_two(20)
    p = np.concatenate([R.C2_GOAL, R.C2_GOAL])[None]
    r = O.solve_batch(O.make_config(cfg), p, R.cold_start(cfg, R.C2_GOAL)[None])
    assert r["status"][0] == 0 and abs(r["f"][0]) < 1e-7 and np.max(np.abs(r["x"][0][cfg.nx * 21:])) < 1e-5   # item 3: U = 0
    # stage-0 pair row violated by the pinned x0 (the ten-robot literal x0 has coincident robots, C10:389)
    bad = np.array([[0.0, 0.0, 0.0, 0.05, 0.0, 0.0, 1.0, 1.0, 0.0, -1.0, -1.0, 0.0]])
    w0 = R.cold_start(cfg, bad[0, :6])[None]
    r = O.solve_batch(O.make_config(cfg), bad, w0)
    assert r["status"][0] == 3 and r["iters"][0] == 0 and np.array_equal(r["x"][0], w0[0])


def test_prototype_and_c_oracle_agree():
    """two independent implementations (numpy dense blocks, C scalar loops) of the same algorithm."""
    from tests import helpers as Hh
    for cfg, idx in ((R.cfg_one(20), 0), (R.cfg_two(20), 1)):
        P, W0 = Hh.batch(cfg, 4, idx)
        r = O.solve_batch(O.make_config(cfg, max_iter=300), P, W0)
        for b in range(4):
            o = I.Opts(); o.max_iter = 300
            rp = I.solve(cfg, P[b], W0[b], o)
            assert rp["status"] == 0 and abs(rp["f"] - r["f"][b]) < 1e-6 * max(1, abs(rp["f"]))
            assert np.max(np.abs(rp["x"] - r["x"][b])) < 1e-5


def test_warm_start_closed_loop_runs():
    """three closed-loop steps of AS/casadi_test.py:143-183 with the oracle: solve, plant step, shift."""
    cfg = R.cfg_two(20)
    oc = O.make_config(cfg, max_iter=500)
    p = np.concatenate([R.C2_START, R.C2_GOAL])[None]; w = R.cold_start(cfg, R.C2_START)[None]
    d0 = np.linalg.norm(p[0, :6] - p[0, 6:])
    for _ in range(3):
        r = O.solve_batch(oc, p, w)
        assert r["status"][0] == 0
        w, x0n = O.shift_batch(oc, p, r["x"])
        p = p.copy(); p[:, :6] = x0n
    assert np.linalg.norm(p[0, :6] - p[0, 6:]) < d0


def test_closed_loop_episode_stays_collision_free_and_converged():
    """12 receding-horizon steps on 48 six-robot swarms (a9 + a13 + a11): every solve converges (the dual step is capped
    by the accepted primal step, without which ~0.03% of warm-started solves diverged), pair distances stay >= dmin up to
    the Euler-plant mismatch, and the swarm moves towards its goals."""
    cfg = R.cfg_six(20)
    oc = O.make_config(cfg, max_iter=600)
    P, W = Hh.batch(cfg, 48, 2)
    P = P.copy()
    d0 = np.linalg.norm((P[:, : cfg.nx] - P[:, cfg.nx:]).reshape(48, 6, 3)[:, :, :2], axis=2).sum(axis=1)
    for _ in range(12):
        r = O.solve_batch(oc, P, W)
        assert (r["status"] == 0).all(), (r["status"], r["iters"])
        W, x0n = O.shift_batch(oc, P, r["x"])
        P[:, : cfg.nx] = x0n
        xy = x0n.reshape(48, 6, 3)[:, :, :2]
        for i in range(6):
            for j in range(i + 1, 6):
                assert (np.linalg.norm(xy[:, i] - xy[:, j], axis=1) >= cfg.dmin - 1e-6).all()
    d1 = np.linalg.norm((P[:, : cfg.nx] - P[:, cfg.nx:]).reshape(48, 6, 3)[:, :, :2], axis=2).sum(axis=1)
    assert (d1 < d0).all()


def test_stall_case_and_composite_failures_are_rescued_by_the_cold_start_retry():
    """Fixtures of tests/golden/gen_cold_retry_cases.py (regenerated in round 4: the captured failures are failures of the algorithm as
    shipped, and the partial re-factorisation of the backward sweep changed which solves fail).  All are warm-started closed-loop solves
    of the six-robot + eight-obstacle composite.  stall_case.npz: a solve that converges to an infeasible stationary point (IPOPT would
    enter restoration): three barrier restarts do not help; the restoration of last resort — a restart from the reference's own cold
    start X_k = x0, U = 0 (C6:398-400), at most twice, the second time with mu = 10 mu_init — converges.  cold_retry_cases.npz: the 9 of
    10,240 closed-loop solves that fail without the retry (stall, cycling until max_iter, one numerical failure;
    NMPC_ORACLE_NO_COLD_RETRY=1): all converge, the cyclers through the iteration watchdog (500).  cold_retry_cases2.npz: the 3 root
    failures of 30,720 solves the FIRST retry does not rescue (NMPC_ORACLE_MAX_COLD=1); the second one does."""
    rng = np.random.default_rng(7)
    c = R.cfg_six(25); c.rob_dim = 0.2; c.margin = 0.1
    c.obstacles = [(float(x), float(y), float(r_)) for x, y, r_ in zip(rng.uniform(-1.5, 1.5, 8), rng.uniform(-1.5, 1.5, 8), rng.uniform(0.125, 0.2, 8))]
    gold = os.path.join(os.path.dirname(__file__), "golden")
    d = np.load(os.path.join(gold, "stall_case.npz"))
    r = O.solve_batch(O.make_config(c, max_iter=2000), d["p"][None], d["w"][None])
    assert r["status"][0] == 0 and 300 < r["iters"][0] < 1000 and r["kkt"][0] <= 1e-8, (r["status"], r["iters"])      # measured: 485
    k = R.kkt_report(c, r["x"][0], d["p"], tol_active=1e-3)
    assert k["eq"] < 1e-8 and k["ineq"] < 1e-8, k
    z = np.load(os.path.join(gold, "cold_retry_cases.npz"))
    rr = O.solve_batch(O.make_config(c, max_iter=2000), z["p"], z["w"])
    assert len(z["p"]) == 9 and (rr["status"] == 0).all() and (rr["kkt"] <= 1e-8).all(), (rr["status"], rr["iters"])
    z2 = np.load(os.path.join(gold, "cold_retry_cases2.npz"))
    r2 = O.solve_batch(O.make_config(c, max_iter=2000), z2["p"], z2["w"])
    assert len(z2["p"]) == 3 and (r2["status"] == 0).all() and (r2["kkt"] <= 1e-8).all() and (r2["iters"] > 2 * 500).all(), (r2["status"], r2["iters"])
    # chaotic_composite_case.npz: the one solve of a 512 x 120 composite soak the HIP kernels of round 3's final build did not converge
    # (a chaotic solve: rounding differences grow ~30x per six iterations); kept as an input after the algorithm change of round 4
    zc = np.load(os.path.join(gold, "chaotic_composite_case.npz"))
    rc_ = O.solve_batch(O.make_config(c, max_iter=2000), zc["p"][None], zc["w"][None])
    assert rc_["status"][0] == 0 and rc_["kkt"][0] <= 1e-8, (rc_["status"], rc_["iters"])
    # without the retry the stall case fails (what the fixture was captured for), and with one retry only the cases2 do
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, sys; sys.path.insert(0, %r); from tests import helpers as Hh; from oracle import nlp_ref as R, oracle_lib as O; "
            "d = np.load(%r); p = d['p'] if d['p'].ndim == 2 else d['p'][None]; w = d['w'] if d['w'].ndim == 2 else d['w'][None]; "
            "r = O.solve_batch(O.make_config(Hh.bench_batch('composite', 1)[0], max_iter=2000), p, w); print('STATUS', r['status'].tolist())")
    out = subprocess.run([sys.executable, "-c", code % (root, os.path.join(gold, "stall_case.npz"))], capture_output=True, text=True,
                         env=dict(os.environ, NMPC_ORACLE_NO_COLD_RETRY="1"), timeout=300)
    assert "STATUS [4]" in out.stdout, (out.stdout, out.stderr[-500:])
    out = subprocess.run([sys.executable, "-c", code % (root, os.path.join(gold, "cold_retry_cases2.npz"))], capture_output=True, text=True,
                         env=dict(os.environ, NMPC_ORACLE_MAX_COLD="1"), timeout=600)
    assert "STATUS [1, 4, 1]" in out.stdout, (out.stdout, out.stderr[-500:])


def test_ipopt_defaults_variant_and_basin_sensitivity(tmp_path):
    """the oracle's IPOPT-faithful variant (NMPC_ORACLE_IPOPT_DEFAULTS=1: mu_init 0.1, filter line search of Waechter &
    Biegler alg. A, independent dual step, no cold-start retry) against the shipped algorithm on the same seeded bench instances — the only
    available estimate of how often the reference's IPOPT would end at another KKT point of this non-convex NLP.  256 instances per team size
    here (docs/basin_sensitivity_r4.jsonl holds the 1024-instance run of tools/basin_sensitivity.py on round 4's algorithm: same point on 97.8 % / 50.6 % / 70.8 %; round 3: 97.8 % / 55.3 % / 71.4 %
    of the two / six / ten-robot instances; where the points differ the objectives do by a median 0.1-0.7 %, and neither variant is
    systematically lower).  Asserted: both variants converge everywhere, two robots agree almost always, a different end point is a
    different LOCAL MINIMUM of similar quality (not a failure), and the literal C2 set ends at the same point, the literal, perfectly
    symmetric C6 swap at two local minima of similar quality (550.93 / 552.88; round 3: mirror images of one)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import basin_sensitivity as BS
    res = {nm: BS.compare(nm, 256, str(tmp_path)) for nm in ("two", "six", "ten20")}
    for nm, r in res.items():
        print(nm, r)
        assert r["shipped"]["converged_frac"] == 1.0 and r["ipopt_defaults"]["converged_frac"] >= 0.99, r
        assert r["other_point"]["median_rel_objective_gap"] is None or r["other_point"]["median_rel_objective_gap"] < 0.02, r
    assert res["two"]["same_point_frac_1e-4"] >= 0.95
    assert 0.3 <= res["six"]["same_point_frac_1e-4"] <= 0.8 and 0.5 <= res["ten20"]["same_point_frac_1e-4"] <= 0.9      # measured 0.54 / 0.70: many basins
    assert res["two"]["instance0_same_point"]
    f0 = res["six"]["instance0_objectives"]      # the literal, perfectly symmetric C6 swap: two local minima of similar quality (round 3: mirror images, equal objectives)
    assert abs(f0[0] - f0[1]) <= 1e-2 * abs(f0[0])


def test_partial_refactorisation_constants_agree(tmp_path):
    """Round 4: a rejected pivot resumes the backward sweep at NMPC_RESUME_STAGE(k, N) (include/nmpc_constants.h), the nearest saved stage at
    or above k + NMPC_REFACTOR_BACK.  The macro the kernels use, the oracle's inline form and the numpy prototype's constants must be one
    rule: the macro is compiled and compared with the definition for every (k, N); the prototype (third implementation) with the C oracle."""
    import re, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "nmpc_constants.h")).read()
    C_ = int(re.search(r"#define NMPC_CKPT_EVERY (\d+)", hdr).group(1)); J_ = int(re.search(r"#define NMPC_REFACTOR_BACK (\d+)", hdr).group(1))
    assert (I.CKPT_EVERY, I.REFACTOR_BACK) == (C_, J_)
    src = tmp_path / "rs.c"
    src.write_text('#include <stdio.h>\n#include "%s"\nint main(void){for(int N=2;N<=64;N++)for(int k=0;k<N;k++)printf("%%d %%d %%d\\n",N,k,NMPC_RESUME_STAGE(k,N));return 0;}\n'
                   % os.path.join(root, "include", "nmpc_constants.h"))
    exe = tmp_path / "rs"
    subprocess.check_call(["gcc", "-O0", "-o", str(exe), str(src)])
    for line in subprocess.check_output([str(exe)], text=True).split("\n"):
        if line:
            N, k, kr = map(int, line.split())
            want = min(k + J_, N - 1); want += (N - 1 - want) % C_
            assert kr == want and kr >= k and kr <= N - 1 and (N - 1 - kr) % C_ == 0, (N, k, kr, want)
    # cases that exercise the resume path in both implementations: random six-robot instances (the perfectly symmetric literal swap is decided
    # by the last bits; not used here)
    import ctypes as C
    cfg = R.cfg_six(20)
    P, W0 = Hh.batch(cfg, 3, 2)
    L = O.lib(); L.nmpc_oracle_stats.argtypes = [C.POINTER(C.c_double), C.c_int]
    st4 = (C.c_double * 4)(); L.nmpc_oracle_stats(st4, 1)
    r = O.solve_batch(O.make_config(cfg, max_iter=60), P, W0)
    L.nmpc_oracle_stats(st4, 1)
    assert st4[2] > 0 and st4[0] > st4[3], list(st4)      # rejected pivots occurred, and stages were factored again
    for b in range(3):
        o = I.Opts(); o.max_iter = 60
        rp = I.solve(cfg, P[b], W0[b], o)
        assert rp["status"] == 0 == r["status"][b] and rp["iters"] == r["iters"][b]
        assert np.max(np.abs(rp["x"] - r["x"][b])) < 1e-9


def test_elastic_phase_rescues_the_captured_soak_failures():
    """Round 4: the second restart of last resort is an ELASTIC phase — pair and obstacle rows relaxed to
    h + t - s = 0, t >= 0 under the penalty NMPC_ELASTIC_RHO sum t, from the cold start — where IPOPT would run its restoration phase (it
    replaces round 2's second cold retry with 10 mu_init).  tests/golden/elastic_cases.npz: the 16 root failures of two closed-loop soaks
    of the composite on the GPU (tools/soak_composite.py 512 120, seed indices 5 and 6, round 4's build before this phase; info = seed
    index, period, swarm, status, iterations): status 1 / 4 after the warm start and both cold retries there.  With the elastic phase the
    oracle converges on all of them to feasible KKT points (every elastic variable closed), as it does on every older fixture."""
    rng = np.random.default_rng(7)
    c = R.cfg_six(25); c.rob_dim = 0.2; c.margin = 0.1
    c.obstacles = [(float(x), float(y), float(r_)) for x, y, r_ in zip(rng.uniform(-1.5, 1.5, 8), rng.uniform(-1.5, 1.5, 8), rng.uniform(0.125, 0.2, 8))]
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "elastic_cases.npz"))
    assert len(z["p"]) == 16 and set(z["info"][:, 3].tolist()) <= {1, 4}
    r = O.solve_batch(O.make_config(c, max_iter=2000), z["p"], z["w"])
    assert (r["status"] == 0).all() and (r["kkt"] <= 1e-8).all(), (r["status"], r["iters"])
    assert (r["iters"] > 500).sum() >= 12      # most of them only converge in the elastic phase (after the warm start and the cold retry)
    for b in range(0, 16, 3):
        k = R.kkt_report(c, r["x"][b], z["p"][b], tol_active=1e-3)
        assert k["eq"] < 1e-7 and k["ineq"] < 1e-7 and k["bnd"] < 1e-9, (b, k)      # feasible for the ORIGINAL rows: the elastic variables have closed
