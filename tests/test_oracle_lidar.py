"""CPU: the LIDAR-ray distance-state NMPC oracle pinned by what the reference text fixes
(V4 = AllScripts/obs_avoid_static_first_scenario_v4.py, V3 = ..._v3.py): layout, cold-start values, bounds as built,
finite-difference derivatives, the least-squares KKT report and scipy-SLSQP on a small instance ("scipy-SLSQP, not CasADi/IPOPT")."""
import numpy as np
import pytest

from oracle import lidar_ref as LR, oracle_lib as O

OBST = [(1.2, 0.9, 0.25), (2.0, 2.2, 0.3), (0.8, -0.6, 0.2)]


def _instance(cfg, pose, xs):
    scan = LR.scan_of_world(pose, OBST, cfg.R)
    return LR.make_p(cfg, pose, xs, scan), LR.cold_start(cfg, np.concatenate([pose, scan]))


def test_layout_and_bounds_as_the_script_builds_them():
    cfg = LR.lidar_v4()
    assert (cfg.ns, cfg.n_var, cfg.n_g, cfg.n_p) == (13, 13 * 101 + 2 * 50, 13 * 101, 26)        # 1413 / 1313
    lbx, ubx, lbg, ubg = LR.bounds(cfg)
    assert lbx.shape == (1413,) and not lbg.any() and not ubg.any()                                # all rows are equalities (V4:158-159)
    # V4:161-172: pose pattern for the first 3(N+1) = 303 entries, distance bounds for the remaining 1010 of the state part
    assert np.array_equal(lbx[:6], [-10, -10, -np.inf, -10, -10, -np.inf]) and (lbx[303:1313] == 0.15).all() and (ubx[303:1313] == 10.0).all()
    assert np.array_equal(lbx[1313:1317], [-0.15, -1.5, -0.15, -1.5])
    # consequence of the misalignment, reproduced: from stage 24 on the POSE entries of w carry the distance bounds [0.15, 10]
    assert lbx[24 * 13] == 0.15 and lbx[24 * 13 + 2] == 0.15 and lbx[23 * 13] == -np.inf and lbx[23 * 13 + 1] == -10.0      # 299 % 3 == 2: the theta pattern
    c3 = LR.lidar_v3()
    assert (c3.N, c3.Nc, c3.lw, c3.n_var) == (125, 125, 0.0, 13 * 126 + 2 * 125)
    # rows: all gx then all gd (V4:151); cold start (V4:184-196): defects and distance rows vanish except the distance rows of
    # the stages, which equal scan - ||p0 - pObs||_1 with pObs one scan length away along each ray
    pose = np.array([0.2, -0.1, 0.3]); xs = np.array([3.0, 2.5, 0.0])
    p, w0 = _instance(cfg, pose, xs)
    g = LR.constraints(cfg, w0, p)
    assert np.abs(g[: 3 * 101]).max() == 0.0 and np.abs(g[3 * 101: 3 * 101 + 10]).max() == 0.0
    B = LR.ray_angles(10); scan = p[6:16]
    want = scan - scan * (np.abs(np.cos(pose[2] + B)) + np.abs(np.sin(pose[2] + B)))
    assert np.abs(g[3 * 101 + 10: 3 * 101 + 20] - want).max() < 1e-14
    f0 = LR.objective(cfg, w0, p)
    assert f0 == pytest.approx(100 * (np.dot(cfg.q, (pose - xs) ** 2) + 0.1 * np.sum(1.0 / scan ** 2)), rel=1e-13)
    # shift (V4:258-270): row N-1 appended, controls drop first / repeat last
    X, U = LR.unpack(cfg, np.arange(cfg.n_var, dtype=float))
    Xs, Us = LR.unpack(cfg, LR.shift_guess(cfg, np.arange(cfg.n_var, dtype=float)))
    assert np.array_equal(Xs[:-1], X[1:]) and np.array_equal(Xs[-1], X[cfg.N - 1]) and np.array_equal(Us[:-1], U[1:]) and np.array_equal(Us[-1], U[-1])


def test_derivatives_and_c_evaluation():
    cfg = LR.LidarConfig(N=8, Nc=4, R=4, aligned_bounds=True)
    pose = np.array([0.0, 0.0, 0.1]); xs = np.array([1.5, 0.8, 0.0])
    p, w0 = _instance(cfg, pose, xs)
    rng = np.random.default_rng(0)
    w = w0 + 0.05 * rng.normal(size=w0.size); w[3: 3 + cfg.R] = np.abs(w[3: 3 + cfg.R]) + 0.5
    J = LR.jacobian(cfg, w, p); gf = LR.grad_objective(cfg, w, p)
    h = 1e-6
    for i in range(w.size):
        e = np.zeros(w.size); e[i] = h
        assert np.abs((LR.constraints(cfg, w + e, p) - LR.constraints(cfg, w - e, p)) / (2 * h) - J[:, i]).max() < 1e-7, i
        assert abs((LR.objective(cfg, w + e, p) - LR.objective(cfg, w - e, p)) / (2 * h) - gf[i]) < 1e-6, i
    f, g = O.lidar_eval_batch(cfg, p[None], w[None])
    assert abs(f[0] - LR.objective(cfg, w, p)) < 1e-12 and np.abs(g[0] - LR.constraints(cfg, w, p)).max() < 1e-14


@pytest.mark.parametrize("name,cfg", [("v4", LR.lidar_v4()), ("v3", LR.lidar_v3()), ("v4_aligned", LR.LidarConfig(aligned_bounds=True)),
                                      ("small", LR.LidarConfig(N=12, Nc=6, R=4, aligned_bounds=True))])
def test_oracle_solutions_are_kkt_points(name, cfg):
    """the solve against the solver-independent KKT report (least-squares multipliers on the active set, full-space Jacobian
    including the dependence of the lidar points on the stage-0 variables)."""
    pose = np.array([0.0, 0.0, 0.0]); xs = np.array([3.0, 2.5, 0.0])
    p, w0 = _instance(cfg, pose, xs)
    r = O.lidar_solve_batch(cfg, p[None], w0[None], max_iter=1500)
    assert r["status"][0] == 0 and r["kkt"][0] <= 1e-8, (r["status"], r["iters"], r["kkt"])
    k = LR.kkt_report(cfg, r["x"][0], p, tol_active=1e-4)
    assert k["stat"] < 1e-5 and k["eq"] < 1e-9 and k["bnd"] == 0.0, k
    X, U = LR.unpack(cfg, r["x"][0])
    assert np.array_equal(X[0, :3], pose) and np.array_equal(X[0, 3:], p[6: 6 + cfg.R])              # X_0 pinned
    if name == "v4":       # the misaligned bounds are active: the pose of stage 24 sits on its "distance" bound 0.15
        assert abs(X[24, 0] - 0.15) < 1e-6


def test_oracle_matches_slsqp_small():
    """an INDEPENDENT solver (scipy SLSQP) on the restated NLP, small enough for its dense algebra."""
    from scipy.optimize import Bounds, minimize
    cfg = LR.LidarConfig(N=10, Nc=5, R=3, aligned_bounds=True)
    for pose, xs in ((np.array([0.0, 0.0, 0.1]), np.array([1.5, 0.8, 0.0])), (np.array([0.3, -0.2, 1.0]), np.array([-1.0, 0.5, 0.5]))):
        p, w0 = _instance(cfg, pose, xs)
        lbx, ubx, _, _ = LR.bounds(cfg)
        res = minimize(lambda w: LR.objective(cfg, w, p), w0, jac=lambda w: LR.grad_objective(cfg, w, p), bounds=Bounds(lbx, ubx),
                       constraints=[{"type": "eq", "fun": lambda w: LR.constraints(cfg, w, p), "jac": lambda w: LR.jacobian(cfg, w, p)}],
                       method="SLSQP", options={"ftol": 1e-14, "maxiter": 500})
        r = O.lidar_solve_batch(cfg, p[None], w0[None], max_iter=500)
        assert r["status"][0] == 0
        assert abs(r["f"][0] - res.fun) <= 1e-6 * max(1.0, abs(res.fun)), (r["f"][0], res.fun)
        assert np.abs(r["x"][0] - res.x).max() < 2e-4


def _lidar_golden():
    """tests/golden/slsqp_lidar.npz (generator tests/golden/gen_golden.py lidar): (name, cfg, p, w0, w*, f*) — "scipy-SLSQP, not CasADi/IPOPT" """
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "slsqp_lidar.npz"))
    for name in sorted({k.rsplit("_cfg", 1)[0] for k in z.files if k.endswith("_cfg")}):
        N, Nc, Rr, al = (int(v) for v in z[name + "_cfg"])
        yield name, LR.LidarConfig(N=N, Nc=Nc, R=Rr, aligned_bounds=bool(al)), z[name + "_p"], z[name + "_w0"], z[name + "_w_pol"], float(z[name + "_f_pol"])


def test_oracle_matches_slsqp_lidar_golden():
    """the committed SLSQP triples of the LIDAR-state NLP (R = 3, 4 and 10 rays), one of them with the bounds exactly as the script builds them."""
    n = 0
    for name, cfg, p, w0, ws, fs in _lidar_golden():
        lbx, ubx, _, _ = LR.bounds(cfg)
        r = O.lidar_solve_batch(cfg, p[None], w0[None], max_iter=500, lbx=lbx, ubx=ubx)
        assert r["status"][0] == 0, (name, r["status"], r["iters"])
        assert abs(r["f"][0] - fs) <= 1e-6 * max(1.0, abs(fs)), (name, r["f"][0], fs)
        assert np.abs(r["x"][0] - ws).max() < 2e-4, (name, np.abs(r["x"][0] - ws).max())
        n += 1
    assert n == 5


def test_oracles_are_clean_under_asan_and_ubsan():
    """VERDICT r1 / SURVEY 5: `-fsanitize=address,undefined` on the CPU restatement (GPU sanitizers are not available on this
    pool): both oracles solve fixed problems under ASan + UBSan + leak check (oracle/asan_driver.c, `make -C oracle asan_check`)."""
    import os, subprocess
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    r = subprocess.run(["make", "-C", here, "-B", "asan_check"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ASAN_DRIVER_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def test_cold_start_retry_rescues_the_crawling_instance(monkeypatch):
    """tests/golden/lidar_cold_retry_case.npz: the one instance of 16,384 seeded V4 instances that needs 2074 iterations at
    mu_init = 0.5 (and 23-42 from any other initial barrier parameter).  Without the retry: max_iter; with it: the third attempt
    (cold start, mu = 10 mu_init) converges, 1024 iterations in all."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lidar_cold_retry_case.npz"))
    cfg = LR.lidar_v4()
    lbx, ubx, _, _ = LR.bounds(cfg)
    r = O.lidar_solve_batch(cfg, g["p"], g["w0"], max_iter=2000, lbx=lbx, ubx=ubx)
    assert r["status"][0] == 0 and 1000 <= r["iters"][0] <= 1100, (r["status"], r["iters"])
    k = LR.kkt_report(cfg, r["x"][0], g["p"][0], tol_active=1e-4)
    assert k["stat"] < 1e-5 and k["eq"] < 1e-9 and k["bnd"] == 0.0, k
    monkeypatch.setenv("NMPC_ORACLE_NO_COLD_RETRY", "1")
    r0 = O.lidar_solve_batch(cfg, g["p"], g["w0"], max_iter=2000, lbx=lbx, ubx=ubx)
    assert r0["status"][0] == 1 and r0["iters"][0] == 2000, (r0["status"], r0["iters"])


def test_line_search_watchdog_cuts_the_long_solves(monkeypatch):
    """tests/golden/lidar_watchdog_cases.npz (gen_lidar_watchdog_cases.py): the three instances of bench.py's LIDAR batch that need 175 / 154 /
    145 iterations without the line-search watchdog (include/nmpc_constants.h: after ten shortened steps the next step with a
    fraction-to-the-boundary length >= 0.5 is taken without the merit test) and the five longest solves that remain with it.  With the
    watchdog every one converges in at most 61 iterations, to the same objective as without."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lidar_watchdog_cases.npz"))
    cfg = LR.lidar_v4()
    lbx, ubx, _, _ = LR.bounds(cfg)
    r = O.lidar_solve_batch(cfg, g["p"], g["w0"], max_iter=2000, lbx=lbx, ubx=ubx)
    assert (r["status"] == 0).all() and (r["iters"] == g["iters_with"]).all() and r["iters"].max() <= 61, (r["status"], r["iters"])
    for b in range(len(g["p"])):
        k = LR.kkt_report(cfg, r["x"][b], g["p"][b], tol_active=1e-4)
        assert k["stat"] < 1e-5 and k["eq"] < 1e-9 and k["bnd"] == 0.0, (b, k)
    monkeypatch.setenv("NMPC_LIDAR_ORACLE_WD", "0")
    r0 = O.lidar_solve_batch(cfg, g["p"], g["w0"], max_iter=2000, lbx=lbx, ubx=ubx)
    assert (r0["status"] == 0).all() and (r0["iters"] == g["iters_without"]).all() and sorted(r0["iters"])[-3:] == [145, 154, 175], r0["iters"]
    assert np.max(np.abs(r["f"] - r0["f"]) / np.abs(r0["f"])) < 1e-8 and np.max(np.abs(r["x"] - r0["x"])) < 1e-4, (r["f"] - r0["f"], np.max(np.abs(r["x"] - r0["x"])))
