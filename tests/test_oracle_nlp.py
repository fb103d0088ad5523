"""CPU: the numpy / C restatement of the reference NLP against everything the reference text pins
without a solver."""
import numpy as np
import pytest

from oracle import nlp_ref as R, oracle_lib as O

CFGS = {"one": R.cfg_one(20), "two": R.cfg_two(20), "six": R.cfg_six(20), "ten": R.cfg_ten(30), "obs3": R.cfg_obs3(20),
        "two70": R.cfg_two(70), "six35": R.cfg_six(35), "one100": R.cfg_one(100)}


@pytest.mark.parametrize("name", list(CFGS))
def test_layout_invariants(name):
    """item 1: len(w) = n_var, len(g) = (n_x+M)(N+1) for the padded multi-robot scripts, block order."""
    c = CFGS[name]
    lbx, ubx, lbg, ubg = R.bounds(c)
    assert lbx.size == ubx.size == c.n_var == c.nx * (c.N + 1) + c.nu * c.N
    assert lbg.size == ubg.size == c.n_g
    if c.pad_rows and not c.obstacles:
        assert c.n_g == (c.nx + c.M) * (c.N + 1)                       # C6:349
    rng = np.random.default_rng(0)
    w = rng.normal(size=c.n_var); p = rng.normal(size=2 * c.nx)
    g = R.constraints(c, w, p)
    X, U = R.unpack(c, w)
    np.testing.assert_array_equal(g[: c.nx], X[0] - p[: c.nx])          # first block [X_0 - x0; 3.5 x M]  (C6:278)
    if c.pad_rows:
        np.testing.assert_array_equal(g[c.nx: c.rows0], np.full(c.M, 3.5))
        np.testing.assert_array_equal(lbg[c.nx: c.rows0], np.full(c.M, c.dmin ** 2))
    # stage block: defect rows first, then d_12, d_13, ... evaluated at X_k (not X_{k+1})
    k = 3
    blk = g[c.rows0 + k * c.rows_k: c.rows0 + (k + 1) * c.rows_k]
    np.testing.assert_allclose(blk[: c.nx], X[k + 1] - (X[k] + c.T * R.rhs(c, X[k], U[k])), rtol=0, atol=1e-15)
    for q, (i, j) in enumerate(c.pairs()):
        assert blk[c.nx + q] == (X[k][3 * i] - X[k][3 * j]) ** 2 + (X[k][3 * i + 1] - X[k][3 * j + 1]) ** 2
    assert c.pairs() == sorted(c.pairs())                                # lexicographic 12,13,..,23,..
    # packing is stage-major, robot-major inside a stage (column-major reshape of n_x x (N+1), C6:339)
    assert w[c.nx * 2 + 1] == X[2, 1] and w[c.nx * (c.N + 1) + c.nu * 3 + 1] == U[3, 1]
    # bounds literals (C6:349-352)
    assert set(np.unique(ubx[: c.nx * (c.N + 1)])) <= {10.0, np.inf, 2 * np.pi}
    assert ubx[-2] == c.v_max and ubx[-1] == c.w_max and lbx[-1] == -c.w_max


@pytest.mark.parametrize("cfg,start,goal", [(R.cfg_two(20), R.C2_START, R.C2_GOAL), (R.cfg_six(20), R.C6_START, R.C6_GOAL),
                                             (R.cfg_two(70), R.C2_START, R.C2_GOAL), (R.cfg_six(35), R.C6_START, R.C6_GOAL)])
def test_cold_start_known_answers(cfg, start, goal):
    """item 2: at w = [repmat(x0); 0] defects are exactly 0, pair rows equal d_ij(x0), f = N (x0-xs)^T Q (x0-xs)."""
    p = np.concatenate([start, goal]); w = R.cold_start(cfg, start)
    g = R.constraints(cfg, w, p)
    q = np.tile(cfg.q, cfg.m)
    assert R.objective(cfg, w, p) == pytest.approx(cfg.N * np.sum(q * (start - goal) ** 2), rel=1e-14)
    for k in range(cfg.N):
        blk = g[cfg.rows0 + k * cfg.rows_k: cfg.rows0 + (k + 1) * cfg.rows_k]
        assert np.all(blk[: cfg.nx] == 0.0)
        for qi, (i, j) in enumerate(cfg.pairs()):
            assert blk[cfg.nx + qi] == (start[3 * i] - start[3 * j]) ** 2 + (start[3 * i + 1] - start[3 * j + 1]) ** 2
    # the scripts' own scenarios are feasible at the start
    _, _, lbg, _ = R.bounds(cfg)
    assert np.all(g >= lbg - 1e-15)


def test_fixed_point_is_optimal():
    """item 3: x0 = xs (pairwise feasible) => U = 0, X_k = xs is a KKT point with f = 0."""
    for cfg, xs in ((R.cfg_two(20), R.C2_GOAL), (R.cfg_six(20), R.C6_GOAL)):
        p = np.concatenate([xs, xs]); w = R.cold_start(cfg, xs)
        assert R.objective(cfg, w, p) == 0.0
        assert np.all(R.grad_objective(cfg, w, p) == 0.0)
        k = R.kkt_report(cfg, w, p)
        assert k["stat"] < 1e-12 and k["eq"] == 0.0 and k["ineq"] == 0.0


def test_shift_known_answer():
    """item 4: u = [[1,2],[3,4],[5,6]] -> [[3,4],[5,6],[5,6]]; the state guess appends row N-1 (C6:465)."""
    t0, u0 = R.shift(0.05, 1.0, np.array([[1, 2], [3, 4], [5, 6]]))
    assert t0 == 1.05
    np.testing.assert_array_equal(u0, [[3, 4], [5, 6], [5, 6]])
    cfg = R.NLPConfig(m=1, N=3)
    X = np.arange(12.0).reshape(4, 3)
    np.testing.assert_array_equal(R.shift_states(cfg, X), np.vstack([X[1:], X[2:3]]))


@pytest.mark.parametrize("name", ["one", "two", "six", "obs3"])
def test_derivatives_against_finite_differences(name):
    """item 5: analytic grad f, J_g and the Lagrangian Hessian vs central differences."""
    c = CFGS[name]
    rng = np.random.default_rng(3)
    w = rng.normal(scale=0.7, size=c.n_var); p = rng.normal(size=2 * c.nx); lam = rng.normal(size=c.n_g)
    h = 1e-6
    gf = R.grad_objective(c, w, p); J = R.jacobian(c, w, p); H = R.hess_lagrangian(c, w, p, lam)
    idx = rng.choice(c.n_var, size=min(c.n_var, 40), replace=False)
    for i in idx:
        e = np.zeros(c.n_var); e[i] = h
        assert (R.objective(c, w + e, p) - R.objective(c, w - e, p)) / (2 * h) == pytest.approx(gf[i], abs=1e-6)
        np.testing.assert_allclose((R.constraints(c, w + e, p) - R.constraints(c, w - e, p)) / (2 * h), J[:, i], atol=2e-7)
        gl = lambda v: R.grad_objective(c, v, p) + R.jacobian(c, v, p).T @ lam
        np.testing.assert_allclose((gl(w + e) - gl(w - e)) / (2 * h), H[:, i], atol=5e-6)


@pytest.mark.parametrize("name", ["one", "two", "six", "ten", "obs3"])
def test_c_eval_and_shift_match_numpy(name):
    """the C restatement evaluates f, g (reference row order) and shift exactly like the numpy one."""
    c = CFGS[name]
    rng = np.random.default_rng(5)
    B = 5
    P = rng.normal(size=(B, 2 * c.nx)); W = rng.normal(size=(B, c.n_var))
    cc = O.make_config(c)
    f, g = O.eval_batch(cc, P, W)
    for b in range(B):
        assert f[b] == pytest.approx(R.objective(c, W[b], P[b]), rel=1e-13)
        np.testing.assert_allclose(g[b], R.constraints(c, W[b], P[b]), rtol=0, atol=1e-13)
    wn, x0n = O.shift_batch(cc, P, W)
    for b in range(B):
        X, U = R.unpack(c, W[b])
        np.testing.assert_array_equal(wn[b], R.pack(c, R.shift_states(c, X), R.shift(c.T, 0.0, U)[1]))
        np.testing.assert_allclose(x0n[b], R.plant_step(c, P[b, : c.nx], U[0]), atol=1e-15)
    assert O.lib().nmpc_oracle_n_var(cc) == c.n_var and O.lib().nmpc_oracle_n_g(cc) == c.n_g


def test_odometry_known_answers():
    """C2:18-37: identity start frame leaves the position alone; a quarter-turn start frame maps x_r onto +y."""
    p = R.odom_to_global([[0.3, -0.2, 0.0, 1.0]], [[0.0, 0.0, 0.0]])[0]
    assert np.allclose(p, [0.3, -0.2, 0.0], atol=1e-16)
    p = R.odom_to_global([[1.0, 0.0, np.sin(np.pi / 8), np.cos(np.pi / 8)]], [[2.0, 3.0, np.pi / 2]])[0]
    assert np.allclose(p, [2.0, 4.0, np.pi / 4 + np.pi / 2], atol=1e-15)
