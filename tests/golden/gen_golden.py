"""Generates tests/golden/slsqp_*.npz: (p, w0, w*, f*) triples of the reference NLP solved by an INDEPENDENT solver.

    python tests/golden/gen_golden.py

"scipy-SLSQP oracle, not CasADi/IPOPT": the reference's own solver stack (casadi, IPOPT) is not
installed and the reference holds no vectors, so the golden solutions come from
scipy.optimize.minimize(method='SLSQP') applied to oracle/nlp_ref.py's restatement of the NLP
(C6:273-339), started from the reference's cold start (C6:398-400).  The literal start/goal
sets of the scripts are instance 0 where they exist.  Each triple also stores the SLSQP
solution re-polished from itself (w_pol) as a convergence witness.
"""
import os
import sys

import numpy as np
from scipy.optimize import Bounds, minimize

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nlp_ref as R  # noqa: E402
from tests import helpers as Hh  # noqa: E402


def slsqp(cfg, p, w0):
    lbx, ubx, lbg, ubg = R.bounds(cfg)
    eq = np.where(lbg == ubg)[0]; iq = np.where(lbg != ubg)[0]
    cons = [{"type": "eq", "fun": lambda w: R.constraints(cfg, w, p)[eq] - lbg[eq], "jac": lambda w: R.jacobian(cfg, w, p)[eq]}]
    if iq.size:
        cons.append({"type": "ineq", "fun": lambda w: R.constraints(cfg, w, p)[iq] - lbg[iq], "jac": lambda w: R.jacobian(cfg, w, p)[iq]})
    r = minimize(lambda w: R.objective(cfg, w, p), w0, jac=lambda w: R.grad_objective(cfg, w, p), bounds=Bounds(lbx, ubx),
                 constraints=cons, method="SLSQP", options={"ftol": 1e-14, "maxiter": 1000})
    return r


def make(name, cfg, P):
    rows = []
    for p in P:
        w0 = R.cold_start(cfg, p[: cfg.nx])
        r = slsqp(cfg, p, w0)
        r2 = slsqp(cfg, p, r.x)
        # status 8 ("positive directional derivative") is SLSQP's way of saying it cannot improve below ftol = 1e-14
        assert r.status in (0, 8) and r2.status in (0, 8), (name, r.message, r2.message)
        k = R.kkt_report(cfg, r2.x, p, tol_active=1e-6)
        assert k["stat"] < 1e-4 and k["eq"] < 1e-9 and k["ineq"] < 1e-9 and k["bnd"] < 1e-12, (name, k)
        rows.append((p, w0, r.x, r.fun, r2.x, r2.fun))
        print(name, "f* = %.9f  (polish moved w by %.1e)" % (r.fun, np.max(np.abs(r.x - r2.x))))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "slsqp_%s.npz" % name)
    np.savez_compressed(out, p=np.stack([r[0] for r in rows]), w0=np.stack([r[1] for r in rows]), w=np.stack([r[2] for r in rows]),
                        f=np.array([r[3] for r in rows]), w_pol=np.stack([r[4] for r in rows]), f_pol=np.array([r[5] for r in rows]))


def lidar_cases():
    """(name, LidarConfig, pose, goal) of tests/golden/slsqp_lidar.npz: four triples with aligned bounds and one with the bounds exactly as
    the script builds them (V4:161-176: misaligned with its own packing — from some stage on the POSE entries carry the distance
    bounds [0.15, 10])."""
    from oracle import lidar_ref as LR
    return [("a", LR.LidarConfig(N=10, Nc=5, R=3, aligned_bounds=True), np.array([0.0, 0.0, 0.1]), np.array([1.5, 0.8, 0.0])),
            ("b", LR.LidarConfig(N=10, Nc=5, R=3, aligned_bounds=True), np.array([0.3, -0.2, 1.0]), np.array([-1.0, 0.5, 0.5])),
            ("c", LR.LidarConfig(N=15, Nc=8, R=4, aligned_bounds=True), np.array([0.1, 0.0, 0.4]), np.array([1.8, 1.2, 0.3])),
            ("d_script_bounds", LR.LidarConfig(N=12, Nc=6, R=4), np.array([0.2, 0.25, 0.3]), np.array([1.5, 1.0, 0.5])),
            # R = 10 rays as in V3 / V4: the ray count the HIP kernel is specialised for (round 3)
            ("e_ten_rays", LR.LidarConfig(N=8, Nc=4, R=10, aligned_bounds=True), np.array([0.1, 0.05, 0.5]), np.array([1.6, 1.1, 0.2]))]


LIDAR_WORLD = [(1.2, 0.9, 0.25), (2.0, 2.2, 0.3)]      # the obstacle world of tests/test_oracle_lidar.py


def make_lidar():
    """scipy-SLSQP on oracle/lidar_ref.py's restatement of the LIDAR-state NLP (V4:78-151), cold start V4:184-196."""
    from oracle import lidar_ref as LR
    out = {}
    for name, cfg, pose, xs in lidar_cases():
        scan = LR.scan_of_world(pose, LIDAR_WORLD, cfg.R)
        p = LR.make_p(cfg, pose, xs, scan); w0 = LR.cold_start(cfg, np.concatenate([pose, scan]))
        lbx, ubx, _, _ = LR.bounds(cfg)

        def run(wstart):
            return minimize(lambda w: LR.objective(cfg, w, p), wstart, jac=lambda w: LR.grad_objective(cfg, w, p), bounds=Bounds(lbx, ubx),
                            constraints=[{"type": "eq", "fun": lambda w: LR.constraints(cfg, w, p), "jac": lambda w: LR.jacobian(cfg, w, p)}],
                            method="SLSQP", options={"ftol": 1e-14, "maxiter": 1000})
        r = run(w0); r2 = run(r.x)
        assert r.status in (0, 8) and r2.status in (0, 8), (name, r.message, r2.message)
        k = LR.kkt_report(cfg, r2.x, p, tol_active=1e-4)
        assert k["eq"] < 1e-9 and k["bnd"] < 1e-12, (name, k)
        print("lidar", name, "f* = %.9f  (polish moved w by %.1e), kkt_report %s" % (r.fun, np.max(np.abs(r.x - r2.x)), k))
        out.update({name + "_p": p, name + "_w0": w0, name + "_w": r.x, name + "_f": r.fun, name + "_w_pol": r2.x, name + "_f_pol": r2.fun,
                    name + "_cfg": np.array([cfg.N, cfg.Nc, cfg.R, int(cfg.aligned_bounds)])})
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "slsqp_lidar.npz"), **out)


if __name__ == "__main__":
    if "lidar" in sys.argv[1:]:
        make_lidar()
        sys.argv.remove("lidar")
        if not sys.argv[1:]:
            sys.exit(0)
    # usage: gen_golden.py [name ...]   (default: the four small sets; "six" takes ~8 min, "ten" ~25 min of one core)
    want = sys.argv[1:] or ["one", "two", "obs3", "three"]
    rng = np.random.Generator(np.random.PCG64(Hh.SEED0))
    c1 = R.cfg_one(20)
    P1 = [np.array([0.0, 0.0, 0.0, 1.5, 1.5, 0.0])] + [Hh.instance(rng, c1) for _ in range(5)]   # C1:177 first goal
    c2 = R.cfg_two(20)
    close = np.array([-0.2, 0.0, 0.0, 0.2, 0.02, np.pi, 0.6, 0.0, 0.0, -0.6, 0.0, np.pi])
    P2 = [np.concatenate([R.C2_START, R.C2_GOAL]), close] + [Hh.instance(rng, c2) for _ in range(4)]
    co = R.cfg_obs3(20)
    c3 = R.NLPConfig(m=3, N=10, T=0.3, dmin=0.4, v_max=0.15, w_max=1.5)
    P3 = [Hh.instance(rng, c3) for _ in range(3)]
    if "one" in want: make("one", c1, P1)
    if "two" in want: make("two", c2, P2)
    if "obs3" in want: make("obs3", co, [np.array([0.0, 0.2 + 0.1 * t, 1.2, 0.2 * t, 2.6, 1.57]) for t in range(4)])
    if "three" in want: make("three", c3, P3)
    # pair rows AND obstacle rows in one NLP (the composite's row types): three robots crossing a field of two obstacles; own stream
    if "mix3" in want:
        cm = Hh.cfg_mix3(10); rm = np.random.Generator(np.random.PCG64(Hh.SEED0 + 333))
        Pm = []
        for _ in range(4):
            s = rm.uniform(-0.15, 0.15, (3, 2)) + np.array([[-0.6, -0.5], [1.0, -0.4], [0.2, 1.2]])
            gq = rm.uniform(-0.15, 0.15, (3, 2)) + np.array([[0.9, 0.7], [-0.6, 0.6], [0.1, -0.7]])
            Pm.append(np.concatenate([np.concatenate([s, rm.uniform(-np.pi, np.pi, (3, 1))], axis=1).reshape(-1), np.concatenate([gq, rm.uniform(-np.pi, np.pi, (3, 1))], axis=1).reshape(-1)]))
        make("mix3", cm, Pm)
    # the headline configuration (BASELINE configs[2]): the literal antipodal swap of C6:364-388 plus four drawn instances,
    # and two ten-robot instances at the file's own horizon N=20; own generator streams so that the
    # small sets above stay byte-identical
    if "six" in want:
        c6 = R.cfg_six(20); r6 = np.random.Generator(np.random.PCG64(Hh.SEED0 + 600))
        make("six", c6, [np.concatenate([R.C6_START, R.C6_GOAL])] + [Hh.instance(r6, c6) for _ in range(4)])
    if "ten" in want:
        c10 = R.cfg_ten(20); r10 = np.random.Generator(np.random.PCG64(Hh.SEED0 + 1000))
        make("ten", c10, [Hh.instance(r10, c10) for _ in range(2)])
