"""-m gpu: the probe for the one route to a pin against the real CasADi/IPOPT solve.

If `import casadi` succeeds on the GPU box, the build's own generator (oracle/casadi_probe.py, no reference file involved) builds the NLP of
a1-a7 through the CasADi API with the options of C6:345, solves the literal C2 / C6 start-goal sets and the five scipy-SLSQP six-robot
inputs single-threaded, and the HIP output is compared with it: the same point at 1e-6 where the basins agree, two KKT points
(solver-independent report) where they do not.  Either way ONE line says whether casadi was found, so the GPU test record answers the
question."""
import os

import numpy as np
import pytest

from oracle import casadi_probe as CP, nlp_ref as R
from tests import helpers as Hh

pytestmark = pytest.mark.gpu


def test_casadi_ipopt_pin_if_casadi_is_importable(built, capsys):
    ok, what = CP.available()
    with capsys.disabled():
        print("\n[casadi probe] " + ("FOUND: " + what if ok else "not importable on this box (%s): parity against CasADi/IPOPT stays unpinned" % what))
    if not ok:
        pytest.skip("casadi is not importable here: " + what)
    import torch
    import nmpc_amd
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "slsqp_six.npz"))
    cases = [("C2 literal", R.cfg_two(20), np.concatenate([R.C2_START, R.C2_GOAL])[None]),
             ("C6 literal + SLSQP inputs", R.cfg_six(20), np.concatenate([np.concatenate([R.C6_START, R.C6_GOAL])[None], z["p"]]))]
    for name, ocfg, P in cases:
        W0 = np.stack([R.cold_start(ocfg, p[: ocfg.nx]) for p in P])
        ref = CP.solve(ocfg, P, W0)
        r = {k: v.cpu().numpy() for k, v in nmpc_amd.NmpcSolver(Hh.to_product_cfg(ocfg, max_iter=2000), max_batch=len(P)).solve_batch(P, W0).items()}
        torch.cuda.synchronize()
        same = np.max(np.abs(r["x"] - ref["x"]), axis=1) <= 1e-6
        with capsys.disabled():
            print("[casadi probe] %s: same point as nlpsol('ipopt') on %d of %d; IPOPT %s; %.3f s per IPOPT solve (one thread)" % (
                name, same.sum(), len(same), sorted(set(ref["return_status"])), ref["seconds"].mean()))
        assert (r["status"] == 0).all()
        for b in range(len(P)):
            if same[b]:
                assert abs(r["f"][b] - ref["f"][b]) <= 1e-6 * max(1.0, abs(ref["f"][b]))
            else:       # another basin: both must be KKT points of the same NLP (solver-independent report)
                for x in (r["x"][b], ref["x"][b]):
                    k = R.kkt_report(ocfg, x, P[b], tol_active=1e-4)
                    assert k["stat"] < 1e-4 and k["eq"] < 1e-6 and k["ineq"] < 1e-6, (name, b, k)
        assert same[0] or name != "C2 literal"      # two robots: one basin expected
