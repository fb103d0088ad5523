"""CPU only: how often does an IPOPT-faithful globalisation end in another KKT point than the shipped algorithm?  

The reference's solver (CasADi + IPOPT) cannot be run here.  oracle/nmpc_oracle.c carries an IPOPT-faithful variant (NMPC_ORACLE_IPOPT_DEFAULTS=1:
mu_init 0.1, filter line search of Waechter & Biegler alg. A, independent dual step length, no cold-start retry) next to the shipped algorithm
(mu_init 0.5, l1 merit with non-monotone reference, dual step capped by the primal one, cold-start retry).  Both are run on the same seeded
instances (the bench batches, bench.make_batch: instance 0 is the literal start/goal set of the script where one exists) from the reference's
cold start; reported per team size: converged fractions, iteration counts, the fraction of instances on which both end at the SAME point
(max |dw| <= 1e-6, and <= 1e-4), objective agreement, and which of two different end points has the lower objective.

    python tools/basin_sensitivity.py [n=1024] [names=two,six,ten20]      -> one JSON line per team size
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import bench
from oracle import oracle_lib as O
name, n, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
from tests import helpers as Hh
ocfg, B, P, W0 = Hh.bench_batch(name, n)
r = O.solve_batch(O.make_config(ocfg), P, W0)
np.savez(out, x=r["x"], f=r["f"], status=r["status"], iters=r["iters"], kkt=r["kkt"])
''' % ROOT


def run(name, n, mode, tmp):
    env = dict(os.environ)
    env.pop("NMPC_ORACLE_IPOPT_DEFAULTS", None)
    if mode:
        env["NMPC_ORACLE_IPOPT_DEFAULTS"] = "1"
    out = os.path.join(tmp, "basin_%s_%d_%d.npz" % (name, n, mode))
    subprocess.check_call([sys.executable, "-c", CHILD, name, str(n), out], env=env)      # a child: the mode is read when the library's workspace is built
    return np.load(out)


def compare(name, n, tmp="/tmp"):
    a, b = run(name, n, 0, tmp), run(name, n, 1, tmp)
    both = (a["status"] == 0) & (b["status"] == 0)
    dw = np.max(np.abs(a["x"] - b["x"]), axis=1)
    df = (b["f"] - a["f"]) / np.maximum(1.0, np.abs(a["f"]))
    other = both & (dw > 1e-4)
    return {"workload": name, "instances": int(n),
            "shipped": {"converged_frac": float((a["status"] == 0).mean()), "mean_iters": float(a["iters"].mean()), "max_iters": int(a["iters"].max())},
            "ipopt_defaults": {"converged_frac": float((b["status"] == 0).mean()), "mean_iters": float(b["iters"].mean()), "max_iters": int(b["iters"].max()),
                               "status_counts": {str(k): int((b["status"] == k).sum()) for k in np.unique(b["status"])}},
            "same_point_frac_1e-6": float((dw[both] <= 1e-6).mean()), "same_point_frac_1e-4": float((dw[both] <= 1e-4).mean()),
            "same_objective_frac_1e-6": float((np.abs(df[both]) <= 1e-6).mean()),
            "other_point": {"count": int(other.sum()), "ipopt_defaults_lower_objective_frac": float((df[other] < 0).mean()) if other.any() else None,
                            "median_rel_objective_gap": float(np.median(np.abs(df[other]))) if other.any() else None},
            "instance0_same_point": bool(dw[0] <= 1e-4), "instance0_objectives": [float(a["f"][0]), float(b["f"][0])]}


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["two", "six", "ten20"]
    for nm in names:
        print(json.dumps(compare(nm, n)), flush=True)
