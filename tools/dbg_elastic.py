"""Development aid (GPU box): the captured composite failures / fixtures on a kernel pin against the oracle (elastic phase)."""
import os, sys, glob
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nmpc_amd
from oracle import nlp_ref as R, oracle_lib as O
from tests import helpers as Hh
oc, _, _, _ = Hh.bench_batch("composite", 1)
sets = {}
for f in ["tests/golden/elastic_cases.npz", "tests/golden/cold_retry_cases.npz", "tests/golden/cold_retry_cases2.npz"]:
    z = np.load(f); sets[os.path.basename(f)] = (z["p"], z["w"])
for k, (p, w) in sets.items():
    ref = O.solve_batch(O.make_config(oc, max_iter=2000), p, w)
    for kern in (sys.argv[1:] or ["3"]):
        s = nmpc_amd.NmpcSolver(Hh.to_product_cfg(oc, max_iter=2000), max_batch=len(p), kernel=int(kern))
        r = {a: b.cpu().numpy() for a, b in s.solve_batch(p, w).items()}
        same = np.max(np.abs(r["x"] - ref["x"]), axis=1) <= 1e-6
        print("%-28s kernel %s: status hip %s oracle %s | iters hip %s oracle %s | same point %d of %d | kkt ok %s" % (
            k, kern, r["status"].tolist(), ref["status"].tolist(), r["iters"].tolist(), ref["iters"].tolist(), same.sum(), len(same), bool((r["kkt"][r["status"] == 0] <= 1e-8).all())), flush=True)
