"""Development aid (CPU only): the predictor-corrector experiment of VERDICT r1 item 7 on the CPU oracle.

    NMPC_ORACLE_PC=0|1|2|3 python tools/pc_experiment.py [B]

0 = the algorithm as shipped; 1 = second-order complementarity corrector (ds dz / s of the plain step) with the barrier parameter
unchanged; 2 = Mehrotra (affine-scaling predictor, mu scaled by (mu_aff/mu)^3); 3 = corrector taken only at its full
fraction-to-the-boundary length and only when it meets the plain step's Armijo bound, else the plain step with its line search.
Results: DESIGN.md 7."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nlp_ref as R, oracle_lib as O
from tests import helpers as Hh
def run(name, ocfg, B, idx):
    P, W0 = Hh.batch(ocfg, B, idx)
    t=time.time(); r = O.solve_batch(O.make_config(ocfg, max_iter=2000), P, W0); dt=time.time()-t
    it=r['iters']; st=r['status']
    print(f"PC={os.environ.get('NMPC_ORACLE_PC','0')} {name}: mean iters {it.mean():.2f} p50 {np.percentile(it,50):.0f} p99 {np.percentile(it,99):.0f} max {it.max()} converged {(st==0).mean():.4f} status counts {dict(zip(*np.unique(st, return_counts=True)))} [{dt:.1f}s]", flush=True)
    return r
B=int(sys.argv[1]) if len(sys.argv)>1 else 1024
run('six N=20', R.cfg_six(20), B, 2)
run('two N=20', R.cfg_two(20), B, 1)
run('ten N=20', R.cfg_ten(20), B//4, 3)
