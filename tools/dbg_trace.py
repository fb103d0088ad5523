"""Development aid: per-iteration trace of one instance (NMPC_PROFILE build)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NMPC_PROFILE"] = "1"
inst = int(sys.argv[1]); B = int(sys.argv[2]); mi = int(sys.argv[3]) if len(sys.argv) > 3 else 100
os.environ["NMPC_TRACE_INST"] = str(inst)
import importlib, torch, nmpc_amd
from oracle import nlp_ref as R
from tests import helpers as Hh
importlib.import_module("nmpc_amd.build").build(force=True)
ocfg = R.cfg_six(20)
P, W0 = Hh.batch(ocfg, B, 2)
s = nmpc_amd.NmpcSolver(Hh.to_product_cfg(ocfg, max_iter=mi), max_batch=B)
r = s.solve_batch(P, W0); torch.cuda.synchronize()
n = int(r["iters"][inst]) + 1
out = np.zeros((n, 16))
s.lib.nmpc_debug_trace(s._h, out.ctypes.data, n)
out2 = np.zeros((n, 8))
s.lib.nmpc_debug_trace2(s._h, out2.ctypes.data, n)
print("it      E0      e_d      e_c      e_h    szmax       mu    alpha      a_p      a_d    delta       nu     dphi      th0            f   s_d  mult")
for i in range(n):
    t = out[i]
    print("%3d %8.1e %8.1e %8.1e %8.1e %8.1e %8.1e %8.1e %8.1e %8.1e %8.1e %8.1e %8.1e %8.1e %12.6f %5.2f %8.1e" % ((i,) + tuple(t[:16])), " | code %6.0f  % .2e % .2e % .2e % .2e | lin % .2e |lamR-lamA| % .2e linR % .2e" % tuple(out2[i][:8]))
