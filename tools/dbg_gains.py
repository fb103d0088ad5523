"""Development aid: compare the Riccati feedback gains of the two kernels after one iteration."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nmpc_amd
from oracle import nlp_ref as R
from tests import helpers as Hh
m = int(sys.argv[1]); N = 8
ocfg = R.NLPConfig(m=m, N=N, T=0.1, dmin=0.3, v_max=0.22, w_max=2.84)
nx, nu = 3 * m, 2 * m
P, W0 = Hh.batch(ocfg, 4, 3)
res = {}
for kern in ("1", "2"):
    os.environ["NMPC_KERNEL"] = kern
    s = nmpc_amd.NmpcSolver(Hh.to_product_cfg(ocfg, max_iter=1), max_batch=4)
    s.solve_batch(P, W0); torch.cuda.synchronize()
    f = s.lib.nmpc_debug_workspace; f.restype = C.c_int64; f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
    offs = (C.c_int64 * 5)()
    per = f(s._h, 0, None, 0, offs)
    buf = np.zeros(per); f(s._h, 0, buf.ctypes.data, per, offs)
    if kern == "1":
        K = buf[offs[1]: offs[1] + N * nu * nx].reshape(N, nu, nx); kff = buf[offs[2]: offs[2] + N * nu].reshape(N, nu)
    else:
        kts = ((nx + 1) * nu + 7) // 8 * 8
        KT = np.stack([buf[offs[4] + k * kts: offs[4] + k * kts + (nx + 1) * nu].reshape(nx + 1, nu) for k in range(N)])
        K = KT[:, :nx, :].transpose(0, 2, 1); kff = KT[:, nx, :]
    res[kern] = (K, kff)
for k in range(N - 1, -1, -1):
    dK = np.abs(res["1"][0][k] - res["2"][0][k]); dk = np.abs(res["1"][1][k] - res["2"][1][k])
    print("stage %d: max|dK| %.2e at %s   max|dkff| %.2e at %d   (|K| %.2e)" % (k, dK.max(), np.unravel_index(dK.argmax(), dK.shape), dk.max(), dk.argmax(), np.abs(res["1"][0][k]).max()))
np.set_printoptions(linewidth=250, precision=3, suppress=False)
k = N - 1
print("kff v1", res["1"][1][k]); print("kff v2", res["2"][1][k])
print("K v1 rows 10..15, cols 0..8\n", res["1"][0][k][10:16, :9]); print("K v2\n", res["2"][0][k][10:16, :9])
