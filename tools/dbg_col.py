"""Development aid (GPU box): column-per-lane kernel (NMPC_KERNEL=3) against the element-per-lane kernel (NMPC_KERNEL=2) after ONE
iteration: iterate difference and the stage factors (pivot rows, reciprocal pivots) of the per-instance workspace, entry by entry."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nmpc_amd
from oracle import nlp_ref as R
from tests import helpers as Hh

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1
its = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ocfg = {1: R.cfg_one(20), 2: R.cfg_two(20), 6: R.cfg_six(20), 10: R.cfg_ten(20)}[m]
P, W0 = Hh.batch(ocfg, 4, {1: 0, 2: 1, 6: 2, 10: 3}[m])
out = {}
for kern in ("2", "3"):
    os.environ["NMPC_KERNEL"] = kern
    s = nmpc_amd.NmpcSolver(Hh.to_product_cfg(ocfg, max_iter=its), max_batch=4)
    r = s.solve_batch(P, W0); torch.cuda.synchronize()
    L = s.lib
    offs = (C.c_int64 * 5)()
    per = L.nmpc_debug_workspace(s._h, 0, None, 0, offs)
    buf = np.zeros(per)
    L.nmpc_debug_workspace(s._h, 0, buf.ctypes.data_as(C.c_void_p), per, offs)
    out[kern] = (r["x"].cpu().numpy(), r["kkt"].cpu().numpy(), buf, list(offs), r["f"].cpu().numpy())
x2, k2, b2, o2, f2 = out["2"]; x3, k3, b3, o3, f3 = out["3"]
print("kkt", k2, k3); print("f", f2, f3)
print("max|dx| after %d iteration(s): %.3e" % (its, np.abs(x2 - x3).max()))
nx, nu = ocfg.nx, ocfg.nu; nz = nx + nu; LD = nz + 1
kts = (nu * LD + nu + 7) // 8 * 8
oKT = o2[4]
N = ocfg.N
for k in range(N - 1, -1, -1):
    a = b2[oKT + k * kts: oKT + (k + 1) * kts]; b = b3[oKT + k * kts: oKT + (k + 1) * kts]
    A = a[: nu * LD].reshape(nu, LD); Bm = b[: nu * LD].reshape(nu, LD)
    D = np.abs(A - Bm)
    for j in range(nu):
        D[j, :j] = 0          # below the diagonal: dead entries
    print("stage", k, "max factor diff", D.max(), "at", np.unravel_index(D.argmax(), D.shape), "inv diff", np.abs(a[nu * LD: nu * LD + nu] - b[nu * LD: nu * LD + nu]).max())
    if D.max() > float(os.environ.get('DBG_TOL', '1e-9')):
        np.set_printoptions(linewidth=250, precision=int(os.environ.get('DBG_PREC', '4')), suppress=False)
        print("old\n", A); print("new\n", Bm); print("diff\n", A - Bm)
        if not os.environ.get('DBG_ALL'): break

# ---- recompute the backward sweep of the LAST iteration in numpy from the stage packs left in the workspace and compare the
#      factors of both kernels with it (delta = 0 assumed)
M_ = m; NP = M_ * (M_ - 1) // 2; NX = nx; NU = nu; NZ = nz
PK_G, PK_HD, PK_HXY, PK_HVT = 0, NZ, 2 * NZ, 2 * NZ + M_
PK_E = 2 * NZ + 2 * M_; PK_C = PK_E + 3 * NP; PK_CF = PK_C + NX; PK_ZERO = PK_CF + 3 * NZ
PACK = (PK_ZERO + 1 + 7) // 8 * 8
oPACK = o2[3]


def pidx(a, b):
    return a * (2 * M_ - a - 1) // 2 + (b - a - 1)


def sweep(buf):
    pk = buf[oPACK: oPACK + (N + 1) * PACK].reshape(N + 1, PACK)
    Pm = np.diag(pk[N, PK_HD + NU: PK_HD + NZ]); pv = pk[N, PK_G + NU: PK_G + NZ].copy()
    rows = {}
    for k in range(N - 1, -1, -1):
        q = pk[k]
        cf = q[PK_CF: PK_CF + 3 * NZ].reshape(NZ, 3)
        W = np.zeros((NX, NZ))
        for i in range(M_):
            W[3 * i, 2 * i] = cf[2 * i, 0]; W[3 * i + 1, 2 * i] = cf[2 * i, 1]; W[3 * i + 2, 2 * i + 1] = cf[2 * i + 1, 0]
            for d in range(3):
                W[3 * i + d, NU + 3 * i + d] = cf[NU + 3 * i + d, 0]
            W[3 * i, NU + 3 * i + 2] = cf[NU + 3 * i + 2, 1]; W[3 * i + 1, NU + 3 * i + 2] = cf[NU + 3 * i + 2, 2]
        H = np.diag(q[PK_HD: PK_HD + NZ])
        for i in range(M_):
            H[2 * i, NU + 3 * i + 2] += q[PK_HVT + i]; H[NU + 3 * i + 2, 2 * i] += q[PK_HVT + i]
            H[NU + 3 * i, NU + 3 * i + 1] += q[PK_HXY + i]; H[NU + 3 * i + 1, NU + 3 * i] += q[PK_HXY + i]
            for j in range(i + 1, M_):
                e = q[PK_E + 3 * pidx(i, j): PK_E + 3 * pidx(i, j) + 3]
                for da in range(2):
                    for dc in range(2):
                        H[NU + 3 * i + da, NU + 3 * j + dc] += e[da + dc]; H[NU + 3 * j + dc, NU + 3 * i + da] += e[da + dc]
        pb = pv - Pm @ q[PK_C: PK_C + NX]
        Mx = W.T @ Pm @ W + H
        qq = W.T @ pb + q[PK_G: PK_G + NZ]
        R_ = np.zeros((NU, LD))
        for j in range(NU):
            R_[j, :NZ] = Mx[j]; R_[j, NZ] = qq[j]
            d = Mx[j, j]; rj = Mx[j] / d
            for a in range(j + 1, NZ):
                Mx[a] -= Mx[j, a] * rj
            qq = qq - qq[j] * rj if False else qq - R_[j, NZ] * rj
        rows[k] = R_
        Pm = Mx[NU:, NU:]; Pm = 0.5 * (Pm + Pm.T); pv = qq[NU:]
    return rows


for nm, bb in (("old", b2), ("new", b3)):
    rows = sweep(bb)
    print("---- kernel", nm, "against the numpy recomputation from its own stage packs")
    for k in range(N - 1, -1, -1):
        a = bb[oKT + k * kts: oKT + (k + 1) * kts][: nu * LD].reshape(nu, LD)
        D = np.abs(a - rows[k])
        for j in range(nu):
            D[j, :j] = 0
        print("  stage %2d max|factor - numpy| %.3e at %s (|value| %.3e)" % (k, D.max(), np.unravel_index(D.argmax(), D.shape), np.abs(rows[k]).max()))

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
try:
    from lane_emu import lane_sweep
    rows_e = lane_sweep(b3[oPACK: oPACK + (N + 1) * PACK].reshape(N + 1, PACK), M_, N, NX, NU, (PK_G, PK_HD, PK_HXY, PK_HVT, PK_E, PK_C, PK_CF, PK_ZERO))
    rows_n = sweep(b3)
    print("---- lane-level emulation of the column kernel against numpy / against the kernel's factors")
    for k in range(N - 1, -1, -1):
        a = b3[oKT + k * kts: oKT + (k + 1) * kts][: nu * LD].reshape(nu, LD)
        D1 = np.abs(rows_e[k] - rows_n[k]); D2 = np.abs(rows_e[k] - a)
        for j in range(nu):
            D1[j, :j] = 0; D2[j, :j] = 0
        print("  stage %2d emu-numpy %.3e   emu-kernel %.3e" % (k, D1.max(), D2.max()))
except ImportError:
    pass
np.savez(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "dbg_ws_m%d_it%d.npz" % (m, its)), b2=b2, b3=b3, o2=np.array(o2), P=P, W0=W0)
