"""Development aid (no GPU): literal lane-level emulation of the ROW-PAIRED backward sweep of solve_col_kernel (csrc/nmpc_solve_col.hip,
`if constexpr (RP)`): 64 lanes, register m[i] = rows (half 0, slot i) and (half 1, slot i), DPP row_newbcast / v_permlane16_swap /
v_permlane32_swap / ds_bpermute / v_readlane written out with numpy index arithmetic.  Checked against tools/lane_emu.lane_sweep (the
one-row-per-register layout) on random stage packs:   python tools/rp_emu.py
"""
import numpy as np
from lane_emu import lane_sweep


def offsets(M_):
    NX, NU = 3 * M_, 2 * M_; NZ = NX + NU; NP = M_ * (M_ - 1) // 2
    PK_G = 0; PK_HD = NZ; PK_HXY = 2 * NZ; PK_HVT = 2 * NZ + M_; PK_E = 2 * NZ + 2 * M_; PK_C = PK_E + 3 * NP; PK_CF = PK_C + NX; PK_ZERO = PK_CF + 3 * NZ
    return (PK_G, PK_HD, PK_HXY, PK_HVT, PK_E, PK_C, PK_CF, PK_ZERO), ((PK_ZERO + 1 + 7) // 8) * 8


def swap16(a, b):      # odd rows (of 16 lanes) of a <-> even rows of b
    a = a.copy(); b = b.copy()
    for r in (0, 2):
        t = a[16 * (r + 1):16 * (r + 2)].copy(); a[16 * (r + 1):16 * (r + 2)] = b[16 * r:16 * (r + 1)]; b[16 * r:16 * (r + 1)] = t
    return a, b


def swap32(a, b):      # upper 32 lanes of a <-> lower 32 lanes of b
    a = a.copy(); b = b.copy()
    t = a[32:].copy(); a[32:] = b[:32]; b[:32] = t
    return a, b


def rowb(u, n):        # row_newbcast:n — lane n of the executing lane's own row of 16
    lanes = np.arange(64)
    return u[(lanes & ~15) + n]


def rp_slot(j): return 2 * ((j >> 1) >> 1) + (j & 1)
def rp_half(j): return (j >> 1) & 1
def rp_live(i, jj, nc, nu): return i >= nc or (4 * (i >> 1) + (i & 1) > jj) or (4 * (i >> 1) + 2 + (i & 1) > jj and 4 * (i >> 1) + 2 + (i & 1) < nu)


def rp_sweep(pk, M_, N, off):
    NX, NU = 3 * M_, 2 * M_; NZ = NX + NU
    PK_G, PK_HD, PK_HXY, PK_HVT, PK_E, PK_C, PK_CF, PK_ZERO = off
    MP = (M_ + 1) // 2; NC = 2 * MP; NS = 3 * MP; RR = NC + NS
    tid = np.arange(64)
    rh = tid >> 5; rw = (tid >> 4) & 1; rn = tid & 15
    cc_ = rn < NC
    cp = np.where(cc_, rn >> 1, (rn - NC) // 3); cd = np.where(cc_, rn & 1, (rn - NC) - 3 * cp); crob = 2 * cp + rw
    cvalid = (rn < RR) & (crob < M_); c_state = cvalid & ~cc_; c_ctrl = cvalid & cc_
    rcol = np.where(cvalid, np.where(cc_, 2 * crob + cd, NU + 3 * crob + cd), 0)
    hoff = np.zeros((RR, 64), dtype=int)
    for i in range(RR):
        au = i < NC; p = (i >> 1) if au else (i - NC) // 3; d = (i & 1) if au else (i - NC) - 3 * p
        rob = 2 * p + rh
        a = np.where(au, 2 * rob + d, NU + 3 * rob + d)
        h = np.full(64, PK_ZERO)
        if (not au) and d < 2:
            lo = np.minimum(rob, crob); hi = np.maximum(rob, crob)
            pe = PK_E + 3 * (lo * (2 * M_ - lo - 1) // 2 + (hi - lo - 1)) + d + cd
            h = np.where(c_state & (cd < 2), np.where(rob == crob, PK_HXY + rob, pe), h)
        if au and d == 0: h = np.where(c_state & (cd == 2) & (crob == rob), PK_HVT + rob, h)
        if (not au) and d == 2: h = np.where(c_ctrl & (cd == 0) & (crob == rob), PK_HVT + rob, h)
        h = np.where(rcol == a, PK_HD + a, h)
        hoff[i] = np.where(cvalid & (rob < M_), h, PK_ZERO)
    goff = np.where(cvalid, PK_G + rcol, PK_ZERO); cfo = PK_CF + 3 * rcol
    cbo = PK_C + 3 * rh; cf1 = PK_CF + 6 * rh; cf2 = PK_CF + 3 * NU + 9 * rh
    lb = tid & ~15
    needxy = (c_state & (cd == 2)) | (c_ctrl & (cd == 0)); needth = c_ctrl & (cd == 1)
    srcA = np.where(needxy, lb + NC + 3 * cp, np.where(needth, lb + NC + 3 * cp + 2, tid))
    srcB = np.where(needxy, lb + NC + 3 * cp + 1, np.where(needth, lb + NC + 3 * cp + 2, tid))
    m = np.zeros((RR + 1, 64))
    pkN = np.concatenate([pk[N], np.zeros(64)])
    hdN = np.where(c_state, pkN[PK_HD + rcol], 0.0); gN = np.where(c_state, pkN[PK_G + rcol], 0.0)
    for r in range(NC, RR): m[r] = np.where(c_state & (rw == rh) & (rn == r), hdN, 0.0)
    m[RR] = gN
    rows = {}
    for k in range(N - 1, -1, -1):
        PK = np.concatenate([pk[k], np.full(64, np.nan)])      # reads past the pack must not matter
        c0, c1, c2 = PK[cfo], PK[cfo + 1], PK[cfo + 2]
        kO = np.where(c_state, c0, 0.0); kA = np.where(c_state, c1, np.where(c_ctrl, c0, 0.0)); kB = np.where(c_state, c2, np.where(c_ctrl, c1, 0.0))
        acc = np.zeros(64)
        for s in range(NS):
            p = s // 3; d = s - 3 * p
            cv = PK[cbo + 6 * p + d]
            if (M_ & 1) and p == MP - 1: cv = np.where(rh == 1, 0.0, cv)
            acc = m[NC + s] * cv + acc
        a2, b2 = swap32(acc, acc)
        m[RR] = m[RR] - (a2 + b2)
        tA = [m[NC + q][srcA] for q in range(NS + 1)]; tB = [m[NC + q][srcB] for q in range(NS + 1)]
        for q in range(NS + 1): m[NC + q] = kB * tB[q] + (kA * tA[q] + kO * m[NC + q])
        for p in range(MP):
            Tc, Ts, Tt = PK[cf1 + 12 * p], PK[cf1 + 12 * p + 1], PK[cf1 + 12 * p + 3]
            ai, bi = PK[cf2 + 18 * p + 7], PK[cf2 + 18 * p + 8]
            if (M_ & 1) and p == MP - 1:
                z = rh == 1
                Tc, Ts, Tt, ai, bi = [np.where(z, 0.0, v) for v in (Tc, Ts, Tt, ai, bi)]
            gx, gy, gt = m[NC + 3 * p].copy(), m[NC + 3 * p + 1].copy(), m[NC + 3 * p + 2].copy()
            m[2 * p] = Ts * gy + Tc * gx; m[2 * p + 1] = Tt * gt; m[NC + 3 * p + 2] = bi * gy + (ai * gx + gt)
        for r in range(RR): m[r] = m[r] + PK[hoff[r]]
        m[RR] = m[RR] + PK[goff]
        R_ = np.zeros((NU, NZ + 1))
        for j in range(NU):
            hj, ij = rp_half(j), rp_slot(j)
            d = m[ij][48 * hj + ij]; inv = 1.0 / d
            rhs_j = m[RR][16 * hj + ij]
            oc, od = swap32(m[ij], m[ij])
            ua, ub = swap16(m[ij], m[ij]); ua, ub = swap32(ua, ub)
            own = od if hj else oc; ur = ub if hj else ua
            for lane in range(32):          # what the lower half stores: the pivot row by natural column index
                if cvalid[lane]: R_[j, rcol[lane]] = own[lane]
            R_[j, NZ] = rhs_j
            rjv = own * inv; nrjv = -rjv
            for i in range(RR):
                if rp_live(i, j, NC, NU): m[i] = m[i] + rowb(ur, i) * nrjv
            m[RR] = m[RR] - rhs_j * rjv
        rows[k] = R_
    # cost-to-go left for the next stage, by natural state index (both halves hold their robots' rows)
    Pk = np.zeros((NX, NX)); pv = np.zeros(NX)
    for lane in range(64):
        if not c_state[lane]: continue
        for s in range(NS):
            p = s // 3; d = s - 3 * p; rob = 2 * p + rh[lane]
            if rob < M_: Pk[3 * rob + d, rcol[lane] - NU] = m[NC + s][lane]
        pv[rcol[lane] - NU] = m[RR][lane]
    return rows, Pk, pv


def random_packs(M_, N, rng):
    off, PACK = offsets(M_)
    PK_G, PK_HD, PK_HXY, PK_HVT, PK_E, PK_C, PK_CF, PK_ZERO = off
    NX, NU = 3 * M_, 2 * M_; NZ = NX + NU; NP = M_ * (M_ - 1) // 2
    pk = np.zeros((N + 1, PACK))
    for k in range(N + 1):
        p = pk[k]
        p[PK_G:PK_G + NZ] = rng.normal(size=NZ)
        p[PK_HD:PK_HD + NZ] = rng.uniform(2.0, 6.0, NZ)
        p[PK_HXY:PK_HXY + M_] = rng.normal(size=M_) * 0.3
        p[PK_HVT:PK_HVT + M_] = rng.normal(size=M_) * 0.3
        p[PK_E:PK_E + 3 * NP] = rng.normal(size=3 * NP) * 0.2
        p[PK_C:PK_C + NX] = rng.normal(size=NX) * 0.1
        T = 0.3
        for i in range(M_):
            th, v = rng.uniform(-3, 3), rng.uniform(-0.2, 0.2)
            c, s = np.cos(th), np.sin(th); cf = p[PK_CF:PK_CF + 3 * NZ]
            cf[3 * (2 * i):3 * (2 * i) + 3] = [T * c, T * s, 0]; cf[3 * (2 * i + 1):3 * (2 * i + 1) + 3] = [T, 0, 0]
            cf[3 * (NU + 3 * i):3 * (NU + 3 * i) + 3] = [1, 0, 0]; cf[3 * (NU + 3 * i + 1):3 * (NU + 3 * i + 1) + 3] = [1, 0, 0]
            cf[3 * (NU + 3 * i + 2):3 * (NU + 3 * i + 2) + 3] = [1, -T * v * s, T * v * c]
    return pk, off


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for M_ in (2, 3, 4, 5, 6):
        N = 4
        pk, off = random_packs(M_, N, rng)
        NX, NU = 3 * M_, 2 * M_
        # lane_emu puts control column a on lane 32 + a; its pack argument carries no padding requirement
        ref = lane_sweep(pk, M_, N, NX, NU, off)
        got, Pk, pv = rp_sweep(pk, M_, N, off)
        err = max(np.abs(ref[k] - got[k]).max() / max(1.0, np.abs(ref[k]).max()) for k in range(N))
        print("m=%d: max rel diff of the pivot rows / right-hand sides over %d stages: %.2e; P_0 symmetric to %.1e" % (M_, N, err, np.abs(Pk - Pk.T).max()))
        assert err < 1e-12
