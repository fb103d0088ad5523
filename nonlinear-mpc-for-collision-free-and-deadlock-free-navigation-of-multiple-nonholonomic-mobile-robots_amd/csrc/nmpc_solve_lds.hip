// nmpc_solve_lds.hip — latency-oriented gfx950 solve kernel (the hot kernel).
//
// A launch takes as long as "start of the longest instance + its length", so this kernel minimises the latency of ONE
// interior-point iteration of ONE instance:
//   * one wavefront (64 lanes) per swarm instance up to six robots (2 / 4 waves for 8 / 10 robots, and for small batches) —
//     every "barrier" of a single-wave instance is a wave-local LDS fence;
//   * the iterate (X, U, lambda, pair / obstacle slacks and duals, control-bound slacks and duals, state-bound duals, sin/cos
//     cache) lives in LDS for the entire solve; the step and the adjoint residuals overlay the Riccati working set;
//   * everything of a stage that does not depend on the cost-to-go (gradients, barrier Hessian terms, the 3x2 Jacobian entries
//     T cos, T sin, -T v sin, T v cos per robot) is computed for all stages in parallel ("stage packs", written to HBM/L2 and
//     prefetched one stage ahead);
//   * one Riccati stage = assemble the augmented symmetric matrix
//         [ Quu Qux | qu ]      = [B A]^T P [B A] + H     (upper triangle + rhs column)
//         [ Qxu Qxx | qx ]
//     directly from P with the robot-sparse A, B (<= 3 terms per index), held in REGISTERS, one static set of elements per
//     lane (per-lane offset tables rebuilt at the start of every sweep); NU pivot steps of symmetric elimination publish one
//     pivot row per step through LDS; what remains in the registers is [P_k | p_k].  No feedback gains are formed: the pivot
//     rows and reciprocal pivots stream to HBM/L2 and the forward sweep does one triangular solve per stage with v_readlane;
//   * every wave reduction ends in v_readfirstlane (uniform_f64): the control decisions of the solve are scalar branches;
//   * multipliers: stage-parallel residual pass + robot-local adjoint recursion in registers;
//   * a stall (5 steps below 1e-10) restarts the barrier iteration from the interior-pushed current point, at most 3 times.
//
// Reference blocks replaced: see nmpc_kernels.hip (same algorithm, same constants as the oracle).
#include "nmpc_solve_common.h"

namespace nmpc {

template <int M_, int THB, int TPB>
__global__ __launch_bounds__(TPB) void solve_lds_kernel(const KParams P, const double *__restrict__ p_in, const double *__restrict__ w0,
                                                         double *__restrict__ w_out, double *__restrict__ obj_out,
                                                         int32_t *__restrict__ status_out, int32_t *__restrict__ iters_out,
                                                         double *__restrict__ kkt_out, double *__restrict__ ws, long long *__restrict__ prof_out)
{
    using G = G2<M_, THB>;
    constexpr int NX = G::NX, NU = G::NU, NP = G::NP, NZ = G::NZ, LD = G::LD, NXB = G::NXB, NT = G::NT;
    constexpr int NTP = (NT + TPB - 1) / TPB;
    constexpr int HG = (TPB / NZ) > 0 ? (TPB / NZ) : 1;   // row groups of the G pass (thread -> column, rows strided)
    constexpr int NPd = NP > 0 ? NP : 1;   // divisor that stays legal for M_ == 1 (those loops have zero trips)
    const int tid = threadIdx.x;
    const int N = P.N, N1 = P.N + 1, K = P.K, MK = M_ * P.K;
    const double T = P.T;
    const size_t inst = (P.order && *P.order_bad == 0) ? (size_t)P.order[blockIdx.x] : (size_t)blockIdx.x;      // dispatch-order hint: long solves first (ignored unless it is a permutation)
    const bool prs = P.pairs != 0;                 // pair rows present (the no-pair multi-robot NLP keeps NP slots that are never touched)
    const int NPA = prs ? G::NP : 0;

    extern __shared__ double sm[];
    double *X = sm;                       // [N1*NX]
    double *U = X + N1 * NX;              // [N*NU]
    double *LAM = U + N * NU;             // [N1*NX]   lam[k] pairs with defect c_{k-1}
    double *SPp = LAM + N1 * NX;          // [N1*NP]   pair slacks
    double *ZPp = SPp + N1 * NP;          // [N1*NP]   pair duals
    double *SO = ZPp + N1 * NP;           // [N1*MK]   obstacle slacks
    double *ZO = SO + N1 * MK;            // [N1*MK]
    double *ZUL = ZO + N1 * MK;           // [N*NU]
    double *ZUU = ZUL + N * NU;           // [N*NU]
    double *ZXL = ZUU + N * NU;           // [N1*NXB]
    double *ZXU = ZXL + N1 * NXB;         // [N1*NXB]
    double *SUL = ZXU + N1 * NXB;         // [N*NU]    explicit slacks of the simple bounds: s = u - lb computed on the fly
    double *SUU = SUL + N * NU;           // [N*NU]    loses all relative accuracy once s ~ 1e-12 (active bound at mu = 1e-9)
    double *SN = SUU + N * NU;            // [N*M]      (state-bound slacks are implicit: s = x + b, b - x; |x| <= 10 keeps them O(1) accurate)
    double *CS = SN + N * M_;             // [N*M]
    double *Pf = CS + N * M_;             // [NX*NX]   cost-to-go Hessian P (full symmetric storage)
    double *PV = Pf + NX * NX;            // [NX]      cost-to-go gradient p
    double *Gb = PV + NX;                 // [NX*LDG]  P [B A] and, in column NZ, p + P b
    double *UR = Gb + NX * G::LDG;        // [NU*LD]   published pivot rows
    double *INV = UR + NU * LD;           // [NU]      reciprocal pivots (contiguous with UR: streamed out together)
    double *PK = INV + NU;                // [PACK]    current stage pack
    double *D0 = PK + G::PACK;            // [NU]      pivots as assembled
    double *XS = D0 + NU;                 // [NX]
    double *RED = XS + NX;                // [8]
    // Overlays.  W1 = [Pf | PV | Gb] and W2 = [UR | INV | PK] are only live during the Riccati sweep; the step (DX, DU) is
    // produced after it and consumed before the next one, so it lives in W1 when it fits; the stage residuals RV of the
    // adjoint recursion and the forward-sweep staging of the factors live in W2.
    constexpr int W1S = NX * NX + NX + NX * G::LDG, W2S = NU * LD + NU + G::PACK;
    double *extra = RED + 8;
    const bool ovA = N1 * NX + N * NU <= W1S, ovB = N1 * NX <= W2S;
    double *DX = ovA ? Pf : extra;        // [N1*NX]
    double *DU = DX + N1 * NX;            // [N*NU]
    if (!ovA) extra += N1 * NX + N * NU;
    double *RV = ovB ? UR : extra;        // [N1*NX]   stage residuals / QP multipliers of the adjoint recursion (ACT follows, see below)
    double *FS = UR;                      // forward-sweep staging of one stage factor ([UR rows | INV], KTS doubles <= W2S)

    double *gpack = ws + inst * P.stride2 + P.oPACK;   // [N][PACK] + terminal [2*NX]
    double *gkt = ws + inst * P.stride2 + P.oKT;       // [N][KTS]
    double *TPp = ws + inst * P.stride2 + P.oELAS;     // [N1*NP]  elastic variables of the pair rows (elastic phase only), obstacle rows behind them
    double *TOb = TPp + N1 * NP;                       // [N1*MK]
    const double rho = P.rho_el;
    double *gck = ws + inst * P.stride2 + P.oCKPT;     // saved cost-to-go [P | p] of the backward sweep, one slot of (NX + 1) * 64 doubles per NMPC_CKPT_EVERY stages
    const double *pp = p_in + inst * (2 * NX);
    const double *wi = w0 + inst * (size_t)P.nvar;
    double *wo = w_out + inst * (size_t)P.nvar;

#ifdef NMPC_PROFILE
    long long prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = clock64();
#endif

    auto lbu = [&](int c) { return (c & 1) ? -P.wmax : -P.vmax; };
    auto bst = [&](int s) { return THB ? s : 3 * (s >> 1) + (s & 1); };            // bounded-state slot -> state index
    auto bvl = [&](int s) { return (THB && (s % 3 == 2)) ? P.thmax : P.xymax; };

    // ---- static element ownership of the augmented matrix: e -> (row a, col c), c == NZ is the rhs column.
    // Everything an element needs per stage is an LDS offset fixed for the whole solve (branch-free assembly):
    //   value = sum_t CF[a][t] * G[ix(a,t)][c] + PK[hoff] + dl * delta
    auto term_ix = [&](int a, int t) -> int {     // state index of term t of row/column a of [B A]
        if (a < NU) { int i = a >> 1; return (a & 1) ? 3 * i + 2 : ((t == 1) ? 3 * i + 1 : 3 * i); }
        int s = a - NU, i = s / 3, d = s - 3 * i;
        return (d < 2) ? s : ((t == 0) ? s : (t == 1 ? 3 * i : 3 * i + 1));
    };
    // Static element ownership e -> (row a, col c) is decoded once and parked in LDS as 16-bit (a | c << 8) words; the
    // per-element offset tables are rebuilt from it at the start of every sweep, so they only occupy registers while
    // a sweep runs (kept for the whole kernel they cost ~100 VGPRs in every other phase and pushed the kernel into scratch).
    const int oUR = (int)(UR - sm) * 8, oGb = (int)(Gb - sm) * 8, oPK = (int)(PK - sm) * 8, oPf = (int)(Pf - sm) * 8;
    unsigned short *ACT = reinterpret_cast<unsigned short *>(extra + ((ovB ? 0 : N1 * NX)));    // [NTP*TPB]
#ifdef NMPC_POISON
    {   // debug build: every LDS word and the instance's HBM workspace start as NMPC_POISON, so that a read of anything this
        // solve did not write shows up as a parity failure instead of depending on what ran on the CU before
        const int nl = (int)(reinterpret_cast<double *>(ACT) - sm) + (NTP * TPB * 2 + 7) / 8;
        for (int e = tid; e < nl; e += TPB) sm[e] = NMPC_POISON;
        for (int e = tid; e < P.stride2; e += TPB) ws[inst * P.stride2 + e] = NMPC_POISON;
        __syncthreads();
    }
#endif
#pragma unroll
    for (int t = 0; t < NTP; t++) {
        int e = tid + t * TPB, a = 0;
        unsigned short w = 0xFFFF;
        if (e < NT) {
            while (e >= NZ - a + 1) { e -= NZ - a + 1; a++; }
            w = (unsigned short)(a | ((a + e) << 8));          // e in [0, NZ-a]: columns a..NZ
        }
        ACT[t * TPB + tid] = w;
    }
    __syncthreads();
    // the strictly lower part of the pivot-row buffer stays zero for the whole solve: the branch-free rank-1 update
    // multiplies by UR[j][a], which must vanish for rows a < j that are already eliminated
    for (int e = tid; e < NU * LD; e += TPB) UR[e] = 0.0;
    // G pass: lane -> column gcol of [B A], rows grow0, grow0 + HG, ...
    const int gcol = tid % NZ, grow0 = tid / NZ;
    const bool gact = tid < HG * NZ;
    const int gx0 = term_ix(gcol, 0), gx1 = term_ix(gcol, 1), gx2 = term_ix(gcol, 2);
    constexpr int GRPT = (NX + HG - 1) / HG;                                   // rows per thread in the G pass
    constexpr int PBP = (64 / NX >= 3 && NX % 3 == 0) ? 3 : ((64 / NX >= 2 && NX % 2 == 0) ? 2 : 1), PBC = (NX + PBP - 1) / PBP;   // parts / columns per part of the P b product (exact splits only: nine robots, NX = 27, take one part)
    static_assert(PBP * PBC == NX, "P b split must be exact");
    const int gpb0 = oPf + 8 * (grow0 * NX + gx0), gpb1 = oPf + 8 * (grow0 * NX + gx1), gpb2 = oPf + 8 * (grow0 * NX + gx2);
    const int gwb = oGb + 8 * (grow0 * G::LDG + gcol);

    // ---- load start, pin X_0, push into the interior of the simple bounds (IPOPT bound_push)
    const double bp = 1e-2;
    for (int c = tid; c < NX; c += TPB) XS[c] = pp[NX + c];
    for (int e = tid; e < N1 * NX; e += TPB) {
        int k = e / NX, c = e - k * NX;
        double v = (k == 0) ? pp[c] : wi[e];
        if (k >= 1) {
            int d = c % 3;
            if (d < 2 || THB) {
                double b = (d == 2) ? P.thmax : P.xymax, px = fmin(bp * fmax(1.0, b), bp * 2.0 * b);
                v = fmin(fmax(v, -b + px), b - px);
            }
        }
        X[e] = v; LAM[e] = 0.0; DX[e] = 0.0;     // the step is read (times a = 0) by the first merit evaluation: stale LDS may hold NaN patterns
    }
    for (int e = tid; e < N * NU; e += TPB) {
        int c = e % NU;
        double lo = lbu(c), hi = -lo, pu = fmin(bp * fmax(1.0, fabs(lo)), bp * (hi - lo));
        U[e] = fmin(fmax(wi[(size_t)N1 * NX + e], lo + pu), hi - pu);
        DU[e] = 0.0;
    }
    __syncthreads();

    // ---- stage-0 pair / obstacle rows act on the pinned state: feasibility pre-check
    {
        double bad = 0.0;
        for (int q = tid; q < NPA; q += TPB) {
            int i = 0, r = q;
            while (r >= M_ - 1 - i) { r -= M_ - 1 - i; i++; }
            int j = i + 1 + r;
            double dx = X[3 * i] - X[3 * j], dy = X[3 * i + 1] - X[3 * j + 1];
            if (h_pair(dx, dy, P.dmin2) < -NMPC_X0_TOL) bad = 1.0;
        }
        for (int e = tid; e < MK; e += TPB) {
            int i = e / K, o = e - i * K;
            double dx = X[3 * i] - P.obs[3 * o], dy = X[3 * i + 1] - P.obs[3 * o + 1];
            if (h_obs(r_obs(dx, dy), P.robdim, P.obs[3 * o + 2], P.margin) < -NMPC_X0_TOL) bad = 1.0;
        }
        bad = wmax<TPB>(bad, RED);
        if (bad > 0.0) {
            for (int e = tid; e < N1 * NX; e += TPB) wo[e] = X[e];
            for (int e = tid; e < N * NU; e += TPB) wo[(size_t)N1 * NX + e] = U[e];
            if (tid == 0) {
                if (obj_out) obj_out[inst] = NAN;
                if (status_out) status_out[inst] = NMPC_STATUS_INFEASIBLE_X0;
                if (iters_out) iters_out[inst] = 0;
                if (kkt_out) kkt_out[inst] = INFINITY;
            }
            return;
        }
    }

    // pair (i,j) of flat index q
    double mu = P.mu_init;      // barrier parameter (declared here: the merit function of the elastic phase needs it)
    bool el = false;            // elastic phase (the second restart of last resort; oracle/nmpc_oracle.c, nmpc_solve_col.hip)
    auto pair_ij = [&](int q, int &i, int &j) { i = 0; while (q >= M_ - 1 - i) { q -= M_ - 1 - i; i++; } j = i + 1 + q; };

    // ---- trig cache of the current iterate
    auto trig = [&]() {
        for (int it = tid; it < N * M_; it += TPB) {
            int k = it / M_, i = it - k * M_;
            double s, c;
            sincos(X[k * NX + 3 * i + 2], &s, &c);
            SN[it] = s; CS[it] = c;
        }
    };
    // ---- merit pieces at (X + a DX, U + a DU, S + a DS): objective, sum log s, l1 infeasibility, max defect / slack residual.
    //      a == 0 evaluates the current point.  Every inequality row carries an explicit slack.
    auto merit = [&](double a, double &fv, double &lg, double &th, double &ec_, double &eh_) {
        double fs = 0.0, l = 0.0, t = 0.0, mc = 0.0, mh = 0.0;
        // one bound row: current slack sv, current value h0, step of the value jd, trial value ht
        double pr = 1.0;       // product of the slacks of one (stage, robot) item: one logarithm per item (see nmpc_solve_col.hip)
        auto brow = [&](double sv, double h0, double jd, double ht) {
            double st = sv + a * (jd + (h0 - sv));
            pr *= st;
            double r = fabs(ht - st);
            t += r; mh = fmax(mh, r);
        };
        for (int it = tid; it < N * M_; it += TPB) {
            int k = it / M_, i = it - k * M_;
            const int ox = k * NX + 3 * i, ou = k * NU + 2 * i;
            double x0 = X[ox] + a * DX[ox], x1 = X[ox + 1] + a * DX[ox + 1], x2 = X[ox + 2] + a * DX[ox + 2];
            double n0 = X[ox + NX] + a * DX[ox + NX], n1 = X[ox + NX + 1] + a * DX[ox + NX + 1], n2 = X[ox + NX + 2] + a * DX[ox + NX + 2];
            double u0 = U[ou] + a * DU[ou], u1 = U[ou + 1] + a * DU[ou + 1];
            double s, c;
            if (a == 0.0) { s = SN[it]; c = CS[it]; } else sincos(x2, &s, &c);
            double c0 = fabs(defect_xy(n0, x0, T * u0, c)), c1 = fabs(defect_xy(n1, x1, T * u0, s)), c2 = fabs(defect_th(n2, x2, T, u1));
            t += c0 + c1 + c2; mc = fmax(mc, fmax(c0, fmax(c1, c2)));
            double e0 = x0 - XS[3 * i], e1 = x1 - XS[3 * i + 1], e2 = x2 - XS[3 * i + 2];
            fs += P.q[0] * e0 * e0 + P.q[1] * e1 * e1 + P.q[2] * e2 * e2 + P.r[0] * u0 * u0 + P.r[1] * u1 * u1;
            // control bounds of stage k
            brow(SUL[ou], U[ou] + P.vmax, DU[ou], u0 + P.vmax); brow(SUU[ou], P.vmax - U[ou], -DU[ou], P.vmax - u0);
            brow(SUL[ou + 1], U[ou + 1] + P.wmax, DU[ou + 1], u1 + P.wmax); brow(SUU[ou + 1], P.wmax - U[ou + 1], -DU[ou + 1], P.wmax - u1);
            // state bounds of stage k+1
            pr *= ((n0 + P.xymax) * (P.xymax - n0)) * ((n1 + P.xymax) * (P.xymax - n1));
            if (THB) pr *= (n2 + P.thmax) * (P.thmax - n2);
            l += log(pr);
            pr = 1.0;
        }
        for (int it = tid; it < (N - 1) * NPA; it += TPB) {
            int k = 1 + it / NPd, q = it - (k - 1) * NP, i, j;
            pair_ij(q, i, j);
            const int oi = k * NX + 3 * i, oj = k * NX + 3 * j;
            double dx = (X[oi] + a * DX[oi]) - (X[oj] + a * DX[oj]), dy = (X[oi + 1] + a * DX[oi + 1]) - (X[oj + 1] + a * DX[oj + 1]);
            double h = h_pair(dx, dy, P.dmin2);
            // trial slack: s + a ds, ds = J dx + (h0 - s)
            double sv = SPp[k * NP + q], tv = 0.0;
            if (el) {
                tv = TPp[k * NP + q];
                if (a != 0.0) {
                    double ex = X[oi] - X[oj], ey = X[oi + 1] - X[oj + 1], ds, dz, dt;
                    el_step(mu, rho, sv, ZPp[k * NP + q], tv, h_pair(ex, ey, P.dmin2), jd_pair(ex, ey, DX[oi] - DX[oj], DX[oi + 1] - DX[oj + 1]), ds, dz, dt);
                    sv += a * ds; tv += a * dt;
                }
                l += log(tv); fs += rho * tv;
            } else if (a != 0.0) {
                double ex = X[oi] - X[oj], ey = X[oi + 1] - X[oj + 1];
                sv += a * ds_pair(ex, ey, DX[oi] - DX[oj], DX[oi + 1] - DX[oj + 1], P.dmin2, sv);
            }
            l += log(sv);
            double r = fabs(h - sv + tv);
            t += r; mh = fmax(mh, r);
        }
        for (int it = tid; it < (N - 1) * MK; it += TPB) {
            int k = 1 + it / MK, e = it - (k - 1) * MK, i = e / K, o = e - i * K;
            const int oi = k * NX + 3 * i;
            double px = X[oi] + a * DX[oi], py = X[oi + 1] + a * DX[oi + 1];
            double h = h_obs(r_obs(px - P.obs[3 * o], py - P.obs[3 * o + 1]), P.robdim, P.obs[3 * o + 2], P.margin);
            double sv = SO[k * MK + e], tv = 0.0;
            if (el) {
                tv = TOb[k * MK + e];
                if (a != 0.0) {
                    double ex = X[oi] - P.obs[3 * o], ey = X[oi + 1] - P.obs[3 * o + 1], rr = r_obs(ex, ey), ds, dz, dt;
                    el_step(mu, rho, sv, ZO[k * MK + e], tv, h_obs(rr, P.robdim, P.obs[3 * o + 2], P.margin), jd_obs(ex, ey, rr, DX[oi], DX[oi + 1]), ds, dz, dt);
                    sv += a * ds; tv += a * dt;
                }
                l += log(tv); fs += rho * tv;
            } else if (a != 0.0) {
                double ex = X[oi] - P.obs[3 * o], ey = X[oi + 1] - P.obs[3 * o + 1], rr = r_obs(ex, ey);
                sv += a * ds_obs(ex, ey, rr, DX[oi], DX[oi + 1], h_obs(rr, P.robdim, P.obs[3 * o + 2], P.margin), sv);
            }
            l += log(sv);
            double r = fabs(h - sv + tv);
            t += r; mh = fmax(mh, r);
        }
        fv = wsum<TPB>(fs, RED); lg = wsum<TPB>(l, RED); th = wsum<TPB>(t, RED);
        ec_ = wmax<TPB>(mc, RED); eh_ = wmax<TPB>(mh, RED);
    };

    // ---- slacks and duals from the current primal point (also the barrier restart after a stall)
    auto init_barrier = [&]() {
        for (int it = tid; it < N1 * NP; it += TPB) {
            int k = it / NPd, q = it - k * NP;
            if (k >= 1 && k <= N - 1 && prs) {
                int i, j; pair_ij(q, i, j);
                double dx = X[k * NX + 3 * i] - X[k * NX + 3 * j], dy = X[k * NX + 3 * i + 1] - X[k * NX + 3 * j + 1];
                const double hv = h_pair(dx, dy, P.dmin2);
                if (el) { const double tv = fmax(bp, bp - hv), sv = hv + tv; TPp[it] = tv; SPp[it] = sv; ZPp[it] = fmin(mu / sv, 0.5 * rho); }
                else { double sv = fmax(hv, bp); SPp[it] = sv; ZPp[it] = mu / sv; }
            } else { SPp[it] = 1.0; ZPp[it] = 0.0; }
        }
        for (int it = tid; it < N1 * MK; it += TPB) {
            int k = it / MK, e = it - k * MK;
            if (k >= 1 && k <= N - 1) {
                int i = e / K, o = e - i * K;
                double dx = X[k * NX + 3 * i] - P.obs[3 * o], dy = X[k * NX + 3 * i + 1] - P.obs[3 * o + 1];
                const double hv = h_obs(r_obs(dx, dy), P.robdim, P.obs[3 * o + 2], P.margin);
                if (el) { const double tv = fmax(bp, bp - hv), sv = hv + tv; TOb[it] = tv; SO[it] = sv; ZO[it] = fmin(mu / sv, 0.5 * rho); }
                else { double sv = fmax(hv, bp); SO[it] = sv; ZO[it] = mu / sv; }
            } else { SO[it] = 1.0; ZO[it] = 0.0; }
        }
        for (int e = tid; e < N * NU; e += TPB) {
            int c = e % NU;
            double lo = lbu(c);
            double sl = fmax(U[e] - lo, 1e-12), su = fmax(-lo - U[e], 1e-12);
            SUL[e] = sl; SUU[e] = su; ZUL[e] = mu / sl; ZUU[e] = mu / su;
        }
        for (int e = tid; e < N1 * NXB; e += TPB) {
            int k = e / NXB, s = e - k * NXB;
            if (k >= 1) {
                double v = X[k * NX + bst(s)], b = bvl(s), sl = fmax(v + b, bp), su = fmax(b - v, bp);
                ZXL[e] = mu / sl; ZXU[e] = mu / su;
            } else { ZXL[e] = 0.0; ZXU[e] = 0.0; }
        }
        __syncthreads();
    };
    double f, lgs, th0, e_c, e_h;
    double delta_last, nu_pen, kkt = INFINITY;
    bool need_shift, restarting = false, cold = false;
    int n_cold = 0, it_base = 0;      // cold-start retries done / iteration at which the current attempt started (NMPC_COLD_RETRY_ITERS)
    int n_tiny = 0, n_restart = 0;      // consecutive iterations with a step length below 1e-10 (stall -> restart, then NMPC_STATUS_STALLED)
    double mh0 = 0.0, mh1 = 0.0, mh2 = 0.0, mh_mu = -1.0, mh_nu = -1.0;   // merit values of the last three iterates (same mu, nu)
    int mcount;
    int iter = 0, status = NMPC_STATUS_MAX_ITER;
    const double n_ineq = (double)P.n_ineq;

    // (Re)start of the barrier iteration.  The first pass is the start of the solve; a later pass is the barrier restart after a
    // stall (restoration in miniature, see the oracle): the primal point goes back strictly inside the simple bounds (a control
    // sitting on its bound would restart with a slack of ~1e-6 and a dual of mu / 1e-6), slacks return onto the constraint
    // values, duals = mu / s, multipliers = 0.  One call site for the initialisation keeps the register budget of the main loop.
    for (;;) {
    if (restarting) {
        if (cold) {       // the reference's cold start (C6:398-400): X_k = x0, U = 0
            for (int e = tid + NX; e < N1 * NX; e += TPB) X[e] = X[e % NX];
            for (int e = tid; e < N * NU; e += TPB) U[e] = 0.0;
            cold = false;
            __syncthreads();
        }
        for (int e = tid + NX; e < N1 * NX; e += TPB) {
            const int d = (e % NX) % 3;
            if (d < 2 || THB) {
                double b = (d == 2) ? P.thmax : P.xymax, px = fmin(bp * fmax(1.0, b), bp * 2.0 * b);
                X[e] = fmin(fmax(X[e], -b + px), b - px);
            }
            LAM[e] = 0.0;
        }
        for (int e = tid; e < N * NU; e += TPB) {
            double lo = lbu(e % NU), hi = -lo, pu = fmin(bp * fmax(1.0, fabs(lo)), bp * (hi - lo));
            U[e] = fmin(fmax(U[e], lo + pu), hi - pu);
        }
        __syncthreads();
    }
    trig();
    __syncthreads();
    init_barrier();
    merit(0.0, f, lgs, th0, e_c, e_h);
    delta_last = 0.0; nu_pen = 1.0; need_shift = false; mcount = 0; restarting = false;
    PROF_T(0);

    for (;;) {
        // ============ A. optimality error (IPOPT eq. 5): stationarity per (stage, robot), complementarity per slot
        double e_d = 0.0, lsum = 0.0, zsum = 0.0, szmax = 0.0, szmin = INFINITY;
#ifdef NMPC_PROFILE
        double dbg_code = -1.0, dbg_v[4] = {0, 0, 0, 0};
#endif
        for (int it = tid; it < N * M_; it += TPB) {
            int k = it / M_, i = it - k * M_, kk = k + 1;
            const double *x = X + kk * NX;
            double r0 = LAM[kk * NX + 3 * i], r1 = LAM[kk * NX + 3 * i + 1], r2 = LAM[kk * NX + 3 * i + 2];
            lsum += fabs(r0) + fabs(r1) + fabs(r2);
            if (kk < N) {
                const double *ln = LAM + (kk + 1) * NX + 3 * i;
                double v = U[kk * NU + 2 * i];
                double a = -T * v * SN[kk * M_ + i], b = T * v * CS[kk * M_ + i];
                r0 += 2 * P.q[0] * (x[3 * i] - XS[3 * i]) - ln[0];
                r1 += 2 * P.q[1] * (x[3 * i + 1] - XS[3 * i + 1]) - ln[1];
                r2 += 2 * P.q[2] * (x[3 * i + 2] - XS[3 * i + 2]) - (ln[2] + a * ln[0] + b * ln[1]);
            }
            // Jx^T z : bounds, pairs (ascending partner), obstacles
            double j0, j1, j2 = 0.0;
            {
                const int sb = kk * NXB + (2 + THB) * i;
                double zl0 = ZXL[sb], zu0 = ZXU[sb], zl1 = ZXL[sb + 1], zu1 = ZXU[sb + 1];
                j0 = zl0 - zu0; j1 = zl1 - zu1;
                zsum += zl0 + zu0 + zl1 + zu1;
                double p0 = (x[3 * i] + P.xymax) * zl0, p1 = (P.xymax - x[3 * i]) * zu0, p2 = (x[3 * i + 1] + P.xymax) * zl1, p3 = (P.xymax - x[3 * i + 1]) * zu1;
                szmax = fmax(szmax, fmax(fmax(p0, p1), fmax(p2, p3))); szmin = fmin(szmin, fmin(fmin(p0, p1), fmin(p2, p3)));
                if (THB) {
                    double zl2 = ZXL[sb + 2], zu2 = ZXU[sb + 2];
                    j2 = zl2 - zu2; zsum += zl2 + zu2;
                    double p4 = (x[3 * i + 2] + P.thmax) * zl2, p5 = (P.thmax - x[3 * i + 2]) * zu2;
                    szmax = fmax(szmax, fmax(p4, p5)); szmin = fmin(szmin, fmin(p4, p5));
                }
            }
            if (kk <= N - 1) {
                const double xi = x[3 * i], yi = x[3 * i + 1];
#pragma unroll 1
                for (int j = 0; j < (prs ? M_ : 0); j++) {
                    if (j == i) continue;
                    int q = (i < j) ? pidx<M_>(i, j) : pidx<M_>(j, i);
                    double z = ZPp[kk * NP + q];
                    j0 += 2 * (xi - x[3 * j]) * z; j1 += 2 * (yi - x[3 * j + 1]) * z;
                }
                for (int o = 0; o < K; o++) {
                    double dx = xi - P.obs[3 * o], dy = yi - P.obs[3 * o + 1], rr = r_obs(dx, dy), z = ZO[kk * MK + i * K + o];
                    j0 += dx / rr * z; j1 += dy / rr * z;
                }
            }
#ifdef NMPC_PROFILE
            { double m3 = fmax(fabs(r0 - j0), fmax(fabs(r1 - j1), fabs(r2 - j2)));
              if (m3 > e_d) { dbg_code = 1000.0 * kk + 10.0 * i + (fabs(r0 - j0) == m3 ? 0 : (fabs(r1 - j1) == m3 ? 1 : 2)); dbg_v[0] = r0 - j0; dbg_v[1] = r1 - j1; dbg_v[2] = r2 - j2; dbg_v[3] = r2; } }
#endif
            e_d = fmax(e_d, fmax(fabs(r0 - j0), fmax(fabs(r1 - j1), fabs(r2 - j2))));
            // control rows of stage k
            const double *ln = LAM + (k + 1) * NX + 3 * i, *u = U + k * NU + 2 * i;
            double c = CS[it], s = SN[it];
            double zl0 = ZUL[k * NU + 2 * i], zu0 = ZUU[k * NU + 2 * i], zl1 = ZUL[k * NU + 2 * i + 1], zu1 = ZUU[k * NU + 2 * i + 1];
            double rv = 2 * P.r[0] * u[0] - T * (c * ln[0] + s * ln[1]) - (zl0 - zu0);
            double rw = 2 * P.r[1] * u[1] - T * ln[2] - (zl1 - zu1);
#ifdef NMPC_PROFILE
            if (fmax(fabs(rv), fabs(rw)) > e_d) { dbg_code = 1000.0 * k + 10.0 * i + (fabs(rv) > fabs(rw) ? 5 : 6); dbg_v[0] = rv; dbg_v[1] = rw; dbg_v[2] = zl0 - zu0; dbg_v[3] = zl1 - zu1; }
#endif
            e_d = fmax(e_d, fmax(fabs(rv), fabs(rw)));
            zsum += zl0 + zu0 + zl1 + zu1;
            double p0 = SUL[k * NU + 2 * i] * zl0, p1 = SUU[k * NU + 2 * i] * zu0, p2 = SUL[k * NU + 2 * i + 1] * zl1, p3 = SUU[k * NU + 2 * i + 1] * zu1;
            szmax = fmax(szmax, fmax(fmax(p0, p1), fmax(p2, p3))); szmin = fmin(szmin, fmin(fmin(p0, p1), fmin(p2, p3)));
        }
        for (int it = tid; it < (N - 1) * NPA; it += TPB) {
            double zv = ZPp[NP + it], pz = SPp[NP + it] * zv;
            zsum += zv; szmax = fmax(szmax, pz); szmin = fmin(szmin, pz);
            if (el) { const double pt = TPp[NP + it] * (rho - zv); szmax = fmax(szmax, pt); szmin = fmin(szmin, pt); }
        }
        for (int it = tid; it < (N - 1) * MK; it += TPB) {
            double zv = ZO[MK + it], pz = SO[MK + it] * zv;
            zsum += zv; szmax = fmax(szmax, pz); szmin = fmin(szmin, pz);
            if (el) { const double pt = TOb[MK + it] * (rho - zv); szmax = fmax(szmax, pt); szmin = fmin(szmin, pt); }
        }
#ifdef NMPC_PROFILE
        { double my = e_d; double gm = wmax<TPB>(e_d, RED);
          if (prof_out && (int)inst == P.trace_inst && iter < 2040 && my == gm) {
              double *tr = reinterpret_cast<double *>(prof_out + 12 + 16 * 2048) + (size_t)iter * 8;
              tr[0] = dbg_code; tr[1] = dbg_v[0]; tr[2] = dbg_v[1]; tr[3] = dbg_v[2]; tr[4] = dbg_v[3]; } }
#endif
        e_d = wmax<TPB>(e_d, RED); lsum = wsum<TPB>(lsum, RED); zsum = wsum<TPB>(zsum, RED);
        szmax = wmax<TPB>(szmax, RED); szmin = wmin<TPB>(szmin, RED);
        const double smax = 100.0;
        double s_d = fmax(smax, (lsum + zsum) / ((double)(N * NX) + n_ineq)) / smax;
        double s_c = fmax(smax, zsum / fmax(n_ineq, 1.0)) / smax;
        double E0 = fmax(fmax(e_d / s_d, e_c), fmax(e_h, szmax / s_c));
        kkt = E0;
        auto cold_retry = [&]() { cold = true; n_cold++; it_base = iter; restarting = true; el = n_cold >= 2; mu = P.mu_init; n_tiny = 0; n_restart = 0; };      // the second restart of last resort is the elastic phase
        if (!(E0 == E0)) { if (n_cold < NMPC_COLD_RETRIES && iter < P.max_iter) cold_retry(); else status = NMPC_STATUS_NUMERIC; break; }
        if (E0 <= P.tol) {
            status = NMPC_STATUS_CONVERGED;
            if (el) {       // the penalty problem's solution solves the NLP only if every elastic variable has closed
                double tmax = 0.0;
                for (int it = tid; it < (N - 1) * NPA; it += TPB) tmax = fmax(tmax, TPp[NP + it]);
                for (int it = tid; it < (N - 1) * MK; it += TPB) tmax = fmax(tmax, TOb[MK + it]);
                if (wmax<TPB>(tmax, RED) > NMPC_X0_TOL) status = NMPC_STATUS_STALLED;
            }
            break;
        }
        if (iter >= P.max_iter) { status = NMPC_STATUS_MAX_ITER; break; }
        if (n_cold < NMPC_COLD_RETRIES && iter - it_base >= NMPC_COLD_RETRY_ITERS) { cold_retry(); break; }
        const double mu_min = P.tol / 10.0;
        for (;;) {
            double cm = fmax(fabs(szmax - mu), fabs(szmin - mu));
            double Emu = fmax(fmax(e_d / s_d, e_c), fmax(e_h, cm / s_c));
            if (mu > mu_min && Emu <= 10.0 * mu) mu = fmax(mu_min, fmin(0.2 * mu, pow(mu, 1.5)));
            else break;
        }
        const double tau = fmax(0.99, 1.0 - mu);
        PROF_T(1);

        // ============ B0. stage packs: everything of stage k that does not depend on the cost-to-go
        // (a) per (stage, robot): x-gradient, diagonal additions, cross terms, coefficients, defects
        for (int it = tid; it < N1 * M_; it += TPB) {
            int k = it / M_, i = it - k * M_;
            const double *x = X + k * NX;
            double *pk = gpack + (size_t)k * G::PACK;          // k == N: terminal block reuses the pack layout
            double g0 = 0.0, g1 = 0.0, g2 = 0.0, h0 = 0.0, h1 = 0.0, h2 = 0.0, hxy = 0.0;
            if (k >= 1) {
                if (k < N) {
                    g0 = 2 * P.q[0] * (x[3 * i] - XS[3 * i]); g1 = 2 * P.q[1] * (x[3 * i + 1] - XS[3 * i + 1]); g2 = 2 * P.q[2] * (x[3 * i + 2] - XS[3 * i + 2]);
                    h0 = 2 * P.q[0]; h1 = 2 * P.q[1]; h2 = 2 * P.q[2];
                }
                const int sb = k * NXB + (2 + THB) * i;
                {   // bounds: v = mu/s - sigma (h - s), sigma = z/s, lower row gradient +1, upper row -1
                    auto bv = [&](double sv, double zv, double hv, double &hd) { double sg = zv / sv; hd += sg; return mu / sv - sg * (hv - sv); };
                    g0 -= bv(x[3 * i] + P.xymax, ZXL[sb], x[3 * i] + P.xymax, h0) - bv(P.xymax - x[3 * i], ZXU[sb], P.xymax - x[3 * i], h0);
                    g1 -= bv(x[3 * i + 1] + P.xymax, ZXL[sb + 1], x[3 * i + 1] + P.xymax, h1) - bv(P.xymax - x[3 * i + 1], ZXU[sb + 1], P.xymax - x[3 * i + 1], h1);
                    if (THB) g2 -= bv(x[3 * i + 2] + P.thmax, ZXL[sb + 2], x[3 * i + 2] + P.thmax, h2) - bv(P.thmax - x[3 * i + 2], ZXU[sb + 2], P.thmax - x[3 * i + 2], h2);
                }
                if (k <= N - 1) {
                    const double xi = x[3 * i], yi = x[3 * i + 1];
#pragma unroll 1
                    for (int j = 0; j < (prs ? M_ : 0); j++) {
                        if (j == i) continue;
                        int q = (i < j) ? pidx<M_>(i, j) : pidx<M_>(j, i);
                        double dx = xi - x[3 * j], dy = yi - x[3 * j + 1];
                        double sv = SPp[k * NP + q], zv = ZPp[k * NP + q], sg, v;
                        if (el) el_sigma(mu, rho, sv, zv, TPp[k * NP + q], h_pair(dx, dy, P.dmin2), sg, v);
                        else { sg = zv / sv; v = mu / sv - sg * (h_pair(dx, dy, P.dmin2) - sv); }
                        g0 -= 2 * dx * v; g1 -= 2 * dy * v;
                        h0 += 4 * sg * dx * dx - 2 * zv; hxy += 4 * sg * dx * dy; h1 += 4 * sg * dy * dy - 2 * zv;
                    }
                    for (int o = 0; o < K; o++) {
                        double dx = xi - P.obs[3 * o], dy = yi - P.obs[3 * o + 1], rr = r_obs(dx, dy), n0 = dx / rr, n1 = dy / rr;
                        double sv = SO[k * MK + i * K + o], zv = ZO[k * MK + i * K + o], sg, v, zz = zv / rr;
                        if (el) el_sigma(mu, rho, sv, zv, TOb[k * MK + i * K + o], h_obs(rr, P.robdim, P.obs[3 * o + 2], P.margin), sg, v);
                        else { sg = zv / sv; v = mu / sv - sg * (h_obs(rr, P.robdim, P.obs[3 * o + 2], P.margin) - sv); }
                        g0 -= n0 * v; g1 -= n1 * v;
                        h0 += sg * n0 * n0 - zz * (1 - n0 * n0); hxy += sg * n0 * n1 + zz * n0 * n1; h1 += sg * n1 * n1 - zz * (1 - n1 * n1);
                    }
                }
            }
            double hvt = 0.0;
            if (k < N) {
                const double *u = U + k * NU + 2 * i, *ln = LAM + (k + 1) * NX + 3 * i;
                double c = CS[it], s = SN[it];
                if (k >= 1) h2 += T * u[0] * (ln[0] * c + ln[1] * s);
                hvt = T * (ln[0] * s - ln[1] * c);
                // control gradient / diagonal
#pragma unroll
                for (int d = 0; d < 2; d++) {
                    const int eu = k * NU + 2 * i + d;
                    double lo = d ? -P.wmax : -P.vmax, sl = SUL[eu], su = SUU[eu], zl = ZUL[eu], zu = ZUU[eu];
                    double vl = mu / sl - zl / sl * ((u[d] - lo) - sl), vu = mu / su - zu / su * ((-lo - u[d]) - su);
                    pk[G::PK_G + 2 * i + d] = 2 * P.r[d] * u[d] - (vl - vu);
                    pk[G::PK_HD + 2 * i + d] = 2 * P.r[d] + zl / sl + zu / su;
                }
                {   // rows/columns of [B A] belonging to robot i: v_i, omega_i, x_i, y_i, theta_i
                    double *cf = pk + G::PK_CF;
                    cf[3 * (2 * i)] = T * c; cf[3 * (2 * i) + 1] = T * s; cf[3 * (2 * i) + 2] = 0.0;
                    cf[3 * (2 * i + 1)] = T; cf[3 * (2 * i + 1) + 1] = 0.0; cf[3 * (2 * i + 1) + 2] = 0.0;
                    cf[3 * (NU + 3 * i)] = 1.0; cf[3 * (NU + 3 * i) + 1] = 0.0; cf[3 * (NU + 3 * i) + 2] = 0.0;
                    cf[3 * (NU + 3 * i + 1)] = 1.0; cf[3 * (NU + 3 * i + 1) + 1] = 0.0; cf[3 * (NU + 3 * i + 1) + 2] = 0.0;
                    cf[3 * (NU + 3 * i + 2)] = 1.0; cf[3 * (NU + 3 * i + 2) + 1] = -T * u[0] * s; cf[3 * (NU + 3 * i + 2) + 2] = T * u[0] * c;
                    if (i == 0) pk[G::PK_ZERO] = 0.0;
                }
                const double *xn = x + NX;
                pk[G::PK_C + 3 * i] = defect_xy(xn[3 * i], x[3 * i], T * u[0], c);
                pk[G::PK_C + 3 * i + 1] = defect_xy(xn[3 * i + 1], x[3 * i + 1], T * u[0], s);
                pk[G::PK_C + 3 * i + 2] = defect_th(xn[3 * i + 2], x[3 * i + 2], T, u[1]);
                pk[G::PK_HVT + i] = hvt;
            }
            pk[G::PK_G + NU + 3 * i] = g0; pk[G::PK_G + NU + 3 * i + 1] = g1; pk[G::PK_G + NU + 3 * i + 2] = g2;
            pk[G::PK_HD + NU + 3 * i] = h0; pk[G::PK_HD + NU + 3 * i + 1] = h1; pk[G::PK_HD + NU + 3 * i + 2] = h2;
            pk[G::PK_HXY + i] = hxy;
        }
        // (b) pair blocks E_ij = 4 sigma dp dp^T - 2 z I per (stage, pair)
        for (int it = tid; it < (N - 1) * NPA; it += TPB) {
            int k = 1 + it / NPd, q = it - (k - 1) * NP, i, j;
            pair_ij(q, i, j);
            double dx = X[k * NX + 3 * i] - X[k * NX + 3 * j], dy = X[k * NX + 3 * i + 1] - X[k * NX + 3 * j + 1];
            double zz = ZPp[k * NP + q], sg;
            if (el) { double v_; el_sigma(mu, rho, SPp[k * NP + q], zz, TPp[k * NP + q], h_pair(dx, dy, P.dmin2), sg, v_); }
            else sg = zz / SPp[k * NP + q];
            double *pk = gpack + (size_t)k * G::PACK + G::PK_E + 3 * q;
            pk[0] = -(4 * sg * dx * dx - 2 * zz); pk[1] = -(4 * sg * dx * dy); pk[2] = -(4 * sg * dy * dy - 2 * zz);
        }
        for (int q = tid; q < 3 * NP; q += TPB) gpack[G::PK_E + q] = 0.0;     // stage 0 carries no pair rows
        if (!prs)                                                              // no pair rows at all: the E slots of every stage are zero
            for (int e = tid; e < (N - 1) * 3 * NP; e += TPB) gpack[(size_t)(1 + e / (3 * NPd)) * G::PACK + G::PK_E + e % (3 * NPd)] = 0.0;
        __syncthreads();
        PROF_T(2);

        // ============ B. Riccati sweep with inertia correction (IPOPT alg. IC)
        // first trial: delta = 0, except right after an iteration that needed a shift (then a quarter of that shift directly)
        // Partial re-factorisation (include/nmpc_constants.h): [P | p] entering every NMPC_CKPT_EVERY-th stage is saved in the workspace; a rejected
        // pivot at stage kf escalates the shift and resumes at the saved stage NMPC_RESUME_STAGE(kf, N), the stages above keep their factors.
        double delta = need_shift ? fmax(1e-20, 0.25 * delta_last) : 0.0;
        int ntry = 0;
        bool ok;
        int kst = N - 1;
        for (;;) {
            ok = true;
            int kf = 0;
            // terminal cost-to-go P_N = diag(hd_N), p_N = g_N — or the cost-to-go saved on entry to stage kst (Pf and PV are contiguous)
            if (kst == N - 1) {
                const double *pkN = gpack + (size_t)N * G::PACK;
                for (int e = tid; e < NX * NX; e += TPB) {
                    int r = e / NX, c = e - r * NX;
                    Pf[e] = (r == c) ? pkN[G::PK_HD + NU + r] : 0.0;
                }
                for (int r = tid; r < NX; r += TPB) PV[r] = pkN[G::PK_G + NU + r];
            } else {
                const double *ck = gck + (size_t)((N - 1 - kst) / NMPC_CKPT_EVERY) * ((NX + 1) * 64);
                for (int e = tid; e < NX * NX + NX; e += TPB) Pf[e] = ck[e];
            }
            // ---- per-element LDS byte offsets (relative to sm): pivot-row entries UR[.][a], UR[.][c]; the three G entries,
            //      the coefficient triple and the Hessian addition of the assembly; the mirror positions of the Schur block
            int ea[NTP], uoa[NTP], uoc[NTP], pg0[NTP], pg1[NTP], pg2[NTP], pca[NTP], pho[NTP], wa1[NTP], wa2[NTP];
            double dl[NTP];
        #pragma unroll
            for (int t = 0; t < NTP; t++) {
                // branch-free decode (selects only): this runs at the start of every sweep
                const int w = ACT[t * TPB + tid];
                const bool valid = w != 0xFFFF;
                const int a = valid ? (w & 0xFF) : 0, c = valid ? (w >> 8) : 0;
                const bool au = a < NU, odd = (a & 1) != 0, rhs = c == NZ;
                const int iu = a >> 1;                                          // robot of a control row
                const int sa = au ? 0 : a - NU, sc = (c < NU || rhs) ? 0 : c - NU;   // state indices of row / column (0 when not a state)
                const int ia = sa / 3, da = sa - 3 * ia, ic = sc / 3, dc = sc - 3 * ic;
                // state index of term t of row/column a of [B A] (term_ix)
                const int ix0 = au ? (odd ? 3 * iu + 2 : 3 * iu) : sa;
                const int ix1 = au ? (odd ? 3 * iu + 2 : 3 * iu + 1) : (da < 2 ? sa : 3 * ia);
                const int ix2 = au ? (odd ? 3 * iu + 2 : 3 * iu) : (da < 2 ? sa : 3 * ia + 1);
                ea[t] = valid ? a : -1;
                uoa[t] = oUR + 8 * a; uoc[t] = oUR + 8 * c;
                pg0[t] = oGb + 8 * (ix0 * G::LDG + c); pg1[t] = oGb + 8 * (ix1 * G::LDG + c); pg2[t] = oGb + 8 * (ix2 * G::LDG + c);
                pca[t] = oPK + 8 * (G::PK_CF + 3 * a);
                dl[t] = (valid && c == a && au) ? 1.0 : 0.0;      // the inertia shift delta acts on the control diagonal only
                // Hessian addition: lowest priority first, later selects override
                int h = G::PK_ZERO;
                h = (!au && !rhs && c >= NU && da < 2 && dc < 2) ? ((ia == ic) ? (G::PK_HXY + ia) : (G::PK_E + 3 * pidx<M_>(ia, ic) + da + dc)) : h;
                h = (au && !odd && c == NU + 3 * iu + 2) ? (G::PK_HVT + iu) : h;
                h = (c == a) ? (G::PK_HD + a) : h;
                h = rhs ? (G::PK_G + a) : h;
                pho[t] = oPK + 8 * (valid ? h : G::PK_ZERO);
                // mirror positions of the Schur block [P | p] (state rows only; PV follows Pf)
                const int w1 = au ? 0 : (rhs ? NX * NX + sa : sa * NX + sc), w2 = au ? 0 : (rhs ? NX * NX + sa : sc * NX + sa);
                wa1[t] = oPf + 8 * w1; wa2[t] = oPf + 8 * w2;
            }
            // the pivot-row area doubles as staging / residual storage between sweeps: restore the zero lower part the
            // select-free rank-1 update relies on
            for (int e = tid; e < NU * LD; e += TPB) UR[e] = 0.0;
            // prefetch pack N-1 into registers
            constexpr int PKR = (G::PACK + TPB - 1) / TPB;
            double pkr[PKR];
#pragma unroll
            for (int t = 0; t < PKR; t++) { int e = tid + t * TPB; pkr[t] = (e < G::PACK) ? gpack[(size_t)kst * G::PACK + e] : 0.0; }
            lds_sync<TPB>();
            for (int k = kst; k >= 0; k--) {
                if (k < kst && (kst - k) % NMPC_CKPT_EVERY == 0) {      // save [P | p] entering this stage
                    double *ck = gck + (size_t)((N - 1 - k) / NMPC_CKPT_EVERY) * ((NX + 1) * 64);
                    for (int e = tid; e < NX * NX + NX; e += TPB) ck[e] = Pf[e];
                }
                // ---- stage pack -> LDS; prefetch the next one
#pragma unroll
                for (int t = 0; t < PKR; t++) { int e = tid + t * TPB; if (e < G::PACK) PK[e] = pkr[t]; }
                if (k > 0) {
#pragma unroll
                    for (int t = 0; t < PKR; t++) { int e = tid + t * TPB; pkr[t] = (e < G::PACK) ? gpack[(size_t)(k - 1) * G::PACK + e] : 0.0; }
                }
                lds_sync<TPB>();
                // ---- G = P [B A] (<= 3 terms per column, thread = column, rows strided) and its last column p + P b, b = -c_k
                if (gact) {
                    const double c0 = PK[G::PK_CF + 3 * gcol], c1 = PK[G::PK_CF + 3 * gcol + 1], c2 = PK[G::PK_CF + 3 * gcol + 2];
                    double p0[GRPT], p1[GRPT], p2[GRPT];          // all reads first (one LDS latency instead of GRPT)
                    static_for<0, GRPT>([&](auto nc) {
                        constexpr int n = decltype(nc)::value;
                        p0[n] = lds_ld(sm, gpb0, 8 * n * HG * NX); p1[n] = lds_ld(sm, gpb1, 8 * n * HG * NX); p2[n] = lds_ld(sm, gpb2, 8 * n * HG * NX);
                    });
                    static_for<0, GRPT>([&](auto nc) {
                        constexpr int n = decltype(nc)::value;
                        if ((NX % HG == 0) || grow0 + n * HG < NX) lds_st(sm, gwb, 8 * n * HG * G::LDG, c0 * p0[n] + c1 * p1[n] + c2 * p2[n]);
                    });
                }
                // last column p + P b (b = -c_k): the NX dot products are split into PBP column parts over PBP * NX lanes of the
                // first wave and recombined with two lane shuffles (was: NX lanes doing NX serial LDS reads each)
                if (tid < 64) {
                    const int pr_ = tid % NX, part = tid / NX;
                    double a = 0.0;
                    if (part < PBP) {
                        const int pb = oPf + 8 * (pr_ * NX + part * PBC), cb = oPK + 8 * (G::PK_C + part * PBC);
                        double pq[PBC], cq[PBC];
#pragma unroll
                        for (int c = 0; c < PBC; c++) { pq[c] = lds_ld(sm, pb, 8 * c); cq[c] = lds_ld(sm, cb, 8 * c); }
#pragma unroll
                        for (int c = 0; c < PBC; c++) a = fma(-pq[c], cq[c], a);
                    }
                    double ssum = a;
                    if constexpr (PBP >= 2) ssum += __shfl(a, tid + NX);
                    if constexpr (PBP >= 3) ssum += __shfl(a, tid + 2 * NX);
                    if (tid < NX) Gb[tid * G::LDG + NZ] = PV[tid] + ssum;
                }
                lds_sync<TPB>();
                PROF_T(9);
                // ---- my elements of [B A]^T G + H (and the rhs column), branch-free, into registers
                double mv[NTP], d0r[NTP];      // d0r: values as assembled (the owners of the control diagonals test their pivot against it)
                // reads first, in batches of AB elements (one LDS latency per batch instead of one per element; the batch size
                // bounds the registers held by in-flight loads)
                constexpr int AB = 4;
                static_for<0, (NTP + AB - 1) / AB>([&](auto bc) {
                    constexpr int b0 = decltype(bc)::value * AB, b1 = (b0 + AB < NTP) ? b0 + AB : NTP;
                    double q0[AB], q1[AB], q2[AB], r0[AB], r1[AB], r2[AB], hh[AB];
#pragma unroll
                    for (int t = b0; t < b1; t++) {
                        q0[t - b0] = lds_ld(sm, pca[t], 0); q1[t - b0] = lds_ld(sm, pca[t], 8); q2[t - b0] = lds_ld(sm, pca[t], 16);
                        r0[t - b0] = lds_ld(sm, pg0[t], 0); r1[t - b0] = lds_ld(sm, pg1[t], 0); r2[t - b0] = lds_ld(sm, pg2[t], 0);
                        hh[t - b0] = lds_ld(sm, pho[t], 0);
                    }
#pragma unroll
                    for (int t = b0; t < b1; t++) {
                        double v = q0[t - b0] * r0[t - b0] + q1[t - b0] * r1[t - b0] + q2[t - b0] * r2[t - b0];
                        v += hh[t - b0] + dl[t] * delta;
                        mv[t] = v;
                        d0r[t] = v;
                    }
                });
                PROF_T(10);
                // ---- NU pivot steps of symmetric elimination; pivot rows are published through LDS
                // pivot test + reciprocal of pivot j: evaluated by every thread on its own slice element, meaningful on the owner
                // of (j,j); < 0 flags a non-positive pivot.  It is computed right after the slice holding (j,j) got its last
                // update, so its latency hides under the rank-1 updates of the other slices.
                auto pivot_inv = [&](double d, double d0) { return (d > 1e-9 * fabs(d0) && d > 0.0) ? rcp_nr(d) : -1.0; };
                double inv_own = pivot_inv(mv[G::diag_e(0) / TPB], d0r[G::diag_e(0) / TPB]);
                static_for<0, NU>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    constexpr int td = G::diag_e(j) / TPB;
                    if (ok) {
                        // publish pivot row j (only the at most two slices that hold it) and its reciprocal pivot
                        static_for<0, NTP>([&](auto tc) {
                            constexpr int t = decltype(tc)::value;
                            if constexpr (G::slice_lo(t, TPB) <= j && j <= G::slice_hi(t, TPB)) { if (ea[t] == j) lds_st(sm, uoc[t], 8 * j * LD, mv[t]); }
                        });
                        if (ea[td] == j && dl[td] != 0.0) INV[j] = inv_own;
                        lds_sync<TPB>();
                        const double inv = uniform_f64(INV[j]);
                        if (!(inv > 0.0)) ok = false;
                        else {
                            // rank-1 update of every live slice, branch-free and select-free: for rows a <= j the factor
                            // UR[j][a] is 0 (a < j) or annihilates the already published pivot row itself (a == j)
                            if constexpr (j + 1 < NU) {
                                constexpr int tn = G::diag_e(j + 1) / TPB;       // slice of the next pivot: update it first
                                mv[tn] = fma(-(lds_ld(sm, uoa[tn], 8 * j * LD) * inv), lds_ld(sm, uoc[tn], 8 * j * LD), mv[tn]);
                                inv_own = pivot_inv(mv[tn], d0r[tn]);
                                static_for<0, NTP>([&](auto tc) {
                                    constexpr int t = decltype(tc)::value;
                                    if constexpr (G::slice_hi(t, TPB) > j && t != tn)
                                        mv[t] = fma(-(lds_ld(sm, uoa[t], 8 * j * LD) * inv), lds_ld(sm, uoc[t], 8 * j * LD), mv[t]);
                                });
                            } else {
                                static_for<0, NTP>([&](auto tc) {
                                    constexpr int t = decltype(tc)::value;
                                    if constexpr (G::slice_hi(t, TPB) > j)
                                        mv[t] = fma(-(lds_ld(sm, uoa[t], 8 * j * LD) * inv), lds_ld(sm, uoc[t], 8 * j * LD), mv[t]);
                                });
                            }
                        }
                    }
                });
                PROF_T(11);
                if (!ok) { kf = k; break; }
                // ---- what is left is [P_k | p_k]: back to LDS (both triangles)
                if (k >= 1) {
#pragma unroll
                    for (int t = 0; t < NTP; t++) if (ea[t] >= NU) { lds_st(sm, wa1[t], 0, mv[t]); lds_st(sm, wa2[t], 0, mv[t]); }
                }
                lds_sync<TPB>();
                // ---- stream the pivot rows and reciprocal pivots of this stage to HBM/L2 (coalesced).  No feedback gains are
                //      formed: the forward sweep needs K dx + k for ONE dx only, i.e. one triangular solve per stage.
                for (int e = tid; e < NU * LD + NU; e += TPB) gkt[(size_t)k * G::KTS + e] = UR[e];
            }
            if (ok) break;
            __syncthreads();
            ntry++;
            if (delta == 0.0) delta = (delta_last == 0.0) ? 1e-4 : fmax(1e-20, delta_last / 3.0);
            else delta *= (delta_last == 0.0) ? 100.0 : NMPC_SHIFT_ESCALATION;
            if (delta > 1e20) break;
            kst = NMPC_RESUME_STAGE(kf, N);
        }
        if (!ok) { if (n_cold < NMPC_COLD_RETRIES) { cold_retry(); iter++; } else status = NMPC_STATUS_NUMERIC; break; }
        if (delta > 0.0) delta_last = delta;
        need_shift = delta > 0.0 && (ntry > 0 || delta > 1e-6);
        __syncthreads();   // s_waitcnt vmcnt(0): the stage-0 gains were stored a moment ago by other lanes of this wave
        PROF_T(3);

        // ============ C. forward sweep.  The stage factors stream back from HBM/L2 through a register ring filled PD stages
        // ahead by all threads (coalesced), then through an LDS staging buffer (the pivot-row area, free here).  Per stage the control
        // lanes form t = Uux dx + rhs and solve Uuu du = -t by back substitution with v_readlane broadcasts.
        {
            constexpr int KPT = (G::KTS + TPB - 1) / TPB, PD = 4;
            double kq[PD][KPT];
#pragma unroll
            for (int d = 0; d < PD; d++)
#pragma unroll
                for (int t = 0; t < KPT; t++) { int e = tid + t * TPB; kq[d][t] = (d < N && e < G::KTS) ? gkt[(size_t)d * G::KTS + e] : 0.0; }
            for (int c = tid; c < NX; c += TPB) DX[c] = 0.0;
            const int jrow = (tid < NU) ? tid : 0;
            for (int k0 = 0; k0 < N; k0 += PD) {
#pragma unroll
                for (int d = 0; d < PD; d++) {
                    const int k = k0 + d;
                    if (k < N) {
#pragma unroll
                        for (int t = 0; t < KPT; t++) { int e = tid + t * TPB; if (e < G::KTS) FS[e] = kq[d][t]; }
                        if (k + PD < N) {
#pragma unroll
                            for (int t = 0; t < KPT; t++) { int e = tid + t * TPB; if (e < G::KTS) kq[d][t] = gkt[(size_t)(k + PD) * G::KTS + e]; }
                        }
                        lds_sync<TPB>();
                        if (tid < 64) {      // the NU control lanes live in the first wave
                            const double *row = FS + jrow * LD;
                            double t0 = row[NZ], t1 = 0.0, t2 = 0.0;
#pragma unroll
                            for (int c = 0; c + 2 < NX; c += 3) {
                                t0 = fma(row[NU + c], DX[k * NX + c], t0);
                                t1 = fma(row[NU + c + 1], DX[k * NX + c + 1], t1);
                                t2 = fma(row[NU + c + 2], DX[k * NX + c + 2], t2);
                            }
                            double tj = t0 + (t1 + t2);
                            const double invj = FS[NU * LD + jrow];
                            double urow[NU];
#pragma unroll
                            for (int c = 0; c < NU; c++) urow[c] = row[c];
                            double duj = 0.0;
                            static_for<0, NU>([&](auto cc) {
                                constexpr int c = NU - 1 - decltype(cc)::value;
                                const double cand = -tj * invj;            // meaningful on lane c: all its later columns are in
                                const double duc = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(cand), c),
                                                                    __builtin_amdgcn_readlane(__double2loint(cand), c));
                                if (tid == c) duj = duc;
                                tj = fma(urow[c], duc, tj);                // rows j < c use it; rows j >= c are done (their tj is dead)
                            });
                            if (tid < NU) DU[k * NU + tid] = duj;
                        }
                        lds_sync<TPB>();
                        for (int i = tid; i < M_; i += TPB) {
                            const double *x = X + k * NX + 3 * i, *xn = x + NX, *u = U + k * NU + 2 * i, *dx = DX + k * NX + 3 * i, *du = DU + k * NU + 2 * i;
                            double s = SN[k * M_ + i], c = CS[k * M_ + i];
                            double c0 = defect_xy(xn[0], x[0], T * u[0], c), c1 = defect_xy(xn[1], x[1], T * u[0], s), c2 = defect_th(xn[2], x[2], T, u[1]);
                            double *dn = DX + (k + 1) * NX + 3 * i;
                            dn[0] = dx[0] + (-T * u[0] * s) * dx[2] + T * c * du[0] - c0;
                            dn[1] = dx[1] + (T * u[0] * c) * dx[2] + T * s * du[0] - c1;
                            dn[2] = dx[2] + T * du[1] - c2;
                        }
                    }
                }
            }
            __syncthreads();
        }
        PROF_T(4);

        // ============ D. fraction to the boundary (IPOPT eq. 15) over all inequality slots (ds, dz recomputed)
        double a_p = 1.0, a_d = 1.0, mult_max = 0.0;   // mult_max: inf-norm of the QP multipliers of the rows that enter theta
        auto fb = [&](double sv, double zv, double ds) -> double {
            double dz = dz_of(mu, sv, zv, ds);
            if (ds < 0.0) a_p = fmin(a_p, -tau * sv / ds);
            if (dz < 0.0) a_d = fmin(a_d, -tau * zv / dz);
            return zv + dz;
        };
        double dphi_el = 0.0;       // elastic phase: the rows' part of the merit derivative, -mu (ds/s + dt/t) + rho dt
        auto fbe = [&](double sv, double zv, double tv, double hv, double jd) -> double {
            double ds, dz, dt;
            el_step(mu, rho, sv, zv, tv, hv, jd, ds, dz, dt);
            if (ds < 0.0) a_p = fmin(a_p, -tau * sv / ds);
            if (dt < 0.0) a_p = fmin(a_p, -tau * tv / dt);
            if (dz < 0.0) a_d = fmin(a_d, -tau * zv / dz);
            if (dz > 0.0) a_d = fmin(a_d, tau * (rho - zv) / dz);
            dphi_el += rho * dt - mu * (ds / sv + dt / tv);
            return zv + dz;
        };
        for (int e = tid; e < N * NU; e += TPB) {
            int c = e % NU;
            double lo = lbu(c), u = U[e], du = DU[e], sl = SUL[e], su = SUU[e];
            fb(sl, ZUL[e], du + ((u - lo) - sl)); fb(su, ZUU[e], -du + ((-lo - u) - su));
        }
        for (int e = tid; e < N * NXB; e += TPB) {
            int k = 1 + e / NXB, s = e - (k - 1) * NXB;
            double v = X[k * NX + bst(s)], dv = DX[k * NX + bst(s)], b = bvl(s), sl = v + b, su = b - v;
            fb(sl, ZXL[k * NXB + s], dv + ((v + b) - sl)); fb(su, ZXU[k * NXB + s], -dv + ((b - v) - su));
        }
        for (int it = tid; it < (N - 1) * NPA; it += TPB) {
            int k = 1 + it / NPd, q = it - (k - 1) * NP, i, j;
            pair_ij(q, i, j);
            const int oi = k * NX + 3 * i, oj = k * NX + 3 * j;
            double ex = X[oi] - X[oj], ey = X[oi + 1] - X[oj + 1], sv = SPp[k * NP + q];
            if (el) { mult_max = fmax(mult_max, fabs(fbe(sv, ZPp[k * NP + q], TPp[k * NP + q], h_pair(ex, ey, P.dmin2), jd_pair(ex, ey, DX[oi] - DX[oj], DX[oi + 1] - DX[oj + 1])))); continue; }
            double ds = ds_pair(ex, ey, DX[oi] - DX[oj], DX[oi + 1] - DX[oj + 1], P.dmin2, sv);
            mult_max = fmax(mult_max, fabs(fb(sv, ZPp[k * NP + q], ds)));
        }
        for (int it = tid; it < (N - 1) * MK; it += TPB) {
            int k = 1 + it / MK, e = it - (k - 1) * MK, i = e / K, o = e - i * K;
            const int oi = k * NX + 3 * i;
            double ex = X[oi] - P.obs[3 * o], ey = X[oi + 1] - P.obs[3 * o + 1], rr = r_obs(ex, ey), sv = SO[k * MK + e];
            if (el) { mult_max = fmax(mult_max, fabs(fbe(sv, ZO[k * MK + e], TOb[k * MK + e], h_obs(rr, P.robdim, P.obs[3 * o + 2], P.margin), jd_obs(ex, ey, rr, DX[oi], DX[oi + 1])))); continue; }
            double ds = ds_obs(ex, ey, rr, DX[oi], DX[oi + 1], h_obs(rr, P.robdim, P.obs[3 * o + 2], P.margin), sv);
            mult_max = fmax(mult_max, fabs(fb(sv, ZO[k * MK + e], ds)));
        }
        a_p = wmin<TPB>(a_p, RED); a_d = wmin<TPB>(a_d, RED);
        PROF_T(5);
        // ============ F. multipliers of the QP: stage-parallel residuals, then the robot-local adjoint recursion in registers
        //   lam+_k = A_k^T lam+_{k+1} - (grad f_k + W_k dx_k + W_xu du_k) + Jx_k^T (z + dz)_k
        for (int it = tid; it < N * M_; it += TPB) {
            int k = 1 + it / M_, i = it - (k - 1) * M_;
            const double *x = X + k * NX, *dx = DX + k * NX;
            double l0, l1, l2 = 0.0;
            {   // bounds: z + dz, dz = (mu - s z - z ds)/s, ds = +-dx + (h - s)
                const int sb = k * NXB + (2 + THB) * i;
                auto zn = [&](double sv, double zv, double hv, double jd) { double ds = jd + (hv - sv); return zv + dz_of(mu, sv, zv, ds); };
                l0 = zn(x[3 * i] + P.xymax, ZXL[sb], x[3 * i] + P.xymax, dx[3 * i]) - zn(P.xymax - x[3 * i], ZXU[sb], P.xymax - x[3 * i], -dx[3 * i]);
                l1 = zn(x[3 * i + 1] + P.xymax, ZXL[sb + 1], x[3 * i + 1] + P.xymax, dx[3 * i + 1]) - zn(P.xymax - x[3 * i + 1], ZXU[sb + 1], P.xymax - x[3 * i + 1], -dx[3 * i + 1]);
                if (THB) l2 = zn(x[3 * i + 2] + P.thmax, ZXL[sb + 2], x[3 * i + 2] + P.thmax, dx[3 * i + 2]) - zn(P.thmax - x[3 * i + 2], ZXU[sb + 2], P.thmax - x[3 * i + 2], -dx[3 * i + 2]);
            }
            if (k < N) {
                const double xi = x[3 * i], yi = x[3 * i + 1];
#pragma unroll 1
                for (int j = 0; j < (prs ? M_ : 0); j++) {
                    if (j == i) continue;
                    int q = (i < j) ? pidx<M_>(i, j) : pidx<M_>(j, i);
                    double ex = xi - x[3 * j], ey = yi - x[3 * j + 1];
                    double ddx = dx[3 * i] - dx[3 * j], ddy = dx[3 * i + 1] - dx[3 * j + 1];
                    double sv = SPp[k * NP + q], zv = ZPp[k * NP + q], znew;
                    if (el) { double ds, dz, dt; el_step(mu, rho, sv, zv, TPp[k * NP + q], h_pair(ex, ey, P.dmin2), jd_pair(ex, ey, ddx, ddy), ds, dz, dt); znew = zv + dz; }
                    else { double ds = ds_pair(ex, ey, ddx, ddy, P.dmin2, sv); znew = zv + dz_of(mu, sv, zv, ds); }
                    l0 += 2 * ex * znew + 2 * zv * ddx;      // Jx^T (z+dz)  -  (-2 z (ddx))  [exact-Hessian term of the pair row]
                    l1 += 2 * ey * znew + 2 * zv * ddy;
                }
                for (int o = 0; o < K; o++) {
                    double ex = xi - P.obs[3 * o], ey = yi - P.obs[3 * o + 1], rr = r_obs(ex, ey), n0 = ex / rr, n1 = ey / rr;
                    double sv = SO[k * MK + i * K + o], zv = ZO[k * MK + i * K + o];
                    double nd = n0 * dx[3 * i] + n1 * dx[3 * i + 1];
                    double znew, zz = zv / rr;
                    if (el) { double ds, dz, dt; el_step(mu, rho, sv, zv, TOb[k * MK + i * K + o], h_obs(rr, P.robdim, P.obs[3 * o + 2], P.margin), jd_obs(ex, ey, rr, dx[3 * i], dx[3 * i + 1]), ds, dz, dt); znew = zv + dz; }
                    else { double ds = ds_obs(ex, ey, rr, dx[3 * i], dx[3 * i + 1], h_obs(rr, P.robdim, P.obs[3 * o + 2], P.margin), sv); znew = zv + dz_of(mu, sv, zv, ds); }
                    l0 += n0 * znew + zz * (dx[3 * i] - n0 * nd);
                    l1 += n1 * znew + zz * (dx[3 * i + 1] - n1 * nd);
                }
                l0 -= 2 * P.q[0] * (x[3 * i] - XS[3 * i]) + 2 * P.q[0] * dx[3 * i];
                l1 -= 2 * P.q[1] * (x[3 * i + 1] - XS[3 * i + 1]) + 2 * P.q[1] * dx[3 * i + 1];
                l2 -= 2 * P.q[2] * (x[3 * i + 2] - XS[3 * i + 2]) + 2 * P.q[2] * dx[3 * i + 2];
                const double *ln = LAM + (k + 1) * NX + 3 * i;
                double v = U[k * NU + 2 * i], c = CS[k * M_ + i], s = SN[k * M_ + i];
                double htt = T * v * (ln[0] * c + ln[1] * s), hvt = T * (ln[0] * s - ln[1] * c);
                l2 -= htt * dx[3 * i + 2] + hvt * DU[k * NU + 2 * i];
            }
            RV[k * NX + 3 * i] = l0; RV[k * NX + 3 * i + 1] = l1; RV[k * NX + 3 * i + 2] = l2;
        }
        __syncthreads();
        for (int i = tid; i < M_; i += TPB) {
            double n0 = 0.0, n1 = 0.0, n2 = 0.0;
            for (int k = N; k >= 1; k--) {
                double l0 = RV[k * NX + 3 * i], l1 = RV[k * NX + 3 * i + 1], l2 = RV[k * NX + 3 * i + 2];
                if (k < N) {
                    double v = U[k * NU + 2 * i];
                    double a = -T * v * SN[k * M_ + i], b = T * v * CS[k * M_ + i];
                    l0 += n0; l1 += n1; l2 += n2 + a * n0 + b * n1;
                }
                n0 = l0; n1 = l1; n2 = l2;
                RV[k * NX + 3 * i] = l0; RV[k * NX + 3 * i + 1] = l1; RV[k * NX + 3 * i + 2] = l2;     // lam+_k
                mult_max = fmax(mult_max, fmax(fabs(l0), fmax(fabs(l1), fabs(l2))));
            }
        }
        mult_max = wmax<TPB>(mult_max, RED);
        __syncthreads();
        PROF_T(7);


        // ============ E. l1 merit backtracking line search; (f, sum log s, theta) of the current point are carried
        double dphi = 0.0;
        for (int it = tid; it < N1 * M_; it += TPB) {
            int k = it / M_, i = it - k * M_;
            if (k >= 1 && k < N) {
#pragma unroll
                for (int d = 0; d < 3; d++) dphi += 2 * P.q[d] * (X[k * NX + 3 * i + d] - XS[3 * i + d]) * DX[k * NX + 3 * i + d];
            }
            if (k < N) {
#pragma unroll
                for (int d = 0; d < 2; d++) {
                    const int eu = k * NU + 2 * i + d;
                    double u = U[eu], du = DU[eu], lo = d ? -P.wmax : -P.vmax, sl = SUL[eu], su = SUU[eu];
                    dphi += 2 * P.r[d] * u * du - mu * ((du + ((u - lo) - sl)) / sl + (-du + ((-lo - u) - su)) / su);
                }
            }
            if (k >= 1) {
#pragma unroll
                for (int d = 0; d < 2 + THB; d++) {
                    const int sb = k * NXB + (2 + THB) * i + d;
                    double v = X[k * NX + 3 * i + d], dv = DX[k * NX + 3 * i + d], b = (d == 2) ? P.thmax : P.xymax, sl = v + b, su = b - v;
                    dphi -= mu * ((dv + ((v + b) - sl)) / sl + (-dv + ((b - v) - su)) / su);
                }
            }
        }
        for (int it = tid; it < (N - 1) * NPA; it += TPB) {
            int k = 1 + it / NPd, q = it - (k - 1) * NP, i, j;
            pair_ij(q, i, j);
            const int oi = k * NX + 3 * i, oj = k * NX + 3 * j;
            double ex = X[oi] - X[oj], ey = X[oi + 1] - X[oj + 1], sv = SPp[k * NP + q];
            if (el) continue;      // elastic rows: summed by the step-length pass (dphi_el)
            double ds = ds_pair(ex, ey, DX[oi] - DX[oj], DX[oi + 1] - DX[oj + 1], P.dmin2, sv);
            dphi -= mu * ds / sv;
        }
        for (int it = tid; it < (N - 1) * MK; it += TPB) {
            int k = 1 + it / MK, e = it - (k - 1) * MK, i = e / K, o = e - i * K;
            const int oi = k * NX + 3 * i;
            double ex = X[oi] - P.obs[3 * o], ey = X[oi + 1] - P.obs[3 * o + 1], rr = r_obs(ex, ey), sv = SO[k * MK + e];
            if (el) continue;
            double ds = ds_obs(ex, ey, rr, DX[oi], DX[oi + 1], h_obs(rr, P.robdim, P.obs[3 * o + 2], P.margin), sv);
            dphi -= mu * ds / sv;
        }
        dphi += dphi_el;
        dphi = wsum<TPB>(dphi, RED);
        const double phi0 = f - mu * lgs;
        if (th0 > 0.0) {
            // Nocedal-Wright (18.36), rho = 0.1; capped by the multiplier norm it cannot exceed in exact arithmetic
            double nut = fmin(dphi / ((1.0 - 0.1) * th0), mult_max / (1.0 - 0.1));
            nu_pen = fmax(1.0, 0.5 * nu_pen);      // the penalty may relax again: one bad step must not cripple the rest of the solve
            if (nu_pen < nut) nu_pen = nut + 1.0;
        }
        const double Dm = dphi - nu_pen * th0;
        double alpha = a_p, ft, lgt, tht, ect, eht;
        // non-monotone (Grippo-type) reference value: max of the current and the last three merit values of this barrier problem
        if (mh_mu != mu || mh_nu != nu_pen) { mcount = 0; mh_mu = mu; mh_nu = nu_pen; }
        const double m0 = phi0 + nu_pen * th0;
        double mref = m0;
        if (mcount > 0) mref = fmax(mref, mh0);
        if (mcount > 1) mref = fmax(mref, mh1);
        if (mcount > 2) mref = fmax(mref, mh2);
        mh2 = mh1; mh1 = mh0; mh0 = m0; if (mcount < 3) mcount++;
        for (int ls = 0; ls < 30; ls++) {
            merit(alpha, ft, lgt, tht, ect, eht);
            if ((ft - mu * lgt) + nu_pen * tht <= mref + 1e-4 * alpha * Dm + 1e-13 * fabs(phi0)) break;
            if (ls < 29) alpha *= 0.5;
        }
        // the duals never step further than the primal variables actually moved: a dual step taken
        // without its primal counterpart (line search cut alpha) blows up the dual infeasibility of rows with tiny slacks
        a_d = fmin(a_d, alpha);
        n_tiny = (alpha < 1e-10) ? n_tiny + 1 : 0;
        PROF_T(6);
#ifdef NMPC_PROFILE
        if (prof_out && tid == 0 && (int)inst == P.trace_inst && iter < 2040) {
            // linearised stationarity of the v-row of robot 0 at stage 0 with the QP multipliers (should be ~1e-14)
            double u0 = U[0], du0 = DU[0], sl = SUL[0], su = SUU[0], zl = ZUL[0], zu = ZUU[0];
            double dsl = du0 + ((u0 + P.vmax) - sl), dsu = -du0 + ((P.vmax - u0) - su);
            double znl = zl + (mu - sl * zl - zl * dsl) / sl, znu = zu + (mu - su * zu - zu * dsu) / su;
            double lin = 2 * P.r[0] * (u0 + du0) - T * (CS[0] * RV[NX] + SN[0] * RV[NX + 1]) - (znl - znu);
            double *tr2 = reinterpret_cast<double *>(prof_out + 12 + 16 * 2048) + (size_t)iter * 8;
            tr2[5] = lin; tr2[6] = du0; tr2[7] = dsu;
        }
        if (prof_out && tid == 0 && (int)inst == P.trace_inst && iter < 2040) {
            double *tr = reinterpret_cast<double *>(prof_out + 12) + (size_t)iter * 16;
            tr[0] = E0; tr[1] = e_d; tr[2] = e_c; tr[3] = e_h; tr[4] = szmax; tr[5] = mu; tr[6] = alpha; tr[7] = a_p; tr[8] = a_d;
            tr[9] = delta; tr[10] = nu_pen; tr[11] = dphi; tr[12] = th0; tr[13] = f; tr[14] = s_d; tr[15] = mult_max;
        }
#endif

        // ============ G. accept: duals first (they need the old primal point), then primal / slack step
        auto zup = [&](double sv, double zv, double ds, double &snew) {
            double dz = dz_of(mu, sv, zv, ds);
            snew = sv + alpha * ds;
            double z = zv + a_d * dz;
            return fmin(fmax(z, mu / (1e10 * snew)), 1e10 * mu / snew);
        };
        auto zupe = [&](double sv, double zv, double tv, double hv, double jd, double &snew, double &tnew) {      // an elastic row
            double ds, dz, dt;
            el_step(mu, rho, sv, zv, tv, hv, jd, ds, dz, dt);
            snew = sv + alpha * ds; tnew = tv + alpha * dt;
            double z = zv + a_d * dz;
            return fmin(fmax(z, mu / (1e10 * snew)), fmin(1e10 * mu / snew, rho - mu / (1e10 * tnew)));
        };
        for (int e = tid; e < N * NU; e += TPB) {
            int c = e % NU;
            double lo = lbu(c), u = U[e], du = DU[e], sl = SUL[e], su = SUU[e], sn;
            ZUL[e] = zup(sl, ZUL[e], du + ((u - lo) - sl), sn); SUL[e] = sn;
            ZUU[e] = zup(su, ZUU[e], -du + ((-lo - u) - su), sn); SUU[e] = sn;
        }
        for (int e = tid; e < N * NXB; e += TPB) {
            int k = 1 + e / NXB, s = e - (k - 1) * NXB;
            const int es = k * NXB + s;
            double v = X[k * NX + bst(s)], dv = DX[k * NX + bst(s)], b = bvl(s), sl = v + b, su = b - v, sn;
            ZXL[es] = zup(sl, ZXL[es], dv + ((v + b) - sl), sn);
            ZXU[es] = zup(su, ZXU[es], -dv + ((b - v) - su), sn);
        }
        for (int it = tid; it < (N - 1) * NPA; it += TPB) {
            int k = 1 + it / NPd, q = it - (k - 1) * NP, i, j;
            pair_ij(q, i, j);
            const int oi = k * NX + 3 * i, oj = k * NX + 3 * j;
            double ex = X[oi] - X[oj], ey = X[oi + 1] - X[oj + 1], sv = SPp[k * NP + q], sn;
            if (el) {
                double tn;
                ZPp[k * NP + q] = zupe(sv, ZPp[k * NP + q], TPp[k * NP + q], h_pair(ex, ey, P.dmin2), jd_pair(ex, ey, DX[oi] - DX[oj], DX[oi + 1] - DX[oj + 1]), sn, tn);
                SPp[k * NP + q] = sn; TPp[k * NP + q] = tn;
                continue;
            }
            double ds = ds_pair(ex, ey, DX[oi] - DX[oj], DX[oi + 1] - DX[oj + 1], P.dmin2, sv);
            ZPp[k * NP + q] = zup(sv, ZPp[k * NP + q], ds, sn);
            SPp[k * NP + q] = sn;
        }
        for (int it = tid; it < (N - 1) * MK; it += TPB) {
            int k = 1 + it / MK, e = it - (k - 1) * MK, i = e / K, o = e - i * K;
            const int oi = k * NX + 3 * i;
            double ex = X[oi] - P.obs[3 * o], ey = X[oi + 1] - P.obs[3 * o + 1], rr = r_obs(ex, ey), sv = SO[k * MK + e], sn;
            if (el) {
                double tn;
                ZO[k * MK + e] = zupe(sv, ZO[k * MK + e], TOb[k * MK + e], h_obs(rr, P.robdim, P.obs[3 * o + 2], P.margin), jd_obs(ex, ey, rr, DX[oi], DX[oi + 1]), sn, tn);
                SO[k * MK + e] = sn; TOb[k * MK + e] = tn;
                continue;
            }
            double ds = ds_obs(ex, ey, rr, DX[oi], DX[oi + 1], h_obs(rr, P.robdim, P.obs[3 * o + 2], P.margin), sv);
            ZO[k * MK + e] = zup(sv, ZO[k * MK + e], ds, sn);
            SO[k * MK + e] = sn;
        }
        __syncthreads();
        for (int e = tid; e < N1 * NX; e += TPB) { X[e] += alpha * DX[e]; if (e >= NX) LAM[e] += alpha * (RV[e] - LAM[e]); }
        for (int e = tid; e < N * NU; e += TPB) U[e] += alpha * DU[e];
        __syncthreads();
        trig();
        __syncthreads();
        f = ft; lgs = lgt; th0 = tht; e_c = ect; e_h = eht;
        iter++;
        PROF_T(8);
        if (n_tiny >= 5) {
            if (n_restart >= 3) { if (n_cold < NMPC_COLD_RETRIES) cold_retry(); else status = NMPC_STATUS_STALLED; break; }
            n_restart++; n_tiny = 0;
            mu = fmax(mu, P.mu_init);
            restarting = true;
            break;
        }
    }
    if (!restarting) break;
    }

    __syncthreads();
    for (int e = tid; e < N1 * NX; e += TPB) wo[e] = X[e];
    for (int e = tid; e < N * NU; e += TPB) wo[(size_t)N1 * NX + e] = U[e];
    if (tid == 0) {
        if (obj_out) obj_out[inst] = f;
        if (status_out) status_out[inst] = status;
        if (iters_out) iters_out[inst] = iter;
        if (kkt_out) kkt_out[inst] = kkt;
    }
#ifdef NMPC_PROFILE
    if (tid == 0 && prof_out) for (int i = 0; i < 12; i++) atomicAdd((unsigned long long *)&prof_out[i], (unsigned long long)prof[i]);
#endif
}

// LDS bytes of one instance
template <int M_, int THB> static size_t lds_bytes(const KParams &P, int tpb)
{
    using G = G2<M_, THB>;
    const size_t N = P.N, N1 = P.N + 1, MK = (size_t)M_ * P.K;
    const size_t W1S = (size_t)G::NX * G::NX + G::NX + (size_t)G::NX * G::LDG, W2S = (size_t)G::NU * G::LD + G::NU + G::PACK;
    size_t d = N1 * G::NX * 2 + N * G::NU * 5 + N1 * G::NP * 2 + N1 * MK * 2 + N1 * G::NXB * 2 + N * M_ * 2 + W1S + W2S + G::NU + G::NX + 8;
    if (N1 * G::NX + N * G::NU > W1S) d += N1 * G::NX + N * G::NU;      // step does not fit the Riccati working area
    if (N1 * G::NX > W2S) d += N1 * G::NX;
    d += ((size_t)((G::NT + tpb - 1) / tpb) * tpb * 2 + 7) / 8;      // 16-bit element table
    return d * sizeof(double);
}

template <int M_, int THB, int TPB> static hipError_t launch2_mtt(const KParams &P, int B, const double *p, const double *w0, double *w_out, double *obj,
                                                                  int32_t *status, int32_t *iters, double *kkt, double *ws, long long *prof, hipStream_t st)
{
    size_t lds = lds_bytes<M_, THB>(P, TPB);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto kern = solve_lds_kernel<M_, THB, TPB>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(B), dim3(TPB), lds, st, P, p, w0, w_out, obj, status, iters, kkt, ws, prof);
    return hipGetLastError();
}

template <int M_, int THB> static hipError_t launch2_mt(const KParams &P, int B, const double *p, const double *w0, double *w_out, double *obj,
                                                        int32_t *status, int32_t *iters, double *kkt, double *ws, long long *prof, hipStream_t st)
{
    // Throughput shape: one wave per instance up to six robots (4 instances per CU, 1024 resident on the chip); the augmented
    // matrix of 8 / 10 robots (860 / 1325 elements) is spread over 2 / 4 waves so that the per-thread element tables stay in
    // registers (64 threads spill to scratch there).
    // Latency shape (five and six robots): a batch that leaves most of the chip idle is solved with 2 or 4 waves per instance —
    // measured 189 / 148 / 139 us per iteration at 64 / 128 / 256 threads (m=6, N=20, one instance per CU).  This is the
    // reference's own use (one swarm per control period) and the tail of small closed-loop batches.
    // (the wider shapes carry a larger 16-bit element table: a horizon that only fits the 160 KB of LDS in the throughput
    // shape stays on it)
    if constexpr (M_ == 5 || M_ == 6) {
        if (B <= 256 && lds_bytes<M_, THB>(P, 256) <= (size_t)160 * 1024) return launch2_mtt<M_, THB, 256>(P, B, p, w0, w_out, obj, status, iters, kkt, ws, prof, st);
        if (B <= 512 && lds_bytes<M_, THB>(P, 128) <= (size_t)160 * 1024) return launch2_mtt<M_, THB, 128>(P, B, p, w0, w_out, obj, status, iters, kkt, ws, prof, st);
    }
    constexpr int TPB = (M_ <= 6) ? 64 : (M_ <= 8 ? 128 : 256);
    return launch2_mtt<M_, THB, TPB>(P, B, p, w0, w_out, obj, status, iters, kkt, ws, prof, st);
}
template <int M_> static hipError_t launch2_m(const KParams &P, int B, const double *p, const double *w0, double *w_out, double *obj, int32_t *status,
                                              int32_t *iters, double *kkt, double *ws, long long *prof, hipStream_t st)
{
    return P.thb ? launch2_mt<M_, 1>(P, B, p, w0, w_out, obj, status, iters, kkt, ws, prof, st)
                 : launch2_mt<M_, 0>(P, B, p, w0, w_out, obj, status, iters, kkt, ws, prof, st);
}

hipError_t launch_solve_lds(const KParams &P, int m, int B, const double *p, const double *w0, double *w_out, double *obj, int32_t *status,
                            int32_t *iters, double *kkt, double *ws, long long *prof, hipStream_t st)
{
    switch (m) {
    case 1: return launch2_m<1>(P, B, p, w0, w_out, obj, status, iters, kkt, ws, prof, st);
    case 2: return launch2_m<2>(P, B, p, w0, w_out, obj, status, iters, kkt, ws, prof, st);
    case 3: return launch2_m<3>(P, B, p, w0, w_out, obj, status, iters, kkt, ws, prof, st);
    case 4: return launch2_m<4>(P, B, p, w0, w_out, obj, status, iters, kkt, ws, prof, st);
    case 5: return launch2_m<5>(P, B, p, w0, w_out, obj, status, iters, kkt, ws, prof, st);
    case 6: return launch2_m<6>(P, B, p, w0, w_out, obj, status, iters, kkt, ws, prof, st);
    case 7: return launch2_m<7>(P, B, p, w0, w_out, obj, status, iters, kkt, ws, prof, st);
    case 8: return launch2_m<8>(P, B, p, w0, w_out, obj, status, iters, kkt, ws, prof, st);
    case 9: return launch2_m<9>(P, B, p, w0, w_out, obj, status, iters, kkt, ws, prof, st);
    case 10: return launch2_m<10>(P, B, p, w0, w_out, obj, status, iters, kkt, ws, prof, st);
    default: return hipErrorInvalidValue;
    }
}

// LDS bytes one instance of the LDS-resident kernel needs in its throughput shape (0 if m is not supported); the C ABI falls
// back to the HBM-resident kernel when this exceeds the 160 KB of a CU
size_t lds_kernel_bytes(const KParams &P, int m)
{
#define LB(M, T) case M: return P.thb ? lds_bytes<M, 1>(P, T) : lds_bytes<M, 0>(P, T);
    switch (m) {
        LB(1, 64) LB(2, 64) LB(3, 64) LB(4, 64) LB(5, 64) LB(6, 64) LB(7, 128) LB(8, 128) LB(9, 256) LB(10, 256)
    default: return 0;
    }
#undef LB
}

// workspace doubles per instance for this kernel (packs + transposed gains)
void lds_kernel_workspace(const KParams &P, int m, int64_t *pack_off, int64_t *kt_off, int64_t *stride)
{
    const int thb = P.thb;
    int64_t pack = 0, kts = 0;
#define SZ(M)                                                                                                                     \
    case M:                                                                                                                       \
        pack = thb ? G2<M, 1>::PACK : G2<M, 0>::PACK;                                                                             \
        kts = thb ? G2<M, 1>::KTS : G2<M, 0>::KTS;                                                                                \
        break;
    switch (m) { SZ(1) SZ(2) SZ(3) SZ(4) SZ(5) SZ(6) SZ(7) SZ(8) SZ(9) SZ(10) default: break; }
#undef SZ
    int64_t o = 0;
    *pack_off = o; o += (int64_t)(P.N + 1) * pack; o = (o + 15) / 16 * 16;
    *kt_off = o; o += (int64_t)P.N * kts; o = (o + 15) / 16 * 16;
    *stride = o;
}

}  // namespace nmpc
