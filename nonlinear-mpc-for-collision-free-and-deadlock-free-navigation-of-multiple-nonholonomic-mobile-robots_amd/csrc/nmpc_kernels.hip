// nmpc_kernels.hip — gfx950 (MI355X, CDNA4) kernels of the batched NMPC solve.
//
// One workgroup = one swarm instance (one NLP).  The whole interior-point solve runs inside a
// single launch: no host round trips, no inter-workgroup communication, so converged
// workgroups simply retire and the dispatcher back-fills the CU with the next instances.
//
// Reference blocks replaced (C6 = AllScripts/centralized_six_robots_implementation.py):
//   unicycle rhs C6:207-237, Euler defects C6:318-323, stage cost C6:314 (Q,R C6:252-266),
//   pair rows C6:288-306, obstacle rows third_scenario_mpc_obstacle_avoidance.py:145-150,
//   bounds C6:349-352, and the nlpsol('ipopt') call C6:345-346,432 itself.
//
// Data layout in HBM: per instance one contiguous fp64 workspace (stride multiple of 16
// doubles = 128 B) holding the iterate (X, U, lambda, slacks S, duals Z), the step, the
// per-stage gradient pieces and the Riccati feedback gains K_k; every per-stage block is
// contiguous so a wave reads it with unit-stride 8-byte lanes.  The stage working set of the
// Riccati sweep (P, P[A B], Qxx/Qux/Quu, the 3x2 Jacobian entries sin/cos per robot) lives in
// LDS.  All arithmetic is IEEE fp64.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "nmpc_device.h"

namespace nmpc {

// ------------------------------------------------------------------------------------------
// block-wide reductions (wave64 butterfly, then LDS across waves)
template <int TPB> __device__ __forceinline__ double blk_sum(double v, double *red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if constexpr (TPB > 64) {
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < TPB / 64; w++) t += red[w];
        v = t;
    }
    return v;
}
template <int TPB> __device__ __forceinline__ double blk_max(double v, double *red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    if constexpr (TPB > 64) {
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        double t = red[0];
#pragma unroll
        for (int w = 1; w < TPB / 64; w++) t = fmax(t, red[w]);
        v = t;
    }
    return v;
}
template <int TPB> __device__ __forceinline__ double blk_min(double v, double *red) { return -blk_max<TPB>(-v, red); }

// ------------------------------------------------------------------------------------------
template <int M_> struct Geo {
    static constexpr int NX = 3 * M_, NU = 2 * M_, NP = M_ * (M_ - 1) / 2, NZ = NX + NU;
};

// index of pair (a,b), a<b, in lexicographic order (C6:288-306)
template <int M_> __device__ __forceinline__ int pair_index(int a, int b) { return a * (2 * M_ - a - 1) / 2 + (b - a - 1); }

__device__ __forceinline__ bool slot_active(const KParams &P, int k, int s)
{
    if (s < P.o_xl) return k < P.N;
    if (s < P.o_pr) return k >= 1;
    return k >= 1 && k <= P.N - 1;
}
__device__ __forceinline__ int bnd_state(const KParams &P, int s) { return P.thb ? s : 3 * (s >> 1) + (s & 1); }
__device__ __forceinline__ double bnd_val(const KParams &P, int s) { return (P.thb && (s % 3 == 2)) ? P.thmax : P.xymax; }
__device__ __forceinline__ double lbu(const KParams &P, int c) { return (c & 1) ? -P.wmax : -P.vmax; }

// value of inequality slot s at stage k; x,u point at the stage's state / control block
template <int M_>
__device__ __forceinline__ double slot_h(const KParams &P, const int *sPi, const int *sPj, int k, int s, const double *x, const double *u)
{
    constexpr int NU = Geo<M_>::NU;
    if (s < P.o_xl) {
        if (k >= P.N) return 1.0;
        return (s < NU) ? (u[s] - lbu(P, s)) : (-lbu(P, s - NU) - u[s - NU]);
    }
    if (s < P.o_pr) {
        if (k < 1) return 1.0;
        int sl = (s < P.o_xu) ? s - P.o_xl : s - P.o_xu;
        double v = x[bnd_state(P, sl)], b = bnd_val(P, sl);
        return (s < P.o_xu) ? (v + b) : (b - v);
    }
    if (k < 1 || k > P.N - 1) return 1.0;
    if (s < P.o_ob) {
        int pq = s - P.o_pr, i = sPi[pq], j = sPj[pq];
        double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1];
        return dx * dx + dy * dy - P.dmin2;
    }
    int io = s - P.o_ob, i = io / P.K, o = io - i * P.K;
    double dx = x[3 * i] - P.obs[3 * o], dy = x[3 * i + 1] - P.obs[3 * o + 1];
    return sqrt(dx * dx + dy * dy) - P.robdim - P.obs[3 * o + 2] - P.margin;
}

// (Jx_k d)[s] for x-slots, (Ju_k du)[s] for u-slots
template <int M_>
__device__ __forceinline__ double slot_jd(const KParams &P, const int *sPi, const int *sPj, int k, int s, const double *x, const double *dx_,
                                          const double *du)
{
    constexpr int NU = Geo<M_>::NU;
    if (s < P.o_xl) return (s < NU) ? du[s] : -du[s - NU];
    if (s < P.o_pr) {
        int sl = (s < P.o_xu) ? s - P.o_xl : s - P.o_xu;
        double v = dx_[bnd_state(P, sl)];
        return (s < P.o_xu) ? v : -v;
    }
    if (s < P.o_ob) {
        int pq = s - P.o_pr, i = sPi[pq], j = sPj[pq];
        double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1];
        return 2 * dx * (dx_[3 * i] - dx_[3 * j]) + 2 * dy * (dx_[3 * i + 1] - dx_[3 * j + 1]);
    }
    int io = s - P.o_ob, i = io / P.K, o = io - i * P.K;
    double dx = x[3 * i] - P.obs[3 * o], dy = x[3 * i + 1] - P.obs[3 * o + 1];
    double rr = sqrt(dx * dx + dy * dy);
    return (dx * dx_[3 * i] + dy * dx_[3 * i + 1]) / rr;
}

// robot i's three components of Jx_k^T v, with v(slot) supplied by a functor (partners in ascending order)
template <int M_, class F>
__device__ __forceinline__ void jxT_robot(const KParams &P, int k, int i, const double *x, F v, double &o0, double &o1, double &o2)
{
    o0 = 0.0; o1 = 0.0; o2 = 0.0;
    if (k < 1) return;
    if (P.thb) {
        o0 = v(P.o_xl + 3 * i) - v(P.o_xu + 3 * i);
        o1 = v(P.o_xl + 3 * i + 1) - v(P.o_xu + 3 * i + 1);
        o2 = v(P.o_xl + 3 * i + 2) - v(P.o_xu + 3 * i + 2);
    } else {
        o0 = v(P.o_xl + 2 * i) - v(P.o_xu + 2 * i);
        o1 = v(P.o_xl + 2 * i + 1) - v(P.o_xu + 2 * i + 1);
    }
    if (k > P.N - 1) return;
    const double xi = x[3 * i], yi = x[3 * i + 1];
#pragma unroll
    for (int j = 0; j < (P.pairs ? M_ : 0); j++) {
        if (j == i) continue;
        int pq = (i < j) ? pair_index<M_>(i, j) : pair_index<M_>(j, i);
        double z = v(P.o_pr + pq);
        o0 += 2 * (xi - x[3 * j]) * z;
        o1 += 2 * (yi - x[3 * j + 1]) * z;
    }
    for (int o = 0; o < P.K; o++) {
        double dx = xi - P.obs[3 * o], dy = yi - P.obs[3 * o + 1];
        double rr = sqrt(dx * dx + dy * dy), z = v(P.o_ob + i * P.K + o);
        o0 += dx / rr * z; o1 += dy / rr * z;
    }
}

// ------------------------------------------------------------------------------------------
// The solve.  Grid: one workgroup per instance.  TPB threads cooperate on one NLP.
template <int M_, int TPB>
__global__ __launch_bounds__(TPB) void solve_kernel(const KParams P, const double *__restrict__ p_in, const double *__restrict__ w0,
                                                     double *__restrict__ w_out, double *__restrict__ obj_out,
                                                     int32_t *__restrict__ status_out, int32_t *__restrict__ iters_out,
                                                     double *__restrict__ kkt_out, double *__restrict__ ws)
{
    constexpr int NX = Geo<M_>::NX, NU = Geo<M_>::NU, NP = Geo<M_>::NP, NZ = Geo<M_>::NZ;
    const int tid = threadIdx.x;
    const int N = P.N, NH = P.nh;
    const double T = P.T;
    const size_t inst = (P.order && *P.order_bad == 0) ? (size_t)P.order[blockIdx.x] : (size_t)blockIdx.x;
    const int NPA = P.pairs ? NP : 0;

    __shared__ double sP[NX * NX];   // P_{k+1}, then Qxx, then P_k
    __shared__ double sG[NX * NZ];   // P [A B]
    __shared__ double sQux[NU * NX]; // Qux, then Y = L^-1 Qux, then K
    __shared__ double sQuu[NU * NU]; // Quu, then its Cholesky factor
    __shared__ double sPv[NX], sPb[NX], sQx[NX], sQu[NU], sD0[NU];
    __shared__ double sX[NX], sU[NU], sCk[NX], sA[M_], sB[M_], sSn[M_], sCs[M_];
    __shared__ double sE[3 * (NP > 0 ? NP : 1)];
    __shared__ double sXs[NX], sRed[TPB / 64 + 1];
    __shared__ int sPi[NP > 0 ? NP : 1], sPj[NP > 0 ? NP : 1];
    __shared__ int sFail;

    // ---- workspace carve-up (doubles)
    double *base = ws + inst * P.stride;
    double *X = base + P.oX, *U = base + P.oU, *LAM = base + P.oLAM, *S = base + P.oS, *Z = base + P.oZ;
    double *DX = base + P.oDX, *DU = base + P.oDU, *LAMN = base + P.oLAMN, *DS = base + P.oDS, *DZ = base + P.oDZ;
    double *EL = base + P.oEL, *DEL = EL + (size_t)(P.N + 1) * P.nh;      // elastic variables t and their steps (elastic phase only; pair / obstacle slots)
    const double rho = P.rho_el;
    bool el = false;      // elastic phase: the second restart of last resort (oracle/nmpc_oracle.c has the derivation)
#define ELSLOT(s) (el && (s) >= P.o_pr)
    double *SN = base + P.oSN, *CS = base + P.oCS, *Cd = base + P.oC, *H = base + P.oH, *GX = base + P.oGX;
    double *HUU = base + P.oHUU, *GU = base + P.oGU, *HVT = base + P.oHVT, *HTT = base + P.oHTT;
    double *KG = base + P.oKG, *KFF = base + P.oKFF;
    double *CKP = base + P.oCKP;      // saved cost-to-go [P | p] of the backward sweep, one slot of NX * NX + NX doubles per NMPC_CKPT_EVERY stages

    const double *pp = p_in + inst * (2 * NX);
    const double *wi = w0 + inst * (size_t)P.nvar;
    double *wo = w_out + inst * (size_t)P.nvar;

    // pair tables
    if (tid == 0) {
        int q = 0;
        for (int i = 0; i < M_; i++) for (int j = i + 1; j < M_; j++) { sPi[q] = i; sPj[q] = j; q++; }
        sFail = 0;
    }
    for (int c = tid; c < NX; c += TPB) sXs[c] = pp[NX + c];
    // ---- load the start: X_0 := x0 (C6:278 pins it), push into the interior of the simple bounds
    const double bp = 1e-2;
    for (int e = tid; e < (N + 1) * NX; e += TPB) {
        int k = e / NX, c = e - k * NX;
        double v = (k == 0) ? pp[c] : wi[e];
        if (k >= 1) {
            int d = c % 3;
            if (d < 2 || P.thb) {
                double b = (d == 2) ? P.thmax : P.xymax, px = fmin(bp * fmax(1.0, b), bp * 2.0 * b);
                v = fmin(fmax(v, -b + px), b - px);
            }
        }
        X[e] = v;
        LAM[e] = 0.0;
    }
    for (int e = tid; e < N * NU; e += TPB) {
        int c = e % NU;
        double lo = lbu(P, c), hi = -lo, pu = fmin(bp * fmax(1.0, fabs(lo)), bp * (hi - lo));
        U[e] = fmin(fmax(wi[(size_t)(N + 1) * NX + e], lo + pu), hi - pu);
    }
    __syncthreads();

    // ---- stage-0 pair / obstacle rows act on the pinned state: feasibility pre-check
    {
        double bad = 0.0;
        for (int q = tid; q < NPA; q += TPB) {
            int i = sPi[q], j = sPj[q];
            double dx = pp[3 * i] - pp[3 * j], dy = pp[3 * i + 1] - pp[3 * j + 1];
            if (dx * dx + dy * dy < P.dmin2 - NMPC_X0_TOL) bad = 1.0;
        }
        for (int e = tid; e < M_ * P.K; e += TPB) {
            int i = e / P.K, o = e - i * P.K;
            double dx = pp[3 * i] - P.obs[3 * o], dy = pp[3 * i + 1] - P.obs[3 * o + 1];
            if (sqrt(dx * dx + dy * dy) - P.robdim - P.obs[3 * o + 2] < P.margin - NMPC_X0_TOL) bad = 1.0;
        }
        bad = blk_max<TPB>(bad, sRed);
        if (bad > 0.0) {
            for (int e = tid; e < (N + 1) * NX; e += TPB) wo[e] = X[e];
            for (int e = tid; e < N * NU; e += TPB) wo[(size_t)(N + 1) * NX + e] = U[e];
            if (tid == 0) {
                if (obj_out) obj_out[inst] = NAN;
                if (status_out) status_out[inst] = NMPC_STATUS_INFEASIBLE_X0;
                if (iters_out) iters_out[inst] = 0;
                if (kkt_out) kkt_out[inst] = INFINITY;
            }
            return;
        }
    }

    // ---- point evaluation: trig cache, defects, inequality values, objective
    auto eval_point = [&]() -> double {
        double fs = 0.0;
        for (int it = tid; it < N * M_; it += TPB) {
            int k = it / M_, i = it - k * M_;
            const double *x = X + k * NX + 3 * i, *xn = x + NX, *u = U + k * NU + 2 * i;
            double s, c;
            sincos(x[2], &s, &c);
            SN[it] = s; CS[it] = c;
            Cd[k * NX + 3 * i] = xn[0] - (x[0] + T * u[0] * c);
            Cd[k * NX + 3 * i + 1] = xn[1] - (x[1] + T * u[0] * s);
            Cd[k * NX + 3 * i + 2] = xn[2] - (x[2] + T * u[1]);
            double e0 = x[0] - sXs[3 * i], e1 = x[1] - sXs[3 * i + 1], e2 = x[2] - sXs[3 * i + 2];
            fs += P.q[0] * e0 * e0 + P.q[1] * e1 * e1 + P.q[2] * e2 * e2 + P.r[0] * u[0] * u[0] + P.r[1] * u[1] * u[1];
        }
        for (int k = 0; k <= N; k++)
            for (int s = tid; s < NH; s += TPB) H[k * NH + s] = slot_h<M_>(P, sPi, sPj, k, s, X + k * NX, U + (k < N ? k : 0) * NU);
        return blk_sum<TPB>(fs, sRed);
    };
    // merit pieces at the trial point (X + a dX, U + a dU, S + a dS); nothing is stored
    auto eval_trial = [&](double a, double mu, double &phi, double &theta) {
        double fs = 0.0, th = 0.0, lg = 0.0;
        for (int it = tid; it < N * M_; it += TPB) {
            int k = it / M_, i = it - k * M_;
            const int ox = k * NX + 3 * i, ou = k * NU + 2 * i;
            double x0 = X[ox] + a * DX[ox], x1 = X[ox + 1] + a * DX[ox + 1], x2 = X[ox + 2] + a * DX[ox + 2];
            double n0 = X[ox + NX] + a * DX[ox + NX], n1 = X[ox + NX + 1] + a * DX[ox + NX + 1], n2 = X[ox + NX + 2] + a * DX[ox + NX + 2];
            double u0 = U[ou] + a * DU[ou], u1 = U[ou + 1] + a * DU[ou + 1];
            double s, c;
            sincos(x2, &s, &c);
            th += fabs(n0 - (x0 + T * u0 * c)) + fabs(n1 - (x1 + T * u0 * s)) + fabs(n2 - (x2 + T * u1));
            double e0 = x0 - sXs[3 * i], e1 = x1 - sXs[3 * i + 1], e2 = x2 - sXs[3 * i + 2];
            fs += P.q[0] * e0 * e0 + P.q[1] * e1 * e1 + P.q[2] * e2 * e2 + P.r[0] * u0 * u0 + P.r[1] * u1 * u1;
        }
        for (int k = 0; k <= N; k++) {
            // stage k trial state / control staged in LDS so slot_h can index it
            __syncthreads();
            for (int c = tid; c < NX; c += TPB) sX[c] = X[k * NX + c] + a * DX[k * NX + c];
            if (k < N) for (int c = tid; c < NU; c += TPB) sU[c] = U[k * NU + c] + a * DU[k * NU + c];
            __syncthreads();
            for (int s = tid; s < NH; s += TPB)
                if (slot_active(P, k, s)) {
                    double st = S[k * NH + s] + a * DS[k * NH + s];
                    double h = slot_h<M_>(P, sPi, sPj, k, s, sX, sU);
                    lg += log(st);
                    if (ELSLOT(s)) { const double tt = EL[k * NH + s] + a * DEL[k * NH + s]; lg += log(tt); fs += rho * tt; th += fabs(h + tt - st); }
                    else
                    th += fabs(h - st);
                }
        }
        fs = blk_sum<TPB>(fs, sRed);
        lg = blk_sum<TPB>(lg, sRed);
        theta = blk_sum<TPB>(th, sRed);
        phi = fs - mu * lg;
    };

    double mu = P.mu_init;
    double f = eval_point();
    __syncthreads();
    // slacks and duals from the constraint values H of the current point (also the barrier restart after a stall)
    auto init_barrier = [&]() {
        for (int e = tid; e < (N + 1) * NH; e += TPB) {
            int k = e / NH, s = e - k * NH;
            if (!slot_active(P, k, s)) { S[e] = 1.0; Z[e] = 0.0; continue; }
            double fl = (s < P.o_xl) ? 1e-12 : bp;
            DEL[e] = 0.0;
            if (ELSLOT(s)) { const double tv = fmax(bp, bp - H[e]), sv = H[e] + tv; EL[e] = tv; S[e] = sv; Z[e] = fmin(mu / sv, 0.5 * rho); continue; }
            double sv = fmax(H[e], fl);
            S[e] = sv; Z[e] = mu / sv;
        }
        __syncthreads();
    };
    init_barrier();

    double delta_last = 0.0, nu_pen = 1.0, kkt = INFINITY;
    bool need_shift = false;
    int n_tiny = 0, n_restart = 0;      // consecutive iterations with a step length below 1e-10 (stall -> restart, then NMPC_STATUS_STALLED)
    double mh0 = 0.0, mh1 = 0.0, mh2 = 0.0, mh_mu = -1.0, mh_nu = -1.0;
    int mcount = 0;
    int iter = 0, status = NMPC_STATUS_MAX_ITER;
    const double n_ineq = (double)P.n_ineq;
    // (Re)start of the barrier iteration from the current primal point, or — cold — from the reference's cold start X_k = x0, U = 0
    // (C6:398-400; the cold-start retry of NMPC_COLD_RETRY_ITERS): interior push, slacks onto the constraint values, duals mu/s, multipliers 0
    int n_cold = 0, it_base = 0;
    auto do_restart = [&](bool cold) {
        if (cold) { n_cold++; it_base = iter; el = n_cold >= 2; mu = P.mu_init; n_tiny = 0; n_restart = 0; }      // the second restart of last resort is the elastic phase
        for (int e = tid + NX; e < (N + 1) * NX; e += TPB) {
            if (cold) X[e] = X[e % NX];
            const int d = (e % NX) % 3;
            if (d < 2 || P.thb) {
                double b = (d == 2) ? P.thmax : P.xymax, px = fmin(bp * fmax(1.0, b), bp * 2.0 * b);
                X[e] = fmin(fmax(X[e], -b + px), b - px);
            }
            LAM[e] = 0.0;
        }
        for (int e = tid; e < N * NU; e += TPB) {
            double lo = lbu(P, e % NU), hi = -lo, pu = fmin(bp * fmax(1.0, fabs(lo)), bp * (hi - lo));
            U[e] = fmin(fmax(cold ? 0.0 : U[e], lo + pu), hi - pu);
        }
        __syncthreads();
        f = eval_point();
        __syncthreads();
        init_barrier();
        delta_last = 0.0; nu_pen = 1.0; need_shift = false; mcount = 0;
    };

    for (;;) {
        // ================= optimality error (IPOPT eq. 5), residuals per (stage, robot)
        double e_d = 0.0, lsum = 0.0;
        for (int it = tid; it < N * M_; it += TPB) {
            int k = it / M_, i = it - k * M_, kk = k + 1;
            const double *x = X + kk * NX;
            double r0 = LAM[kk * NX + 3 * i], r1 = LAM[kk * NX + 3 * i + 1], r2 = LAM[kk * NX + 3 * i + 2];
            lsum += fabs(r0) + fabs(r1) + fabs(r2);
            if (kk < N) {
                const double *ln = LAM + (kk + 1) * NX + 3 * i;
                double v = U[kk * NU + 2 * i];
                double a = -T * v * SN[kk * M_ + i], b = T * v * CS[kk * M_ + i];
                r0 += 2 * P.q[0] * (x[3 * i] - sXs[3 * i]) - ln[0];
                r1 += 2 * P.q[1] * (x[3 * i + 1] - sXs[3 * i + 1]) - ln[1];
                r2 += 2 * P.q[2] * (x[3 * i + 2] - sXs[3 * i + 2]) - (ln[2] + a * ln[0] + b * ln[1]);
            }
            double j0, j1, j2;
            const double *zk = Z + kk * NH;
            jxT_robot<M_>(P, kk, i, x, [&](int s) { return zk[s]; }, j0, j1, j2);
            e_d = fmax(e_d, fmax(fabs(r0 - j0), fmax(fabs(r1 - j1), fabs(r2 - j2))));
            // control rows of stage k
            const double *ln = LAM + (k + 1) * NX + 3 * i, *u = U + k * NU + 2 * i, *z = Z + k * NH;
            double c = CS[it], s = SN[it];
            double rv = 2 * P.r[0] * u[0] - T * (c * ln[0] + s * ln[1]) - (z[P.o_ul + 2 * i] - z[P.o_uu + 2 * i]);
            double rw = 2 * P.r[1] * u[1] - T * ln[2] - (z[P.o_ul + 2 * i + 1] - z[P.o_uu + 2 * i + 1]);
            e_d = fmax(e_d, fmax(fabs(rv), fabs(rw)));
        }
        double e_c = 0.0, e_h = 0.0, zsum = 0.0, szmax = 0.0, szmin = INFINITY;
        for (int e = tid; e < N * NX; e += TPB) e_c = fmax(e_c, fabs(Cd[e]));
        for (int e = tid; e < (N + 1) * NH; e += TPB) {
            int k = e / NH, s = e - k * NH;
            if (slot_active(P, k, s)) {
                double sv = S[e], zv = Z[e];
                if (ELSLOT(s)) { const double pt = EL[e] * (rho - zv); e_h = fmax(e_h, fabs(H[e] + EL[e] - sv)); szmax = fmax(szmax, pt); szmin = fmin(szmin, pt); }
                else
                e_h = fmax(e_h, fabs(H[e] - sv));
                zsum += zv;
                szmax = fmax(szmax, sv * zv); szmin = fmin(szmin, sv * zv);
            }
        }
        e_d = blk_max<TPB>(e_d, sRed); e_c = blk_max<TPB>(e_c, sRed); e_h = blk_max<TPB>(e_h, sRed);
        lsum = blk_sum<TPB>(lsum, sRed); zsum = blk_sum<TPB>(zsum, sRed);
        szmax = blk_max<TPB>(szmax, sRed); szmin = blk_min<TPB>(szmin, sRed);
        const double smax = 100.0;
        double s_d = fmax(smax, (lsum + zsum) / ((double)(N * NX) + n_ineq)) / smax;
        double s_c = fmax(smax, zsum / fmax(n_ineq, 1.0)) / smax;
        double E0 = fmax(fmax(e_d / s_d, e_c), fmax(e_h, szmax / s_c));
        kkt = E0;
        if (!(E0 == E0)) { if (n_cold < NMPC_COLD_RETRIES && iter < P.max_iter) { do_restart(true); continue; } status = NMPC_STATUS_NUMERIC; break; }
        if (E0 <= P.tol) {
            status = NMPC_STATUS_CONVERGED;
            if (el) {       // the penalty problem's solution solves the NLP only if every elastic variable has closed
                double tmax = 0.0;
                for (int e = tid; e < (N + 1) * NH; e += TPB) { int k = e / NH, s = e - k * NH; if (s >= P.o_pr && slot_active(P, k, s)) tmax = fmax(tmax, EL[e]); }
                if (blk_max<TPB>(tmax, sRed) > NMPC_X0_TOL) status = NMPC_STATUS_STALLED;
            }
            break;
        }
        if (iter >= P.max_iter) { status = NMPC_STATUS_MAX_ITER; break; }
        if (n_cold < NMPC_COLD_RETRIES && iter - it_base >= NMPC_COLD_RETRY_ITERS) { do_restart(true); continue; }
        // ================= monotone barrier update (IPOPT eq. 7)
        const double mu_min = P.tol / 10.0;
        for (;;) {
            double cm = fmax(fabs(szmax - mu), fabs(szmin - mu));
            double Emu = fmax(fmax(e_d / s_d, e_c), fmax(e_h, cm / s_c));
            if (mu > mu_min && Emu <= 10.0 * mu) mu = fmax(mu_min, fmin(0.2 * mu, pow(mu, 1.5)));
            else break;
        }
        const double tau = fmax(0.99, 1.0 - mu);

        // ================= per-stage gradient pieces (all stages in parallel)
        for (int it = tid; it < (N + 1) * M_; it += TPB) {
            int k = it / M_, i = it - k * M_;
            const double *x = X + k * NX;
            double g0 = 0.0, g1 = 0.0, g2 = 0.0;
            if (k >= 1) {
                if (k < N) {
                    g0 = 2 * P.q[0] * (x[3 * i] - sXs[3 * i]);
                    g1 = 2 * P.q[1] * (x[3 * i + 1] - sXs[3 * i + 1]);
                    g2 = 2 * P.q[2] * (x[3 * i + 2] - sXs[3 * i + 2]);
                }
                const double *sk = S + k * NH, *zk = Z + k * NH, *hk = H + k * NH;
                double j0, j1, j2;
                jxT_robot<M_>(P, k, i, x, [&](int s) {
                    double sv = sk[s], zv = zk[s];
                    if (ELSLOT(s)) { const double rz = rho - zv, D_ = sv / zv + EL[k * NH + s] / rz, q_ = hk[s] - mu / zv + mu / rz; return zv - q_ / D_; }
                    return mu / sv - zv / sv * (hk[s] - sv); }, j0, j1, j2);
                g0 -= j0; g1 -= j1; g2 -= j2;
            }
            GX[k * NX + 3 * i] = g0; GX[k * NX + 3 * i + 1] = g1; GX[k * NX + 3 * i + 2] = g2;
            if (k < N) {
                const double *u = U + k * NU + 2 * i, *ln = LAM + (k + 1) * NX + 3 * i;
                const double *sk = S + k * NH, *zk = Z + k * NH, *hk = H + k * NH;
#pragma unroll
                for (int d = 0; d < 2; d++) {
                    int cl = P.o_ul + 2 * i + d, cu = P.o_uu + 2 * i + d;
                    double sl = sk[cl], su = sk[cu], zl = zk[cl], zu = zk[cu];
                    HUU[k * NU + 2 * i + d] = 2 * P.r[d] + zl / sl + zu / su;
                    double vl = mu / sl - zl / sl * (hk[cl] - sl), vu = mu / su - zu / su * (hk[cu] - su);
                    GU[k * NU + 2 * i + d] = 2 * P.r[d] * u[d] - (vl - vu);
                }
                double c = CS[it], s = SN[it];
                HTT[it] = T * u[0] * (ln[0] * c + ln[1] * s);
                HVT[it] = T * (ln[0] * s - ln[1] * c);
            }
        }
        __syncthreads();

        // ================= Riccati sweep with inertia correction (IPOPT alg. IC)
        // first trial: delta = 0, except right after an iteration that needed a shift (then a quarter of that shift directly)
        double delta = need_shift ? fmax(1e-20, 0.25 * delta_last) : 0.0;
        // Partial re-factorisation (include/nmpc_constants.h): [P | p] entering every NMPC_CKPT_EVERY-th stage is saved; a rejected pivot at
        // stage kf escalates the shift and resumes at the saved stage NMPC_RESUME_STAGE(kf, N), the stages above keep their gains.
        int ntry = 0;
        bool ok;
        int kst = N - 1;
        for (;;) {
            ok = true;
            int kf = 0;
            if (kst == N - 1) {
                // terminal: P_N = Hxx_N (bound barrier terms only), p_N = gx_N; the inertia shift acts on the controls only
                for (int e = tid; e < NX * NX; e += TPB) sP[e] = 0.0;
                __syncthreads();
                for (int c = tid; c < NX; c += TPB) sPv[c] = GX[N * NX + c];
                __syncthreads();
                for (int s = tid; s < P.nxb; s += TPB) {
                    int c = bnd_state(P, s);
                    sP[c * NX + c] += Z[N * NH + P.o_xl + s] / S[N * NH + P.o_xl + s] + Z[N * NH + P.o_xu + s] / S[N * NH + P.o_xu + s];
                }
            } else {
                const double *ck = CKP + (size_t)((N - 1 - kst) / NMPC_CKPT_EVERY) * (NX * NX + NX);
                for (int e = tid; e < NX * NX; e += TPB) sP[e] = ck[e];
                for (int c = tid; c < NX; c += TPB) sPv[c] = ck[NX * NX + c];
            }
            __syncthreads();
            for (int k = kst; k >= 0; k--) {
                if (k < kst && (kst - k) % NMPC_CKPT_EVERY == 0) {      // save [P | p] entering this stage (each thread reads back its own words)
                    double *ck = CKP + (size_t)((N - 1 - k) / NMPC_CKPT_EVERY) * (NX * NX + NX);
                    for (int e = tid; e < NX * NX; e += TPB) ck[e] = sP[e];
                    for (int c = tid; c < NX; c += TPB) ck[NX * NX + c] = sPv[c];
                }
                // ---- stage data to LDS
                for (int c = tid; c < NX; c += TPB) { sX[c] = X[k * NX + c]; sCk[c] = Cd[k * NX + c]; }
                for (int c = tid; c < NU; c += TPB) sU[c] = U[k * NU + c];
                for (int i = tid; i < M_; i += TPB) {
                    double s = SN[k * M_ + i], c = CS[k * M_ + i], v = U[k * NU + 2 * i];
                    sSn[i] = s; sCs[i] = c; sA[i] = -T * v * s; sB[i] = T * v * c;
                }
                __syncthreads();
                // ---- Pb = p + P b, b = -c_k
                for (int r = tid; r < NX; r += TPB) {
                    double a = sPv[r];
                    for (int c = 0; c < NX; c++) a -= sP[r * NX + c] * sCk[c];
                    sPb[r] = a;
                }
                // ---- G = P [A B]
                for (int e = tid; e < NX * NZ; e += TPB) {
                    int r = e / NZ, col = e - r * NZ;
                    const double *Pr = sP + r * NX;
                    double g;
                    if (col < NX) {
                        int i = col / 3, d = col - 3 * i;
                        g = (d < 2) ? Pr[col] : (Pr[col] + sA[i] * Pr[col - 2] + sB[i] * Pr[col - 1]);
                    } else {
                        int cu = col - NX, i = cu >> 1;
                        g = (cu & 1) ? (T * Pr[3 * i + 2]) : (T * (sCs[i] * Pr[3 * i] + sSn[i] * Pr[3 * i + 1]));
                    }
                    sG[e] = g;
                }
                // pair blocks E_ij = 4 sigma dp dp^T - 2 z I of this stage's Hessian (k>=1)
                if (k >= 1) {
                    for (int q = tid; q < NP; q += TPB) {
                        int i = sPi[q], j = sPj[q];
                        double dx = sX[3 * i] - sX[3 * j], dy = sX[3 * i + 1] - sX[3 * j + 1];
                        double zz = P.pairs ? Z[k * NH + P.o_pr + q] : 0.0, sg = P.pairs ? zz / S[k * NH + P.o_pr + q] : 0.0;
                        if (el && P.pairs) sg = 1.0 / (S[k * NH + P.o_pr + q] / zz + EL[k * NH + P.o_pr + q] / (rho - zz));
                        sE[3 * q] = 4 * sg * dx * dx - 2 * zz; sE[3 * q + 1] = 4 * sg * dx * dy; sE[3 * q + 2] = 4 * sg * dy * dy - 2 * zz;
                    }
                }
                __syncthreads();
                // ---- [Qxx Qxu; Qux Quu] = [A B]^T G ; q = [A B]^T Pb   (P is dead: Qxx overwrites it)
                for (int e = tid; e < NX * NX; e += TPB) {
                    int r = e / NX, c = e - r * NX, i = r / 3, d = r - 3 * i;
                    sP[e] = (d < 2) ? sG[r * NZ + c] : (sG[r * NZ + c] + sA[i] * sG[(r - 2) * NZ + c] + sB[i] * sG[(r - 1) * NZ + c]);
                }
                for (int e = tid; e < NU * NX; e += TPB) {
                    int r = e / NX, c = e - r * NX, i = r >> 1;
                    sQux[e] = (r & 1) ? (T * sG[(3 * i + 2) * NZ + c]) : (T * (sCs[i] * sG[(3 * i) * NZ + c] + sSn[i] * sG[(3 * i + 1) * NZ + c]));
                }
                for (int e = tid; e < NU * NU; e += TPB) {
                    int r = e / NU, c = e - r * NU, i = r >> 1;
                    sQuu[e] = (r & 1) ? (T * sG[(3 * i + 2) * NZ + NX + c])
                                      : (T * (sCs[i] * sG[(3 * i) * NZ + NX + c] + sSn[i] * sG[(3 * i + 1) * NZ + NX + c]));
                }
                for (int i = tid; i < M_; i += TPB) {
                    sQx[3 * i] = sPb[3 * i]; sQx[3 * i + 1] = sPb[3 * i + 1];
                    sQx[3 * i + 2] = sPb[3 * i + 2] + sA[i] * sPb[3 * i] + sB[i] * sPb[3 * i + 1];
                    sQu[2 * i] = T * (sCs[i] * sPb[3 * i] + sSn[i] * sPb[3 * i + 1]);
                    sQu[2 * i + 1] = T * sPb[3 * i + 2];
                }
                __syncthreads();
                // ---- add the stage Hessian / gradient
                for (int c = tid; c < NU; c += TPB) {
                    double d = sQuu[c * NU + c] + HUU[k * NU + c] + delta;
                    sQuu[c * NU + c] = d; sD0[c] = d;
                    sQu[c] += GU[k * NU + c];
                }
                for (int i = tid; i < M_; i += TPB) sQux[(2 * i) * NX + 3 * i + 2] += HVT[k * M_ + i];
                if (k >= 1) {
                    for (int c = tid; c < NX; c += TPB) {
                        int i = c / 3, d = c - 3 * i;
                        double add = (k < N) ? 2 * P.q[d] : 0.0;
                        if (d == 2) add += HTT[k * M_ + i];
                        sP[c * NX + c] += add;
                        sQx[c] += GX[k * NX + c];
                    }
                    __syncthreads();
                    for (int s = tid; s < P.nxb; s += TPB) {
                        int c = bnd_state(P, s);
                        sP[c * NX + c] += Z[k * NH + P.o_xl + s] / S[k * NH + P.o_xl + s] + Z[k * NH + P.o_xu + s] / S[k * NH + P.o_xu + s];
                    }
                    __syncthreads();
                    // off-diagonal 2x2 blocks: -E_ij ; diagonal blocks: sum_j E_ij (ascending partner) + obstacle terms
                    for (int e = tid; e < NP * 4; e += TPB) {
                        int q = e >> 2, w = e & 3, i = sPi[q], j = sPj[q];
                        int rr = w >> 1, cc = w & 1;
                        double ev = sE[3 * q + rr + cc];
                        sP[(3 * i + rr) * NX + 3 * j + cc] -= ev;
                        sP[(3 * j + rr) * NX + 3 * i + cc] -= ev;
                    }
                    for (int i = tid; i < M_; i += TPB) {
                        double e00 = 0.0, e01 = 0.0, e11 = 0.0;
#pragma unroll
                        for (int j = 0; j < M_; j++) {
                            if (j == i) continue;
                            int q = (i < j) ? pair_index<M_>(i, j) : pair_index<M_>(j, i);
                            e00 += sE[3 * q]; e01 += sE[3 * q + 1]; e11 += sE[3 * q + 2];
                        }
                        for (int o = 0; o < P.K; o++) {
                            double dx = sX[3 * i] - P.obs[3 * o], dy = sX[3 * i + 1] - P.obs[3 * o + 1];
                            double rr = sqrt(dx * dx + dy * dy), n0 = dx / rr, n1 = dy / rr;
                            int sl = k * NH + P.o_ob + i * P.K + o;
                            double sg = el ? 1.0 / (S[sl] / Z[sl] + EL[sl] / (rho - Z[sl])) : Z[sl] / S[sl], zz = Z[sl] / rr;
                            e00 += sg * n0 * n0 - zz * (1 - n0 * n0);
                            e01 += sg * n0 * n1 + zz * n0 * n1;
                            e11 += sg * n1 * n1 - zz * (1 - n1 * n1);
                        }
                        int a = 3 * i;
                        sP[a * NX + a] += e00; sP[a * NX + a + 1] += e01; sP[(a + 1) * NX + a] += e01; sP[(a + 1) * NX + a + 1] += e11;
                    }
                }
                __syncthreads();
                // ---- Cholesky of Quu in LDS (right-looking); pivot test against the original diagonal
                for (int j = 0; j < NU; j++) {
                    if (tid == 0) {
                        double d = sQuu[j * NU + j];
                        if (!(d > 1e-9 * fabs(sD0[j])) || !(d > 0.0)) { sFail = 1; d = 1.0; }
                        sQuu[j * NU + j] = sqrt(d);
                    }
                    __syncthreads();
                    double dj = sQuu[j * NU + j];
                    for (int i = j + 1 + tid; i < NU; i += TPB) sQuu[i * NU + j] /= dj;
                    __syncthreads();
                    const int rem = NU - j - 1;
                    for (int e = tid; e < rem * rem; e += TPB) {
                        int i = j + 1 + e / rem, c = j + 1 + e % rem;
                        if (c <= i) sQuu[i * NU + c] -= sQuu[i * NU + j] * sQuu[c * NU + j];
                    }
                    __syncthreads();
                }
                if (sFail) { ok = false; kf = k; break; }
                // ---- Y = L^-1 [Qux | qu]: one lane per column, forward substitution
                for (int c = tid; c <= NX; c += TPB) {
                    double *col = (c < NX) ? (sQux + c) : sQu;
                    const int ld = (c < NX) ? NX : 1;
                    for (int r = 0; r < NU; r++) {
                        double a = col[r * ld];
                        for (int t = 0; t < r; t++) a -= sQuu[r * NU + t] * col[t * ld];
                        col[r * ld] = a / sQuu[r * NU + r];
                    }
                }
                __syncthreads();
                // ---- P_k = sym(Qxx) - Y^T Y ; p_k = qx - Y^T y   (k >= 1)
                if (k >= 1) {
                    for (int e = tid; e < NX * NX; e += TPB) {
                        int r = e / NX, c = e - r * NX;
                        if (r <= c) {
                            double acc = 0.0;
                            for (int t = 0; t < NU; t++) acc += sQux[t * NX + r] * sQux[t * NX + c];
                            double v = 0.5 * (sP[r * NX + c] + sP[c * NX + r]) - acc;
                            sP[r * NX + c] = v; sP[c * NX + r] = v;
                        }
                    }
                    for (int r = tid; r < NX; r += TPB) {
                        double a = sQx[r];
                        for (int t = 0; t < NU; t++) a -= sQux[t * NX + r] * sQu[t];
                        sPv[r] = a;
                    }
                }
                __syncthreads();
                // ---- K = -L^-T Y, kff = -L^-T y: one lane per column, back substitution; gains to HBM
                for (int c = tid; c <= NX; c += TPB) {
                    double *col = (c < NX) ? (sQux + c) : sQu;
                    const int ld = (c < NX) ? NX : 1;
                    for (int r = NU - 1; r >= 0; r--) {
                        double a = col[r * ld];
                        for (int t = r + 1; t < NU; t++) a += sQuu[t * NU + r] * col[t * ld];
                        a = -a / sQuu[r * NU + r];
                        col[r * ld] = a;
                        if (c < NX) KG[(size_t)k * NU * NX + r * NX + c] = a; else KFF[k * NU + r] = a;
                    }
                }
                __syncthreads();
            }
            if (ok) break;
            __syncthreads();
            if (tid == 0) sFail = 0;
            __syncthreads();
            ntry++;
            if (delta == 0.0) delta = (delta_last == 0.0) ? 1e-4 : fmax(1e-20, delta_last / 3.0);
            else delta *= (delta_last == 0.0) ? 100.0 : NMPC_SHIFT_ESCALATION;
            if (delta > 1e20) break;
            kst = NMPC_RESUME_STAGE(kf, N);
        }
        if (!ok) { if (n_cold < NMPC_COLD_RETRIES) { do_restart(true); iter++; continue; } status = NMPC_STATUS_NUMERIC; break; }
        if (delta > 0.0) delta_last = delta;
        need_shift = delta > 0.0 && (ntry > 0 || delta > 1e-6);

        // ================= forward sweep (serial over stages; dx_k lives in LDS)
        for (int c = tid; c < NX; c += TPB) { sX[c] = 0.0; DX[c] = 0.0; }
        __syncthreads();
        for (int k = 0; k < N; k++) {
            for (int e = tid; e < NU * NX; e += TPB) sQux[e] = KG[(size_t)k * NU * NX + e];
            for (int i = tid; i < M_; i += TPB) {
                double s = SN[k * M_ + i], c = CS[k * M_ + i], v = U[k * NU + 2 * i];
                sSn[i] = s; sCs[i] = c; sA[i] = -T * v * s; sB[i] = T * v * c;
            }
            for (int c = tid; c < NX; c += TPB) sCk[c] = Cd[k * NX + c];
            __syncthreads();
            for (int r = tid; r < NU; r += TPB) {
                double a = KFF[k * NU + r];
                for (int c = 0; c < NX; c++) a += sQux[r * NX + c] * sX[c];
                sU[r] = a; DU[k * NU + r] = a;
            }
            __syncthreads();
            for (int i = tid; i < M_; i += TPB) {
                double d0 = sX[3 * i] + sA[i] * sX[3 * i + 2] + T * sCs[i] * sU[2 * i] - sCk[3 * i];
                double d1 = sX[3 * i + 1] + sB[i] * sX[3 * i + 2] + T * sSn[i] * sU[2 * i] - sCk[3 * i + 1];
                double d2 = sX[3 * i + 2] + T * sU[2 * i + 1] - sCk[3 * i + 2];
                sPb[3 * i] = d0; sPb[3 * i + 1] = d1; sPb[3 * i + 2] = d2;
            }
            __syncthreads();
            for (int c = tid; c < NX; c += TPB) { sX[c] = sPb[c]; DX[(k + 1) * NX + c] = sPb[c]; }
            __syncthreads();
        }

        // ================= slack / dual steps, fraction to the boundary (IPOPT eq. 15)
        double a_p = 1.0, a_d = 1.0, mult_max = 0.0;
        for (int k = 0; k <= N; k++) {
            const double *x = X + k * NX, *dxk = DX + k * NX, *duk = DU + (k < N ? k : 0) * NU;
            for (int s = tid; s < NH; s += TPB) {
                int e = k * NH + s;
                if (!slot_active(P, k, s)) { DS[e] = 0.0; DZ[e] = 0.0; continue; }
                double sv = S[e], zv = Z[e];
                double ds, dz;
                if (ELSLOT(s)) {
                    const double tv = EL[e], rz = rho - zv, D_ = sv / zv + tv / rz, q_ = H[e] - mu / zv + mu / rz;
                    dz = -(slot_jd<M_>(P, sPi, sPj, k, s, x, dxk, duk) + q_) / D_;
                    ds = (mu - sv * zv - sv * dz) / zv;
                    const double dt = (mu - tv * rz + tv * dz) / rz;
                    DEL[e] = dt;
                    if (dt < 0.0) a_p = fmin(a_p, -tau * tv / dt);
                    if (dz > 0.0) a_d = fmin(a_d, tau * rz / dz);
                } else {
                ds = slot_jd<M_>(P, sPi, sPj, k, s, x, dxk, duk) + (H[e] - sv);
                dz = (mu - sv * zv - zv * ds) / sv;
                }
                DS[e] = ds; DZ[e] = dz;
                if (s >= P.o_pr) mult_max = fmax(mult_max, fabs(zv + dz));
                if (ds < 0.0) a_p = fmin(a_p, -tau * sv / ds);
                if (dz < 0.0) a_d = fmin(a_d, -tau * zv / dz);
            }
        }
        a_p = blk_min<TPB>(a_p, sRed);
        a_d = blk_min<TPB>(a_d, sRed);
        __syncthreads();

        // ================= multipliers of the QP: adjoint recursion, serial over stages
        //   lam+_k = A_k^T lam+_{k+1} - (grad f_k + W_k dx_k + W_xu du_k + delta dx_k) + Jx_k^T (z + dz)_k
        for (int k = N; k >= 1; k--) {
            for (int i = tid; i < M_; i += TPB) {
                const double *x = X + k * NX, *dx = DX + k * NX;
                const double *zk = Z + k * NH, *dzk = DZ + k * NH;
                double j0, j1, j2;
                jxT_robot<M_>(P, k, i, x, [&](int s) { return zk[s] + dzk[s]; }, j0, j1, j2);
                double l0 = j0, l1 = j1, l2 = j2;
                if (k < N) {
                    l0 -= 2 * P.q[0] * (x[3 * i] - sXs[3 * i]) + 2 * P.q[0] * dx[3 * i];
                    l1 -= 2 * P.q[1] * (x[3 * i + 1] - sXs[3 * i + 1]) + 2 * P.q[1] * dx[3 * i + 1];
                    l2 -= 2 * P.q[2] * (x[3 * i + 2] - sXs[3 * i + 2]) + 2 * P.q[2] * dx[3 * i + 2];
                    // exact-Hessian terms of the pair / obstacle rows: -z hess(h) dx
                    double w0_ = 0.0, w1_ = 0.0;
#pragma unroll
                    for (int j = 0; j < (P.pairs ? M_ : 0); j++) {
                        if (j == i) continue;
                        int q = (i < j) ? pair_index<M_>(i, j) : pair_index<M_>(j, i);
                        double zz = zk[P.o_pr + q];
                        w0_ -= 2 * zz * (dx[3 * i] - dx[3 * j]);
                        w1_ -= 2 * zz * (dx[3 * i + 1] - dx[3 * j + 1]);
                    }
                    for (int o = 0; o < P.K; o++) {
                        double ex = x[3 * i] - P.obs[3 * o], ey = x[3 * i + 1] - P.obs[3 * o + 1];
                        double rr = sqrt(ex * ex + ey * ey), n0 = ex / rr, n1 = ey / rr, zz = zk[P.o_ob + i * P.K + o] / rr;
                        double nd = n0 * dx[3 * i] + n1 * dx[3 * i + 1];
                        w0_ -= zz * (dx[3 * i] - n0 * nd);
                        w1_ -= zz * (dx[3 * i + 1] - n1 * nd);
                    }
                    l0 -= w0_; l1 -= w1_;
                    const double *ln = LAMN + (k + 1) * NX + 3 * i;
                    double v = U[k * NU + 2 * i];
                    double a = -T * v * SN[k * M_ + i], b = T * v * CS[k * M_ + i];
                    l0 += ln[0]; l1 += ln[1];
                    l2 += ln[2] + a * ln[0] + b * ln[1] - HTT[k * M_ + i] * dx[3 * i + 2] - HVT[k * M_ + i] * DU[k * NU + 2 * i];
                }
                LAMN[k * NX + 3 * i] = l0; LAMN[k * NX + 3 * i + 1] = l1; LAMN[k * NX + 3 * i + 2] = l2;
            }
            __syncthreads();
        }

        // ================= l1 merit backtracking line search
        double phi0, th0;
        eval_trial(0.0, mu, phi0, th0);
        double dphi = 0.0;
        for (int it = tid; it < (N + 1) * M_; it += TPB) {
            int k = it / M_, i = it - k * M_;
            if (k >= 1 && k < N)
#pragma unroll
                for (int d = 0; d < 3; d++) dphi += 2 * P.q[d] * (X[k * NX + 3 * i + d] - sXs[3 * i + d]) * DX[k * NX + 3 * i + d];
            if (k < N)
#pragma unroll
                for (int d = 0; d < 2; d++) dphi += 2 * P.r[d] * U[k * NU + 2 * i + d] * DU[k * NU + 2 * i + d];
        }
        for (int e = tid; e < (N + 1) * NH; e += TPB) {
            int k = e / NH, s = e - k * NH;
            if (slot_active(P, k, s)) { dphi -= mu * DS[e] / S[e]; if (ELSLOT(s)) dphi += (rho - mu / EL[e]) * DEL[e]; }
        }
        dphi = blk_sum<TPB>(dphi, sRed);
        for (int e = NX + tid; e < (N + 1) * NX; e += TPB) mult_max = fmax(mult_max, fabs(LAMN[e]));
        mult_max = blk_max<TPB>(mult_max, sRed);
        if (th0 > 0.0) {
            // Nocedal-Wright (18.36), rho = 0.1; capped by the multiplier norm it cannot exceed in exact arithmetic
            double nut = fmin(dphi / ((1.0 - 0.1) * th0), mult_max / (1.0 - 0.1));
            nu_pen = fmax(1.0, 0.5 * nu_pen);
            if (nu_pen < nut) nu_pen = nut + 1.0;
        }
        const double D = dphi - nu_pen * th0;
        double alpha = a_p;
        if (mh_mu != mu || mh_nu != nu_pen) { mcount = 0; mh_mu = mu; mh_nu = nu_pen; }
        const double m0 = phi0 + nu_pen * th0;
        double mref = m0;
        if (mcount > 0) mref = fmax(mref, mh0);
        if (mcount > 1) mref = fmax(mref, mh1);
        if (mcount > 2) mref = fmax(mref, mh2);
        mh2 = mh1; mh1 = mh0; mh0 = m0; if (mcount < 3) mcount++;
        for (int ls = 0; ls < 30; ls++) {
            double phit, tht;
            eval_trial(alpha, mu, phit, tht);
            if (phit + nu_pen * tht <= mref + 1e-4 * alpha * D + 1e-13 * fabs(phi0)) break;
            if (ls < 29) alpha *= 0.5;
        }
        a_d = fmin(a_d, alpha);      // the duals never step further than the primal variables actually moved
        n_tiny = (alpha < 1e-10) ? n_tiny + 1 : 0;
        __syncthreads();
        // ================= accept: primal / slack step alpha, dual step a_d with safeguard (IPOPT eq. 16)
        for (int e = tid; e < (N + 1) * NX; e += TPB) {
            X[e] += alpha * DX[e];
            if (e >= NX) LAM[e] += alpha * (LAMN[e] - LAM[e]);
        }
        for (int e = tid; e < N * NU; e += TPB) U[e] += alpha * DU[e];
        for (int e = tid; e < (N + 1) * NH; e += TPB) {
            int k = e / NH, s = e - k * NH;
            if (slot_active(P, k, s)) {
                double sv = S[e] + alpha * DS[e];
                double zv = Z[e] + a_d * DZ[e];
                S[e] = sv;
                double hi = 1e10 * mu / sv;
                if (ELSLOT(s)) { const double tn = EL[e] + alpha * DEL[e]; EL[e] = tn; hi = fmin(hi, rho - mu / (1e10 * tn)); }
                Z[e] = fmin(fmax(zv, mu / (1e10 * sv)), hi);
            }
        }
        __syncthreads();
        f = eval_point();
        __syncthreads();
        iter++;
        if (n_tiny >= 5) {
            if (n_restart >= 3) { if (n_cold < NMPC_COLD_RETRIES) { do_restart(true); continue; } status = NMPC_STATUS_STALLED; break; }
            // barrier restart from the current primal point (restoration in miniature, see the oracle)
            n_restart++; n_tiny = 0;
            mu = fmax(mu, P.mu_init);
            do_restart(false);
        }
    }

    // ---- write back sol['x'] = [vec(X); vec(U)] (C6:436,440)
    __syncthreads();
    for (int e = tid; e < (N + 1) * NX; e += TPB) wo[e] = X[e];
    for (int e = tid; e < N * NU; e += TPB) wo[(size_t)(N + 1) * NX + e] = U[e];
    if (tid == 0) {
        if (obj_out) obj_out[inst] = f;
        if (status_out) status_out[inst] = status;
        if (iters_out) iters_out[inst] = iter;
        if (kkt_out) kkt_out[inst] = kkt;
    }
}

// ------------------------------------------------------------------------------------------
// f and g in the reference's row order (C6:278,314,318-331): streaming, one thread per (instance, stage)
template <int M_>
__global__ __launch_bounds__(256) void eval_kernel(const KParams P, int B, const double *__restrict__ p_in, const double *__restrict__ w,
                                                    double *__restrict__ f_out, double *__restrict__ g_out)
{
    constexpr int NX = Geo<M_>::NX, NU = Geo<M_>::NU, NP = Geo<M_>::NP;
    const int N = P.N;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)B * (N + 1);
    if (gid >= total) return;
    const int b = (int)(gid / (N + 1)), k = (int)(gid - (long)b * (N + 1));   // k = N handles the initial block
    const double *X = w + (size_t)b * P.nvar, *U = X + (size_t)(N + 1) * NX, *pp = p_in + (size_t)b * 2 * NX;
    double *g = g_out ? g_out + (size_t)b * P.ng : nullptr;
    if (k == N) {
        if (g) {
            for (int c = 0; c < NX; c++) g[c] = X[c] - pp[c];
            if (P.pad_rows && P.pairs) for (int c = 0; c < NP; c++) g[NX + c] = P.pad_value;
        }
        return;
    }
    const double *x = X + (size_t)k * NX, *xn = x + NX, *u = U + (size_t)k * NU;
    double fs = 0.0;
    double *gk = g ? g + P.rows0 + (size_t)k * P.rowsk : nullptr;
#pragma unroll
    for (int i = 0; i < M_; i++) {
        double s, c;
        sincos(x[3 * i + 2], &s, &c);
        double e0 = x[3 * i] - pp[NX + 3 * i], e1 = x[3 * i + 1] - pp[NX + 3 * i + 1], e2 = x[3 * i + 2] - pp[NX + 3 * i + 2];
        fs += P.q[0] * e0 * e0 + P.q[1] * e1 * e1 + P.q[2] * e2 * e2 + P.r[0] * u[2 * i] * u[2 * i] + P.r[1] * u[2 * i + 1] * u[2 * i + 1];
        if (gk) {
            gk[3 * i] = xn[3 * i] - (x[3 * i] + P.T * u[2 * i] * c);
            gk[3 * i + 1] = xn[3 * i + 1] - (x[3 * i + 1] + P.T * u[2 * i] * s);
            gk[3 * i + 2] = xn[3 * i + 2] - (x[3 * i + 2] + P.T * u[2 * i + 1]);
        }
    }
    if (gk) {
        int o = NX;
        for (int i = 0; i < (P.pairs ? M_ : 0); i++)
            for (int j = i + 1; j < M_; j++) {
                double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1];
                gk[o++] = dx * dx + dy * dy;
            }
        for (int i = 0; i < M_; i++)
            for (int q = 0; q < P.K; q++) {
                double dx = x[3 * i] - P.obs[3 * q], dy = x[3 * i + 1] - P.obs[3 * q + 1];
                gk[o++] = sqrt(dx * dx + dy * dy) - P.robdim - P.obs[3 * q + 2];
            }
    }
    if (f_out) atomicAdd(&f_out[b], fs);   // <= N adds per instance; order-dependent in the last bits only
}

// warm-start shift (C6:160-169,460-465) + optional plant step (casadi_test.py:17-26)
template <int M_>
__global__ __launch_bounds__(256) void shift_kernel(const KParams P, int B, const double *p_in, const double *__restrict__ w_in,
                                                     double *__restrict__ w_next, double *x0_next, int x0_stride, const int32_t *__restrict__ keep_status)
{
    // keep_status (nmpc_step_batch): an instance whose solve ended with a numerical failure (2) or an infeasible x0 (3) — possibly a non-finite
    // iterate — keeps its guess and its x0: nothing of it is overwritten, the caller can still apply a fallback
    constexpr int NX = Geo<M_>::NX, NU = Geo<M_>::NU;
    const int N = P.N;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)B * P.nvar;
    if (gid < total) {
        const int b = (int)(gid / P.nvar), e = (int)(gid - (long)b * P.nvar);
        const double *src = w_in + (size_t)b * P.nvar;
        const int nX = (N + 1) * NX;
        double v;
        if (e < nX) {
            int k = e / NX, c = e - k * NX;
            v = (k < N) ? src[(k + 1) * NX + c] : src[(N - 1) * NX + c];   // appended row is row N-1 (C6:465)
        } else {
            int eu = e - nX, k = eu / NU, c = eu - k * NU;
            v = (k < N - 1) ? src[nX + (k + 1) * NU + c] : src[nX + (N - 1) * NU + c];
        }
        const bool keep = keep_status && (keep_status[b] == NMPC_STATUS_NUMERIC || keep_status[b] == NMPC_STATUS_INFEASIBLE_X0);
        if (!keep) w_next[gid] = v;
    }
    if (x0_next && gid < (long)B * M_) {
        const int b = (int)(gid / M_), i = (int)(gid - (long)b * M_);
        if (keep_status && (keep_status[b] == NMPC_STATUS_NUMERIC || keep_status[b] == NMPC_STATUS_INFEASIBLE_X0)) return;
        const double *x0 = p_in + (size_t)b * 2 * NX + 3 * i;
        const double *u = w_in + (size_t)b * P.nvar + (size_t)(N + 1) * NX + 2 * i;
        double s, c;
        sincos(x0[2], &s, &c);
        // x0_next may be the x0 part of p_in itself (nmpc_step_batch: stride 2 n_x): this thread reads its robot's three entries
        // before it writes them, and no other thread touches them
        const double n0 = x0[0] + P.T * u[0] * c, n1 = x0[1] + P.T * u[0] * s, n2 = x0[2] + P.T * u[1];
        double *o = x0_next + (size_t)b * x0_stride + 3 * i;
        o[0] = n0; o[1] = n1; o[2] = n2;
    }
}

// ------------------------------------------------------------------------------------------
// host-side launchers (called from nmpc_api.cpp)
template <int M_> static hipError_t launch_solve_m(const KParams &P, int B, const double *p, const double *w0, double *w_out, double *obj,
                                                   int32_t *status, int32_t *iters, double *kkt, double *ws, hipStream_t st)
{
    constexpr int TPB = (M_ <= 6) ? 64 : 128;
    hipLaunchKernelGGL((solve_kernel<M_, TPB>), dim3(B), dim3(TPB), 0, st, P, p, w0, w_out, obj, status, iters, kkt, ws);
    return hipGetLastError();
}
template <int M_> static hipError_t launch_eval_m(const KParams &P, int B, const double *p, const double *w, double *f, double *g, hipStream_t st)
{
    if (f) { hipError_t e = hipMemsetAsync(f, 0, sizeof(double) * (size_t)B, st); if (e != hipSuccess) return e; }
    long total = (long)B * (P.N + 1);
    hipLaunchKernelGGL((eval_kernel<M_>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, P, B, p, w, f, g);
    return hipGetLastError();
}
template <int M_> static hipError_t launch_shift_m(const KParams &P, int B, const double *p, const double *w_in, double *w_next, double *x0n, int x0_stride, const int32_t *keep_status, hipStream_t st)
{
    long total = (long)B * P.nvar;
    hipLaunchKernelGGL((shift_kernel<M_>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, P, B, p, w_in, w_next, x0n, x0_stride ? x0_stride : Geo<M_>::NX, keep_status);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Odometry front-end (C2:18-37): per robot, wheel-odometry pose in the robot's start frame -> pose in the global frame.
//   th = 2 asin(q_z); phi = th + th_init; [x y] = R(th_init) [x_r y_r] + [x_init y_init].   Streaming, 7 doubles in, 3 out.
__global__ __launch_bounds__(256) void odometry_kernel(long n, const double *__restrict__ odom, const double *__restrict__ init, double *__restrict__ pose, int wrap)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double xr = odom[4 * i], yr = odom[4 * i + 1], qz = odom[4 * i + 2];
    const double xi = init[3 * i], yi = init[3 * i + 1], thi = init[3 * i + 2];
    double s, c;
    sincos(thi, &s, &c);
    pose[3 * i] = (c * xr - s * yr) + xi;
    pose[3 * i + 1] = (s * xr + c * yr) + yi;
    double th = 2.0 * asin(qz);
    // modify() of the scripts without collision rows (AS/mpc_online_casadi.py:28-33): th in [-pi, 0) -> th + 2 pi
    if (wrap && th >= -3.14159265358979323846 && th < 0.0) th += 2.0 * 3.14159265358979323846;
    pose[3 * i + 2] = th + thi;
}
hipError_t launch_odometry(long n, const double *odom, const double *init, double *pose, int wrap, hipStream_t st)
{
    hipLaunchKernelGGL(odometry_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, odom, init, pose, wrap);
    return hipGetLastError();
}

// Permutation check of a dispatch-order hint (nmpc_solve_batch_ordered): count[o]++ for every entry, then any count != 1 (or an
// entry outside 0..B-1) raises *bad; the solve kernels ignore the hint when *bad != 0.  Three tiny launches on the call's stream.
__global__ __launch_bounds__(256) void order_count_kernel(int B, const int32_t *__restrict__ order, int32_t *__restrict__ count, int32_t *__restrict__ bad)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= B) return;
    const int o = order[g];
    if (o < 0 || o >= B) atomicOr(bad, 1);
    else atomicAdd(&count[o], 1);
}
__global__ __launch_bounds__(256) void order_verify_kernel(int B, const int32_t *__restrict__ count, int32_t *__restrict__ bad)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < B && count[g] != 1) atomicOr(bad, 1);
}
hipError_t launch_order_check(int B, const int32_t *order, int32_t *count, int32_t *bad, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(count, 0, sizeof(int32_t) * (size_t)(B + 1), st);      // count[B] is the flag
    if (e != hipSuccess) return e;
    const unsigned nb = (unsigned)((B + 255) / 256);
    hipLaunchKernelGGL(order_count_kernel, dim3(nb), dim3(256), 0, st, B, order, count, bad);
    hipLaunchKernelGGL(order_verify_kernel, dim3(nb), dim3(256), 0, st, B, count, bad);
    return hipGetLastError();
}

// Dispatch order of the next control period (nmpc_step_batch): the instances sorted by this period's iteration counts, longest first —
// a counting sort over the keys min(iters, 2047) by ONE workgroup (histogram in LDS, exclusive scan over descending keys, scatter).
// Instances with equal counts land in arbitrary relative order: the order only schedules, it never changes a result.
__global__ __launch_bounds__(1024) void order_by_iters_kernel(int B, const int32_t *__restrict__ iters, int32_t *__restrict__ order)
{
    constexpr int NB = 2048;
    __shared__ int hist[NB], offs[NB];
    const int t = threadIdx.x;
    for (int k = t; k < NB; k += 1024) hist[k] = 0;
    __syncthreads();
    for (int i = t; i < B; i += 1024) { int k = iters[i]; k = k < 0 ? 0 : (k > NB - 1 ? NB - 1 : k); atomicAdd(&hist[NB - 1 - k], 1); }      // bin 0 = the longest solves
    __syncthreads();
    // exclusive scan of hist (2 bins per thread; Hillis-Steele over the per-thread sums)
    const int a0 = hist[2 * t], a1 = hist[2 * t + 1];
    offs[t] = a0 + a1;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = (t >= d) ? offs[t - d] : 0;
        __syncthreads();
        offs[t] += v;
        __syncthreads();
    }
    const int base = offs[t] - (a0 + a1);
    __syncthreads();
    hist[2 * t] = base; hist[2 * t + 1] = base + a0;
    __syncthreads();
    for (int i = t; i < B; i += 1024) { int k = iters[i]; k = k < 0 ? 0 : (k > NB - 1 ? NB - 1 : k); order[atomicAdd(&hist[NB - 1 - k], 1)] = i; }
}
hipError_t launch_order_by_iters(int B, const int32_t *iters, int32_t *order, hipStream_t st)
{
    hipLaunchKernelGGL(order_by_iters_kernel, dim3(1), dim3(1024), 0, st, B, iters, order);
    return hipGetLastError();
}

#define NMPC_DISPATCH(M, CALL)                                                                                                    \
    switch (M) {                                                                                                                  \
    case 1: return CALL(1);                                                                                                       \
    case 2: return CALL(2);                                                                                                       \
    case 3: return CALL(3);                                                                                                       \
    case 4: return CALL(4);                                                                                                       \
    case 5: return CALL(5);                                                                                                       \
    case 6: return CALL(6);                                                                                                       \
    case 7: return CALL(7);                                                                                                       \
    case 8: return CALL(8);                                                                                                       \
    case 9: return CALL(9);                                                                                                       \
    case 10: return CALL(10);                                                                                                     \
    default: return hipErrorInvalidValue;                                                                                         \
    }

hipError_t launch_solve(const KParams &P, int m, int B, const double *p, const double *w0, double *w_out, double *obj, int32_t *status,
                        int32_t *iters, double *kkt, double *ws, hipStream_t st)
{
#define C_(M) launch_solve_m<M>(P, B, p, w0, w_out, obj, status, iters, kkt, ws, st)
    NMPC_DISPATCH(m, C_)
#undef C_
}
hipError_t launch_eval(const KParams &P, int m, int B, const double *p, const double *w, double *f, double *g, hipStream_t st)
{
#define C_(M) launch_eval_m<M>(P, B, p, w, f, g, st)
    NMPC_DISPATCH(m, C_)
#undef C_
}
hipError_t launch_shift(const KParams &P, int m, int B, const double *p, const double *w_in, double *w_next, double *x0n, int x0_stride, const int32_t *keep_status, hipStream_t st)
{
#define C_(M) launch_shift_m<M>(P, B, p, w_in, w_next, x0n, x0_stride, keep_status, st)
    NMPC_DISPATCH(m, C_)
#undef C_
}

}  // namespace nmpc
