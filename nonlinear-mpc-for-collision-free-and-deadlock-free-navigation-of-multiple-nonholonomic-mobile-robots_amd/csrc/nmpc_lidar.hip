// nmpc_lidar.hip — LIDAR-ray distance-state NMPC solve for gfx950 (include/nmpc_lidar.h).
//
// Reference blocks replaced (V4 = AllScripts/obs_avoid_static_first_scenario_v4.py; V3 = ..._v3.py is Nc = N, lw = 0):
//   NLP V4:78-151 (13 states per stage [x y theta d_1..d_10], move blocking U[:, min(k, Nc-1)], stage cost + 0.1 sum 1/d^2,
//   rows gx then gd with the 1-norm distance to the lidar points of stage 0), nlpsol('ipopt') V4:156-157 and the call V4:245.
//
// Mapping.  This NLP has ONE robot: the pose block is 3 x 3, the control block 2 x 2, and the R distance states of a stage are
// eliminated through their own linearised equality rows (d = ||p - pObs||_1), so the Newton system is a 3-state Riccati
// recursion on 5 x 5 matrices (the held control of the move-blocked stages rides along as two states).  One WAVEFRONT solves one
// instance: the phases that are parallel over the horizon (evaluation, optimality error, condensed stage blocks, step lengths,
// merit function, update: ~95 % of the memory accesses) give every lane its own stages / variables of the instance's
// component-major workspace in HBM/L2 (lane-adjacent addresses, wave reductions with v_readfirstlane so that every decision is a
// scalar branch); the Riccati recursion keeps one COLUMN of the 5 x 5 matrix per lane (DPP broadcasts, as the swarm kernel does
// for its 30 x 30 stage), the forward and adjoint recursions run uniformly on all lanes, out of LDS, on stage maps formed in
// parallel.  DESIGN.md 4.5 has the measurements behind each of these choices.  (A first version gave every LANE its own instance:
// 64 serial solves per wavefront whose every access waited on HBM, 0.5 k solves/s — slower than the CPU.)
// Algorithm: the interior-point iteration of nmpc_kernels.hip / the oracle (barrier rule, fraction to the boundary, non-monotone
// l1-merit search, inertia shift on the control diagonal, barrier restart).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include "../../include/nmpc_constants.h"      // cold-start retry and inertia constants shared with the main solver and the oracles
#include "../../include/nmpc_lidar.h"

namespace nmpc_lidar {

struct LParams {
    int32_t N, Nc, R, ns, max_iter, nvar, ng, np, n_ineq;      // n_ineq: finite bounds of the free variables (stages 1..N and the controls)
    double T, q[3], r[2], lw, tol, mu_init;
    // device copies of the caller's lbx / ubx in the solve kernel's layout: the state part component-major [3 + R][N + 1] with the pinned
    // stage 0 set to -+inf (no slack rows), then the controls [2 Nc]; lb0 / ub0 [3 + R]: the caller's stage-0 bounds (feasibility of x0)
    const double *lb, *ub, *lb0, *ub0;
    int64_t S;                  // instances the workspace holds (each `total` doubles, contiguous)
    // element offsets of the per-instance arrays: [3 + R][N + 1] V Vt dV SL ZL SU ZU; [R][N] eta etan Wd gdv; [3][N] lam; [2 Nc] the control ones
    int64_t oV, oU, olam, oeta, oSL, oZL, oSU, oZU, oSLu, oZLu, oSUu, oZUu, odV, odU, oetan, oVt, oUt, oWd, ogdv, total;
};

__device__ __forceinline__ double sgn(double a) { return (double)((a > 0.0) - (a < 0.0)); }

__device__ __forceinline__ double push_in(double v, double lo, double hi)
{
    const double bp = 1e-2;
    if (isfinite(lo) && isfinite(hi)) {
        double pu = fmin(bp * fmax(1.0, fabs(lo)), bp * (hi - lo)), qu = fmin(bp * fmax(1.0, fabs(hi)), bp * (hi - lo));
        return fmin(fmax(v, lo + pu), hi - qu);
    }
    if (isfinite(lo)) return fmax(v, lo + bp * fmax(1.0, fabs(lo)));
    if (isfinite(hi)) return fmin(v, hi - bp * fmax(1.0, fabs(hi)));
    return v;
}

// sum of log(s_i), s_i > 0, accumulated as (product of the mantissas, sum of the exponents): four instructions per term instead of a log
// (~90, a fifth of the two phases that need the barrier function), one log per wavefront at the end.  The mantissa product is renormalised
// after every term; a term <= 0 or NaN makes the result NaN like the log would.
struct LogSum {
    double m = 1.0, lo = INFINITY;
    int e = 0;
    __device__ __forceinline__ void add(double s_)
    {
        const double t = m * __builtin_amdgcn_frexp_mant(s_);
        e += __builtin_amdgcn_frexp_exp(s_) + __builtin_amdgcn_frexp_exp(t);
        m = __builtin_amdgcn_frexp_mant(t);
        lo = fmin(lo, s_);
    }
};

// reciprocal: v_rcp_f64 seed + two Newton steps (full fp64 accuracy; five dependent operations where the IEEE division chain has eleven — the
// Riccati stage of a lone wavefront is one long dependency chain with two pivots in it)
__device__ __forceinline__ double rcp_nr(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

// Divisions of the stage-parallel phases: v_rcp_f64 + one Newton step and a multiply instead of the IEEE division chain (round 4; as qdiv of
// nmpc_solve_common.h: <= 2.3e-15 relative, tools/rcp_probe.hip).  NMPC_LIDAR_FAST_DIV=0 restores the divisions (A/B).
#ifndef NMPC_LIDAR_FAST_DIV
#define NMPC_LIDAR_FAST_DIV 1
#endif
__device__ __forceinline__ double qdiv(double a, double b)
{
#if NMPC_LIDAR_FAST_DIV
    double r = __builtin_amdgcn_rcp(b);
    return a * fma(fma(-b, r, 1.0), r, r);
#else
    return a / b;
#endif
}

#ifndef NMPC_LIDAR_DPP
#define NMPC_LIDAR_DPP 1        // 1: column-per-lane Riccati stage (DPP broadcasts); 0: every lane walks the upper triangle (A/B)
#endif
// acc += acc[lane L_ of my row of 16 lanes] * nr — a column operation of the column-per-lane Riccati stage.  The s_nop gives the two wait states a DPP
// read needs after a VALU write of its source register whatever the compiler placed in front of the statement (its hazard recognizer does not
// look into inline assembly).
template <int L_> __device__ __forceinline__ void dpp_fmac(double &acc, double nr)
{
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(nr), "n"(L_));
}
template <int L_> __device__ __forceinline__ void dpp_fmac2(double &acc, double u, double nr)      // acc += u[lane L_ of my row] * nr
{
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(u), "v"(nr), "n"(L_));
}

// wave reductions whose result the compiler knows to be uniform (scalar control flow)
__device__ __forceinline__ double uni(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ double wsum_(double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return uni(v); }
__device__ __forceinline__ double wmax_(double v) { for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o)); return uni(v); }
__device__ __forceinline__ double wmin_(double v) { for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o)); return uni(v); }
__device__ __forceinline__ double wlogsum_(LogSum a)
{
    for (int o = 32; o > 0; o >>= 1) {
        const double t = a.m * __shfl_xor(a.m, o);
        a.e += __shfl_xor(a.e, o) + __builtin_amdgcn_frexp_exp(t);
        a.m = __builtin_amdgcn_frexp_mant(t);
        a.lo = fmin(a.lo, __shfl_xor(a.lo, o));
    }
    const double r = fma((double)a.e, 0.693147180559945309417, log(a.m));
    return uni((a.lo > 0.0) ? r : NAN);
}

// One wavefront per instance.  Data layout: every per-stage array of the instance's workspace is COMPONENT-MAJOR ([component][stage]),
// and every stage-parallel phase gives lane l the stages l, l + 64, ...: lane-adjacent addresses in every load (a stage-major layout made
// each of them touch 64 cache lines), and the loops over the components of a stage are unrolled at compile time (ray count = template
// parameter), their loads issued in chunks before the arithmetic: a phase costs a few memory round trips instead of one per component —
// a lone wavefront's iteration is bounded by exactly those round trips.  What the serial recursions over the horizon read
// per stage (condensed blocks, trig, pose, gains) is written straight into LDS by the phase that computes it; the recursions run
// uniformly on all lanes, lane 0 storing to LDS, and stage-parallel passes move their results (pose step, multipliers) from there.
#ifndef NMPC_LIDAR_UNROLL
#define NMPC_LIDAR_UNROLL 4      // stages of the forward / adjoint recursions unrolled together: their LDS operand reads issue as one batch
#endif
#ifndef NMPC_LIDAR_WAVES
#define NMPC_LIDAR_WAVES 1      // resident waves per SIMD the register budget of the default instantiation is set for (the ten-ray kernel also exists for 2: see nmpc_lidar_solve_batch).  Measured (B = 4096 / 1024): 1 -> 63.1 k / 38.4 k solves/s, 2 -> 49.2 k / 30.0 k: with 256 registers the 5 x 5 recursion and the phases around it spill into their hot loops (every reload a full vmcnt wait), and a lone wave iterates 1.9x faster than one of a pair — which is what the longest solve of a launch sees
#endif
#ifndef NMPC_LIDAR_CH
#define NMPC_LIDAR_CH 5         // components of a stage whose loads are issued together
#endif
// Phase boundary: the lane index passes through an empty asm, so nothing derived from it (the per-array element indices of a phase) can be hoisted out of the
// phase or out of the interior-point loop.  The compiler otherwise computes those indices once per solve, keeps ~100 of them live across the whole
// iteration and — inside a 256-register budget — spills them, reloading each at its use behind a full vmcnt wait.  NMPC_LIDAR_RELANE = 0 switches it off (A/B).
#ifndef NMPC_LIDAR_RELANE
#define NMPC_LIDAR_RELANE 1
#endif
#if NMPC_LIDAR_RELANE
#define RELANE() asm volatile("" : "+v"(lane))
#else
#define RELANE() do { } while (0)
#endif
#ifdef NMPC_LIDAR_PROFILE      // development: cycles per phase, returned in the first entries of w_out (tools/lidar_phase_profile.py)
#define LP(i) do { long long t_ = clock64(); prof[i] += t_ - tlast; tlast = t_; RELANE(); } while (0)
#else
#define LP(i) RELANE()
#endif
// Wave-uniform base pointer indexed by a 32-bit element index: the byte offset is formed in 32 bits and zero-extended, which the compiler addresses as
// `global_load v, v_offset, s[base:base+1]` (scalar base + 32-bit vector offset) — with a plain `double *` every access carries a 64-bit vector address
// (two registers and two or three address instructions each: 973 of them in this kernel).  The per-instance workspace and the bound arrays are far
// below 4 GB.  NMPC_LIDAR_SADDR = 0 keeps plain pointers (A/B).
#ifndef NMPC_LIDAR_SADDR
#define NMPC_LIDAR_SADDR 1
#endif
template <class T> struct UPtr {
    T *b;
    __device__ __forceinline__ T &operator[](int i) const
    {
#if NMPC_LIDAR_SADDR
        return *reinterpret_cast<T *>(reinterpret_cast<char *>(const_cast<typename std::remove_const<T>::type *>(b)) + (size_t)((uint32_t)i * (uint32_t)sizeof(T)));
#else
        return b[i];
#endif
    }
};
#define SV(off, k, c) wsb[(off) + (c) * NP1 + (k)]            // state-like arrays [3 + R][N + 1]
#define RV(off, k, m) wsb[(off) + (m) * N + ((k) - 1)]        // per-ray arrays [R][N], stages 1..N
#define LV(k, i) wsb[olam + (i) * N + ((k) - 1)]              // multipliers of the pose rows [3][N], stages 1..N
// R_ >= 0: ray count known at compile time (10 in V3 / V4); R_ = -1: run-time count up to NMPC_LIDAR_MAX_RAYS (loops unrolled to the
// maximum, the surplus predicated off with its loads clamped to a valid component)
template <int R_, int W_ = NMPC_LIDAR_WAVES>      // W_: resident waves per SIMD the register budget is set for (1: 512 registers, 2: 256)
__global__ __launch_bounds__(64, W_) void lidar_solve_kernel(const LParams P, int B, const double *__restrict__ p_in, const double *__restrict__ w0,
                                                          double *__restrict__ w_out, double *__restrict__ obj_out, int32_t *__restrict__ status_out,
                                                          int32_t *__restrict__ iters_out, double *__restrict__ kkt_out, double *__restrict__ ws)
{
    constexpr int RM = R_ >= 0 ? R_ : NMPC_LIDAR_MAX_RAYS, CM = 3 + RM, CH = NMPC_LIDAR_CH;
    const int b = blockIdx.x;
    int lane = threadIdx.x;      // re-read through RELANE() at every phase boundary
    if (b >= B) return;
    const int N = P.N, Nc = P.Nc, R = R_ >= 0 ? R_ : P.R, ns = 3 + R, NP1 = N + 1;
    const double T = P.T;
    const UPtr<double> wsb{ws + (size_t)b * (size_t)P.total};
    extern __shared__ double lsm[];
    double *SB = lsm;                              // [N+1][13]  Hxx(4) gx(3) hvt sin cos pose(3)
    double *SC = SB + (size_t)NP1 * 13;            // [Nc][16]   u(2) huu(2) gu(2) K(6) kff(2) du(2)
    double *SD = SC + (size_t)Nc * 16;             // [N+1][3]   pose step; the adjoint recursion overwrites it with lambda+
    double *SP = SD + (size_t)NP1 * 3;             // [R][2]     lidar points of stage 0 (16 ray slots)
    double *SF = SP + 2 * NMPC_LIDAR_MAX_RAYS;     // [N][12]    stage maps of the forward recursion, then the stage terms of the adjoint one
    const double *pp = p_in + (size_t)b * P.np, *wi = w0 + (size_t)b * P.nvar;
    double *wo = w_out + (size_t)b * P.nvar;
    const UPtr<const double> lbS{P.lb}, ubS{P.ub}, lbu{P.lb + (size_t)NP1 * ns}, ubu{P.ub + (size_t)NP1 * ns};      // component-major, stage 0 = -+inf
    int oV = (int)P.oV, oVt = (int)P.oVt, oU = (int)P.oU, oUt = (int)P.oUt;
    const int odV = (int)P.odV, odU = (int)P.odU, oSL = (int)P.oSL, oZL = (int)P.oZL, oSU = (int)P.oSU, oZU = (int)P.oZU, oSLu = (int)P.oSLu, oZLu = (int)P.oZLu,
              oSUu = (int)P.oSUu, oZUu = (int)P.oZUu, olam = (int)P.olam, oeta = (int)P.oeta, oetan = (int)P.oetan, oWd = (int)P.oWd, ogdv = (int)P.ogdv;
    const double xs0 = pp[3], xs1 = pp[4], xs2 = pp[5];
    const int nV = NP1 * ns;
    const double lw = P.lw, q0 = P.q[0], q1 = P.q[1], q2 = P.q[2];
    auto cof = [&](int k) { return k < Nc - 1 ? k : Nc - 1; };
    auto gdist = [&](int m, double x, double y, double &sx, double &sy) {
        double ax = x - SP[2 * m], ay = y - SP[2 * m + 1];
        sx = sgn(ax); sy = sgn(ay);
        return fabs(ax) + fabs(ay);
    };
// component / ray c of an unrolled loop: live?  and the index its (unconditional) loads use
#define C_ON(c) (R_ >= 0 || (c) < ns)
#define C_IX(c, c0) (C_ON(c) ? (c) : (c0))
#define M_ON(m) (R_ >= 0 || (m) < R)
#define M_IX(m, m0) (M_ON(m) ? (m) : (m0))
    // ---- load the start; X_0 (pose and scan) pinned to the parameters (V4:110-111); lidar points (V4:114-118)
    for (int k = lane; k <= N; k += 64)
        for (int c = 0; c < ns; c++) {
            SV(oV, k, c) = (k == 0) ? ((c < 3) ? pp[c] : pp[6 + (c - 3)]) : wi[(size_t)k * ns + c];
            if (k == 0) SV(odV, 0, c) = 0.0;
        }
    for (int e = lane; e < 2 * Nc; e += 64) wsb[oU + e] = wi[(size_t)NP1 * ns + e];
    for (int m = lane; m < R; m += 64) {
        double a = pp[2] + pp[6 + R + m], s, c;
        sincos(a, &s, &c);
        SP[2 * m] = pp[0] + pp[6 + m] * c; SP[2 * m + 1] = pp[1] + pp[6 + m] * s;
    }
    __syncthreads();
    {   // the pinned stage-0 variables must respect their own bounds
        double bad = 0.0;
        for (int c = lane; c < ns; c += 64) { double v = (c < 3) ? pp[c] : pp[6 + (c - 3)]; if (v < P.lb0[c] || v > P.ub0[c]) bad = 1.0; }
        if (wmax_(bad) > 0.0) {
            for (int k = lane; k <= N; k += 64) for (int c = 0; c < ns; c++) wo[(size_t)k * ns + c] = SV(oV, k, c);
            for (int e = lane; e < 2 * Nc; e += 64) wo[(size_t)NP1 * ns + e] = wsb[oU + e];
            if (lane == 0) {
                if (obj_out) obj_out[b] = NAN;
                if (status_out) status_out[b] = NMPC_STATUS_INFEASIBLE_X0;
                if (iters_out) iters_out[b] = 0;
                if (kkt_out) kkt_out[b] = INFINITY;
            }
            return;
        }
    }
    const int n_ineq = P.n_ineq;

    // objective, sum / max of the equality residuals at (oVx, oUx): one stage per lane; leaves sin / cos of the point in SB (the last
    // point evaluated in an iteration is the accepted one)
    auto eval_point = [&](int oVx, int oUx, double &th, double &ec) {
        double f = 0.0, t_ = 0.0, e_ = 0.0;
        for (int k = lane; k < N; k += 64) {
            const int j = cof(k);
            const double x = SV(oVx, k, 0), y = SV(oVx, k, 1), t = SV(oVx, k, 2), u0 = wsb[oUx + 2 * j], u1 = wsb[oUx + 2 * j + 1];
            const double xn = SV(oVx, k + 1, 0), yn = SV(oVx, k + 1, 1), tn = SV(oVx, k + 1, 2);
            double dk[RM > 0 ? RM : 1], dn[RM > 0 ? RM : 1];
            if (R_ > 0 || (R_ < 0 && R > 0)) {
#pragma unroll
                for (int m = 0; m < RM; m++) { dk[m] = SV(oVx, k, 3 + M_IX(m, 0)); dn[m] = SV(oVx, k + 1, 3 + M_IX(m, 0)); }
            }
            double s, c;
            sincos(t, &s, &c);
            SB[k * 13 + 8] = s; SB[k * 13 + 9] = c;
            const double c0 = xn - (x + T * u0 * c), c1 = yn - (y + T * u0 * s), c2 = tn - (t + T * u1);
            t_ += fabs(c0) + fabs(c1) + fabs(c2); e_ = fmax(e_, fmax(fabs(c0), fmax(fabs(c1), fabs(c2))));
            f += q0 * (x - xs0) * (x - xs0); f += q1 * (y - xs1) * (y - xs1); f += q2 * (t - xs2) * (t - xs2);
            f += P.r[0] * u0 * u0 + P.r[1] * u1 * u1;
#pragma unroll
            for (int m = 0; m < RM; m++)
                if (M_ON(m)) {
                    if (lw != 0.0) { double d = dk[m]; f += qdiv(lw, d * d); }
                    double sx, sy, e = dn[m] - gdist(m, xn, yn, sx, sy);
                    t_ += fabs(e); e_ = fmax(e_, fabs(e));
                }
        }
        th = wsum_(t_); ec = wmax_(e_);
        return wsum_(f);
    };

    double mu = P.mu_init, f = 0.0, th0 = 0.0, e_c = 0.0;
    int it = 0, n_tiny = 0, n_restart = 0, status = NMPC_STATUS_MAX_ITER;
#ifdef NMPC_LIDAR_PROFILE
    long long prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
#endif
    // cold-start retry (the restoration of last resort of the main solver, same constants; see oracle/lidar_oracle.c)
    int n_cold = 0, it_base = 0;
    bool cold = false;
    bool need_shift = false, restarting = false;
    double delta_last = 0.0, nu_pen = 1.0, kkt = INFINITY;
    double mh0 = 0, mh1 = 0, mh2 = 0, mh_mu = -1, mh_nu = -1;
    int mcount = 0, n_short = 0;      // n_short: successive steps the backtracking shortened (watchdog, include/nmpc_constants.h)

    auto cold_retry = [&]() { cold = true; n_cold++; it_base = it; restarting = true; mu = (n_cold == 1) ? P.mu_init : 10.0 * P.mu_init; n_tiny = 0; n_restart = 0; };
    for (;;) {      // (re)start of the barrier iteration
        if (cold) {       // the reference's cold start (V4:184-196): X_k = x0 (pose and scan), U = 0
            for (int k = 1 + lane; k <= N; k += 64) for (int c = 0; c < ns; c++) SV(oV, k, c) = SV(oV, 0, c);
            for (int e = lane; e < 2 * Nc; e += 64) wsb[oU + e] = 0.0;
            cold = false;
            __syncthreads();
        }
        for (int k = lane; k <= N; k += 64)
            for (int c = 0; c < ns; c++) {      // stage 0 is pinned: its bounds read -+inf here, its slots stay (1, 0)
                const double lo = lbS[c * NP1 + k], hi = ubS[c * NP1 + k], v = push_in(SV(oV, k, c), lo, hi);
                SV(oV, k, c) = v;
                const double sl = isfinite(lo) ? fmax(v - lo, 1e-12) : 1.0, su = isfinite(hi) ? fmax(hi - v, 1e-12) : 1.0;
                SV(oSL, k, c) = sl; SV(oZL, k, c) = isfinite(lo) ? mu / sl : 0.0;
                SV(oSU, k, c) = su; SV(oZU, k, c) = isfinite(hi) ? mu / su : 0.0;
            }
        for (int e = lane; e < 2 * Nc; e += 64) {
            const double u = push_in(wsb[oU + e], lbu[e], ubu[e]), sl = fmax(u - lbu[e], 1e-12), su = fmax(ubu[e] - u, 1e-12);
            wsb[oU + e] = u;
            wsb[oSLu + e] = sl; wsb[oZLu + e] = mu / sl; wsb[oSUu + e] = su; wsb[oZUu + e] = mu / su;
        }
        for (int e = lane; e < N * 3; e += 64) wsb[olam + e] = 0.0;
        for (int e = lane; e < N * R; e += 64) wsb[oeta + e] = 0.0;
        __syncthreads();
        f = eval_point(oV, oU, th0, e_c);
        __syncthreads();
        delta_last = 0.0; nu_pen = 1.0; need_shift = false; mcount = 0; restarting = false; n_short = 0;

        for (;;) {
            LP(0);
            // ---- A. optimality error (IPOPT eq. 5): one stage per lane
            double e_d = 0.0, e_h = 0.0, zsum = 0.0, lsum = 0.0, cmax = 0.0, cmin = INFINITY, pu0 = 0.0, pu1 = 0.0;
            auto ctrl_rows = [&](int j, double ru0, double ru1) {      // stationarity / complementarity of control j given its Lagrangian gradient
                for (int e = 0; e < 2; e++) {
                    const int o = 2 * j + e;
                    const double zl = wsb[oZLu + o], zu = wsb[oZUu + o], sl = wsb[oSLu + o], su = wsb[oSUu + o], u = wsb[oU + o];
                    const double ru = (e ? ru1 : ru0) - (zl - zu);
                    e_d = fmax(e_d, fabs(ru)); zsum += zl + zu;
                    cmax = fmax(cmax, fmax(sl * zl, su * zu)); cmin = fmin(cmin, fmin(sl * zl, su * zu));
                    e_h = fmax(e_h, fmax(fabs((u - lbu[o]) - sl), fabs((ubu[o] - u) - su)));
                }
            };
            // complementarity / bound-row residual of one variable (its slots hold (1, 0) where a bound is absent)
            auto comp = [&](double v, double lo, double hi, double sl, double zl, double su, double zu) {
                if (isfinite(lo)) { const double pz = sl * zl; zsum += zl; cmax = fmax(cmax, pz); cmin = fmin(cmin, pz); e_h = fmax(e_h, fabs((v - lo) - sl)); }
                if (isfinite(hi)) { const double pz = su * zu; zsum += zu; cmax = fmax(cmax, pz); cmin = fmin(cmin, pz); e_h = fmax(e_h, fabs((hi - v) - su)); }
            };
            for (int k = lane; k <= N; k += 64) {
                const int j = cof(k);
                const double u0 = (k < N) ? wsb[oU + 2 * j] : 0.0, u1 = (k < N) ? wsb[oU + 2 * j + 1] : 0.0;
                const double ln0 = (k < N) ? LV(k + 1, 0) : 0.0, ln1 = (k < N) ? LV(k + 1, 1) : 0.0, ln2 = (k < N) ? LV(k + 1, 2) : 0.0;
                const double sn = SB[k * 13 + 8], cs = SB[k * 13 + 9];
                if (k >= 1) {
                    const double x = SV(oV, k, 0), y = SV(oV, k, 1), t = SV(oV, k, 2);
                    double r0 = LV(k, 0), r1 = LV(k, 1), r2 = LV(k, 2);
                    double pv[3], plo[3], phi[3], psl[3], pzl[3], psu[3], pzu[3];
                    pv[0] = x; pv[1] = y; pv[2] = t;
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        plo[c] = lbS[c * NP1 + k]; phi[c] = ubS[c * NP1 + k];
                        psl[c] = SV(oSL, k, c); pzl[c] = SV(oZL, k, c); psu[c] = SV(oSU, k, c); pzu[c] = SV(oZU, k, c);
                    }
                    lsum += fabs(r0) + fabs(r1) + fabs(r2);
                    if (k < N) {
                        const double a = -T * u0 * sn, bq = T * u0 * cs;
                        r0 += 2 * q0 * (x - xs0) - ln0; r1 += 2 * q1 * (y - xs1) - ln1; r2 += 2 * q2 * (t - xs2) - ln2;
                        r2 -= a * ln0 + bq * ln1;
                    }
#pragma unroll
                    for (int m0 = 0; m0 < RM; m0 += CH)
                        if (R_ >= 0 || m0 < R) {
                            double et[CH], d[CH], lo[CH], hi[CH], sl[CH], zl[CH], su[CH], zu[CH];
#pragma unroll
                            for (int u = 0; u < CH; u++)
                                if (m0 + u < RM) {
                                    const int m = M_IX(m0 + u, m0);
                                    et[u] = RV(oeta, k, m); d[u] = SV(oV, k, 3 + m); lo[u] = lbS[(3 + m) * NP1 + k]; hi[u] = ubS[(3 + m) * NP1 + k];
                                    sl[u] = SV(oSL, k, 3 + m); zl[u] = SV(oZL, k, 3 + m); su[u] = SV(oSU, k, 3 + m); zu[u] = SV(oZU, k, 3 + m);
                                }
#pragma unroll
                            for (int u = 0; u < CH; u++)
                                if (m0 + u < RM && M_ON(m0 + u)) {
                                    double sx, sy; gdist(m0 + u, x, y, sx, sy);
                                    r0 -= et[u] * sx; r1 -= et[u] * sy;
                                    double rd = et[u] + ((k < N && lw != 0.0) ? qdiv(-2.0 * lw, d[u] * d[u] * d[u]) : 0.0);
                                    rd -= zl[u] - zu[u];
                                    e_d = fmax(e_d, fabs(rd)); lsum += fabs(et[u]);
                                    comp(d[u], lo[u], hi[u], sl[u], zl[u], su[u], zu[u]);
                                }
                        }
                    r0 -= pzl[0] - pzu[0]; r1 -= pzl[1] - pzu[1]; r2 -= pzl[2] - pzu[2];
                    e_d = fmax(e_d, fmax(fabs(r0), fmax(fabs(r1), fabs(r2))));
#pragma unroll
                    for (int c = 0; c < 3; c++) comp(pv[c], plo[c], phi[c], psl[c], pzl[c], psu[c], pzu[c]);
                }
                if (k < N) {       // control rows: stage k contributes to the Lagrangian gradient of control cof(k)
                    const double g0 = 2 * P.r[0] * u0 - T * (cs * ln0 + sn * ln1);
                    const double g1 = 2 * P.r[1] * u1 - T * ln2;
                    if (k < Nc - 1) ctrl_rows(j, g0, g1); else { pu0 += g0; pu1 += g1; }
                }
            }
            pu0 = wsum_(pu0); pu1 = wsum_(pu1);
            if (lane == 0) ctrl_rows(Nc - 1, pu0, pu1);       // the held control: the sum over its stages
            e_d = wmax_(e_d); e_h = wmax_(e_h); zsum = wsum_(zsum); lsum = wsum_(lsum); cmax = wmax_(cmax); cmin = wmin_(cmin);
            const double smax = 100.0;
            const double s_d = fmax(smax, (lsum + zsum) / (double)(N * ns + n_ineq)) / smax;
            const double s_c = fmax(smax, zsum / (double)(n_ineq > 0 ? n_ineq : 1)) / smax;
            const double E0 = uni(fmax(fmax(e_d / s_d, e_c), fmax(e_h, cmax / s_c)));
            kkt = E0;
            if (!(E0 == E0)) { if (n_cold < NMPC_COLD_RETRIES && it < P.max_iter) { cold_retry(); break; } status = NMPC_STATUS_NUMERIC; break; }
            if (E0 <= P.tol) { status = NMPC_STATUS_CONVERGED; break; }
            if (it >= P.max_iter) { status = NMPC_STATUS_MAX_ITER; break; }
            if (n_cold < NMPC_COLD_RETRIES && it - it_base >= NMPC_COLD_RETRY_ITERS) { cold_retry(); break; }
            const double mu_min = P.tol / 10.0;
            for (;;) {
                const double cm = fmax(fabs(cmax - mu), fabs(cmin - mu));
                const double Emu = fmax(fmax(e_d / s_d, e_c), fmax(e_h, cm / s_c));
                if (mu > mu_min && Emu <= 10.0 * mu) mu = fmax(mu_min, fmin(0.2 * mu, pow(mu, 1.5)));
                else break;
            }
            // wave-uniform values that live across phases go back to scalar registers (arithmetic on them runs on the vector ALU and would
            // leave them in vector registers — the ones the allocator then spills to scratch and reloads inside the hot loops)
            mu = uni(mu);
            const double tau = uni(fmax(0.99, 1.0 - mu));
            LP(1);

            // ---- B0. condensed stage blocks, one stage per lane, straight into LDS.  slot: v = mu/s - sigma (h - s), sigma = z/s
            for (int k = lane; k <= N; k += 64) {
                double *sb = SB + k * 13;
                const double x = SV(oV, k, 0), y = SV(oV, k, 1), t = SV(oV, k, 2);
                const double sn = sb[8], cs = sb[9];
                const double ln0 = (k < N) ? LV(k + 1, 0) : 0.0, ln1 = (k < N) ? LV(k + 1, 1) : 0.0, u0 = (k < N) ? wsb[oU + 2 * cof(k)] : 0.0;
                sb[10] = x; sb[11] = y; sb[12] = t;
                if (k < N) sb[7] = T * (ln0 * sn - ln1 * cs);
                else { sb[7] = 0.0; sb[8] = 0.0; sb[9] = 0.0; }
                if (k >= 1) {
                    double H0 = 0.0, H1 = 0.0, H2 = 0.0, hd[3] = {0, 0, 0}, g[3] = {0, 0, 0};
                    if (k < N) {
                        hd[0] = 2 * q0; hd[1] = 2 * q1; hd[2] = 2 * q2;
                        g[0] = 2 * q0 * (x - xs0); g[1] = 2 * q1 * (y - xs1); g[2] = 2 * q2 * (t - xs2);
                    }
                    // Hessian / gradient contribution of the bound slots of one variable
                    auto slots = [&](double v, double lo, double hi, double sl, double zl, double su, double zu, double &hs, double &gs) {
                        hs = 0.0; gs = 0.0;
                        if (isfinite(lo)) { const double sg = qdiv(zl, sl); hs += sg; gs -= qdiv(mu, sl) - sg * ((v - lo) - sl); }
                        if (isfinite(hi)) { const double sg = qdiv(zu, su); hs += sg; gs += qdiv(mu, su) - sg * ((hi - v) - su); }
                    };
                    {
                        const double pv[3] = {x, y, t};
                        double plo[3], phi[3], psl[3], pzl[3], psu[3], pzu[3];
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            plo[c] = lbS[c * NP1 + k]; phi[c] = ubS[c * NP1 + k];
                            psl[c] = SV(oSL, k, c); pzl[c] = SV(oZL, k, c); psu[c] = SV(oSU, k, c); pzu[c] = SV(oZU, k, c);
                        }
#pragma unroll
                        for (int c = 0; c < 3; c++) { double hs, gs; slots(pv[c], plo[c], phi[c], psl[c], pzl[c], psu[c], pzu[c], hs, gs); hd[c] += hs; g[c] += gs; }
                    }
#pragma unroll
                    for (int m0 = 0; m0 < RM; m0 += CH)
                        if (R_ >= 0 || m0 < R) {
                            double d[CH], lo[CH], hi[CH], sl[CH], zl[CH], su[CH], zu[CH];
#pragma unroll
                            for (int u = 0; u < CH; u++)
                                if (m0 + u < RM) {
                                    const int m = M_IX(m0 + u, m0);
                                    d[u] = SV(oV, k, 3 + m); lo[u] = lbS[(3 + m) * NP1 + k]; hi[u] = ubS[(3 + m) * NP1 + k];
                                    sl[u] = SV(oSL, k, 3 + m); zl[u] = SV(oZL, k, 3 + m); su[u] = SV(oSU, k, 3 + m); zu[u] = SV(oZU, k, 3 + m);
                                }
#pragma unroll
                            for (int u = 0; u < CH; u++)
                                if (m0 + u < RM && M_ON(m0 + u)) {
                                    const int m = m0 + u;
                                    const double v = d[u];
                                    double hs, gs;
                                    slots(v, lo[u], hi[u], sl[u], zl[u], su[u], zu[u], hs, gs);
                                    const bool cost = k < N && lw != 0.0;
                                    const double Wd = hs + (cost ? qdiv(6.0 * lw, v * v * v * v) : 0.0), gd = gs + (cost ? qdiv(-2.0 * lw, v * v * v) : 0.0);
                                    RV(oWd, k, m) = Wd; RV(ogdv, k, m) = gd;
                                    double sx, sy;
                                    const double re = gdist(m, x, y, sx, sy) - v, tq = gd + Wd * re;      // linearised row: dd = G dx + re
                                    g[0] += sx * tq; g[1] += sy * tq;
                                    H0 += Wd * sx * sx; H1 += Wd * sx * sy; H2 += Wd * sy * sy;
                                }
                        }
                    double H3 = hd[2];
                    if (k < N) H3 += T * u0 * (ln0 * cs + ln1 * sn);
                    sb[0] = H0 + hd[0]; sb[1] = H1; sb[2] = H2 + hd[1]; sb[3] = H3;
                    sb[4] = g[0]; sb[5] = g[1]; sb[6] = g[2];
                }
            }
            for (int o = lane; o < 2 * Nc; o += 64) {
                const int j = o >> 1, e = o & 1, cnt = (j < Nc - 1) ? 1 : N - Nc + 1;
                const double sl = wsb[oSLu + o], su = wsb[oSUu + o], zl = wsb[oZLu + o], zu = wsb[oZUu + o], u = wsb[oU + o];
                const double vl = qdiv(mu, sl) - qdiv(zl, sl) * ((u - lbu[o]) - sl), vu = qdiv(mu, su) - qdiv(zu, su) * ((ubu[o] - u) - su);
                SC[j * 16 + e] = u;
                SC[j * 16 + 2 + e] = cnt * 2 * P.r[e] + qdiv(zl, sl) + qdiv(zu, su);
                SC[j * 16 + 4 + e] = cnt * 2 * P.r[e] * u - (vl - vu);
            }
            __syncthreads();
            LP(2);
            // ---- B. Riccati sweep on z = (pose (3), held control (2)) with inertia correction; uniform on all lanes, stage in registers
            double delta = uni(need_shift ? fmax(1e-20, 0.25 * delta_last) : 0.0);
            int ntry = 0;
            bool ok;
            for (;;) {
                ok = true;
#if NMPC_LIDAR_DPP
                // Column per lane: lane q of every row of 16 lanes owns column q of the 5 x 5 cost-to-go matrix (q = 0..2 pose, 3..4 held control) and
                // lane 5 the linear term; its five rows are the registers m0..m4.  With At = [[A, B], [0, I]] the product P At is a combination
                // of columns — v_fmac_f64_dpp row_newbcast of the source column with a per-lane coefficient (the defect correction p - P c of
                // the linear term is one more coefficient of the same three instructions per row) —, At' (P At) a combination of rows, i.e. of
                // registers with uniform coefficients; the two control pivots are eliminated like in the swarm kernel (multiplier column by
                // DPP broadcast, normalised pivot row per lane).  ~150 instructions per stage for all entries at once, against ~260 when every
                // lane walks the 15 entries of the upper triangle by itself.
                const int q16 = lane & 15;
                const double i0_ = q16 == 0 ? 1.0 : 0.0, i1_ = q16 == 1 ? 1.0 : 0.0, i2_ = q16 == 2 ? 1.0 : 0.0, i3_ = q16 == 3 ? 1.0 : 0.0, i4_ = q16 == 4 ? 1.0 : 0.0,
                             i5_ = q16 == 5 ? 1.0 : 0.0;
                const bool isctl = q16 == 3 || q16 == 4, stor = lane < 16 && (q16 < 3 || q16 == 5);
                const int ko0 = q16 < 3 ? 6 + q16 : 12, ko1 = q16 < 3 ? 9 + q16 : 13;
                double m0, m1, m2, m3 = 0.0, m4 = 0.0;
                {
                    const double *sb = SB + N * 13;
                    m0 = sb[0] * i0_ + sb[1] * i1_ + sb[4] * i5_; m1 = sb[1] * i0_ + sb[2] * i1_ + sb[5] * i5_; m2 = sb[3] * i2_ + sb[6] * i5_;
                }
                auto fetch = [&](int k, double *o_, double *on_, double *oc_) {
                    const double *sb = SB + k * 13, *sc = SC + cof(k) * 16;
#pragma unroll
                    for (int q_ = 0; q_ < 13; q_++) o_[q_] = sb[q_];
                    on_[0] = sb[13 + 10]; on_[1] = sb[13 + 11]; on_[2] = sb[13 + 12];
#pragma unroll
                    for (int q_ = 0; q_ < 6; q_++) oc_[q_] = sc[q_];
                };
                auto lane_val = [&](double v, int l) {      // v of lane l, as a wave-uniform value
                    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
                };
                auto stage = [&](int k, const double *o_, const double *on_, const double *oc_) -> bool {
                    const int j = cof(k);
                    const double H0 = o_[0], H1 = o_[1], H2 = o_[2], H3 = o_[3], g0 = o_[4], g1 = o_[5], g2 = o_[6], hv = o_[7], s = o_[8], c = o_[9];
                    const double u0 = oc_[0], u1 = oc_[1], huu0 = oc_[2], huu1 = oc_[3], gu0 = oc_[4], gu1 = oc_[5];
                    const double a = -T * u0 * s, bq = T * u0 * c, Tc = T * c, Ts = T * s;
                    const double cd0 = on_[0] - (o_[10] + T * u0 * c), cd1 = on_[1] - (o_[11] + T * u0 * s), cd2 = on_[2] - (o_[12] + T * u1);
                    // columns: col2 += a col0 + bq col1, col3 += Tc col0 + Ts col1, col4 += T col2, linear term -= c0 col0 + c1 col1 + c2 col2
                    // (column 2 is a source before it is a target; columns 0 and 1 never change)
                    const double cf2 = T * i4_ - cd2 * i5_, cf0 = a * i2_ + Tc * i3_ - cd0 * i5_, cf1 = bq * i2_ + Ts * i3_ - cd1 * i5_;
                    dpp_fmac<2>(m0, cf2); dpp_fmac<2>(m1, cf2); dpp_fmac<2>(m2, cf2); dpp_fmac<2>(m3, cf2); dpp_fmac<2>(m4, cf2);
                    dpp_fmac<0>(m0, cf0); dpp_fmac<0>(m1, cf0); dpp_fmac<0>(m2, cf0); dpp_fmac<0>(m3, cf0); dpp_fmac<0>(m4, cf0);
                    dpp_fmac<1>(m0, cf1); dpp_fmac<1>(m1, cf1); dpp_fmac<1>(m2, cf1); dpp_fmac<1>(m3, cf1); dpp_fmac<1>(m4, cf1);
                    // rows: row4 += T row2 (the old row 2), row2 += a row0 + bq row1, row3 += Tc row0 + Ts row1
                    m4 = fma(T, m2, m4);
                    m2 = fma(a, m0, fma(bq, m1, m2));
                    m3 = fma(Tc, m0, fma(Ts, m1, m3));
                    if (k >= 1) {
                        m0 = fma(H0, i0_, fma(H1, i1_, fma(g0, i5_, m0)));
                        m1 = fma(H1, i0_, fma(H2, i1_, fma(g1, i5_, m1)));
                        m2 = fma(H3, i2_, fma(g2, i5_, m2));
                    }
                    m2 = fma(hv, i3_, m2); m3 = fma(hv, i2_, m3);
                    if (k <= Nc - 1) {        // the stage where control j is decided carries its whole diagonal / gradient
                        m3 = fma(huu0 + delta, i3_, fma(gu0, i5_, m3));
                        m4 = fma(huu1 + delta, i4_, fma(gu1, i5_, m4));
                        const double dv = lane_val(m3, 3), d1o = lane_val(m4, 4);
                        if (!(dv > 0.0)) return false;
                        const double rdv = rcp_nr(dv);
                        const double t3 = m3 * rdv, nt3 = -t3;        // pivot row 3 / pivot, per column
                        dpp_fmac<3>(m0, nt3); dpp_fmac<3>(m1, nt3); dpp_fmac<3>(m2, nt3); dpp_fmac<3>(m4, nt3);
                        const double d1 = lane_val(m4, 4);
                        if (!(d1 > 1e-9 * fabs(d1o)) || !(d1 > 0.0)) return false;
                        const double rd1 = rcp_nr(d1);
                        const double t4 = m4 * rd1, nt4 = -t4;
                        dpp_fmac<4>(m0, nt4); dpp_fmac<4>(m1, nt4); dpp_fmac<4>(m2, nt4);
                        // gains [K | kff] (columns 0..2 | column 5): second control -t4, first -(t3 - (M34 / d3) t4), M34 / d3 = t3 of column 4
                        double k0v = nt3;
                        dpp_fmac2<4>(k0v, t3, t4);
                        if (stor) { SC[j * 16 + ko0] = k0v; SC[j * 16 + ko1] = nt4; }
                        m3 = 0.0; m4 = 0.0;
                        if (isctl) { m0 = 0.0; m1 = 0.0; m2 = 0.0; }
                    }
                    return true;
                };
                for (int k0 = N - 1; k0 >= 0; k0 -= 2) {
                    double oa[13], na[3], ca[6], ob[13], nb[3], cb[6];
                    fetch(k0, oa, na, ca); fetch(k0 >= 1 ? k0 - 1 : 0, ob, nb, cb);
                    if (!stage(k0, oa, na, ca)) { ok = false; break; }
                    if (k0 >= 1 && !stage(k0 - 1, ob, nb, cb)) { ok = false; break; }
                }
#else
                // Cost-to-go z' P z + 2 p' z kept as the 15 entries of its upper triangle: with At = [[A, B], [0, I]] (A = I + (a, bq) in column
                // theta, B = [[T c, 0], [T s, 0], [0, T]]) the columns w_j = P At e_j and then M = At' P At are formed for j >= i only — 30
                // multiply-adds instead of the 50 + 20 of the full product followed by a symmetrisation.
                double P00, P01, P02, P03 = 0.0, P04 = 0.0, P11, P12, P13 = 0.0, P14 = 0.0, P22, P23 = 0.0, P24 = 0.0, P33 = 0.0, P34 = 0.0, P44 = 0.0;
                double p0, p1, p2, p3 = 0.0, p4 = 0.0;
                P00 = SB[N * 13]; P01 = SB[N * 13 + 1]; P11 = SB[N * 13 + 2]; P22 = SB[N * 13 + 3]; P02 = 0.0; P12 = 0.0;
                p0 = SB[N * 13 + 4]; p1 = SB[N * 13 + 5]; p2 = SB[N * 13 + 6];
                // operands of the stages: read for two stages at a time, before the gain stores of the first (the compiler keeps LDS reads behind
                // those): one exposed LDS round trip per pair
                auto fetch = [&](int k, double *o_, double *on_, double *oc_) {
                    const double *sb = SB + k * 13, *sc = SC + cof(k) * 16;
#pragma unroll
                    for (int q_ = 0; q_ < 13; q_++) o_[q_] = sb[q_];
                    on_[0] = sb[13 + 10]; on_[1] = sb[13 + 11]; on_[2] = sb[13 + 12];
#pragma unroll
                    for (int q_ = 0; q_ < 6; q_++) oc_[q_] = sc[q_];
                };
                auto stage = [&](int k, const double *o_, const double *on_, const double *oc_) -> bool {
                    const int j = cof(k);
                    const double H0 = o_[0], H1 = o_[1], H2 = o_[2], H3 = o_[3], g0 = o_[4], g1 = o_[5], g2 = o_[6], hv = o_[7], s = o_[8], c = o_[9];
                    const double u0 = oc_[0], u1 = oc_[1], huu0 = oc_[2], huu1 = oc_[3], gu0 = oc_[4], gu1 = oc_[5];
                    const double a = -T * u0 * s, bq = T * u0 * c, Tc = T * c, Ts = T * s;
                    const double cd0 = on_[0] - (o_[10] + T * u0 * c), cd1 = on_[1] - (o_[11] + T * u0 * s), cd2 = on_[2] - (o_[12] + T * u1);
                    const double pb0 = p0 - (P00 * cd0 + P01 * cd1 + P02 * cd2), pb1 = p1 - (P01 * cd0 + P11 * cd1 + P12 * cd2),
                                 pb2 = p2 - (P02 * cd0 + P12 * cd1 + P22 * cd2), pb3 = p3 - (P03 * cd0 + P13 * cd1 + P23 * cd2),
                                 pb4 = p4 - (P04 * cd0 + P14 * cd1 + P24 * cd2);
                    const double w20 = a * P00 + bq * P01 + P02, w21 = a * P01 + bq * P11 + P12, w22 = a * P02 + bq * P12 + P22;
                    const double w30 = Tc * P00 + Ts * P01 + P03, w31 = Tc * P01 + Ts * P11 + P13, w32 = Tc * P02 + Ts * P12 + P23, w33 = Tc * P03 + Ts * P13 + P33;
                    const double w40 = T * P02 + P04, w41 = T * P12 + P14, w42 = T * P22 + P24, w43 = T * P23 + P34, w44 = T * P24 + P44;
                    double M00 = P00, M01 = P01, M11 = P11, M02 = w20, M12 = w21, M03 = w30, M13 = w31, M04 = w40, M14 = w41;
                    double M22 = a * w20 + bq * w21 + w22, M23 = a * w30 + bq * w31 + w32, M24 = a * w40 + bq * w41 + w42;
                    double M33 = Tc * w30 + Ts * w31 + w33, M34 = Tc * w40 + Ts * w41 + w43, M44 = T * w42 + w44;
                    double mv0 = pb0, mv1 = pb1, mv2 = a * pb0 + bq * pb1 + pb2, mv3 = Tc * pb0 + Ts * pb1 + pb3, mv4 = T * pb2 + pb4;
                    if (k >= 1) { M00 += H0; M01 += H1; M11 += H2; M22 += H3; mv0 += g0; mv1 += g1; mv2 += g2; }
                    M23 += hv;
                    if (k <= Nc - 1) {        // the stage where control j is decided carries its whole diagonal / gradient
                        M33 += huu0 + delta; M44 += huu1 + delta;
                        mv3 += gu0; mv4 += gu1;
                        const double dv = M33;
                        if (!(uni(dv) > 0.0)) return false;
                        const double rdv = 1.0 / dv;          // two reciprocals per stage (the recursion is uniform: every lane pays them)
                        const double l43 = M34 * rdv, d1o = M44, d1 = d1o - l43 * M34;
                        if (!(uni(d1) > 1e-9 * fabs(uni(d1o))) || !(uni(d1) > 0.0)) return false;
                        const double rd1 = 1.0 / d1;
                        // gains: [K | kff] = -[[M33 M34], [M34 M44]]^-1 [M3q M4q | mv3 mv4], q = x, y, theta
                        const double r3[4] = {M03, M13, M23, mv3}, r4[4] = {M04, M14, M24, mv4};
                        double K0[4], K1[4];
#pragma unroll
                        for (int q_ = 0; q_ < 4; q_++) {
                            const double y4 = (r4[q_] - l43 * r3[q_]) * rd1, y3 = (r3[q_] - M34 * y4) * rdv;
                            K0[q_] = -y3; K1[q_] = -y4;
                        }
                        if (lane == 0) {
                            double *sk = SC + j * 16 + 6;
                            sk[0] = K0[0]; sk[1] = K0[1]; sk[2] = K0[2]; sk[3] = K1[0]; sk[4] = K1[1]; sk[5] = K1[2]; sk[6] = K0[3]; sk[7] = K1[3];
                        }
                        P00 = M00 + M03 * K0[0] + M04 * K1[0]; P01 = M01 + M03 * K0[1] + M04 * K1[1]; P02 = M02 + M03 * K0[2] + M04 * K1[2];
                        P11 = M11 + M13 * K0[1] + M14 * K1[1]; P12 = M12 + M13 * K0[2] + M14 * K1[2]; P22 = M22 + M23 * K0[2] + M24 * K1[2];
                        p0 = mv0 + M03 * K0[3] + M04 * K1[3]; p1 = mv1 + M13 * K0[3] + M14 * K1[3]; p2 = mv2 + M23 * K0[3] + M24 * K1[3];
                        P03 = 0.0; P04 = 0.0; P13 = 0.0; P14 = 0.0; P23 = 0.0; P24 = 0.0; P33 = 0.0; P34 = 0.0; P44 = 0.0; p3 = 0.0; p4 = 0.0;
                    } else {                  // held control: it stays a parameter of the cost-to-go
                        P00 = M00; P01 = M01; P02 = M02; P03 = M03; P04 = M04; P11 = M11; P12 = M12; P13 = M13; P14 = M14; P22 = M22; P23 = M23; P24 = M24;
                        P33 = M33; P34 = M34; P44 = M44;
                        p0 = mv0; p1 = mv1; p2 = mv2; p3 = mv3; p4 = mv4;
                    }
                    return true;
                };
                for (int k0 = N - 1; k0 >= 0; k0 -= 2) {
                    double oa[13], na[3], ca[6], ob[13], nb[3], cb[6];
                    fetch(k0, oa, na, ca); fetch(k0 >= 1 ? k0 - 1 : 0, ob, nb, cb);
                    if (!stage(k0, oa, na, ca)) { ok = false; break; }
                    if (k0 >= 1 && !stage(k0 - 1, ob, nb, cb)) { ok = false; break; }
                }
#endif
                if (ok) break;
                ntry++;
                if (delta == 0.0) delta = (delta_last == 0.0) ? 1e-4 : fmax(1e-20, delta_last / 3.0);
                else delta *= (delta_last == 0.0) ? 100.0 : NMPC_SHIFT_ESCALATION;
                delta = uni(delta);
                if (delta > 1e20) break;
            }
            if (!ok) { if (n_cold < NMPC_COLD_RETRIES) { cold_retry(); it++; break; } status = NMPC_STATUS_NUMERIC; break; }
            if (delta > 0.0) delta_last = delta;
            need_shift = delta > 0.0 && (ntry > 0 || delta > 1e-6);
            __syncthreads();

            LP(3);
            // ---- C. forward sweep.  The closed-loop maps of the stages are formed in parallel (one stage per lane): with the gains of the sweep,
            //      dx+ = (A + B K) dx + (B kff - c) below the control horizon and dx+ = A dx + (B du_held - c) above it, so that the recursion
            //      over the horizon — uniform on all lanes, operands and results in LDS — is three dependent multiply-adds per stage
            for (int k = lane; k < N; k += 64) {
                const double *sb = SB + k * 13, *sc = SC + cof(k) * 16;
                double *sf = SF + k * 12;
                const double u0 = sc[0], u1 = sc[1], sn = sb[8], cs = sb[9];
                const double a = -T * u0 * sn, bq = T * u0 * cs, Tc = T * cs, Ts = T * sn;
                const double c0 = sb[13 + 10] - (sb[10] + T * u0 * cs), c1 = sb[13 + 11] - (sb[11] + T * u0 * sn), c2 = sb[13 + 12] - (sb[12] + T * u1);
                if (k <= Nc - 1) {
                    const double K00 = sc[6], K01 = sc[7], K02 = sc[8], K10 = sc[9], K11 = sc[10], K12 = sc[11], kf0 = sc[12], kf1 = sc[13];
                    sf[0] = 1.0 + Tc * K00; sf[1] = Tc * K01; sf[2] = a + Tc * K02;
                    sf[3] = Ts * K00; sf[4] = 1.0 + Ts * K01; sf[5] = bq + Ts * K02;
                    sf[6] = T * K10; sf[7] = T * K11; sf[8] = 1.0 + T * K12;
                    sf[9] = Tc * kf0 - c0; sf[10] = Ts * kf0 - c1; sf[11] = T * kf1 - c2;
                } else { sf[0] = a; sf[1] = bq; sf[2] = Tc; sf[3] = Ts; sf[4] = c0; sf[5] = c1; sf[6] = c2; }
            }
            __syncthreads();
            {
                // stages in batches: the LDS operands of a batch are read FIRST (the compiler does not move LDS reads across the lane-0 stores of
                // the previous stages — one exposed LDS round trip per stage otherwise), then the stages are walked, then the results stored
                constexpr int UB = NMPC_LIDAR_UNROLL;
                double dx0 = 0.0, dx1 = 0.0, dx2 = 0.0;
                if (lane == 0) { SD[0] = 0.0; SD[1] = 0.0; SD[2] = 0.0; }
                for (int k0 = 0; k0 < Nc; k0 += UB) {
                    double F_[UB][12], n_[UB][3];
#pragma unroll
                    for (int i = 0; i < UB; i++) {
                        const double *sf = SF + (k0 + i < Nc ? k0 + i : Nc - 1) * 12;
#pragma unroll
                        for (int q_ = 0; q_ < 12; q_++) F_[i][q_] = sf[q_];
                    }
#pragma unroll
                    for (int i = 0; i < UB; i++)
                        if (k0 + i < Nc) {
                            const double n0 = F_[i][9] + F_[i][0] * dx0 + F_[i][1] * dx1 + F_[i][2] * dx2, n1 = F_[i][10] + F_[i][3] * dx0 + F_[i][4] * dx1 + F_[i][5] * dx2,
                                         n2 = F_[i][11] + F_[i][6] * dx0 + F_[i][7] * dx1 + F_[i][8] * dx2;
                            dx0 = n0; dx1 = n1; dx2 = n2;
                            n_[i][0] = n0; n_[i][1] = n1; n_[i][2] = n2;
                        }
                    if (lane == 0) {
#pragma unroll
                        for (int i = 0; i < UB; i++)
                            if (k0 + i < Nc) { SD[3 * (k0 + i + 1)] = n_[i][0]; SD[3 * (k0 + i + 1) + 1] = n_[i][1]; SD[3 * (k0 + i + 1) + 2] = n_[i][2]; }
                    }
                }
                if (Nc < N) {      // the held control: the one of stage Nc - 1, from the pose step that stage started from
                    const double *sc = SC + (Nc - 1) * 16, *sd = SD + 3 * (Nc - 1);
                    const double du0 = sc[12] + sc[6] * sd[0] + sc[7] * sd[1] + sc[8] * sd[2], du1 = sc[13] + sc[9] * sd[0] + sc[10] * sd[1] + sc[11] * sd[2];
                    for (int k0 = Nc; k0 < N; k0 += UB) {
                        double F_[UB][7], n_[UB][3];
#pragma unroll
                        for (int i = 0; i < UB; i++) {
                            const double *sf = SF + (k0 + i < N ? k0 + i : N - 1) * 12;
#pragma unroll
                            for (int q_ = 0; q_ < 7; q_++) F_[i][q_] = sf[q_];
                        }
#pragma unroll
                        for (int i = 0; i < UB; i++)
                            if (k0 + i < N) {
                                const double n0 = dx0 + F_[i][0] * dx2 + (F_[i][2] * du0 - F_[i][4]), n1 = dx1 + F_[i][1] * dx2 + (F_[i][3] * du0 - F_[i][5]),
                                             n2 = dx2 + (T * du1 - F_[i][6]);
                                dx0 = n0; dx1 = n1; dx2 = n2;
                                n_[i][0] = n0; n_[i][1] = n1; n_[i][2] = n2;
                            }
                        if (lane == 0) {
#pragma unroll
                            for (int i = 0; i < UB; i++)
                                if (k0 + i < N) { SD[3 * (k0 + i + 1)] = n_[i][0]; SD[3 * (k0 + i + 1) + 1] = n_[i][1]; SD[3 * (k0 + i + 1) + 2] = n_[i][2]; }
                        }
                    }
                }
            }
            __syncthreads();
            // control steps du_j = kff_j + K_j dx_j (one control per lane): to the workspace and, for the adjoint recursion, to LDS
            for (int j = lane; j < Nc; j += 64) {
                double *sc = SC + j * 16;
                const double *sd = SD + 3 * j;
                const double du0 = sc[12] + sc[6] * sd[0] + sc[7] * sd[1] + sc[8] * sd[2], du1 = sc[13] + sc[9] * sd[0] + sc[10] * sd[1] + sc[11] * sd[2];
                sc[14] = du0; sc[15] = du1;
                wsb[odU + 2 * j] = du0; wsb[odU + 2 * j + 1] = du1;
            }
            __syncthreads();
            LP(4);
            double mult_max = 0.0;
            double dphi_f = 0.0;      // directional derivative of the objective (states; the controls' part is added in D)
            for (int k = 1 + lane; k <= N; k += 64) {      // dd = G dx + (g - d);  eta+ = -(gd + Wd dd)
                const double d0 = SD[3 * k], d1 = SD[3 * k + 1], d2 = SD[3 * k + 2], x = SB[k * 13 + 10], y = SB[k * 13 + 11];
                SV(odV, k, 0) = d0; SV(odV, k, 1) = d1; SV(odV, k, 2) = d2;
                if (k < N) { dphi_f += 2 * q0 * (x - xs0) * d0; dphi_f += 2 * q1 * (y - xs1) * d1; dphi_f += 2 * q2 * (SB[k * 13 + 12] - xs2) * d2; }
                {   // the adjoint recursion lambda_k = r_k + A_k' lambda_k+1: its stage terms r_k = -(g + H dx) - (0, 0, hvt du0) and the two entries of A_k
                    const double *sb = SB + k * 13;
                    double *sf = SF + (k - 1) * 12;
                    const double H0 = sb[0], H1 = sb[1], H2 = sb[2], H3 = sb[3];
                    double r2 = -(sb[6] + H3 * d2), aa = 0.0, bb = 0.0;
                    if (k < N) { const double *sc = SC + cof(k) * 16; const double u0 = sc[0]; r2 -= sb[7] * sc[14]; aa = -T * u0 * sb[8]; bb = T * u0 * sb[9]; }
                    sf[0] = -(sb[4] + H0 * d0 + H1 * d1); sf[1] = -(sb[5] + H1 * d0 + H2 * d1); sf[2] = r2; sf[3] = aa; sf[4] = bb;
                }
#pragma unroll
                for (int m0 = 0; m0 < RM; m0 += CH)
                    if (R_ >= 0 || m0 < R) {
                        double d[CH], gv[CH], wd[CH];
#pragma unroll
                        for (int u = 0; u < CH; u++)
                            if (m0 + u < RM) { const int m = M_IX(m0 + u, m0); d[u] = SV(oV, k, 3 + m); gv[u] = RV(ogdv, k, m); wd[u] = RV(oWd, k, m); }
#pragma unroll
                        for (int u = 0; u < CH; u++)
                            if (m0 + u < RM && M_ON(m0 + u)) {
                                const int m = m0 + u;
                                double sx, sy;
                                const double gg = gdist(m, x, y, sx, sy), dd = sx * d0 + sy * d1 + (gg - d[u]);
                                SV(odV, k, 3 + m) = dd;
                                if (k < N && lw != 0.0) dphi_f += qdiv(-2.0 * lw, d[u] * d[u] * d[u]) * dd;
                                const double v = -(gv[u] + wd[u] * dd);
                                RV(oetan, k, m) = v; mult_max = fmax(mult_max, fabs(v));
                            }
                    }
            }
            __syncthreads();
            LP(5);
            // ---- multipliers of the QP: lambda+ by the adjoint recursion (uniform; it replaces the pose step in SD stage by stage)
            {
                constexpr int UB = NMPC_LIDAR_UNROLL;      // batches as in the forward recursion: operands first, then the stages, then the stores
                double ln0 = 0.0, ln1 = 0.0, ln2 = 0.0;
                for (int k0 = N; k0 >= 1; k0 -= UB) {
                    double F_[UB][5], l_[UB][3];
#pragma unroll
                    for (int i = 0; i < UB; i++) {
                        const double *sf = SF + ((k0 - i >= 1 ? k0 - i : 1) - 1) * 12;
#pragma unroll
                        for (int q_ = 0; q_ < 5; q_++) F_[i][q_] = sf[q_];
                    }
#pragma unroll
                    for (int i = 0; i < UB; i++)
                        if (k0 - i >= 1) {       // (a, bq are zero at k = N and lambda_N+1 = 0)
                            const double l0 = F_[i][0] + ln0, l1 = F_[i][1] + ln1, l2 = F_[i][2] + ln2 + F_[i][3] * ln0 + F_[i][4] * ln1;
                            ln0 = l0; ln1 = l1; ln2 = l2;
                            l_[i][0] = l0; l_[i][1] = l1; l_[i][2] = l2;
                            mult_max = fmax(mult_max, fmax(fabs(l0), fmax(fabs(l1), fabs(l2))));
                        }
                    if (lane == 0) {
#pragma unroll
                        for (int i = 0; i < UB; i++)
                            if (k0 - i >= 1) { SD[3 * (k0 - i)] = l_[i][0]; SD[3 * (k0 - i) + 1] = l_[i][1]; SD[3 * (k0 - i) + 2] = l_[i][2]; }
                    }
                }
                mult_max = wmax_(mult_max);
            }
            __syncthreads();
            LP(6);
            // ---- D. fraction to the boundary; directional derivative of the barrier function.  One variable per lane (the arrays of a
            //      stage-like quantity share the flat index e = c (N+1) + k), two trips' loads issued together
            double a_p = 1.0, a_d = 1.0, dphi = dphi_f, thh = 0.0;
            LogSum lgs_;
            auto slot = [&](double s_, double z_, double h_, double jd_) {
                const double ds_ = jd_ + (h_ - s_), dz_ = qdiv(mu - s_ * z_ - z_ * ds_, s_);
                if (ds_ < 0.0) a_p = fmin(a_p, qdiv(-tau * s_, ds_));
                if (dz_ < 0.0) a_d = fmin(a_d, qdiv(-tau * z_, dz_));
                dphi -= qdiv(mu * ds_, s_); lgs_.add(s_); thh += fabs(h_ - s_);
            };
            {
                auto item = [&](int e, double v, double dv, double lo, double hi, double sl, double zl, double su, double zu) {
                    if (isfinite(lo)) slot(sl, zl, v - lo, dv);
                    if (isfinite(hi)) slot(su, zu, hi - v, -dv);
                };
                for (int e0 = lane; e0 < nV; e0 += 128) {
                    const int e1 = e0 + 64, i1 = e1 < nV ? e1 : e0;
                    const double v0 = wsb[oV + e0], dv0 = wsb[odV + e0], lo0 = lbS[e0], hi0 = ubS[e0], sl0 = wsb[oSL + e0], zl0 = wsb[oZL + e0], su0 = wsb[oSU + e0],
                                 zu0 = wsb[oZU + e0];
                    const double v1 = wsb[oV + i1], dv1 = wsb[odV + i1], lo1 = lbS[i1], hi1 = ubS[i1], sl1 = wsb[oSL + i1], zl1 = wsb[oZL + i1], su1 = wsb[oSU + i1],
                                 zu1 = wsb[oZU + i1];
                    item(e0, v0, dv0, lo0, hi0, sl0, zl0, su0, zu0);
                    if (e1 < nV) item(e1, v1, dv1, lo1, hi1, sl1, zl1, su1, zu1);
                }
            }
            for (int o = lane; o < 2 * Nc; o += 64) {
                const double u = wsb[oU + o], du = wsb[odU + o];
                slot(wsb[oSLu + o], wsb[oZLu + o], u - lbu[o], du);
                slot(wsb[oSUu + o], wsb[oZUu + o], ubu[o] - u, -du);
                const int j = o >> 1, cnt = (j < Nc - 1) ? 1 : N - Nc + 1;
                dphi += cnt * 2 * P.r[o & 1] * u * du;
            }
            a_p = wmin_(a_p); a_d = wmin_(a_d); dphi = wsum_(dphi); thh = wsum_(thh);
            const double lgs = wlogsum_(lgs_);
            LP(7);
            // ---- E. l1 merit backtracking (non-monotone reference: max of the last merit values of this barrier problem)
            const double theta0 = uni(th0 + thh), phi0 = uni(f - mu * lgs);
            if (theta0 > 0.0) {
                const double nut = fmin(dphi / ((1.0 - 0.1) * theta0), mult_max / (1.0 - 0.1));
                nu_pen = fmax(1.0, 0.5 * nu_pen);
                if (nu_pen < nut) nu_pen = nut + 1.0;
                nu_pen = uni(nu_pen);
            }
            const double Dm = uni(dphi - nu_pen * theta0);
            double alpha = a_p, ft = f, tht = th0, ect = e_c;
            if (mh_mu != mu || mh_nu != nu_pen) { mcount = 0; mh_mu = mu; mh_nu = nu_pen; }
            const double m0 = uni(phi0 + nu_pen * theta0);
            double mref = m0;
            if (mcount > 0) mref = fmax(mref, mh0);
            if (mcount > 1) mref = fmax(mref, mh1);
            if (mcount > 2) mref = fmax(mref, mh2);
            mref = uni(mref);
            mh2 = mh1; mh1 = mh0; mh0 = m0; if (mcount < 3) mcount++;
            const bool wd_fire = n_short >= NMPC_WATCHDOG_TRIGGER && a_p >= NMPC_WATCHDOG_MIN_AP;
            bool wd_took = false;
            for (int ls = 0; ls < 30; ls++) {
                double tb = 0.0;
                LogSum lgt_;
                auto trial = [&](double s_, double h0_, double jd_, double ht_) { const double st_ = s_ + alpha * (jd_ + (h0_ - s_)); lgt_.add(st_); tb += fabs(ht_ - st_); };
                {
                    auto item = [&](int e, double v, double dv, double lo, double hi, double sl, double su) {
                        const double vt = v + alpha * dv;
                        wsb[oVt + e] = vt;
                        if (isfinite(lo)) trial(sl, v - lo, dv, vt - lo);
                        if (isfinite(hi)) trial(su, hi - v, -dv, hi - vt);
                    };
                    for (int e0 = lane; e0 < nV; e0 += 128) {
                        const int e1 = e0 + 64, i1 = e1 < nV ? e1 : e0;
                        const double v0 = wsb[oV + e0], dv0 = wsb[odV + e0], lo0 = lbS[e0], hi0 = ubS[e0], sl0 = wsb[oSL + e0], su0 = wsb[oSU + e0];
                        const double v1 = wsb[oV + i1], dv1 = wsb[odV + i1], lo1 = lbS[i1], hi1 = ubS[i1], sl1 = wsb[oSL + i1], su1 = wsb[oSU + i1];
                        item(e0, v0, dv0, lo0, hi0, sl0, su0);
                        if (e1 < nV) item(e1, v1, dv1, lo1, hi1, sl1, su1);
                    }
                }
                for (int o = lane; o < 2 * Nc; o += 64) {
                    const double u = wsb[oU + o], du = wsb[odU + o], ut = u + alpha * du;
                    wsb[oUt + o] = ut;
                    trial(wsb[oSLu + o], u - lbu[o], du, ut - lbu[o]);
                    trial(wsb[oSUu + o], ubu[o] - u, -du, ubu[o] - ut);
                }
                __syncthreads();
                ft = eval_point(oVt, oUt, tht, ect);
                const double lgt = wlogsum_(lgt_);
                tb = wsum_(tb);
                const double mt = (ft - mu * lgt) + nu_pen * (tht + tb);
                if (wd_fire && isfinite(mt)) { mcount = 0; wd_took = true; break; }      // watchdog: the fraction-to-the-boundary step without the merit test
                if (mt <= mref + 1e-4 * alpha * Dm + 1e-13 * fabs(phi0)) break;
                if (ls < 29) alpha = uni(alpha * 0.5);
                __syncthreads();
            }
            a_d = uni(fmin(a_d, alpha));
            LP(8);
            n_tiny = (alpha < 1e-10) ? n_tiny + 1 : 0;
            n_short = (!wd_took && alpha < a_p) ? n_short + 1 : 0;
            // ---- G. accept: duals and slacks (they need the old primal point), then the primal point
            auto upd = [&](double s_, double z_, double h_, double jd_, double &sn_, double &zo_) {
                const double ds_ = jd_ + (h_ - s_), dz_ = qdiv(mu - s_ * z_ - z_ * ds_, s_);
                sn_ = s_ + alpha * ds_;
                const double zn_ = z_ + a_d * dz_;
                zo_ = fmin(fmax(zn_, qdiv(1e-10 * mu, sn_)), qdiv(1e10 * mu, sn_));
            };
            {
                auto item = [&](int e, double v, double dv, double lo, double hi, double sl, double zl, double su, double zu) {
                    double s2, z2;
                    if (isfinite(lo)) { upd(sl, zl, v - lo, dv, s2, z2); wsb[oSL + e] = s2; wsb[oZL + e] = z2; }
                    if (isfinite(hi)) { upd(su, zu, hi - v, -dv, s2, z2); wsb[oSU + e] = s2; wsb[oZU + e] = z2; }
                };
                for (int e0 = lane; e0 < nV; e0 += 128) {
                    const int e1 = e0 + 64, i1 = e1 < nV ? e1 : e0;
                    const double v0 = wsb[oV + e0], dv0 = wsb[odV + e0], lo0 = lbS[e0], hi0 = ubS[e0], sl0 = wsb[oSL + e0], zl0 = wsb[oZL + e0], su0 = wsb[oSU + e0],
                                 zu0 = wsb[oZU + e0];
                    const double v1 = wsb[oV + i1], dv1 = wsb[odV + i1], lo1 = lbS[i1], hi1 = ubS[i1], sl1 = wsb[oSL + i1], zl1 = wsb[oZL + i1], su1 = wsb[oSU + i1],
                                 zu1 = wsb[oZU + i1];
                    item(e0, v0, dv0, lo0, hi0, sl0, zl0, su0, zu0);
                    if (e1 < nV) item(e1, v1, dv1, lo1, hi1, sl1, zl1, su1, zu1);
                }
            }
            for (int k = 1 + lane; k <= N; k += 64) {
                // multipliers of the equality rows of stage k: lambda (lambda+ is in SD), eta
                LV(k, 0) += alpha * (SD[3 * k] - LV(k, 0)); LV(k, 1) += alpha * (SD[3 * k + 1] - LV(k, 1)); LV(k, 2) += alpha * (SD[3 * k + 2] - LV(k, 2));
#pragma unroll
                for (int m0 = 0; m0 < RM; m0 += CH)
                    if (R_ >= 0 || m0 < R) {
                        double e0[CH], e1[CH];
#pragma unroll
                        for (int u = 0; u < CH; u++)
                            if (m0 + u < RM) { const int m = M_IX(m0 + u, m0); e0[u] = RV(oeta, k, m); e1[u] = RV(oetan, k, m); }
#pragma unroll
                        for (int u = 0; u < CH; u++)
                            if (m0 + u < RM && M_ON(m0 + u)) RV(oeta, k, m0 + u) = e0[u] + alpha * (e1[u] - e0[u]);
                    }
            }
            for (int o = lane; o < 2 * Nc; o += 64) {
                const double u = wsb[oU + o], du = wsb[odU + o];
                double s2, z2;
                upd(wsb[oSLu + o], wsb[oZLu + o], u - lbu[o], du, s2, z2); wsb[oSLu + o] = s2; wsb[oZLu + o] = z2;
                upd(wsb[oSUu + o], wsb[oZUu + o], ubu[o] - u, -du, s2, z2); wsb[oSUu + o] = s2; wsb[oZUu + o] = z2;
            }
            { int t_ = oV; oV = oVt; oVt = t_; t_ = oU; oU = oUt; oUt = t_; }      // the trial point of the accepted step length is in (oVt, oUt)
            f = ft; th0 = tht; e_c = ect;      // ... and its evaluation (objective, residual norms, sin / cos in SB) is the line search's last one
            __syncthreads();
            LP(9);
            it++;
            if (n_tiny >= 5) {
                if (n_restart >= 3) { if (n_cold < NMPC_COLD_RETRIES) { cold_retry(); break; } status = NMPC_STATUS_STALLED; break; }
                n_restart++; n_tiny = 0; mu = uni(fmax(mu, P.mu_init)); restarting = true;
                break;
            }
        }
        if (!restarting) break;
    }
    __syncthreads();
    for (int k = lane; k <= N; k += 64) for (int c = 0; c < ns; c++) wo[(size_t)k * ns + c] = SV(oV, k, c);
    for (int e = lane; e < 2 * Nc; e += 64) wo[(size_t)NP1 * ns + e] = wsb[oU + e];
    if (lane == 0) {
        if (obj_out) obj_out[b] = f;
        if (status_out) status_out[b] = status;
        if (iters_out) iters_out[b] = it;
        if (kkt_out) kkt_out[b] = kkt;
    }
#ifdef NMPC_LIDAR_PROFILE
    __syncthreads();
    if (lane == 0) for (int i = 0; i < 12; i++) wo[i] = (double)prof[i];
#endif
}
#undef SV
#undef RV
#undef LV
#undef C_ON
#undef C_IX
#undef M_ON
#undef M_IX

// f (V4:135-136) and g = [gx; gd] (V4:151): one wavefront per instance, lane = stage (+64, ...); trip k == N handles the initial rows.
// The objective is summed in a fixed order (per lane over its stages, then a butterfly over the wave): bit-reproducible from run to run.
__global__ __launch_bounds__(64) void lidar_eval_kernel(const LParams P, int B, const double *__restrict__ p_in, const double *__restrict__ w,
                                                         double *__restrict__ f_out, double *__restrict__ g_out)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= B) return;
    const int N = P.N, R = P.R, ns = P.ns;
    const double *X = w + (size_t)b * P.nvar, *U = X + (size_t)(N + 1) * ns, *pp = p_in + (size_t)b * P.np;
    double *gx = g_out ? g_out + (size_t)b * P.ng : nullptr, *gd = g_out ? gx + 3 * (N + 1) : nullptr;
    double fs = 0.0;
    for (int k = lane; k <= N; k += 64) {
        if (k == N) {
            if (g_out) { for (int i = 0; i < 3; i++) gx[i] = X[i] - pp[i]; for (int m = 0; m < R; m++) gd[m] = X[3 + m] - pp[6 + m]; }
            continue;
        }
        const double *v = X + (size_t)k * ns, *vn = v + ns, *u = U + 2 * (k < P.Nc - 1 ? k : P.Nc - 1);
        double fk = P.q[0] * (v[0] - pp[3]) * (v[0] - pp[3]) + P.q[1] * (v[1] - pp[4]) * (v[1] - pp[4]) + P.q[2] * (v[2] - pp[5]) * (v[2] - pp[5]) +
                    P.r[0] * u[0] * u[0] + P.r[1] * u[1] * u[1];
        if (P.lw != 0.0) for (int m = 0; m < R; m++) fk += P.lw / (v[3 + m] * v[3 + m]);
        fs += fk;
        if (g_out) {
            double s, c;
            sincos(v[2], &s, &c);
            gx[3 * (k + 1)] = vn[0] - (v[0] + P.T * u[0] * c); gx[3 * (k + 1) + 1] = vn[1] - (v[1] + P.T * u[0] * s); gx[3 * (k + 1) + 2] = vn[2] - (v[2] + P.T * u[1]);
            for (int m = 0; m < R; m++) {
                double sa, ca;
                sincos(X[2] + pp[6 + R + m], &sa, &ca);       // pObs from the STAGE-0 variables of w (V4:114-118)
                gd[R * (k + 1) + m] = vn[3 + m] - (fabs(vn[0] - (X[0] + X[3 + m] * ca)) + fabs(vn[1] - (X[1] + X[3 + m] * sa)));
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) fs += __shfl_xor(fs, o);
    if (f_out && lane == 0) f_out[b] = fs;
}

// warm-start shuffle V4:258-270
__global__ __launch_bounds__(256) void lidar_shift_kernel(const LParams P, int B, const double *__restrict__ w_in, double *__restrict__ w_next)
{
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)B * P.nvar) return;
    const int b = (int)(gid / P.nvar), e = (int)(gid - (long)b * P.nvar), N = P.N, ns = P.ns, nX = (N + 1) * ns;
    const double *src = w_in + (size_t)b * P.nvar;
    double v;
    if (e < nX) { int k = e / ns, c = e - k * ns; v = (k < N) ? src[(k + 1) * ns + c] : src[(N - 1) * ns + c]; }
    else { int eu = e - nX, j = eu >> 1, c = eu & 1; v = (j < P.Nc - 1) ? src[nX + 2 * (j + 1) + c] : src[nX + 2 * (P.Nc - 1) + c]; }
    w_next[gid] = v;
}

// synthetic LaserScan (callback_lidar, V4:29-36): one thread per (instance, ray); range to the nearest circular obstacle, clipped
__global__ __launch_bounds__(256) void lidar_scan_kernel(long B, int R, int K, const double *__restrict__ pose, const double *__restrict__ world, double scan_max,
                                                          double *__restrict__ scan)
{
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= B * R) return;
    const long b = gid / R;
    const int m = (int)(gid - b * R);
    const double x = pose[3 * b], y = pose[3 * b + 1], th = pose[3 * b + 2];
    double s, c;
    sincos(th + (double)m * (2.0 * 3.14159265358979323846) / (double)R, &s, &c);
    double best = scan_max;
    for (int o = 0; o < K; o++) {
        const double *ob = world + ((size_t)b * K + o) * 3;
        const double fx = ob[0] - x, fy = ob[1] - y, t = fx * c + fy * s, h2 = fx * fx + fy * fy - t * t, r2 = ob[2] * ob[2];
        // the near intersection of the ray with the circle; a pose inside (or touching) an obstacle reads range 0 on every ray (the
        // near root is negative there, and a negative range would enter the NLP as a distance state)
        if (fx * fx + fy * fy <= r2) best = 0.0;
        else if (t > 0.0 && h2 < r2) best = fmin(best, fmax(t - sqrt(r2 - h2), 0.0));
    }
    scan[gid] = best;
}

// plant step pose + T f(pose, u_0) (V4:78-87 as the robot; casadi_test.py:17-26)
__global__ __launch_bounds__(256) void lidar_plant_kernel(const LParams P, int B, const double *p_in, const double *__restrict__ w_sol, double *pose_next, int stride)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double *x0 = p_in + (size_t)b * P.np, *u = w_sol + (size_t)b * P.nvar + (size_t)(P.N + 1) * P.ns;
    double s, c;
    sincos(x0[2], &s, &c);
    const double n0 = x0[0] + P.T * u[0] * c, n1 = x0[1] + P.T * u[0] * s, n2 = x0[2] + P.T * u[1];      // read before written: pose_next may be p itself
    double *o = pose_next + (size_t)b * stride;
    o[0] = n0; o[1] = n1; o[2] = n2;
}

}  // namespace nmpc_lidar

struct nmpc_lidar_handle {
    nmpc_lidar_config_t cfg;
    nmpc_lidar::LParams P;
    int32_t max_batch;
    size_t lds_bytes;           // dynamic LDS of the solve kernel: per-stage operands of the three recursions
    int device;
    int n_cu;                   // compute units of the device (256 on MI355X): batches beyond one instance per SIMD run the two-waves-per-SIMD instantiation
    double *ws, *lb, *ub;
    int64_t ws_bytes;
};

extern "C" {

int32_t nmpc_lidar_n_var(const nmpc_lidar_config_t *c) { return c ? (3 + c->R) * (c->N + 1) + 2 * c->Nc : NMPC_E_ARG; }
int32_t nmpc_lidar_n_g(const nmpc_lidar_config_t *c) { return c ? (3 + c->R) * (c->N + 1) : NMPC_E_ARG; }
int32_t nmpc_lidar_n_p(const nmpc_lidar_config_t *c) { return c ? 6 + 2 * c->R : NMPC_E_ARG; }

int32_t nmpc_lidar_create(const nmpc_lidar_config_t *cfg, const double *lbx, const double *ubx, int32_t max_batch, nmpc_lidar_handle_t **out)
{
    if (!cfg || !lbx || !ubx || !out || max_batch < 1) return NMPC_E_ARG;
    if (cfg->N < 1 || cfg->N > 4096 || cfg->Nc < 1 || cfg->Nc > cfg->N || cfg->R < 0 || cfg->R > NMPC_LIDAR_MAX_RAYS) return NMPC_E_ARG;
    if (!(cfg->T > 0.0) || !(cfg->tol > 0.0) || !(cfg->mu_init > 0.0) || cfg->max_iter < 0 || !(cfg->lw >= 0.0)) return NMPC_E_ARG;
    if (!(cfg->q[0] >= 0.0) || !(cfg->q[1] >= 0.0) || !(cfg->q[2] >= 0.0) || !(cfg->r[0] > 0.0) || !(cfg->r[1] > 0.0)) return NMPC_E_ARG;
    const int nv = nmpc_lidar_n_var(cfg), ns = 3 + cfg->R, N = cfg->N, Nc = cfg->Nc, R = cfg->R;
    for (int i = 0; i < nv; i++) if (!(lbx[i] < ubx[i])) return NMPC_E_ARG;                                   // NaN or empty interval
    for (int i = (N + 1) * ns; i < nv; i++) if (!isfinite(lbx[i]) || !isfinite(ubx[i])) return NMPC_E_ARG;   // controls are boxed (V4:166,172)
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return NMPC_E_HIP;   // fail loudly: no CPU path exists
    nmpc_lidar_handle *h = (nmpc_lidar_handle *)calloc(1, sizeof(nmpc_lidar_handle));
    if (!h) return NMPC_E_NOMEM;
    if (hipGetDevice(&h->device) != hipSuccess) { free(h); return NMPC_E_HIP; }
    { hipDeviceProp_t prop; h->n_cu = (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256; }
    h->cfg = *cfg; h->max_batch = max_batch;
    h->lds_bytes = sizeof(double) * ((size_t)(N + 1) * 13 + (size_t)Nc * 16 + (size_t)(N + 1) * 3 + 2 * NMPC_LIDAR_MAX_RAYS + (size_t)N * 12);
    if (h->lds_bytes > 160 * 1024) { free(h); return NMPC_E_ARG; }       // horizon beyond the LDS of a CU: 8 (16 (N + 1) + 16 Nc + 12 N + 32) bytes <= 160 KB, i.e. N <= ~560 with Nc = N / 2
    nmpc_lidar::LParams &P = h->P;
    memset(&P, 0, sizeof(P));
    P.N = N; P.Nc = Nc; P.R = R; P.ns = ns; P.max_iter = cfg->max_iter; P.nvar = nv; P.ng = nmpc_lidar_n_g(cfg); P.np = nmpc_lidar_n_p(cfg);
    P.T = cfg->T; P.lw = cfg->lw; P.tol = cfg->tol; P.mu_init = cfg->mu_init;
    for (int i = 0; i < 3; i++) P.q[i] = cfg->q[i];
    for (int i = 0; i < 2; i++) P.r[i] = cfg->r[i];
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t at = o; o += n; return at; };
    const int64_t nV = (int64_t)(N + 1) * ns, nU = 2 * Nc, nL = (int64_t)N * 3, nE = (int64_t)N * R;
    P.oeta = take(nE); P.oetan = take(nE); P.oWd = take(nE); P.ogdv = take(nE); P.olam = take(nL);
    P.oU = take(nU); P.oUt = take(nU); P.odU = take(nU); P.oSLu = take(nU); P.oZLu = take(nU); P.oSUu = take(nU); P.oZUu = take(nU);
    P.oV = take(nV); P.oVt = take(nV); P.odV = take(nV); P.oSL = take(nV); P.oZL = take(nV); P.oSU = take(nV); P.oZU = take(nV);
    P.total = o;
    P.total = (P.total + 15) / 16 * 16;      // every instance starts on a 128-byte boundary
    P.S = max_batch;
    h->ws_bytes = (int64_t)sizeof(double) * P.total * P.S;
    if (hipMalloc((void **)&h->ws, (size_t)h->ws_bytes) != hipSuccess) { free(h); return NMPC_E_NOMEM; }
    // bounds in the solve kernel's layout (LParams): state part component-major with stage 0 opened, controls, then the caller's stage-0 bounds
    const size_t nb = (size_t)nv + ns;
    double *hl = (double *)malloc(sizeof(double) * nb), *hu = (double *)malloc(sizeof(double) * nb);
    if (!hl || !hu) { free(hl); free(hu); (void)hipFree(h->ws); free(h); return NMPC_E_NOMEM; }
    int n_ineq = 4 * Nc;
    for (int k = 0; k <= N; k++)
        for (int c = 0; c < ns; c++) {
            const double lo = lbx[(size_t)k * ns + c], hi = ubx[(size_t)k * ns + c];
            hl[(size_t)c * (N + 1) + k] = k ? lo : -INFINITY; hu[(size_t)c * (N + 1) + k] = k ? hi : INFINITY;
            if (k) n_ineq += (isfinite(lo) ? 1 : 0) + (isfinite(hi) ? 1 : 0);
            else { hl[(size_t)nv + c] = lo; hu[(size_t)nv + c] = hi; }
        }
    for (int e = 0; e < 2 * Nc; e++) { hl[nV + e] = lbx[nV + e]; hu[nV + e] = ubx[nV + e]; }
    P.n_ineq = n_ineq;
    bool okb = hipMalloc((void **)&h->lb, sizeof(double) * nb) == hipSuccess && hipMalloc((void **)&h->ub, sizeof(double) * nb) == hipSuccess;
    okb = okb && hipMemcpy(h->lb, hl, sizeof(double) * nb, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(h->ub, hu, sizeof(double) * nb, hipMemcpyHostToDevice) == hipSuccess;
    free(hl); free(hu);
    if (!okb) { (void)hipFree(h->ws); if (h->lb) (void)hipFree(h->lb); if (h->ub) (void)hipFree(h->ub); free(h); return NMPC_E_HIP; }
    P.lb = h->lb; P.ub = h->ub; P.lb0 = h->lb + nv; P.ub0 = h->ub + nv;
    *out = h;
    return NMPC_OK;
}

int32_t nmpc_lidar_destroy(nmpc_lidar_handle_t *h)
{
    if (!h) return NMPC_E_ARG;
    if (h->ws) (void)hipFree(h->ws);
    if (h->lb) (void)hipFree(h->lb);
    if (h->ub) (void)hipFree(h->ub);
    free(h);
    return NMPC_OK;
}

struct LidarDeviceScope {
    int prev = -1; bool ok = true, switched = false;
    explicit LidarDeviceScope(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev) { ok = hipSetDevice(dev) == hipSuccess; switched = ok; }
    }
    ~LidarDeviceScope() { if (switched) (void)hipSetDevice(prev); }
};

int32_t nmpc_lidar_solve_batch(nmpc_lidar_handle_t *h, int32_t B, const double *p, const double *w0, double *w_out, double *obj, int32_t *status,
                               int32_t *iters, double *kkt, void *stream)
{
    if (!h || B < 0 || B > h->max_batch) return NMPC_E_ARG;
    if (B == 0) return NMPC_OK;
    if (!p || !w0 || !w_out) return NMPC_E_ARG;
    LidarDeviceScope dev(h->device);
    if (!dev.ok) return NMPC_E_HIP;
    // the ray count of the scripts gets its own instantiation (component loops unrolled, loads of a stage issued together); any other count
    // runs the predicated one.  The dynamic-LDS limit is an attribute of the kernel FUNCTION, not of a handle: set per launch when this
    // handle needs more than HIP's default of 64 KB (a second handle with another horizon would otherwise change the limit under this one)
    // Two register budgets of the ten-ray kernel: one wave per SIMD (no spill; a lone wave iterates fastest: batches that fit the machine, whose
    // launch is its longest solve) and two (256 registers, 68 spilled dwords outside the recursions: 5 instead of 4 instances per CU by LDS —
    // batches beyond one instance per SIMD, whose launch is the batch's work).  Measured (round 4, V4, mean of three): B = 4096 192 k -> 199 k
    // solves/s with the second, B = 1024 115 k -> 109 k.
    const bool crowded = B > 4 * h->n_cu;
    auto kern = (h->cfg.R == 10) ? (crowded ? nmpc_lidar::lidar_solve_kernel<10, 2> : nmpc_lidar::lidar_solve_kernel<10, 1>) : nmpc_lidar::lidar_solve_kernel<-1, 1>;
    if (h->lds_bytes > 64 * 1024 && hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes) != hipSuccess) return NMPC_E_HIP;
    hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(64), h->lds_bytes, (hipStream_t)stream, h->P, B, p, w0, w_out, obj, status, iters, kkt, h->ws);
    return hipGetLastError() == hipSuccess ? NMPC_OK : NMPC_E_HIP;
}

int32_t nmpc_lidar_eval_batch(nmpc_lidar_handle_t *h, int32_t B, const double *p, const double *w, double *f, double *g, void *stream)
{
    if (!h || B < 0) return NMPC_E_ARG;
    if (B == 0) return NMPC_OK;
    if (!p || !w) return NMPC_E_ARG;
    LidarDeviceScope dev(h->device);
    if (!dev.ok) return NMPC_E_HIP;
    hipLaunchKernelGGL(nmpc_lidar::lidar_eval_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, h->P, B, p, w, f, g);
    return hipGetLastError() == hipSuccess ? NMPC_OK : NMPC_E_HIP;
}

int32_t nmpc_lidar_shift_batch(nmpc_lidar_handle_t *h, int32_t B, const double *w_in, double *w_next, void *stream)
{
    if (!h || B < 0) return NMPC_E_ARG;
    if (B == 0) return NMPC_OK;
    if (!w_in || !w_next || w_in == w_next) return NMPC_E_ARG;
    LidarDeviceScope dev(h->device);
    if (!dev.ok) return NMPC_E_HIP;
    const long total = (long)B * h->P.nvar;
    hipLaunchKernelGGL(nmpc_lidar::lidar_shift_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h->P, B, w_in, w_next);
    return hipGetLastError() == hipSuccess ? NMPC_OK : NMPC_E_HIP;
}

int32_t nmpc_lidar_scan_batch(int64_t B, int32_t R, int32_t K, const double *pose, const double *world, double scan_max, double *scan, void *stream)
{
    if (B < 0 || R < 0 || R > NMPC_LIDAR_MAX_RAYS || K < 0 || !(scan_max > 0.0)) return NMPC_E_ARG;
    if (B == 0 || R == 0) return NMPC_OK;
    if (!pose || !scan || (K > 0 && !world)) return NMPC_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return NMPC_E_HIP;
    const long total = (long)B * R;
    hipLaunchKernelGGL(nmpc_lidar::lidar_scan_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long)B, R, K, pose, world, scan_max, scan);
    return hipGetLastError() == hipSuccess ? NMPC_OK : NMPC_E_HIP;
}

int32_t nmpc_lidar_plant_batch(nmpc_lidar_handle_t *h, int32_t B, const double *p, const double *w_sol, double *pose_next, int32_t pose_stride, void *stream)
{
    if (!h || B < 0 || pose_stride < 0 || (pose_stride > 0 && pose_stride < 3)) return NMPC_E_ARG;
    if (B == 0) return NMPC_OK;
    if (!p || !w_sol || !pose_next) return NMPC_E_ARG;
    LidarDeviceScope dev(h->device);
    if (!dev.ok) return NMPC_E_HIP;
    hipLaunchKernelGGL(nmpc_lidar::lidar_plant_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h->P, B, p, w_sol, pose_next, pose_stride ? pose_stride : 3);
    return hipGetLastError() == hipSuccess ? NMPC_OK : NMPC_E_HIP;
}

}  // extern "C"
