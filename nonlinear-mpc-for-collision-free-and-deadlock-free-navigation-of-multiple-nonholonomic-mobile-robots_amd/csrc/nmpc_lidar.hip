// nmpc_lidar.hip — LIDAR-ray distance-state NMPC solve for gfx950 (include/nmpc_lidar.h).
//
// Reference blocks replaced (V4 = AllScripts/obs_avoid_static_first_scenario_v4.py; V3 = ..._v3.py is Nc = N, lw = 0):
//   NLP V4:78-151 (13 states per stage [x y theta d_1..d_10], move blocking U[:, min(k, Nc-1)], stage cost + 0.1 sum 1/d^2,
//   rows gx then gd with the 1-norm distance to the lidar points of stage 0), nlpsol('ipopt') V4:156-157 and the call V4:245.
//
// Mapping.  This NLP has ONE robot: the pose block is 3 x 3, the control block 2 x 2, and the R distance states of a stage are
// eliminated through their own linearised equality rows (d = ||p - pObs||_1), so the Newton system is a 3-state Riccati
// recursion whose matrices (5 x 5 with the held control of the move-blocked stages) fit the registers of one lane.  One
// WAVEFRONT solves one instance: the phases that are parallel over the horizon (evaluation, optimality error, condensed stage
// blocks, step lengths, merit function, update: ~95 % of the memory accesses) give every lane its own stages / variables of the
// instance's contiguous workspace in HBM/L2 (coalesced, wave reductions with v_readfirstlane so that every decision is a scalar
// branch), and the two recursions over the horizon run uniformly on all lanes with lane 0 storing.  (A first version gave every
// LANE its own instance: 64 serial solves per wavefront whose every access waited on HBM, 0.5 k solves/s — slower than the CPU.)
// Algorithm: the interior-point iteration of nmpc_kernels.hip / the oracle (barrier rule, fraction to the boundary, non-monotone
// l1-merit search, inertia shift on the control diagonal, barrier restart).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/nmpc_constants.h"      // cold-start retry and inertia constants shared with the main solver and the oracles
#include "../../include/nmpc_lidar.h"

namespace nmpc_lidar {

struct LParams {
    int32_t N, Nc, R, ns, max_iter, nvar, ng, np;
    double T, q[3], r[2], lw, tol, mu_init;
    const double *lb, *ub;      // device copies of the caller's lbx / ubx [n_var]
    int64_t S;                  // instances the workspace holds (each `total` doubles, contiguous)
    // element offsets of the per-instance arrays
    int64_t oV, oU, olam, oeta, oSL, oZL, oSU, oZU, oSLu, oZLu, oSUu, oZUu, odV, odU, olamn, oetan, oVt, oUt, osn, ocs, oHxx, ogx, oWd,
        ogdv, ohuu, ogu, ohvt, oKg, okff, opo, total;
};

#define W_(off, i) wsb[(off) + (i)]

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

// wave reductions whose result the compiler knows to be uniform (scalar control flow)
__device__ __forceinline__ double uni(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ double wsum_(double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return uni(v); }
__device__ __forceinline__ double wmax_(double v) { for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o)); return uni(v); }
__device__ __forceinline__ double wmin_(double v) { for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o)); return uni(v); }

// One wavefront per instance.  The stage-parallel phases (evaluation, optimality error, condensed blocks, step lengths, merit,
// update) spread stages / variables over the 64 lanes (coalesced accesses of the instance's contiguous workspace, wave
// reductions); the two recursions over the horizon (Riccati sweep with its 5 x 5 blocks in registers, forward sweep / adjoint
// recursion) run uniformly on all lanes, lane 0 storing.
#define N_LDS_STAGES(N_) ((N_) + 1)
#ifndef NMPC_LIDAR_UNROLL
#define NMPC_LIDAR_UNROLL 4      // stages of the forward / adjoint recursions unrolled together: their LDS operand reads issue as one batch
#endif
#ifndef NMPC_LIDAR_WAVES
#define NMPC_LIDAR_WAVES 2      // resident waves per SIMD the register budget is set for (measured: 1 -> 27.5 k, 2 -> 29.2 k, 3 -> 21.7 k, 4 -> 19.2 k solves/s)
#endif
__global__ __launch_bounds__(64, NMPC_LIDAR_WAVES) void lidar_solve_kernel(const LParams P, int B, const double *__restrict__ p_in, const double *__restrict__ w0,
                                                          double *__restrict__ w_out, double *__restrict__ obj_out, int32_t *__restrict__ status_out,
                                                          int32_t *__restrict__ iters_out, double *__restrict__ kkt_out, double *__restrict__ ws)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= B) return;
    const int N = P.N, Nc = P.Nc, R = P.R, ns = P.ns;
    const double T = P.T;
    double *wsb = ws + (size_t)b * (size_t)P.total;
    // LDS: what the three serial recursions over the horizon read per stage.  They run uniformly on all lanes and every stage
    // depends on the one before: an L2/HBM round trip per stage (~1 us) is their whole cost when the operands come from the
    // workspace, so a stage-parallel pass copies them here first and the recursions touch global memory only to store.
    extern __shared__ double lsm[];
    double *SB = lsm;                              // [N+1][13]  Hxx(4) gx(3) hvt sin cos pose(3)
    double *SC = SB + (size_t)(N_LDS_STAGES(P.N)) * 13;      // [Nc][16]   u(2) huu(2) gu(2) K(6) kff(2) du(2)
    double *SD = SC + (size_t)P.Nc * 16;           // [N+1][3]   pose step
    const double *pp = p_in + (size_t)b * P.np, *wi = w0 + (size_t)b * P.nvar;
    double *wo = w_out + (size_t)b * P.nvar;
    const double *lbv = P.lb, *ubv = P.ub, *lbu = P.lb + (size_t)(N + 1) * ns, *ubu = P.ub + (size_t)(N + 1) * ns;
    int64_t oV = P.oV, oVt = P.oVt, oU = P.oU, oUt = P.oUt;
    const double xs0 = pp[3], xs1 = pp[4], xs2 = pp[5];
    const int nV = (N + 1) * ns;
    auto cof = [&](int k) { return k < Nc - 1 ? k : Nc - 1; };
    auto gdist = [&](int m, double x, double y, double &sx, double &sy) {
        double ax = x - W_(P.opo, 2 * m), ay = y - W_(P.opo, 2 * m + 1);
        sx = sgn(ax); sy = sgn(ay);
        return fabs(ax) + fabs(ay);
    };
    // ---- load the start; X_0 (pose and scan) pinned to the parameters (V4:110-111); lidar points (V4:114-118)
    for (int e = lane; e < nV; e += 64) W_(oV, e) = (e < 3) ? pp[e] : ((e < ns) ? pp[6 + (e - 3)] : wi[e]);
    for (int e = lane; e < 2 * Nc; e += 64) W_(oU, e) = wi[(size_t)nV + e];
    for (int m = lane; m < R; m += 64) {
        double a = pp[2] + pp[6 + R + m], s, c;
        sincos(a, &s, &c);
        W_(P.opo, 2 * m) = pp[0] + pp[6 + m] * c; W_(P.opo, 2 * m + 1) = pp[1] + pp[6 + m] * s;
    }
    __syncthreads();
    {   // the pinned stage-0 variables must respect their own bounds
        double bad = 0.0;
        for (int c = lane; c < ns; c += 64) { double v = W_(oV, c); if (v < lbv[c] || v > ubv[c]) bad = 1.0; }
        if (wmax_(bad) > 0.0) {
            for (int e = lane; e < nV; e += 64) wo[e] = W_(oV, e);
            for (int e = lane; e < 2 * Nc; e += 64) wo[(size_t)nV + e] = W_(oU, e);
            if (lane == 0) {
                if (obj_out) obj_out[b] = NAN;
                if (status_out) status_out[b] = NMPC_STATUS_INFEASIBLE_X0;
                if (iters_out) iters_out[b] = 0;
                if (kkt_out) kkt_out[b] = INFINITY;
            }
            return;
        }
    }
    int n_ineq;
    { double c = 0.0; for (int e = ns + lane; e < nV; e += 64) c += (isfinite(lbv[e]) ? 1.0 : 0.0) + (isfinite(ubv[e]) ? 1.0 : 0.0); n_ineq = 4 * Nc + (int)wsum_(c); }

    // objective, sum / max of the equality residuals at (oVx, oUx): one stage per lane; optionally the trig cache
    auto eval_point = [&](int64_t oVx, int64_t oUx, bool trig, double &th, double &ec) {
        double f = 0.0, t_ = 0.0, e_ = 0.0;
        for (int k = lane; k < N; k += 64) {
            const int j = cof(k);
            const double x = W_(oVx, k * ns), y = W_(oVx, k * ns + 1), t = W_(oVx, k * ns + 2), u0 = W_(oUx, 2 * j), u1 = W_(oUx, 2 * j + 1);
            const double xn = W_(oVx, (k + 1) * ns), yn = W_(oVx, (k + 1) * ns + 1), tn = W_(oVx, (k + 1) * ns + 2);
            double s, c;
            sincos(t, &s, &c);
            if (trig) { W_(P.osn, k) = s; W_(P.ocs, k) = c; }
            const double c0 = xn - (x + T * u0 * c), c1 = yn - (y + T * u0 * s), c2 = tn - (t + T * u1);
            t_ += fabs(c0) + fabs(c1) + fabs(c2); e_ = fmax(e_, fmax(fabs(c0), fmax(fabs(c1), fabs(c2))));
            f += P.q[0] * (x - xs0) * (x - xs0); f += P.q[1] * (y - xs1) * (y - xs1); f += P.q[2] * (t - xs2) * (t - xs2);
            f += P.r[0] * u0 * u0 + P.r[1] * u1 * u1;
            for (int m = 0; m < R; m++) {
                if (P.lw != 0.0) { double d = W_(oVx, k * ns + 3 + m); f += P.lw / (d * d); }
                double sx, sy, e = W_(oVx, (k + 1) * ns + 3 + m) - gdist(m, xn, yn, sx, sy);
                t_ += fabs(e); e_ = fmax(e_, fabs(e));
            }
        }
        th = wsum_(t_); ec = wmax_(e_);
        return wsum_(f);
    };

    double mu = P.mu_init, f = 0.0, th0 = 0.0, e_c = 0.0;
    int it = 0, n_tiny = 0, n_restart = 0, status = NMPC_STATUS_MAX_ITER;
    // cold-start retry (the restoration of last resort of the main solver, same constants; see oracle/lidar_oracle.c)
    int n_cold = 0, it_base = 0;
    bool cold = false;
    bool need_shift = false, restarting = false;
    double delta_last = 0.0, nu_pen = 1.0, kkt = INFINITY;
    double mh0 = 0, mh1 = 0, mh2 = 0, mh_mu = -1, mh_nu = -1;
    int mcount = 0;

    auto cold_retry = [&]() { cold = true; n_cold++; it_base = it; restarting = true; mu = (n_cold == 1) ? P.mu_init : 10.0 * P.mu_init; n_tiny = 0; n_restart = 0; };
    for (;;) {      // (re)start of the barrier iteration
        if (cold) {       // the reference's cold start (V4:184-196): X_k = x0 (pose and scan), U = 0
            for (int e = ns + lane; e < nV; e += 64) W_(oV, e) = W_(oV, e % ns);
            for (int e = lane; e < 2 * Nc; e += 64) W_(oU, e) = 0.0;
            cold = false;
            __syncthreads();
        }
        for (int e = ns + lane; e < nV; e += 64) {
            const double lo = lbv[e], hi = ubv[e], v = push_in(W_(oV, e), lo, hi);
            W_(oV, e) = v;
            const double sl = isfinite(lo) ? fmax(v - lo, 1e-12) : 1.0, su = isfinite(hi) ? fmax(hi - v, 1e-12) : 1.0;
            W_(P.oSL, e) = sl; W_(P.oZL, e) = isfinite(lo) ? mu / sl : 0.0;
            W_(P.oSU, e) = su; W_(P.oZU, e) = isfinite(hi) ? mu / su : 0.0;
        }
        for (int e = lane; e < 2 * Nc; e += 64) {
            const double u = push_in(W_(oU, e), lbu[e], ubu[e]), sl = fmax(u - lbu[e], 1e-12), su = fmax(ubu[e] - u, 1e-12);
            W_(oU, e) = u;
            W_(P.oSLu, e) = sl; W_(P.oZLu, e) = mu / sl; W_(P.oSUu, e) = su; W_(P.oZUu, e) = mu / su;
        }
        for (int e = lane; e < (N + 1) * 3; e += 64) W_(P.olam, e) = 0.0;
        for (int e = lane; e < (N + 1) * R; e += 64) W_(P.oeta, e) = 0.0;
        __syncthreads();
        f = eval_point(oV, oU, true, th0, e_c);
        __syncthreads();
        delta_last = 0.0; nu_pen = 1.0; need_shift = false; mcount = 0; restarting = false;

        for (;;) {
            // ---- A. optimality error (IPOPT eq. 5): one stage per lane
            double e_d = 0.0, e_h = 0.0, zsum = 0.0, lsum = 0.0, cmax = 0.0, cmin = INFINITY, pu0 = 0.0, pu1 = 0.0;
            auto ctrl_rows = [&](int j, double ru0, double ru1) {      // stationarity / complementarity of control j given its Lagrangian gradient
                for (int e = 0; e < 2; e++) {
                    const int o = 2 * j + e;
                    const double zl = W_(P.oZLu, o), zu = W_(P.oZUu, o), sl = W_(P.oSLu, o), su = W_(P.oSUu, o), u = W_(oU, o);
                    const double ru = (e ? ru1 : ru0) - (zl - zu);
                    e_d = fmax(e_d, fabs(ru)); zsum += zl + zu;
                    cmax = fmax(cmax, fmax(sl * zl, su * zu)); cmin = fmin(cmin, fmin(sl * zl, su * zu));
                    e_h = fmax(e_h, fmax(fabs((u - lbu[o]) - sl), fabs((ubu[o] - u) - su)));
                }
            };
            for (int k = lane; k <= N; k += 64) {
                if (k >= 1) {
                    const double x = W_(oV, k * ns), y = W_(oV, k * ns + 1), t = W_(oV, k * ns + 2);
                    double r0 = W_(P.olam, 3 * k), r1 = W_(P.olam, 3 * k + 1), r2 = W_(P.olam, 3 * k + 2);
                    lsum += fabs(r0) + fabs(r1) + fabs(r2);
                    if (k < N) {
                        const double l0 = W_(P.olam, 3 * k + 3), l1 = W_(P.olam, 3 * k + 4), l2 = W_(P.olam, 3 * k + 5), u0 = W_(oU, 2 * cof(k));
                        const double a = -T * u0 * W_(P.osn, k), bq = T * u0 * W_(P.ocs, k);
                        r0 += 2 * P.q[0] * (x - xs0) - l0; r1 += 2 * P.q[1] * (y - xs1) - l1; r2 += 2 * P.q[2] * (t - xs2) - l2;
                        r2 -= a * l0 + bq * l1;
                    }
                    for (int m = 0; m < R; m++) {
                        double sx, sy; gdist(m, x, y, sx, sy);
                        const double et = W_(P.oeta, k * R + m), d = W_(oV, k * ns + 3 + m);
                        r0 -= et * sx; r1 -= et * sy;
                        double rd = et + ((k < N && P.lw != 0.0) ? -2.0 * P.lw / (d * d * d) : 0.0);
                        rd -= W_(P.oZL, k * ns + 3 + m) - W_(P.oZU, k * ns + 3 + m);
                        e_d = fmax(e_d, fabs(rd)); lsum += fabs(et);
                    }
                    r0 -= W_(P.oZL, k * ns) - W_(P.oZU, k * ns); r1 -= W_(P.oZL, k * ns + 1) - W_(P.oZU, k * ns + 1); r2 -= W_(P.oZL, k * ns + 2) - W_(P.oZU, k * ns + 2);
                    e_d = fmax(e_d, fmax(fabs(r0), fmax(fabs(r1), fabs(r2))));
                    for (int c = 0; c < ns; c++) {
                        const int e = k * ns + c;
                        const double v = W_(oV, e), lo = lbv[e], hi = ubv[e];
                        if (isfinite(lo)) { double sl = W_(P.oSL, e), zl = W_(P.oZL, e), pz = sl * zl; zsum += zl; cmax = fmax(cmax, pz); cmin = fmin(cmin, pz); e_h = fmax(e_h, fabs((v - lo) - sl)); }
                        if (isfinite(hi)) { double su = W_(P.oSU, e), zu = W_(P.oZU, e), pz = su * zu; zsum += zu; cmax = fmax(cmax, pz); cmin = fmin(cmin, pz); e_h = fmax(e_h, fabs((hi - v) - su)); }
                    }
                }
                if (k < N) {       // control rows: stage k contributes to the Lagrangian gradient of control cof(k)
                    const int j = cof(k);
                    const double g0 = 2 * P.r[0] * W_(oU, 2 * j) - T * (W_(P.ocs, k) * W_(P.olam, 3 * k + 3) + W_(P.osn, k) * W_(P.olam, 3 * k + 4));
                    const double g1 = 2 * P.r[1] * W_(oU, 2 * j + 1) - T * W_(P.olam, 3 * k + 5);
                    if (k < Nc - 1) ctrl_rows(j, g0, g1); else { pu0 += g0; pu1 += g1; }
                }
            }
            pu0 = wsum_(pu0); pu1 = wsum_(pu1);
            if (lane == 0) ctrl_rows(Nc - 1, pu0, pu1);       // the held control: the sum over its stages
            e_d = wmax_(e_d); e_h = wmax_(e_h); zsum = wsum_(zsum); lsum = wsum_(lsum); cmax = wmax_(cmax); cmin = wmin_(cmin);
            const double smax = 100.0;
            const double s_d = fmax(smax, (lsum + zsum) / (double)(N * ns + n_ineq)) / smax;
            const double s_c = fmax(smax, zsum / (double)(n_ineq > 0 ? n_ineq : 1)) / smax;
            const double E0 = fmax(fmax(e_d / s_d, e_c), fmax(e_h, cmax / s_c));
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
            const double tau = fmax(0.99, 1.0 - mu);

            // ---- B0. condensed stage blocks, one stage per lane.  slot: v = mu/s - sigma (h - s), sigma = z/s
            for (int k = 1 + lane; k <= N; k += 64) {
                const double x = W_(oV, k * ns), y = W_(oV, k * ns + 1);
                double H0 = 0.0, H1 = 0.0, H2 = 0.0, hd[3] = {0, 0, 0}, g[3] = {0, 0, 0};
                if (k < N) {
                    hd[0] = 2 * P.q[0]; hd[1] = 2 * P.q[1]; hd[2] = 2 * P.q[2];
                    g[0] = 2 * P.q[0] * (x - xs0); g[1] = 2 * P.q[1] * (y - xs1); g[2] = 2 * P.q[2] * (W_(oV, k * ns + 2) - xs2);
                }
                for (int c = 0; c < ns; c++) {
                    const int e = k * ns + c;
                    const double v = W_(oV, e), lo = lbv[e], hi = ubv[e];
                    double hs = 0.0, gs = 0.0;
                    if (isfinite(lo)) { double sl = W_(P.oSL, e), sg = W_(P.oZL, e) / sl; hs += sg; gs -= mu / sl - sg * ((v - lo) - sl); }
                    if (isfinite(hi)) { double su = W_(P.oSU, e), sg = W_(P.oZU, e) / su; hs += sg; gs += mu / su - sg * ((hi - v) - su); }
                    if (c < 3) { hd[c] += hs; g[c] += gs; }
                    else {
                        const int m = c - 3;
                        const bool cost = k < N && P.lw != 0.0;
                        const double Wd = hs + (cost ? 6.0 * P.lw / (v * v * v * v) : 0.0), gd = gs + (cost ? -2.0 * P.lw / (v * v * v) : 0.0);
                        W_(P.oWd, k * R + m) = Wd; W_(P.ogdv, k * R + m) = gd;
                        double sx, sy;
                        const double re = gdist(m, x, y, sx, sy) - v, tq = gd + Wd * re;      // linearised row: dd = G dx + re
                        g[0] += sx * tq; g[1] += sy * tq;
                        H0 += Wd * sx * sx; H1 += Wd * sx * sy; H2 += Wd * sy * sy;
                    }
                }
                double H3 = hd[2];
                if (k < N) H3 += T * W_(oU, 2 * cof(k)) * (W_(P.olam, 3 * k + 3) * W_(P.ocs, k) + W_(P.olam, 3 * k + 4) * W_(P.osn, k));
                W_(P.oHxx, 4 * k) = H0 + hd[0]; W_(P.oHxx, 4 * k + 1) = H1; W_(P.oHxx, 4 * k + 2) = H2 + hd[1]; W_(P.oHxx, 4 * k + 3) = H3;
                W_(P.ogx, 3 * k) = g[0]; W_(P.ogx, 3 * k + 1) = g[1]; W_(P.ogx, 3 * k + 2) = g[2];
            }
            for (int k = lane; k < N; k += 64) W_(P.ohvt, k) = T * (W_(P.olam, 3 * k + 3) * W_(P.osn, k) - W_(P.olam, 3 * k + 4) * W_(P.ocs, k));
            for (int o = lane; o < 2 * Nc; o += 64) {
                const int j = o >> 1, e = o & 1, cnt = (j < Nc - 1) ? 1 : N - Nc + 1;
                const double sl = W_(P.oSLu, o), su = W_(P.oSUu, o), zl = W_(P.oZLu, o), zu = W_(P.oZUu, o), u = W_(oU, o);
                W_(P.ohuu, o) = cnt * 2 * P.r[e] + zl / sl + zu / su;
                const double vl = mu / sl - zl / sl * ((u - lbu[o]) - sl), vu = mu / su - zu / su * ((ubu[o] - u) - su);
                W_(P.ogu, o) = cnt * 2 * P.r[e] * u - (vl - vu);
            }
            __syncthreads();
            for (int k = lane; k <= N; k += 64) {
                double *sb = SB + k * 13;
                sb[0] = W_(P.oHxx, 4 * k); sb[1] = W_(P.oHxx, 4 * k + 1); sb[2] = W_(P.oHxx, 4 * k + 2); sb[3] = W_(P.oHxx, 4 * k + 3);
                sb[4] = W_(P.ogx, 3 * k); sb[5] = W_(P.ogx, 3 * k + 1); sb[6] = W_(P.ogx, 3 * k + 2);
                sb[7] = (k < N) ? W_(P.ohvt, k) : 0.0; sb[8] = (k < N) ? W_(P.osn, k) : 0.0; sb[9] = (k < N) ? W_(P.ocs, k) : 0.0;
                sb[10] = W_(oV, k * ns); sb[11] = W_(oV, k * ns + 1); sb[12] = W_(oV, k * ns + 2);
            }
            for (int j = lane; j < Nc; j += 64) {
                double *sc = SC + j * 16;
                sc[0] = W_(oU, 2 * j); sc[1] = W_(oU, 2 * j + 1); sc[2] = W_(P.ohuu, 2 * j); sc[3] = W_(P.ohuu, 2 * j + 1);
                sc[4] = W_(P.ogu, 2 * j); sc[5] = W_(P.ogu, 2 * j + 1);
            }
            __syncthreads();

            // ---- B. Riccati sweep on z = (pose (3), held control (2)) with inertia correction; uniform on all lanes, stage in registers
            double delta = need_shift ? fmax(1e-20, 0.25 * delta_last) : 0.0;
            int ntry = 0;
            bool ok;
            for (;;) {
                ok = true;
                double P5[25], p5[5];
#pragma unroll
                for (int z = 0; z < 25; z++) P5[z] = 0.0;
#pragma unroll
                for (int z = 0; z < 5; z++) p5[z] = 0.0;
                P5[0] = SB[N * 13]; P5[1] = P5[5] = SB[N * 13 + 1]; P5[6] = SB[N * 13 + 2]; P5[12] = SB[N * 13 + 3];
                p5[0] = SB[N * 13 + 4]; p5[1] = SB[N * 13 + 5]; p5[2] = SB[N * 13 + 6];
                for (int k = N - 1; k >= 0; k--) {
                    const int j = cof(k);
                    const double *sb = SB + k * 13, *sc = SC + j * 16;
                    const double u0 = sc[0], u1 = sc[1], s = sb[8], c = sb[9], a = -T * u0 * s, bq = T * u0 * c;
                    const double cd0 = sb[13 + 10] - (sb[10] + T * u0 * c), cd1 = sb[13 + 11] - (sb[11] + T * u0 * s),
                                 cd2 = sb[13 + 12] - (sb[12] + T * u1);
                    // At = [[A, B], [0, I]] acts on columns, then on rows: G = P At, M = At^T G (At is sparse: written out)
                    double pb[5], G5[25], Mx[25], mv[5];
#pragma unroll
                    for (int r_ = 0; r_ < 5; r_++) {
                        pb[r_] = p5[r_] - (P5[r_ * 5] * cd0 + P5[r_ * 5 + 1] * cd1 + P5[r_ * 5 + 2] * cd2);
                        const double c0_ = P5[r_ * 5], c1_ = P5[r_ * 5 + 1], c2_ = P5[r_ * 5 + 2], c3_ = P5[r_ * 5 + 3], c4_ = P5[r_ * 5 + 4];
                        G5[r_ * 5] = c0_; G5[r_ * 5 + 1] = c1_; G5[r_ * 5 + 2] = a * c0_ + bq * c1_ + c2_;
                        G5[r_ * 5 + 3] = T * c * c0_ + T * s * c1_ + c3_; G5[r_ * 5 + 4] = T * c2_ + c4_;
                    }
#pragma unroll
                    for (int q_ = 0; q_ < 5; q_++) {
                        const double r0_ = G5[q_], r1_ = G5[5 + q_], r2_ = G5[10 + q_], r3_ = G5[15 + q_], r4_ = G5[20 + q_];
                        Mx[q_] = r0_; Mx[5 + q_] = r1_; Mx[10 + q_] = a * r0_ + bq * r1_ + r2_;
                        Mx[15 + q_] = T * c * r0_ + T * s * r1_ + r3_; Mx[20 + q_] = T * r2_ + r4_;
                    }
                    mv[0] = pb[0]; mv[1] = pb[1]; mv[2] = a * pb[0] + bq * pb[1] + pb[2]; mv[3] = T * c * pb[0] + T * s * pb[1] + pb[3]; mv[4] = T * pb[2] + pb[4];
                    if (k >= 1) {
                        const double h1 = sb[1];
                        Mx[0] += sb[0]; Mx[1] += h1; Mx[5] += h1; Mx[6] += sb[2]; Mx[12] += sb[3];
                        mv[0] += sb[4]; mv[1] += sb[5]; mv[2] += sb[6];
                    }
                    { const double hv = sb[7]; Mx[2 * 5 + 3] += hv; Mx[3 * 5 + 2] += hv; }
                    if (k <= Nc - 1) {        // the stage where control j is decided carries its whole diagonal / gradient
                        Mx[18] += sc[2] + delta; Mx[24] += sc[3] + delta;
                        mv[3] += sc[4]; mv[4] += sc[5];
                        const double dv = Mx[18];
                        if (!(uni(dv) > 0.0)) { ok = false; break; }
                        const double rdv = 1.0 / dv;          // two reciprocals per stage instead of nine divisions (the recursion is uniform: every lane pays them)
                        const double l43 = Mx[23] * rdv, d1o = Mx[24], d1 = d1o - l43 * Mx[19];
                        if (!(uni(d1) > 1e-9 * fabs(uni(d1o))) || !(uni(d1) > 0.0)) { ok = false; break; }
                        const double rd1 = 1.0 / d1;
                        double Kk[6], kk[2];
#pragma unroll
                        for (int q_ = 0; q_ < 4; q_++) {
                            const double r3 = (q_ < 3) ? Mx[15 + q_] : mv[3], r4 = (q_ < 3) ? Mx[20 + q_] : mv[4];
                            const double y4 = (r4 - l43 * r3) * rd1, y3 = (r3 - Mx[19] * y4) * rdv;
                            if (q_ < 3) { Kk[q_] = -y3; Kk[3 + q_] = -y4; } else { kk[0] = -y3; kk[1] = -y4; }
                        }
                        if (lane == 0) {
                            double *sk = SC + j * 16 + 6;
#pragma unroll
                            for (int q_ = 0; q_ < 6; q_++) sk[q_] = Kk[q_];
                            sk[6] = kk[0]; sk[7] = kk[1];
                        }
                        double Pn[9], pn[3];
#pragma unroll
                        for (int r_ = 0; r_ < 3; r_++) {
#pragma unroll
                            for (int q_ = 0; q_ < 3; q_++) Pn[r_ * 3 + q_] = Mx[r_ * 5 + q_] + Mx[r_ * 5 + 3] * Kk[q_] + Mx[r_ * 5 + 4] * Kk[3 + q_];
                            pn[r_] = mv[r_] + Mx[r_ * 5 + 3] * kk[0] + Mx[r_ * 5 + 4] * kk[1];
                        }
#pragma unroll
                        for (int z = 0; z < 25; z++) P5[z] = 0.0;
#pragma unroll
                        for (int z = 0; z < 5; z++) p5[z] = 0.0;
#pragma unroll
                        for (int r_ = 0; r_ < 3; r_++) {
#pragma unroll
                            for (int q_ = 0; q_ < 3; q_++) P5[r_ * 5 + q_] = 0.5 * (Pn[r_ * 3 + q_] + Pn[q_ * 3 + r_]);
                            p5[r_] = pn[r_];
                        }
                    } else {                  // held control: it stays a parameter of the cost-to-go
#pragma unroll
                        for (int r_ = 0; r_ < 5; r_++) {
#pragma unroll
                            for (int q_ = 0; q_ < 5; q_++) P5[r_ * 5 + q_] = (q_ == r_) ? Mx[r_ * 5 + q_] : 0.5 * (Mx[r_ * 5 + q_] + Mx[q_ * 5 + r_]);
                            p5[r_] = mv[r_];
                        }
                    }
                }
                if (ok) break;
                ntry++;
                if (delta == 0.0) delta = (delta_last == 0.0) ? 1e-4 : fmax(1e-20, delta_last / 3.0);
                else delta *= (delta_last == 0.0) ? 100.0 : NMPC_SHIFT_ESCALATION;
                if (delta > 1e20) break;
            }
            if (!ok) { if (n_cold < NMPC_COLD_RETRIES) { cold_retry(); it++; break; } status = NMPC_STATUS_NUMERIC; break; }
            if (delta > 0.0) delta_last = delta;
            need_shift = delta > 0.0 && (ntry > 0 || delta > 1e-6);
            __syncthreads();

            // ---- C. forward sweep: the pose recursion uniform on all lanes (operands from LDS), then the R distance states of every
            //      stage in parallel, one (stage, ray) item per lane
            for (int c = lane; c < ns; c += 64) W_(P.odV, c) = 0.0;
            {
                double dx0 = 0.0, dx1 = 0.0, dx2 = 0.0, du0 = 0.0, du1 = 0.0;
                if (lane == 0) { SD[0] = 0.0; SD[1] = 0.0; SD[2] = 0.0; }
#pragma unroll NMPC_LIDAR_UNROLL
                for (int k = 0; k < N; k++) {
                    const int j = cof(k);
                    const double *sb = SB + k * 13;
                    double *sc = SC + j * 16;
                    if (k <= Nc - 1) {
                        du0 = sc[12] + sc[6] * dx0 + sc[7] * dx1 + sc[8] * dx2;
                        du1 = sc[13] + sc[9] * dx0 + sc[10] * dx1 + sc[11] * dx2;
                        if (lane == 0) { W_(P.odU, 2 * j) = du0; W_(P.odU, 2 * j + 1) = du1; sc[14] = du0; sc[15] = du1; }
                    }
                    const double u0 = sc[0], u1 = sc[1], s = sb[8], c = sb[9];
                    const double xn = sb[13 + 10], yn = sb[13 + 11], tn = sb[13 + 12];
                    const double n0 = dx0 + (-T * u0 * s) * dx2 + T * c * du0 - (xn - (sb[10] + T * u0 * c));
                    const double n1 = dx1 + (T * u0 * c) * dx2 + T * s * du0 - (yn - (sb[11] + T * u0 * s));
                    const double n2 = dx2 + T * du1 - (tn - (sb[12] + T * u1));
                    dx0 = n0; dx1 = n1; dx2 = n2;
                    if (lane == 0) {
                        W_(P.odV, (k + 1) * ns) = n0; W_(P.odV, (k + 1) * ns + 1) = n1; W_(P.odV, (k + 1) * ns + 2) = n2;
                        SD[3 * (k + 1)] = n0; SD[3 * (k + 1) + 1] = n1; SD[3 * (k + 1) + 2] = n2;
                    }
                }
            }
            __syncthreads();
            for (int e = lane; e < N * R; e += 64) {      // dd = G dx + (g - d) at stage k1 = 1..N
                const int k1 = 1 + e / R, m = e - (k1 - 1) * R;
                double sx, sy;
                const double gg = gdist(m, SB[k1 * 13 + 10], SB[k1 * 13 + 11], sx, sy);
                W_(P.odV, k1 * ns + 3 + m) = sx * SD[3 * k1] + sy * SD[3 * k1 + 1] + (gg - W_(oV, k1 * ns + 3 + m));
            }
            __syncthreads();
            // ---- multipliers of the QP: lambda+ by the adjoint recursion (uniform), eta+ from the distance rows (one per lane)
            double mult_max = 0.0;
            {
                double ln0 = 0.0, ln1 = 0.0, ln2 = 0.0;
#pragma unroll NMPC_LIDAR_UNROLL
                for (int k = N; k >= 1; k--) {
                    const double *sb = SB + k * 13;
                    const double dx0 = SD[3 * k], dx1 = SD[3 * k + 1], dx2 = SD[3 * k + 2];
                    const double H0 = sb[0], H1 = sb[1], H2 = sb[2], H3 = sb[3];
                    double l0 = -(sb[4] + H0 * dx0 + H1 * dx1), l1 = -(sb[5] + H1 * dx0 + H2 * dx1), l2 = -(sb[6] + H3 * dx2);
                    if (k < N) {
                        const double *sc = SC + cof(k) * 16;
                        const double u0 = sc[0];
                        l0 += ln0; l1 += ln1;
                        l2 += ln2 + (-T * u0 * sb[8]) * ln0 + (T * u0 * sb[9]) * ln1 - sb[7] * sc[14];
                    }
                    ln0 = l0; ln1 = l1; ln2 = l2;
                    if (lane == 0) { W_(P.olamn, 3 * k) = l0; W_(P.olamn, 3 * k + 1) = l1; W_(P.olamn, 3 * k + 2) = l2; }
                    mult_max = fmax(mult_max, fmax(fabs(l0), fmax(fabs(l1), fabs(l2))));
                }
                for (int e = R + lane; e < (N + 1) * R; e += 64) {
                    const int k = e / R, m = e - k * R;
                    const double v = -(W_(P.ogdv, e) + W_(P.oWd, e) * W_(P.odV, k * ns + 3 + m));
                    W_(P.oetan, e) = v; mult_max = fmax(mult_max, fabs(v));
                }
                mult_max = wmax_(mult_max);
            }
            // ---- D. fraction to the boundary; directional derivative of the barrier function (one variable per lane)
            double a_p = 1.0, a_d = 1.0, dphi = 0.0, lgs = 0.0, thh = 0.0;
            auto slot = [&](double s_, double z_, double h_, double jd_) {
                const double ds_ = jd_ + (h_ - s_), dz_ = (mu - s_ * z_ - z_ * ds_) / s_;
                if (ds_ < 0.0) a_p = fmin(a_p, -tau * s_ / ds_);
                if (dz_ < 0.0) a_d = fmin(a_d, -tau * z_ / dz_);
                dphi -= mu * ds_ / s_; lgs += log(s_); thh += fabs(h_ - s_);
            };
            for (int e = ns + lane; e < nV; e += 64) {
                const double v = W_(oV, e), dv = W_(P.odV, e), lo = lbv[e], hi = ubv[e];
                if (isfinite(lo)) slot(W_(P.oSL, e), W_(P.oZL, e), v - lo, dv);
                if (isfinite(hi)) slot(W_(P.oSU, e), W_(P.oZU, e), hi - v, -dv);
                const int k = e / ns, c = e - k * ns;
                if (k < N) {
                    if (c < 3) dphi += 2 * P.q[c] * (v - (c == 0 ? xs0 : (c == 1 ? xs1 : xs2))) * dv;
                    else if (P.lw != 0.0) dphi += -2.0 * P.lw / (v * v * v) * dv;
                }
            }
            for (int o = lane; o < 2 * Nc; o += 64) {
                const double u = W_(oU, o), du = W_(P.odU, o);
                slot(W_(P.oSLu, o), W_(P.oZLu, o), u - lbu[o], du);
                slot(W_(P.oSUu, o), W_(P.oZUu, o), ubu[o] - u, -du);
                const int j = o >> 1, cnt = (j < Nc - 1) ? 1 : N - Nc + 1;
                dphi += cnt * 2 * P.r[o & 1] * u * du;
            }
            a_p = wmin_(a_p); a_d = wmin_(a_d); dphi = wsum_(dphi); lgs = wsum_(lgs); thh = wsum_(thh);
            // ---- E. l1 merit backtracking (non-monotone reference: max of the last merit values of this barrier problem)
            const double theta0 = th0 + thh, phi0 = f - mu * lgs;
            if (theta0 > 0.0) {
                const double nut = fmin(dphi / ((1.0 - 0.1) * theta0), mult_max / (1.0 - 0.1));
                nu_pen = fmax(1.0, 0.5 * nu_pen);
                if (nu_pen < nut) nu_pen = nut + 1.0;
            }
            const double Dm = dphi - nu_pen * theta0;
            double alpha = a_p, ft = f, tht = th0, ect = e_c;
            if (mh_mu != mu || mh_nu != nu_pen) { mcount = 0; mh_mu = mu; mh_nu = nu_pen; }
            const double m0 = phi0 + nu_pen * theta0;
            double mref = m0;
            if (mcount > 0) mref = fmax(mref, mh0);
            if (mcount > 1) mref = fmax(mref, mh1);
            if (mcount > 2) mref = fmax(mref, mh2);
            mh2 = mh1; mh1 = mh0; mh0 = m0; if (mcount < 3) mcount++;
            for (int ls = 0; ls < 30; ls++) {
                double lgt = 0.0, tb = 0.0;
                auto trial = [&](double s_, double h0_, double jd_, double ht_) { const double st_ = s_ + alpha * (jd_ + (h0_ - s_)); lgt += log(st_); tb += fabs(ht_ - st_); };
                for (int e = lane; e < nV; e += 64) {
                    const double v = W_(oV, e), dv = W_(P.odV, e), vt = v + alpha * dv;
                    W_(oVt, e) = vt;
                    if (e >= ns) {
                        const double lo = lbv[e], hi = ubv[e];
                        if (isfinite(lo)) trial(W_(P.oSL, e), v - lo, dv, vt - lo);
                        if (isfinite(hi)) trial(W_(P.oSU, e), hi - v, -dv, hi - vt);
                    }
                }
                for (int o = lane; o < 2 * Nc; o += 64) {
                    const double u = W_(oU, o), du = W_(P.odU, o), ut = u + alpha * du;
                    W_(oUt, o) = ut;
                    trial(W_(P.oSLu, o), u - lbu[o], du, ut - lbu[o]);
                    trial(W_(P.oSUu, o), ubu[o] - u, -du, ubu[o] - ut);
                }
                __syncthreads();
                ft = eval_point(oVt, oUt, false, tht, ect);
                lgt = wsum_(lgt); tb = wsum_(tb);
                if ((ft - mu * lgt) + nu_pen * (tht + tb) <= mref + 1e-4 * alpha * Dm + 1e-13 * fabs(phi0)) break;
                if (ls < 29) alpha *= 0.5;
                __syncthreads();
            }
            a_d = fmin(a_d, alpha);
            n_tiny = (alpha < 1e-10) ? n_tiny + 1 : 0;
            // ---- G. accept: duals and slacks (they need the old primal point), then the primal point
            auto upd = [&](int64_t oS, int64_t oZ, int e, double h_, double jd_) {
                const double s_ = W_(oS, e), z_ = W_(oZ, e);
                const double ds_ = jd_ + (h_ - s_), dz_ = (mu - s_ * z_ - z_ * ds_) / s_;
                const double sn_ = s_ + alpha * ds_, zn_ = z_ + a_d * dz_;
                W_(oS, e) = sn_; W_(oZ, e) = fmin(fmax(zn_, mu / (1e10 * sn_)), 1e10 * mu / sn_);
            };
            for (int e = ns + lane; e < nV; e += 64) {
                const double v = W_(oV, e), dv = W_(P.odV, e), lo = lbv[e], hi = ubv[e];
                if (isfinite(lo)) upd(P.oSL, P.oZL, e, v - lo, dv);
                if (isfinite(hi)) upd(P.oSU, P.oZU, e, hi - v, -dv);
            }
            for (int o = lane; o < 2 * Nc; o += 64) {
                const double u = W_(oU, o), du = W_(P.odU, o);
                upd(P.oSLu, P.oZLu, o, u - lbu[o], du);
                upd(P.oSUu, P.oZUu, o, ubu[o] - u, -du);
            }
            { int64_t t_ = oV; oV = oVt; oVt = t_; t_ = oU; oU = oUt; oUt = t_; }      // the trial point of the accepted step length is in (oVt, oUt)
            for (int e = 3 + lane; e < (N + 1) * 3; e += 64) W_(P.olam, e) += alpha * (W_(P.olamn, e) - W_(P.olam, e));
            for (int e = R + lane; e < (N + 1) * R; e += 64) W_(P.oeta, e) += alpha * (W_(P.oetan, e) - W_(P.oeta, e));
            __syncthreads();
            f = eval_point(oV, oU, true, th0, e_c);
            __syncthreads();
            it++;
            if (n_tiny >= 5) {
                if (n_restart >= 3) { if (n_cold < NMPC_COLD_RETRIES) { cold_retry(); break; } status = NMPC_STATUS_STALLED; break; }
                n_restart++; n_tiny = 0; mu = fmax(mu, P.mu_init); restarting = true;
                break;
            }
        }
        if (!restarting) break;
    }
    __syncthreads();
    for (int e = lane; e < nV; e += 64) wo[e] = W_(oV, e);
    for (int e = lane; e < 2 * Nc; e += 64) wo[(size_t)nV + e] = W_(oU, e);
    if (lane == 0) {
        if (obj_out) obj_out[b] = f;
        if (status_out) status_out[b] = status;
        if (iters_out) iters_out[b] = it;
        if (kkt_out) kkt_out[b] = kkt;
    }
}

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
        if (t > 0.0 && h2 < r2) best = fmin(best, t - sqrt(r2 - h2));
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
    h->cfg = *cfg; h->max_batch = max_batch;
    h->lds_bytes = sizeof(double) * ((size_t)(N + 1) * 13 + (size_t)Nc * 16 + (size_t)(N + 1) * 3);
    if (h->lds_bytes > 160 * 1024) { free(h); return NMPC_E_ARG; }       // horizon beyond the LDS of a CU (N ~ 1000 with Nc = N / 2)
    nmpc_lidar::LParams &P = h->P;
    memset(&P, 0, sizeof(P));
    P.N = N; P.Nc = Nc; P.R = R; P.ns = ns; P.max_iter = cfg->max_iter; P.nvar = nv; P.ng = nmpc_lidar_n_g(cfg); P.np = nmpc_lidar_n_p(cfg);
    P.T = cfg->T; P.lw = cfg->lw; P.tol = cfg->tol; P.mu_init = cfg->mu_init;
    for (int i = 0; i < 3; i++) P.q[i] = cfg->q[i];
    for (int i = 0; i < 2; i++) P.r[i] = cfg->r[i];
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t at = o; o += n; return at; };
    const int64_t nV = (int64_t)(N + 1) * ns, nU = 2 * Nc, nL = (int64_t)(N + 1) * 3, nE = (int64_t)(N + 1) * R;
    P.oV = take(nV); P.oU = take(nU); P.olam = take(nL); P.oeta = take(nE); P.oSL = take(nV); P.oZL = take(nV); P.oSU = take(nV); P.oZU = take(nV);
    P.oSLu = take(nU); P.oZLu = take(nU); P.oSUu = take(nU); P.oZUu = take(nU); P.odV = take(nV); P.odU = take(nU); P.olamn = take(nL); P.oetan = take(nE);
    P.oVt = take(nV); P.oUt = take(nU); P.osn = take(N); P.ocs = take(N); P.oHxx = take(4 * (int64_t)(N + 1)); P.ogx = take(nL); P.oWd = take(nE); P.ogdv = take(nE);
    P.ohuu = take(nU); P.ogu = take(nU); P.ohvt = take(N); P.oKg = take(6 * (int64_t)Nc); P.okff = take(nU); P.opo = take(2 * R);
    P.total = o;
    P.total = (P.total + 15) / 16 * 16;      // every instance starts on a 128-byte boundary
    P.S = max_batch;
    h->ws_bytes = (int64_t)sizeof(double) * P.total * P.S;
    if (hipMalloc((void **)&h->ws, (size_t)h->ws_bytes) != hipSuccess) { free(h); return NMPC_E_NOMEM; }
    if (hipMalloc((void **)&h->lb, sizeof(double) * nv) != hipSuccess || hipMalloc((void **)&h->ub, sizeof(double) * nv) != hipSuccess) {
        (void)hipFree(h->ws); if (h->lb) (void)hipFree(h->lb); free(h); return NMPC_E_NOMEM;
    }
    if (hipMemcpy(h->lb, lbx, sizeof(double) * nv, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(h->ub, ubx, sizeof(double) * nv, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(h->ws); (void)hipFree(h->lb); (void)hipFree(h->ub); free(h); return NMPC_E_HIP;
    }
    P.lb = h->lb; P.ub = h->ub;
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
    // the dynamic-LDS limit is an attribute of the kernel FUNCTION, not of a handle: set per launch when this handle needs more than HIP's
    // default of 64 KB (a second handle with another horizon would otherwise change the limit under this one)
    if (h->lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute((const void *)nmpc_lidar::lidar_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes) != hipSuccess) return NMPC_E_HIP;
    hipLaunchKernelGGL(nmpc_lidar::lidar_solve_kernel, dim3((unsigned)B), dim3(64), h->lds_bytes, (hipStream_t)stream, h->P, B, p, w0, w_out, obj, status, iters, kkt, h->ws);
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
