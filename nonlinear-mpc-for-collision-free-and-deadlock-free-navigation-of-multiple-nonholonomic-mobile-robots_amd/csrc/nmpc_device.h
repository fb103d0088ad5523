// nmpc_device.h — kernel parameter block shared by nmpc_kernels.hip and nmpc_api.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nmpc.h"

// shared with the CPU oracles: NMPC_SHIFT_ESCALATION, NMPC_COLD_RETRY_ITERS, NMPC_COLD_RETRIES, NMPC_X0_TOL
#include "../../include/nmpc_constants.h"

namespace nmpc {

// Passed by value to every kernel (lands in SGPRs / the kernarg segment).
struct KParams {
    int32_t m, N, K, thb, nxb, nh, n_ineq, nvar, ng, rows0, rowsk, pad_rows, max_iter;
    int32_t o_ul, o_uu, o_xl, o_xu, o_pr, o_ob;     // inequality slot offsets inside one stage block
    double T, dmin2, vmax, wmax, xymax, thmax, robdim, margin, pad_value, tol, mu_init;
    double q[3], r[2];
    double rho_el;        // penalty of the elastic phase (NMPC_ELASTIC_RHO)
    double obs[3 * NMPC_MAX_OBSTACLES];
    // per-instance workspace carve-up, in doubles
    int64_t stride;
    int32_t trace_inst;   // NMPC_PROFILE builds: instance whose per-iteration trace is recorded
    int32_t pairs;        // 1: pair rows present (C6:288-306); 0: multi-robot NLP without them (mpc_online_casadi_tb3_multi_centralized.py)
    const int32_t *order;       // per call: dispatch order (workgroup g solves instance order[g]) or nullptr = identity
    const int32_t *order_bad;   // device flag written by the permutation check of that order: non-zero -> the hint is ignored
    int64_t stride2;      // workspace stride of the LDS-resident kernel (stage packs + transposed gains)
    int64_t oPACK, oKT;
    int64_t oCKPT;        // LDS-resident kernels: cost-to-go saved every NMPC_CKPT_EVERY stages of the backward sweep, [(N-1)/NMPC_CKPT_EVERY + 1] slots of (3m + 1) * 64 doubles
    int64_t oELAS;        // LDS-resident kernels: elastic variables t of the pair and obstacle rows, [(N+1) * (NP + m K)]: touched only in the elastic phase
    int64_t oDUAL;        // column kernel: slacks and duals of the inequality rows (touched by the stage-parallel phases only) live here, not in LDS
    int32_t oX, oU, oLAM, oS, oZ, oDX, oDU, oLAMN, oDS, oDZ, oSN, oCS, oC, oH, oGX, oHUU, oGU, oHVT, oHTT, oKG, oKFF, oCKP, oEL;      // oCKP: the HBM-resident kernel's saved cost-to-go, [(N-1)/NMPC_CKPT_EVERY + 1][nx * nx + nx]; oEL: its elastic variables, one per inequality slot
};

hipError_t launch_solve(const KParams &P, int m, int B, const double *p, const double *w0, double *w_out, double *obj, int32_t *status,
                        int32_t *iters, double *kkt, double *ws, hipStream_t st);
hipError_t launch_solve_lds(const KParams &P, int m, int B, const double *p, const double *w0, double *w_out, double *obj, int32_t *status,
                            int32_t *iters, double *kkt, double *ws, long long *prof, hipStream_t st);
size_t lds_kernel_bytes(const KParams &P, int m);
hipError_t launch_solve_col(const KParams &P, int m, int B, const double *p, const double *w0, double *w_out, double *obj, int32_t *status,
                            int32_t *iters, double *kkt, double *ws, long long *prof, hipStream_t st, int shape);      // shape: 0 throughput, 1 latency (two wavefronts per instance)
size_t col_kernel_bytes(const KParams &P, int m, int shape);
void lds_kernel_workspace(const KParams &P, int m, int64_t *pack_off, int64_t *kt_off, int64_t *stride);
hipError_t launch_eval(const KParams &P, int m, int B, const double *p, const double *w, double *f, double *g, hipStream_t st);
hipError_t launch_shift(const KParams &P, int m, int B, const double *p, const double *w_in, double *w_next, double *x0n, int x0_stride, const int32_t *keep_status, hipStream_t st);      // x0_stride: doubles between the x0_next rows (0 = n_x); keep_status: instances with status 2 / 3 there are left untouched (or nullptr)
hipError_t launch_order_by_iters(int B, const int32_t *iters, int32_t *order, hipStream_t st);
hipError_t launch_odometry(long n, const double *odom, const double *init, double *pose, int wrap, hipStream_t st);
hipError_t launch_order_check(int B, const int32_t *order, int32_t *count, int32_t *bad, hipStream_t st);

}  // namespace nmpc
