// nmpc_solve_col.hip — column-per-lane, register-resident Riccati stage (gfx950).  The hot kernel of the throughput path.
//
// Same interior-point algorithm, same phases and the same stage packs / stage factors as nmpc_solve_lds.hip (see there and
// DESIGN.md 4); what differs is how one Riccati stage is executed.  There, the augmented symmetric matrix
//         [ Quu Qux | qu ]      = [B A]^T P [B A] + H
//         [ Qxu Qxx | qx ]
// is spread element-per-lane and every pivot step goes through LDS twice (publish the pivot row, gather two factors per
// element).  Here ONE LANE OWNS ONE COLUMN and keeps all of its rows in registers (m[r], r = 0..NZ-1, plus the right-hand side
// as row NZ), state column s on lane s, control column a on lane NX + a, so that a whole stage needs no LDS traffic for the
// matrix at all:
//   * P_k+1 is simply what the previous stage left in rows NU.. of the state lanes; P c_k is NX FMAs with broadcast defects;
//   * G = P [B A] is a lane gather of at most two foreign columns per lane (ds_bpermute, <= 3 terms per column) and
//     [B A]^T G is a ROW operation, i.e. a handful of in-lane FMAs with wave-uniform coefficients per robot;
//   * the Hessian additions arrive through per-lane LDS offsets into the staged pack (fixed per sweep);
//   * a pivot step is  m[a] -= M[j][a] * (m[j] / d_j)  for all remaining rows a: the multiplier M[j][a] sits in lane L(a) of
//     register m[j] (symmetry) and is broadcast by the DPP operand of the multiply-add itself (v_fmac_f64_dpp row_newbcast),
//     one DP instruction per row — no LDS round trip, no wave synchronisation, no scalar round trip;
//   * pivot rows stream to HBM/L2 already multiplied by -1/pivot (zero up to the diagonal): the forward sweep below loads them
//     row-per-lane and needs neither a division nor LDS staging.
// One wavefront per swarm instance for every team size (ten robots: 50 of 64 lanes, 51 rows = 102 VGPRs).
#include "nmpc_solve_common.h"

namespace nmpc {

__device__ __forceinline__ double lane_read(double v, int lane)      // uniform result (SGPR pair)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// Elimination multipliers.  NMPC_COL_DPP=1 (default): the multiplier of row a, lane LC(a) of the pivot row register, reaches
// every lane INSIDE the multiply-add: v_fmac_f64_dpp ... row_newbcast:n reads lane n of the executing lane's own row of 16
// lanes, so one DP instruction per row replaces v_readlane x2 + v_fma_f64.  Teams whose NZ columns span several rows of 16
// first replicate each row of the pivot register into the other rows (v_permlane16_swap / v_permlane32_swap, gfx950),
// 4 (two rows) or 12 (four rows) 32-bit VALU moves per pivot step.  NMPC_COL_DPP=0 keeps the readlane form (A/B).
// tools/dpp_probe.hip checks the lane semantics of both instructions on the device.
#ifndef NMPC_COL_GB
#define NMPC_COL_GB ((M_ == 6) ? (NX + 2) / 2 : NX + 1)      // rows per batch of the G = P [B A] gathers.  Measured: six robots in two batches +2.7 % (batch of 3 -> 166 k, 5 -> 171 k, 10 -> 172 k, 13 -> 173 k, all 19 rows at once -> 165 k solves/s); two robots +-0, ten robots -2.5 %: one batch there
#endif
#ifndef NMPC_FW_PD
#define NMPC_FW_PD ((M_ <= 6) ? 2 : 1)         // stages the forward sweep requests its rows ahead.  The rows come from L2 and the ring costs registers: with the spills of the earlier builds 1 was best (1 -> 198.5 k, 2 -> 194 k, 3 -> 191 k, 4 -> 176 k solves/s); on the spill-free build (A/B, one session) 2: six robots +1.3..2.5 %, B=16384 +1 %, two robots +2 %, 3: -22 % (the ring is spilled); ten robots 2: -1.6 %, stays 1
#endif
#ifndef NMPC_FW_FAKE
#define NMPC_FW_FAKE 0       // development: 1 = the forward sweep re-reads stage 0's rows (wrong results; separates compute from memory time)
#endif
#ifndef NMPC_COL_DPP
#define NMPC_COL_DPP 1
#endif
#ifndef NMPC_COL_WAVES_SMALL
#define NMPC_COL_WAVES_SMALL 1      // occupancy the compiler is ASKED for, up to three robots (see col_min_waves; two robots: 1 -> +1.3 % over 2, 3 = 168 VGPRs was slower)
#endif
#ifndef NMPC_COL_WAVES_MID
#define NMPC_COL_WAVES_MID 1        // occupancy the compiler is ASKED for, four to six robots (see col_min_waves)
#endif
#ifndef NMPC_COL_WAVES_FOUR
#define NMPC_COL_WAVES_FOUR 2       // four robots: asked for 1 the allocator lands on 257..267 registers (one wave per SIMD) since the elastic phase went in; asked for 2 it must stay within 256
#endif
#ifndef NMPC_COL_WAVES_LAT
#define NMPC_COL_WAVES_LAT 2        // five / six robots, latency shape, no heading bound: with the elastic phase's code the allocator, asked for 1, lands on 255 VGPRs + 2 AGPRs = 258 > 256, i.e. ONE wave per SIMD (warm closed loop 364 k -> 256 k solves/s); asked for 2 it must stay within 256
#endif
#ifndef NMPC_COL_DUMMY_ST
#define NMPC_COL_DUMMY_ST (NU - 1)
#endif
#ifndef NMPC_FUSE_DPHI
#define NMPC_FUSE_DPHI 1     // the barrier part of the merit function's directional derivative is summed by the step-length pass (same slots, same ds): one pass over the slack arrays less; A/B +0.7..1 %
#endif
#ifndef NMPC_COL_RP
#define NMPC_COL_RP 1        // row-paired backward sweep for two to six robots (see the sweep); 0 keeps one row per register (A/B)
#endif

// lane that owns column c of the augmented matrix: state column NU + s on lane s, control column a on lane NX + a
#define LC(c) (((c) < NU) ? NX + (c) : (c) - NU)

// row-paired layout: register slot / half of control row j (natural order v_0, omega_0, v_1, ...), and whether a control register
// still holds a row that has not been a pivot after pivot jj
constexpr int rp_slot(int j) { return 2 * ((j >> 1) >> 1) + (j & 1); }
constexpr int rp_half(int j) { return (j >> 1) & 1; }
constexpr bool rp_live(int i, int jj, int nc, int nu) { return i >= nc || (4 * (i >> 1) + (i & 1) > jj) || (4 * (i >> 1) + 2 + (i & 1) > jj && 4 * (i >> 1) + 2 + (i & 1) < nu); }

// Second argument of __launch_bounds__.  Two waves per SIMD (8 instances per CU) is what runs for up to six robots either way: asked for 1, the
// allocator lands on 230..254 registers WITHOUT a spill and the hardware still places two waves; asked for 2 it stops at 256 and spills 56..64
// dwords to scratch (reloaded at the phase boundaries of every iteration).  Measured A/B in one session, six robots: B = 2048 +3.5 %, 4096
// +2.5 %, 16384 +3 %.  The one variant that lands above 256 when asked for 1 (four robots, heading bounds, slacks in the workspace: 257) keeps 2.
constexpr int col_min_waves(int m, int thb, int dl)
{
    return m > 6 ? 1 : (m <= 3 ? NMPC_COL_WAVES_SMALL : (m == 4 ? NMPC_COL_WAVES_FOUR : ((m >= 5 && !thb && dl) ? NMPC_COL_WAVES_LAT : NMPC_COL_WAVES_MID)));
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int N_> __device__ __forceinline__ void fmac_rowb(double &acc, double u, double nr)      // acc += u[lane N_ of my row of 16] * nr
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(u), "v"(nr), "n"(N_));
}
__device__ __forceinline__ void swap16(double &a, double &b)      // odd rows (of 16 lanes) of a <-> even rows of b
{
    u32x2 lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    u32x2 hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    a = __hiloint2double(hi.x, lo.x); b = __hiloint2double(hi.y, lo.y);
}
__device__ __forceinline__ void swap32(double &a, double &b)      // upper 32 lanes of a <-> lower 32 lanes of b
{
    u32x2 lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    u32x2 hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    a = __hiloint2double(hi.x, lo.x); b = __hiloint2double(hi.y, lo.y);
}

__device__ __forceinline__ double lane_gather(int src4, double v)     // v of lane src4 / 4
{
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(src4, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(src4, __double2loint(v)));
}

#ifndef NMPC_COL_SADDR
#define NMPC_COL_SADDR 1
#endif
struct UArr {      // array of the instance's workspace (or of LDS): wave-uniform base, uniform element offset, 32-bit element index (see the kernel)
    double *b;
    int o;
    __device__ __forceinline__ double &operator[](int i) const
    {
#if NMPC_COL_SADDR
        return *reinterpret_cast<double *>(reinterpret_cast<char *>(b) + (size_t)((uint32_t)(o + i) * 8u));
#else
        return b[o + i];
#endif
    }
};

// Stage pack of this kernel: G2's layout with a COMPACT coefficient table.  G2 keeps three coefficients for each of the NZ columns of [B A]
// (3 NZ doubles, most of them 0, 1 or T); here PK_CF holds the four entries per robot that depend on the iterate — T cos, T sin, -T v sin,
// T v cos — followed by one slot with T and the zero slot: 168 instead of 232 doubles for six robots, i.e. three 64-lane slices to stage per
// stage instead of four and 26 KB less traffic per iteration.  (The workspace is sized for G2's pack; the element-per-lane kernel keeps G2.)
template <int M_, int THB> struct GC : G2<M_, THB> {
    using B = G2<M_, THB>;
    static constexpr int PK_CF = B::PK_CF;           // [4 M]
    static constexpr int PK_T = PK_CF + 4 * M_;      // T
    static constexpr int PK_ZERO = PK_T + 1;         // one zero entry (target of "no Hessian addition" / "no coefficient")
    static constexpr int PACKC = ((PK_ZERO + 1 + 7) / 8) * 8;
    static constexpr int PACK = (PACKC < 64 && B::PACK >= 64) ? 64 : PACKC;      // two robots: one whole 64-lane slice, as before (no exec-masked staging)
    static_assert(PACK <= B::PACK, "the workspace is sized for G2's pack");
};

// TPBK: threads per instance.  64 = the throughput shape (one wavefront per instance).  128 (256) = the LATENCY shape: a second wavefront (three more) shares
// the stage-parallel phases (evaluation, optimality error, stage packs, step lengths, multipliers, merit function, update — 41 % of a lone
// wave's iteration for six robots) and waits at a barrier while wave 0 runs the two sweeps; for batches whose launch is as long as their
// longest solve.
template <int M_, int THB, int DL, int TPBK = 64>
__global__ __launch_bounds__(TPBK, col_min_waves(M_, THB, DL)) void solve_col_kernel(const KParams P, const double *__restrict__ p_in, const double *__restrict__ w0,
                                                        double *__restrict__ w_out, double *__restrict__ obj_out,
                                                        int32_t *__restrict__ status_out, int32_t *__restrict__ iters_out,
                                                        double *__restrict__ kkt_out, double *__restrict__ ws, long long *__restrict__ prof_out)
{
    using G = GC<M_, THB>;
    constexpr int TPB = TPBK;
    static_assert(TPBK == 64 || TPBK == 128 || TPBK == 256, "one, two or four wavefronts per instance");
    constexpr int NX = G::NX, NU = G::NU, NP = G::NP, NZ = G::NZ, LD = G::LD, NXB = G::NXB;
    constexpr int NPd = NP > 0 ? NP : 1;   // divisor that stays legal for M_ == 1 (those loops have zero trips)
    constexpr int MPN = M_ > 1 ? M_ - 1 : 1;   // partners of a robot in the pair rows
    static_assert(NZ <= 64, "one lane per column of the augmented matrix");
    constexpr int NR = (NZ + 15) / 16;     // rows of 16 lanes that hold columns
    // row-paired backward sweep (two to six robots): MP robots per half, NC control + NS state register slots, slot RR = right-hand side
    constexpr bool RP = NMPC_COL_RP && M_ >= 2 && M_ <= 6;
    constexpr int MP = (M_ + 1) / 2, NC = 2 * MP, NS = 3 * MP, RR = NC + NS;
    const int tid = threadIdx.x;
    const int N = P.N, N1 = P.N + 1, K = P.K, MK = M_ * P.K;
    const double T = P.T;
    const size_t inst = (P.order && *P.order_bad == 0) ? (size_t)P.order[blockIdx.x] : (size_t)blockIdx.x;      // dispatch-order hint: long solves first (ignored unless it is a permutation)
    double mu = P.mu_init;                         // barrier parameter (declared here: the merit function of the elastic phase needs it)
    bool el = false;                               // elastic phase (the second restart of last resort): pair / obstacle rows h + t - s = 0 with penalty rho t
    const bool prs = P.pairs != 0;                 // pair rows present (the no-pair multi-robot NLP keeps NP slots that are never touched)
    const int NPA = prs ? G::NP : 0;

    // LDS holds what the serial sweeps touch (states, controls, multipliers, trig cache, step) plus one scratch region; the
    // slacks and duals of the inequality rows are only ever visited by the stage-parallel phases (coalesced, each lane its own
    // items) and live in the instance's workspace in HBM/L2 — that is what lets two instances share a SIMD (<= 20 KB of LDS each)
    extern __shared__ double sm[];
    double *X = sm;                       // [N1*NX]
    double *U = X + N1 * NX;              // [N*NU]
    double *LAM = U + N * NU;             // [N1*NX]   lam[k] pairs with defect c_{k-1}
    double *SN = LAM + N1 * NX;           // [N*M]
    double *CS = SN + N * M_;             // [N*M]
    double *DX = CS + N * M_;             // [N1*NX]   step
    double *DU = DX + N1 * NX;            // [N*NU]
    double *RV = DU + N * NU;             // one region, two lives: PK [PACK] the staged pack of the backward sweep; RV [N1*NX] the residuals /
    double *FS = RV, *PK = RV;            // QP multipliers of the adjoint recursion (FS [KTS]: staged stage factor, only the > 32-control forward path)
    constexpr int RG0 = G::KTS > G::PACK ? G::KTS : G::PACK;
    double *XS = RV + ((N1 * NX > RG0) ? N1 * NX : RG0);   // [NX]
    double *RED = XS + NX;                // [8]
    // slacks and duals of the inequality rows: in the instance's workspace (HBM/L2) for the larger teams, whose LDS budget of
    // 20 KB (eight instances per CU) they would break; in LDS (DL = 1) for up to four robots when they fit that budget — there the
    // stage-parallel phases are most of an iteration and their loops make one trip, i.e. one exposed memory latency each
    // The arrays are (wave-uniform base, uniform offset) pairs indexed by a 32-bit element index: the byte offset is formed in 32 bits and zero-extended,
    // which the compiler addresses as scalar base + 32-bit vector offset (`global_load v, v_off, s[base:base+1]`) instead of a 64-bit vector address
    // per access (two registers and two or three address instructions each).  NMPC_COL_SADDR = 0 keeps plain pointers (A/B).
    double *gd = DL ? (RED + 8) : (ws + inst * P.stride2 + P.oDUAL);
    const int oZPp = N1 * NP, oSO = oZPp + N1 * NP, oZO = oSO + N1 * MK, oZUL = oZO + N1 * MK, oZUU = oZUL + N * NU, oZXL = oZUU + N * NU, oZXU = oZXL + N1 * NXB,
              oSUL = oZXU + N1 * NXB, oSUU = oSUL + N * NU;
    const UArr SPp{gd, 0};                // [N1*NP]   pair slacks
    const UArr ZPp{gd, oZPp};             // [N1*NP]   pair duals
    const UArr SO{gd, oSO};               // [N1*MK]   obstacle slacks
    const UArr ZO{gd, oZO};               // [N1*MK]
    const UArr ZUL{gd, oZUL};             // [N*NU]
    const UArr ZUU{gd, oZUU};             // [N*NU]
    const UArr ZXL{gd, oZXL};             // [N1*NXB]
    const UArr ZXU{gd, oZXU};             // [N1*NXB]
    const UArr SUL{gd, oSUL};             // [N*NU]    explicit slacks of the simple bounds
    const UArr SUU{gd, oSUU};             // [N*NU]

    // DL >= 2 (small teams, throughput shape): the stage factors — written by the backward sweep, read by the forward sweep — live in LDS as well, DL >= 3:
    // the stage packs too; otherwise both are in the instance's workspace (HBM/L2)
    double *const dl_end = DL ? gd + oSUU + N * NU : RED + 8;
    double *gkt = (DL >= 2) ? dl_end : ws + inst * P.stride2 + P.oKT;       // [N][KTS]
    double *gpack = (DL >= 3) ? dl_end + (size_t)N * G::KTS : ws + inst * P.stride2 + P.oPACK;   // [N][PACK] + terminal [2*NX]
    const UArr TPp{ws + inst * P.stride2 + P.oELAS, 0};           // [N1*NP]   elastic variables of the pair rows (elastic phase only; always in the workspace)
    const UArr TOb{ws + inst * P.stride2 + P.oELAS, N1 * NP};     // [N1*MK]   ... of the obstacle rows
    const double rho = P.rho_el;
    double *gck = ws + inst * P.stride2 + P.oCKPT;     // [(N-1)/NMPC_CKPT_EVERY + 1][NX + 1][64]  saved cost-to-go of the backward sweep (column-per-lane registers as they are)
    const double *pp = p_in + inst * (2 * NX);
    const double *wi = w0 + inst * (size_t)P.nvar;
    double *wo = w_out + inst * (size_t)P.nvar;

#ifdef NMPC_POISON
    {   // debug build: every LDS word and the instance's HBM workspace start as NMPC_POISON, so that a read of anything this solve did not
        // write shows up as a parity failure instead of depending on what ran on the CU before
        const int nl = (int)((DL ? gd + oSUU + N * NU : RED + 8) - sm) + (DL >= 2 ? N * G::KTS : 0) + (DL >= 3 ? N1 * G::PACK : 0);
        for (int e = tid; e < nl; e += TPB) sm[e] = NMPC_POISON;
        for (size_t e = tid; e < (size_t)P.stride2; e += TPB) ws[inst * P.stride2 + e] = NMPC_POISON;
        __syncthreads();
    }
#endif
#ifdef NMPC_PROFILE
    long long prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = clock64();
#endif

    auto lbu = [&](int c) { return (c & 1) ? -P.wmax : -P.vmax; };
    auto bst = [&](int s) { return THB ? s : 3 * (s >> 1) + (s & 1); };            // bounded-state slot -> state index
    auto bvl = [&](int s) { return (THB && (s % 3 == 2)) ? P.thmax : P.xymax; };

    // ---- column ownership: lane -> column of the augmented matrix (controls 0..NU-1, states NU..NZ-1)
    const bool is_state = tid < NX, is_ctrl = tid >= NX && tid < NZ, lvalid = tid < NZ;
    const int ms = is_state ? tid : 0;                       // my state index
    const int ma = is_ctrl ? tid - NX : 0;                   // my control index
    const int mycol = is_state ? NU + ms : ma;               // 0 on unused lanes (they carry zero coefficients)
    const int mrob = is_state ? ms / 3 : ma >> 1;
    // gather sources of G = P [B A]: at most two foreign columns per lane
    //   x_i, y_i: none;  theta_i: x_i, y_i;  v_i: x_i, y_i;  omega_i: theta_i
    const bool th_lane = is_state && (ms - 3 * mrob == 2);
    const int srcA = 4 * (th_lane ? 3 * mrob : (is_ctrl ? ((ma & 1) ? 3 * mrob + 2 : 3 * mrob) : tid));
    const int srcB = 4 * (th_lane ? 3 * mrob + 1 : (is_ctrl ? ((ma & 1) ? 3 * mrob + 2 : 3 * mrob + 1) : tid));

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
            int i, j;
            pair_of<M_>(q, i, j);
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
    auto pair_ij = [&](int q, int &i, int &j) { pair_of<M_>(q, i, j); };      // loop-free (nmpc_solve_common.h): the while-loop form was a divergent loop, ~60 issue slots per call

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
    auto merit_t = [&](auto elc, double a, double &fv, double &lg, double &th, double &ec_, double &eh_) {
        constexpr bool EL = decltype(elc)::value;      // elastic phase: a separate instantiation, the plain one carries no trace of it
        double fs = 0.0, t = 0.0, mc = 0.0, mh = 0.0;
        LogSum ls;      // sum of log(slack): mantissa product and exponent sum per lane, ONE logarithm per wavefront at the end (nmpc_solve_common.h)
        // one bound row: current slack sv, current value h0, step of the value jd, trial value ht
        // (the slacks of one (stage, robot) item enter the barrier as ONE logarithm of their product — 8 to 10 factors between 1e-12
        // and 2 xy_max, far from over/underflow: fp64 log is ~80 instructions, the item would otherwise spend most of its time there)
        double pr = 1.0;
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
            ls.add(pr);
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
            if constexpr (EL) {       // elastic row: s and t both move, the penalty rho t joins the objective
                tv = TPp[k * NP + q];
                if (a != 0.0) {
                    double ex = X[oi] - X[oj], ey = X[oi + 1] - X[oj + 1], ds, dz, dt;
                    el_step(mu, rho, sv, ZPp[k * NP + q], tv, h_pair(ex, ey, P.dmin2), jd_pair(ex, ey, DX[oi] - DX[oj], DX[oi + 1] - DX[oj + 1]), ds, dz, dt);
                    sv += a * ds; tv += a * dt;
                }
                ls.add(tv); fs += rho * tv;
            } else if (a != 0.0) {
                double ex = X[oi] - X[oj], ey = X[oi + 1] - X[oj + 1];
                sv += a * ds_pair(ex, ey, DX[oi] - DX[oj], DX[oi + 1] - DX[oj + 1], P.dmin2, sv);
            }
            ls.add(sv);
            double r = fabs(h - sv + tv);
            t += r; mh = fmax(mh, r);
        }
        for (int it = tid; it < (N - 1) * MK; it += TPB) {
            int k = 1 + it / MK, e = it - (k - 1) * MK, i = e / K, o = e - i * K;
            const int oi = k * NX + 3 * i;
            double px = X[oi] + a * DX[oi], py = X[oi + 1] + a * DX[oi + 1];
            double h = h_obs(r_obs(px - P.obs[3 * o], py - P.obs[3 * o + 1]), P.robdim, P.obs[3 * o + 2], P.margin);
            double sv = SO[k * MK + e], tv = 0.0;
            if constexpr (EL) {
                tv = TOb[k * MK + e];
                if (a != 0.0) {
                    double ex = X[oi] - P.obs[3 * o], ey = X[oi + 1] - P.obs[3 * o + 1], rr = r_obs(ex, ey), ds, dz, dt;
                    el_step(mu, rho, sv, ZO[k * MK + e], tv, h_obs(rr, P.robdim, P.obs[3 * o + 2], P.margin), jd_obs(ex, ey, rr, DX[oi], DX[oi + 1]), ds, dz, dt);
                    sv += a * ds; tv += a * dt;
                }
                ls.add(tv); fs += rho * tv;
            } else if (a != 0.0) {
                double ex = X[oi] - P.obs[3 * o], ey = X[oi + 1] - P.obs[3 * o + 1], rr = r_obs(ex, ey);
                sv += a * ds_obs(ex, ey, rr, DX[oi], DX[oi + 1], h_obs(rr, P.robdim, P.obs[3 * o + 2], P.margin), sv);
            }
            ls.add(sv);
            double r = fabs(h - sv + tv);
            t += r; mh = fmax(mh, r);
        }
        fv = wsum<TPB>(fs, RED); lg = wlogsum<TPB>(ls, RED); th = wsum<TPB>(t, RED);
        ec_ = wmax<TPB>(mc, RED); eh_ = wmax<TPB>(mh, RED);
    };
    auto merit = [&](double a, double &fv, double &lg, double &th, double &ec_, double &eh_) {
        if (el) merit_t(std::true_type{}, a, fv, lg, th, ec_, eh_); else merit_t(std::false_type{}, a, fv, lg, th, ec_, eh_);
    };

    // ---- constant entries of the stage packs (the T and zero slots, stage 0's pair blocks): once per solve
    for (int it = tid; it < N * M_; it += TPB) {
        int k = it / M_, i = it - k * M_;
        if (i == 0) { gpack[(size_t)k * G::PACK + G::PK_T] = T; gpack[(size_t)k * G::PACK + G::PK_ZERO] = 0.0; }
    }
    for (int q = tid; q < 3 * NP; q += TPB) gpack[G::PK_E + q] = 0.0;     // stage 0 carries no pair rows
    if (!prs)                                                              // no pair rows at all: the E slots of every stage are zero
        for (int e = tid; e < (N - 1) * 3 * NP; e += TPB) gpack[(size_t)(1 + e / (3 * NPd)) * G::PACK + G::PK_E + e % (3 * NPd)] = 0.0;
    // ---- slacks and duals from the current primal point (also the barrier restart after a stall)
    auto init_barrier = [&]() {
        for (int it = tid; it < N1 * NP; it += TPB) {
            int k = it / NPd, q = it - k * NP;
            if (k >= 1 && k <= N - 1 && prs) {
                int i, j; pair_ij(q, i, j);
                double dx = X[k * NX + 3 * i] - X[k * NX + 3 * j], dy = X[k * NX + 3 * i + 1] - X[k * NX + 3 * j + 1];
                const double hv = h_pair(dx, dy, P.dmin2);
                if (el) {       // t absorbs the violation: s = h + t >= bp, 0 < z < rho
                    const double tv = fmax(bp, bp - hv), sv = hv + tv;
                    TPp[it] = tv; SPp[it] = sv; ZPp[it] = fmin(mu / sv, 0.5 * rho);
                } else {
                double sv = fmax(hv, bp);
                SPp[it] = sv; ZPp[it] = mu / sv;
                }
            } else { SPp[it] = 1.0; ZPp[it] = 0.0; }
        }
        for (int it = tid; it < N1 * MK; it += TPB) {
            int k = it / MK, e = it - k * MK;
            if (k >= 1 && k <= N - 1) {
                int i = e / K, o = e - i * K;
                double dx = X[k * NX + 3 * i] - P.obs[3 * o], dy = X[k * NX + 3 * i + 1] - P.obs[3 * o + 1];
                const double hv = h_obs(r_obs(dx, dy), P.robdim, P.obs[3 * o + 2], P.margin);
                if (el) {
                    const double tv = fmax(bp, bp - hv), sv = hv + tv;
                    TOb[it] = tv; SO[it] = sv; ZO[it] = fmin(mu / sv, 0.5 * rho);
                } else {
                double sv = fmax(hv, bp);
                SO[it] = sv; ZO[it] = mu / sv;
                }
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
                if (prs) {       // partners in ascending order (p-th partner: j = p, or p + 1 from the own index on); their duals requested first
                    double zq[MPN];
                    static_for<0, M_ - 1>([&](auto pc) { constexpr int p = decltype(pc)::value; zq[p] = ZPp[kk * NP + pidx_any<M_>(i, p + (p >= i ? 1 : 0))]; });
                    static_for<0, M_ - 1>([&](auto pc) {
                        constexpr int p = decltype(pc)::value;
                        const int j = p + (p >= i ? 1 : 0);
                        j0 += 2 * (xi - x[3 * j]) * zq[p]; j1 += 2 * (yi - x[3 * j + 1]) * zq[p];
                    });
                }
                for (int o = 0; o < K; o++) {
                    double dx = xi - P.obs[3 * o], dy = yi - P.obs[3 * o + 1], rr = r_obs(dx, dy), z = ZO[kk * MK + i * K + o];
                    j0 += qdiv(dx, rr) * z; j1 += qdiv(dy, rr) * z;
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
        }
        for (int it = tid; it < (N - 1) * MK; it += TPB) {
            double zv = ZO[MK + it], pz = SO[MK + it] * zv;
            zsum += zv; szmax = fmax(szmax, pz); szmin = fmin(szmin, pz);
        }
        if (el) {       // elastic phase: the products t (rho - z) of the elastic variables with their duals
            for (int it = tid; it < (N - 1) * NPA; it += TPB) { const double pt = TPp[NP + it] * (rho - ZPp[NP + it]); szmax = fmax(szmax, pt); szmin = fmin(szmin, pt); }
            for (int it = tid; it < (N - 1) * MK; it += TPB) { const double pt = TOb[MK + it] * (rho - ZO[MK + it]); szmax = fmax(szmax, pt); szmin = fmin(szmin, pt); }
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
        auto cold_retry = [&]() { cold = true; n_cold++; it_base = iter; restarting = true; el = n_cold >= 2; mu = P.mu_init; n_tiny = 0; n_restart = 0; };      // the second restart of last resort is the elastic phase (include/nmpc_constants.h)
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
        auto stage_packs = [&](auto elc) {
        constexpr bool EL = decltype(elc)::value;      // elastic phase: its own instantiation (the plain one is the code of rounds 1-3)
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
                    auto bv = [&](double sv, double zv, double hv, double &hd) { double sg = qdiv(zv, sv); hd += sg; return qdiv(mu, sv) - sg * (hv - sv); };
                    g0 -= bv(x[3 * i] + P.xymax, ZXL[sb], x[3 * i] + P.xymax, h0) - bv(P.xymax - x[3 * i], ZXU[sb], P.xymax - x[3 * i], h0);
                    g1 -= bv(x[3 * i + 1] + P.xymax, ZXL[sb + 1], x[3 * i + 1] + P.xymax, h1) - bv(P.xymax - x[3 * i + 1], ZXU[sb + 1], P.xymax - x[3 * i + 1], h1);
                    if (THB) g2 -= bv(x[3 * i + 2] + P.thmax, ZXL[sb + 2], x[3 * i + 2] + P.thmax, h2) - bv(P.thmax - x[3 * i + 2], ZXU[sb + 2], P.thmax - x[3 * i + 2], h2);
                }
                if (k <= N - 1) {
                    const double xi = x[3 * i], yi = x[3 * i + 1];
                    if (prs) {
                        double sq[MPN], zq[MPN];
                        static_for<0, M_ - 1>([&](auto pc) {
                            constexpr int p = decltype(pc)::value;
                            const int q = pidx_any<M_>(i, p + (p >= i ? 1 : 0));
                            sq[p] = SPp[k * NP + q]; zq[p] = ZPp[k * NP + q];
                        });
                        static_for<0, M_ - 1>([&](auto pc) {
                            constexpr int p = decltype(pc)::value;
                            const int j = p + (p >= i ? 1 : 0);
                            double dx = xi - x[3 * j], dy = yi - x[3 * j + 1];
                            double sv = sq[p], zv = zq[p], sg, v;
                            if constexpr (EL) el_sigma(mu, rho, sv, zv, TPp[k * NP + pidx_any<M_>(i, j)], h_pair(dx, dy, P.dmin2), sg, v);
                            else { sg = qdiv(zv, sv); v = qdiv(mu, sv) - sg * (h_pair(dx, dy, P.dmin2) - sv); }
                            g0 -= 2 * dx * v; g1 -= 2 * dy * v;
                            h0 += 4 * sg * dx * dx - 2 * zv; hxy += 4 * sg * dx * dy; h1 += 4 * sg * dy * dy - 2 * zv;
                        });
                    }
                    for (int o = 0; o < K; o++) {
                        double dx = xi - P.obs[3 * o], dy = yi - P.obs[3 * o + 1], rr = r_obs(dx, dy), n0 = qdiv(dx, rr), n1 = qdiv(dy, rr);
                        double sv = SO[k * MK + i * K + o], zv = ZO[k * MK + i * K + o], sg, v, zz = qdiv(zv, rr);
                        if constexpr (EL) el_sigma(mu, rho, sv, zv, TOb[k * MK + i * K + o], h_obs(rr, P.robdim, P.obs[3 * o + 2], P.margin), sg, v);
                        else { sg = qdiv(zv, sv); v = qdiv(mu, sv) - sg * (h_obs(rr, P.robdim, P.obs[3 * o + 2], P.margin) - sv); }
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
                    double vl = qdiv(mu, sl) - qdiv(zl, sl) * ((u[d] - lo) - sl), vu = qdiv(mu, su) - qdiv(zu, su) * ((-lo - u[d]) - su);
                    pk[G::PK_G + 2 * i + d] = 2 * P.r[d] * u[d] - (vl - vu);
                    pk[G::PK_HD + 2 * i + d] = 2 * P.r[d] + qdiv(zl, sl) + qdiv(zu, su);
                }
                {   // coefficients of [B A] belonging to robot i that depend on the iterate: (v_i: x, y) = T cos, T sin; (theta_i: x, y) = -T v sin, T v cos
                    double *cf = pk + G::PK_CF + 4 * i;
                    cf[0] = T * c; cf[1] = T * s; cf[2] = -T * u[0] * s; cf[3] = T * u[0] * c;
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
            if constexpr (EL) { double v_; el_sigma(mu, rho, SPp[k * NP + q], zz, TPp[k * NP + q], h_pair(dx, dy, P.dmin2), sg, v_); }
            else sg = qdiv(zz, SPp[k * NP + q]);
            double *pk = gpack + (size_t)k * G::PACK + G::PK_E + 3 * q;
            pk[0] = -(4 * sg * dx * dx - 2 * zz); pk[1] = -(4 * sg * dx * dy); pk[2] = -(4 * sg * dy * dy - 2 * zz);
        }
        };
        if (el) stage_packs(std::true_type{}); else stage_packs(std::false_type{});
        __syncthreads();
        PROF_T(2);

        // ============ B. Riccati sweep with inertia correction (IPOPT alg. IC), column-per-lane
        // first trial: delta = 0, except right after an iteration that needed a shift (then a quarter of that shift directly).
        // Partial re-factorisation (round 4, include/nmpc_constants.h): the cost-to-go entering every NMPC_CKPT_EVERY-th stage is saved in the
        // workspace; a rejected pivot at stage kf escalates the shift and resumes the sweep at the saved stage NMPC_RESUME_STAGE(kf, N) — the
        // stages above keep their factorisation (the pivot rows already stored) and the smaller shift it was made with.
        double delta = need_shift ? fmax(1e-20, 0.25 * delta_last) : 0.0;
        int ntry = 0;
        bool ok;
        int kst = N - 1;        // stage this pass of the sweep starts at
        for (;;) {
            ok = true;
            int kf = 0;         // stage of the rejected pivot
            if (TPBK == 64 || tid < 64) {          // the sweep is wave 0's (the second wave of the latency shape waits at the barrier below)
            constexpr int TPB = 64;
            if constexpr (RP) {
            // ==== row-paired sweep (up to six robots): BOTH 32-lane halves of the wave hold every column, and the rows are split between
            // them by robot parity — the lower half keeps the rows of robots 0, 2, 4 (and the right-hand side), the upper half those of
            // robots 1, 3, 5.  Register m[i] therefore carries TWO rows, (half 0, slot i) and (half 1, slot i); slots: 2p + d = control d of
            // the half's p-th robot, NC + 3p + d = its state d, RR = right-hand side.  The columns mirror that inside each half: wave-row
            // (16 lanes) w holds the columns of the robots of parity w, at position = slot.  A pivot step needs, on the lanes of half h, the
            // multipliers M[j][col(h, slot)]: position `slot` of wave-row h of the pivot row — so with ur = [P0 P0 P1 P1] (Pw = wave-row w of
            // the pivot row's half, v_permlane16/32_swap) ONE v_fmac_f64_dpp row_newbcast:slot eliminates the two rows of a register:
            // 159 instead of 282 DP instructions for six robots, 16 instead of 31 matrix registers, and half as many rows in the
            // gathers of P [B A], the row operations of [B A]^T G and the Hessian additions.  Pivot order, the stored pivot rows and the
            // cost-to-go left for the next stage are those of the single-row layout.
            const int rh = tid >> 5, rw = (tid >> 4) & 1, rn = tid & 15;
            const bool cc_ = rn < NC;                                          // my column is a control
            const int cp = cc_ ? (rn >> 1) : (rn - NC) / 3, cd = cc_ ? (rn & 1) : (rn - NC) - 3 * cp, crob = 2 * cp + rw;
            const bool cvalid = rn < RR && crob < M_, c_state = cvalid && !cc_, c_ctrl = cvalid && cc_;
            const int rcol = cvalid ? (cc_ ? 2 * crob + cd : NU + 3 * crob + cd) : 0;      // natural column index
            auto rob_ok = [&](int p) { return (M_ % 2 == 0) || p < MP - 1 || rh == 0; };  // the upper half's last robot slot is empty for odd teams
            int hoff[RR];
            static_for<0, RR>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr bool au = i < NC;
                constexpr int p = au ? (i >> 1) : (i - NC) / 3, d = au ? (i & 1) : (i - NC) - 3 * p;
                const int rob = 2 * p + rh;                                    // robot of my half's row in this register
                const int a = au ? 2 * rob + d : NU + 3 * rob + d;             // its natural row index
                int h = G::PK_ZERO;
                if constexpr (!au && d < 2) {        // x / y row: pair blocks (negated in the pack) and the xy term of the own robot
                    const int lo = rob < crob ? rob : crob, hi = rob < crob ? crob : rob;
                    const int pe = G::PK_E + 3 * (lo * (2 * M_ - lo - 1) / 2 + (hi - lo - 1)) + d + cd;
                    h = (c_state && cd < 2) ? ((rob == crob) ? G::PK_HXY + rob : pe) : h;
                }
                if constexpr (au && d == 0) h = (c_state && cd == 2 && crob == rob) ? G::PK_HVT + rob : h;       // (v_i, theta_i)
                if constexpr (!au && d == 2) h = (c_ctrl && cd == 0 && crob == rob) ? G::PK_HVT + rob : h;       // (theta_i, v_i)
                h = (rcol == a) ? G::PK_HD + a : h;
                hoff[i] = 8 * ((cvalid && rob < M_) ? h : G::PK_ZERO);
            });
            const int goff = 8 * (cvalid ? G::PK_G + rcol : G::PK_ZERO);
            // coefficients of the two gathered terms of my column of [B A]: theta_i <- (-T v sin, T v cos), v_i <- (T cos, T sin), omega_i <- (T, 0), else none
            const int cfA = 8 * ((c_state && cd == 2) ? G::PK_CF + 4 * crob + 2 : ((c_ctrl && cd == 0) ? G::PK_CF + 4 * crob : ((c_ctrl && cd == 1) ? G::PK_T : G::PK_ZERO)));
            const int cfB = 8 * ((c_state && cd == 2) ? G::PK_CF + 4 * crob + 3 : ((c_ctrl && cd == 0) ? G::PK_CF + 4 * crob + 1 : G::PK_ZERO));
            const int cbo = 8 * (G::PK_C + 3 * rh), cf1 = 8 * (G::PK_CF + 4 * rh);
            // gather sources of G = P [B A] inside my wave-row:  theta_i, v_i <- x_i, y_i;  omega_i <- theta_i
            const int lb = tid & ~15;
            const bool needxy = (c_state && cd == 2) || (c_ctrl && cd == 0), needth = c_ctrl && cd == 1;
            const int srcA2 = 4 * (needxy ? lb + NC + 3 * cp : (needth ? lb + NC + 3 * cp + 2 : tid));
            const int srcB2 = 4 * (needxy ? lb + NC + 3 * cp + 1 : (needth ? lb + NC + 3 * cp + 2 : tid));
            // ---- terminal cost-to-go P_N = diag(hd_N), p_N = g_N — or, resuming after a rejected pivot, the cost-to-go saved on entry to stage kst
            double m[RR + 1];
            static_for<0, NC>([&](auto rc) { m[decltype(rc)::value] = 0.0; });
            if (kst == N - 1) {
                const double *pkN = gpack + (size_t)N * G::PACK;
                const double hdN = c_state ? pkN[G::PK_HD + rcol] : 0.0, gN = c_state ? pkN[G::PK_G + rcol] : 0.0;
                static_for<NC, RR>([&](auto rc) { constexpr int r = decltype(rc)::value; m[r] = (c_state && rw == rh && rn == r) ? hdN : 0.0; });
                m[RR] = gN;
            } else {
                const double *ck = gck + (size_t)((N - 1 - kst) / NMPC_CKPT_EVERY) * ((NX + 1) * 64) + tid;
                static_for<0, NS + 1>([&](auto rc) { constexpr int r = decltype(rc)::value; const double v = ck[r * 64]; m[NC + r] = cvalid ? v : 0.0; });
            }
            constexpr int PKR = (G::PACK + TPB - 1) / TPB;
            constexpr bool FULL = PKR * TPB <= RG0 && PKR * TPB <= 2 * G::PACK;
            double pkr[PKR];
#pragma unroll
            for (int t = 0; t < PKR; t++) { int e = tid + t * TPB; pkr[t] = (FULL || e < G::PACK) ? gpack[(size_t)kst * G::PACK + e] : 0.0; }
            double pend_row = 0.0, pend_rhs = 0.0;
            int ckc = 0;        // stages since the last saved one (the pass starts at a saved stage, or at N-1)
#if NMPC_COL_DUMMY_ST
            // vmcnt counts loads and stores in issue order.  The wait for a stage's pack (requested at the top of the previous stage) may leave
            // the NU - 1 pivot-row stores issued after that request in flight — but the compiler merges the counter state of the loop's two
            // entries, and on the path from here nothing follows the first request, so it waited for vmcnt(3..0): every stage began with the
            // acknowledgement latency of the previous stage's last stores.  As many (padding-slot) stores here make the two paths alike.
            static_for<0, NMPC_COL_DUMMY_ST>([&](auto jc) { gkt[(size_t)(N - 1) * G::KTS + decltype(jc)::value * G::LDC + NZ + 1] = 0.0; });
#endif
            for (int k = kst; k >= 0; k--) {
                lds_sync<TPB>();
#pragma unroll
                for (int t = 0; t < PKR; t++) {
                    const int e = tid + t * TPB;
                    if (FULL || e < G::PACK) PK[e] = (e >= G::PK_HD && e < G::PK_HD + NU) ? pkr[t] + delta : pkr[t];
                }
                if (k > 0) {
#pragma unroll
                    for (int t = 0; t < PKR; t++) { int e = tid + t * TPB; pkr[t] = (FULL || e < G::PACK) ? gpack[(size_t)(k - 1) * G::PACK + e] : 0.0; }
                }
                lds_sync<TPB>();
                // the lower half stores the pivot rows; every other lane shares the padding slot of the row
                double *const gkrow = gkt + (size_t)k * G::KTS + ((cvalid && rh == 0) ? rcol : NZ + 1);
                if (k < kst) {
                    gkrow[G::KTS + (NU - 1) * G::LDC] = pend_row;
                    if (tid < NU) gkt[(size_t)(k + 1) * G::KTS + tid * G::LDC + NZ] = pend_rhs;
                }
                if (ckc == NMPC_CKPT_EVERY) {      // save the cost-to-go entering this stage: state rows and right-hand side as the lanes hold them
                    ckc = 0;
                    double *ck = gck + (size_t)((N - 1 - k) / NMPC_CKPT_EVERY) * ((NX + 1) * 64) + tid;
                    if (cvalid) static_for<0, NS + 1>([&](auto rc) { constexpr int r = decltype(rc)::value; ck[r * 64] = m[NC + r]; });
                }
                ckc++;
                double kO, kA, kB;
                {
                    kO = c_state ? 1.0 : 0.0; kA = lds_ld(PK, cfA, 0); kB = lds_ld(PK, cfB, 0);
                }
                // ---- 1. right-hand side p + P b, b = -c_k: each half sums over its own state rows, the halves are then added
                {
                    double acc = 0.0;
                    static_for<0, NS>([&](auto sc) {
                        constexpr int s = decltype(sc)::value, p = s / 3, d = s - 3 * p;
                        double cv = lds_ld(PK, cbo, 8 * (6 * p + d));
                        if constexpr ((M_ & 1) && p == MP - 1) cv = rh ? 0.0 : cv;
                        acc = fma(m[NC + s], cv, acc);
                    });
                    double a2 = acc, b2 = acc;
                    swap32(a2, b2);                       // a2 = lower half's sum everywhere, b2 = upper half's
                    m[RR] -= a2 + b2;
                }
                // ---- 2. G = P [B A] (and the same combination of the right-hand side): own column + two gathered columns
                {
#ifndef NMPC_RP_GB
#define NMPC_RP_GB (NS + 1)      // rows per batch of the gathers (all of a batch issued before its first use)
#endif
                    constexpr int GB2 = NMPC_RP_GB;
                    static_for<0, (NS + 1 + GB2 - 1) / GB2>([&](auto bc) {
                        constexpr int q0 = decltype(bc)::value * GB2, q1 = (q0 + GB2 < NS + 1) ? q0 + GB2 : NS + 1;
                        double tA[GB2], tB[GB2];
                        static_for<q0, q1>([&](auto rc) { constexpr int q = decltype(rc)::value; tA[q - q0] = lane_gather(srcA2, m[NC + q]); tB[q - q0] = lane_gather(srcB2, m[NC + q]); });
                        static_for<q0, q1>([&](auto rc) { constexpr int q = decltype(rc)::value; m[NC + q] = fma(kB, tB[q - q0], fma(kA, tA[q - q0], kO * m[NC + q])); });
                    });
                }
                PROF_T(9);
                // ---- 3. [B A]^T G: row operations inside each half (a robot's rows live in one half), coefficients per half
                static_for<0, MP>([&](auto pc) {
                    constexpr int p = decltype(pc)::value;
                    double Tc = lds_ld(PK, cf1, 8 * (8 * p)), Ts = lds_ld(PK, cf1, 8 * (8 * p + 1)), Tt = T;
                    double ai = lds_ld(PK, cf1, 8 * (8 * p + 2)), bi = lds_ld(PK, cf1, 8 * (8 * p + 3));
                    if constexpr ((M_ & 1) && p == MP - 1) { Tc = rh ? 0.0 : Tc; Ts = rh ? 0.0 : Ts; Tt = rh ? 0.0 : Tt; ai = rh ? 0.0 : ai; bi = rh ? 0.0 : bi; }
                    const double gx = m[NC + 3 * p], gy = m[NC + 3 * p + 1], gt = m[NC + 3 * p + 2];
                    m[2 * p] = fma(Ts, gy, Tc * gx);
                    m[2 * p + 1] = Tt * gt;
                    m[NC + 3 * p + 2] = fma(bi, gy, fma(ai, gx, gt));
                });
                // ---- 4. Hessian additions and gradient
                static_for<0, RR>([&](auto rc) { constexpr int r = decltype(rc)::value; m[r] += lds_ld(PK, hoff[r], 0); });
                m[RR] += lds_ld(PK, goff, 0);
                PROF_T(10);
                // ---- 5. NU pivot steps in natural control order.  Pivot j is row (half hj, slot ij); its diagonal sits on lane 48 hj + ij,
                //      the right-hand side of column j on lane 16 hj + ij of the lower half.  The pivot test runs on the lanes
                //      themselves (thr = 1e-9 |diagonal as assembled| on the lane that holds the diagonal of its column) and the
                //      reciprocal — or -1 for a rejected pivot — is read from the diagonal's lane; a rejected pivot does not leave the
                //      stage early (the rest of the stage computes garbage that the retry overwrites): no branch per pivot
                double thr = 0.0;
                static_for<0, NC>([&](auto ic) { constexpr int i = decltype(ic)::value; thr = (rh == rw && rn == i) ? m[i] : thr; });
                thr = 1e-9 * fabs(thr);
                auto pivot_inv = [&](double d) { return (d > thr) ? rcp_nr(d) : -1.0; };      // thr >= 0 (or NaN, which rejects): d > thr already says d > 0
                double inv_cur = lane_read(pivot_inv(m[rp_slot(0)]), 48 * rp_half(0) + rp_slot(0));
                double rhsv = 0.0;
                bool okk = true;
                static_for<0, NU>([&](auto jc) {
                    constexpr int j = decltype(jc)::value, hj = rp_half(j), ij = rp_slot(j);
                    const double inv = inv_cur;
                    okk = okk && (inv > 0.0);
                    const double rhs_j = lane_read(m[RR], 16 * hj + ij);
                    rhsv = (tid == j) ? -rhs_j * inv : rhsv;
                    // own = the pivot row on every lane's own column, in both halves.  The swap leaves the OTHER row of the register in
                    // both halves of m[ij]: the pivot row itself is dead from here on, the other one keeps its place
                    double own = m[ij];
                    if constexpr (hj == 0) swap32(own, m[ij]); else swap32(m[ij], own);
                    const double rjv = own * inv;
                    double nrjv = -rjv;
                    {
                        int mc = rcol;
                        asm("" : "+v"(mc));
                        if constexpr (j == NU - 1) pend_row = (mc > j) ? nrjv : 0.0;
                        else gkrow[j * G::LDC] = (mc > j) ? nrjv : 0.0;
                    }
                    // ur = [P0 P0 P1 P1] from own = [P0 P1 P0 P1]
                    double ur = own, ub = own;
                    swap16(ur, ub);                   // ur = [P0 P0 P0 P0], ub = [P1 P1 P1 P1]
                    swap32(ur, ub);                   // ur = [P0 P0 P1 P1]
                    asm("s_nop 1" : "+v"(ur));        // VALU write -> DPP read of the same VGPR: 2 wait states
                    auto elim = [&](auto ac) { constexpr int i = decltype(ac)::value; fmac_rowb<i>(m[i], ur, nrjv); };
                    if constexpr (j + 1 < NU) {       // the next pivot row first: its reciprocal overlaps the other rows
                        elim(std::integral_constant<int, rp_slot(j + 1)>{});
                        inv_cur = lane_read(pivot_inv(m[rp_slot(j + 1)]), 48 * rp_half(j + 1) + rp_slot(j + 1));
                    }
                    // a control register stays live while one of its two rows has not been a pivot yet
                    static_for<0, RR>([&](auto ac) {
                        constexpr int i = decltype(ac)::value;
                        if constexpr (rp_live(i, j, NC, NU) && !(j + 1 < NU && i == rp_slot(j + 1))) elim(ac);
                    });
                    m[RR] = fma(-rhs_j, rjv, m[RR]);
                });
                ok = okk;
                PROF_T(11);
                if (!ok) { kf = k; break; }
                pend_rhs = rhsv;
                if (k == 0) {
                    gkrow[(NU - 1) * G::LDC] = pend_row;
                    if (tid < NU) gkt[tid * G::LDC + NZ] = pend_rhs;
                }
            }
            } else {
            // ---- per-lane byte offsets (into the staged pack) of the Hessian addition of element (row a, my column); rebuilt
            //      per sweep so that they only occupy registers while a sweep runs
            int hoff[NZ];
            static_for<0, NZ>([&](auto ac) {
                constexpr int a = decltype(ac)::value;
                constexpr bool au = a < NU;
                constexpr int sa = au ? 0 : a - NU, ia = au ? (a >> 1) : sa / 3, da = sa - 3 * (sa / 3);
                int h = G::PK_ZERO;
                if constexpr (!au && da < 2) {       // x / y row: pair blocks (negated in the pack) and the xy term of the own robot
                    const int dc = ms - 3 * mrob;
                    const int lo = ia < mrob ? ia : mrob, hi = ia < mrob ? mrob : ia;
                    const int pe = G::PK_E + 3 * (lo * (2 * M_ - lo - 1) / 2 + (hi - lo - 1)) + da + dc;
                    h = (is_state && dc < 2) ? ((ia == mrob) ? G::PK_HXY + ia : pe) : h;
                }
                if constexpr (au && (a & 1) == 0) h = (is_state && ms == 3 * ia + 2) ? G::PK_HVT + ia : h;       // (v_i, theta_i)
                if constexpr (!au && da == 2) h = (is_ctrl && ma == 2 * ia) ? G::PK_HVT + ia : h;               // (theta_i, v_i)
                h = (lvalid && mycol == a) ? G::PK_HD + a : h;
                hoff[a] = 8 * (lvalid ? h : G::PK_ZERO);
            });
            const int goff = 8 * (lvalid ? G::PK_G + mycol : G::PK_ZERO);
            const int cfA = 8 * (th_lane ? G::PK_CF + 4 * mrob + 2 : ((is_ctrl && !(ma & 1)) ? G::PK_CF + 4 * mrob : ((is_ctrl && (ma & 1)) ? G::PK_T : G::PK_ZERO)));
            const int cfB = 8 * (th_lane ? G::PK_CF + 4 * mrob + 3 : ((is_ctrl && !(ma & 1)) ? G::PK_CF + 4 * mrob + 1 : G::PK_ZERO));
            // ---- terminal cost-to-go P_N = diag(hd_N), p_N = g_N: rows NU.. of the state lanes — or the cost-to-go saved on entry to stage kst
            double m[NZ + 1];
            static_for<0, NU>([&](auto rc) { m[decltype(rc)::value] = 0.0; });
            if (kst == N - 1) {
                const double *pkN = gpack + (size_t)N * G::PACK;
                const double hdN = is_state ? pkN[G::PK_HD + NU + ms] : 0.0, gN = is_state ? pkN[G::PK_G + NU + ms] : 0.0;
                static_for<0, NX>([&](auto rc) { constexpr int r = decltype(rc)::value; m[NU + r] = (is_state && ms == r) ? hdN : 0.0; });
                m[NZ] = gN;
            } else {
                const double *ck = gck + (size_t)((N - 1 - kst) / NMPC_CKPT_EVERY) * ((NX + 1) * 64) + tid;
                static_for<0, NX + 1>([&](auto rc) { constexpr int r = decltype(rc)::value; const double v = ck[r * 64]; m[NU + r] = is_state ? v : 0.0; });
            }
            // prefetch pack N-1 into registers
            constexpr int PKR = (G::PACK + TPB - 1) / TPB;
            // whole 64-lane slices where they fit the scratch region (the tail of the last slice reads into the next pack / the terminal
            // block, in bounds: (N+1) packs are allocated and PKR*64 <= 2*PACK): no exec-masked tail in every stage.  Measured +3.3 %
            constexpr bool FULL = PKR * TPB <= RG0 && PKR * TPB <= 2 * G::PACK;
            double pkr[PKR];
#pragma unroll
            for (int t = 0; t < PKR; t++) { int e = tid + t * TPB; pkr[t] = (FULL || e < G::PACK) ? gpack[(size_t)kst * G::PACK + e] : 0.0; }
            // the last pivot row and the right-hand sides of a stage are stored at the top of the NEXT stage, behind the wait for that
            // stage's pack: vmcnt counts loads and stores in order, so a store issued just before the wait adds its whole
            // acknowledgement latency to it
            double pend_row = 0.0, pend_rhs = 0.0;
            int ckc = 0;        // stages since the last saved one (see the row-paired sweep)
#if NMPC_COL_DUMMY_ST
            static_for<0, NMPC_COL_DUMMY_ST>([&](auto jc) { gkt[(size_t)(N - 1) * G::KTS + decltype(jc)::value * G::LDC + NZ + 1] = 0.0; });      // (see the row-paired sweep)
#endif
            for (int k = kst; k >= 0; k--) {
                // ---- stage pack -> LDS (the inertia shift delta joins the control diagonal here); prefetch the next one
                lds_sync<TPB>();         // every read of the previous stage's pack is done
#pragma unroll
                for (int t = 0; t < PKR; t++) {
                    const int e = tid + t * TPB;
                    if (FULL || e < G::PACK) PK[e] = (e >= G::PK_HD && e < G::PK_HD + NU) ? pkr[t] + delta : pkr[t];
                }
                if (k > 0) {
#pragma unroll
                    for (int t = 0; t < PKR; t++) { int e = tid + t * TPB; pkr[t] = (FULL || e < G::PACK) ? gpack[(size_t)(k - 1) * G::PACK + e] : 0.0; }
                }
                lds_sync<TPB>();
                double *const gkrow = gkt + (size_t)k * G::KTS + (lvalid ? mycol : NZ + 1);
                if (k < kst) {
                    gkrow[G::KTS + (NU - 1) * G::LDC] = pend_row;
                    if (tid < NU) gkt[(size_t)(k + 1) * G::KTS + tid * G::LDC + NZ] = pend_rhs;
                }
                if (ckc == NMPC_CKPT_EVERY) {      // save the cost-to-go entering this stage (state lanes: rows NU.. and the right-hand side)
                    ckc = 0;
                    double *ck = gck + (size_t)((N - 1 - k) / NMPC_CKPT_EVERY) * ((NX + 1) * 64) + tid;
                    if (is_state) static_for<0, NX + 1>([&](auto rc) { constexpr int r = decltype(rc)::value; ck[r * 64] = m[NU + r]; });
                }
                ckc++;
                // coefficients of the <= 3 terms of my column of [B A]: own, gathered A, gathered B
                double kO, kA, kB;
                {
                    kO = is_state ? 1.0 : 0.0; kA = lds_ld(PK, cfA, 0); kB = lds_ld(PK, cfB, 0);
                }
                // ---- 1. right-hand side p + P b, b = -c_k (defects broadcast from the pack)
                static_for<0, NX>([&](auto sc) { constexpr int s = decltype(sc)::value; m[NZ] = fma(-m[NU + s], PK[G::PK_C + s], m[NZ]); });
                // ---- 2. G = P [B A] (and the same combination of the right-hand side): own column + two gathered columns
                {
                    // gathers in batches of GB rows, all of a batch before its first use: one LDS-crossbar latency per batch, and
                    // no more than 2 GB gathered values alive next to the matrix (register budget of two waves per SIMD)
                    constexpr int GB = NMPC_COL_GB;
                    static_for<0, (NX + 1 + GB - 1) / GB>([&](auto bc) {
                        constexpr int q0 = decltype(bc)::value * GB, q1 = (q0 + GB < NX + 1) ? q0 + GB : NX + 1;
                        double tA[GB], tB[GB];
                        static_for<q0, q1>([&](auto rc) { constexpr int q = decltype(rc)::value; tA[q - q0] = lane_gather(srcA, m[NU + q]); tB[q - q0] = lane_gather(srcB, m[NU + q]); });
                        static_for<q0, q1>([&](auto rc) { constexpr int q = decltype(rc)::value; m[NU + q] = fma(kB, tB[q - q0], fma(kA, tA[q - q0], kO * m[NU + q])); });
                    });
                }
                PROF_T(9);
                // ---- 3. [B A]^T G: a row operation, i.e. in-lane, with the wave-uniform coefficients of each robot
                static_for<0, M_>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    const double Tc = PK[G::PK_CF + 4 * i], Ts = PK[G::PK_CF + 4 * i + 1], Tt = T;
                    const double ai = PK[G::PK_CF + 4 * i + 2], bi = PK[G::PK_CF + 4 * i + 3];
                    const double gx = m[NU + 3 * i], gy = m[NU + 3 * i + 1], gt = m[NU + 3 * i + 2];
                    m[2 * i] = fma(Ts, gy, Tc * gx);
                    m[2 * i + 1] = Tt * gt;
                    m[NU + 3 * i + 2] = fma(bi, gy, fma(ai, gx, gt));
                });
                // ---- 4. Hessian additions and gradient
                static_for<0, NZ>([&](auto rc) { constexpr int r = decltype(rc)::value; m[r] += lds_ld(PK, hoff[r], 0); });
                m[NZ] += lds_ld(PK, goff, 0);
                PROF_T(10);
                // ---- 5. NU pivot steps.  Row a loses M[j][a] / d_j times the pivot row; M[j][a] is lane L(a) of m[j] (symmetry) and is
                //      broadcast inside the multiply-add (row_newbcast DPP operand), one DP instruction per remaining row
                double d0s[NU];                          // the control diagonals as assembled (pivot test reference), wave-uniform
                static_for<0, NU>([&](auto jc) { constexpr int j = decltype(jc)::value; d0s[j] = lane_read(m[j], LC(j)); });
                auto pivot_inv = [&](double d, double d0) { return (d > 1e-9 * fabs(d0)) ? rcp_nr(d) : -1.0; };      // the bound is >= 0 (or NaN, which rejects): no separate d > 0
                double inv_cur = pivot_inv(d0s[0], d0s[0]);
                double rhsv = 0.0;                       // lane j <- right-hand side of pivot row j, times -1/pivot
                bool okk = true;                         // a rejected pivot does not leave the stage early (as in the row-paired sweep): no branch per pivot
                static_for<0, NU>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    {
                        const double inv = inv_cur;
                        okk = okk && (inv > 0.0);
                        {
                            const double rhs_j = lane_read(m[NZ], LC(j));
                            rhsv = (tid == j) ? -rhs_j * inv : rhsv;
                            const double rjv = m[j] * inv;
                            double nrjv = -rjv;
                            // pivot row j as the forward sweep wants it: times -1/pivot, zero up to the diagonal.  Every lane stores (the
                            // lanes without a column share the padding slot of the row); the comparison is made here, from an opaque copy
                            // of the column index: hoisted out of the sweep its 12 masks are spilled and reloaded lane by lane
                            {
                                int mc = mycol;
                                asm("" : "+v"(mc));
                                if constexpr (j == NU - 1) pend_row = (mc > j) ? nrjv : 0.0;
                                else gkrow[j * G::LDC] = (mc > j) ? nrjv : 0.0;
                            }
#if NMPC_COL_DPP
                            // the pivot row register, each of its rows of 16 lanes replicated into all rows (ur[r] = row r everywhere)
                            double ur[NR < 3 ? NR : 4];
                            if constexpr (NR == 1) asm("s_nop 1" : "+v"(m[j]));      // (see below; in place, no copy of the register)
                            ur[0] = m[j];
                            if constexpr (NR == 2) { ur[1] = m[j]; swap16(ur[0], ur[1]); }
                            if constexpr (NR >= 3) {
                                ur[1] = m[j]; swap16(ur[0], ur[1]);
                                ur[2] = ur[0]; swap32(ur[0], ur[2]);
                                ur[3] = ur[1]; swap32(ur[1], ur[3]);
                            }
                            // VALU write -> DPP read of the same VGPR needs 2 wait states (no VALU writes EXEC here); hipcc does not look into asm
                            // ONE s_nop behind the last of the writes covers all the replicas (it was one per register: four issue slots per pivot)
                            if constexpr (NR == 2) asm("s_nop 1" : "+v"(ur[0]), "+v"(ur[1]));
                            if constexpr (NR >= 3) asm("s_nop 1" : "+v"(ur[0]), "+v"(ur[1]), "+v"(ur[2]), "+v"(ur[3]));
                            auto elim = [&](auto ac) {
                                constexpr int a = decltype(ac)::value;
                                fmac_rowb<(LC(a) & 15)>(m[a], ur[LC(a) >> 4], nrjv);
                            };
#else
                            auto elim = [&](auto ac) {
                                constexpr int a = decltype(ac)::value;
                                m[a] = fma(-lane_read(m[j], LC(a)), rjv, m[a]);
                            };
#endif
                            if constexpr (j + 1 < NU) {       // the next pivot row first: its reciprocal overlaps the other rows
                                elim(std::integral_constant<int, j + 1>{});
                                inv_cur = pivot_inv(lane_read(m[j + 1], LC(j + 1)), d0s[j + 1]);
                            }
                            static_for<(j + 1 < NU ? j + 2 : j + 1), NZ>([&](auto ac) { elim(ac); });
                            m[NZ] = fma(-rhs_j, rjv, m[NZ]);
                        }
                    }
                });
                ok = okk;
                PROF_T(11);
                if (!ok) { kf = k; break; }
                // the right-hand sides of the pivot rows (lanes 0..NU-1 hold them)
                pend_rhs = rhsv;
                if (k == 0) {
                    gkrow[(NU - 1) * G::LDC] = pend_row;
                    if (tid < NU) gkt[tid * G::LDC + NZ] = pend_rhs;
                }
                // what is left in rows NU.. of the state lanes is [P_k | p_k]
            }
            }
            }
            if constexpr (TPBK > 64) {      // wave 0's verdict on the factorisation reaches the other wave through LDS
                if (tid == 0) RED[7] = ok ? 1.0 : 0.0;
                __syncthreads();
                ok = __builtin_amdgcn_readfirstlane(RED[7] != 0.0 ? 1 : 0) != 0;
                __syncthreads();
            }
            if (ok) break;
            ntry++;
            if (delta == 0.0) delta = (delta_last == 0.0) ? 1e-4 : fmax(1e-20, delta_last / 3.0);
            else delta *= (delta_last == 0.0) ? 100.0 : NMPC_SHIFT_ESCALATION;
            if (delta > 1e20) break;
            kst = NMPC_RESUME_STAGE(kf, N);        // (wave 0's: the other waves of the latency shape do not sweep)
        }
        if (!ok) { if (n_cold < NMPC_COLD_RETRIES) { cold_retry(); iter++; } else status = NMPC_STATUS_NUMERIC; break; }
        if (delta > 0.0) delta_last = delta;
        need_shift = delta > 0.0 && (ntry > 0 || delta > 1e-6);
        __syncthreads();   // s_waitcnt vmcnt(0): the stage-0 gains were stored a moment ago by other lanes of this wave
        PROF_T(3);

        // ============ C. forward sweep: du_k = -(t + Uuu' du), t = rhs' + Uux' dx_k with the pivot rows as stored (times -1/pivot),
        // then dx_{k+1} = A dx_k + B du_k - c_k per robot.
        // No LDS staging and no scalar round trips.  Three lane roles, one register each:
        //   S (lane r < NX)        holds dx[r]; updates its own state with its robot's (a, b, c) coefficients
        //   T (lane TB + j)        holds the state part and the right-hand side of pivot row j: t_j = sum_c bcast(dx_c) * row[c],
        //                          NX multiply-adds whose DPP operand is dx (its rows of 16 lanes replicated by v_permlane16_swap)
        //   B (lane TB + 32 + j)   holds the control part of pivot row j (zero up to the diagonal): back substitution, step c
        //                          adds bcast(du_c) * row[c] to every later row's accumulator — one multiply-add per control
        //                          (controls 16.. of the larger teams sit in the second row of 16 B lanes: v_readlane broadcast)
        // Each T / B lane loads ITS row straight from HBM/L2 into registers (dwordx4, PD stages ahead).
        if (TPBK == 64 || tid < 64) {              // wave 0's as well
        constexpr int TPB = 64;
        if constexpr (NU <= 32 && NX <= 32) {
            constexpr int LDC = G::LDC, CT = (NX + 3) & ~1, CH = CT / 2, PD = NMPC_FW_PD;
            static_assert(NU + CT <= LDC && NU <= CT, "row chunks of the forward sweep");
            // first T lane / first B lane (32 apart: v_permlane32_swap carries t across).  Up to 16 controls both roles fit one row of
            // 16 lanes each; measured, rows 1 and 3 (lanes 16.., 48..) are 2.5 % faster for six robots than rows 0 and 2
            constexpr int TB = (NU <= 16) ? 16 : 0, BB = TB + 32;
            const bool isT = tid >= TB && tid < TB + NU, isB = tid >= BB && tid < BB + NU;
            const double2 *rowp = reinterpret_cast<const double2 *>(gkt + (isT ? (tid - TB) * LDC + NU : (isB ? (tid - BB) * LDC : NU)));
            double kq[PD][CT];
#pragma unroll
            for (int d = 0; d < PD; d++)        // definitely assigned: a ring that is only conditionally loaded counts as live across the whole
#pragma unroll
                for (int t = 0; t < CT; t++) kq[d][t] = 0.0;      // interior-point loop (its undefined start value meets the loaded one at every join)
            auto fetch = [&](auto dc, int k) {
                constexpr int d = decltype(dc)::value;
#pragma unroll
                for (int t = 0; t < CH; t++) {      // every lane loads (the lanes without a role read what lane 16 reads): no exec-masked regions
                    const double2 v = rowp[(size_t)(NMPC_FW_FAKE ? 0 : k) * (G::KTS / 2) + t];
                    kq[d][2 * t] = v.x; kq[d][2 * t + 1] = v.y;
                }
            };
            static_for<0, PD>([&](auto dc) { if (decltype(dc)::value < N) fetch(dc, decltype(dc)::value); });
            double dx = 0.0;
            if (is_state) DX[tid] = 0.0;
            for (int k0 = 0; k0 < N; k0 += PD) {
                static_for<0, PD>([&](auto dc) {
                    constexpr int d = decltype(dc)::value;
                    const int k = k0 + d;
                    if (k < N) {
                        // lane constants rebuilt from the lane id per stage (a handful of integer operations): kept across the loop they are
                        // spilled, and a spill reload inside the loop waits (vmcnt is in order) for the row prefetches in flight
                        int ln = tid;
                        asm volatile("" : "+v"(ln));
                        const int fr = ln < NX ? ln : 0, fi = fr / 3, fd = fr - 3 * fi;
                        const int srcT = 4 * (3 * fi + 2), srcU = 4 * (BB + (fd < 2 ? 2 * fi : 2 * fi + 1));
                        // coefficients of my state's row of [A B | c] (independent of the recursion: issued first)
                        const double sn = SN[k * M_ + fi], cs = CS[k * M_ + fi], uv = U[k * NU + 2 * fi], uw = U[k * NU + 2 * fi + 1];
                        const double xk = X[k * NX + fr], xn = X[(k + 1) * NX + fr];
                        const double dxt = lane_gather(srcT, dx);
                        double u0 = dx, u1 = dx;
                        swap16(u0, u1);                   // u0 = lanes 0..15 of dx in both rows of each half, u1 = lanes 16..31
                        asm("s_nop 1" : "+v"(u0));
                        if constexpr (NX > 16) asm("s_nop 0" : "+v"(u1));
                        double t = kq[d][NX];
                        static_for<0, NX>([&](auto cc) {
                            constexpr int c = decltype(cc)::value;
                            if constexpr (c < 16) fmac_rowb<(c & 15)>(t, u0, kq[d][c]);
                            else fmac_rowb<(c & 15)>(t, u1, kq[d][c]);
                        });
                        double tb = t, tl = t;
                        swap32(tb, tl);                   // lanes BB + j <- t of lanes TB + j
                        static_for<1, NU>([&](auto cc) {
                            constexpr int c = NU - decltype(cc)::value;      // NU-1 .. 1: du_c is final on lane BB + c when its turn comes
                            if constexpr (c < 16) {
                                asm("s_nop 1" : "+v"(tb));
                                double ub = tb;
                                fmac_rowb<c>(tb, ub, kq[d][c]);
                            } else {
                                tb = fma(kq[d][c], lane_read(tb, BB + c), tb);
                            }
                        });
                        if (ln >= BB && ln < BB + NU) DU[k * NU + (ln - BB)] = tb;
                        const double du = lane_gather(srcU, tb);
                        if (k + PD < N) fetch(dc, k + PD);
                        // dx+ = dx + a dx_theta + b du - c:  x: a = -T v sin, b = T cos;  y: a = T v cos, b = T sin;  theta: a = 0, b = T
                        const double Tv = T * uv;
                        const double ca = (fd == 0) ? -Tv * sn : ((fd == 1) ? Tv * cs : 0.0);
                        const double cb = (fd == 0) ? T * cs : ((fd == 1) ? T * sn : T);
                        const double dfc = (fd == 2) ? defect_th(xn, xk, T, uw) : defect_xy(xn, xk, Tv, (fd == 0) ? cs : sn);
                        dx = fma(cb, du, fma(ca, dxt, dx)) - dfc;
                        if (ln < NX) DX[(k + 1) * NX + ln] = dx;
                    }
                });
            }
        } else {
            // larger teams: the stage factors stream back through a register ring filled PD stages ahead by all lanes (coalesced),
            // then through an LDS staging buffer (the pack area, free here); the control lanes form t and substitute back with
            // v_readlane broadcasts
            constexpr int KPT = (G::KTS + TPB - 1) / TPB, PD = 4, LDC = G::LDC;
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
                        {
                            const double *row = FS + jrow * LDC;
                            double t0 = row[NZ], t1 = 0.0, t2 = 0.0;
#pragma unroll
                            for (int c = 0; c + 2 < NX; c += 3) {
                                t0 = fma(row[NU + c], DX[k * NX + c], t0);
                                t1 = fma(row[NU + c + 1], DX[k * NX + c + 1], t1);
                                t2 = fma(row[NU + c + 2], DX[k * NX + c + 2], t2);
                            }
                            double tj = t0 + (t1 + t2);
                            double urow[NU];
#pragma unroll
                            for (int c = 0; c < NU; c++) urow[c] = row[c];
                            double duj = 0.0;
                            static_for<0, NU>([&](auto cc) {
                                constexpr int c = NU - 1 - decltype(cc)::value;
                                const double duc = lane_read(tj, c);       // final on lane c: all its later columns are in
                                if (tid == c) duj = duc;
                                tj = fma(urow[c], duc, tj);                // rows j < c use it (urow[c] is zero for rows j >= c)
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
        }
        }
        __syncthreads();
        PROF_T(4);

        // ============ D. fraction to the boundary (IPOPT eq. 15) over all inequality slots (ds, dz recomputed)
        double a_p = 1.0, a_d = 1.0, mult_max = 0.0;   // mult_max: inf-norm of the QP multipliers of the rows that enter theta
#if NMPC_FUSE_DPHI
        double dphi_b = 0.0;        // barrier part of the merit function's directional derivative, -mu sum ds / s: same slots, same ds (one pass less)
#endif
        auto fb = [&](double sv, double zv, double ds) -> double {
            double dz = dz_of(mu, sv, zv, ds);
            if (ds < 0.0) a_p = fmin(a_p, qdiv(-tau * sv, ds));
            if (dz < 0.0) a_d = fmin(a_d, qdiv(-tau * zv, dz));
#if NMPC_FUSE_DPHI
            dphi_b += qdiv(ds, sv);
#endif
            return zv + dz;
        };
        auto fbe = [&](double sv, double zv, double tv, double hv, double jd) -> double {      // an elastic row: s, t >= 0, 0 < z < rho
            double ds, dz, dt;
            el_step(mu, rho, sv, zv, tv, hv, jd, ds, dz, dt);
            if (ds < 0.0) a_p = fmin(a_p, qdiv(-tau * sv, ds));
            if (dt < 0.0) a_p = fmin(a_p, qdiv(-tau * tv, dt));
            if (dz < 0.0) a_d = fmin(a_d, qdiv(-tau * zv, dz));
            if (dz > 0.0) a_d = fmin(a_d, qdiv(tau * (rho - zv), dz));
            dphi_b += qdiv(ds, sv) + qdiv(dt, tv) - qdiv(rho, mu) * dt;      // -mu dphi_b = -mu (ds/s + dt/t) + rho dt: the penalty's part rides along (no second accumulator)
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
        if (!el) {
        for (int it = tid; it < (N - 1) * NPA; it += TPB) {
            int k = 1 + it / NPd, q = it - (k - 1) * NP, i, j;
            pair_ij(q, i, j);
            const int oi = k * NX + 3 * i, oj = k * NX + 3 * j;
            double ex = X[oi] - X[oj], ey = X[oi + 1] - X[oj + 1], sv = SPp[k * NP + q];
            double ds = ds_pair(ex, ey, DX[oi] - DX[oj], DX[oi + 1] - DX[oj + 1], P.dmin2, sv);
            mult_max = fmax(mult_max, fabs(fb(sv, ZPp[k * NP + q], ds)));
        }
        for (int it = tid; it < (N - 1) * MK; it += TPB) {
            int k = 1 + it / MK, e = it - (k - 1) * MK, i = e / K, o = e - i * K;
            const int oi = k * NX + 3 * i;
            double ex = X[oi] - P.obs[3 * o], ey = X[oi + 1] - P.obs[3 * o + 1], rr = r_obs(ex, ey), sv = SO[k * MK + e];
            double ds = ds_obs(ex, ey, rr, DX[oi], DX[oi + 1], h_obs(rr, P.robdim, P.obs[3 * o + 2], P.margin), sv);
            mult_max = fmax(mult_max, fabs(fb(sv, ZO[k * MK + e], ds)));
        }
        } else {        // elastic phase: the same two loops over elastic rows
        for (int it = tid; it < (N - 1) * NPA; it += TPB) {
            int k = 1 + it / NPd, q = it - (k - 1) * NP, i, j;
            pair_ij(q, i, j);
            const int oi = k * NX + 3 * i, oj = k * NX + 3 * j;
            double ex = X[oi] - X[oj], ey = X[oi + 1] - X[oj + 1];
            mult_max = fmax(mult_max, fabs(fbe(SPp[k * NP + q], ZPp[k * NP + q], TPp[k * NP + q], h_pair(ex, ey, P.dmin2), jd_pair(ex, ey, DX[oi] - DX[oj], DX[oi + 1] - DX[oj + 1]))));
        }
        for (int it = tid; it < (N - 1) * MK; it += TPB) {
            int k = 1 + it / MK, e = it - (k - 1) * MK, i = e / K, o = e - i * K;
            const int oi = k * NX + 3 * i;
            double ex = X[oi] - P.obs[3 * o], ey = X[oi + 1] - P.obs[3 * o + 1], rr = r_obs(ex, ey);
            mult_max = fmax(mult_max, fabs(fbe(SO[k * MK + e], ZO[k * MK + e], TOb[k * MK + e], h_obs(rr, P.robdim, P.obs[3 * o + 2], P.margin), jd_obs(ex, ey, rr, DX[oi], DX[oi + 1]))));
        }
        }
        a_p = wmin<TPB>(a_p, RED); a_d = wmin<TPB>(a_d, RED);
        PROF_T(5);
        // ============ F. multipliers of the QP: stage-parallel residuals, then the robot-local adjoint recursion in registers
        //   lam+_k = A_k^T lam+_{k+1} - (grad f_k + W_k dx_k + W_xu du_k) + Jx_k^T (z + dz)_k
        auto multiplier_residuals = [&](auto elc) {
        constexpr bool EL = decltype(elc)::value;
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
                if (prs) {
                    double sq[MPN], zq[MPN];
                    static_for<0, M_ - 1>([&](auto pc) {
                        constexpr int p = decltype(pc)::value;
                        const int q = pidx_any<M_>(i, p + (p >= i ? 1 : 0));
                        sq[p] = SPp[k * NP + q]; zq[p] = ZPp[k * NP + q];
                    });
                    static_for<0, M_ - 1>([&](auto pc) {
                        constexpr int p = decltype(pc)::value;
                        const int j = p + (p >= i ? 1 : 0);
                        double ex = xi - x[3 * j], ey = yi - x[3 * j + 1];
                        double ddx = dx[3 * i] - dx[3 * j], ddy = dx[3 * i + 1] - dx[3 * j + 1];
                        double sv = sq[p], zv = zq[p], znew;
                        if constexpr (EL) {
                            double ds, dz, dt;
                            el_step(mu, rho, sv, zv, TPp[k * NP + pidx_any<M_>(i, j)], h_pair(ex, ey, P.dmin2), jd_pair(ex, ey, ddx, ddy), ds, dz, dt);
                            znew = zv + dz;
                        } else {
                        double ds = ds_pair(ex, ey, ddx, ddy, P.dmin2, sv);
                        znew = zv + dz_of(mu, sv, zv, ds);
                        }
                        l0 += 2 * ex * znew + 2 * zv * ddx;      // Jx^T (z+dz)  -  (-2 z (ddx))  [exact-Hessian term of the pair row]
                        l1 += 2 * ey * znew + 2 * zv * ddy;
                    });
                }
                for (int o = 0; o < K; o++) {
                    double ex = xi - P.obs[3 * o], ey = yi - P.obs[3 * o + 1], rr = r_obs(ex, ey), n0 = qdiv(ex, rr), n1 = qdiv(ey, rr);
                    double sv = SO[k * MK + i * K + o], zv = ZO[k * MK + i * K + o];
                    double nd = n0 * dx[3 * i] + n1 * dx[3 * i + 1];
                    double znew, zz = qdiv(zv, rr);
                    if constexpr (EL) {
                        double ds, dz, dt;
                        el_step(mu, rho, sv, zv, TOb[k * MK + i * K + o], h_obs(rr, P.robdim, P.obs[3 * o + 2], P.margin), jd_obs(ex, ey, rr, dx[3 * i], dx[3 * i + 1]), ds, dz, dt);
                        znew = zv + dz;
                    } else {
                    double ds = ds_obs(ex, ey, rr, dx[3 * i], dx[3 * i + 1], h_obs(rr, P.robdim, P.obs[3 * o + 2], P.margin), sv);
                    znew = zv + dz_of(mu, sv, zv, ds);
                    }
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
        };
        if (el) multiplier_residuals(std::true_type{}); else multiplier_residuals(std::false_type{});
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
#if NMPC_FUSE_DPHI
        for (int it = tid; it < N * M_; it += TPB) {      // objective part only: the barrier part was summed by the step-length pass (D)
            int k = it / M_, i = it - k * M_;
            if (k >= 1) {
#pragma unroll
                for (int d = 0; d < 3; d++) dphi += 2 * P.q[d] * (X[k * NX + 3 * i + d] - XS[3 * i + d]) * DX[k * NX + 3 * i + d];
            }
#pragma unroll
            for (int d = 0; d < 2; d++) { const int eu = k * NU + 2 * i + d; dphi += 2 * P.r[d] * U[eu] * DU[eu]; }
        }
        dphi -= mu * dphi_b;
#else
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
#endif
#if !NMPC_FUSE_DPHI
        for (int it = tid; it < (N - 1) * NPA; it += TPB) {
            int k = 1 + it / NPd, q = it - (k - 1) * NP, i, j;
            pair_ij(q, i, j);
            const int oi = k * NX + 3 * i, oj = k * NX + 3 * j;
            double ex = X[oi] - X[oj], ey = X[oi + 1] - X[oj + 1], sv = SPp[k * NP + q];
            double ds = ds_pair(ex, ey, DX[oi] - DX[oj], DX[oi + 1] - DX[oj + 1], P.dmin2, sv);
            dphi -= mu * ds / sv;
        }
        for (int it = tid; it < (N - 1) * MK; it += TPB) {
            int k = 1 + it / MK, e = it - (k - 1) * MK, i = e / K, o = e - i * K;
            const int oi = k * NX + 3 * i;
            double ex = X[oi] - P.obs[3 * o], ey = X[oi + 1] - P.obs[3 * o + 1], rr = r_obs(ex, ey), sv = SO[k * MK + e];
            double ds = ds_obs(ex, ey, rr, DX[oi], DX[oi + 1], h_obs(rr, P.robdim, P.obs[3 * o + 2], P.margin), sv);
            dphi -= mu * ds / sv;
        }
#endif
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
            return fmin(fmax(z, qdiv(1e-10 * mu, snew)), qdiv(1e10 * mu, snew));
        };
        auto zupe = [&](double sv, double zv, double tv, double hv, double jd, double &snew, double &tnew) {      // an elastic row
            double ds, dz, dt;
            el_step(mu, rho, sv, zv, tv, hv, jd, ds, dz, dt);
            snew = sv + alpha * ds; tnew = tv + alpha * dt;
            double z = zv + a_d * dz;
            return fmin(fmax(z, qdiv(1e-10 * mu, snew)), fmin(qdiv(1e10 * mu, snew), rho - qdiv(1e-10 * mu, tnew)));
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
        if (!el) {
        for (int it = tid; it < (N - 1) * NPA; it += TPB) {
            int k = 1 + it / NPd, q = it - (k - 1) * NP, i, j;
            pair_ij(q, i, j);
            const int oi = k * NX + 3 * i, oj = k * NX + 3 * j;
            double ex = X[oi] - X[oj], ey = X[oi + 1] - X[oj + 1], sv = SPp[k * NP + q], sn;
            double ds = ds_pair(ex, ey, DX[oi] - DX[oj], DX[oi + 1] - DX[oj + 1], P.dmin2, sv);
            ZPp[k * NP + q] = zup(sv, ZPp[k * NP + q], ds, sn);
            SPp[k * NP + q] = sn;
        }
        for (int it = tid; it < (N - 1) * MK; it += TPB) {
            int k = 1 + it / MK, e = it - (k - 1) * MK, i = e / K, o = e - i * K;
            const int oi = k * NX + 3 * i;
            double ex = X[oi] - P.obs[3 * o], ey = X[oi + 1] - P.obs[3 * o + 1], rr = r_obs(ex, ey), sv = SO[k * MK + e], sn;
            double ds = ds_obs(ex, ey, rr, DX[oi], DX[oi + 1], h_obs(rr, P.robdim, P.obs[3 * o + 2], P.margin), sv);
            ZO[k * MK + e] = zup(sv, ZO[k * MK + e], ds, sn);
            SO[k * MK + e] = sn;
        }
        } else {        // elastic phase
        for (int it = tid; it < (N - 1) * NPA; it += TPB) {
            int k = 1 + it / NPd, q = it - (k - 1) * NP, i, j;
            pair_ij(q, i, j);
            const int oi = k * NX + 3 * i, oj = k * NX + 3 * j;
            double ex = X[oi] - X[oj], ey = X[oi + 1] - X[oj + 1], sn, tn;
            ZPp[k * NP + q] = zupe(SPp[k * NP + q], ZPp[k * NP + q], TPp[k * NP + q], h_pair(ex, ey, P.dmin2), jd_pair(ex, ey, DX[oi] - DX[oj], DX[oi + 1] - DX[oj + 1]), sn, tn);
            SPp[k * NP + q] = sn; TPp[k * NP + q] = tn;
        }
        for (int it = tid; it < (N - 1) * MK; it += TPB) {
            int k = 1 + it / MK, e = it - (k - 1) * MK, i = e / K, o = e - i * K;
            const int oi = k * NX + 3 * i;
            double ex = X[oi] - P.obs[3 * o], ey = X[oi + 1] - P.obs[3 * o + 1], rr = r_obs(ex, ey), sn, tn;
            ZO[k * MK + e] = zupe(SO[k * MK + e], ZO[k * MK + e], TOb[k * MK + e], h_obs(rr, P.robdim, P.obs[3 * o + 2], P.margin), jd_obs(ex, ey, rr, DX[oi], DX[oi + 1]), sn, tn);
            SO[k * MK + e] = sn; TOb[k * MK + e] = tn;
        }
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

// LDS bytes of one instance of the column-per-lane kernel
template <int M_, int THB> static size_t col_lds_bytes(const KParams &P, bool duals, int fl = 0)      // fl: 2 = + stage factors, 3 = + stage packs (see the kernel)
{
    using G = G2<M_, THB>;
    const size_t N = P.N, N1 = P.N + 1, MK = (size_t)M_ * P.K;
    size_t d = N1 * G::NX * 3 + N * G::NU * 2 + N * M_ * 2;
    const size_t rg0 = G::KTS > G::PACK ? G::KTS : G::PACK;
    d += (N1 * G::NX > rg0) ? N1 * G::NX : rg0;
    d += G::NX + 8;
    if (duals) d += 2 * N1 * G::NP + 2 * N1 * MK + 4 * N * G::NU + 2 * N1 * G::NXB;
    if (duals && fl >= 2) d += N * G::KTS;
    if (duals && fl >= 3) d += N1 * GC<M_, THB>::PACK;
    return d * sizeof(double);
}
// slacks / duals in LDS: up to four robots, when eight instances still share a CU
#ifndef NMPC_COL_DL_MAXM
#define NMPC_COL_DL_MAXM 4
#endif
#ifndef NMPC_COL_DL_BYTES
#define NMPC_COL_DL_BYTES (20 * 1024)
#endif
template <int M_, int THB> static bool col_duals_in_lds(const KParams &P) { return M_ <= NMPC_COL_DL_MAXM && col_lds_bytes<M_, THB>(P, true) <= NMPC_COL_DL_BYTES; }
// Stage factors in LDS too (kernel mode DL = 2; 3 = the stage packs as well): TWO robots only.  Measured A/B in one session (round 4, B = 4096 /
// 1024 / 16384): two robots +7.5 % / +4.5 % / +5 % with the factors in LDS (18 KB per instance at N = 20: eight instances per CU instead of
// fifteen, still faster: the forward sweep reads its rows from LDS, the backward sweep stores them there), -6 % / +3 % / -22 % with the packs
// in LDS as well (29 KB); one robot -8 % (5 -> 7.5 KB: its launch lives on occupancy), three robots -28 % (34 KB).  Identical results.
#ifndef NMPC_COL_FL_MINM
#define NMPC_COL_FL_MINM 2
#endif
#ifndef NMPC_COL_FL_MAXM
#define NMPC_COL_FL_MAXM 2
#endif
#ifndef NMPC_COL_FL_MODE
#define NMPC_COL_FL_MODE 2
#endif
#ifndef NMPC_COL_FL_BYTES
#define NMPC_COL_FL_BYTES (20 * 1024)
#endif
template <int M_, int THB> static int col_factor_mode(const KParams &P)
{
    if (M_ < NMPC_COL_FL_MINM || M_ > NMPC_COL_FL_MAXM || !col_duals_in_lds<M_, THB>(P)) return 0;
    return col_lds_bytes<M_, THB>(P, true, NMPC_COL_FL_MODE) <= NMPC_COL_FL_BYTES ? NMPC_COL_FL_MODE : 0;
}

template <int M_, int THB> static hipError_t launch3_mt(const KParams &P, int B, const double *p, const double *w0, double *w_out, double *obj,
                                                        int32_t *status, int32_t *iters, double *kkt, double *ws, long long *prof, hipStream_t st, int shape)
{
    constexpr int DLmax = (M_ <= NMPC_COL_DL_MAXM) ? 1 : 0;
    // latency shape (two wavefronts per instance): slacks and duals in LDS up to six robots — occupancy is not what a launch that lasts as
    // long as its longest solve is short of — and in the workspace for eight and ten
    constexpr int DLlat = (M_ <= 6) ? 1 : 0;
    const bool lat = shape >= 1;
    const bool dl = lat ? (DLlat != 0) : (DLmax && col_duals_in_lds<M_, THB>(P));
    constexpr int FLM = (M_ >= NMPC_COL_FL_MINM && M_ <= NMPC_COL_FL_MAXM) ? NMPC_COL_FL_MODE : 1;      // instantiated only for the team sizes of the switch
    const int fl = (!lat && dl) ? col_factor_mode<M_, THB>(P) : 0;
    size_t lds = col_lds_bytes<M_, THB>(P, dl, fl);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    // four wavefronts per instance (shape 2) are instantiated for five and six robots only: they pay where a phase has > 1000 items (eight
    // obstacles: composite B=1024 28.7 k -> 33.5 k solves/s; six robots without obstacles B=512 42.3 k -> 43.4 k) and the build time counts
    constexpr int TPB4 = (M_ == 5 || M_ == 6) ? 256 : 128;
    if (shape == 2 && TPB4 == 128) shape = 1;
    auto kern = shape == 2 ? solve_col_kernel<M_, THB, DLlat, TPB4> : (lat ? solve_col_kernel<M_, THB, DLlat, 128> : (fl ? solve_col_kernel<M_, THB, DLmax * FLM, 64> : (dl ? solve_col_kernel<M_, THB, DLmax, 64> : solve_col_kernel<M_, THB, 0, 64>)));
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(B), dim3(shape == 2 ? 256 : (lat ? 128 : 64)), lds, st, P, p, w0, w_out, obj, status, iters, kkt, ws, prof);
    return hipGetLastError();
}
template <int M_> static hipError_t launch3_m(const KParams &P, int B, const double *p, const double *w0, double *w_out, double *obj, int32_t *status,
                                              int32_t *iters, double *kkt, double *ws, long long *prof, hipStream_t st, int shape)
{
    return P.thb ? launch3_mt<M_, 1>(P, B, p, w0, w_out, obj, status, iters, kkt, ws, prof, st, shape)
                 : launch3_mt<M_, 0>(P, B, p, w0, w_out, obj, status, iters, kkt, ws, prof, st, shape);
}

// The team sizes are instantiated in up to three objects so that the build spreads over the cores (one hipcc -c per part, see build.py;
// the ten team sizes take ~350 s in one translation unit): NMPC_COL_PART 0 = everything here (variants, NMPC_COL_ONLY_M), 1 = the entry
// points below plus 1..5 robots, 2 = 6..8 robots, 3 = 9..10 robots.
#ifndef NMPC_COL_PART
#define NMPC_COL_PART 0
#endif
#define COL_ARGS_DECL const KParams &P, int m, int B, const double *p, const double *w0, double *w_out, double *obj, int32_t *status, int32_t *iters, \
                      double *kkt, double *ws, long long *prof, hipStream_t st, int shape
#define COL_CASE(M) case M: return launch3_m<M>(P, B, p, w0, w_out, obj, status, iters, kkt, ws, prof, st, shape);
#if NMPC_COL_PART == 2
hipError_t launch_solve_col_part2(COL_ARGS_DECL)
{
    switch (m) {
        COL_CASE(6) COL_CASE(7) COL_CASE(8)
    default: return hipErrorInvalidValue;
    }
}
#elif NMPC_COL_PART == 3
hipError_t launch_solve_col_part3(COL_ARGS_DECL)
{
    switch (m) {
        COL_CASE(9) COL_CASE(10)
    default: return hipErrorInvalidValue;
    }
}
#else
#if NMPC_COL_PART == 1
hipError_t launch_solve_col_part2(COL_ARGS_DECL);
hipError_t launch_solve_col_part3(COL_ARGS_DECL);
#endif
// shape: 0 = throughput (one wavefront per instance), 1 / 2 = latency (two / four wavefronts per instance, see the kernel)
hipError_t launch_solve_col(COL_ARGS_DECL)
{
    switch (m) {
#ifdef NMPC_COL_ONLY_M
        COL_CASE(NMPC_COL_ONLY_M)
#else
        COL_CASE(1) COL_CASE(2) COL_CASE(3) COL_CASE(4) COL_CASE(5)
#if NMPC_COL_PART == 1
    case 6: case 7: case 8: return launch_solve_col_part2(P, m, B, p, w0, w_out, obj, status, iters, kkt, ws, prof, st, shape);
    case 9: case 10: return launch_solve_col_part3(P, m, B, p, w0, w_out, obj, status, iters, kkt, ws, prof, st, shape);
#else
        COL_CASE(6) COL_CASE(7) COL_CASE(8) COL_CASE(9) COL_CASE(10)
#endif
#endif
    default: return hipErrorInvalidValue;
    }
}

// LDS bytes one instance of the column-per-lane kernel needs in the given shape (0 if m is not supported)
size_t col_kernel_bytes(const KParams &P, int m, int shape)
{
#define LB(M) case M: return P.thb ? col_lds_bytes<M, 1>(P, shape == 1 ? (M <= 6) : col_duals_in_lds<M, 1>(P), shape == 0 ? col_factor_mode<M, 1>(P) : 0) : col_lds_bytes<M, 0>(P, shape == 1 ? (M <= 6) : col_duals_in_lds<M, 0>(P), shape == 0 ? col_factor_mode<M, 0>(P) : 0);
    switch (m) {
        LB(1) LB(2) LB(3) LB(4) LB(5) LB(6) LB(7) LB(8) LB(9) LB(10)
    default: return 0;
    }
#undef LB
}

#endif      // NMPC_COL_PART
#undef COL_CASE
#undef COL_ARGS_DECL

}  // namespace nmpc
