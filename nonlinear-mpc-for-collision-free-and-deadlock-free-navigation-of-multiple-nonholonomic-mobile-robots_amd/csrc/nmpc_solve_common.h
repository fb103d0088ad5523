// nmpc_solve_common.h — geometry, pinned-order arithmetic helpers and wave primitives shared by the LDS-resident solve kernels
// (nmpc_solve_lds.hip: element-per-lane Riccati stage; nmpc_solve_col.hip: column-per-lane, register-resident Riccati stage).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <type_traits>

#include "nmpc_device.h"

namespace nmpc {

template <int M_, int THB> struct G2 {
    static constexpr int NX = 3 * M_, NU = 2 * M_, NP = M_ * (M_ - 1) / 2, NZ = NX + NU, LD = NZ + 1;
    static constexpr int NXB = M_ * (2 + THB);
    static constexpr int NT = NZ * (NZ + 1) / 2 + NZ;   // upper triangle + rhs column
    // stage pack layout (doubles)
    static constexpr int PK_G = 0;                 // [NZ]  rhs gradient  (u part then x part)
    static constexpr int PK_HD = NZ;               // [NZ]  diagonal Hessian additions (without delta)
    static constexpr int PK_HXY = 2 * NZ;          // [M]   (x_i,y_i) entry of robot i's diagonal block
    static constexpr int PK_HVT = 2 * NZ + M_;     // [M]   (v_i,theta_i) cross term
    static constexpr int PK_E = 2 * NZ + 2 * M_;   // [3 NP] NEGATED pair blocks -E_ij
    static constexpr int PK_C = PK_E + 3 * (NP > 0 ? NP : 0);   // [NX]  defects c_k
    static constexpr int PK_CF = PK_C + NX;        // [NZ][3] coefficients of the <=3 terms of each row/column of [B A]
    static constexpr int PK_ZERO = PK_CF + 3 * NZ; // one zero entry (target of "no Hessian addition")
    static constexpr int PACK = ((PK_ZERO + 1 + 7) / 8) * 8;
    static constexpr int LDG = NZ + 1;             // row stride of G = P [B A | b-part]
    // element e of the row-major upper triangle (+ rhs column) lives in row row_of(e); thread tid owns e = tid + t*TPB, so
    // "slice" t holds rows slice_lo(t)..slice_hi(t): compile-time knowledge of which slices a pivot step touches
    static constexpr int row_of(int e) { int a = 0; while (e >= NZ - a + 1) { e -= NZ - a + 1; a++; } return a; }
    static constexpr int diag_e(int j) { int e = 0; for (int a = 0; a < j; a++) e += NZ - a + 1; return e; }   // flat index of element (j,j)
    static constexpr int slice_lo(int t, int tpb) { return row_of(t * tpb); }
    static constexpr int slice_hi(int t, int tpb) { return row_of((t * tpb + tpb - 1 < NT) ? (t * tpb + tpb - 1) : (NT - 1)); }
    // per stage: the NU pivot rows.  Element-per-lane kernel: [Uuu | Uux | rhs] at row stride LD, then the NU reciprocal pivots.
    // Column-per-lane kernel: the rows scaled by -1/pivot, [-Uuu/d (strictly upper part, else 0) | -Uux/d | -rhs/d | pad] at the
    // even row stride LDC (16-byte aligned rows: the forward sweep loads its row with dwordx4)
    static constexpr int LDC = (NZ + 3) & ~1;
    static constexpr int KTS = ((NU * LD + NU > NU * LDC ? NU * LD + NU : NU * LDC) + 7) / 8 * 8;
};

// ---- single evaluation points.  Constraint values, slack steps and defects are recomputed at several places of an
// iteration (Newton right-hand side, step-size rule, multiplier recursion, update).  With sigma = z/s up to 1e13 a
// last-bit difference between two sites (e.g. a differently contracted a*b+c) shows up as 1e-7 in the dual residual, so
// every site goes through these helpers, whose operation order is pinned with explicit fma().
// Divisions of the stage-parallel phases (sigma = z / s, mu / s, dz, the step-length quotients): NMPC_FAST_DIV = 1 replaces the IEEE division
// chain (~12 instructions: div_scale x2, rcp, four multiply-adds, div_fmas, div_fixup) by v_rcp_f64 + one Newton step and a multiply (4): relative
// error <= 2.3e-15 against the division (tools/rcp_probe.hip).  Every site of one quantity goes through the same helper, so the sites stay
// consistent with each other (see above).
// A/B in one session (round 4): six robots B=4096 262 k -> 280 k solves/s, B=16384 388 k -> 410 k, a lone wave +10 %, mean iterations 29.0735 -> 29.0720.
#ifndef NMPC_FAST_DIV
#define NMPC_FAST_DIV 1
#endif
__device__ __forceinline__ double rcp1(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    return fma(fma(-d, r, 1.0), r, r);
}
__device__ __forceinline__ double qdiv(double a, double b)
{
#if NMPC_FAST_DIV
    return a * rcp1(b);
#else
    return a / b;
#endif
}
__device__ __forceinline__ double h_pair(double ex, double ey, double dmin2) { return fma(ex, ex, ey * ey) - dmin2; }
__device__ __forceinline__ double ds_pair(double ex, double ey, double ddx, double ddy, double dmin2, double sv)
{
    return fma(2.0 * ex, ddx, (2.0 * ey) * ddy) + (h_pair(ex, ey, dmin2) - sv);
}
__device__ __forceinline__ double r_obs(double ex, double ey) { return sqrt(fma(ex, ex, ey * ey)); }
__device__ __forceinline__ double h_obs(double rr, double robdim, double orad, double margin) { return rr - robdim - orad - margin; }
__device__ __forceinline__ double ds_obs(double ex, double ey, double rr, double d0, double d1, double hv, double sv)
{
    return qdiv(fma(ex, d0, ey * d1), rr) + (hv - sv);
}
__device__ __forceinline__ double defect_xy(double xn, double x, double tu, double cs) { return xn - fma(tu, cs, x); }   // tu = T*v
__device__ __forceinline__ double defect_th(double xn, double x, double T, double w) { return xn - fma(T, w, x); }
__device__ __forceinline__ double ds_bound(double jd, double hv, double sv) { return jd + (hv - sv); }
__device__ __forceinline__ double dz_of(double mu, double sv, double zv, double ds) { return qdiv(fma(-zv, ds, fma(-sv, zv, mu)), sv); }

// ---- elastic phase (the second restart of last resort; derivation in oracle/nmpc_oracle.c): a pair / obstacle row h + t - s = 0, s, t >= 0,
// dual z in (0, rho).  jd = J dx of the row (ds_pair / ds_obs minus their (h - s) term).
__device__ __forceinline__ double jd_pair(double ex, double ey, double ddx, double ddy) { return fma(2.0 * ex, ddx, (2.0 * ey) * ddy); }
__device__ __forceinline__ double jd_obs(double ex, double ey, double rr, double d0, double d1) { return qdiv(fma(ex, d0, ey * d1), rr); }
__device__ __forceinline__ void el_sigma(double mu, double rho, double sv, double zv, double tv, double hv, double &sg, double &v)
{
    const double rz = rho - zv, D = qdiv(sv, zv) + qdiv(tv, rz), q = hv - qdiv(mu, zv) + qdiv(mu, rz);
    sg = qdiv(1.0, D); v = zv - q * sg;
}
__device__ __forceinline__ void el_step(double mu, double rho, double sv, double zv, double tv, double hv, double jd, double &ds, double &dz, double &dt)
{
    const double rz = rho - zv, D = qdiv(sv, zv) + qdiv(tv, rz), q = hv - qdiv(mu, zv) + qdiv(mu, rz);
    dz = -qdiv(jd + q, D);
    ds = qdiv(mu - sv * zv - sv * dz, zv);
    dt = qdiv(mu - tv * rz + tv * dz, rz);
}

// compile-time loop: f(std::integral_constant<int, I>) for I = B .. E-1 (indices usable in `if constexpr`)
template <int B, int E, class F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// LDS access through a per-thread byte offset fixed at kernel start plus a compile-time constant: lowers to
// ds_read_b64 / ds_write_b64 with the constant in the instruction's offset field (no address arithmetic in the hot loops)
__device__ __forceinline__ double lds_ld(const double *sm, int byteoff, int cbytes)
{
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(sm) + byteoff + cbytes);
}
__device__ __forceinline__ void lds_st(double *sm, int byteoff, int cbytes, double v)
{
    *reinterpret_cast<double *>(reinterpret_cast<char *>(sm) + byteoff + cbytes) = v;
}

// LDS-only synchronisation of the threads of one instance.  __syncthreads() also drains vmcnt, i.e. it waits for every
// global load/store in flight — which would turn the stage-ahead prefetches of the sweeps into synchronous loads.  One
// wave executes its LDS instructions in order, so a single-wave instance only needs the LDS counter and a compiler fence;
// multi-wave instances add the hardware barrier.
template <int TPB> __device__ __forceinline__ void lds_sync()
{
    // release/acquire fences restricted to the LDS address space ("local"): s_waitcnt lgkmcnt(0) only, vmcnt untouched
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if constexpr (TPB > 64) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// reciprocal of a pivot: v_rcp_f64 seed + Newton steps.  NMPC_RCP_STEPS = 1 (round 4): the seed is good to about single precision, one step
// leaves ~1e-14 relative — the factorisation of an interior-point step does not need more (iteration counts of the bench batches equal to
// 1e-5 relative, parity tests unchanged) and the two dependent multiply-adds it saves sit on the pivot chain of every stage: A/B in one
// session, six robots B=4096 +2.3 %, B=16384 +1.2 %.  2 = full fp64 accuracy (rounds 2-3).
#ifndef NMPC_RCP_STEPS
#define NMPC_RCP_STEPS 1
#endif
__device__ __forceinline__ double rcp_nr(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
#if NMPC_RCP_STEPS >= 2
    r = fma(fma(-d, r, 1.0), r, r);
#endif
    return r;
}

template <int M_> __device__ __forceinline__ int pidx(int a, int b) { return a * (2 * M_ - a - 1) / 2 + (b - a - 1); }

// flat index of the unordered pair {a, b}, a != b
template <int M_> __device__ __forceinline__ int pidx_any(int a, int b) { const int lo = a < b ? a : b, hi = a < b ? b : a; return pidx<M_>(lo, hi); }

// Inverse of pidx: pair (i < j) of flat index q (lexicographic order 12, 13, .., 1m, 23, .. of C6:288-306), without a loop.  The obvious
// `while (q >= M-1-i) { q -= M-1-i; i++; }` compiles to a DIVERGENT loop (11 instructions per trip, every lane waits for the lane with the
// largest i): ~60 issue slots per call, 25 calls per interior-point iteration of six robots.  Up to six robots (<= 15 pairs) i and j are
// 4-bit fields of two 64-bit literals; beyond, i counts the row starts c_r = r (2M-1-r)/2 that q has passed.
template <int M_> struct PairTab {
    static constexpr int NP = M_ * (M_ - 1) / 2;
    static constexpr unsigned long long tab(bool want_j)
    {
        unsigned long long t = 0;
        int q = 0;
        for (int i = 0; i < M_ && q < 16; i++)
            for (int j = i + 1; j < M_ && q < 16; j++, q++) t |= (unsigned long long)(want_j ? j : i) << (4 * q);
        return t;
    }
};
template <int M_> __host__ __device__ __forceinline__ constexpr void pair_of(int q, int &i, int &j)
{
    if constexpr (M_ * (M_ - 1) / 2 <= 16) {
        constexpr unsigned long long TI = PairTab<M_>::tab(false), TJ = PairTab<M_>::tab(true);
        const int sh = 4 * q;
        i = (int)((TI >> sh) & 15ull); j = (int)((TJ >> sh) & 15ull);
    } else {
        int r = 0;
#pragma unroll
        for (int s = 1; s < M_ - 1; s++) r += (q >= s * (2 * M_ - 1 - s) / 2) ? 1 : 0;
        i = r; j = q - r * (2 * M_ - 1 - r) / 2 + r + 1;
    }
}
// compile-time check of pair_of against the definition (the lexicographic enumeration) for every team size the kernels are instantiated for
template <int M_> constexpr bool pair_of_ok()
{
    int q = 0;
    for (int i = 0; i < M_; i++)
        for (int j = i + 1; j < M_; j++, q++) {
            int a = -1, b = -1;
            pair_of<M_>(q, a, b);
            if (a != i || b != j || a * (2 * M_ - a - 1) / 2 + (b - a - 1) != q) return false;
        }
    return q == M_ * (M_ - 1) / 2;
}
static_assert(pair_of_ok<1>() && pair_of_ok<2>() && pair_of_ok<3>() && pair_of_ok<4>() && pair_of_ok<5>() && pair_of_ok<6>() && pair_of_ok<7>() &&
              pair_of_ok<8>() && pair_of_ok<9>() && pair_of_ok<10>(), "pair_of is the inverse of pidx");

// A value every lane of the wave holds identically (a finished reduction, an LDS word read at a wave-uniform address),
// moved through v_readfirstlane: the compiler then KNOWS it is wave-uniform, so the branches that depend on it (line-search
// acceptance, convergence, pivot failure, barrier update) become scalar branches instead of exec-masked divergent regions.
__device__ __forceinline__ double uniform_f64(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

template <int TPB> __device__ __forceinline__ double wsum(double v, double *red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if constexpr (TPB > 64) {
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        double t = 0.0;
        for (int w = 0; w < TPB / 64; w++) t += red[w];
        v = t;
    }
    return uniform_f64(v);
}
template <int TPB> __device__ __forceinline__ double wmax(double v, double *red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    if constexpr (TPB > 64) {
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        double t = red[0];
        for (int w = 1; w < TPB / 64; w++) t = fmax(t, red[w]);
        v = t;
    }
    return uniform_f64(v);
}
template <int TPB> __device__ __forceinline__ double wmin(double v, double *red) { return -wmax<TPB>(-v, red); }

// Sum of log(s_i), s_i > 0, accumulated as (product of the mantissas, sum of the exponents): four instructions per term instead of a
// logarithm (~90), one logarithm per wavefront at the end.  The mantissa product is renormalised after every term (no underflow however
// many terms); a term <= 0 or NaN makes the result NaN, as the logarithm would.
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
template <int TPB> __device__ __forceinline__ double wlogsum(LogSum a, double *red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double t = a.m * __shfl_xor(a.m, o);
        a.e += __shfl_xor(a.e, o) + __builtin_amdgcn_frexp_exp(t);
        a.m = __builtin_amdgcn_frexp_mant(t);
        a.lo = fmin(a.lo, __shfl_xor(a.lo, o));
    }
    double v = (a.lo > 0.0) ? fma((double)a.e, 0.693147180559945309417, log(a.m)) : NAN;
    if constexpr (TPB > 64) {
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        double t = 0.0;
        for (int w = 0; w < TPB / 64; w++) t += red[w];
        v = t;
    }
    return uniform_f64(v);
}

#ifdef NMPC_PROFILE
#define PROF_T(i)                                                                                                                 \
    do {                                                                                                                          \
        long long _t = clock64();                                                                                                 \
        prof[i] += _t - tlast;                                                                                                    \
        tlast = _t;                                                                                                               \
    } while (0)
#else
#define PROF_T(i) do { } while (0)
#endif

}  // namespace nmpc
