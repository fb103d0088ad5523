// mfma_f64_probe.hip — development measurement: is v_mfma_f64_16x16x4_f64 a faster way to do the trailing update of one
// Riccati stage,  P(18x18 | 19 with the right-hand side) -= Y^T Y  with K = NU = 12 (six robots), than the fp64 VALU?
//
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_probe tools/mfma_f64_probe.hip && ./mfma_f64_probe
//
// One wavefront per SIMD (and optionally two), REP dependent updates each (stage k+1 needs stage k), s_memtime around the loop.
//   valu : the column-per-lane form of nmpc_solve_col.hip — 19 rows x 12 pivots = 228 v_fma_f64 on 19 accumulators, multipliers in VGPRs
//   mfma : 2 x 2 output tiles of 16 x 16 (19 of 32 rows / columns used) x 3 k-steps = 12 v_mfma_f64_16x16x4_f64, operands already in
//          MFMA lane order (no staging) — the best case for the matrix core
//   mfma+staging : the same with the operand shuffle a column-per-lane producer needs (ds_bpermute of the pivot rows into the
//          A[row = lane & 15][k = lane >> 4] / B[k][col] order), which is what the kernel would really pay
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_valu(double *out, long long *cyc, int rep)
{
    double acc[19], y = 1.0 + 1e-9 * threadIdx.x, mul[12];
    for (int r = 0; r < 19; r++) acc[r] = 0.001 * (r + threadIdx.x);
    for (int j = 0; j < 12; j++) mul[j] = 1e-3 * (j + 1);
    long long t0 = clock64();
    for (int it = 0; it < rep; it++) {
#pragma unroll
        for (int j = 0; j < 12; j++) {
#pragma unroll
            for (int r = 0; r < 19; r++) acc[r] = fma(-mul[j], y, acc[r]);
            y = acc[j] * 1e-3 + 1.0;      // next pivot row depends on this step
        }
    }
    long long t1 = clock64();
    double s = 0; for (int r = 0; r < 19; r++) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void k_mfma(double *out, long long *cyc, int rep, int staging)
{
    d4 c[4];
    for (int t = 0; t < 4; t++) c[t] = (d4){0.001 * threadIdx.x, 0.002, 0.003, 0.004};
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    const int src = 4 * (((threadIdx.x & 15) * 3 + (threadIdx.x >> 4)) & 63);
    long long t0 = clock64();
    for (int it = 0; it < rep; it++) {
#pragma unroll
        for (int ks = 0; ks < 3; ks++) {
            double a0 = a, a1 = a * 1.0000001, b0 = b, b1 = b * 0.9999999;
            if (staging) {      // pivot rows of a column-per-lane producer -> MFMA operand order
                a0 = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(a0)), __builtin_amdgcn_ds_bpermute(src, __double2loint(a0)));
                a1 = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(a1)), __builtin_amdgcn_ds_bpermute(src, __double2loint(a1)));
                b0 = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(b0)), __builtin_amdgcn_ds_bpermute(src, __double2loint(b0)));
                b1 = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(b1)), __builtin_amdgcn_ds_bpermute(src, __double2loint(b1)));
            }
            c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c[0], 0, 0, 0);
            c[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, c[1], 0, 0, 0);
            c[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, c[2], 0, 0, 0);
            c[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c[3], 0, 0, 0);
        }
        a = c[0][0] * 1e-6 + 1.0; b = c[3][1] * 1e-6 + 1.0;      // the next stage depends on this one
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    const int rep = 2000;
    double *out; long long *cyc;
    hipMalloc(&out, sizeof(double) * 64 * 2048); hipMalloc(&cyc, sizeof(long long) * 2048);
    long long h[2048];
    for (int waves_per_simd = 1; waves_per_simd <= 2; waves_per_simd++) {
        const int blocks = 256 * 4 * waves_per_simd;
        for (int v = 0; v < 3; v++) {
            for (int w = 0; w < 2; w++) {
                if (v == 0) hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(64), 0, 0, out, cyc, rep);
                else hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(64), 0, 0, out, cyc, rep, v == 2);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < blocks; i++) s += (double)h[i];
            const char *nm = v == 0 ? "valu (228 v_fma_f64)      " : (v == 1 ? "mfma (12 x 16x16x4)       " : "mfma + operand staging    ");
            printf("%d wave(s)/SIMD  %s %8.1f cycles per trailing update  (useful flop 8664; executed %d)\n", waves_per_simd, nm, s / blocks / rep, v == 0 ? 228 * 2 * 64 : 12 * 2048);
        }
    }
    return 0;
}
