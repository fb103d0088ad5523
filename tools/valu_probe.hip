// Development probe (gfx950): issue cost of the instruction forms the solve kernels lean on, in s_memtime ticks per instruction.
//   independent / dependent v_fma_f64, independent / dependent v_fmac_f64_dpp row_newbcast, v_readlane x2 + v_fma_f64 (scalar operand)
//   at 1 and 2 resident waves per SIMD.   hipcc --offload-arch=gfx950 -O2 tools/valu_probe.hip -o tools/bin/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
template <int N_> __device__ __forceinline__ void fmac_rowb(double &acc, double u, double nr)
{
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(u), "v"(nr), "n"(N_));
}
__global__ void probe(double *out, long long *ticks, const double *in, int mode, int iters)
{
    const int t = threadIdx.x;
    double a[16], x = in[t], y = in[64 + t];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = in[128 + t] + i;
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (mode == 0) {          // 16 independent accumulators
#pragma unroll
            for (int r = 0; r < REP / 16; r++)
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
        } else if (mode == 1) {   // one dependent chain
#pragma unroll
            for (int r = 0; r < REP; r++) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[0]) : "v"(x), "v"(y));
        } else if (mode == 2) {   // independent DPP multiply-adds
#pragma unroll
            for (int r = 0; r < REP / 16; r++)
#pragma unroll
                for (int i = 0; i < 16; i++) fmac_rowb<3>(a[i], x, y);
        } else if (mode == 3) {   // dependent DPP multiply-adds (accumulator chain)
#pragma unroll
            for (int r = 0; r < REP; r++) fmac_rowb<3>(a[0], x, y);
        } else if (mode == 4) {   // readlane x2 + fma with the scalar, independent accumulators
#pragma unroll
            for (int r = 0; r < REP / 16; r++)
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    double s = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), i), __builtin_amdgcn_readlane(__double2loint(x), i));
                    asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "s"(s), "v"(y));
                }
        } else if (mode == 5) {   // independent f32 fma (reference: 4-cycle issue)
            float f[16];
#pragma unroll
            for (int i = 0; i < 16; i++) f[i] = (float)a[i];
#pragma unroll
            for (int r = 0; r < REP / 16; r++)
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"((float)x), "v"((float)y));
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = f[i];
        } else if (mode == 6) {   // DPP chain where the DPP operand is the accumulator itself (back substitution form) + s_nop 1
#pragma unroll
            for (int r = 0; r < REP; r++) { asm volatile("s_nop 1"); asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[0]) : "v"(y)); }
        }
    }
    long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i];
    out[blockIdx.x * 64 + t] = s;
    if (t == 0) ticks[blockIdx.x] = t1 - t0;
}
int main()
{
    double h[192], *di, *dout; long long *dt; static long long ht[8192];
    for (int i = 0; i < 192; i++) h[i] = 1.0 + 1e-9 * i;
    hipMalloc(&di, sizeof h); hipMalloc(&dout, 8192 * 64 * 8); hipMalloc(&dt, sizeof ht);
    hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
    const char *names[] = {"v_fma_f64 independent", "v_fma_f64 dependent chain", "v_fmac_f64_dpp independent", "v_fmac_f64_dpp dependent chain",
                           "readlane x2 + v_fma_f64 (sgpr)", "v_fma_f32 independent", "v_fmac_f64_dpp acc=dpp operand + s_nop 1"};
    const int iters = 20000;
    for (int wps = 1; wps <= 8; wps *= 2) {
        // 256 CUs x 4 SIMDs x wps waves; LDS request keeps one block per wave slot: 160 KB / (4 * wps) per block
        const int blocks = 256 * 4 * wps; const size_t lds = 160 * 1024 / (4 * wps) - 1024;
        for (int mode = 0; mode < 7; mode++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            probe<<<blocks, 64, lds>>>(dout, dt, di, mode, iters);
            hipEventRecord(e0);
            probe<<<blocks, 64, lds>>>(dout, dt, di, mode, iters);
            hipEventRecord(e1);
            hipMemcpy(ht, dt, blocks * sizeof(long long), hipMemcpyDeviceToHost);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double avg = 0; for (int b = 0; b < blocks; b++) avg += ht[b]; avg /= blocks;
            const double ninst = (double)iters * REP * (mode == 4 ? 1 : 1);
            printf("waves/SIMD %d  %-42s %7.2f ticks/instr-group   kernel %.3f ms -> %.2f ns per group per wave\n", wps, names[mode], avg / ninst, ms, ms * 1e6 / ninst);
        }
    }
    return 0;
}
