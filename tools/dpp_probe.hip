// Development probe (gfx950): semantics of the two cross-lane forms the column kernel's pivot step uses.
//   1. v_fmac_f64_dpp ... row_newbcast:n  — acc += (lane n of MY row of 16 lanes of u) * nr, one DP instruction
//   2. v_permlane16_swap / v_permlane32_swap — replicate one 16-lane row of a register into the other rows
// Build + run:  hipcc --offload-arch=gfx950 -O2 tools/dpp_probe.hip -o /tmp/dpp_probe && /tmp/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u2v __attribute__((ext_vector_type(2)));

template <int N_> __device__ __forceinline__ void fmac_rowb(double &acc, double u, double nr)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(u), "v"(nr), "n"(N_));
}
__device__ __forceinline__ void swap16(double &a, double &b)      // odd rows of a <-> even rows of b
{
    u2v lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    u2v hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    a = __hiloint2double(hi.x, lo.x); b = __hiloint2double(hi.y, lo.y);
}
__device__ __forceinline__ void swap32(double &a, double &b)      // upper half of a <-> lower half of b
{
    u2v lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    u2v hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    a = __hiloint2double(hi.x, lo.x); b = __hiloint2double(hi.y, lo.y);
}
__global__ void probe(double *o, const double *in)
{
    const int t = threadIdx.x;
    double u = in[t], nr = in[64 + t], acc3 = in[128 + t], acc11 = in[128 + t];
    asm("s_nop 4" : "+v"(u));
    fmac_rowb<3>(acc3, u, nr);
    fmac_rowb<11>(acc11, u, nr);
    o[t] = acc3; o[64 + t] = acc11;
    double a = in[t], b = in[t];
    swap16(a, b);
    o[128 + t] = a; o[192 + t] = b;
    double c = a, d = a;
    swap32(c, d);
    o[256 + t] = c; o[320 + t] = d;
}
int main()
{
    double h[192], r[384], *di, *dout;
    for (int i = 0; i < 64; i++) { h[i] = 100.0 + i; h[64 + i] = 0.5 + i * 0.25; h[128 + i] = 1000.0 * i; }
    hipMalloc(&di, sizeof h); hipMalloc(&dout, sizeof r);
    hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dout, di);
    if (hipMemcpy(r, dout, sizeof r, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAIL: hip error\n"); return 2; }
    int bad = 0;
    for (int t = 0; t < 64; t++) {
        const int row = t / 16;
        if (r[t] != h[128 + t] + h[16 * row + 3] * h[64 + t]) bad++;
        if (r[64 + t] != h[128 + t] + h[16 * row + 11] * h[64 + t]) bad++;
        // swap16(a = v, b = v): a = [v0 v0 v2 v2], b = [v1 v1 v3 v3] (rows of 16)
        if (r[128 + t] != h[16 * (row & 2) + (t & 15)]) bad++;
        if (r[192 + t] != h[16 * ((row & 2) + 1) + (t & 15)]) bad++;
        // swap32(c = a, d = a): c = [v0 v0 v0 v0], d = [v2 v2 v2 v2]
        if (r[256 + t] != h[t & 15]) bad++;
        if (r[320 + t] != h[32 + (t & 15)]) bad++;
    }
    printf("row_newbcast fmac + permlane swaps: %s (%d mismatches)\n", bad ? "MISMATCH" : "as expected", bad);
    if (bad) for (int t = 0; t < 64; t += 5) printf("  t=%2d acc3 %.3f a %.0f b %.0f c %.0f d %.0f\n", t, r[t], r[128 + t], r[192 + t], r[256 + t], r[320 + t]);
    return bad ? 1 : 0;
}
