// Development aid (GPU box): relative error of v_rcp_f64 and of v_rcp_f64 + 1 / 2 Newton steps against an IEEE division (what
// nmpc_solve_common.h: rcp_nr relies on).   hipcc --offload-arch=gfx950 -O2 tools/rcp_probe.hip -o /tmp/rcp_probe && /tmp/rcp_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
__global__ void k(const double *x, double *e, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = x[i], ref = 1.0 / d;
    double r0 = __builtin_amdgcn_rcp(d);
    double r1 = fma(fma(-d, r0, 1.0), r0, r0);
    double r2 = fma(fma(-d, r1, 1.0), r1, r1);
    e[3 * i] = fabs(r0 - ref) / fabs(ref); e[3 * i + 1] = fabs(r1 - ref) / fabs(ref); e[3 * i + 2] = fabs(r2 - ref) / fabs(ref);
}
int main()
{
    const int n = 1 << 22;
    double *hx = (double *)malloc(n * sizeof(double)), *he = (double *)malloc(3 * n * sizeof(double)), *dx, *de;
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        double u = (double)(s >> 11) / 9007199254740992.0;
        int ex = (int)((s >> 3) % 80) - 40;      // magnitudes 1e-12 .. 1e12
        hx[i] = ldexp(1.0 + u, ex) * ((s & 1) ? 1.0 : -1.0);
    }
    hipMalloc(&dx, n * sizeof(double)); hipMalloc(&de, 3 * n * sizeof(double));
    hipMemcpy(dx, hx, n * sizeof(double), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, de, n);
    hipMemcpy(he, de, 3 * n * sizeof(double), hipMemcpyDeviceToHost);
    double m[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) if (he[3 * i + j] > m[j]) m[j] = he[3 * i + j];
    printf("max relative error over %d values: v_rcp_f64 %.3e, + 1 Newton step %.3e, + 2 Newton steps %.3e\n", n, m[0], m[1], m[2]);
    return 0;
}
