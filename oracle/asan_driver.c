/*
 * asan_driver.c — TEST INFRASTRUCTURE.  Runs the two CPU oracles under AddressSanitizer + UBSan on a few fixed problems
 * (GPU sanitizers are not available on this pool; the CPU restatement is what can be checked).  `make -C oracle asan_check`.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/nmpc_lidar.h"

int32_t nmpc_oracle_solve_batch(const nmpc_config_t *, int32_t, const double *, const double *, double *, double *, int32_t *, int32_t *, double *, int32_t);
int32_t nmpc_oracle_eval_batch(const nmpc_config_t *, int32_t, const double *, const double *, double *, double *);
int32_t nmpc_oracle_shift_batch(const nmpc_config_t *, int32_t, const double *, const double *, double *, double *);
int32_t nmpc_lidar_oracle_solve_batch(const nmpc_lidar_config_t *, const double *, const double *, int32_t, const double *, const double *, double *, double *,
                                      int32_t *, int32_t *, double *, int32_t);
int32_t nmpc_lidar_oracle_eval_batch(const nmpc_lidar_config_t *, int32_t, const double *, const double *, double *, double *);

static int swarm(int m, int N, int n_obs)
{
    nmpc_config_t c;
    nmpc_oracle_config_default(&c, m, N);
    c.T = 0.3; c.dmin = 0.4; c.v_max = 0.15; c.w_max = 1.5; c.max_iter = 400; c.n_obs = n_obs;
    if (m == 1) { c.pad_rows = 0; c.th_max = 6.283185307179586; }
    for (int o = 0; o < n_obs; o++) { c.obs[3 * o] = 2.0 + o; c.obs[3 * o + 1] = -2.0; c.obs[3 * o + 2] = 0.15; }
    const int B = 3, nv = nmpc_oracle_n_var(&c), ng = nmpc_oracle_n_g(&c), nx = 3 * m;
    double *p = calloc((size_t)B * 2 * nx, sizeof(double)), *w0 = calloc((size_t)B * nv, sizeof(double)), *w = calloc((size_t)B * nv, sizeof(double));
    double *wn = calloc((size_t)B * nv, sizeof(double)), *x0n = calloc((size_t)B * nx, sizeof(double)), *g = calloc((size_t)B * ng, sizeof(double));
    double obj[3], kkt[3], f[3]; int32_t st[3], it[3];
    for (int b = 0; b < B; b++)
        for (int i = 0; i < m; i++) {
            double a = 6.283185307179586 * i / m + 0.1 * b;
            p[b * 2 * nx + 3 * i] = cos(a); p[b * 2 * nx + 3 * i + 1] = sin(a); p[b * 2 * nx + 3 * i + 2] = a + 3.0;
            p[b * 2 * nx + nx + 3 * i] = 0.6 * cos(a + 0.4); p[b * 2 * nx + nx + 3 * i + 1] = 0.6 * sin(a + 0.4); p[b * 2 * nx + nx + 3 * i + 2] = a + 3.0;
            for (int k = 0; k <= N; k++) memcpy(w0 + (size_t)b * nv + (size_t)k * nx + 3 * i, p + b * 2 * nx + 3 * i, 3 * sizeof(double));
        }
    int rc = nmpc_oracle_solve_batch(&c, B, p, w0, w, obj, st, it, kkt, 1);
    rc |= nmpc_oracle_eval_batch(&c, B, p, w, f, g);
    rc |= nmpc_oracle_shift_batch(&c, B, p, w, wn, x0n);
    int bad = rc != 0;
    for (int b = 0; b < B; b++) bad |= st[b] != 0 || !(fabs(f[b] - obj[b]) <= 1e-9 * fmax(1.0, fabs(obj[b])));
    printf("swarm m=%d N=%d K=%d: status %d %d %d, iters %d %d %d -> %s\n", m, N, n_obs, st[0], st[1], st[2], it[0], it[1], it[2], bad ? "FAIL" : "ok");
    free(p); free(w0); free(w); free(wn); free(x0n); free(g);
    return bad;
}

static int lidar(int N, int Nc, int R)
{
    nmpc_lidar_config_t c = {N, Nc, R, 400, 0.075, {1.0, 5.0, 0.1}, {0.5, 0.05}, 0.1, 1e-8, 0.5};
    const int nv = nmpc_lidar_oracle_n_var(&c), ng = nmpc_lidar_oracle_n_g(&c), np_ = nmpc_lidar_oracle_n_p(&c), ns = 3 + R;
    double *lb = malloc(sizeof(double) * nv), *ub = malloc(sizeof(double) * nv), *p = calloc(np_, sizeof(double)), *w0 = calloc(nv, sizeof(double)), *w = calloc(nv, sizeof(double)),
           *g = calloc(ng, sizeof(double));
    for (int k = 0; k <= N; k++)
        for (int cc = 0; cc < ns; cc++) { lb[k * ns + cc] = cc < 2 ? -10.0 : (cc == 2 ? -INFINITY : 0.15); ub[k * ns + cc] = cc < 2 ? 10.0 : (cc == 2 ? INFINITY : 10.0); }
    for (int j = 0; j < Nc; j++) { lb[(N + 1) * ns + 2 * j] = -0.15; ub[(N + 1) * ns + 2 * j] = 0.15; lb[(N + 1) * ns + 2 * j + 1] = -1.5; ub[(N + 1) * ns + 2 * j + 1] = 1.5; }
    p[2] = 0.2; p[3] = 1.5; p[4] = 0.8;
    for (int m = 0; m < R; m++) { p[6 + m] = 1.0 + 0.5 * m; p[6 + R + m] = 6.283185307179586 * m / R; }
    for (int k = 0; k <= N; k++) { w0[k * ns + 2] = 0.2; for (int m = 0; m < R; m++) w0[k * ns + 3 + m] = p[6 + m]; }
    double obj, kkt, f; int32_t st, it;
    int rc = nmpc_lidar_oracle_solve_batch(&c, lb, ub, 1, p, w0, w, &obj, &st, &it, &kkt, 1);
    rc |= nmpc_lidar_oracle_eval_batch(&c, 1, p, w, &f, g);
    int bad = rc != 0 || st != 0 || !(fabs(f - obj) <= 1e-9 * fmax(1.0, fabs(obj)));
    printf("lidar N=%d Nc=%d R=%d: status %d, iters %d -> %s\n", N, Nc, R, st, it, bad ? "FAIL" : "ok");
    free(lb); free(ub); free(p); free(w0); free(w); free(g);
    return bad;
}

int main(void)
{
    int bad = swarm(1, 6, 2) | swarm(2, 10, 0) | swarm(3, 8, 1) | swarm(6, 8, 0) | lidar(12, 6, 4) | lidar(10, 10, 3);
    printf(bad ? "ASAN_DRIVER_FAIL\n" : "ASAN_DRIVER_OK\n");
    return bad;
}
