// nmpc_api.cpp — the C ABI of include/nmpc.h on top of the gfx950 kernels.
//
// There is deliberately no CPU fallback in this library: every entry point either runs the
// HIP kernels or returns an error code.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nmpc_device.h"
#include "../../include/nmpc_debug.h"

struct nmpc_handle {
    nmpc_config_t cfg;
    nmpc::KParams P;
    int32_t max_batch;
    double *ws;          // device workspace: max_batch * stride doubles
    int64_t ws_bytes;
    int kernel;          // 3 = column-per-lane LDS-resident kernel (default), 2 = element-per-lane LDS-resident kernel, 1 = HBM-resident workgroup kernel, 4 = 3 pinned to its latency shape
    long long *prof;     // device counters of the NMPC_PROFILE build (12 x int64), else unused
    int device;          // device the workspace lives on; made current for the duration of every call
    bool lat_ok;         // the element-per-lane kernel's latency shapes fit the LDS for this configuration
    int col_lat;         // column kernel shape pin: 1 latency, -1 throughput, 0 by batch size
    bool col_lat_ok;     // the column kernel's latency shape (two wavefronts per instance, duals in LDS up to six robots) fits the LDS
    int32_t lat_slots;   // instances of that shape the device holds at once
    int32_t lat_slots4;  // the same for four wavefronts per instance (2 per CU at 256 VGPRs per wave)
    int lat_waves;       // 0: two or four wavefronts by the rule of lat_waves_shape(); 2 / 4: pinned (nmpc_options_t.kernel = 4 / 5)
    int32_t *ord_chk;    // [max_batch + 1] permutation check of a dispatch-order hint: counts, then the "bad" flag
    int32_t *it_buf;     // [max_batch] iteration counts of nmpc_step_batch when the caller passes iters == NULL
    int32_t *st_buf;     // [max_batch] statuses of nmpc_step_batch when the caller passes status == NULL (the in-place update skips failed instances)
};

// makes the handle's device current for one call and restores the caller's on return
struct DeviceScope {
    int prev = -1; bool ok = true, switched = false;
    explicit DeviceScope(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev) { ok = hipSetDevice(dev) == hipSuccess; switched = ok; }
    }
    ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
};

static bool m_supported(int m) { return m >= 1 && m <= NMPC_MAX_ROBOTS; }      // every team size 1..10 is instantiated (7 and 9 since round 4)

extern "C" {

int32_t nmpc_n_var(const nmpc_config_t *c) { return c ? 3 * c->m * (c->N + 1) + 2 * c->m * c->N : NMPC_E_ARG; }
int32_t nmpc_n_p(const nmpc_config_t *c) { return c ? 6 * c->m : NMPC_E_ARG; }
int32_t nmpc_n_g(const nmpc_config_t *c)
{
    if (!c) return NMPC_E_ARG;
    int M = c->pair_rows ? c->m * (c->m - 1) / 2 : 0;
    return 3 * c->m + (c->pad_rows ? M : 0) + (3 * c->m + M + c->m * c->n_obs) * c->N;
}

void nmpc_config_default(nmpc_config_t *c, int32_t m, int32_t N)
{
    memset(c, 0, sizeof(*c));
    c->m = m; c->N = N; c->n_obs = 0; c->pad_rows = m > 1;
    c->T = 0.05; c->dmin = 0.15;
    c->q[0] = 1.0; c->q[1] = 5.0; c->q[2] = 0.1;
    c->r[0] = 0.5; c->r[1] = 0.05;
    c->v_max = 0.22; c->w_max = 2.84; c->xy_max = 10.0; c->th_max = INFINITY;
    c->rob_dim = 0.2; c->margin = 0.1; c->pad_value = 3.5;
    c->tol = 1e-8; c->mu_init = 0.5; c->max_iter = 2000; c->pair_rows = 1;
}

static int fill_params(const nmpc_config_t *c, nmpc::KParams *P)
{
    memset(P, 0, sizeof(*P));
    const int m = c->m, N = c->N, nx = 3 * m, nu = 2 * m, M = c->pair_rows ? m * (m - 1) / 2 : 0, K = c->n_obs;
    P->m = m; P->N = N; P->K = K; P->pairs = (c->pair_rows && m > 1) ? 1 : 0;
    P->thb = isfinite(c->th_max) ? 1 : 0;
    P->nxb = m * (P->thb ? 3 : 2);
    P->nh = 2 * nu + 2 * P->nxb + M + m * K;
    P->o_ul = 0; P->o_uu = nu; P->o_xl = 2 * nu; P->o_xu = P->o_xl + P->nxb; P->o_pr = P->o_xu + P->nxb; P->o_ob = P->o_pr + M;
    P->n_ineq = N * 2 * nu + N * 2 * P->nxb + (N - 1) * (M + m * K);
    P->nvar = nmpc_n_var(c); P->ng = nmpc_n_g(c);
    P->rows0 = nx + (c->pad_rows ? M : 0); P->rowsk = nx + M + m * K;
    P->pad_rows = c->pad_rows; P->pad_value = c->pad_value; P->max_iter = c->max_iter;
    P->T = c->T; P->dmin2 = c->dmin * c->dmin; P->vmax = c->v_max; P->wmax = c->w_max; P->xymax = c->xy_max; P->thmax = c->th_max;
    P->robdim = c->rob_dim; P->margin = c->margin; P->tol = c->tol; P->mu_init = c->mu_init;
    P->rho_el = NMPC_ELASTIC_RHO;
    for (int d = 0; d < 3; d++) P->q[d] = c->q[d];
    for (int d = 0; d < 2; d++) P->r[d] = c->r[d];
    memcpy(P->obs, c->obs, sizeof(P->obs));
    // workspace carve-up; every array starts on a 128-byte boundary
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t at = o; o += (n + 15) / 16 * 16; return (int32_t)at; };
    const int64_t nX = (int64_t)(N + 1) * nx, nU = (int64_t)N * nu, nH = (int64_t)(N + 1) * P->nh;
    P->oX = take(nX); P->oU = take(nU); P->oLAM = take(nX); P->oS = take(nH); P->oZ = take(nH);
    P->oDX = take(nX); P->oDU = take(nU); P->oLAMN = take(nX); P->oDS = take(nH); P->oDZ = take(nH);
    P->oSN = take((int64_t)N * m); P->oCS = take((int64_t)N * m); P->oC = take(nX); P->oH = take(nH); P->oGX = take(nX);
    P->oHUU = take(nU); P->oGU = take(nU); P->oHVT = take((int64_t)N * m); P->oHTT = take((int64_t)N * m);
    P->oKG = take((int64_t)N * nu * nx); P->oKFF = take(nU);
    P->oEL = take(2 * nH);  // elastic variables of the HBM-resident kernel and their steps (one per inequality slot; only the pair / obstacle slots are used)
    P->oCKP = take((int64_t)((N - 1) / NMPC_CKPT_EVERY + 1) * (nx * nx + nx));      // saved cost-to-go of the backward sweep (partial re-factorisation)
    P->stride = o;
    return 0;
}

int32_t nmpc_create(const nmpc_config_t *cfg, int32_t max_batch, nmpc_handle_t **out) { return nmpc_create_opts(cfg, max_batch, nullptr, out); }

int32_t nmpc_create_opts(const nmpc_config_t *cfg, int32_t max_batch, const nmpc_options_t *opts, nmpc_handle_t **out)
{
    if (!cfg || !out || max_batch < 1) return NMPC_E_ARG;
    if (opts && (opts->kernel < 0 || opts->kernel > 5)) return NMPC_E_ARG;
    if (cfg->N < 2 || cfg->N > 4096 || cfg->n_obs < 0 || cfg->n_obs > NMPC_MAX_OBSTACLES) return NMPC_E_ARG;
    if (!(cfg->T > 0.0) || !(cfg->v_max > 0.0) || !(cfg->w_max > 0.0) || !(cfg->xy_max > 0.0) || !(cfg->th_max > 0.0)) return NMPC_E_ARG;
    if (!(cfg->tol > 0.0) || !(cfg->mu_init > 0.0) || cfg->max_iter < 0) return NMPC_E_ARG;
    // weights: Q >= 0, R > 0 (a non-positive R makes every Quu indefinite: it would only surface as status 2 per instance);
    // distances and margins are lengths
    if (!(cfg->q[0] >= 0.0) || !(cfg->q[1] >= 0.0) || !(cfg->q[2] >= 0.0) || !(cfg->r[0] > 0.0) || !(cfg->r[1] > 0.0)) return NMPC_E_ARG;
    if (!(cfg->dmin >= 0.0) || (cfg->n_obs > 0 && (!(cfg->rob_dim >= 0.0) || !(cfg->margin >= 0.0)))) return NMPC_E_ARG;
    for (int o = 0; o < cfg->n_obs; o++) if (!(cfg->obs[3 * o + 2] >= 0.0)) return NMPC_E_ARG;
    if (cfg->pad_rows && !cfg->pair_rows) return NMPC_E_ARG;      // the padding rows only exist next to pair rows (C6:278)
    if (!m_supported(cfg->m)) return NMPC_E_UNSUPPORTED;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return NMPC_E_HIP;   // fail loudly: no CPU path exists
    nmpc_handle *h = (nmpc_handle *)calloc(1, sizeof(nmpc_handle));
    if (!h) return NMPC_E_NOMEM;
    if (hipGetDevice(&h->device) != hipSuccess) { free(h); return NMPC_E_HIP; }
    h->cfg = *cfg;
    fill_params(cfg, &h->P);
    h->max_batch = max_batch;
    const int pin = opts ? opts->kernel : 0;      // 0: chosen per batch size (kernel_for_batch)
    h->kernel = pin ? (pin >= 4 ? 3 : pin) : 3;
    h->col_lat = pin >= 4 ? 1 : (pin == 3 ? -1 : 0);      // 1: latency shape always, -1: throughput shape always, 0: by batch size
    h->lat_waves = pin == 4 ? 2 : (pin == 5 ? 4 : 0);
    // horizons whose iterate does not fit the 160 KB of LDS of a CU run on the HBM-resident kernel (same algorithm, slower)
    if (h->kernel == 3 && nmpc::col_kernel_bytes(h->P, cfg->m, 0) > (size_t)160 * 1024) h->kernel = 2;
    {
        const size_t lb = nmpc::col_kernel_bytes(h->P, cfg->m, 1);
        h->col_lat_ok = lb > 0 && lb <= (size_t)160 * 1024;
        int cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        const int by_regs = cfg->m <= 6 ? 4 : 2, by_lds = h->col_lat_ok ? (int)((size_t)160 * 1024 / lb) : 0;
        h->lat_slots = cus * (by_lds < by_regs ? by_lds : by_regs);
        h->lat_slots4 = cus * (by_lds < 2 ? by_lds : 2);
    }
    if (h->kernel == 2 && nmpc::lds_kernel_bytes(h->P, cfg->m) > (size_t)160 * 1024) h->kernel = 1;
    h->lat_ok = pin != 3 && pin < 4 && nmpc::lds_kernel_bytes(h->P, cfg->m) <= (size_t)160 * 1024;      // NMPC_KERNEL=3 pins the column kernel for every batch size
    nmpc::lds_kernel_workspace(h->P, cfg->m, &h->P.oPACK, &h->P.oKT, &h->P.stride2);
    {   // slack / dual arrays of the column kernel: pair, obstacle, control-bound (slacks + duals) and state-bound (duals) rows
        const int64_t N = cfg->N, N1 = N + 1, m = cfg->m, NPd = m * (m - 1) / 2, MK = m * cfg->n_obs, NU = 2 * m, NXB = h->P.nxb;
        const int64_t dual = 2 * N1 * NPd + 2 * N1 * MK + 4 * N * NU + 2 * N1 * NXB;
        h->P.oDUAL = h->P.stride2;
        h->P.stride2 += (dual + 15) / 16 * 16;
    }
    {   // cost-to-go saved every NMPC_CKPT_EVERY stages of the backward sweep (partial re-factorisation after a rejected pivot): the column
        // kernel stores its registers lane by lane, (3m + 1) x 64 doubles per slot; the element-per-lane kernel its [P | p] (smaller)
        const int64_t slots = (cfg->N - 1) / NMPC_CKPT_EVERY + 1;
        h->P.oCKPT = h->P.stride2;
        h->P.stride2 += slots * (3 * cfg->m + 1) * 64;
    }
    {   // elastic variables t of the pair and obstacle rows (elastic phase only)
        const int64_t N1 = cfg->N + 1, m = cfg->m, el = N1 * (m * (m - 1) / 2 + m * cfg->n_obs);
        h->P.oELAS = h->P.stride2;
        h->P.stride2 += (el + 15) / 16 * 16;
    }
    int64_t per = h->kernel == 1 ? h->P.stride : h->P.stride2;
    h->ws_bytes = (int64_t)sizeof(double) * per * max_batch;
    if (hipMalloc((void **)&h->ws, (size_t)h->ws_bytes) != hipSuccess) { free(h); return NMPC_E_NOMEM; }
    if (hipMalloc((void **)&h->prof, (12 + 24 * 2048) * sizeof(long long)) != hipSuccess) { (void)hipFree(h->ws); free(h); return NMPC_E_NOMEM; }
    (void)hipMemset(h->prof, 0, (12 + 24 * 2048) * sizeof(long long));
    if (hipMalloc((void **)&h->ord_chk, sizeof(int32_t) * ((size_t)max_batch + 1)) != hipSuccess) { (void)hipFree(h->ws); (void)hipFree(h->prof); free(h); return NMPC_E_NOMEM; }
    (void)hipMemset(h->ord_chk, 0, sizeof(int32_t) * ((size_t)max_batch + 1));
    if (hipMalloc((void **)&h->it_buf, sizeof(int32_t) * (size_t)max_batch) != hipSuccess) { (void)hipFree(h->ws); (void)hipFree(h->prof); (void)hipFree(h->ord_chk); free(h); return NMPC_E_NOMEM; }
    if (hipMalloc((void **)&h->st_buf, sizeof(int32_t) * (size_t)max_batch) != hipSuccess) { (void)hipFree(h->ws); (void)hipFree(h->prof); (void)hipFree(h->ord_chk); (void)hipFree(h->it_buf); free(h); return NMPC_E_NOMEM; }
    h->P.trace_inst = opts ? opts->trace_instance : -1;
    *out = h;
    return NMPC_OK;
}

int32_t nmpc_destroy(nmpc_handle_t *h)
{
    if (!h) return NMPC_E_ARG;
    if (h->ws) (void)hipFree(h->ws);
    if (h->prof) (void)hipFree(h->prof);
    if (h->ord_chk) (void)hipFree(h->ord_chk);
    if (h->it_buf) (void)hipFree(h->it_buf);
    if (h->st_buf) (void)hipFree(h->st_buf);
    free(h);
    return NMPC_OK;
}

int64_t nmpc_workspace_bytes(const nmpc_handle_t *h) { return h ? h->ws_bytes : 0; }

// which solve kernel a batch of B instances runs on: 1 HBM-resident, 2 element-per-lane (latency shapes), 3 column-per-lane (throughput)
static int kernel_for_batch(const nmpc_handle_t *h, int32_t B, bool ordered = false)
{
    int kern = h->kernel;
    // With a dispatch-order hint (long solves first: the receding-horizon loop, nmpc_step_batch) the latency shape pays up to four rounds of
    // instances — measured, six robots B=4096: sorted longest-first 15.5 ms against 18.4 ms in the throughput shape; warm closed loop 281 k
    // against 248 k solves/s — because the order keeps a long solve from starting in the last round
    if (ordered && kern == 3 && h->col_lat == 0 && h->col_lat_ok && h->cfg.m >= 4 && B <= 4 * h->lat_slots) return 4;
    if (kern == 3 && h->col_lat == 1) return h->col_lat_ok ? 4 : 3;      // pinned to the column kernel's latency shape
    if (kern == 3 && h->col_lat == 0 && h->col_lat_ok && h->cfg.m >= 4) {
        // The column kernel's latency shape (two wavefronts per instance) where the launch lasts as long as its longest solve.  It holds
        // lat_slots instances at once (4 per CU up to six robots: 256 VGPRs per wave; 2 per CU beyond; fewer when the LDS says so: the
        // composite's 63 KB with its duals make it 2 per CU = 512).
        // Measured, solves/s (throughput | latency shape): six robots B=512 34.6 k | 42.4 k, 1024 67 k | 83 k, 2048 122 k | 151 k, 4096 188 k | 174 k;
        // composite 512 13.2 k | 18.6 k, 1024 25.4 k | 28.7 k, 2048 43.4 k | 41.2 k; ten robots N=20 512 13.5 k | 16.0 k, 1024 26.5 k | 31.5 k,
        // 2048 49.5 k | 41.6 k; N=30 512 5.9 k | 6.9 k, 1024 11.2 k | 11.2 k.  Up to three robots a phase has no more than 64 items: nothing
        // for a second wave to do (two robots 1024: 457 k | 462 k).
        if (B <= 2 * h->lat_slots) return 4;      // every measured crossover lies between two and four rounds of instances
    }
    if (kern == 3 && h->col_lat == 0 && !h->col_lat_ok && h->lat_ok && ((h->cfg.m >= 8 && B <= 256) || (h->cfg.m >= 5 && h->cfg.m <= 6 && B <= 512))) kern = 2;
    return kern;
}

// latency shape: two wavefronts per instance (1), or four (2) for five / six robots whose stage-parallel phases have more than 1000 items
// (obstacle rows: (N-1) m K) while the batch fits twice the instances that shape holds at once (2 per CU)
static int lat_waves_shape(const nmpc_handle_t *h, int32_t B)
{
    if (h->lat_waves == 2 || h->lat_waves == 4) return h->lat_waves == 4 ? 2 : 1;
    const int64_t items = (int64_t)(h->cfg.N - 1) * h->cfg.m * h->cfg.n_obs;
    return (h->cfg.m >= 5 && h->cfg.m <= 6 && items > 1000 && B <= 2 * h->lat_slots4) ? 2 : 1;
}

static int32_t solve_impl(nmpc_handle_t *h, int32_t B, const double *p, const double *w0, double *w_out, double *obj, int32_t *status,
                          int32_t *iters, double *kkt, const int32_t *order, void *stream)
{
    if (!h || B < 0 || B > h->max_batch) return NMPC_E_ARG;
    if (B == 0) return NMPC_OK;      /* empty batch: nothing to read or write, pointers may be null */
    if (!p || !w0 || !w_out) return NMPC_E_ARG;
    DeviceScope dev(h->device);
    if (!dev.ok) return NMPC_E_HIP;
    nmpc::KParams P = h->P;
    P.order = order;
    P.order_bad = h->ord_chk + B;
    if (order && nmpc::launch_order_check(B, order, h->ord_chk, h->ord_chk + B, (hipStream_t)stream) != hipSuccess) return NMPC_E_HIP;
    // Launch shape.  The column-per-lane kernel (one wave per instance, two instances per SIMD up to six robots) is the
    // throughput path.  A batch that cannot fill those slots is a latency problem instead (the launch lasts as long as its longest
    // solve): there the element-per-lane kernel, one instance per SIMD and, for still smaller batches, 2-4 waves per instance, is
    // faster.  Measured (solves/s, column | element), six robots: B=256 14.7 k | 16.7 k, 512 28.6 k | 31.0 k, 1024 54.0 k | 46.4 k,
    // 2048 97 k | 89 k, 4096 164 k | 104 k, 8192 205 k | 131 k, 16384 251 k | 145 k; ten robots N=20: B=256 10.4 k | 11.9 k,
    // 512 13.1 k | 13.4 k, 1024 25.5 k | 23.0 k, 2048 41.5 k | 25.5 k, 4096 59 k | 27 k; N=30: B=256 3.4 k | 3.8 k, 512 5.7 k | 4.7 k.
    const int kern = kernel_for_batch(h, B, order != nullptr);
    hipError_t e = (kern == 1)   ? nmpc::launch_solve(P, h->cfg.m, B, p, w0, w_out, obj, status, iters, kkt, h->ws, (hipStream_t)stream)
                   : (kern == 2) ? nmpc::launch_solve_lds(P, h->cfg.m, B, p, w0, w_out, obj, status, iters, kkt, h->ws, h->prof, (hipStream_t)stream)
                                 : nmpc::launch_solve_col(P, h->cfg.m, B, p, w0, w_out, obj, status, iters, kkt, h->ws, h->prof, (hipStream_t)stream, kern == 4 ? lat_waves_shape(h, B) : 0);
    return e == hipSuccess ? NMPC_OK : NMPC_E_HIP;
}

int32_t nmpc_solve_batch(nmpc_handle_t *h, int32_t B, const double *p, const double *w0, double *w_out, double *obj, int32_t *status,
                         int32_t *iters, double *kkt, void *stream)
{
    return solve_impl(h, B, p, w0, w_out, obj, status, iters, kkt, nullptr, stream);
}

int32_t nmpc_solve_batch_ordered(nmpc_handle_t *h, int32_t B, const double *p, const double *w0, double *w_out, double *obj,
                                 int32_t *status, int32_t *iters, double *kkt, const int32_t *order, void *stream)
{
    return solve_impl(h, B, p, w0, w_out, obj, status, iters, kkt, order, stream);
}

int32_t nmpc_step_batch(nmpc_handle_t *h, int32_t B, double *p, double *w, double *w_sol, double *obj, int32_t *status, int32_t *iters, double *kkt,
                        int32_t *order, void *stream)
{
    if (!h || B < 0 || B > h->max_batch) return NMPC_E_ARG;
    if (B == 0) return NMPC_OK;
    if (!p || !w || !w_sol || w == w_sol) return NMPC_E_ARG;
    int32_t *it = iters ? iters : h->it_buf;
    int32_t *stt = status ? status : h->st_buf;
    // 1. the solve, dispatched in the caller's order (checked to be a permutation; ignored otherwise)
    int32_t rc = solve_impl(h, B, p, w, w_sol, obj, stt, it, kkt, order, stream);
    if (rc != NMPC_OK) return rc;
    DeviceScope dev(h->device);
    if (!dev.ok) return NMPC_E_HIP;
    // 2. guess <- shift(solution) in place of the old guess; x0 <- x0 + T f(x0, u_0) in place of the x0 half of p.  An instance whose solve
    //    ended with status 2 (numerical failure: the iterate may be non-finite) or 3 (infeasible x0) keeps its guess and its x0;
    //    status 1 / 4 return a finite last iterate and are shifted like a converged one, as the scripts do with any IPOPT return
    if (nmpc::launch_shift(h->P, h->cfg.m, B, p, w_sol, w, p, 2 * 3 * h->cfg.m, stt, (hipStream_t)stream) != hipSuccess) return NMPC_E_HIP;
    // 3. the next period's dispatch order: longest solves of this period first
    if (order && nmpc::launch_order_by_iters(B, it, order, (hipStream_t)stream) != hipSuccess) return NMPC_E_HIP;
    return NMPC_OK;
}

int32_t nmpc_eval_batch(nmpc_handle_t *h, int32_t B, const double *p, const double *w, double *f, double *g, void *stream)
{
    if (!h || B < 0) return NMPC_E_ARG;
    if (B == 0) return NMPC_OK;
    if (!p || !w) return NMPC_E_ARG;
    DeviceScope dev(h->device);
    if (!dev.ok) return NMPC_E_HIP;
    hipError_t e = nmpc::launch_eval(h->P, h->cfg.m, B, p, w, f, g, (hipStream_t)stream);
    return e == hipSuccess ? NMPC_OK : NMPC_E_HIP;
}

int32_t nmpc_shift_batch(nmpc_handle_t *h, int32_t B, const double *p_in, const double *w_in, double *w_next, double *x0_next, void *stream)
{
    if (!h || B < 0) return NMPC_E_ARG;
    if (B == 0) return NMPC_OK;
    if (!w_in || !w_next || w_in == w_next) return NMPC_E_ARG;
    if (x0_next && !p_in) return NMPC_E_ARG;
    DeviceScope dev(h->device);
    if (!dev.ok) return NMPC_E_HIP;
    hipError_t e = nmpc::launch_shift(h->P, h->cfg.m, B, p_in, w_in, w_next, x0_next, 0, nullptr, (hipStream_t)stream);
    return e == hipSuccess ? NMPC_OK : NMPC_E_HIP;
}

int32_t nmpc_odometry_batch(int64_t n, const double *odom, const double *init, double *pose, int32_t wrap_2pi, void *stream)
{
    if (n < 0) return NMPC_E_ARG;
    if (n == 0) return NMPC_OK;
    if (!odom || !init || !pose) return NMPC_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return NMPC_E_HIP;
    return nmpc::launch_odometry((long)n, odom, init, pose, wrap_2pi != 0, (hipStream_t)stream) == hipSuccess ? NMPC_OK : NMPC_E_HIP;
}

int64_t nmpc_query(const nmpc_handle_t *h, int32_t what, int64_t arg)
{
    if (!h) return NMPC_E_ARG;
    switch (what) {
    case NMPC_QUERY_KERNEL_FOR_BATCH: return (arg < 0 || arg > h->max_batch) ? NMPC_E_ARG : kernel_for_batch(h, (int32_t)arg);
    case NMPC_QUERY_KERNEL_FOR_ORDERED_BATCH: return (arg < 0 || arg > h->max_batch) ? NMPC_E_ARG : kernel_for_batch(h, (int32_t)arg, true);
    case NMPC_QUERY_WORKSPACE_BYTES: return h->ws_bytes;
    case NMPC_QUERY_LDS_BYTES: {
        if (arg < 0 || arg > h->max_batch) return NMPC_E_ARG;
        const int k = kernel_for_batch(h, (int32_t)arg);
        return (k == 3 || k == 4) ? (int64_t)nmpc::col_kernel_bytes(h->P, h->cfg.m, k == 4) : (k == 2 ? (int64_t)nmpc::lds_kernel_bytes(h->P, h->cfg.m) : 0);
    }
    case NMPC_QUERY_MAX_BATCH: return h->max_batch;
    default: return NMPC_E_ARG;
    }
}

/* ---- development aids, declared in include/nmpc_debug.h (not part of the drop-in boundary) */
/* per-phase cycle counters of an NMPC_PROFILE build */
int32_t nmpc_debug_profile(nmpc_handle_t *h, int64_t *out12, int32_t reset)
{
    if (!h || !out12) return NMPC_E_ARG;
    if (hipDeviceSynchronize() != hipSuccess) return NMPC_E_HIP;
    if (hipMemcpy(out12, h->prof, 12 * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) return NMPC_E_HIP;
    if (reset) (void)hipMemset(h->prof, 0, 12 * sizeof(long long));
    return NMPC_OK;
}
int32_t nmpc_debug_trace(nmpc_handle_t *h, double *out, int32_t rows)
{
    if (!h || !out || rows < 0 || rows > 2048) return NMPC_E_ARG;
    if (rows == 0) return NMPC_OK;
    if (hipDeviceSynchronize() != hipSuccess) return NMPC_E_HIP;
    if (hipMemcpy(out, h->prof + 12, (size_t)rows * 16 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return NMPC_E_HIP;
    return NMPC_OK;
}
int32_t nmpc_debug_trace2(nmpc_handle_t *h, double *out, int32_t rows)
{
    if (!h || !out || rows < 0 || rows > 2048) return NMPC_E_ARG;
    if (rows == 0) return NMPC_OK;
    if (hipMemcpy(out, h->prof + 12 + 16 * 2048, (size_t)rows * 8 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return NMPC_E_HIP;
    return NMPC_OK;
}

/* copy the per-instance workspace of one instance to the host; returns its length in doubles */
int64_t nmpc_debug_workspace(nmpc_handle_t *h, int32_t inst, double *out, int64_t cap, int64_t *offs)
{
    if (!h) return NMPC_E_ARG;
    int64_t per = h->kernel == 1 ? h->P.stride : h->P.stride2;
    if (offs) { offs[0] = h->kernel; offs[1] = h->P.oKG; offs[2] = h->P.oKFF; offs[3] = h->P.oPACK; offs[4] = h->P.oKT; }
    if (!out) return per;
    if (cap < per || inst < 0 || inst >= h->max_batch) return NMPC_E_ARG;
    if (hipDeviceSynchronize() != hipSuccess) return NMPC_E_HIP;
    if (hipMemcpy(out, h->ws + (size_t)inst * per, (size_t)per * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return NMPC_E_HIP;
    return per;
}

#ifndef NMPC_SRC_HASH
#define NMPC_SRC_HASH "unknown"
#endif
const char *nmpc_version(void) { return "nmpc_hip 0.2 (gfx950, fp64, one wavefront per swarm instance) src=" NMPC_SRC_HASH; }

}  // extern "C"
