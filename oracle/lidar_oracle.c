/*
 * lidar_oracle.c — TEST INFRASTRUCTURE.  CPU fp64 restatement of the LIDAR-ray distance-state NMPC solve of the reference
 * (SURVEY.md 8(f) row 1 / a14).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * library; the product (libnmpc_hip.so) never links or calls it.
 *
 * PARITY UNPINNED against CasADi/IPOPT (neither is installed; the reference holds no vectors).  Pinned by the solver-independent
 * known answers of tests/test_oracle_lidar.py (layout, cold-start values, finite-difference derivatives of oracle/lidar_ref.py),
 * scipy-SLSQP golden triples and the least-squares KKT report.
 *
 * What is restated (V4 = AllScripts/obs_avoid_static_first_scenario_v4.py, V3 = ..._v3.py):
 *   NLP    V4:78-151 — 3 + R states per stage [x y theta d_1..d_R]; controls U[:, min(k, Nc-1)] (move blocking, V4:128-131);
 *          cost V4:135-136 (stage cost + 0.1 sum 1/d^2); rows gx (V4:110,137-140) then gd (V4:111,142-148: d_{k+1} minus the
 *          1-norm distance to the lidar point fixed at stage 0, V4:114-118); packing V4:155; bounds as the caller passes them
 *          (V4:161-176 builds them misaligned with the packing; that array is an input here).
 *   solve  V4:157,245 nlpsol('ipopt') — the same interior-point restatement as oracle/nmpc_oracle.c (see its header: Waechter &
 *          Biegler barrier rule, fraction to the boundary, l1-merit non-monotone search, inertia shift on the control diagonal,
 *          barrier restart), applied to this NLP, plus — this solve only — the line-search watchdog of include/nmpc_constants.h (after
 *          IPOPT's watchdog_shortened_iter_trigger: the 1-norm rows make the l1 merit reject good steps).  The Newton system is solved in the reduced space of the pose: the distance
 *          states are eliminated stage by stage through their own (linearised) equality rows, d = ||p - pObs||_1, which leaves
 *          a 3-state Riccati recursion; stages k >= Nc carry the held control as two extra columns of the cost-to-go.
 *   shift  V4:258-270.
 * X_0 (pose and scan) is pinned to the parameters, as in nmpc_oracle.c.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/nmpc_constants.h"      /* cold-start retry and inertia constants shared with the HIP kernels */
#include "../include/nmpc_lidar.h"

#define RMAX NMPC_LIDAR_MAX_RAYS
#define NS (3 + RMAX)

typedef struct {
    int N, Nc, R, ns, max_iter, max_restarts;
    double T, q[3], r[2], lw, tol, mu_init;
    const double *lb, *ub;             /* [n_var] */
    /* iterate */
    double *V, *U, *lam, *eta;         /* V [N+1][ns], U [Nc][2], lam [N+1][3], eta [N+1][R] */
    double *SL, *ZL, *SU, *ZU;         /* [N+1][ns] slots of the variable bounds of stages >= 1 */
    double *SLu, *ZLu, *SUu, *ZUu;     /* [Nc][2] */
    double *dV, *dU, *lamn, *etan;
    double *Vt, *Ut;
    double *sn, *cs;                   /* [N] */
    double *Hxx, *gx, *Wd, *gdv;       /* [N+1][6] (xx xy yy tt + 2 spare), [N+1][3], [N+1][R], [N+1][R] */
    double *huu, *gu, *hvt;            /* [Nc][2], [Nc][2], [N] */
    double *Kg, *kff;                  /* [Nc][2*3], [Nc][2] */
    double po[RMAX][2], xs[3];
} lw_t;

static int ctrl_of(const lw_t *w, int k) { return k < w->Nc - 1 ? k : w->Nc - 1; }

static lw_t *lw_new(const nmpc_lidar_config_t *c, const double *lb, const double *ub)
{
    lw_t *w = (lw_t *)calloc(1, sizeof(lw_t));
    w->N = c->N; w->Nc = c->Nc; w->R = c->R; w->ns = 3 + c->R; w->T = c->T; w->lw = c->lw; w->tol = c->tol; w->mu_init = c->mu_init;
    w->max_iter = c->max_iter; w->max_restarts = 3;
    for (int i = 0; i < 3; i++) w->q[i] = c->q[i];
    for (int i = 0; i < 2; i++) w->r[i] = c->r[i];
    w->lb = lb; w->ub = ub;
    size_t nV = (size_t)(c->N + 1) * w->ns, nU = (size_t)c->Nc * 2;
#define AL(p, n) w->p = (double *)calloc((n), sizeof(double))
    AL(V, nV); AL(U, nU); AL(lam, (size_t)(c->N + 1) * 3); AL(eta, (size_t)(c->N + 1) * c->R);
    AL(SL, nV); AL(ZL, nV); AL(SU, nV); AL(ZU, nV); AL(SLu, nU); AL(ZLu, nU); AL(SUu, nU); AL(ZUu, nU);
    AL(dV, nV); AL(dU, nU); AL(lamn, (size_t)(c->N + 1) * 3); AL(etan, (size_t)(c->N + 1) * c->R);
    AL(Vt, nV); AL(Ut, nU); AL(sn, c->N); AL(cs, c->N);
    AL(Hxx, (size_t)(c->N + 1) * 6); AL(gx, (size_t)(c->N + 1) * 3); AL(Wd, (size_t)(c->N + 1) * c->R); AL(gdv, (size_t)(c->N + 1) * c->R);
    AL(huu, nU); AL(gu, nU); AL(hvt, c->N); AL(Kg, (size_t)c->Nc * 6); AL(kff, nU);
#undef AL
    return w;
}
static void lw_free(lw_t *w)
{
    double **ps[] = {&w->V, &w->U, &w->lam, &w->eta, &w->SL, &w->ZL, &w->SU, &w->ZU, &w->SLu, &w->ZLu, &w->SUu, &w->ZUu, &w->dV, &w->dU, &w->lamn,
                     &w->etan, &w->Vt, &w->Ut, &w->sn, &w->cs, &w->Hxx, &w->gx, &w->Wd, &w->gdv, &w->huu, &w->gu, &w->hvt, &w->Kg, &w->kff};
    for (size_t i = 0; i < sizeof(ps) / sizeof(ps[0]); i++) free(*ps[i]);
    free(w);
}

/* bounds of variable (stage k, component c) / control (j, e) in the caller's arrays (packing V4:155) */
static inline double LBV(const lw_t *w, int k, int c) { return w->lb[(size_t)k * w->ns + c]; }
static inline double UBV(const lw_t *w, int k, int c) { return w->ub[(size_t)k * w->ns + c]; }
static inline double LBU(const lw_t *w, int j, int e) { return w->lb[(size_t)(w->N + 1) * w->ns + 2 * j + e]; }
static inline double UBU(const lw_t *w, int j, int e) { return w->ub[(size_t)(w->N + 1) * w->ns + 2 * j + e]; }

/* one-norm distance to lidar point m and its (sub)gradient signs */
static inline double gdist(const lw_t *w, int m, double x, double y, double *sx, double *sy)
{
    double ax = x - w->po[m][0], ay = y - w->po[m][1];
    *sx = (ax > 0) - (ax < 0); *sy = (ay > 0) - (ay < 0);
    return fabs(ax) + fabs(ay);
}

/* objective, max |defect|, sum |defect| of (V,U); trig cache of the stages optional */
static double eval_point(const lw_t *w, const double *V, const double *U, double *sn, double *cs, double *th_out, double *ec_out)
{
    const int N = w->N, ns = w->ns, R = w->R;
    double f = 0.0, th = 0.0, ec = 0.0;
    for (int k = 0; k < N; k++) {
        const double *v = V + (size_t)k * ns, *vn = v + ns, *u = U + 2 * ctrl_of(w, k);
        double s = sin(v[2]), c = cos(v[2]);
        if (sn) { sn[k] = s; cs[k] = c; }
        double c0 = vn[0] - (v[0] + w->T * u[0] * c), c1 = vn[1] - (v[1] + w->T * u[0] * s), c2 = vn[2] - (v[2] + w->T * u[1]);
        th += fabs(c0) + fabs(c1) + fabs(c2); ec = fmax(ec, fmax(fabs(c0), fmax(fabs(c1), fabs(c2))));
        for (int i = 0; i < 3; i++) { double e = v[i] - w->xs[i]; f += w->q[i] * e * e; }
        f += w->r[0] * u[0] * u[0] + w->r[1] * u[1] * u[1];
        if (w->lw != 0.0) for (int m = 0; m < R; m++) f += w->lw / (v[3 + m] * v[3 + m]);
        for (int m = 0; m < R; m++) {
            double sx, sy, e = vn[3 + m] - gdist(w, m, vn[0], vn[1], &sx, &sy);
            th += fabs(e); ec = fmax(ec, fabs(e));
        }
    }
    *th_out = th; *ec_out = ec;
    return f;
}

static int solve_one(lw_t *w, const double *p, const double *w0, double *wout, double *obj_out, int *iters_out, double *kkt_out)
{
    const int N = w->N, Nc = w->Nc, R = w->R, ns = w->ns;
    const double T = w->T, bp = 1e-2;
    int status = NMPC_STATUS_MAX_ITER;
    memcpy(w->V, w0, sizeof(double) * (size_t)(N + 1) * ns);
    memcpy(w->U, w0 + (size_t)(N + 1) * ns, sizeof(double) * (size_t)Nc * 2);
    /* X_0 pinned to the parameters: pose P[0:3], scan P[6:6+R] */
    for (int i = 0; i < 3; i++) { w->V[i] = p[i]; w->xs[i] = p[3 + i]; }
    for (int m = 0; m < R; m++) w->V[3 + m] = p[6 + m];
    for (int m = 0; m < R; m++) {       /* V4:114-118 */
        double a = p[2] + p[6 + R + m];
        w->po[m][0] = p[0] + p[6 + m] * cos(a); w->po[m][1] = p[1] + p[6 + m] * sin(a);
    }
    /* the pinned stage-0 variables must respect their own bounds */
    for (int c = 0; c < ns; c++)
        if (w->V[c] < LBV(w, 0, c) || w->V[c] > UBV(w, 0, c)) {
            memcpy(wout, w->V, sizeof(double) * (size_t)(N + 1) * ns);
            memcpy(wout + (size_t)(N + 1) * ns, w->U, sizeof(double) * (size_t)Nc * 2);
            *obj_out = NAN; *iters_out = 0; *kkt_out = INFINITY;
            return NMPC_STATUS_INFEASIBLE_X0;
        }
    /* push the start strictly inside the simple bounds (IPOPT bound_push = bound_frac = 1e-2) */
#define PUSH(val, lo, hi)                                                                                                         \
    do {                                                                                                                          \
        double lo_ = (lo), hi_ = (hi), v_ = (val);                                                                                \
        if (isfinite(lo_) && isfinite(hi_)) { double pu = fmin(bp * fmax(1.0, fabs(lo_)), bp * (hi_ - lo_)), qu = fmin(bp * fmax(1.0, fabs(hi_)), bp * (hi_ - lo_)); v_ = fmin(fmax(v_, lo_ + pu), hi_ - qu); } \
        else if (isfinite(lo_)) v_ = fmax(v_, lo_ + bp * fmax(1.0, fabs(lo_)));                                                    \
        else if (isfinite(hi_)) v_ = fmin(v_, hi_ - bp * fmax(1.0, fabs(hi_)));                                                    \
        (val) = v_;                                                                                                               \
    } while (0)
    int n_ineq = 0;
    for (int k = 1; k <= N; k++) for (int c = 0; c < ns; c++) { n_ineq += isfinite(LBV(w, k, c)) + isfinite(UBV(w, k, c)); }
    n_ineq += 4 * Nc;
    double mu = w->mu_init, f, th0, e_c;
    int it = 0, need_shift = 0, n_tiny = 0, n_restart = 0, restarting = 0;
    /* Cold-start retry, as in oracle/nmpc_oracle.c: an attempt that stalls after its restarts, fails numerically or runs
       NMPC_COLD_RETRY_ITERS iterations without converging is restarted from the reference's own cold start (V4:184-196: X_k = x0
       pose and scan, U = 0), at most twice, the second time with mu = 10 mu_init.  Of 16,384 random V4 instances one crawls for
       2074 iterations at mu_init = 0.5 and converges in 23-42 from any other initial barrier parameter. */
    int n_cold = getenv("NMPC_ORACLE_NO_COLD_RETRY") ? NMPC_COLD_RETRIES : 0, it_base = 0, cold = 0;
    /* watchdog of the line search (include/nmpc_constants.h); NMPC_LIDAR_ORACLE_WD=<trigger> overrides the trigger for experiments and tests (0 = off) */
    const int wd_trig = getenv("NMPC_LIDAR_ORACLE_WD") ? atoi(getenv("NMPC_LIDAR_ORACLE_WD")) : NMPC_WATCHDOG_TRIGGER; int n_short = 0;
    const int trace = getenv("NMPC_LIDAR_ORACLE_TRACE") != NULL;      /* development: one line per iteration on stderr (single instance) */
#define LIDAR_COLD_RETRY() do { cold = 1; n_cold++; it_base = it; restarting = 1; mu = (n_cold == 1) ? w->mu_init : 10.0 * w->mu_init; n_tiny = 0; n_restart = 0; } while (0)
    double delta_last = 0.0, nu_pen = 1.0, kkt = INFINITY;
    double mh0 = 0, mh1 = 0, mh2 = 0, mh_mu = -1, mh_nu = -1; int mcount = 0;

    for (;;) {      /* (re)start of the barrier iteration */
        if (cold) {
            for (int k = 1; k <= N; k++) for (int c = 0; c < ns; c++) w->V[(size_t)k * ns + c] = w->V[c];
            for (int j = 0; j < 2 * Nc; j++) w->U[j] = 0.0;
            cold = 0;
        }
        for (int k = 1; k <= N; k++) for (int c = 0; c < ns; c++) PUSH(w->V[(size_t)k * ns + c], LBV(w, k, c), UBV(w, k, c));
        for (int j = 0; j < Nc; j++) for (int e = 0; e < 2; e++) PUSH(w->U[2 * j + e], LBU(w, j, e), UBU(w, j, e));
        for (int k = 1; k <= N; k++)
            for (int c = 0; c < ns; c++) {
                size_t o = (size_t)k * ns + c;
                double lo = LBV(w, k, c), hi = UBV(w, k, c);
                w->SL[o] = isfinite(lo) ? fmax(w->V[o] - lo, 1e-12) : 1.0; w->ZL[o] = isfinite(lo) ? mu / w->SL[o] : 0.0;
                w->SU[o] = isfinite(hi) ? fmax(hi - w->V[o], 1e-12) : 1.0; w->ZU[o] = isfinite(hi) ? mu / w->SU[o] : 0.0;
            }
        for (int j = 0; j < 2 * Nc; j++) {
            w->SLu[j] = fmax(w->U[j] - LBU(w, j / 2, j % 2), 1e-12); w->ZLu[j] = mu / w->SLu[j];
            w->SUu[j] = fmax(UBU(w, j / 2, j % 2) - w->U[j], 1e-12); w->ZUu[j] = mu / w->SUu[j];
        }
        memset(w->lam, 0, sizeof(double) * (size_t)(N + 1) * 3);
        memset(w->eta, 0, sizeof(double) * (size_t)(N + 1) * R);
        f = eval_point(w, w->V, w->U, w->sn, w->cs, &th0, &e_c);
        delta_last = 0.0; nu_pen = 1.0; need_shift = 0; mcount = 0; restarting = 0; n_short = 0;

        for (;;) {
            /* ---- A. optimality error (IPOPT eq. 5) */
            double e_d = 0.0, e_h = 0.0, zsum = 0.0, lsum = 0.0, cmax = 0.0, cmin = INFINITY;
            for (int k = 1; k <= N; k++) {
                const double *v = w->V + (size_t)k * ns, *l = w->lam + 3 * k, *et = w->eta + (size_t)k * R;
                double r[3] = {l[0], l[1], l[2]};
                lsum += fabs(l[0]) + fabs(l[1]) + fabs(l[2]);
                if (k < N) {
                    const double *ln = w->lam + 3 * (k + 1), *u = w->U + 2 * ctrl_of(w, k);
                    double a = -T * u[0] * w->sn[k], b = T * u[0] * w->cs[k];
                    for (int i = 0; i < 3; i++) r[i] += 2 * w->q[i] * (v[i] - w->xs[i]) - ln[i];
                    r[2] -= a * ln[0] + b * ln[1];
                }
                for (int m = 0; m < R; m++) {
                    double sx, sy; gdist(w, m, v[0], v[1], &sx, &sy);
                    r[0] -= et[m] * sx; r[1] -= et[m] * sy;
                    double rd = et[m] + ((k < N && w->lw != 0.0) ? -2.0 * w->lw / (v[3 + m] * v[3 + m] * v[3 + m]) : 0.0);
                    size_t o = (size_t)k * ns + 3 + m;
                    rd -= w->ZL[o] - w->ZU[o];
                    e_d = fmax(e_d, fabs(rd)); lsum += fabs(et[m]);
                }
                for (int i = 0; i < 3; i++) { size_t o = (size_t)k * ns + i; r[i] -= w->ZL[o] - w->ZU[o]; e_d = fmax(e_d, fabs(r[i])); }
                for (int c = 0; c < ns; c++) {
                    size_t o = (size_t)k * ns + c;
                    if (isfinite(LBV(w, k, c))) { double pz = w->SL[o] * w->ZL[o]; zsum += w->ZL[o]; cmax = fmax(cmax, pz); cmin = fmin(cmin, pz); e_h = fmax(e_h, fabs((v[c] - LBV(w, k, c)) - w->SL[o])); }
                    if (isfinite(UBV(w, k, c))) { double pz = w->SU[o] * w->ZU[o]; zsum += w->ZU[o]; cmax = fmax(cmax, pz); cmin = fmin(cmin, pz); e_h = fmax(e_h, fabs((UBV(w, k, c) - v[c]) - w->SU[o])); }
                }
            }
            for (int j = 0; j < Nc; j++) {
                double ru[2] = {0, 0};
                for (int k = j; k < N; k++) {
                    if (ctrl_of(w, k) != j) break;
                    const double *ln = w->lam + 3 * (k + 1), *u = w->U + 2 * j;
                    ru[0] += 2 * w->r[0] * u[0] - T * (w->cs[k] * ln[0] + w->sn[k] * ln[1]);
                    ru[1] += 2 * w->r[1] * u[1] - T * ln[2];
                }
                for (int e = 0; e < 2; e++) {
                    int o = 2 * j + e;
                    ru[e] -= w->ZLu[o] - w->ZUu[o];
                    e_d = fmax(e_d, fabs(ru[e]));
                    zsum += w->ZLu[o] + w->ZUu[o];
                    double p0 = w->SLu[o] * w->ZLu[o], p1 = w->SUu[o] * w->ZUu[o];
                    cmax = fmax(cmax, fmax(p0, p1)); cmin = fmin(cmin, fmin(p0, p1));
                    e_h = fmax(e_h, fmax(fabs((w->U[o] - LBU(w, j, e)) - w->SLu[o]), fabs((UBU(w, j, e) - w->U[o]) - w->SUu[o])));
                }
            }
            const double smax = 100.0;
            double s_d = fmax(smax, (lsum + zsum) / (double)(N * ns + n_ineq)) / smax;
            double s_c = fmax(smax, zsum / (double)(n_ineq > 0 ? n_ineq : 1)) / smax;
            double E0 = fmax(fmax(e_d / s_d, e_c), fmax(e_h, cmax / s_c));
            kkt = E0;
            if (!(E0 == E0)) { if (n_cold < NMPC_COLD_RETRIES && it < w->max_iter) { LIDAR_COLD_RETRY(); break; } status = NMPC_STATUS_NUMERIC; break; }
            if (E0 <= w->tol) { status = NMPC_STATUS_CONVERGED; break; }
            if (it >= w->max_iter) { status = NMPC_STATUS_MAX_ITER; break; }
            if (n_cold < NMPC_COLD_RETRIES && it - it_base >= NMPC_COLD_RETRY_ITERS) { LIDAR_COLD_RETRY(); break; }
            const double mu_min = w->tol / 10.0;
            for (;;) {
                double cm = fmax(fabs(cmax - mu), fabs(cmin - mu));
                double Emu = fmax(fmax(e_d / s_d, e_c), fmax(e_h, cm / s_c));
                if (mu > mu_min && Emu <= 10.0 * mu) mu = fmax(mu_min, fmin(0.2 * mu, pow(mu, 1.5)));
                else break;
            }
            const double tau = fmax(0.99, 1.0 - mu);

            /* ---- B0. condensed stage blocks.  slot: v = mu/s - sigma (h - s), sigma = z/s */
            for (int k = 1; k <= N; k++) {
                const double *v = w->V + (size_t)k * ns;
                double *H = w->Hxx + 6 * k, *g = w->gx + 3 * k;
                H[0] = H[1] = H[2] = H[3] = 0.0; g[0] = g[1] = g[2] = 0.0;        /* xx, xy, yy, tt */
                double hd[3] = {0, 0, 0};
                if (k < N) for (int i = 0; i < 3; i++) { hd[i] = 2 * w->q[i]; g[i] = 2 * w->q[i] * (v[i] - w->xs[i]); }
                for (int c = 0; c < ns; c++) {
                    size_t o = (size_t)k * ns + c;
                    double hs = 0.0, gs = 0.0;
                    if (isfinite(LBV(w, k, c))) { double sg = w->ZL[o] / w->SL[o]; hs += sg; gs -= mu / w->SL[o] - sg * ((v[c] - LBV(w, k, c)) - w->SL[o]); }
                    if (isfinite(UBV(w, k, c))) { double sg = w->ZU[o] / w->SU[o]; hs += sg; gs += mu / w->SU[o] - sg * ((UBV(w, k, c) - v[c]) - w->SU[o]); }
                    if (c < 3) { hd[c] += hs; g[c] += gs; }
                    else {
                        int m = c - 3;
                        double d = v[c];
                        double Wd = hs + ((k < N && w->lw != 0.0) ? 6.0 * w->lw / (d * d * d * d) : 0.0);
                        double gd = gs + ((k < N && w->lw != 0.0) ? -2.0 * w->lw / (d * d * d) : 0.0);
                        w->Wd[(size_t)k * R + m] = Wd; w->gdv[(size_t)k * R + m] = gd;
                        double sx, sy, re = gdist(w, m, v[0], v[1], &sx, &sy) - d;      /* linearised row: dd = G dx + re */
                        double t = gd + Wd * re;
                        g[0] += sx * t; g[1] += sy * t;
                        H[0] += Wd * sx * sx; H[1] += Wd * sx * sy; H[2] += Wd * sy * sy;
                    }
                }
                H[0] += hd[0]; H[2] += hd[1]; H[3] += hd[2];
                if (k < N) {
                    const double *ln = w->lam + 3 * (k + 1), *u = w->U + 2 * ctrl_of(w, k);
                    H[3] += T * u[0] * (ln[0] * w->cs[k] + ln[1] * w->sn[k]);
                }
            }
            for (int k = 0; k < N; k++) { const double *ln = w->lam + 3 * (k + 1); w->hvt[k] = T * (ln[0] * w->sn[k] - ln[1] * w->cs[k]); }
            for (int j = 0; j < Nc; j++)
                for (int e = 0; e < 2; e++) {
                    int o = 2 * j + e, cnt = (j < Nc - 1) ? 1 : N - Nc + 1;
                    double sl = w->SLu[o], su = w->SUu[o], zl = w->ZLu[o], zu = w->ZUu[o], u = w->U[o];
                    w->huu[o] = cnt * 2 * w->r[e] + zl / sl + zu / su;
                    double vl = mu / sl - zl / sl * ((u - LBU(w, j, e)) - sl), vu = mu / su - zu / su * ((UBU(w, j, e) - u) - su);
                    w->gu[o] = cnt * 2 * w->r[e] * u - (vl - vu);
                }

            /* ---- B. Riccati sweep on z = (x (3), held control (2)) with inertia correction */
            double delta = need_shift ? fmax(1e-20, 0.25 * delta_last) : 0.0;
            int ntry = 0, ok = 0;
            double P5[25], p5[5], Mx[25], mv[5];
            for (;;) {
                ok = 1;
                memset(P5, 0, sizeof(P5)); memset(p5, 0, sizeof(p5));
                { const double *H = w->Hxx + 6 * N, *g = w->gx + 3 * N;
                  P5[0] = H[0]; P5[1] = P5[5] = H[1]; P5[6] = H[2]; P5[12] = H[3]; p5[0] = g[0]; p5[1] = g[1]; p5[2] = g[2]; }
                for (int k = N - 1; k >= 0; k--) {
                    const int j = ctrl_of(w, k);
                    const double *v = w->V + (size_t)k * ns, *vn = v + ns, *u = w->U + 2 * j;
                    const double s = w->sn[k], c = w->cs[k], a = -T * u[0] * s, b = T * u[0] * c;
                    const double cd[3] = {vn[0] - (v[0] + T * u[0] * c), vn[1] - (v[1] + T * u[0] * s), vn[2] - (v[2] + T * u[1])};
                    /* At = [[A, B], [0, I]]: columns x y t v w -> next (x y t v w) */
                    double At[25] = {1, 0, a, T * c, 0,   0, 1, b, T * s, 0,   0, 0, 1, 0, T,   0, 0, 0, 1, 0,   0, 0, 0, 0, 1};
                    double pb[5], G5[25];
                    for (int r_ = 0; r_ < 5; r_++) { double t = p5[r_]; for (int q_ = 0; q_ < 3; q_++) t -= P5[r_ * 5 + q_] * cd[q_]; pb[r_] = t; }
                    for (int r_ = 0; r_ < 5; r_++) for (int q_ = 0; q_ < 5; q_++) { double t = 0; for (int z = 0; z < 5; z++) t += P5[r_ * 5 + z] * At[z * 5 + q_]; G5[r_ * 5 + q_] = t; }
                    for (int r_ = 0; r_ < 5; r_++) {
                        for (int q_ = 0; q_ < 5; q_++) { double t = 0; for (int z = 0; z < 5; z++) t += At[z * 5 + r_] * G5[z * 5 + q_]; Mx[r_ * 5 + q_] = t; }
                        double t = 0; for (int z = 0; z < 5; z++) t += At[z * 5 + r_] * pb[z]; mv[r_] = t;
                    }
                    /* stage Hessian / gradient: pose block (k >= 1), control cost of this stage, cross term (theta, v) */
                    if (k >= 1) {
                        const double *H = w->Hxx + 6 * k, *g = w->gx + 3 * k;
                        Mx[0] += H[0]; Mx[1] += H[1]; Mx[5] += H[1]; Mx[6] += H[2]; Mx[12] += H[3];
                        mv[0] += g[0]; mv[1] += g[1]; mv[2] += g[2];
                    }
                    Mx[2 * 5 + 3] += w->hvt[k]; Mx[3 * 5 + 2] += w->hvt[k];
                    if (k <= Nc - 1) {        /* the stage where control j is decided carries its whole diagonal / gradient */
                        Mx[3 * 5 + 3] += w->huu[2 * j] + delta; Mx[4 * 5 + 4] += w->huu[2 * j + 1] + delta;
                        mv[3] += w->gu[2 * j]; mv[4] += w->gu[2 * j + 1];
                        /* eliminate (v, w): 2 pivots */
                        double d0 = Mx[18], dv = d0;
                        if (!(dv > 1e-9 * fabs(d0)) || !(dv > 0.0)) { ok = 0; break; }
                        double l43 = Mx[4 * 5 + 3] / dv;
                        double d1o = Mx[24], d1 = d1o - l43 * Mx[3 * 5 + 4];
                        if (!(d1 > 1e-9 * fabs(d1o)) || !(d1 > 0.0)) { ok = 0; break; }
                        /* gains: solve [Mvv Mvw; Mwv Mww] [dv; dw] = -([Mvx; Mwx] dx + [mv3; mv4]) */
                        double *Kk = w->Kg + 6 * j, *kk = w->kff + 2 * j;
                        for (int q_ = 0; q_ < 4; q_++) {
                            double r3 = (q_ < 3) ? Mx[3 * 5 + q_] : mv[3], r4 = (q_ < 3) ? Mx[4 * 5 + q_] : mv[4];
                            double y4 = (r4 - l43 * r3) / d1, y3 = (r3 - Mx[3 * 5 + 4] * y4) / dv;
                            if (q_ < 3) { Kk[q_] = -y3; Kk[3 + q_] = -y4; } else { kk[0] = -y3; kk[1] = -y4; }
                        }
                        /* Schur complement on the pose block */
                        double Pn[9], pn[3];
                        for (int r_ = 0; r_ < 3; r_++) {
                            for (int q_ = 0; q_ < 3; q_++) Pn[r_ * 3 + q_] = Mx[r_ * 5 + q_] + Mx[r_ * 5 + 3] * Kk[q_] + Mx[r_ * 5 + 4] * Kk[3 + q_];
                            pn[r_] = mv[r_] + Mx[r_ * 5 + 3] * kk[0] + Mx[r_ * 5 + 4] * kk[1];
                        }
                        memset(P5, 0, sizeof(P5)); memset(p5, 0, sizeof(p5));
                        for (int r_ = 0; r_ < 3; r_++) { for (int q_ = 0; q_ < 3; q_++) P5[r_ * 5 + q_] = 0.5 * (Pn[r_ * 3 + q_] + Pn[q_ * 3 + r_]); p5[r_] = pn[r_]; }
                    } else {                  /* held control: it stays a parameter of the cost-to-go; its stage cost 2R is part of huu at Nc-1 */
                        for (int z = 0; z < 25; z++) P5[z] = Mx[z];
                        for (int r_ = 0; r_ < 5; r_++) for (int q_ = r_ + 1; q_ < 5; q_++) { double t = 0.5 * (P5[r_ * 5 + q_] + P5[q_ * 5 + r_]); P5[r_ * 5 + q_] = t; P5[q_ * 5 + r_] = t; }
                        for (int z = 0; z < 5; z++) p5[z] = mv[z];
                    }
                }
                if (ok) break;
                ntry++;
                if (delta == 0.0) delta = (delta_last == 0.0) ? 1e-4 : fmax(1e-20, delta_last / 3.0);
                else delta *= (delta_last == 0.0) ? 100.0 : NMPC_SHIFT_ESCALATION;
                if (delta > 1e20) break;
            }
            if (!ok) { if (n_cold < NMPC_COLD_RETRIES) { LIDAR_COLD_RETRY(); it++; break; } status = NMPC_STATUS_NUMERIC; break; }
            if (delta > 0.0) delta_last = delta;
            need_shift = delta > 0.0 && (ntry > 0 || delta > 1e-6);

            /* ---- C. forward sweep */
            for (int c = 0; c < ns; c++) w->dV[c] = 0.0;
            for (int k = 0; k < N; k++) {
                const int j = ctrl_of(w, k);
                const double *dx = w->dV + (size_t)k * ns, *v = w->V + (size_t)k * ns, *vn = v + ns, *u = w->U + 2 * j;
                double *du = w->dU + 2 * j, *dxn = w->dV + (size_t)(k + 1) * ns;
                if (k <= Nc - 1) {
                    const double *Kk = w->Kg + 6 * j, *kk = w->kff + 2 * j;
                    du[0] = kk[0] + Kk[0] * dx[0] + Kk[1] * dx[1] + Kk[2] * dx[2];
                    du[1] = kk[1] + Kk[3] * dx[0] + Kk[4] * dx[1] + Kk[5] * dx[2];
                }
                const double s = w->sn[k], c = w->cs[k];
                dxn[0] = dx[0] + (-T * u[0] * s) * dx[2] + T * c * du[0] - (vn[0] - (v[0] + T * u[0] * c));
                dxn[1] = dx[1] + (T * u[0] * c) * dx[2] + T * s * du[0] - (vn[1] - (v[1] + T * u[0] * s));
                dxn[2] = dx[2] + T * du[1] - (vn[2] - (v[2] + T * u[1]));
                for (int m = 0; m < R; m++) {     /* dd = G dx + (g - d) at stage k+1 */
                    double sx, sy, gg = gdist(w, m, vn[0], vn[1], &sx, &sy);
                    dxn[3 + m] = sx * dxn[0] + sy * dxn[1] + (gg - vn[3 + m]);
                }
            }
            /* ---- multipliers of the QP: eta+ from the distance rows, lambda+ by the adjoint recursion */
            double mult_max = 0.0;
            for (int k = N; k >= 1; k--) {
                const double *H = w->Hxx + 6 * k, *g = w->gx + 3 * k, *dx = w->dV + (size_t)k * ns;
                double *l = w->lamn + 3 * k;
                l[0] = -(g[0] + H[0] * dx[0] + H[1] * dx[1]); l[1] = -(g[1] + H[1] * dx[0] + H[2] * dx[1]); l[2] = -(g[2] + H[3] * dx[2]);
                if (k < N) {
                    const double *ln = w->lamn + 3 * (k + 1), *u = w->U + 2 * ctrl_of(w, k);
                    l[0] += ln[0]; l[1] += ln[1];
                    l[2] += ln[2] + (-T * u[0] * w->sn[k]) * ln[0] + (T * u[0] * w->cs[k]) * ln[1] - w->hvt[k] * w->dU[2 * ctrl_of(w, k)];
                }
                for (int i = 0; i < 3; i++) mult_max = fmax(mult_max, fabs(l[i]));
                for (int m = 0; m < R; m++) {
                    double e = -(w->gdv[(size_t)k * R + m] + w->Wd[(size_t)k * R + m] * dx[3 + m]);
                    w->etan[(size_t)k * R + m] = e; mult_max = fmax(mult_max, fabs(e));
                }
            }
            /* ---- D. fraction to the boundary; dphi */
            double a_p = 1.0, a_d = 1.0, dphi = 0.0, lgs = 0.0, thh = 0.0;
#define SLOT(s_, z_, h_, jd_)                                                                                                     \
    do {                                                                                                                          \
        double ds_ = (jd_) + ((h_) - (s_)), dz_ = (mu - (s_) * (z_) - (z_) * ds_) / (s_);                                         \
        if (ds_ < 0.0) a_p = fmin(a_p, -tau * (s_) / ds_);                                                                        \
        if (dz_ < 0.0) a_d = fmin(a_d, -tau * (z_) / dz_);                                                                        \
        dphi -= mu * ds_ / (s_); lgs += log(s_); thh += fabs((h_) - (s_));                                                        \
    } while (0)
            for (int k = 1; k <= N; k++)
                for (int c = 0; c < ns; c++) {
                    size_t o = (size_t)k * ns + c;
                    if (isfinite(LBV(w, k, c))) SLOT(w->SL[o], w->ZL[o], w->V[o] - LBV(w, k, c), w->dV[o]);
                    if (isfinite(UBV(w, k, c))) SLOT(w->SU[o], w->ZU[o], UBV(w, k, c) - w->V[o], -w->dV[o]);
                }
            for (int o = 0; o < 2 * Nc; o++) {
                SLOT(w->SLu[o], w->ZLu[o], w->U[o] - LBU(w, o / 2, o % 2), w->dU[o]);
                SLOT(w->SUu[o], w->ZUu[o], UBU(w, o / 2, o % 2) - w->U[o], -w->dU[o]);
            }
            for (int k = 1; k < N; k++) {
                const double *v = w->V + (size_t)k * ns, *dx = w->dV + (size_t)k * ns;
                for (int i = 0; i < 3; i++) dphi += 2 * w->q[i] * (v[i] - w->xs[i]) * dx[i];
                if (w->lw != 0.0) for (int m = 0; m < R; m++) dphi += -2.0 * w->lw / (v[3 + m] * v[3 + m] * v[3 + m]) * dx[3 + m];
            }
            for (int j = 0; j < Nc; j++) { int cnt = (j < Nc - 1) ? 1 : N - Nc + 1; for (int e = 0; e < 2; e++) dphi += cnt * 2 * w->r[e] * w->U[2 * j + e] * w->dU[2 * j + e]; }
            /* ---- E. l1 merit backtracking (non-monotone, as nmpc_oracle.c) */
            const double theta0 = th0 + thh, phi0 = f - mu * lgs;
            if (theta0 > 0.0) {
                double nut = fmin(dphi / ((1.0 - 0.1) * theta0), mult_max / (1.0 - 0.1));
                nu_pen = fmax(1.0, 0.5 * nu_pen);
                if (nu_pen < nut) nu_pen = nut + 1.0;
            }
            const double D = dphi - nu_pen * theta0;
            double alpha = a_p, ft = f, tht = th0, ect = e_c;
            if (mh_mu != mu || mh_nu != nu_pen) { mcount = 0; mh_mu = mu; mh_nu = nu_pen; }
            const double m0 = phi0 + nu_pen * theta0;
            double mref = m0;
            if (mcount > 0) mref = fmax(mref, mh0);
            if (mcount > 1) mref = fmax(mref, mh1);
            if (mcount > 2) mref = fmax(mref, mh2);
            mh2 = mh1; mh1 = mh0; mh0 = m0; if (mcount < 3) mcount++;
            int wd_took = 0;
            const int wd_fire = wd_trig > 0 && n_short >= wd_trig && a_p >= NMPC_WATCHDOG_MIN_AP;
            for (int ls = 0; ls < 30; ls++) {
                for (size_t i = 0; i < (size_t)(N + 1) * ns; i++) w->Vt[i] = w->V[i] + alpha * w->dV[i];
                for (int i = 0; i < 2 * Nc; i++) w->Ut[i] = w->U[i] + alpha * w->dU[i];
                ft = eval_point(w, w->Vt, w->Ut, NULL, NULL, &tht, &ect);
                double lgt = 0.0, tb = 0.0;
#define TRIAL(s_, h0_, jd_, ht_) do { double st_ = (s_) + alpha * ((jd_) + ((h0_) - (s_))); lgt += log(st_); tb += fabs((ht_) - st_); } while (0)
                for (int k = 1; k <= N; k++)
                    for (int c = 0; c < ns; c++) {
                        size_t o = (size_t)k * ns + c;
                        if (isfinite(LBV(w, k, c))) TRIAL(w->SL[o], w->V[o] - LBV(w, k, c), w->dV[o], w->Vt[o] - LBV(w, k, c));
                        if (isfinite(UBV(w, k, c))) TRIAL(w->SU[o], UBV(w, k, c) - w->V[o], -w->dV[o], UBV(w, k, c) - w->Vt[o]);
                    }
                for (int o = 0; o < 2 * Nc; o++) {
                    TRIAL(w->SLu[o], w->U[o] - LBU(w, o / 2, o % 2), w->dU[o], w->Ut[o] - LBU(w, o / 2, o % 2));
                    TRIAL(w->SUu[o], UBU(w, o / 2, o % 2) - w->U[o], -w->dU[o], UBU(w, o / 2, o % 2) - w->Ut[o]);
                }
                const double mt = (ft - mu * lgt) + nu_pen * (tht + tb);
                if (wd_fire && isfinite(mt)) { mcount = 0; wd_took = 1; break; }      /* watchdog: the fraction-to-the-boundary step, no merit test, merit history dropped */
                if (mt <= mref + 1e-4 * alpha * D + 1e-13 * fabs(phi0)) break;
                if (ls < 29) alpha *= 0.5;
            }
            a_d = fmin(a_d, alpha);
            n_tiny = (alpha < 1e-10) ? n_tiny + 1 : 0;
            n_short = (!wd_took && alpha < a_p) ? n_short + 1 : 0;      /* a shortened step: the backtracking cut it */
            if (trace) fprintf(stderr, "it %3d mu %.3e E0 %.3e e_d %.3e e_c %.3e delta %.3e ntry %d a_p %.3e alpha %.3e a_d %.3e f %.9g theta %.3e nu %.3e\n",
                               it, mu, E0, e_d / s_d, e_c, delta, ntry, a_p, alpha, a_d, f, theta0, nu_pen);
            /* ---- G. accept: duals and slacks (they need the old primal point), then the primal point */
#define UPD(s_, z_, h_, jd_)                                                                                                      \
    do {                                                                                                                          \
        double ds_ = (jd_) + ((h_) - (s_)), dz_ = (mu - (s_) * (z_) - (z_) * ds_) / (s_);                                         \
        double sn_ = (s_) + alpha * ds_, zn_ = (z_) + a_d * dz_;                                                                  \
        (s_) = sn_; (z_) = fmin(fmax(zn_, mu / (1e10 * sn_)), 1e10 * mu / sn_);                                                   \
    } while (0)
            for (int k = 1; k <= N; k++)
                for (int c = 0; c < ns; c++) {
                    size_t o = (size_t)k * ns + c;
                    if (isfinite(LBV(w, k, c))) UPD(w->SL[o], w->ZL[o], w->V[o] - LBV(w, k, c), w->dV[o]);
                    if (isfinite(UBV(w, k, c))) UPD(w->SU[o], w->ZU[o], UBV(w, k, c) - w->V[o], -w->dV[o]);
                }
            for (int o = 0; o < 2 * Nc; o++) {
                UPD(w->SLu[o], w->ZLu[o], w->U[o] - LBU(w, o / 2, o % 2), w->dU[o]);
                UPD(w->SUu[o], w->ZUu[o], UBU(w, o / 2, o % 2) - w->U[o], -w->dU[o]);
            }
            { double *t; t = w->V; w->V = w->Vt; w->Vt = t; t = w->U; w->U = w->Ut; w->Ut = t; }
            for (int k = 1; k <= N; k++) {
                for (int i = 0; i < 3; i++) w->lam[3 * k + i] += alpha * (w->lamn[3 * k + i] - w->lam[3 * k + i]);
                for (int m = 0; m < R; m++) w->eta[(size_t)k * R + m] += alpha * (w->etan[(size_t)k * R + m] - w->eta[(size_t)k * R + m]);
            }
            f = eval_point(w, w->V, w->U, w->sn, w->cs, &th0, &e_c);
            it++;
            if (n_tiny >= 5) {
                if (n_restart >= w->max_restarts) { if (n_cold < NMPC_COLD_RETRIES) { LIDAR_COLD_RETRY(); break; } status = NMPC_STATUS_STALLED; break; }
                n_restart++; n_tiny = 0; mu = fmax(mu, w->mu_init); restarting = 1;
                break;
            }
        }
        if (!restarting) break;
    }
    memcpy(wout, w->V, sizeof(double) * (size_t)(N + 1) * ns);
    memcpy(wout + (size_t)(N + 1) * ns, w->U, sizeof(double) * (size_t)Nc * 2);
    *obj_out = f; *iters_out = it; *kkt_out = kkt;
    return status;
}

int32_t nmpc_lidar_oracle_n_var(const nmpc_lidar_config_t *c) { return c ? (3 + c->R) * (c->N + 1) + 2 * c->Nc : NMPC_E_ARG; }
int32_t nmpc_lidar_oracle_n_g(const nmpc_lidar_config_t *c) { return c ? (3 + c->R) * (c->N + 1) : NMPC_E_ARG; }
int32_t nmpc_lidar_oracle_n_p(const nmpc_lidar_config_t *c) { return c ? 6 + 2 * c->R : NMPC_E_ARG; }

int32_t nmpc_lidar_oracle_solve_batch(const nmpc_lidar_config_t *cfg, const double *lbx, const double *ubx, int32_t B, const double *p, const double *w0,
                                      double *w_out, double *obj, int32_t *status, int32_t *iters, double *kkt, int32_t nthreads)
{
    if (!cfg || cfg->R < 0 || cfg->R > RMAX || cfg->N < 1 || cfg->Nc < 1 || cfg->Nc > cfg->N) return NMPC_E_ARG;
    const int nv = nmpc_lidar_oracle_n_var(cfg), np_ = nmpc_lidar_oracle_n_p(cfg);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel
    {
        lw_t *w = lw_new(cfg, lbx, ubx);
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; b++) {
            double o = 0, k = 0; int it = 0;
            int st = solve_one(w, p + (size_t)b * np_, w0 + (size_t)b * nv, w_out + (size_t)b * nv, &o, &it, &k);
            if (obj) obj[b] = o;
            if (status) status[b] = st;
            if (iters) iters[b] = it;
            if (kkt) kkt[b] = k;
        }
        lw_free(w);
    }
    return NMPC_OK;
}

/* f (V4:135-136) and g = [gx; gd] (V4:151) in the reference's order */
int32_t nmpc_lidar_oracle_eval_batch(const nmpc_lidar_config_t *cfg, int32_t B, const double *p, const double *wv, double *f, double *g)
{
    const int N = cfg->N, R = cfg->R, ns = 3 + R, nv = nmpc_lidar_oracle_n_var(cfg), ng = nmpc_lidar_oracle_n_g(cfg), np_ = nmpc_lidar_oracle_n_p(cfg);
    for (int b = 0; b < B; b++) {
        const double *X = wv + (size_t)b * nv, *U = X + (size_t)(N + 1) * ns, *pp = p + (size_t)b * np_;
        double *gx = g ? g + (size_t)b * ng : NULL, *gd = g ? gx + 3 * (N + 1) : NULL;
        double po[RMAX][2], fv = 0.0;
        for (int m = 0; m < R; m++) { double a = X[2] + pp[6 + R + m]; po[m][0] = X[0] + X[3 + m] * cos(a); po[m][1] = X[1] + X[3 + m] * sin(a); }
        if (g) { for (int i = 0; i < 3; i++) gx[i] = X[i] - pp[i]; for (int m = 0; m < R; m++) gd[m] = X[3 + m] - pp[6 + m]; }
        for (int k = 0; k < N; k++) {
            const double *v = X + (size_t)k * ns, *vn = v + ns, *u = U + 2 * (k < cfg->Nc - 1 ? k : cfg->Nc - 1);
            for (int i = 0; i < 3; i++) { double e = v[i] - pp[3 + i]; fv += cfg->q[i] * e * e; }
            fv += cfg->r[0] * u[0] * u[0] + cfg->r[1] * u[1] * u[1];
            if (cfg->lw != 0.0) for (int m = 0; m < R; m++) fv += cfg->lw / (v[3 + m] * v[3 + m]);
            if (g) {
                gx[3 * (k + 1)] = vn[0] - (v[0] + cfg->T * u[0] * cos(v[2]));
                gx[3 * (k + 1) + 1] = vn[1] - (v[1] + cfg->T * u[0] * sin(v[2]));
                gx[3 * (k + 1) + 2] = vn[2] - (v[2] + cfg->T * u[1]);
                for (int m = 0; m < R; m++) gd[R * (k + 1) + m] = vn[3 + m] - (fabs(vn[0] - po[m][0]) + fabs(vn[1] - po[m][1]));
            }
        }
        if (f) f[b] = fv;
    }
    return NMPC_OK;
}
