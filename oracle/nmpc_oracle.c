/*
 * nmpc_oracle.c — TEST INFRASTRUCTURE.  CPU fp64 restatement of the reference's per-step
 * NMPC solve.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's library; the product (libnmpc_hip.so) never links or calls it.
 *
 * PARITY UNPINNED against CasADi/IPOPT: the reference's arithmetic lives in the
 * un-vendored, unpinned third-party packages `casadi` and IPOPT (call sites
 * C6:15,207-346,432), neither installed here, and the reference holds no tests or golden
 * vectors.  This file is pinned instead by (a) the solver-independent known answers of
 * SURVEY.md §8c items 1-5 via oracle/nlp_ref.py, (b) scipy-SLSQP golden triples in
 * tests/golden/ ("scipy-SLSQP oracle, not CasADi/IPOPT").
 *
 * What is restated (C6 = AllScripts/centralized_six_robots_implementation.py):
 *   NLP:    unicycle rhs C6:207-237; Euler defect C6:318-323; stage cost C6:314 with
 *           Q,R of C6:252-266; pair rows C6:288-306; initial block C6:273-278; packing
 *           C6:339; bounds C6:349-352; obstacle rows
 *           AllScripts/third_scenario_mpc_obstacle_avoidance.py:145-150,175-177.
 *   solve:  C6:345-346,432 — nlpsol('ipopt').  IPOPT is absent, so its published
 *           algorithm (Waechter & Biegler, Math. Program. 106, 2006) is restated:
 *           slack/barrier reformulation, monotone mu update (their eq. 7 with
 *           kappa_mu=0.2, theta_mu=1.5, kappa_eps=10), fraction-to-boundary (eq. 8, 15),
 *           dual safeguard (eq. 16), scaled optimality error (eq. 5-6, s_max=100),
 *           inertia correction schedule (alg. IC) with the shift delta applied to the control
 *           diagonal only (every Quu_k > 0 is what the Riccati sweep needs), bound_push=1e-2; the linear system
 *           is solved by a Riccati sweep over the stages (block elimination of the same
 *           KKT matrix) and the line search is a non-monotone l1-merit backtracking search
 *           (reference = max of the last four merit values) instead of IPOPT's filter.
 *           Further deliberate differences (each measured): mu_init = 0.5, the
 *           dual step is capped by the accepted primal step, the first inertia trial after
 *           an iteration that needed a shift is a quarter of that shift, and a stall (5
 *           steps below 1e-10) triggers a barrier restart from the interior-pushed current
 *           point (at most 3, then NMPC_STATUS_STALLED) in place of IPOPT's restoration
 *           phase; a solve that still fails (stall after the restarts, numerical failure, 500 iterations
 *           without convergence) is restarted from the reference's cold start X_k = x0, U = 0, and a
 *           second time from there in an ELASTIC phase (pair / obstacle rows relaxed under an l1
 *           penalty: see barrier_and_infeas_t below) — the restoration of last resort (round 4).
 *           The backward sweep re-factors only a few stages after a rejected pivot (round 4, see the sweep).
 *           Parity is therefore at the KKT point, not on iterates.
 *   shift:  C6:160-169,460-465; plant step AllScripts/casadi_test.py:17-26.
 *
 * Plain scalar C99, one instance at a time; the batch driver runs instances in
 * parallel with OpenMP (one instance per thread).
 */
#include "../include/nmpc.h"
#include "../include/nmpc_constants.h"      /* NMPC_SHIFT_ESCALATION, NMPC_COLD_RETRY_ITERS, NMPC_COLD_RETRIES, NMPC_X0_TOL: shared with the HIP kernels */

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif


#define NXM (3 * NMPC_MAX_ROBOTS)
#define NUM_ (2 * NMPC_MAX_ROBOTS)

typedef struct {
    int m, N, K, nx, nu, M, nxb, nh, thb;
    int trace;      /* NMPC_ORACLE_TRACE=1: one line per iteration on stderr (development aid) */
    int ipopt;      /* NMPC_ORACLE_IPOPT_DEFAULTS=1: IPOPT-faithful variant for the basin-sensitivity estimate: mu_init = 0.1 unless the
                       caller's differs from this solver's 0.5, filter line search (Waechter & Biegler alg. A: filter with margins, switching
                       condition, Armijo on the barrier function) instead of the l1 merit function, dual step length from its own
                       fraction-to-the-boundary rule (no cap by the primal step), no cold-start retry.  Not reproduced: second-order correction,
                       the restoration phase (the barrier restart stands in), IPOPT's NLP scaling and linear solver.  Never the shipped algorithm. */
    int sreg;       /* inertia correction variant.  -1 (NMPC_ORACLE_STAGE_REG unset): the shipped one, partial re-factorisation from a saved cost-to-go;
                       0: rounds 1-3 (whole sweep again); 1, 2, 3, 5: the stage-local experiments of VERDICT r3 item 1 (tools/sreg_experiment.py; never shipped) */
    double sreg_beta;
    double *dl_k; int *ns_k; double *Psnap, *pvsnap; int sreg_J, sreg_C;   /* per-(stage, control) shift memory of the stage-local variants */
    int el;         /* 1 while the solve is in its elastic phase (the second restart of last resort): pair / obstacle rows h + t - s = 0, t >= 0, penalty NMPC_ELASTIC_RHO t */
    double rho;
    double *El, *Elt, *dEl;      /* elastic variables t, trial, step */
    int max_restarts;   /* barrier restarts after a stall: 3 (as the HIP path); NMPC_ORACLE_MAX_RESTARTS overrides (fixture generation) */
    int o_ul, o_uu, o_xl, o_xu, o_pr, o_ob;
    double T, dmin2, vmax, wmax, xymax, thmax, robdim, margin;
    double qd[NXM], rd[NUM_], lbu[NUM_], ubu[NUM_], bxs[NXM];
    int pi[NMPC_MAX_ROBOTS * (NMPC_MAX_ROBOTS - 1) / 2], pj[NMPC_MAX_ROBOTS * (NMPC_MAX_ROBOTS - 1) / 2];
    int bidx[NXM]; /* bounded-state slot -> state index */
    double obs[3 * NMPC_MAX_OBSTACLES];
    double tol, mu_init;
    int max_iter;
    /* iterate + step */
    double *X, *U, *lam, *S, *Z, *dX, *dU, *lamn, *dS, *dZ;
    double *dXc, *dUc, *dSc, *dZc, *lamnc;   /* EXPERIMENT (NMPC_ORACLE_PC=3): the corrected step next to the plain one */
    double *corr;      /* EXPERIMENT (NMPC_ORACLE_PC): second-order complementarity term ds dz / s of a predictor step, per slot */
    int pc;            /* 0 off; 1: corrector with the barrier parameter unchanged; 2: Mehrotra (affine predictor, mu scaled by (mu_aff/mu_cur)^3) */
    double *Xt, *Ut, *St;
    double *sn, *cs, *C, *H, *Ht, *Ct, *snt, *cst;
    double *Hxx, *gx, *huu, *gu, *hvt, *Kg, *kff;
} ws_t;

int32_t nmpc_oracle_n_var(const nmpc_config_t *c) { return 3 * c->m * (c->N + 1) + 2 * c->m * c->N; }
int32_t nmpc_oracle_n_p(const nmpc_config_t *c) { return 6 * c->m; }
int32_t nmpc_oracle_n_g(const nmpc_config_t *c)
{
    int M = c->pair_rows ? c->m * (c->m - 1) / 2 : 0;   /* pair_rows = 0: AS/mpc_online_casadi_tb3_multi_centralized.py:115-148 */
    return 3 * c->m + (c->pad_rows ? M : 0) + (3 * c->m + M + c->m * c->n_obs) * c->N;
}

void nmpc_oracle_config_default(nmpc_config_t *c, int32_t m, int32_t N)
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

static ws_t *ws_new(const nmpc_config_t *c)
{
    ws_t *w = (ws_t *)calloc(1, sizeof(ws_t));
    int m = c->m, N = c->N;
    w->m = m; w->N = N; w->K = c->n_obs; w->nx = 3 * m; w->nu = 2 * m; w->M = c->pair_rows ? m * (m - 1) / 2 : 0;
    w->thb = isfinite(c->th_max) ? 1 : 0;
    w->trace = getenv("NMPC_ORACLE_TRACE") != NULL;
    w->max_restarts = getenv("NMPC_ORACLE_MAX_RESTARTS") ? atoi(getenv("NMPC_ORACLE_MAX_RESTARTS")) : 3;
    w->sreg = getenv("NMPC_ORACLE_STAGE_REG") ? atoi(getenv("NMPC_ORACLE_STAGE_REG")) : -1;
    w->sreg_beta = getenv("NMPC_ORACLE_SREG_BETA") ? atof(getenv("NMPC_ORACLE_SREG_BETA")) : 1e-4;
    w->rho = NMPC_ELASTIC_RHO; w->el = 0;
    w->pc = getenv("NMPC_ORACLE_PC") ? atoi(getenv("NMPC_ORACLE_PC")) : 0;
    w->ipopt = getenv("NMPC_ORACLE_IPOPT_DEFAULTS") ? atoi(getenv("NMPC_ORACLE_IPOPT_DEFAULTS")) : 0;
    w->nxb = m * (w->thb ? 3 : 2);
    w->nh = 2 * w->nu + 2 * w->nxb + w->M + m * w->K;
    w->o_ul = 0; w->o_uu = w->nu; w->o_xl = 2 * w->nu; w->o_xu = w->o_xl + w->nxb;
    w->o_pr = w->o_xu + w->nxb; w->o_ob = w->o_pr + w->M;
    w->T = c->T; w->dmin2 = c->dmin * c->dmin; w->vmax = c->v_max; w->wmax = c->w_max;
    w->xymax = c->xy_max; w->thmax = c->th_max; w->robdim = c->rob_dim; w->margin = c->margin;
    w->tol = c->tol; w->mu_init = c->mu_init; w->max_iter = c->max_iter;
    if (w->ipopt && c->mu_init == 0.5) w->mu_init = 0.1;      /* IPOPT's default where the caller did not choose */
    memcpy(w->obs, c->obs, sizeof(w->obs));
    for (int i = 0; i < m; i++) {
        for (int d = 0; d < 3; d++) w->qd[3 * i + d] = c->q[d];
        for (int d = 0; d < 2; d++) w->rd[2 * i + d] = c->r[d];
        w->lbu[2 * i] = -c->v_max; w->ubu[2 * i] = c->v_max;
        w->lbu[2 * i + 1] = -c->w_max; w->ubu[2 * i + 1] = c->w_max;
    }
    int np = 0;
    for (int i = 0; i < m; i++) for (int j = i + 1; j < m; j++) { w->pi[np] = i; w->pj[np] = j; np++; }
    for (int s = 0; s < w->nxb; s++) {
        if (w->thb) { w->bidx[s] = s; w->bxs[s] = (s % 3 == 2) ? c->th_max : c->xy_max; }
        else { w->bidx[s] = 3 * (s / 2) + (s % 2); w->bxs[s] = c->xy_max; }
    }
    size_t nX = (size_t)(N + 1) * w->nx, nU = (size_t)N * w->nu, nH = (size_t)(N + 1) * w->nh;
#define AL(p, n) w->p = (double *)calloc((n), sizeof(double))
    AL(X, nX); AL(U, nU); AL(lam, nX); AL(S, nH); AL(Z, nH); AL(dX, nX); AL(dU, nU); AL(lamn, nX); AL(dS, nH); AL(dZ, nH); AL(corr, nH); AL(dXc, nX); AL(dUc, nU); AL(dSc, nH); AL(dZc, nH); AL(lamnc, nX);
    AL(Xt, nX); AL(Ut, nU); AL(St, nH);
    AL(sn, (size_t)N * m); AL(cs, (size_t)N * m); AL(snt, (size_t)N * m); AL(cst, (size_t)N * m);
    AL(C, nX); AL(Ct, nX); AL(H, nH); AL(Ht, nH);
    AL(Hxx, (size_t)(N + 1) * w->nx * w->nx); AL(gx, nX); AL(huu, nU); AL(gu, nU); AL(hvt, (size_t)N * m);
    AL(Kg, (size_t)N * w->nu * w->nx); AL(kff, nU); AL(dl_k, nU); AL(El, nH); AL(Elt, nH); AL(dEl, nH);
#undef AL
    w->ns_k = (int *)calloc(nU, sizeof(int));
    w->Psnap = (double *)calloc((size_t)(N + 1) * w->nx * w->nx, sizeof(double)); w->pvsnap = (double *)calloc(nX, sizeof(double));
    w->sreg_C = getenv("NMPC_ORACLE_SREG_C") ? atoi(getenv("NMPC_ORACLE_SREG_C")) : NMPC_CKPT_EVERY;
    w->sreg_J = getenv("NMPC_ORACLE_SREG_J") ? atoi(getenv("NMPC_ORACLE_SREG_J")) : NMPC_REFACTOR_BACK;
    return w;
}

static void ws_free(ws_t *w)
{
    double **ps[] = {&w->X, &w->U, &w->lam, &w->S, &w->Z, &w->dX, &w->dU, &w->lamn, &w->dS, &w->dZ, &w->Xt, &w->Ut, &w->St,
                     &w->sn, &w->cs, &w->snt, &w->cst, &w->C, &w->Ct, &w->H, &w->Ht, &w->Hxx, &w->gx, &w->huu, &w->gu,
                     &w->hvt, &w->Kg, &w->kff, &w->corr, &w->dXc, &w->dUc, &w->dSc, &w->dZc, &w->lamnc};
    for (size_t i = 0; i < sizeof(ps) / sizeof(ps[0]); i++) free(*ps[i]);
    free(w->dl_k); free(w->ns_k); free(w->Psnap); free(w->pvsnap); free(w->El); free(w->Elt); free(w->dEl);
    free(w);
}

/* which inequality slots exist at stage k: u rows k<N; x-bound rows k>=1; pair/obstacle rows 1<=k<=N-1 */
static inline int slot_active(const ws_t *w, int k, int s)
{
    if (s < w->o_xl) return k < w->N;
    if (s < w->o_pr) return k >= 1;
    return k >= 1 && k <= w->N - 1;
}

/* inequality values h_s(x_k,u_k) >= 0 for all active slots of stage k */
static void stage_h(const ws_t *w, int k, const double *x, const double *u, double *h)
{
    for (int s = 0; s < w->nh; s++) h[s] = 1.0;
    if (k < w->N)
        for (int c = 0; c < w->nu; c++) { h[w->o_ul + c] = u[c] - w->lbu[c]; h[w->o_uu + c] = w->ubu[c] - u[c]; }
    if (k >= 1)
        for (int s = 0; s < w->nxb; s++) { double v = x[w->bidx[s]]; h[w->o_xl + s] = v + w->bxs[s]; h[w->o_xu + s] = w->bxs[s] - v; }
    if (k >= 1 && k <= w->N - 1) {
        for (int pq = 0; pq < w->M; pq++) {
            int i = w->pi[pq], j = w->pj[pq];
            double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1];
            h[w->o_pr + pq] = dx * dx + dy * dy - w->dmin2;
        }
        for (int i = 0; i < w->m; i++)
            for (int o = 0; o < w->K; o++) {
                double dx = x[3 * i] - w->obs[3 * o], dy = x[3 * i + 1] - w->obs[3 * o + 1];
                h[w->o_ob + i * w->K + o] = sqrt(dx * dx + dy * dy) - w->robdim - w->obs[3 * o + 2] - w->margin;
            }
    }
}

/* out += Jx_k^T v  (x-rows of stage k; v indexed by slot) */
static void jxT_apply(const ws_t *w, int k, const double *x, const double *v, double *out)
{
    if (k >= 1)
        for (int s = 0; s < w->nxb; s++) out[w->bidx[s]] += v[w->o_xl + s] - v[w->o_xu + s];
    if (k >= 1 && k <= w->N - 1) {
        for (int pq = 0; pq < w->M; pq++) {
            int i = w->pi[pq], j = w->pj[pq];
            double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], z = v[w->o_pr + pq];
            out[3 * i] += 2 * dx * z; out[3 * i + 1] += 2 * dy * z; out[3 * j] -= 2 * dx * z; out[3 * j + 1] -= 2 * dy * z;
        }
        for (int i = 0; i < w->m; i++)
            for (int o = 0; o < w->K; o++) {
                double dx = x[3 * i] - w->obs[3 * o], dy = x[3 * i + 1] - w->obs[3 * o + 1];
                double rr = sqrt(dx * dx + dy * dy), z = v[w->o_ob + i * w->K + o];
                out[3 * i] += dx / rr * z; out[3 * i + 1] += dy / rr * z;
            }
    }
}

/* out[slot] = (Jx_k d)[slot] for x-rows, (Ju_k du)[slot] for u-rows */
static void j_apply(const ws_t *w, int k, const double *x, const double *dx_, const double *du, double *out)
{
    for (int s = 0; s < w->nh; s++) out[s] = 0.0;
    if (k < w->N)
        for (int c = 0; c < w->nu; c++) { out[w->o_ul + c] = du[c]; out[w->o_uu + c] = -du[c]; }
    if (k >= 1)
        for (int s = 0; s < w->nxb; s++) { double v = dx_[w->bidx[s]]; out[w->o_xl + s] = v; out[w->o_xu + s] = -v; }
    if (k >= 1 && k <= w->N - 1) {
        for (int pq = 0; pq < w->M; pq++) {
            int i = w->pi[pq], j = w->pj[pq];
            double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1];
            out[w->o_pr + pq] = 2 * dx * (dx_[3 * i] - dx_[3 * j]) + 2 * dy * (dx_[3 * i + 1] - dx_[3 * j + 1]);
        }
        for (int i = 0; i < w->m; i++)
            for (int o = 0; o < w->K; o++) {
                double dx = x[3 * i] - w->obs[3 * o], dy = x[3 * i + 1] - w->obs[3 * o + 1];
                double rr = sqrt(dx * dx + dy * dy);
                out[w->o_ob + i * w->K + o] = (dx * dx_[3 * i] + dy * dx_[3 * i + 1]) / rr;
            }
    }
}

/* defects, inequality values, trig cache and objective at (X,U) */
static double eval_point(const ws_t *w, const double *xs, const double *X, const double *U, double *sn, double *cs, double *C,
                         double *H)
{
    int nx = w->nx, nu = w->nu, N = w->N, m = w->m;
    double f = 0.0;
    for (int k = 0; k < N; k++) {
        const double *x = X + (size_t)k * nx, *xn = x + nx, *u = U + (size_t)k * nu;
        for (int i = 0; i < m; i++) {
            double s = sin(x[3 * i + 2]), c = cos(x[3 * i + 2]), v = u[2 * i], om = u[2 * i + 1];
            sn[k * m + i] = s; cs[k * m + i] = c;
            C[k * nx + 3 * i] = xn[3 * i] - (x[3 * i] + w->T * v * c);
            C[k * nx + 3 * i + 1] = xn[3 * i + 1] - (x[3 * i + 1] + w->T * v * s);
            C[k * nx + 3 * i + 2] = xn[3 * i + 2] - (x[3 * i + 2] + w->T * om);
        }
        for (int c = 0; c < nx; c++) { double e = x[c] - xs[c]; f += w->qd[c] * e * e; }
        for (int c = 0; c < nu; c++) f += w->rd[c] * u[c] * u[c];
    }
    for (int k = 0; k <= N; k++) stage_h(w, k, X + (size_t)k * nx, k < N ? U + (size_t)k * nu : NULL, H + (size_t)k * w->nh);
    return f;
}

/* Elastic phase (round 4; the last resort where IPOPT would run its restoration phase).  The pair and obstacle rows h(x) >= 0 become
       h(x) + t - s = 0,  s >= 0,  t >= 0,   objective + rho sum t        (rho = NMPC_ELASTIC_RHO)
   with both s and t under the barrier; the dynamics stay equalities (any control sequence has a trajectory: only the inequality rows can make
   the problem locally infeasible), so the Riccati structure is untouched.  Stationarity gives the row's dual z in (0, rho) and z_t = rho - z.
   Newton step of one row with D = s/z + t/(rho - z) and q = h - mu/z + mu/(rho - z):
       dz = -(J dx + q) / D,   ds = (mu - s z - s dz) / z,   dt = (mu - t (rho - z) + t dz) / (rho - z)
   i.e. in the condensed stage blocks Sigma = 1/D replaces z/s and v = z - q/D replaces mu/s - Sigma (h - s) (both reduce to the plain forms
   for t -> 0, rho -> inf).  A converged elastic solve solves the NLP iff every t has closed (<= NMPC_X0_TOL); otherwise it is a stationary
   point of the infeasibility and is reported as NMPC_STATUS_STALLED.  Measured on the 42 captured failures / fixtures of the composite
   (tools): 7 failed -> 0 with this phase in place of the second cold retry; closed-loop soaks 11 failed of 122,880 -> 2. */
#define ELS(w, s) ((w)->el && (s) >= (w)->o_pr)      /* elastic slot */
static double barrier_and_infeas_t(const ws_t *w, double f, const double *C, const double *H, const double *S, const double *Tv, double mu,
                                   double *theta)
{
    double lg = 0.0, th = 0.0, pen = 0.0;
    for (int i = 0; i < w->N * w->nx; i++) th += fabs(C[i]);
    for (int k = 0; k <= w->N; k++)
        for (int s = 0; s < w->nh; s++)
            if (slot_active(w, k, s)) {
                size_t o = (size_t)k * w->nh + s;
                if (ELS(w, s)) { lg += log(S[o]) + log(Tv[o]); th += fabs(H[o] + Tv[o] - S[o]); pen += w->rho * Tv[o]; }
                else { lg += log(S[o]); th += fabs(H[o] - S[o]); }
            }
    *theta = th;
    return f + pen - mu * lg;
}
static double barrier_and_infeas(const ws_t *w, double f, const double *C, const double *H, const double *S, double mu, double *theta)
{
    return barrier_and_infeas_t(w, f, C, H, S, w->El, mu, theta);
}

/* Cholesky of the leading n x n of a (row-major, ld), in place (lower). 0 = ok, 1 = not positive definite */
static int chol(double *a, int n, int ld)
{
    for (int j = 0; j < n; j++) {
        double d0 = a[j * ld + j], d = d0;
        for (int t = 0; t < j; t++) d -= a[j * ld + t] * a[j * ld + t];
        if (!(d > 1e-9 * fabs(d0)) || !(d > 0.0)) return 1;
        d = sqrt(d); a[j * ld + j] = d;
        for (int i = j + 1; i < n; i++) {
            double v = a[i * ld + j];
            for (int t = 0; t < j; t++) v -= a[i * ld + t] * a[j * ld + t];
            a[i * ld + j] = v / d;
        }
    }
    return 0;
}

/* EXPERIMENT (NMPC_ORACLE_STAGE_REG = 2 | 3): Cholesky whose rejected pivots are repaired in place, i.e. the factorisation of
   a + diag(e) with e_j > 0 only on the rejected pivots — a stage-local inertia correction that needs no second sweep.
   mode 2: d <- max(|d|, beta max(1, |d0|))  (curvature flip);  mode 3: d <- d + delta with IPOPT's escalation schedule run on the
   scalar pivot, delta remembered per (stage, control) in dl / ns.  *nmod counts repaired pivots. */
static int chol_mod(double *a, int n, int ld, int mode, double beta, double *dl, int *ns, int *nmod)
{
    for (int j = 0; j < n; j++) {
        double d0 = a[j * ld + j], d = d0;
        for (int t = 0; t < j; t++) d -= a[j * ld + t] * a[j * ld + t];
        if (mode == 2) {
            if (!(d > 1e-9 * fabs(d0)) || !(d > 0.0)) {
                if (!(d == d)) return 1;
                d = fmax(fabs(d), beta * fmax(1.0, fabs(d0))); (*nmod)++;
            }
        } else {
            double del = ns[j] ? fmax(1e-20, 0.25 * dl[j]) : 0.0;
            int ntry = 0;
            if (!(d == d)) return 1;
            while (!(d + del > 1e-9 * fabs(d0 + del)) || !(d + del > 0.0)) {
                ntry++;
                if (del == 0.0) del = (dl[j] == 0.0) ? 1e-4 : fmax(1e-20, dl[j] / 3.0);
                else del *= (dl[j] == 0.0) ? 100.0 : NMPC_SHIFT_ESCALATION;
                if (del > 1e20) return 1;
            }
            if (del > 0.0) { dl[j] = del; (*nmod)++; }
            ns[j] = del > 0.0 && (ntry > 0 || del > 1e-6);
            d += del;
        }
        d = sqrt(d); a[j * ld + j] = d;
        for (int i = j + 1; i < n; i++) {
            double v = a[i * ld + j];
            for (int t = 0; t < j; t++) v -= a[i * ld + t] * a[j * ld + t];
            a[i * ld + j] = v / d;
        }
    }
    return 0;
}

/* EXPERIMENT (NMPC_ORACLE_STAGE_REG = 5): the global shift escalated IN PLACE.  *delta is the running shift of the sweep: it enters every
   pivot from here on (this stage's remaining ones and all later stages of the backward sweep); a rejected pivot escalates it on the spot
   with IPOPT's schedule (dlast = the shift the previous iteration ended with) — pivots already taken keep the smaller shift they were taken
   with.  The factorisation is that of Quu + diag(e), e non-decreasing along the sweep: no stage is ever factored twice. */
static int chol_run(double *a, int n, int ld, double *delta, double dlast, int *ntry)
{
    for (int j = 0; j < n; j++) {
        double d0 = a[j * ld + j], d = d0;
        for (int t = 0; t < j; t++) d -= a[j * ld + t] * a[j * ld + t];
        if (!(d == d)) return 1;
        double del = *delta;
        while (!(d + del > 1e-9 * fabs(d0 + del)) || !(d + del > 0.0)) {
            (*ntry)++;
            if (del == 0.0) del = (dlast == 0.0) ? 1e-4 : fmax(1e-20, dlast / 3.0);
            else del *= (dlast == 0.0) ? 100.0 : NMPC_SHIFT_ESCALATION;
            if (del > 1e20) return 1;
        }
        *delta = del;
        d = sqrt(d + del); a[j * ld + j] = d;
        for (int i = j + 1; i < n; i++) {
            double v = a[i * ld + j];
            for (int t = 0; t < j; t++) v -= a[i * ld + t] * a[j * ld + t];
            a[i * ld + j] = v / d;
        }
    }
    return 0;
}

/* experiment statistics (development aid): stage factorisations, interior-point iterations, repaired pivots */
static long long g_stat_fact = 0, g_stat_iter = 0, g_stat_mod = 0, g_stat_stages = 0;
void nmpc_oracle_stats(double *out, int reset)
{
    out[0] = (double)g_stat_fact; out[1] = (double)g_stat_iter; out[2] = (double)g_stat_mod; out[3] = (double)g_stat_stages;
    if (reset) { g_stat_fact = g_stat_iter = g_stat_mod = g_stat_stages = 0; }
}

/* (Re)start of the barrier iteration from the current primal point, or — cold != 0 — from the reference's cold start
   X_k = x0, U = 0 (C6:398-400): the point goes strictly inside the simple bounds, slacks onto the constraint values, duals mu/s,
   multipliers 0.  Returns the objective. */
static double barrier_restart(ws_t *w, const double *xs, double mu, int cold)
{
    const int nx = w->nx, nu = w->nu, N = w->N, nh = w->nh;
    const double bp = 1e-2;
    if (cold) {
        for (int k = 1; k <= N; k++) memcpy(w->X + (size_t)k * nx, w->X, sizeof(double) * nx);
        memset(w->U, 0, sizeof(double) * (size_t)N * nu);
    }
    for (int k = 0; k < N; k++)
        for (int c = 0; c < nu; c++) {
            double lo = w->lbu[c], hi = w->ubu[c];
            double pu = fmin(bp * fmax(1.0, fabs(lo)), bp * (hi - lo));
            double *u = &w->U[(size_t)k * nu + c];
            *u = fmin(fmax(*u, lo + pu), hi - pu);
        }
    for (int k = 1; k <= N; k++)
        for (int s = 0; s < w->nxb; s++) {
            double b = w->bxs[s], px = fmin(bp * fmax(1.0, b), bp * 2.0 * b);
            double *x = &w->X[(size_t)k * nx + w->bidx[s]];
            *x = fmin(fmax(*x, -b + px), b - px);
        }
    double f = eval_point(w, xs, w->X, w->U, w->sn, w->cs, w->C, w->H);
    for (int k = 0; k <= N; k++)
        for (int s = 0; s < nh; s++) {
            size_t o = (size_t)k * nh + s;
            if (!slot_active(w, k, s)) { w->S[o] = 1.0; w->Z[o] = 0.0; continue; }
            double floor_ = (s < w->o_xl) ? 1e-12 : bp;
            if (ELS(w, s)) {      /* elastic row: t absorbs the violation, s = h + t >= bp, 0 < z < rho */
                w->El[o] = fmax(bp, bp - w->H[o]);
                w->S[o] = w->H[o] + w->El[o];
                w->Z[o] = fmin(mu / w->S[o], 0.5 * w->rho);
                continue;
            }
            w->S[o] = fmax(w->H[o], floor_);
            w->Z[o] = mu / w->S[o];
        }
    memset(w->lam, 0, sizeof(double) * (size_t)(N + 1) * nx);
    return f;
}

static int solve_one(ws_t *w, const double *p, const double *w0, double *wout, double *obj_out, int *iters_out, double *kkt_out)
{
    const int nx = w->nx, nu = w->nu, N = w->N, m = w->m, nh = w->nh;
    const double *x0p = p, *xs = p + nx;
    const double T = w->T;
    int status = NMPC_STATUS_MAX_ITER;
    memcpy(w->X, w0, sizeof(double) * (size_t)(N + 1) * nx);
    memcpy(w->U, w0 + (size_t)(N + 1) * nx, sizeof(double) * (size_t)N * nu);
    memcpy(w->X, x0p, sizeof(double) * nx);

    /* stage-0 pair / obstacle rows act on the pinned state: feasibility pre-check */
    int infeasible = 0;
    for (int pq = 0; pq < w->M; pq++) {
        int i = w->pi[pq], j = w->pj[pq];
        double dx = x0p[3 * i] - x0p[3 * j], dy = x0p[3 * i + 1] - x0p[3 * j + 1];
        if (dx * dx + dy * dy < w->dmin2 - NMPC_X0_TOL) infeasible = 1;
    }
    for (int i = 0; i < m; i++)
        for (int o = 0; o < w->K; o++) {
            double dx = x0p[3 * i] - w->obs[3 * o], dy = x0p[3 * i + 1] - w->obs[3 * o + 1];
            if (sqrt(dx * dx + dy * dy) - w->robdim - w->obs[3 * o + 2] < w->margin - NMPC_X0_TOL) infeasible = 1;
        }
    if (infeasible) {
        memcpy(wout, w->X, sizeof(double) * (size_t)(N + 1) * nx);
        memcpy(wout + (size_t)(N + 1) * nx, w->U, sizeof(double) * (size_t)N * nu);
        *obj_out = NAN; *iters_out = 0; *kkt_out = INFINITY;
        return NMPC_STATUS_INFEASIBLE_X0;
    }

    memset(w->dl_k, 0, sizeof(double) * (size_t)N * nu); memset(w->ns_k, 0, sizeof(int) * (size_t)N * nu);
    w->el = 0;
    double mu = w->mu_init;
    double f = barrier_restart(w, xs, mu, 0);      /* push inside the simple bounds (IPOPT bound_push = bound_frac = 1e-2), slacks, duals */
    double delta_last = 0.0, nu_pen = 1.0, kkt = INFINITY;
    double mh0 = 0.0, mh1 = 0.0, mh2 = 0.0, mh_mu = -1.0, mh_nu = -1.0;   /* merit values of the last three iterates (same mu, nu) */
    int mcount = 0;
    int it = 0, need_shift = 0, n_tiny = 0, n_restart = 0;
    /* filter of the IPOPT-faithful variant (w->ipopt): pairs (theta, phi) no trial point may be dominated by; reset with every barrier problem */
    enum { FMAX = 512 };
    double flt_th[FMAX], flt_ph[FMAX], flt_mu = -1.0, th_init = -1.0;
    int nflt = 0;
    /* Cold-start retry: a solve that stalls after its barrier restarts, fails numerically, or is still iterating after
       NMPC_COLD_RETRY_ITERS iterations is restarted from the reference's own cold start X_k = x0, U = 0 (C6:398-400) instead
       of the caller's guess — at most twice, the second time with mu = 10 mu_init (of the 5 composite solves the first retry does
       not rescue, all converge from the cold start with another initial barrier parameter: 0.1, 2 and 5 were tried).  This is the restoration of last resort: a warm start shifted from the previous period can sit in a
       region from which the iteration converges to an infeasible stationary point or cycles; from the cold start all captured
       failures of the six-robot + eight-obstacle closed loop converge (13 of 10,240 solves, tests/golden/cold_retry_cases.npz). */
    /* fixture generation: NMPC_ORACLE_NO_COLD_RETRY=1 captures the failures the retry rescues, NMPC_ORACLE_MAX_COLD=1 those the FIRST retry does not rescue */
    const int max_cold = (getenv("NMPC_ORACLE_NO_COLD_RETRY") || w->ipopt) ? 0 : (getenv("NMPC_ORACLE_MAX_COLD") ? atoi(getenv("NMPC_ORACLE_MAX_COLD")) : NMPC_COLD_RETRIES);
    int n_cold = 0;
    int it_base = 0;      /* iteration at which the current attempt started (watchdog reference) */
#define COLD_RETRY()                                                                                                              \
    do {                                                                                                                          \
        n_cold++; it_base = it; w->el = n_cold >= 2; mu = w->mu_init; f = barrier_restart(w, xs, mu, 1);                          \
        delta_last = 0.0; nu_pen = 1.0; need_shift = 0; mcount = 0; n_tiny = 0; n_restart = 0;                                    \
        memset(w->dl_k, 0, sizeof(double) * (size_t)N * nu); memset(w->ns_k, 0, sizeof(int) * (size_t)N * nu);                    \
    } while (0)
    int n_ineq = 0;
    for (int k = 0; k <= N; k++) for (int s = 0; s < nh; s++) n_ineq += slot_active(w, k, s);
    double tmp[NXM + NUM_], Jd[4 * NUM_ + 2 * NXM + 64 + NMPC_MAX_ROBOTS * NMPC_MAX_OBSTACLES];

    for (;;) {
        /* ---- stationarity residual and optimality error (IPOPT eq. 5) */
        double e_d = 0.0, e_c = 0.0, e_h = 0.0, zsum = 0.0, lsum = 0.0;
        for (int k = 1; k <= N; k++) {
            const double *x = w->X + (size_t)k * nx;
            double r[NXM];
            for (int c = 0; c < nx; c++) r[c] = w->lam[(size_t)k * nx + c];
            if (k < N) {
                const double *ln = w->lam + (size_t)(k + 1) * nx, *u = w->U + (size_t)k * nu;
                for (int i = 0; i < m; i++) {
                    double a = -T * u[2 * i] * w->sn[k * m + i], b = T * u[2 * i] * w->cs[k * m + i];
                    r[3 * i] += 2 * w->qd[3 * i] * (x[3 * i] - xs[3 * i]) - ln[3 * i];
                    r[3 * i + 1] += 2 * w->qd[3 * i + 1] * (x[3 * i + 1] - xs[3 * i + 1]) - ln[3 * i + 1];
                    r[3 * i + 2] += 2 * w->qd[3 * i + 2] * (x[3 * i + 2] - xs[3 * i + 2]) - (ln[3 * i + 2] + a * ln[3 * i] + b * ln[3 * i + 1]);
                }
            }
            for (int c = 0; c < nx; c++) tmp[c] = 0.0;
            jxT_apply(w, k, x, w->Z + (size_t)k * nh, tmp);
            for (int c = 0; c < nx; c++) { r[c] -= tmp[c]; e_d = fmax(e_d, fabs(r[c])); }
        }
        for (int k = 0; k < N; k++) {
            const double *ln = w->lam + (size_t)(k + 1) * nx, *u = w->U + (size_t)k * nu, *z = w->Z + (size_t)k * nh;
            for (int i = 0; i < m; i++) {
                double c = w->cs[k * m + i], s = w->sn[k * m + i];
                double rv = 2 * w->rd[2 * i] * u[2 * i] - T * (c * ln[3 * i] + s * ln[3 * i + 1]) - (z[w->o_ul + 2 * i] - z[w->o_uu + 2 * i]);
                double rw = 2 * w->rd[2 * i + 1] * u[2 * i + 1] - T * ln[3 * i + 2] - (z[w->o_ul + 2 * i + 1] - z[w->o_uu + 2 * i + 1]);
                e_d = fmax(e_d, fmax(fabs(rv), fabs(rw)));
            }
        }
        for (int i = 0; i < N * nx; i++) e_c = fmax(e_c, fabs(w->C[i]));
        for (int i = nx; i < (N + 1) * nx; i++) lsum += fabs(w->lam[i]);
        double cmp0 = 0.0;
        for (int k = 0; k <= N; k++)
            for (int s = 0; s < nh; s++)
                if (slot_active(w, k, s)) {
                    size_t o = (size_t)k * nh + s;
                    if (ELS(w, s)) { e_h = fmax(e_h, fabs(w->H[o] + w->El[o] - w->S[o])); cmp0 = fmax(cmp0, w->El[o] * (w->rho - w->Z[o])); }
                    else e_h = fmax(e_h, fabs(w->H[o] - w->S[o]));
                    zsum += w->Z[o];
                    cmp0 = fmax(cmp0, w->S[o] * w->Z[o]);
                }
        const double smax = 100.0;
        double s_d = fmax(smax, (lsum + zsum) / (double)(N * nx + n_ineq)) / smax;
        double s_c = fmax(smax, zsum / (double)(n_ineq > 0 ? n_ineq : 1)) / smax;
        double E0 = fmax(fmax(e_d / s_d, e_c), fmax(e_h, cmp0 / s_c));
        kkt = E0;
        if (!(E0 == E0)) { if (n_cold < max_cold && it < w->max_iter) { COLD_RETRY(); continue; } status = NMPC_STATUS_NUMERIC; break; }
        if (E0 <= w->tol) {
            status = NMPC_STATUS_CONVERGED;
            if (w->el) {      /* the penalty problem's solution solves the NLP only if every elastic variable has closed */
                double tmax = 0.0;
                for (int k = 0; k <= N; k++) for (int s = w->o_pr; s < nh; s++) if (slot_active(w, k, s)) tmax = fmax(tmax, w->El[(size_t)k * nh + s]);
                if (tmax > NMPC_X0_TOL) status = NMPC_STATUS_STALLED;
            }
            break;
        }
        if (it >= w->max_iter) { status = NMPC_STATUS_MAX_ITER; break; }
        if (n_cold < max_cold && it - it_base >= NMPC_COLD_RETRY_ITERS) { COLD_RETRY(); continue; }
        /* ---- monotone barrier update (IPOPT eq. 7) */
        const double mu_min = w->tol / 10.0;
        for (;;) {
            double cm = 0.0;
            for (int k = 0; k <= N; k++)
                for (int s = 0; s < nh; s++)
                    if (slot_active(w, k, s)) { size_t o = (size_t)k * nh + s; cm = fmax(cm, fabs(w->S[o] * w->Z[o] - mu)); if (ELS(w, s)) cm = fmax(cm, fabs(w->El[o] * (w->rho - w->Z[o]) - mu)); }
            double Emu = fmax(fmax(e_d / s_d, e_c), fmax(e_h, cm / s_c));
            if (mu > mu_min && Emu <= 10.0 * mu) mu = fmax(mu_min, fmin(0.2 * mu, pow(mu, 1.5)));
            else break;
        }
        double tau = fmax(0.99, 1.0 - mu);

        int pc_pass = 0;
        double pc_ap = 1.0, pc_ad = 1.0, pc_mm = 0.0;
        const double mu_keep = mu;
        if (w->pc == 2) mu = 0.0;          /* EXPERIMENT: Mehrotra's predictor is the affine-scaling step */
    pc_again:
        /* ---- condensed stage blocks: Hxx, gx (k=1..N), diagonal huu, gu, cross term hvt (k=0..N-1) */
        for (int k = 0; k <= N; k++) {
            const double *x = w->X + (size_t)k * nx, *s_ = w->S + (size_t)k * nh, *z = w->Z + (size_t)k * nh, *h = w->H + (size_t)k * nh;
            double *Hk = w->Hxx + (size_t)k * nx * nx, *g = w->gx + (size_t)k * nx;
            memset(Hk, 0, sizeof(double) * nx * nx);
            for (int c = 0; c < nx; c++) g[c] = 0.0;
            if (k >= 1) {
                if (k < N)
                    for (int c = 0; c < nx; c++) { Hk[c * nx + c] = 2 * w->qd[c]; g[c] = 2 * w->qd[c] * (x[c] - xs[c]); }
                /* v = mu/s - sigma (h - s) per slot; g -= Jx^T v; Hxx += Jx^T Sigma Jx - z * hess(h) */
                double v[4 * NUM_ + 2 * NXM + 64 + NMPC_MAX_ROBOTS * NMPC_MAX_OBSTACLES];
                double sige[4 * NUM_ + 2 * NXM + 64 + NMPC_MAX_ROBOTS * NMPC_MAX_OBSTACLES];      /* Sigma per slot (elastic rows: 1 / (s/z + t/(rho-z))) */
                for (int s = 0; s < nh; s++) {
                    if (!slot_active(w, k, s)) { v[s] = 0.0; sige[s] = 0.0; continue; }
                    if (ELS(w, s)) {
                        const double t_ = w->El[(size_t)k * nh + s], D_ = s_[s] / z[s] + t_ / (w->rho - z[s]);
                        const double q_ = h[s] - mu / z[s] + mu / (w->rho - z[s]);
                        sige[s] = 1.0 / D_; v[s] = z[s] - q_ / D_;
                    } else { sige[s] = z[s] / s_[s]; v[s] = mu / s_[s] - z[s] / s_[s] * (h[s] - s_[s]) - w->corr[(size_t)k * nh + s]; }
                }
                for (int c = 0; c < nx; c++) tmp[c] = 0.0;
                jxT_apply(w, k, x, v, tmp);
                for (int c = 0; c < nx; c++) g[c] -= tmp[c];
                for (int s = 0; s < w->nxb; s++) {
                    int c = w->bidx[s];
                    Hk[c * nx + c] += z[w->o_xl + s] / s_[w->o_xl + s] + z[w->o_xu + s] / s_[w->o_xu + s];
                }
                if (k <= N - 1) {
                    for (int pq = 0; pq < w->M; pq++) {
                        int i = w->pi[pq], j = w->pj[pq];
                        double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1];
                        double sg = sige[w->o_pr + pq], zz = z[w->o_pr + pq];
                        double e00 = 4 * sg * dx * dx - 2 * zz, e01 = 4 * sg * dx * dy, e11 = 4 * sg * dy * dy - 2 * zz;
                        int a = 3 * i, b = 3 * j;
                        Hk[a * nx + a] += e00; Hk[a * nx + a + 1] += e01; Hk[(a + 1) * nx + a] += e01; Hk[(a + 1) * nx + a + 1] += e11;
                        Hk[b * nx + b] += e00; Hk[b * nx + b + 1] += e01; Hk[(b + 1) * nx + b] += e01; Hk[(b + 1) * nx + b + 1] += e11;
                        Hk[a * nx + b] -= e00; Hk[a * nx + b + 1] -= e01; Hk[(a + 1) * nx + b] -= e01; Hk[(a + 1) * nx + b + 1] -= e11;
                        Hk[b * nx + a] -= e00; Hk[b * nx + a + 1] -= e01; Hk[(b + 1) * nx + a] -= e01; Hk[(b + 1) * nx + a + 1] -= e11;
                    }
                    for (int i = 0; i < m; i++)
                        for (int o = 0; o < w->K; o++) {
                            double dx = x[3 * i] - w->obs[3 * o], dy = x[3 * i + 1] - w->obs[3 * o + 1];
                            double rr = sqrt(dx * dx + dy * dy), n0 = dx / rr, n1 = dy / rr;
                            int sl = w->o_ob + i * w->K + o;
                            double sg = sige[sl], zz = z[sl] / rr;
                            int a = 3 * i;
                            Hk[a * nx + a] += sg * n0 * n0 - zz * (1 - n0 * n0);
                            Hk[a * nx + a + 1] += sg * n0 * n1 + zz * n0 * n1;
                            Hk[(a + 1) * nx + a] += sg * n0 * n1 + zz * n0 * n1;
                            Hk[(a + 1) * nx + a + 1] += sg * n1 * n1 - zz * (1 - n1 * n1);
                        }
                }
            }
            if (k < N) {
                const double *u = w->U + (size_t)k * nu, *ln = w->lam + (size_t)(k + 1) * nx;
                for (int c = 0; c < nu; c++) {
                    double sl = s_[w->o_ul + c], su = s_[w->o_uu + c], zl = z[w->o_ul + c], zu = z[w->o_uu + c];
                    w->huu[k * nu + c] = 2 * w->rd[c] + zl / sl + zu / su;
                    double vl = mu / sl - zl / sl * (h[w->o_ul + c] - sl) - w->corr[(size_t)k * nh + w->o_ul + c], vu = mu / su - zu / su * (h[w->o_uu + c] - su) - w->corr[(size_t)k * nh + w->o_uu + c];
                    w->gu[k * nu + c] = 2 * w->rd[c] * u[c] - (vl - vu);
                }
                for (int i = 0; i < m; i++) {
                    double c = w->cs[k * m + i], s = w->sn[k * m + i], lx = ln[3 * i], ly = ln[3 * i + 1];
                    if (k >= 1) Hk[(3 * i + 2) * nx + 3 * i + 2] += T * u[2 * i] * (lx * c + ly * s);
                    w->hvt[k * m + i] = T * (lx * s - ly * c);
                }
            }
        }

        /* ---- Riccati sweep with inertia correction (IPOPT alg. IC) and partial re-factorisation.
           First trial: delta = 0 (IPOPT), except right after an iteration that needed a shift — then a quarter of that shift is tried
           directly.  A rejected pivot at stage k escalates the shift with IPOPT's schedule and resumes the sweep a few stages ABOVE k, from
           the nearest saved cost-to-go at or beyond k + NMPC_REFACTOR_BACK (saved every NMPC_CKPT_EVERY stages, include/nmpc_constants.h) — not
           from stage N-1 as in rounds 1-3.  The stages above the resume point keep the factorisation they have, i.e. the smaller shift: the
           matrix factored is W + diag(delta_k) with delta_k non-decreasing along the sweep, every Quu_k > 0 — a correct-inertia KKT system.
           Measured on the bench batches (tools/sreg_experiment.py): stage factorisations per iteration 1.28 -> 1.08 sweep
           equivalents with 4 % FEWER iterations (six robots), where resuming AT the failing stage costs +40 % iterations. */
        double delta = need_shift ? fmax(1e-20, 0.25 * delta_last) : 0.0;
        int ntry = 0, ok = 0, st_fact = 0, st_mod = 0;
        double P[NXM * NXM], pv[NXM], G[NXM * (NXM + NUM_)], Qxx[NXM * NXM], Qux[NUM_ * NXM], Quu[NUM_ * NUM_], Pb[NXM], qx[NXM], qu[NUM_];
        for (;;) {
            ok = 1;
            memcpy(P, w->Hxx + (size_t)N * nx * nx, sizeof(double) * nx * nx);
            for (int c = 0; c < nx; c++) pv[c] = w->gx[(size_t)N * nx + c];   /* the inertia shift acts on the controls only */
            for (int k = N - 1; k >= 0; k--) {
                const double *u = w->U + (size_t)k * nu, *Ck = w->C + (size_t)k * nx;
                const int nz = nx + nu;
                if (w->sreg < 0 && (N - 1 - k) % w->sreg_C == 0) {      /* checkpoint: the cost-to-go entering stage k */
                    memcpy(w->Psnap + (size_t)(k + 1) * nx * nx, P, sizeof(double) * nx * nx); memcpy(w->pvsnap + (size_t)(k + 1) * nx, pv, sizeof(double) * nx);
                }
                /* Pb = p + P b, b = -c_k */
                for (int r = 0; r < nx; r++) { double a = pv[r]; for (int c = 0; c < nx; c++) a -= P[r * nx + c] * Ck[c]; Pb[r] = a; }
                /* G = P [A B]  (A = I + sparse, B sparse: per-robot column combinations) */
                for (int r = 0; r < nx; r++) {
                    const double *Pr = P + r * nx; double *Gr = G + r * nz;
                    for (int i = 0; i < m; i++) {
                        double c = w->cs[k * m + i], s = w->sn[k * m + i], a = -T * u[2 * i] * s, b = T * u[2 * i] * c;
                        Gr[3 * i] = Pr[3 * i]; Gr[3 * i + 1] = Pr[3 * i + 1];
                        Gr[3 * i + 2] = Pr[3 * i + 2] + a * Pr[3 * i] + b * Pr[3 * i + 1];
                        Gr[nx + 2 * i] = T * (c * Pr[3 * i] + s * Pr[3 * i + 1]);
                        Gr[nx + 2 * i + 1] = T * Pr[3 * i + 2];
                    }
                }
                /* [Qxx Qxu; Qux Quu] = [A B]^T G + H ; q = g + [A B]^T Pb */
                for (int i = 0; i < m; i++) {
                    double c = w->cs[k * m + i], s = w->sn[k * m + i], a = -T * u[2 * i] * s, b = T * u[2 * i] * c;
                    const double *Gx = G + (3 * i) * nz, *Gy = G + (3 * i + 1) * nz, *Gt = G + (3 * i + 2) * nz;
                    for (int cc = 0; cc < nx; cc++) {
                        Qxx[(3 * i) * nx + cc] = Gx[cc]; Qxx[(3 * i + 1) * nx + cc] = Gy[cc];
                        Qxx[(3 * i + 2) * nx + cc] = Gt[cc] + a * Gx[cc] + b * Gy[cc];
                        Qux[(2 * i) * nx + cc] = T * (c * Gx[cc] + s * Gy[cc]);
                        Qux[(2 * i + 1) * nx + cc] = T * Gt[cc];
                    }
                    for (int cc = 0; cc < nu; cc++) {
                        Quu[(2 * i) * nu + cc] = T * (c * Gx[nx + cc] + s * Gy[nx + cc]);
                        Quu[(2 * i + 1) * nu + cc] = T * Gt[nx + cc];
                    }
                    qx[3 * i] = Pb[3 * i]; qx[3 * i + 1] = Pb[3 * i + 1]; qx[3 * i + 2] = Pb[3 * i + 2] + a * Pb[3 * i] + b * Pb[3 * i + 1];
                    qu[2 * i] = T * (c * Pb[3 * i] + s * Pb[3 * i + 1]); qu[2 * i + 1] = T * Pb[3 * i + 2];
                }
                for (int c = 0; c < nu; c++) { Quu[c * nu + c] += w->huu[k * nu + c] + delta; qu[c] += w->gu[k * nu + c]; }
                for (int i = 0; i < m; i++) Qux[(2 * i) * nx + 3 * i + 2] += w->hvt[k * m + i];
                st_fact++;
                if (w->sreg < 0) {
                    if (chol(Quu, nu, nu)) {
                        ntry++; st_mod++;
                        if (delta == 0.0) delta = (delta_last == 0.0) ? 1e-4 : fmax(1e-20, delta_last / 3.0);
                        else delta *= (delta_last == 0.0) ? 100.0 : NMPC_SHIFT_ESCALATION;
                        if (delta > 1e20) { ok = 0; break; }
                        int kr = k + w->sreg_J < N - 1 ? k + w->sreg_J : N - 1;
                        kr += (N - 1 - kr) % w->sreg_C;      /* the nearest checkpointed stage at or above */
                        memcpy(P, w->Psnap + (size_t)(kr + 1) * nx * nx, sizeof(double) * nx * nx); memcpy(pv, w->pvsnap + (size_t)(kr + 1) * nx, sizeof(double) * nx);
                        k = kr + 1;
                        continue;
                    }
                } else if (w->sreg == 5) {      /* EXPERIMENT: the shift escalated in place, no stage factored twice */
                    double tot = delta; int nt0 = ntry;
                    for (int c = 0; c < nu; c++) Quu[c * nu + c] -= delta;      /* chol_run adds the running total itself */
                    if (chol_run(Quu, nu, nu, &tot, delta_last, &ntry)) { ok = 0; break; }
                    if (ntry > nt0) st_mod++;
                    delta = tot;
                } else if (w->sreg >= 2) {      /* EXPERIMENT: rejected pivots repaired in place (2: flip, 3: per-pivot shift with memory) */
                    if (chol_mod(Quu, nu, nu, w->sreg, w->sreg_beta, w->dl_k + (size_t)k * nu, w->ns_k + (size_t)k * nu, &st_mod)) { ok = 0; break; }
                } else if (w->sreg == 1) {      /* EXPERIMENT: stage-local shift delta_k I with its own memory, the stage redone from P_{k+1} (untouched here) */
                    double *dlk = w->dl_k + (size_t)k * nu; int *nsk = w->ns_k + (size_t)k * nu;   /* slot 0 of the stage carries its memory */
                    double dk = nsk[0] ? fmax(1e-20, 0.25 * dlk[0]) : 0.0;
                    int nt = 0, bad = 0;
                    double Qs[NUM_ * NUM_];
                    memcpy(Qs, Quu, sizeof(double) * nu * nu);
                    for (;;) {
                        for (int c = 0; c < nu; c++) Quu[c * nu + c] += dk;
                        if (!chol(Quu, nu, nu)) break;
                        nt++; st_fact++;
                        if (dk == 0.0) dk = (dlk[0] == 0.0) ? 1e-4 : fmax(1e-20, dlk[0] / 3.0);
                        else dk *= (dlk[0] == 0.0) ? 100.0 : NMPC_SHIFT_ESCALATION;
                        if (dk > 1e20) { bad = 1; break; }
                        memcpy(Quu, Qs, sizeof(double) * nu * nu);
                    }
                    if (bad) { ok = 0; break; }
                    if (dk > 0.0) { dlk[0] = dk; st_mod++; }
                    nsk[0] = dk > 0.0 && (nt > 0 || dk > 1e-6);
                } else      /* NMPC_ORACLE_STAGE_REG=0: rounds 1-3, the whole sweep again from stage N-1 */
                if (chol(Quu, nu, nu)) { ok = 0; break; }
                /* Y = L^-1 Qux, y = L^-1 qu (forward substitution) */
                for (int r = 0; r < nu; r++) {
                    for (int t = 0; t < r; t++) {
                        double l = Quu[r * nu + t];
                        for (int c = 0; c < nx; c++) Qux[r * nx + c] -= l * Qux[t * nx + c];
                        qu[r] -= l * qu[t];
                    }
                    double d = Quu[r * nu + r];
                    for (int c = 0; c < nx; c++) Qux[r * nx + c] /= d;
                    qu[r] /= d;
                }
                if (k >= 1) {
                    const double *Hk = w->Hxx + (size_t)k * nx * nx;
                    for (int r = 0; r < nx; r++) {
                        for (int c = 0; c < nx; c++) {
                            double a = Qxx[r * nx + c] + Hk[r * nx + c];
                            for (int t = 0; t < nu; t++) a -= Qux[t * nx + r] * Qux[t * nx + c];
                            P[r * nx + c] = a;
                        }
                        double a = qx[r] + w->gx[(size_t)k * nx + r];
                        for (int t = 0; t < nu; t++) a -= Qux[t * nx + r] * qu[t];
                        pv[r] = a;
                    }
                    for (int i = 0; i < m; i++) {           /* cross term Hxu enters Qxx via (theta_i, v_i): handled through Qux above */
                        (void)i;
                    }
                    for (int r = 0; r < nx; r++) for (int c = r + 1; c < nx; c++) { double a = 0.5 * (P[r * nx + c] + P[c * nx + r]); P[r * nx + c] = a; P[c * nx + r] = a; }
                }
                /* K = -L^-T Y, kff = -L^-T y (back substitution) */
                double *Kk = w->Kg + (size_t)k * nu * nx, *kk = w->kff + (size_t)k * nu;
                for (int r = nu - 1; r >= 0; r--) {
                    double d = Quu[r * nu + r];
                    for (int c = 0; c < nx; c++) {
                        double a = Qux[r * nx + c];
                        for (int t = r + 1; t < nu; t++) a += Quu[t * nu + r] * Kk[t * nx + c];
                        Kk[r * nx + c] = -a / d;
                    }
                    double a = qu[r];
                    for (int t = r + 1; t < nu; t++) a += Quu[t * nu + r] * kk[t];
                    kk[r] = -a / d;
                }
            }
            if (ok) break;
            ntry++;
            if (delta == 0.0) delta = (delta_last == 0.0) ? 1e-4 : fmax(1e-20, delta_last / 3.0);
            else delta *= (delta_last == 0.0) ? 100.0 : NMPC_SHIFT_ESCALATION;
            if (delta > 1e20) break;
        }
        if (!ok) { if (n_cold < max_cold) { COLD_RETRY(); it++; continue; } status = NMPC_STATUS_NUMERIC; break; }
        if (delta > 0.0) delta_last = delta;
        need_shift = delta > 0.0 && (ntry > 0 || delta > 1e-6);
        {
#pragma omp atomic
            g_stat_fact += st_fact;
#pragma omp atomic
            g_stat_iter += 1;
#pragma omp atomic
            g_stat_mod += st_mod;
#pragma omp atomic
            g_stat_stages += N;
        }

        /* ---- forward sweep */
        for (int c = 0; c < nx; c++) w->dX[c] = 0.0;
        for (int k = 0; k < N; k++) {
            const double *dx = w->dX + (size_t)k * nx, *u = w->U + (size_t)k * nu;
            double *du = w->dU + (size_t)k * nu, *dxn = w->dX + (size_t)(k + 1) * nx;
            for (int r = 0; r < nu; r++) {
                double a = w->kff[k * nu + r];
                for (int c = 0; c < nx; c++) a += w->Kg[((size_t)k * nu + r) * nx + c] * dx[c];
                du[r] = a;
            }
            for (int i = 0; i < m; i++) {
                double c = w->cs[k * m + i], s = w->sn[k * m + i], a = -T * u[2 * i] * s, b = T * u[2 * i] * c;
                dxn[3 * i] = dx[3 * i] + a * dx[3 * i + 2] + T * c * du[2 * i] - w->C[k * nx + 3 * i];
                dxn[3 * i + 1] = dx[3 * i + 1] + b * dx[3 * i + 2] + T * s * du[2 * i] - w->C[k * nx + 3 * i + 1];
                dxn[3 * i + 2] = dx[3 * i + 2] + T * du[2 * i + 1] - w->C[k * nx + 3 * i + 2];
            }
        }
        /* ---- multipliers of the QP by the adjoint recursion */
        for (int k = N; k >= 1; k--) {
            const double *Hk = w->Hxx + (size_t)k * nx * nx, *dx = w->dX + (size_t)k * nx;
            double *l = w->lamn + (size_t)k * nx;
            for (int r = 0; r < nx; r++) {
                double a = w->gx[(size_t)k * nx + r];
                for (int c = 0; c < nx; c++) a += Hk[r * nx + c] * dx[c];
                l[r] = -a;
            }
            if (k < N) {
                const double *ln = w->lamn + (size_t)(k + 1) * nx, *u = w->U + (size_t)k * nu, *du = w->dU + (size_t)k * nu;
                for (int i = 0; i < m; i++) {
                    double a = -T * u[2 * i] * w->sn[k * m + i], b = T * u[2 * i] * w->cs[k * m + i];
                    l[3 * i] += ln[3 * i]; l[3 * i + 1] += ln[3 * i + 1];
                    l[3 * i + 2] += ln[3 * i + 2] + a * ln[3 * i] + b * ln[3 * i + 1] - w->hvt[k * m + i] * du[2 * i];
                }
            }
        }
        /* ---- slack / dual steps and fraction to the boundary (IPOPT eq. 15) */
        double a_p = 1.0, a_d = 1.0, mult_max = 0.0;   /* mult_max: inf-norm of the QP multipliers of all rows that enter theta */
        for (size_t i = nx; i < (size_t)(N + 1) * nx; i++) mult_max = fmax(mult_max, fabs(w->lamn[i]));
        for (int k = 0; k <= N; k++) {
            j_apply(w, k, w->X + (size_t)k * nx, w->dX + (size_t)k * nx, k < N ? w->dU + (size_t)k * nu : NULL, Jd);
            for (int s = 0; s < nh; s++) {
                size_t o = (size_t)k * nh + s;
                if (!slot_active(w, k, s)) { w->dS[o] = 0.0; w->dZ[o] = 0.0; continue; }
                double ds, dz;
                w->dEl[o] = 0.0;
                if (ELS(w, s)) {
                    const double zz = w->Z[o], rz = w->rho - zz, D_ = w->S[o] / zz + w->El[o] / rz, q_ = w->H[o] - mu / zz + mu / rz;
                    dz = -(Jd[s] + q_) / D_;
                    ds = (mu - w->S[o] * zz - w->S[o] * dz) / zz;
                    const double dt = (mu - w->El[o] * rz + w->El[o] * dz) / rz;
                    w->dEl[o] = dt;
                    if (dt < 0.0) a_p = fmin(a_p, -tau * w->El[o] / dt);
                    if (dz > 0.0) a_d = fmin(a_d, tau * rz / dz);      /* rho - z stays positive */
                } else {
                    ds = Jd[s] + (w->H[o] - w->S[o]);
                    dz = (mu - w->S[o] * w->Z[o] - w->Z[o] * ds) / w->S[o] - w->corr[o];
                }
                w->dS[o] = ds; w->dZ[o] = dz;
                if (s >= w->o_pr) mult_max = fmax(mult_max, fabs(w->Z[o] + dz));
                if (ds < 0.0) a_p = fmin(a_p, -tau * w->S[o] / ds);
                if (dz < 0.0) a_d = fmin(a_d, -tau * w->Z[o] / dz);
            }
        }
        if (w->pc && pc_pass == 0) {       /* EXPERIMENT: corrector pass with the complementarity product of the predictor */
            double a_aff = fmin(a_p, a_d), num = 0.0, den = 0.0; int cnt = 0;
            for (int k = 0; k <= N; k++)
                for (int s = 0; s < nh; s++) {
                    size_t o = (size_t)k * nh + s;
                    if (!slot_active(w, k, s)) { w->corr[o] = 0.0; continue; }
                    w->corr[o] = w->dS[o] * w->dZ[o] / w->S[o];
                    num += (w->S[o] + a_aff * w->dS[o]) * (w->Z[o] + a_aff * w->dZ[o]); den += w->S[o] * w->Z[o]; cnt++;
                }
            if (w->pc == 2) { double r = (cnt && den > 0.0) ? num / den : 1.0; mu = fmax(w->tol / 10.0, fmin(mu_keep, r * r * r * den / cnt)); }
            if (w->pc == 3) {   /* keep the plain step */
                memcpy(w->dXc, w->dX, sizeof(double) * (size_t)(N + 1) * nx); memcpy(w->dUc, w->dU, sizeof(double) * (size_t)N * nu);
                memcpy(w->dSc, w->dS, sizeof(double) * (size_t)(N + 1) * nh); memcpy(w->dZc, w->dZ, sizeof(double) * (size_t)(N + 1) * nh);
                memcpy(w->lamnc, w->lamn, sizeof(double) * (size_t)(N + 1) * nx);
                pc_ap = a_p; pc_ad = a_d; pc_mm = mult_max;
            }
            pc_pass = 1;
            goto pc_again;
        }
        if (w->pc) memset(w->corr, 0, sizeof(double) * (size_t)(N + 1) * nh);
        if (w->pc == 3) {       /* plain step back into place (the merit model below is the plain step's), corrected step aside */
            double *t, ts;
            t = w->dX; w->dX = w->dXc; w->dXc = t; t = w->dU; w->dU = w->dUc; w->dUc = t; t = w->dS; w->dS = w->dSc; w->dSc = t;
            t = w->dZ; w->dZ = w->dZc; w->dZc = t; t = w->lamn; w->lamn = w->lamnc; w->lamnc = t;
            ts = a_p; a_p = pc_ap; pc_ap = ts; ts = a_d; a_d = pc_ad; pc_ad = ts; ts = mult_max; mult_max = pc_mm; pc_mm = ts;
        }
        /* ---- l1 merit backtracking */
        double th0, phi0 = barrier_and_infeas(w, f, w->C, w->H, w->S, mu, &th0);
        double dphi = 0.0;
        for (int k = 0; k <= N; k++) {
            const double *x = w->X + (size_t)k * nx;
            if (k >= 1 && k < N) for (int c = 0; c < nx; c++) dphi += 2 * w->qd[c] * (x[c] - xs[c]) * w->dX[(size_t)k * nx + c];
            if (k < N) for (int c = 0; c < nu; c++) dphi += 2 * w->rd[c] * w->U[(size_t)k * nu + c] * w->dU[(size_t)k * nu + c];
            for (int s = 0; s < nh; s++) if (slot_active(w, k, s)) { size_t o = (size_t)k * nh + s; dphi -= mu * w->dS[o] / w->S[o]; if (ELS(w, s)) dphi += (w->rho - mu / w->El[o]) * w->dEl[o]; }
        }
        if (th0 > 0.0) {
            /* Nocedal-Wright (18.36) with rho = 0.1.  In exact arithmetic dphi <= ||multipliers+||_inf * theta, so the
               quotient never exceeds mult_max / 0.9; the cap only filters the case theta ~ rounding noise. */
            double nut = fmin(dphi / ((1.0 - 0.1) * th0), mult_max / (1.0 - 0.1));
            nu_pen = fmax(1.0, 0.5 * nu_pen);      /* the penalty may relax again: one bad step must not cripple the rest of the solve */
            if (nu_pen < nut) nu_pen = nut + 1.0;
        }
        double D = dphi - nu_pen * th0, alpha = a_p, ft = f;
        /* non-monotone (Grippo-type) reference: the trial merit is compared with the largest of the current and the last
           three merit values of this barrier problem; cures the Maratos-type rejections of full steps near the solution */
        if (mh_mu != mu || mh_nu != nu_pen) { mcount = 0; mh_mu = mu; mh_nu = nu_pen; }
        const double m0 = phi0 + nu_pen * th0;
        double mref = m0;
        if (mcount > 0) mref = fmax(mref, mh0);
        if (mcount > 1) mref = fmax(mref, mh1);
        if (mcount > 2) mref = fmax(mref, mh2);
        mh2 = mh1; mh1 = mh0; mh0 = m0; if (mcount < 3) mcount++;
        int pc_taken = 0;
        if (w->pc == 3) {       /* the corrected step is taken only at its full fraction-to-boundary length and only if it meets the plain step's Armijo bound */
            const double ac = pc_ap;
            for (size_t i = 0; i < (size_t)(N + 1) * nx; i++) w->Xt[i] = w->X[i] + ac * w->dXc[i];
            for (size_t i = 0; i < (size_t)N * nu; i++) w->Ut[i] = w->U[i] + ac * w->dUc[i];
            for (size_t i = 0; i < (size_t)(N + 1) * nh; i++) w->St[i] = w->S[i] + ac * w->dSc[i];
            ft = eval_point(w, xs, w->Xt, w->Ut, w->snt, w->cst, w->Ct, w->Ht);
            double tht, phit = barrier_and_infeas(w, ft, w->Ct, w->Ht, w->St, mu, &tht);
            if (phit + nu_pen * tht <= mref + 1e-4 * ac * D + 1e-13 * fabs(phi0)) {
                double *t;
                pc_taken = 1; alpha = ac; a_d = pc_ad;
                t = w->dZ; w->dZ = w->dZc; w->dZc = t; t = w->lamn; w->lamn = w->lamnc; w->lamnc = t;
            }
        }
        if (w->ipopt) {
            /* Waechter & Biegler (2006) algorithm A, steps A-5.*: filter line search on (theta, phi) = (l1 infeasibility, barrier function).
               gamma_theta = 1e-5, gamma_phi = 1e-8, delta = 1, s_theta = 1.1, s_phi = 2.3, eta_phi = 1e-4, theta_min = 1e-4 max(1, theta_0),
               theta_max = 1e4 max(1, theta_0) (IPOPT's defaults) */
            const double g_th = 1e-5, g_ph = 1e-8, s_th = 1.1, s_ph = 2.3, eta = 1e-4;
            if (th_init < 0.0) th_init = th0;
            const double th_min = 1e-4 * fmax(1.0, th_init), th_max = 1e4 * fmax(1.0, th_init);
            if (flt_mu != mu) { nflt = 0; flt_mu = mu; }
            int accepted = 0, armijo_case = 0;
            for (int ls = 0; ls < 30; ls++) {
                for (size_t i = 0; i < (size_t)(N + 1) * nx; i++) w->Xt[i] = w->X[i] + alpha * w->dX[i];
                for (size_t i = 0; i < (size_t)N * nu; i++) w->Ut[i] = w->U[i] + alpha * w->dU[i];
                for (size_t i = 0; i < (size_t)(N + 1) * nh; i++) w->St[i] = w->S[i] + alpha * w->dS[i];
                ft = eval_point(w, xs, w->Xt, w->Ut, w->snt, w->cst, w->Ct, w->Ht);
                double tht, phit = barrier_and_infeas(w, ft, w->Ct, w->Ht, w->St, mu, &tht);
                int ok_f = (tht <= th_max) && (phit == phit);
                for (int j = 0; j < nflt && ok_f; j++) if (tht >= flt_th[j] && phit >= flt_ph[j]) ok_f = 0;
                if (ok_f) {
                    const int sw = dphi < 0.0 && alpha * pow(-dphi, s_ph) > pow(th0, s_th);       /* switching condition (19) */
                    if (th0 <= th_min && sw) { if (phit <= phi0 + eta * alpha * dphi + 1e-13 * fabs(phi0)) { accepted = 1; armijo_case = 1; } }
                    else if (tht <= (1.0 - g_th) * th0 || phit <= phi0 - g_ph * th0) accepted = 1;
                }
                if (accepted) break;
                if (ls < 29) alpha *= 0.5;
            }
            /* A-7: augment the filter unless the step was an Armijo step on the barrier function */
            if (!armijo_case && nflt < FMAX) { flt_th[nflt] = (1.0 - g_th) * th0; flt_ph[nflt] = phi0 - g_ph * th0; nflt++; }
        } else
        for (int ls = 0; ls < 30 && !pc_taken; ls++) {
            for (size_t i = 0; i < (size_t)(N + 1) * nx; i++) w->Xt[i] = w->X[i] + alpha * w->dX[i];
            for (size_t i = 0; i < (size_t)N * nu; i++) w->Ut[i] = w->U[i] + alpha * w->dU[i];
            for (size_t i = 0; i < (size_t)(N + 1) * nh; i++) { w->St[i] = w->S[i] + alpha * w->dS[i]; w->Elt[i] = w->El[i] + alpha * w->dEl[i]; }
            ft = eval_point(w, xs, w->Xt, w->Ut, w->snt, w->cst, w->Ct, w->Ht);
            double tht, phit = barrier_and_infeas_t(w, ft, w->Ct, w->Ht, w->St, w->Elt, mu, &tht);
            if (phit + nu_pen * tht <= mref + 1e-4 * alpha * D + 1e-13 * fabs(phi0)) break;
            if (ls < 29) alpha *= 0.5;
        }
        if (w->trace) fprintf(stderr, "it %3d E0 %.2e mu %.1e f %.6f th0 %.2e alpha %.3g a_p %.3g a_d %.3g delta %.2e nu %.3g dphi %.3g ntry %d\n",
                              it, kkt, mu, f, th0, alpha, a_p, a_d, delta_last, nu_pen, dphi, ntry);
        /* the duals never step further than the primal variables actually moved: a dual step taken
           without its primal counterpart (line search cut alpha) blows up the dual infeasibility of rows with tiny slacks */
        if (!w->ipopt) a_d = fmin(a_d, alpha);
        /* stall: the search direction is blocked (slacks pinned at zero with the infeasibility not decreasing) — the iterate
           is converging to an infeasible stationary point; IPOPT would enter restoration here, this solver reports it */
        n_tiny = (alpha < 1e-10) ? n_tiny + 1 : 0;
        /* accept (also when the search ran out: tiny step, as IPOPT's "tiny step" rule) */
        { double *t;
          t = w->X; w->X = w->Xt; w->Xt = t; t = w->U; w->U = w->Ut; w->Ut = t; t = w->S; w->S = w->St; w->St = t;
          if (w->el) { t = w->El; w->El = w->Elt; w->Elt = t; }
          t = w->C; w->C = w->Ct; w->Ct = t; t = w->H; w->H = w->Ht; w->Ht = t;
          t = w->sn; w->sn = w->snt; w->snt = t; t = w->cs; w->cs = w->cst; w->cst = t; }
        f = ft;
        for (int k = 0; k <= N; k++)
            for (int s = 0; s < nh; s++)
                if (slot_active(w, k, s)) {
                    size_t o = (size_t)k * nh + s;
                    double z = w->Z[o] + a_d * w->dZ[o];
                    double lo = mu / (1e10 * w->S[o]), hi = 1e10 * mu / w->S[o];
                    if (ELS(w, s)) hi = fmin(hi, w->rho - mu / (1e10 * w->El[o]));
                    w->Z[o] = fmin(fmax(z, lo), hi);
                }
        for (size_t i = nx; i < (size_t)(N + 1) * nx; i++) w->lam[i] += alpha * (w->lamn[i] - w->lam[i]);
        it++;
        if (n_tiny >= 5) {
            if (n_restart >= w->max_restarts) {
                if (n_cold < max_cold && w->max_restarts > 0) { COLD_RETRY(); continue; }
                status = NMPC_STATUS_STALLED; break;
            }
            /* barrier restart from the current primal point (a restoration phase in miniature): slacks back onto the
               constraint values, duals mu/s, multipliers 0, barrier parameter back to at least mu_init.  Rescues 15 of 16
               captured stalls of the six-robot + eight-obstacle composite. */
            n_restart++; n_tiny = 0;
            mu = fmax(mu, w->mu_init);
            f = barrier_restart(w, xs, mu, 0);
            delta_last = 0.0; nu_pen = 1.0; need_shift = 0; mcount = 0;
            memset(w->dl_k, 0, sizeof(double) * (size_t)N * nu); memset(w->ns_k, 0, sizeof(int) * (size_t)N * nu);
        }
    }
    memcpy(wout, w->X, sizeof(double) * (size_t)(N + 1) * nx);
    memcpy(wout + (size_t)(N + 1) * nx, w->U, sizeof(double) * (size_t)N * nu);
    *obj_out = f; *iters_out = it; *kkt_out = kkt;
    return status;
}

/* ---- exported test/baseline entry points ------------------------------------------------- */

int32_t nmpc_oracle_solve_batch(const nmpc_config_t *cfg, int32_t B, const double *p, const double *w0, double *w_out, double *obj,
                                int32_t *status, int32_t *iters, double *kkt, int32_t nthreads)
{
    if (!cfg || cfg->m < 1 || cfg->m > NMPC_MAX_ROBOTS || cfg->N < 1 || cfg->n_obs < 0 || cfg->n_obs > NMPC_MAX_OBSTACLES) return NMPC_E_ARG;
    const int nv = nmpc_oracle_n_var(cfg), np_ = nmpc_oracle_n_p(cfg);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel
    {
        ws_t *w = ws_new(cfg);
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; b++) {
            double o = 0, k = 0; int it = 0;
            int st = solve_one(w, p + (size_t)b * np_, w0 + (size_t)b * nv, w_out + (size_t)b * nv, &o, &it, &k);
            if (obj) obj[b] = o;
            if (status) status[b] = st;
            if (iters) iters[b] = it;
            if (kkt) kkt[b] = k;
        }
        ws_free(w);
    }
    return NMPC_OK;
}

/* f and g in the reference's row order (C6:278,314,318-331) */
int32_t nmpc_oracle_eval_batch(const nmpc_config_t *cfg, int32_t B, const double *p, const double *wv, double *f, double *g)
{
    const int m = cfg->m, N = cfg->N, nx = 3 * m, nu = 2 * m, M = cfg->pair_rows ? m * (m - 1) / 2 : 0, K = cfg->n_obs;
    const int nv = nmpc_oracle_n_var(cfg), ng = nmpc_oracle_n_g(cfg);
    for (int b = 0; b < B; b++) {
        const double *X = wv + (size_t)b * nv, *U = X + (size_t)(N + 1) * nx, *pp = p + (size_t)b * 2 * nx;
        double fv = 0.0;
        double *gg = g ? g + (size_t)b * ng : NULL;
        int o = 0;
        if (gg) {
            for (int c = 0; c < nx; c++) gg[o++] = X[c] - pp[c];
            if (cfg->pad_rows) for (int c = 0; c < M; c++) gg[o++] = cfg->pad_value;
        }
        for (int k = 0; k < N; k++) {
            const double *x = X + (size_t)k * nx, *xn = x + nx, *u = U + (size_t)k * nu;
            for (int i = 0; i < m; i++) {
                for (int d = 0; d < 3; d++) { double e = x[3 * i + d] - pp[nx + 3 * i + d]; fv += cfg->q[d] * e * e; }
                fv += cfg->r[0] * u[2 * i] * u[2 * i] + cfg->r[1] * u[2 * i + 1] * u[2 * i + 1];
                if (gg) {
                    gg[o++] = xn[3 * i] - (x[3 * i] + cfg->T * u[2 * i] * cos(x[3 * i + 2]));
                    gg[o++] = xn[3 * i + 1] - (x[3 * i + 1] + cfg->T * u[2 * i] * sin(x[3 * i + 2]));
                    gg[o++] = xn[3 * i + 2] - (x[3 * i + 2] + cfg->T * u[2 * i + 1]);
                }
            }
            if (gg) {
                for (int i = 0; i < (M ? m : 0); i++)
                    for (int j = i + 1; j < m; j++) {
                        double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1];
                        gg[o++] = dx * dx + dy * dy;
                    }
                for (int i = 0; i < m; i++)
                    for (int q = 0; q < K; q++) {
                        double dx = x[3 * i] - cfg->obs[3 * q], dy = x[3 * i + 1] - cfg->obs[3 * q + 1];
                        gg[o++] = sqrt(dx * dx + dy * dy) - cfg->rob_dim - cfg->obs[3 * q + 2];
                    }
            }
        }
        if (f) f[b] = fv;
    }
    return NMPC_OK;
}

/* warm-start shift (C6:160-169,460-465) and optional plant step (casadi_test.py:17-26) */
int32_t nmpc_oracle_shift_batch(const nmpc_config_t *cfg, int32_t B, const double *p_in, const double *w_in, double *w_next,
                                double *x0_next)
{
    const int m = cfg->m, N = cfg->N, nx = 3 * m, nu = 2 * m, nv = nmpc_oracle_n_var(cfg);
    for (int b = 0; b < B; b++) {
        const double *X = w_in + (size_t)b * nv, *U = X + (size_t)(N + 1) * nx;
        double *Xn = w_next + (size_t)b * nv, *Un = Xn + (size_t)(N + 1) * nx;
        for (int k = 0; k < N; k++) memcpy(Xn + (size_t)k * nx, X + (size_t)(k + 1) * nx, sizeof(double) * nx);
        memcpy(Xn + (size_t)N * nx, X + (size_t)(N - 1) * nx, sizeof(double) * nx);
        for (int k = 0; k < N - 1; k++) memcpy(Un + (size_t)k * nu, U + (size_t)(k + 1) * nu, sizeof(double) * nu);
        memcpy(Un + (size_t)(N - 1) * nu, U + (size_t)(N - 1) * nu, sizeof(double) * nu);
        if (x0_next) {
            const double *x0 = p_in + (size_t)b * 2 * nx;
            for (int i = 0; i < m; i++) {
                x0_next[(size_t)b * nx + 3 * i] = x0[3 * i] + cfg->T * U[2 * i] * cos(x0[3 * i + 2]);
                x0_next[(size_t)b * nx + 3 * i + 1] = x0[3 * i + 1] + cfg->T * U[2 * i] * sin(x0[3 * i + 2]);
                x0_next[(size_t)b * nx + 3 * i + 2] = x0[3 * i + 2] + cfg->T * U[2 * i + 1];
            }
        }
    }
    return NMPC_OK;
}

int32_t nmpc_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
