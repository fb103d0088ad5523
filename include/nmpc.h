/*
 * nmpc.h — C ABI of the MI355X batched NMPC solver (libnmpc_hip.so).
 *
 * Drop-in boundary for the per-timestep solve of the reference scripts
 *   AS = AllScripts/ of asalimil/Nonlinear-MPC-for-collision-free-and-deadlock-free-
 *        navigation-of-multiple-nonholonomic-mobile-robots
 *   C6 = AS/centralized_six_robots_implementation.py
 *
 * The reference has no FFI of its own: the boundary it exposes is the CasADi solver
 * object call plus shift():
 *     solver = nlpsol('solver','ipopt', nlp_prob, opts)          C6:345-346
 *     sol    = solver(x0=,p=,lbx=,ubx=,lbg=,ubg=)                C6:432
 *     t0, u0 = shift(T, t0, u)                                   C6:160-169,450
 * Each entry point below names the block it replaces.  All buffers are caller-owned
 * DEVICE pointers (fp64 / int32), row-major with the batch index leading.  No global
 * state (the library reads no environment variable; what is not a literal of a reference
 * script is an explicit nmpc_options_t); every call is ordered on the hipStream_t passed as `void *stream` (NULL = the
 * default stream).  Return value: 0 on success, negative NMPC_E_* on argument / HIP
 * errors.  Non-convergence is NOT an error (the reference never reads solver.stats());
 * it is reported per instance in `status`.
 *
 * Concurrency: a handle owns ONE device workspace that every solve on it uses, so one
 * handle must not be used from two streams or two host threads at the same time (the
 * reference's loop is single-threaded and blocking, C6:416-465); different handles are
 * independent.  The handle remembers the device it was created on; every call makes
 * that device current for its own duration, and all buffers and the stream passed to a
 * call must belong to it.
 */
#ifndef NMPC_H_
#define NMPC_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NMPC_MAX_ROBOTS 10
#define NMPC_MAX_OBSTACLES 8

/* per-instance solve status */
#define NMPC_STATUS_CONVERGED 0      /* scaled KKT error <= tol                                  */
#define NMPC_STATUS_MAX_ITER 1       /* iteration limit hit; last iterate returned               */
#define NMPC_STATUS_NUMERIC 2        /* inertia correction exhausted / non-finite step           */
#define NMPC_STATUS_INFEASIBLE_X0 3  /* a stage-0 pair/obstacle row is violated by the pinned x0 */
#define NMPC_STATUS_STALLED 4        /* the restarts of last resort are spent (barrier restarts, the cold start, the elastic phase — where
                                        IPOPT runs its restoration phase) and the iteration still stalls, or the elastic phase converged with
                                        an elastic variable open: a stationary point of the infeasibility; last iterate returned */

/* return codes */
#define NMPC_OK 0
#define NMPC_E_ARG (-1)
#define NMPC_E_UNSUPPORTED (-2)
#define NMPC_E_HIP (-3)
#define NMPC_E_NOMEM (-4)

/*
 * Every literal of one reference script (C6:197-205 T,N,m,dmin,v_max,omega_max;
 * C6:252-266 Q,R; C6:349-352 bounds; obstacle literals
 * AS/third_scenario_mpc_obstacle_avoidance.py:58,97-119,175-177).
 */
typedef struct nmpc_config {
    int32_t m;              /* robots: 1..10 (the reference scripts use 1..6, 8 and 10; 7 and 9 are instantiated since round 4); beyond -> NMPC_E_UNSUPPORTED */
    int32_t N;              /* horizon, 2..4096                                                 */
    int32_t n_obs;          /* static circular obstacles, 0..NMPC_MAX_OBSTACLES                 */
    int32_t pad_rows;       /* 1: initial g block carries M constant rows (C6:278); 0: it does not */
    double T;               /* sample time                                                      */
    double dmin;            /* pair rows bounded below by dmin^2 (C6:349)                       */
    double q[3];            /* Q diagonal per robot                                             */
    double r[2];            /* R diagonal per robot                                             */
    double v_max, w_max;    /* |v| <= v_max, |omega| <= w_max                                   */
    double xy_max;          /* |x|,|y| <= xy_max (10 in every script)                           */
    double th_max;          /* |theta| <= th_max; +inf = unbounded (multi-robot scripts)        */
    double rob_dim, margin; /* obstacle rows: sqrt(.) - rob_dim - obs_r >= margin               */
    double pad_value;       /* 3.5 (C6:278)                                                     */
    double obs[3 * NMPC_MAX_OBSTACLES]; /* (ox, oy, obs_r) per obstacle                         */
    /* solver options: the 'ipopt' dict of C6:345 plus IPOPT defaults that matter */
    double tol;             /* 1e-8 (acceptable_tol of C6:345 == IPOPT tol)                     */
    double mu_init;         /* 0.5: initial barrier parameter (IPOPT ships 0.1; 0.5 saves ~13% iterations) */
    int32_t max_iter;       /* reference: 2000                                                  */
    int32_t pair_rows;      /* 1: pairwise collision rows present (C6:288-306); 0: the multi-robot NLP WITHOUT them
                               (AS/mpc_online_casadi_tb3_multi_centralized.py:115-148: g has 3m(N+1) rows, no padding rows) */
} nmpc_config_t;

typedef struct nmpc_handle nmpc_handle_t;

/* sizes implied by a config */
int32_t nmpc_n_var(const nmpc_config_t *cfg); /* n_x (N+1) + n_u N                    (C6:339) */
int32_t nmpc_n_g(const nmpc_config_t *cfg);   /* rows of g in the reference's order   (C6:278,326-331) */
int32_t nmpc_n_p(const nmpc_config_t *cfg);   /* 2 n_x                                (C6:241) */

/* fills *cfg with the defaults above and the Q/R/bounds literals shared by all scripts */
void nmpc_config_default(nmpc_config_t *cfg, int32_t m, int32_t N);

/*
 * Replaces nlpsol('solver','ipopt',nlp_prob,opts) (C6:342-346): validates the config and
 * allocates a device workspace for up to max_batch instances.
 */
int32_t nmpc_create(const nmpc_config_t *cfg, int32_t max_batch, nmpc_handle_t **out);
int32_t nmpc_destroy(nmpc_handle_t *h);

/*
 * Creation options that are not literals of a reference script.  nmpc_create(cfg, B, out) is nmpc_create_opts(cfg, B, NULL, out).
 */
typedef struct nmpc_options {
    int32_t kernel;         /* 0: the library picks the solve kernel per batch size (default).  1 HBM-resident, 2 element-per-lane,
                               3 column-per-lane in its throughput shape (one wavefront per instance), 4 / 5 column-per-lane in its latency
                               shape with two / four wavefronts per instance (four: five and six robots, else two): that kernel for every
                               batch size it can run (tests, A/B measurements)                                                         */
    int32_t trace_instance; /* -DNMPC_PROFILE builds: instance whose per-iteration trace is recorded (include/nmpc_debug.h); -1 none */
} nmpc_options_t;
int32_t nmpc_create_opts(const nmpc_config_t *cfg, int32_t max_batch, const nmpc_options_t *opts, nmpc_handle_t **out);

/*
 * Facts about a handle.  Returns the value, or a negative NMPC_E_* code.
 *   NMPC_QUERY_KERNEL_FOR_BATCH  arg = B: the solve kernel nmpc_solve_batch launches for a batch of B: 1 / 2 / 3 as in nmpc_options_t, 4 = the
 *                                column kernel's latency shape (two OR four wavefronts per instance: 5 is a pin, never an answer)
 *   NMPC_QUERY_WORKSPACE_BYTES   device workspace held by the handle (same as nmpc_workspace_bytes)
 *   NMPC_QUERY_LDS_BYTES         arg = B: dynamic LDS per swarm instance of the kernel an UNORDERED call of B gets (0 for the HBM-resident kernel's fixed carve-up)
 *   NMPC_QUERY_MAX_BATCH         the max_batch the handle was created for
 *   NMPC_QUERY_KERNEL_FOR_ORDERED_BATCH  arg = B: as NMPC_QUERY_KERNEL_FOR_BATCH for a call that carries a dispatch-order hint
 *                                (nmpc_solve_batch_ordered, nmpc_step_batch): the latency shape is kept for larger batches then
 */
#define NMPC_QUERY_KERNEL_FOR_BATCH 1
#define NMPC_QUERY_WORKSPACE_BYTES 2
#define NMPC_QUERY_LDS_BYTES 3
#define NMPC_QUERY_MAX_BATCH 4
#define NMPC_QUERY_KERNEL_FOR_ORDERED_BATCH 5
int64_t nmpc_query(const nmpc_handle_t *h, int32_t what, int64_t arg);

/* bytes of device workspace held by the handle */
int64_t nmpc_workspace_bytes(const nmpc_handle_t *h);

/*
 * Replaces sol = solver(x0=,p=,lbx=,ubx=,lbg=,ubg=) (C6:432) for B independent swarms.
 *   p      [B][2 n_x]   parameters [x0; xs]                       (C6:419)
 *   w0     [B][n_var]   initial guess [X_0..X_N; U_0..U_{N-1}]     (C6:423)
 *   w_out  [B][n_var]   sol['x']                                   (C6:436,440)
 *   obj    [B]          sol['f']           (may be NULL)
 *   status [B] iters[B] kkt[B]             (may be NULL)
 * Bounds are those of the config (the scripts never change them between calls).
 * Stream-ordered; a handle owns ONE workspace.  A launch lasts as long as its longest solve: to keep the device busy over a stream of batches use two
 * handles on two streams and alternate the launches (bench.py `two_streams`: six robots, B = 4096, 277 k -> 445 k solves/s).
 */
int32_t nmpc_solve_batch(nmpc_handle_t *h, int32_t B, const double *p, const double *w0, double *w_out,
                         double *obj, int32_t *status, int32_t *iters, double *kkt, void *stream);

/*
 * nmpc_solve_batch with a dispatch-order hint: order [B] (device, int32) is a permutation of 0..B-1 and workgroup g solves
 * instance order[g]; results land at the instance's own index, exactly as without the hint.  The launch takes as long as
 * "start of the longest solve + its length", so a caller that can rank the instances by expected effort (in a receding-horizon
 * loop: by the iteration counts of the previous control period, rank correlation 0.6-0.7) puts the long ones first.
 * order == NULL is nmpc_solve_batch.  No reference counterpart (the reference solves one instance per call, C6:432).
 * A hint that is not a permutation (an entry out of range or repeated) is detected on the device before the solve and
 * ignored for that call (the batch is then solved in index order): a bad hint costs time, never results or memory safety.
 */
int32_t nmpc_solve_batch_ordered(nmpc_handle_t *h, int32_t B, const double *p, const double *w0, double *w_out, double *obj,
                                 int32_t *status, int32_t *iters, double *kkt, const int32_t *order, void *stream);

/*
 * Evaluates the NLP functions in the reference's layout at w: f (C6:314) and
 * g [B][n_g] (C6:278,318-331).  Backs sol['f'] / sol['g'] of the host wrapper.
 */
int32_t nmpc_eval_batch(nmpc_handle_t *h, int32_t B, const double *p, const double *w, double *f, double *g,
                        void *stream);

/*
 * Replaces shift() and the warm-start row shuffle (C6:160-169,450,460-465):
 *   U rows: drop first, duplicate last;  X rows: [X_1..X_N; X_{N-1}].
 * x0_next [B][n_x] may be NULL; when given it is written as the plant step
 *   x0 + T f(x0, u_0) of AS/casadi_test.py:17-26 (x0 read from p_in[:, :n_x]).
 * w_in and w_next must not alias.
 */
int32_t nmpc_shift_batch(nmpc_handle_t *h, int32_t B, const double *p_in, const double *w_in, double *w_next,
                         double *x0_next, void *stream);

/*
 * One control period of the receding-horizon loop for B swarms, entirely on the device and on one stream (C6:416-465 with the plant of
 * AS/casadi_test.py:17-26, the stack SURVEY.md 3(C) describes): nmpc_solve_batch_ordered, then nmpc_shift_batch, then the dispatch order
 * of the next period — no host round trip in between.
 *   p      [B][2 n_x]  in: [x0; xs]; out: x0 replaced by the plant step x0 + T f(x0, u_0) of the solution (xs untouched: the caller
 *                      changes goals between calls)                                                       (casadi_test.py:17-26,170)
 *   w      [B][n_var]  in: the guess; out: the next guess, the shifted solution [X_1..X_N; X_{N-1}], [U_1..U_{N-1}; U_{N-1}]   (C6:450,465)
 *   w_sol  [B][n_var]  out: sol['x'] of this period (its row U_0 is the control to apply, C6:444); must not alias w
 *   obj, status, iters, kkt: as nmpc_solve_batch (may be NULL)
 *   order  [B] int32   in: dispatch order of this period (a permutation of 0..B-1; anything else is detected and ignored);
 *                      out: the instances sorted by this period's iteration counts, longest first — the hint for the next call.
 *                      May be NULL (index order, no hint produced).  A caller's first call passes the identity.
 * Failed solves: an instance that ends with NMPC_STATUS_NUMERIC (2: the iterate may be non-finite) or NMPC_STATUS_INFEASIBLE_X0 (3) keeps its
 * x0 and its guess — its rows of p and w are NOT overwritten (w_sol holds what the solve returned); statuses 1 and 4 return a finite last
 * iterate and are shifted like a converged one, as the scripts do with any IPOPT return.
 */
int32_t nmpc_step_batch(nmpc_handle_t *h, int32_t B, double *p, double *w, double *w_sol, double *obj, int32_t *status, int32_t *iters,
                        double *kkt, int32_t *order, void *stream);

/*
 * Odometry front-end of the scripts' callbacks (AS/centralized_two_robots_implementation.py:18-37): for n robots,
 *   odom [n][4] = (x_r, y_r, q_z, q_w) wheel-odometry pose in the robot's own start frame (q_w is carried but, as in the
 *                  reference, not used: yaw = 2 asin(q_z)),
 *   init [n][3] = (x_init, y_init, th_init) pose of that start frame in the global frame,
 *   pose [n][3] = (x, y, phi) in the global frame: [x y] = R(th_init) [x_r y_r] + [x_init y_init], phi = th + th_init with
 *                  th = 2 asin(q_z), or, when wrap_2pi != 0, th = modify(2 asin(q_z)): the heading wrap into [0, 2 pi) of the
 *                  scripts without collision rows (AS/mpc_online_casadi.py:24-33: th in [-pi, 0) -> th + 2 pi); the collision-free
 *                  scripts' modify() is the identity (C2:62-68), i.e. wrap_2pi = 0.
 * Device pointers; no handle (pure function of its inputs).  SURVEY.md 8(f) row 3.
 */
int32_t nmpc_odometry_batch(int64_t n, const double *odom, const double *init, double *pose, int32_t wrap_2pi, void *stream);

/* library / kernel identification string: version, arch, and the hash of the sources it was built from (build.py) */
const char *nmpc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NMPC_H_ */
