/*
 * nmpc_lidar.h — C ABI of the LIDAR-ray distance-state NMPC solve (part of libnmpc_hip.so).
 *
 * Drop-in boundary for the per-timestep solve of
 *   V4 = AllScripts/obs_avoid_static_first_scenario_v4.py   (and V3 = ..._v3.py: Nc = N, lw = 0)
 * of the reference, i.e. of
 *     solver = nlpsol('solver','ipopt', nlp_prob, opts)                     V4:156-157
 *     sol    = solver(x0=,p=,lbx=,ubx=,lbg=,ubg=)                           V4:245
 * for B independent robots at once.  Same conventions as nmpc.h: caller-owned DEVICE buffers (fp64 / int32, batch index
 * leading), stream-ordered, integer return codes, per-instance status instead of errors for non-convergence; one handle must
 * not be used from two streams or threads at once.
 *
 * The NLP (V4:78-151): 3 + R states per stage [x y theta d_1..d_R], controls U[:, min(k, Nc-1)] (move blocking), cost
 * sum_k (x_k-xs)'Q(x_k-xs) + u'Ru + lw sum_m 1/d_mk^2, equality rows g = [gx; gd] with gd: d_{m,k+1} = ||p_{k+1} - pObs_m||_1 and
 * pObs_m the lidar point seen at stage 0.  Variable bounds are an INPUT (lbx / ubx of length n_var, +-inf allowed): the script
 * builds them as "all pose bounds, then all distance bounds" (V4:161-176), which does not line up with its own stage-major
 * packing; the caller passes whatever the script passes and the solve honours it entry by entry.
 */
#ifndef NMPC_LIDAR_H_
#define NMPC_LIDAR_H_

#include "nmpc.h"

#ifdef __cplusplus
extern "C" {
#endif

#define NMPC_LIDAR_MAX_RAYS 16

typedef struct nmpc_lidar_config {
    int32_t N;        /* prediction horizon (V4:59: 100; V3: 125)                                   */
    int32_t Nc;       /* control horizon, 1..N (V4:60: 50; V3: N)                                   */
    int32_t R;        /* numRays, 0..NMPC_LIDAR_MAX_RAYS (V4:62: 10)                                */
    int32_t max_iter; /* 2000 (V4:156)                                                              */
    double T;         /* 0.075                                                                      */
    double q[3], r[2];/* Q = diag(1, 5, 0.1), R = diag(0.5, 0.05)  (V4:119-120)                     */
    double lw;        /* weight of sum 1/d^2 (V4:121: 0.1; V3: 0)                                   */
    double tol;       /* 1e-8 */
    double mu_init;   /* 0.5  */
} nmpc_lidar_config_t;

typedef struct nmpc_lidar_handle nmpc_lidar_handle_t;

int32_t nmpc_lidar_n_var(const nmpc_lidar_config_t *cfg); /* (3+R)(N+1) + 2 Nc   (V4:155) */
int32_t nmpc_lidar_n_g(const nmpc_lidar_config_t *cfg);   /* (3+R)(N+1)          (V4:151,158) */
int32_t nmpc_lidar_n_p(const nmpc_lidar_config_t *cfg);   /* 6 + 2R              (V4:101) */

/* Replaces nlpsol(...) of V4:156-157.  lbx / ubx: HOST arrays of n_var entries (args['lbx'], args['ubx'] of V4:174-176).
 * NMPC_E_ARG also when the per-stage operands of the horizon recursions exceed the LDS of a compute unit:
 * 8 (16 (N+1) + 16 Nc + 12 N + 32) bytes <= 160 KB, i.e. N up to ~560 with Nc = N/2 (the scripts use N = 100 / 125). */
int32_t nmpc_lidar_create(const nmpc_lidar_config_t *cfg, const double *lbx, const double *ubx, int32_t max_batch, nmpc_lidar_handle_t **out);
int32_t nmpc_lidar_destroy(nmpc_lidar_handle_t *h);

/*
 * Replaces sol = solver(x0=,p=,...) (V4:245) for B robots.
 *   p [B][6+2R] = [x0; xs; scan; ray angles] (V4:230-236),  w0 / w_out [B][n_var] = [vec(X); vec(U)] (V4:239-241,247-252)
 *   obj, status, iters, kkt [B]  (may be NULL)
 * One wavefront per robot.  Batches up to one robot per SIMD (4 x the device's compute units) run the one-wave-per-SIMD build of the kernel (a lone
 * wave iterates fastest: the launch is its longest solve), larger ones the two-waves-per-SIMD build (the launch is the batch's work).  Results do not
 * depend on the choice.  Stream-ordered; a handle owns one workspace (two launches in flight need two handles).
 */
int32_t nmpc_lidar_solve_batch(nmpc_lidar_handle_t *h, int32_t B, const double *p, const double *w0, double *w_out, double *obj,
                               int32_t *status, int32_t *iters, double *kkt, void *stream);

/* f (V4:135-136) and g [B][n_g] = [gx; gd] (V4:151) at w: backs sol['f'] / sol['g'] */
int32_t nmpc_lidar_eval_batch(nmpc_lidar_handle_t *h, int32_t B, const double *p, const double *w, double *f, double *g, void *stream);

/* warm-start shuffle of V4:258-270: U rows drop first / repeat last; X rows [X_1..X_N; X_{N-1}] */
int32_t nmpc_lidar_shift_batch(nmpc_lidar_handle_t *h, int32_t B, const double *w_in, double *w_next, void *stream);

/*
 * The two inputs the script's loop gets from the robot, for a SIMULATED one (SURVEY.md 8(f) row 2 applied to V4:209-300); no handle, pure
 * functions of their inputs like nmpc_odometry_batch:
 *
 * nmpc_lidar_scan_batch replaces callback_lidar (V4:29-36) by a synthetic LaserScan: scan [B][R], the range along ray m (angle
 * theta + m 2 pi / R, V4:203-205) from pose [B][3] to the nearest of K circular obstacles world [B][K][3] = (ox, oy, radius), clipped to
 * scan_max (the script turns inf into 3.5).
 *
 * nmpc_lidar_plant_batch replaces the odometry reading of the next period by the Euler model the NLP itself uses (V4:78-87; the offline
 * plant of AS/casadi_test.py:17-26): pose_next [B][3] = pose + T f(pose, u_0), pose = p[:, 0:3] (row stride n_p = 6 + 2R), u_0 = the first
 * control row of w_sol [B][n_var] (V4:247-256).  pose_next may be the pose part of p itself (pose_stride = n_p; 0 means 3).
 */
int32_t nmpc_lidar_scan_batch(int64_t B, int32_t R, int32_t K, const double *pose, const double *world, double scan_max, double *scan, void *stream);
int32_t nmpc_lidar_plant_batch(nmpc_lidar_handle_t *h, int32_t B, const double *p, const double *w_sol, double *pose_next, int32_t pose_stride,
                               void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NMPC_LIDAR_H_ */
