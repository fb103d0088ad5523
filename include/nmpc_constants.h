/*
 * nmpc_constants.h — constants of the interior-point algorithm that the HIP kernels (csrc/) and the CPU oracles (oracle/) must
 * agree on: the GPU parity tests assert equal statuses and (almost everywhere) equal iteration counts, so a constant that differed
 * between the two sides would make them diverge silently.  One definition, included by csrc/nmpc_device.h, csrc/nmpc_lidar.hip,
 * oracle/nmpc_oracle.c and oracle/lidar_oracle.c.  Plain C preprocessor constants, no dependencies.
 */
#ifndef NMPC_CONSTANTS_H_
#define NMPC_CONSTANTS_H_

/* Inertia correction: factor between consecutive trial shifts of one iteration once a previous iteration needed a shift (IPOPT's
   kappa_w^+ = 8).  Measured with 4: the literal antipodal swap drops from 113 to 67 iterations, but the warm closed loop
   falls from 148.8 k to 110.8 k solves/s and the composite from 17.1 k to 12.0 k: kept at 8. */
#ifndef NMPC_SHIFT_ESCALATION
#define NMPC_SHIFT_ESCALATION 8.0
#endif

/* Inertia correction, partial re-factorisation (round 4).  The backward sweep saves the cost-to-go entering every NMPC_CKPT_EVERY-th
   stage (the stages k with (N-1-k) % NMPC_CKPT_EVERY == 0).  A rejected pivot at stage k escalates the shift and resumes the sweep at
   NMPC_RESUME_STAGE(k, N): the nearest saved stage at or above min(k + NMPC_REFACTOR_BACK, N-1) — the stages above keep their
   factorisation (and the smaller shift it was made with).  Measured on the oracle (tools/sreg_experiment.py): resuming AT
   the failing stage (BACK = 0) costs 40 % more iterations, BACK >= 1 with EVERY >= 3 or BACK >= 2 none (4 % fewer than the whole-sweep retry). */
#define NMPC_CKPT_EVERY 5
#define NMPC_REFACTOR_BACK 2
#define NMPC_RESUME_STAGE(k, N) ((((k) + NMPC_REFACTOR_BACK < (N) - 1) ? (k) + NMPC_REFACTOR_BACK : (N) - 1) + \
                                 ((N) - 1 - (((k) + NMPC_REFACTOR_BACK < (N) - 1) ? (k) + NMPC_REFACTOR_BACK : (N) - 1)) % NMPC_CKPT_EVERY)

/* Cold-start retry (restoration of last resort): a solve that stalls after its barrier restarts, fails numerically or is
   still iterating NMPC_COLD_RETRY_ITERS iterations into an attempt is restarted from the reference's cold start X_k = x0, U = 0
   (C6:398-400; LIDAR: V4:184-196), at most NMPC_COLD_RETRIES times, the second time with a ten times larger initial barrier parameter. */
#ifndef NMPC_COLD_RETRY_ITERS
#define NMPC_COLD_RETRY_ITERS 500
#endif
#define NMPC_COLD_RETRIES 2

/* Round 4: the SECOND restart of last resort is the elastic phase (oracle/nmpc_oracle.c has the derivation): from the cold start, with the pair and
   obstacle rows relaxed to h + t - s = 0, t >= 0 under the penalty NMPC_ELASTIC_RHO * sum t — in place of round 2's second cold retry with a
   ten times larger barrier parameter.  Measured on the captured failures of the composite: rho = 1e2 rescues 42 of 42, 1e3 41, 1e4 40. */
#define NMPC_ELASTIC_RHO 100.0

/* LIDAR solve only (round 4): watchdog of the line search, after IPOPT's (watchdog_shortened_iter_trigger = 10): once the l1-merit backtracking
   has shortened NMPC_WATCHDOG_TRIGGER successive steps, the next step whose fraction-to-the-boundary length is at least NMPC_WATCHDOG_MIN_AP
   is taken at that length without the merit test and the non-monotone merit history starts afresh.  The distance rows of that NLP are 1-norms
   (V4:146-149): across their kinks the merit function rejects good steps for hundreds of iterations (the three longest solves of the bench
   batch: 175 / 154 / 145 iterations -> 61 at most; 16,384 other instances: 374 / 319 / 271 / 231 / 155 -> 374 / 126 / 85).  IPOPT's return to
   the stored iterate when the watchdog fails is not reproduced.  The swarm solves never meet the trigger (measured: 0 of 2048 six-robot solves). */
#define NMPC_WATCHDOG_TRIGGER 10
#define NMPC_WATCHDOG_MIN_AP 0.5

/* Slack of the stage-0 feasibility pre-check (status 3): a measured x0 that violates a pair / obstacle row by less than this — the
   previous period's plan holds its rows to the solve tolerance only — is not reported as infeasible. */
#define NMPC_X0_TOL 1e-6

#endif /* NMPC_CONSTANTS_H_ */
