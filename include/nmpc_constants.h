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

/* Slack of the stage-0 feasibility pre-check (status 3): a measured x0 that violates a pair / obstacle row by less than this — the
   previous period's plan holds its rows to the solve tolerance only — is not reported as infeasible. */
#define NMPC_X0_TOL 1e-6

#endif /* NMPC_CONSTANTS_H_ */
