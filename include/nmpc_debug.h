/*
 * nmpc_debug.h — development aids exported by libnmpc_hip.so next to the product ABI of nmpc.h.  Not part of the drop-in boundary:
 * nothing here has a counterpart in the reference, no product path calls it, and the layouts it exposes may change with the kernels.
 * Used by tools/ (phase profiles, per-iteration traces, workspace dumps of one instance).
 */
#ifndef NMPC_DEBUG_H_
#define NMPC_DEBUG_H_

#include "nmpc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* per-phase cycle counters of a -DNMPC_PROFILE build: out12 [12] int64 (all zero in a product build); reset != 0 clears them */
int32_t nmpc_debug_profile(nmpc_handle_t *h, int64_t *out12, int32_t reset);
/* per-iteration trace of nmpc_options_t.trace_instance (NMPC_PROFILE builds): out [rows][16] doubles, rows <= 2048 */
int32_t nmpc_debug_trace(nmpc_handle_t *h, double *out, int32_t rows);
/* second trace block of the same instance: out [rows][8] doubles */
int32_t nmpc_debug_trace2(nmpc_handle_t *h, double *out, int32_t rows);
/* copies the per-instance workspace of one instance to the host; returns its length in doubles (out == NULL: length only);
   offs [5] (may be NULL) <- kernel, oKG, oKFF, oPACK, oKT */
int64_t nmpc_debug_workspace(nmpc_handle_t *h, int32_t inst, double *out, int64_t cap, int64_t *offs);

#ifdef __cplusplus
}
#endif
#endif /* NMPC_DEBUG_H_ */
