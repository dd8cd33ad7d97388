/* roxtrace_diag.h -- measurement and self-test helpers of libroxtrace.so.
 *
 * NOT part of the drop-in boundary (include/roxtrace.h): nothing in the
 * reference corresponds to these.  bench.py, tools/ and tests/ call them.
 */
#ifndef ROXTRACE_DIAG_H
#define ROXTRACE_DIAG_H

#include "roxtrace.h"

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

/* runs `launches` back-to-back launches of the pupil-grid kernel on `stream`,
 * bracketed by HIP events recorded on that same stream, and returns the mean
 * kernel duration in milliseconds (bench.py's roofline figures). */
int rox_time_pupil_grid(rox_system *sys, const rox_field *fld,
                        const rox_grid *grid, int32_t wvl_idx,
                        const rox_opts *opts, const rox_out *out,
                        void *stream, int32_t launches, double *mean_ms);

/* compares the kernels' exponent-band-guarded sqrt / division paths with the
 * plain IEEE operators on n pseudo-random operand sets (whole exponent range,
 * zeros, denormals, inf, nan, and wave-uniform band-edge classes).
 * counts[0] = sqrt mismatches, counts[1] = division mismatches (both must be
 * 0), counts[2] = operand sets that took a guarded path, counts[3] = operand
 * sets of the band-edge classes that took a guarded path. */
int rox_selftest_fp64(uint64_t n, uint64_t seed, uint64_t counts[4]);

/* how many ROX_OUT_HITS_COMPACT launches this process has issued in each form:
 * counts[0] = fused (compaction inside the trace kernel), counts[1] = two-pass (plain HITS
 * launch + pack kernel; deep tables / Newton instances writing device memory, or
 * ROX_PACK_TWO_PASS=1).  Tests use it to see which form a call took. */
int rox_diag_pack_launches(uint64_t counts[2]);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* ROXTRACE_DIAG_H */
