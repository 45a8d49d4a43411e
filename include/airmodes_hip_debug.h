/* airmodes_hip_debug.h -- diagnostics and test hooks of libairmodes_hip.so.
 *
 * NOT part of the drop-in surface (include/airmodes_hip.h: the reference's preamble::make / slicer::make and their four
 * accessors -- include/gr_air_modes/preamble.h:36-46, slicer.h:37-42 -- plus the streaming, time-shard, uploader and resampler
 * calls around them).  What is declared here is exported by the same shared object and may change with the kernels: timing of
 * the last call (bench.py's roofline line), which front end ran, the candidate records of the resident scan (stage-level parity
 * tests), whether the library is the CPU emulation (tests/emu), and the exit word of the time-shard path's synchronous
 * fallback.  The export test (tests/test_capi_host.py) asserts the union of both headers against the library's dynamic symbols.
 */
#ifndef AIRMODES_HIP_DEBUG_H
#define AIRMODES_HIP_DEBUG_H

#include "airmodes_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 0 for the product library.  1 only in the test-only CPU build of the same sources (tests/emu), where device pointers are
 * host pointers: callers that must choose between device-side and host-side message buffers ask this instead of guessing from
 * the library's file name. */
AM_API int am_is_emulated(void);

/* timing of the last am_process_iq / am_shard_scan call, measured with HIP events on the
 * context's own stream: device milliseconds for the whole call and for the dominant
 * (front-end + detection) kernel.  Used by bench.py for the roofline line.  Either pointer may be
 * NULL; asking for total_ms may wait a few microseconds for the call's last event (it is queued
 * behind the completion signal the call itself waits for), dominant_kernel_ms never waits. */
AM_API int am_last_timing(am_ctx *ctx, float *total_ms, float *dominant_kernel_ms);

/* Diagnostic: number of first-stage preamble candidates (positions passing preamble_impl.cc:172-179)
 * the last scan refined and chained.  Negative error code on a null context. */
AM_API long long am_last_num_candidates(const am_ctx *ctx);

/* Diagnostic: which front-end kernel the last scan ran -- 3 = a streaming kernel (am_k_fe3 at 64 Msps, am_k_fe4 at 2, 4, 8,
 * 10, 16, 20, 32 and 40 Msps: persistent workgroups, LDS rings, sparse bb around candidates), 2 = tile kernel (am_k_fe2, dense bb), 1 = rate-generic kernels, 0 = no scan
 * yet.  Results do not depend on it (test builds can keep the tile kernel; tests compare both). */
AM_API int am_last_frontend(const am_ctx *ctx);

/* Diagnostic (stage-level parity tests): the refined record of EVERY first-stage candidate of the last scan, in
 * position order -- absolute stream index of the position the first-stage test fired at (preamble_impl.cc:172-179), of
 * the position after the late-peak search (:182-192), the outcome of the quiet-zone test there (:198-209) and, for a
 * candidate, the reference level at that position (what :220 subtracts).  Any pointer may be NULL.  AM_ECAPACITY
 * (*n_out = needed) if cap is too small. */
AM_API int am_fetch_candidates(am_ctx *ctx, uint64_t *pos, uint64_t *refined, uint8_t *valid, float *inavg, uint64_t cap,
                        uint64_t *n_out);

/* Time shards, synchronous fallback only (air_modes/sharded.py: a step whose tables did not fit, or CPU tensors): where the scan
 * left this context's chunk in its last resolved step -- the word the host-free path keeps on the device and hands on through the
 * message header (am_shard_scan_async).  get: read it (synchronises the context's stream); set: what an accepted synchronous
 * step leaves for the next step's header. */
AM_API int am_shard_get_exit(am_ctx *ctx, uint64_t *pos);
AM_API int am_shard_set_exit(am_ctx *ctx, uint64_t pos);

#ifdef __cplusplus
}
#endif
#endif /* AIRMODES_HIP_DEBUG_H */
