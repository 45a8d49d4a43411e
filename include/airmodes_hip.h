/*
 * airmodes_hip.h -- C ABI of libairmodes_hip.so, the MI355X (gfx950) implementation of the
 * gr-air-modes receive hot path:  IQ -> |.|^2 -> moving-average threshold -> Mode-S preamble
 * detection -> PPM bit slicing with confidence -> CRC-24 syndrome -> packet list.
 *
 * This header is the drop-in boundary.  Each entry point names the reference interface it
 * replaces (paths relative to the gr-air-modes tree).  Plain pointers and sizes only; no
 * C++ or torch types; no exceptions cross the boundary.  All functions returning int return
 * AM_OK (0) or a negative AM_E* code; am_last_error() gives the text.
 *
 * Threading: a context is NOT thread-safe (one context per stream per GPU, as a GNU Radio
 * block instance is single-threaded inside work()); setters take effect at the next call.
 * Ownership: the caller owns every buffer it passes; the context owns its device buffers
 * and the carry-over stream state and is released by am_destroy().
 */
#ifndef AIRMODES_HIP_H
#define AIRMODES_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AM_ABI_VERSION 5

/* Export marker.  The library is built with -fvisibility=hidden: only what carries AM_API leaves the
 * shared object (the role AIR_MODES_API plays in the reference: include/gr_air_modes/api.h:27-31,
 * CMakeLists.txt:65 -- hidden by default, exported by annotation). */
#if defined(_WIN32)
#  define AM_API
#elif defined(__GNUC__) || defined(__clang__)
#  define AM_API __attribute__((visibility("default")))
#else
#  define AM_API
#endif

/* error codes */
#define AM_OK          0
#define AM_EINVAL     (-1)   /* bad argument (null pointer, rate below 2 MHz, ...)           */
#define AM_ENODEV     (-2)   /* no usable HIP device                                        */
#define AM_ENOMEM     (-3)   /* host or device allocation failed                            */
#define AM_EHIP       (-4)   /* a HIP runtime call failed (see am_last_error)               */
#define AM_ECAPACITY  (-5)   /* output did not fit; *n_out holds the required count         */
#define AM_ENOTSUP    (-6)   /* option not implemented (none at present)                    */

/* flags for the *_work / am_process_iq calls */
#define AM_F_DEVICE_IN  0x1u  /* input pointers are device memory on the context's GPU      */
#define AM_F_FLUSH      0x2u  /* end of stream: examine the tail under the reference's
                                 end-of-buffer rule (preamble_impl.cc:150,212)              */
#define AM_F_DEVICE_OUT 0x4u  /* am_frontend_work only: bb/avg are device pointers          */
#define AM_F_MORE       0x10u /* am_shard_scan / am_shard_scan_async: the stream goes on beyond total_n (= the samples so
                                 far): no end-of-stream rule; the chunk must come with its whole right halo            */
#define AM_F_ZERO_GAPS  0x20u /* am_process_multi: the library writes the zeros between the streams into the caller's buffer  */
#define AM_F_KEEP_TAGS  0x8u  /* am_process_iq / am_submit_iq: also keep what the preamble block hands the slicer for
                                 this call's hits -- 240-float bursts + "preamble_found" tags (lib/preamble_impl.cc:
                                 219-232) -- for am_fetch_tags                                  */

/* One accepted Mode-S reply: what slicer_impl.cc:186-194 serialises into its text message
 * (data, crc, reference_level, timestamp) plus the integer sample count behind the
 * timestamp.  56 bytes, little endian, natural alignment.
 * Replaces: struct modes_packet (include/gr_air_modes/types.h:29-40). */
typedef struct am_packet {
    uint8_t  data[14];     /* payload bits, MSB first; bytes beyond nbytes are zero        */
    uint8_t  nbytes;       /* 7 = short (56 bit), 14 = long (112 bit)                      */
    uint8_t  df;           /* downlink format = data[0] >> 3  (modes_packet.message_type)  */
    uint8_t  numlowconf;   /* number of low-confidence bits, saturating at 24              */
    uint8_t  reserved[3];
    uint32_t crc;          /* 24-bit syndrome: crc(data[0..nbytes-3)) ^ last 3 bytes       */
    float    ref;          /* reference_level (mean of the four preamble chips)            */
    uint32_t reserved2;
    uint64_t sample;       /* preamble item count: stream index + 2*spc - 1                */
    uint64_t secs;         /* sample / rate                (tag_to_timestamp, :124)        */
    double   frac;         /* (sample % rate) / rate       (tag_to_timestamp, :125); both   */
                           /* relative to the rx_time tag in force (am_set_rx_time)         */
} am_packet;

/* One preamble hit: the "preamble_found" stream tag (preamble_impl.cc:224-232) whose value
 * is the (uint64 secs, double frac) tuple, attached to sample 0 of a 240-float burst. */
typedef struct am_tag {
    uint64_t sample;
    uint64_t secs;
    double   frac;
    float    inavg;        /* reference level subtracted from the burst (:220)             */
    uint32_t how_late;     /* late-peak shifts applied (:184-192)                          */
} am_tag;

typedef struct am_ctx am_ctx;

AM_API uint32_t am_abi_version(void);

/* ---- context = the rx_path hier block ------------------------------------------------
 * Replaces: rx_path.__init__(rate, threshold, queue, use_pmf, use_dcblock)
 *           (python/rx_path.py:27-65) and the two block factories
 *           gr::air_modes::preamble::make(float channel_rate, float threshold_db)
 *           (include/gr_air_modes/preamble.h:39), gr::air_modes::slicer::make(queue)
 *           (include/gr_air_modes/slicer.h:41).
 * rate: samples per second, a whole number >= 2e6.  It need not be a multiple of 2 MHz: the reference keeps
 * d_samples_per_chip = channel_rate / 2e6 as a FLOAT and truncates every product with int() (lib/preamble_impl.cc:57,150,
 * 158-162,185,192,205-208,212,220,237), the flowgraph in front of it runs its moving averages at int(rate / 2e6) samples per
 * chip (python/rx_path.py:35) -- e.g. 5 Msps is 2.5 samples per chip for the preamble block and 2 for the filters.  That
 * geometry is reproduced as it is (pinned against the reference's own C++).  Multiples of 2 MHz up to 64 Msps (and 40) run the
 * specialised kernels, everything else the rate-generic ones.
 * use_dcblock != 0 puts filter.dc_blocker_cc(100*spc, False) in front of the path
 * (python/rx_path.py:39-41; default off, python/radio.py:118): two cascaded 100-chip moving
 * averages subtracted from the input delayed by 100*spc - 1 samples -- so, as in the reference,
 * every timestamp is later by that delay.  GNU Radio's gr-filter source is not part of the
 * reference tree (parity unpinned): the window sums use the canonical order of DESIGN.md section 3.
 * device < 0 selects the current HIP device.  Returns NULL on failure; *err (optional)
 * receives the code. */
AM_API am_ctx *am_create(int device, double rate, float threshold_db, int use_pmf, int use_dcblock,
                  int *err);
AM_API void am_destroy(am_ctx *ctx);

/* Replaces: preamble::set_rate / set_threshold / get_rate / get_threshold
 *           (include/gr_air_modes/preamble.h:41-44; lib/preamble_impl.cc:56-76) and
 *           rx_path.set_rate / set_threshold / get_threshold / get_pmf (rx_path.py:67-87).
 * am_set_rate also drops the carried stream state (window lengths change). */
AM_API int    am_set_rate(am_ctx *ctx, double rate);
AM_API int    am_set_threshold(am_ctx *ctx, float threshold_db);
/* Replaces: the "rx_time" stream tag a live source (UHD, osmosdr) attaches to the sample stream and
 *           preamble_impl::general_work latches (lib/preamble_impl.cc:165-170); tag_to_timestamp
 *           (lib/preamble_impl.cc:100-137) then stamps each preamble with
 *           (secs, frac) + (item - offset) / rate.
 * offset is the tag's item offset in the input stream (tags pass the front-end blocks 1:1), counted
 * from the first sample after am_create / am_reset / am_set_rate.  Call it before the am_process_iq
 * (am_preamble_work, am_shard_scan) call whose samples the tag belongs to, with non-decreasing offsets
 * (AM_EINVAL otherwise; a second tag at the same offset replaces the first, as .back() does).  The tag
 * is in force for every preamble whose item count (am_packet.sample) is >= offset, until the next tag;
 * before any tag the reference's default applies (offset 0, time 0: a file source).  In the reference a
 * tag is latched as soon as the scheduler's current window contains it, i.e. up to a buffer early
 * (the unsigned difference :124 then wraps for preambles in front of it); that scheduler-dependent
 * early latch is not reproduced.  am_reset and am_set_rate drop the pending tags. */
AM_API int    am_set_rx_time(am_ctx *ctx, uint64_t offset, uint64_t secs, double frac);
AM_API double am_get_rate(const am_ctx *ctx);
AM_API float  am_get_threshold(const am_ctx *ctx);
AM_API int    am_get_pmf(const am_ctx *ctx);
/* ---- batches in flight ---------------------------------------------------------------------------------------
 * Under GNU Radio every block of rx_path runs in its own thread, so the slicer works on burst k while the preamble
 * block scans ahead (thread-per-block scheduler; python/rx_path.py wires five blocks).  The counterpart here: the
 * launch-latency-bound tail of one batch (greedy chain, extraction, slicer) overlaps the streaming front end of the
 * next.  A context works on one batch at a time, so the overlap is between contexts:
 *
 * am_submit_iq   first half of am_process_iq: copies and kernels of one INDEPENDENT batch (AM_F_FLUSH is required: it
 *                is a whole stream) are enqueued, nothing is waited for.  The samples must stay valid until am_collect.
 * am_collect     second half: waits for the batch, returns its packets (same contract as am_process_iq, incl.
 *                AM_ECAPACITY + am_fetch_packets) and resets the stream state.
 * am_pipe_*      `depth` contexts behind one handle, used round-robin by ONE host thread: submit up to `depth`
 *                batches, collect them in submission order.  Results are those of am_process_iq on each batch. */
AM_API int am_submit_iq(am_ctx *ctx, const float *iq, uint64_t n_complex, uint32_t flags);
AM_API int am_collect(am_ctx *ctx, am_packet *out, uint64_t cap, uint64_t *n_out);
typedef struct am_pipe am_pipe;
AM_API am_pipe *am_pipe_create(int device, double rate, float threshold_db, int use_pmf, int use_dcblock, int depth, int *err);
AM_API void am_pipe_destroy(am_pipe *pipe);
AM_API int am_pipe_depth(const am_pipe *pipe);
AM_API int am_pipe_in_flight(const am_pipe *pipe);
/* AM_ECAPACITY when `depth` batches are already in flight (collect one first) */
AM_API int am_pipe_submit(am_pipe *pipe, const float *iq, uint64_t n_complex, uint32_t flags);
/* packets of the OLDEST batch in flight; AM_EINVAL when there is none */
AM_API int am_pipe_collect(am_pipe *pipe, am_packet *out, uint64_t cap, uint64_t *n_out);
/* K whole streams in ONE scan of the next free context (am_submit_multi's arguments); am_pipe_collect hands the packets out stream by
 * stream, am_pipe_multi_counts the number each stream of the scan collected LAST got.  (rx_path.py:35: one receiver chain per
 * stream -- here eight of them share a scan, and several scans are in flight behind one handle.) */
AM_API int am_pipe_submit_multi(am_pipe *pipe, float *iq, uint32_t k, const uint64_t *n_complex, uint32_t flags);
AM_API int am_pipe_multi_counts(am_pipe *pipe, uint64_t *count, uint32_t k);
AM_API const char *am_pipe_last_error(const am_pipe *pipe);
AM_API float am_pipe_last_kernel_ms(const am_pipe *pipe);   /* dominant-kernel time of the batch collected last */

/* ---- ONE continuing stream with several of its chunks in flight ---------------------------------------------------
 * The reference's preamble block is a streaming block: general_work() resumes where the last call stopped (lib/preamble_impl.cc:
 * 139-246; consume_each at :213,237,244).  am_process_iq is that, one chunk at a time.  am_spipe_* keeps `depth` consecutive
 * chunks of ONE stream in flight on one GPU: a chunk's scan (everything that does not depend on where the greedy scan enters the
 * chunk) is enqueued when it is submitted, on a context and stream of its own; the position at which the scan enters it travels on
 * the device, from the word the chunk before it leaves it in -- nothing between two chunks waits for the host.  The packets of all
 * chunks, concatenated in submission order, are those of one am_process_iq over the whole stream (item counts and time stamps keep
 * counting).
 *   am_spipe_front   samples of the stream that must lie IN FRONT of every chunk but the first, in the same allocation (the tail of
 *                    the chunk before it: reference-level history + the positions it takes over).  Where the stream is not
 *                    contiguous in memory the library copies them there from the chunk submitted before (device to device, on the
 *                    chunk's stream): a chunk buffer needs that much writable room in front of it.
 *   am_spipe_submit  iq: DEVICE pointer to the chunk's n_complex samples (n_complex > front); they, and the chunk submitted before
 *                    it, must stay valid until the chunk has been collected.  AM_F_FLUSH: the stream's last chunk (end-of-stream
 *                    rule); the next stream starts once every chunk has been collected.  AM_ECAPACITY: `depth` chunks in flight.
 *   am_spipe_collect the packets of the oldest chunk in flight.
 *   am_spipe_redone  chunks that had to take the synchronous path (exit table larger than its message, more candidates than the
 *                    capacity the scan was launched for): results are the same, the chunks behind such a chunk are scanned twice.
 *   am_spipe_set_rx_time  as am_set_rx_time (stream-absolute offsets), with no chunk in flight. */
typedef struct am_spipe am_spipe;
AM_API am_spipe *am_spipe_create(int device, double rate, float threshold_db, int use_pmf, int use_dcblock, int depth, int *err);
AM_API void am_spipe_destroy(am_spipe *pipe);
AM_API int am_spipe_depth(const am_spipe *pipe);
AM_API int am_spipe_in_flight(const am_spipe *pipe);
AM_API int am_spipe_front(const am_spipe *pipe, uint64_t *front);
AM_API int am_spipe_set_rx_time(am_spipe *pipe, uint64_t offset, uint64_t secs, double frac);
AM_API int am_spipe_submit(am_spipe *pipe, const float *iq, uint64_t n_complex, uint32_t flags);
AM_API int am_spipe_collect(am_spipe *pipe, am_packet *out, uint64_t cap, uint64_t *n_out);
AM_API uint64_t am_spipe_redone(const am_spipe *pipe);
AM_API const char *am_spipe_last_error(const am_spipe *pipe);
AM_API float am_spipe_last_kernel_ms(const am_spipe *pipe);   /* dominant-kernel time of the chunk collected last */

/* Run the context's device work on the caller's HIP stream (hipStream_t passed as a pointer; NULL: back to the
 * context's own stream).  For callers whose input is produced on a stream of their own -- e.g. halo samples that
 * arrive by an RCCL receive on a PyTorch stream: work enqueued here is then ordered behind it without a host
 * synchronisation.  The stream must outlive the context or be replaced before it is destroyed.  (No counterpart in
 * the reference: GNU Radio blocks have no device streams.) */
AM_API int am_set_stream(am_ctx *ctx, void *hip_stream);
/* Order the context's stream behind everything enqueued so far on another stream of the same device (NULL = the legacy
 * default stream, PyTorch's current stream unless the caller changed it): an event recorded there and waited for here,
 * the host does not block.  For inputs another stream produces -- e.g. boundary samples an RCCL receive is still
 * writing (air_modes/sharded.py). */
AM_API int am_wait_for_stream(am_ctx *ctx, void *hip_stream);

/* start a new stream: sample counter, carry-over samples and greedy-scan state are cleared */
AM_API int    am_reset(am_ctx *ctx);

/* ---- the fused hot path ---------------------------------------------------------------
 * Replaces, for one chunk of the input stream, everything rx_path wires together
 * (python/rx_path.py:38-65): complex_to_mag_squared -> moving_average_ff(spc) ->
 * moving_average_ff(48*spc) -> preamble::general_work (lib/preamble_impl.cc:139-246) ->
 * slicer::work (lib/slicer_impl.cc:102-198) -> modes_check_crc (lib/modes_crc.cc:55-63).
 *
 * iq: n_complex interleaved (I,Q) float32 pairs = what blocks.file_source(gr_complex)
 * delivers (python/radio.py:231); host memory, or device memory with AM_F_DEVICE_IN.
 * Chunks may have any length; results do not depend on how the stream is chunked.
 * Decisions that need look-ahead are deferred to a later call; AM_F_FLUSH ends the stream.
 * out/cap: caller's packet array (host).  *n_out = packets produced by this call, in
 * stream order.  AM_ECAPACITY if cap is too small (nothing is lost: call
 * am_fetch_packets with a larger array). */
AM_API int am_process_iq(am_ctx *ctx, const float *iq, uint64_t n_complex, uint32_t flags,
                  am_packet *out, uint64_t cap, uint64_t *n_out);
AM_API int am_fetch_packets(am_ctx *ctx, am_packet *out, uint64_t cap, uint64_t *n_out);

/* K independent streams in ONE scan (many receivers / channels on one GPU: at 2 and 20 Msps one second of one receiver is too
 * small a job for the chip, the scan's launches are what it costs).  Every stream is a WHOLE stream, as am_process_iq(...,
 * AM_F_FLUSH) takes it -- rx_path.work over one finite capture (python/rx_path.py:27-65) -- and gets exactly the packets that call
 * gives (bit for bit; item counts and time stamps are the stream's own, from its item 0).
 * am_multi_layout   where the streams go in the ONE buffer the scan reads: offset[j] (in complex samples) for stream j of n[j]
 *                   samples, *total = samples up to the end of the last one.  The samples between the streams must be ZERO (the
 *                   caller writes them once, or passes AM_F_ZERO_GAPS and the library writes them on every call).
 * am_process_multi  iq: that buffer (host, or device with AM_F_DEVICE_IN); out/cap/n_out as am_process_iq; the packets come
 *                   stream by stream, in stream order inside each; count[j] (may be NULL) = packets of stream j.
 *                   Not with use_dcblock; "rx_time" tags do not apply (every stream starts at 0 s);
 *                   the context's stream state is reset before and after. */
AM_API int am_multi_layout(am_ctx *ctx, uint32_t k, const uint64_t *n, uint64_t *offset, uint64_t *total);
AM_API int am_process_multi(am_ctx *ctx, float *iq, uint32_t k, const uint64_t *n, uint32_t flags,
                            am_packet *out, uint64_t cap, uint64_t *count, uint64_t *n_out);
/* The same in two halves (as am_submit_iq / am_collect): am_submit_multi enqueues the scan and returns (the buffer must stay valid
 * until the scan is collected); am_collect -- the call that collects a single-stream batch -- waits for it and hands out the
 * packets, stream by stream; am_multi_counts(count, k) then gives the packets per stream of the scan just collected. */
AM_API int am_submit_multi(am_ctx *ctx, float *iq, uint32_t k, const uint64_t *n, uint32_t flags);
AM_API int am_multi_counts(am_ctx *ctx, uint64_t *count, uint32_t k);
/* preamble hits (tags) seen by the last am_process_iq call, accepted or not */
AM_API uint64_t am_last_num_tags(const am_ctx *ctx);
/* The inter-block stream of the last am_process_iq / am_collect call that ran with AM_F_KEEP_TAGS: one 240-float burst
 * and one tag per preamble hit, in stream order, exactly what gr::air_modes::preamble would have produced for the
 * slicer (lib/preamble_impl.cc:219-232), from the SAME kernels that produced the call's packets.  bursts: cap*240
 * floats; tags: cap entries; either may be NULL to skip it.  AM_ECAPACITY (*n_out = needed) if cap is too small. */
AM_API int am_fetch_tags(am_ctx *ctx, float *bursts, am_tag *tags, uint64_t cap, uint64_t *n_out);

/* ---- block-level entry points (for block-by-block parity tests) -------------------------
 * am_frontend_work: the three third-party blocks in front of the preamble detector
 *   (rx_path.py:38,49,54) on a whole stream that starts at sample 0: bb = pulse-matched
 *   power, avg = reference level, n floats each.
 * am_preamble_work: gr::air_modes::preamble_impl::general_work run to completion over the
 *   two float streams `in` and `inavg` (n items each, stream starts at item 0; the block's
 *   history of 2*spc-1 zeros is implied).  bursts: cap*240 floats; tags: cap entries.
 * am_slicer_work: gr::air_modes::slicer_impl::work over nbursts tagged bursts.
 *   Only accepted packets are written to out. */
AM_API int am_frontend_work(am_ctx *ctx, const float *iq, uint64_t n_complex, uint32_t flags,
                     float *bb, float *avg);
AM_API int am_preamble_work(am_ctx *ctx, const float *in, const float *inavg, uint64_t n,
                     uint32_t flags, float *bursts, am_tag *tags, uint64_t cap,
                     uint64_t *n_out);
/* The preamble block as a STREAM: what gr::air_modes::preamble is under the scheduler (include/gr_air_modes/preamble.h:36-46;
 * general_work is called again and again on the next items, lib/preamble_impl.cc:139-246).  Consecutive calls on consecutive pieces
 * of the two input streams give, together, what ONE am_preamble_work over the concatenation gives: decisions wait for (240 + 4)
 * samples-per-chip items of look-ahead, the undecided tail of both inputs is carried inside the context, the greedy scan resumes
 * where it stopped (consume_each, :213,237,244), item counts and time stamps keep counting.  AM_F_FLUSH: these are the stream's last
 * items (end-of-buffer rule :150,212); the next call starts a new stream at item 0.  am_reset() drops the carried state.
 * bursts / tags: this call's hits, cap entries; AM_ECAPACITY with *n_out = the number needed if they do not fit (nothing is lost,
 * the stream has moved on: am_fetch_tags hands them out).  With AM_F_DEVICE_IN both inputs are device pointers. */
AM_API int am_preamble_stream(am_ctx *ctx, const float *in, const float *inavg, uint64_t n, uint32_t flags,
                              float *bursts, am_tag *tags, uint64_t cap, uint64_t *n_out);
AM_API int am_slicer_work(am_ctx *ctx, const float *bursts, const am_tag *tags, uint64_t nbursts,
                   uint32_t flags, am_packet *out, uint64_t cap, uint64_t *n_out);

/* ---- host-side helpers ------------------------------------------------------------------
 * am_crc24: modes_check_crc(data, length) (lib/modes_crc.cc:55-63): CRC-24, generator
 *   0xFFF409, zero initial value, over the first nbytes bytes.
 * am_format_message: the text slicer_impl.cc:186-192 posts to the gr::msg_queue:
 *   "<hex payload> <crc %06x> <reference_level> <secs> <frac>".  first != 0 formats the
 *   reference level with 6 significant digits (the first message a slicer instance emits),
 *   otherwise 10 (every later one) -- the member ostringstream keeps setprecision(10).
 *   Returns the length written (excluding NUL) or AM_ECAPACITY. */
AM_API uint32_t am_crc24(const uint8_t *data, int nbytes);
AM_API int am_format_message(const am_packet *pkt, int first, char *buf, size_t cap);
/* The same for n packets in one call (a binding posts a batch per call instead of crossing the FFI per packet): text k is
 * the NUL-terminated string at buf + offsets[k]; offsets has n + 1 entries, offsets[n] = bytes used.  `first` applies to
 * packet 0 only -- the stream's precision is sticky (slicer_impl.cc:186-194; the ostringstream is a member,
 * slicer_impl.h:43).  AM_ECAPACITY when cap is too small: *need (optional) then holds the bytes required (a text is at most
 * 96 bytes with its terminator: 28 hex digits, 6 of the syndrome, two %.10g numbers of up to 16 characters, a 20-digit count of
 * seconds, separators; typical ones take 55-65) and nothing beyond cap was written. */
AM_API int am_format_messages(const am_packet *pkts, uint64_t n, int first, char *buf, size_t cap, uint64_t *offsets,
                              uint64_t *need);

/* ---- time-sharded operation (one context per GPU, one contiguous time chunk each) -------
 * The reference's preamble scan is sequential, but the only state that crosses a chunk
 * boundary is ONE number: the position at which the scan resumes ("cur": after a hit at e the
 * scan skips to e + 240*spc, lib/preamble_impl.cc:237; after a rejected candidate to e + 1,
 * :209).  That position can reach at most 241*spc samples into the next chunk, so a chunk's
 * result depends on its predecessor only through which of its first few candidates (those in
 * its first 241*spc samples, the "lead-in") the scan enters at.  Protocol per step:
 *   1. every rank attaches its neighbours' boundary samples (am_shard_halo) and runs
 *      am_shard_scan on its chunk: front end, detection, refinement, successor array and block
 *      exits of its own greedy chain -- and an EXIT TABLE: for each lead-in candidate (plus the first candidate
 *      after the lead-in, if any) the position at which the scan would leave the chunk if it
 *      entered at that candidate;
 *   2. the small tables are exchanged (all-gather, a few KB), every rank composes them
 *      (am_shard_entry) to learn the position at which the scan enters ITS chunk;
 *   3. am_shard_resolve marks the chain from that entry and extracts + slices the chunk's hits.
 * The concatenation of all chunks' packets equals the single-stream result.
 *
 * am_shard_scan: iq holds samples [abs_start - left, abs_end + right) of the global stream
 *   (left/right from am_shard_halo, clipped to [0, total_n)); AM_F_DEVICE_IN if on the GPU.
 *   table/cap: caller's host array; *n_table entries written (AM_ECAPACITY if cap is too small,
 *   *n_table = needed).
 * am_shard_entry: tables[r] / counts[r] for r = 0..nranks-1 in chunk order, starts[r] = abs_start of
 *   chunk r; writes entry[r] = scan position when it reaches chunk r (entry[0] = 0). Host only.
 * am_shard_resolve: cur_in = entry of this rank's chunk. */
#define AM_SHARD_MSG_HEADER 2   /* header entries of a device message (am_shard_scan_async) */
typedef struct am_shard_exit {
    uint64_t pos;          /* absolute position of the candidate                            */
    uint64_t exit;         /* scan position after the chunk's last visited candidate, if the
                              scan enters the chunk at this candidate                       */
} am_shard_exit;

AM_API int am_shard_halo(const am_ctx *ctx, uint64_t *left, uint64_t *right);
AM_API int am_shard_scan(am_ctx *ctx, const float *iq, uint64_t abs_start, uint64_t abs_end,
                  uint64_t total_n, uint32_t flags, am_shard_exit *table, uint64_t cap,
                  uint64_t *n_table);
AM_API int am_shard_entry(const am_shard_exit *const *tables, const uint64_t *counts, const uint64_t *starts,
                   uint32_t nranks, uint64_t *entry);
AM_API int am_shard_resolve(am_ctx *ctx, uint64_t cur_in, am_packet *out, uint64_t cap, uint64_t *n_out);
/* A receiver, not a batch (round 4): the scan position crosses STEPS as it crosses chunks (lib/preamble_impl.cc:213,237,244:
 * consume_each -- the reference's scan resumes where the last general_work left off, for ever).  Step k covers the samples
 * [k W n, (k + 1) W n); rank r owns the POSITIONS [k W n + r n - H, k W n + (r + 1) n - H), H = `right` of am_shard_halo (the
 * look-ahead a decision needs), so its chunk needs left + H samples of the rank before it (of the last rank's previous
 * step for rank 0) and none of the rank after it; it scans with AM_F_MORE and total_n = the samples so far, and the last
 * step of the stream without the flag and with the true length.
 * am_shard_entry2: am_shard_entry with the position at which the scan left the last chunk of the step before (cur_in; 0 at
 *   the start of a stream); also writes leave[r] = where the scan leaves chunk r (either array may be NULL).
 * am_shard_get_exit / am_shard_set_exit: where the scan left THIS context's chunk in its last resolved step -- the word the
 *   host-free step keeps on the device and sends along in the next step's message header (below); a step that ran on the
 *   synchronous path sets it (leave[rank]), a caller that falls back to that path reads it. */
AM_API int am_shard_entry2(const am_shard_exit *const *tables, const uint64_t *counts, uint32_t nranks, uint64_t cur_in,
                    uint64_t *entry, uint64_t *leave);
/* The samples the NEXT step needs in front of a chunk are this step's last ones, and the caller is about to overwrite them:
 * am_shard_keep_tail registers a device-to-device copy (nbytes from src to dst; 0: none) that every following
 * am_shard_resolve / am_shard_resolve_async enqueues behind its slicing -- which still reads the samples -- and in front of
 * its completion, so that it is done when the call returns (a host-side copy after the call costs a launch and an event on
 * the critical path of the next step).  dst must not be part of what a repeated step would scan.
 * am_stream_copy: a device-to-device copy on the context's stream, ordered with its scans (e.g. the kept tail into the halo
 * in front of the chunk, before the next am_shard_scan_async). */
AM_API int am_shard_keep_tail(am_ctx *ctx, void *dst, const void *src, uint64_t nbytes);
AM_API int am_stream_copy(am_ctx *ctx, void *dst, const void *src, uint64_t nbytes);
/* (am_shard_get_exit / am_shard_set_exit, the exit word of the synchronous fallback: airmodes_hip_debug.h) */

/* The same step without a host round trip in the middle (round 3): the exit table stays on the device.
 * am_shard_scan_async: as am_shard_scan, but everything is only ENQUEUED and the table goes to the device message
 *   msg_dev = AM_SHARD_MSG_HEADER + msg_cap entries: entry 0 = {count, overflow} (count = msg_cap + 1: the table did not
 *   fit; overflow = 1: the scan met more candidates than the capacity it was launched for), entry 1 = {where the scan left
 *   this context's chunk in the step before, -} (how the scan position reaches the next step: every rank reads the last
 *   rank's), then the table.
 * The caller all-gathers the messages (device to device, e.g. torch.distributed over RCCL: am_signal_stream makes the
 *   collective's stream wait for the table, am_wait_for_stream the context for the collective) and hands all `world`
 *   of them, in chunk order, to
 * am_shard_resolve_async: composes the entry position of chunk `rank` on the device (am_shard_entry as a kernel), marks the
 *   chain from there, extracts and slices -- ONE completion wait per step.  *redo != 0: a table did not fit its message,
 *   or some rank's scan met more candidates than the capacity it was launched for (both rare, both read from the message
 *   headers: *redo is the same on every rank): no packets were delivered, repeat the step with am_shard_scan /
 *   am_shard_entry / am_shard_resolve. */
AM_API int am_shard_scan_async(am_ctx *ctx, const float *iq, uint64_t abs_start, uint64_t abs_end, uint64_t total_n,
                        uint32_t flags, am_shard_exit *msg_dev, uint64_t msg_cap);
AM_API int am_shard_resolve_async(am_ctx *ctx, const am_shard_exit *msgs_dev, uint32_t world, uint32_t rank, uint64_t msg_cap,
                           am_packet *out, uint64_t cap, uint64_t *n_out, int *redo);
/* another stream of the context's device (NULL: the legacy default stream) waits, on the device, for what the context has
 * enqueued so far (the counterpart of am_wait_for_stream) */
AM_API int am_signal_stream(am_ctx *ctx, void *hip_stream);

/* The resolve step in two halves (round 6: STEPS of the time-sharded receiver in flight -- the scan and the all-gather of step k + 1
 * are enqueued before step k is resolved, so that the collectives of one step hide behind the kernels of the next).
 * am_shard_resolve_submit only enqueues; am_shard_resolve_collect waits for the step's one completion ticket and hands the packets
 * out (*redo as am_shard_resolve_async).  cur_in_dev (device word, may be NULL): where the scan left the LAST chunk of the step
 * before, read when the entry is composed -- instead of the last rank's message header, which was written before that step was
 * resolved.  carry_out_dev (device word, may be NULL, may be the same word): where the scan leaves THIS step's last chunk, composed
 * by every rank for itself from everybody's tables -- the next step's cur_in. */
AM_API int am_shard_resolve_submit(am_ctx *ctx, const am_shard_exit *msgs_dev, uint32_t world, uint32_t rank, uint64_t msg_cap,
                                   const uint64_t *cur_in_dev, uint64_t *carry_out_dev);
AM_API int am_shard_resolve_collect(am_ctx *ctx, am_packet *out, uint64_t cap, uint64_t *n_out, int *redo);

/* ---- resampling in front of the path (python/radio.py:49-53) ------------------------------------------------
 * modes_radio resamples anything slower than 4 Msps to 4 Msps with pfb.arb_resampler_ccf before rx_path.  GNU Radio's
 * block and tap design are not in the reference tree (parity unpinned); this is the package's own 32-phase x 8-tap
 * polyphase interpolator, DEFINED operation by operation in air_modes/resample.py and repeated bit for bit on the GPU.
 * taps: 32 x 8 doubles [phase][tap] (air_modes.resample.design_taps).  ratio = f_out / f_in >= 1.
 * am_resampler_work: n complex input samples (host, or device with AM_F_DEVICE_IN) -> *n_out complex output samples,
 *   copied to `out` (host, or device with AM_F_DEVICE_OUT; cap complex samples) or, when out is NULL, left in the
 *   handle's device buffer (am_resampler_device_output, valid until the next call: hand it to am_process_iq with
 *   AM_F_DEVICE_IN).  The read position and the last 8 input samples carry over from call to call. */
typedef struct am_resampler am_resampler;
AM_API am_resampler *am_resampler_create(int device, double ratio, const double *taps, int *err);
AM_API void am_resampler_destroy(am_resampler *h);
AM_API int am_resampler_reset(am_resampler *h);
AM_API int am_resampler_work(am_resampler *h, const float *iq, uint64_t n_complex, uint32_t flags, float *out, uint64_t cap,
                      uint64_t *n_out);
AM_API const float *am_resampler_device_output(const am_resampler *h);
AM_API const char *am_resampler_last_error(const am_resampler *h);

/* ---- pinned staging for a host source (apps/modes_rx's file source, python/radio.py:221-234) --------------------
 * nslots pinned host buffers of capacity_complex samples each, with a device twin and a copy stream: the reader fills
 * am_uploader_host(slot), am_uploader_start begins the copy and returns, am_uploader_wait blocks until that slot's
 * samples are on the device and returns the device pointer (for am_process_iq / am_resampler_work with
 * AM_F_DEVICE_IN) -- the transfer of chunk k+1 overlaps the scan of chunk k.  start and wait may be called from
 * different threads (one producer, one consumer); a slot is reused only after its consumer is done with it. */
typedef struct am_uploader am_uploader;
AM_API am_uploader *am_uploader_create(int device, uint64_t capacity_complex, int nslots, int *err);
AM_API void am_uploader_destroy(am_uploader *u);
AM_API float *am_uploader_host(am_uploader *u, int slot);
AM_API int am_uploader_start(am_uploader *u, int slot, uint64_t n_complex);
AM_API const float *am_uploader_wait(am_uploader *u, int slot);

/* last error text of the context (or of am_create when ctx == NULL) */
AM_API const char *am_last_error(const am_ctx *ctx);

/* Diagnostics, test hooks and the time-shard path's exit word (am_last_timing, am_last_num_candidates, am_last_frontend,
 * am_fetch_candidates, am_is_emulated, am_shard_get_exit, am_shard_set_exit) are declared in airmodes_hip_debug.h: they are
 * exported by the same library but are not part of the drop-in surface. */

#ifdef __cplusplus
}
#endif
#endif /* AIRMODES_HIP_H */
