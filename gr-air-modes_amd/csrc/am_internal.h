// am_internal.h -- kernel launch interface between am_capi.hip (host logic) and
// am_kernels.hip (gfx950 kernels).  Not part of the public ABI.
#ifndef AM_INTERNAL_H
#define AM_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "airmodes_hip.h"
#include "airmodes_hip_debug.h"

#ifndef AM_WITH_TILE_KERNEL
#define AM_WITH_TILE_KERNEL 0   /* 1 in TEST builds (tests/gpu_variants, tests/emu): the tile kernel am_k_fe2 and the split refinement behind it */
#endif
#define AM_CHIPS_AVG 48    /* reference level window in chips  (python/rx_path.py:54)      */
#define AM_BURST 240       /* chips handed to the slicer       (lib/preamble_impl.cc:219)  */
#define AM_WAVE 64

int am_device_cus(void);   /* compute units of the current device (cached)                  */

/* Geometry of the preamble block for one sample rate, in the reference's own arithmetic: d_samples_per_chip is a FLOAT
 * (lib/preamble_impl.cc:57) and every use of it is a float product truncated by int() -- for a rate that is not a multiple
 * of 2 MHz (5 Msps: 2.5 samples per chip) that gives a slightly crooked but well-defined geometry.  With whole samples per
 * chip all of it reduces to multiples of spc (o1 = 2 spc, za0 = 3 spc, ..., B = 240 spc, chip j at j spc). */
struct am_geom {
    int S;              /* int(samples per chip): granularity of ninputs (:150), correlation window (:90-98, :185)  */
    int hist0;          /* history items in front of the stream: set_history(d_samples_per_symbol) - 1 (:62)       */
    int o1, o2, o3;     /* pulse offsets int(2 spc), int(7 spc), int(9 spc) (:158-162)                              */
    int late_max;       /* the late-peak loop runs while how_late < d_samples_per_chip (:192): at most this many     */
    int za0, za1;       /* quiet zone 1: j = int(1.5 sps) .. j <= 3 sps, inclusive (:205)                            */
    int zb0, zb1;       /* quiet zone 2: j = int(5 sps) .. j <= 7.5 sps, inclusive (:207)                            */
    int B;              /* items consume_each() skips after a hit: int(240 spc) (:237)                               */
    int room;           /* a hit needs `ninputs - i < 240 spc` to be false (:212): at least this many items          */
    int span;           /* int(239 spc): offset of the last soft chip (:220)                                         */
};

/* ---- front end ---------------------------------------------------------------------- */
#define AM_FE_THREADS 512
#define AM_FE_LDS_BUDGET (80 * 1024)

struct am_fe_args {
    const float *iq;      /* interleaved I,Q; element 0 is absolute sample src_abs0        */
    long long src_abs0;   /* absolute index of the first sample present in iq              */
    long long src_abs1;   /* absolute end (exclusive) of the samples present               */
    long long out_abs0;   /* absolute index of bb[0]/avg[0]; multiple of 48*spc            */
    long long out_n;      /* number of outputs wanted                                      */
    float *bb;
    float *avg;
    int spc;
    int use_pmf;
    int tile;             /* outputs per workgroup; multiple of 48*spc                     */
    float s1;             /* float(1.0/spc)                                                */
    float sL;             /* float(1.0/(48*spc))                                           */
};
size_t am_fe_lds_bytes(int spc, int tile);
int am_fe_pick_tile(int spc);                  /* largest tile within AM_FE_LDS_BUDGET, 0 if none */
hipError_t am_launch_frontend(const am_fe_args &a, hipStream_t s);

/* fused front end + detection, specialised per samples-per-chip (am_fe2.hip).
 * am_fe2_tile(spc): tile length of the specialisation, 0 if spc has none (generic kernels). */
unsigned am_fe2_tile(int spc);
hipError_t am_launch_fe2(int spc, const float *iq, long long src_abs0, long long src_abs1, long long out_abs0,
                         long long out_n, float *bb, float *avg, uint32_t j0, uint32_t j1, int use_pmf, float s1,
                         float sL, float thr_lin, uint32_t *seg_pos, float *avg_sparse, uint32_t *blk_cnt,
                         unsigned *ntiles, unsigned *tile_len, hipStream_t s);
/* streaming fused front ends (am_fe4.hip; am_fe3.hip at 64 Msps): persistent workgroups, LDS rings, sparse outputs.  A lane takes a unit of G chips
 * = am_fe4_unit(spc) samples (one 32-sample chip at 64 Msps).  Candidates leave as a bitmap, a word per unit: bit b of word w =
 * array coordinate w * am_fe4_unit(spc) + b - am_fe4_lag(spc); a step has am_fe4_waves(spc) x am_fe4_words(spc) words (waves of 64
 * words at 20, 10 and 2 Msps, of 48 otherwise).  wg_cnt[g] = candidates workgroup g found; wg_max[g] (nsteps + 8
 * floats are enough) = the largest bb workgroup g formed, +inf if one was not finite; workgroup g formed the bb of the array
 * coordinates [g * steps_per_wg * tile, (g + 1) * steps_per_wg * tile) (and some before them). */
/* (behind am_fe4_* / am_launch_fe4 at 64 Msps: am_k_fe3, am_fe3.hip -- the same machine with one 32-sample chip per lane and
 * 16-byte LDS rows; two segments of 48 words per step, lag 288) */
unsigned am_fe3_tile(void);                 /* positions per step (3072) */
unsigned am_fe3_lag(void);                  /* 288 */
unsigned am_fe3_waves(void);                /* segments (waves, 48 chips = 48 bitmap words each) per step */
unsigned am_fe3_steps(long long out_n);
hipError_t am_launch_fe3(const float *iq, long long src_abs0, long long src_abs1, long long out_abs0, long long out_n,
                         float *bb_sparse, float *avg_sparse, uint32_t j0, uint32_t j1, int use_pmf, float s1, float sL,
                         float thr_lin, uint32_t *bits, uint32_t *wg_cnt, float *wg_max, unsigned *nsteps,
                         unsigned *steps_per_wg, hipStream_t s, int wgs_per_cu, unsigned *n_long = nullptr);
/* (n_long != null: LEVELLED segments -- the first *n_long workgroups take *steps_per_wg steps, the others one fewer, the grid is as
 * many workgroups as are resident; only a caller whose later kernels can place two segment lengths asks for it: am_k_refine_seg) */
int am_fe4_supported(int spc);
unsigned am_fe4_unit(int spc);
unsigned am_fe4_words(int spc);
unsigned am_fe4_waves(int spc);
unsigned am_fe4_tile(int spc);
unsigned am_fe4_lag(int spc);
unsigned am_fe4_steps(long long out_n, int spc);
hipError_t am_launch_fe4(int spc, const float *iq, long long src_abs0, long long src_abs1, long long out_abs0, long long out_n,
                         float *bb_sparse, float *avg_sparse, uint32_t j0, uint32_t j1, int use_pmf, float s1, float sL,
                         float thr_lin, uint32_t *bits, uint32_t *wg_cnt, float *wg_max, unsigned *nsteps,
                         unsigned *steps_per_wg, hipStream_t s, int wgs_per_cu, unsigned *n_long = nullptr);
/* (wgs_per_cu: 0 = as many persistent workgroups as are resident at once; n > 0 = at most n per CU -- am_pipe leaves room on
 * every CU for the small kernels of the other batches in flight; honoured by am_k_fe3) */
/* flat candidate positions from the bitmap, one workgroup per front-end workgroup: workgroup g owns the words
 * [g * words_per_wg, (g + 1) * words_per_wg) (nwords in all) and starts its part of pos[] at the sum of wg_cnt[0 .. g)
 * -- no scan launch, no chain.  Entries at or beyond Mcap are dropped; *total_out = the number of candidates. */
/* rows != null && rows->iq (64 Msps: am_k_fe3 no longer writes them): the same launch also forms, from the samples, the bb rows
 * the refinement reads -- the 17 chips from every candidate's chip on, canonical order (am_rows_segment32). */
struct am_rows_args {
    const float *iq;                      /* null: no rows (the front end wrote them: am_k_fe4 rates)        */
    long long src_abs0, src_abs1;         /* absolute range of samples present in iq                        */
    long long out_abs0;                   /* absolute index of array coordinate 0                           */
    long long out_n;                      /* array coordinates with data (nothing is stored at or beyond)   */
    float *bb_sparse;
    float *bb_max;                        /* [array chip] the largest bb of every row formed (what am_k_refine_late takes for a chip that
                                             lies inside a quiet zone as a whole) */
    int use_pmf;
    float s1;
};
hipError_t am_launch_gather_wg(const uint32_t *bits, const uint32_t *wg_cnt, uint32_t nwg, uint32_t words_per_wg,
                               uint32_t nwords, uint32_t Mcap, uint32_t lag, uint32_t wbits, uint32_t *pos,
                               uint32_t *total_out, hipStream_t s, const am_rows_args *rows = nullptr);
/* 64 Msps (round 6): candidate list + bb rows + refinement in ONE launch, one workgroup per front-end workgroup, the rows in LDS
 * (am_refine_seg.hip): what am_launch_gather_wg(rows) + am_launch_refine_late(bb_max) leave in pos / e / tgt / inavg / valid / jump0,
 * bit for bit, without the bb rows ever reaching memory.  rows: the samples (bb_sparse / bb_max are not used).  Segments: the first
 * n_long front-end workgroups tested words_per_wg words each, the others words_per_wg - words_per_step (levelled: am_launch_fe3) */
hipError_t am_launch_refine_seg(const uint32_t *bits, const uint32_t *wg_cnt, const float *wg_max, uint32_t nwg, uint32_t n_long,
                                uint32_t words_per_wg, uint32_t words_per_step, uint32_t nwords, uint32_t Mcap, uint32_t lag, uint32_t wbits, uint32_t vspan, uint32_t nv,
                                const am_rows_args &rows, const float *avg_sparse, float thr_lin, uint32_t end_j, uint32_t *pos,
                                uint32_t *e, uint32_t *tgt, float *inavg, uint8_t *valid, uint32_t *jump0, uint32_t *total_out,
                                hipStream_t s);
/* split refinement (after the fused kernel in split mode): flat candidate positions, one energy per
 * reachable position (deduplicated across neighbouring candidates), then one lane per candidate */
hipError_t am_launch_gather_pos(const uint32_t *seg_pos, uint32_t seg_stride, const uint32_t *blk_off,
                                uint32_t nseg, uint32_t M, int spc, uint32_t *pos, uint32_t *dcount,
                                hipStream_t s, const uint32_t *Mp = nullptr);
hipError_t am_launch_energy(const float *bb, const uint32_t *pos, const uint32_t *dcount,
                            const uint32_t *off_local, const uint32_t *blk_base, uint32_t M, int spc,
                            double *energy, hipStream_t s, const uint32_t *Mp = nullptr);
hipError_t am_launch_cand(const float *bb, const float *avg_sparse, const uint32_t *pos, const uint32_t *dcount,
                          const uint32_t *off_local, const uint32_t *blk_base, const double *energy, uint32_t M,
                          int spc, float thr_lin, uint32_t end_j, uint32_t *e, uint32_t *tgt, float *inavg,
                          uint8_t *valid, uint32_t *jump0, hipStream_t s, const uint32_t *Mp = nullptr);
/* behind the streaming front ends: late-peak decisions + per-candidate test + record + successor in one launch (the
 * decisions stay in LDS; no dcount / compact offsets).  vmax[array coordinate / vspan] (nv entries) bounds the samples */
hipError_t am_launch_refine_late(const float *bb, const float *avg_sparse, const uint32_t *pos, uint32_t M, int spc,
                                 float thr_lin, uint32_t end_j, uint32_t *e, uint32_t *tgt, float *inavg, uint8_t *valid,
                                 uint32_t *jump0, hipStream_t s, const uint32_t *Mp, const float *vmax, uint32_t vspan,
                                 uint32_t nv, const float *bb_max = nullptr);
/* (bb_max, 32 samples per chip only: bb_max[c] = the largest bb of array chip c for every chip whose row exists -- am_k_gather_wg<1>
 * writes it with the rows; a quiet zone's whole chips are then judged by six numbers instead of 192 samples) */
/* exclusive scan of n counts in ONE launch (2048 per workgroup, chained through slots[]: am_chain_prefix);
 * slots: one 64-bit word per workgroup, zero at allocation; epoch: a value no earlier launch on these slots used */
hipError_t am_launch_exscan_chain(const uint32_t *in, uint32_t *out, uint32_t n, unsigned long long *slots, uint32_t epoch,
                                  uint32_t *total_out, uint32_t *err, uint32_t *ticket, uint32_t *ticket_base, hipStream_t s,
                                  const uint32_t *Mp = nullptr);   /* *err = 1: the chain gave up; ticket: device counter,
                                  *ticket_base: the value it has when this launch starts (advanced by the grid size) */
hipError_t am_launch_ticket(uint32_t *host_word, uint32_t seq, hipStream_t s, const uint32_t *count_src = nullptr,
                            uint32_t *count_dst = nullptr, const uint64_t *word_src = nullptr, uint64_t *word_dst = nullptr);

/* ---- optional DC blocker in front of the path (am_dcblock.hip) ---------------------------- */
#define AM_DC_CHIPS 100              /* rx_path.py:40  dc_blocker_cc(100*spc, False) */
unsigned long long am_dcblock_history(int spc);
hipError_t am_launch_dcblock(const float *raw, long long raw_abs0, long long raw_abs1, long long y_abs0, long long y_n,
                             int spc, float *m1, float *y, hipStream_t s);

/* Launchers that take a candidate count M also take an optional device pointer Mp: when given, the
 * kernels use min(M, *Mp), so that the host may launch for a capacity without knowing the count. */
/* ---- preamble detection / refinement / greedy chain ----------------------------------- */
#define AM_DET_THREADS 256
#define AM_DET_PER_THREAD 8
#define AM_DET_PER_BLOCK (AM_DET_THREADS * AM_DET_PER_THREAD)

struct am_scan_buffers {
    /* per detect block */
    uint32_t *cand_seg;    /* nblk * AM_DET_PER_BLOCK candidate positions (segmented)       */
    uint32_t *blk_cnt;     /* nblk                                                          */
    uint32_t *blk_off;     /* nblk + 1 (exclusive scan, last = total)                       */
    /* per candidate (flat, position order) */
    uint32_t *pos;         /* position where the first-stage test fired                     */
    uint32_t *e;           /* shifted preamble start                                        */
    uint32_t *tgt;         /* where the scan resumes after this candidate                   */
    uint8_t *valid;
    uint8_t *visited;
    uint8_t *emit;
    uint32_t *jump;        /* (levels+1) * (M+1) pointer-doubling tables                    */
    uint32_t *emit_idx;    /* compacted indices of emitted candidates                       */
    uint32_t *cblk_cnt;    /* compaction block counts / offsets                             */
    uint32_t *cblk_off;
    uint32_t *scalars;     /* [0] = final scan position, [1] = visited&valid count          */
};

hipError_t am_launch_detect(const float *bb, const float *avg, uint32_t j0, uint32_t j1, const am_geom &g,
                            float thr_lin, uint32_t *cand_seg, uint32_t *blk_cnt, uint32_t nblk,
                            hipStream_t s);
/* exclusive scan of n counts into off[0..n]; off[n] = total */
hipError_t am_launch_scan_u32(const uint32_t *cnt, uint32_t *off, uint32_t n, hipStream_t s);
hipError_t am_launch_refine(const float *bb, const float *avg, const am_geom &g, float thr_lin,
                            const uint32_t *cand_seg, uint32_t seg_stride, const uint32_t *blk_off,
                            uint32_t nblk, uint32_t M, uint32_t *pos, uint32_t *e, uint32_t *tgt,
                            float *inavg, uint8_t *valid, hipStream_t s);
/* greedy chain (am_kernels.hip): step 1 is independent of where the scan starts */
size_t am_chain_scratch_bytes(uint32_t M);
hipError_t am_launch_chain_prepare(const uint32_t *pos, const uint32_t *tgt, uint32_t M, uint32_t *jump0,
                                   uint32_t *scratch, int want_last, hipStream_t s, const uint32_t *Mp = nullptr,
                                   int have_succ = 0);   /* have_succ: jump0[] was written by am_k_cand */
/* where the greedy scan of a time shard starts: composed on the device from everybody's exit tables (am_k_cblk_walk) */
struct am_entry_src {
    const am_shard_exit *msgs = nullptr;      // null: the start position comes from the host
    uint32_t world = 0, rank = 0, cap = 0;
    uint64_t base_abs = 0;          // absolute index of the chunk's array coordinate 0
    uint32_t *flags = nullptr;      // flags[0] = 1: repeat the step
    uint64_t *exit_out = nullptr;   // where the scan leaves this chunk (absolute): the next step's message carries it
    uint64_t *carry_out = nullptr;  // non-null: where the scan leaves the step's LAST chunk (composed through all ranks' tables), for the next step
    const uint64_t *cur_in = nullptr;   // non-null: where the scan left the chunk BEFORE this one, read when the entry is composed (am_spipe)
};

hipError_t am_launch_chain_visit(const uint32_t *pos, const uint32_t *jump0, uint32_t M, uint32_t cur0,
                                 uint32_t *scratch, const uint8_t *valid, const uint32_t *e, const uint32_t *tgt,
                                 uint32_t emit_max, uint32_t own_lo, uint32_t own_hi, uint4 *emit_idx, uint32_t *n_out,
                                 unsigned long long *slots, uint32_t epoch, uint32_t *ticket, uint32_t *ticket_base,
                                 uint32_t *scalars, int want_resume,
                                 hipStream_t s, const uint32_t *Mp, const am_entry_src *entry_src, const float *inavg,
                                 hipEvent_t after_walk = nullptr);
/* (emit_idx: one 16-byte record per hit -- candidate index, first-stage position, refined position, reference level -- so that
 * the extraction kernels fetch a hit with one load) */
/* lead_end (array coordinate): the table is only needed up to the first candidate at or past it */
hipError_t am_launch_chain_exit_table(const uint32_t *pos, const uint32_t *tgt, uint32_t M, uint32_t n,
                                      uint32_t lead_end, uint32_t *scratch, uint64_t base_abs, am_shard_exit *table,
                                      hipStream_t s, const uint32_t *Mp = nullptr, am_shard_exit *header = nullptr,
                                      const uint64_t *carry = nullptr);   /* header[1] = {*carry, 0}: where the scan left this chunk a step ago */
/* device-side am_shard_entry2: `world` messages of AM_SHARD_MSG_HEADER + cap entries; writes the array coordinate at which the
 * scan enters chunk `rank`, the absolute position at which it leaves it, and sets flags[0] if some table did not fit */
hipError_t am_launch_shard_entry(const am_shard_exit *msgs, uint32_t world, uint32_t rank, uint32_t cap, uint64_t base_abs,
                                 uint32_t *cur0_out, uint32_t *flags, uint64_t *exit_out, hipStream_t s, const uint64_t *cur_in = nullptr,
                                 uint64_t *carry_out = nullptr);
/* the header of a message without a table: {0, 0}, {*carry, 0} */
hipError_t am_launch_shard_header(am_shard_exit *header, const uint64_t *carry, hipStream_t s);

/* ---- burst extraction + slicer + CRC --------------------------------------------------- */
/* "rx_time" stream tag (lib/preamble_impl.cc:165-170): from item `offset` on, time = (secs, frac) +
 * (item - offset) / rate */
struct am_time_tag {
    uint64_t offset;
    uint64_t secs;
    double frac;
};
#define AM_MAX_TIME_TAGS 4096
/* extraction + slicing of the emitted preambles in one launch.  n_ptr: device-side number of hits; n_max:
 * upper bound used for the grid; tt[0..ntt): device array of time tags in ascending offset order;
 * bursts_out / tags_out may be null */
/* chip_idx: device table of the 240 soft chips' sample offsets int(j * samples per chip) (null: j * spc); hist0: history
 * items of the preamble block (the tag's item count = stream index + hist0) */
hipError_t am_launch_extract_slice(const float *bb, const float *inavg, int spc, const int *chip_idx, int hist0,
                                   const uint4 *emit_idx,
                                   const uint32_t *n_ptr, uint32_t n_max, const uint32_t *pos, const uint32_t *e,
                                   uint64_t base_abs, long long e_off, uint64_t rate, const am_time_tag *tt,
                                   uint32_t ntt, float *bursts_out, am_tag *tags_out, const uint32_t *crc_pow,
                                   am_packet *packets, const uint32_t *scalars, uint32_t *host_out, hipStream_t s,
                                   const uint32_t *Mp = nullptr);
/* the same when bb exists only around the candidates: the burst is recomputed from IQ (iq[0] = absolute sample src_abs0) */
hipError_t am_launch_extract_slice_iq(const float *iq, long long src_abs0, long long src_abs1, int use_pmf, float s1,
                                      const float *inavg, int spc, const uint4 *emit_idx, const uint32_t *n_ptr,
                                      uint32_t n_max, const uint32_t *pos, const uint32_t *e, uint64_t base_abs,
                                      uint64_t rate, const am_time_tag *tt, uint32_t ntt, float *bursts_out,
                                      am_tag *tags_out, const uint32_t *crc_pow, am_packet *packets,
                                      const uint32_t *scalars, uint32_t *host_out, hipStream_t s,
                                      const uint32_t *Mp = nullptr);
/* packets[i].reserved[0] = 1 when the reference would post the message, else 0 */
hipError_t am_launch_slice(const float *bursts, const am_tag *tags, const uint32_t *n_ptr, uint32_t n_max,
                           const uint32_t *crc_pow, am_packet *packets, const uint32_t *scalars,
                           uint32_t *host_out, hipStream_t s, const uint32_t *Mp = nullptr);

#endif
