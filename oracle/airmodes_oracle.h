/*
 * airmodes_oracle.h -- CPU oracle for the Mode-S rx_path hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / reported baseline.  The shipped path is
 * gr-air-modes_amd/csrc (HIP, gfx950) behind include/airmodes_hip.h.
 *
 * This is an independent restatement (plain C) of the semantics of the
 * reference hot path, written from the behaviour of:
 *   /root/reference/python/rx_path.py:35-65        (block order, window lengths, scales)
 *   /root/reference/lib/preamble_impl.cc:56-68,90-98,100-137,139-246
 *   /root/reference/lib/slicer_impl.cc:67-100,102-198
 *   /root/reference/lib/modes_crc.cc:33-63
 *   /root/reference/include/gr_air_modes/types.h:26-45
 * Parity status: the preamble/slicer/CRC stages are PINNED against the
 * reference's own C++ compiled from /root/reference (oracle/_ref, see
 * oracle/Makefile and tests/golden/); the |.|^2 + moving-average front end is
 * third-party GNU Radio code that is not in /root/reference, so its summation
 * order is a documented canonical choice ("parity unpinned" for that stage,
 * see DESIGN.md section 3).
 */
#ifndef AIRMODES_ORACLE_H
#define AIRMODES_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same byte layout as am_packet in include/airmodes_hip.h (56 bytes). */
typedef struct amo_packet {
    uint8_t  data[14];     /* MSB-first payload bits, zero padded            */
    uint8_t  nbytes;       /* 7 (short) or 14 (long)                          */
    uint8_t  df;           /* data[0] >> 3                                    */
    uint8_t  numlowconf;   /* min(#low-confidence bits, 24)                   */
    uint8_t  reserved[3];
    uint32_t crc;          /* 24-bit syndrome (0 == clean DF11/DF17)          */
    float    ref;          /* preamble reference level                        */
    uint32_t reserved2;
    uint64_t sample;       /* preamble-block item count = stream index+2*spc-1*/
    uint64_t secs;         /* sample / rate                                   */
    double   frac;         /* (sample % rate) / rate                          */
} amo_packet;

/* One preamble hit as the reference's preamble block tags it. */
typedef struct amo_tag {
    uint64_t sample;       /* item count k of the (shifted) preamble start    */
    uint64_t secs;
    double   frac;
    float    inavg;        /* moving-average level subtracted from the burst  */
    uint32_t how_late;     /* number of late shifts applied (diagnostic)      */
} amo_tag;

/* detection threshold exactly as preamble_impl.cc:67 derives it */
float amo_threshold_lin(float thr_db);

/* a1: |iq|^2, two rounded products and one rounded add, no FMA */
void amo_mag2(const float *iq, uint64_t n, float *m);

/* a3/a4: the canonical front end.  bb = pulse-matched (or raw) power,
 * avg = 48-chip reference level.  Both have n entries. */
int amo_frontend(const float *iq, uint64_t n, int spc, int use_pmf,
                 float *bb, float *avg);

/* Alternative front end used ONLY to quantify sensitivity to GNU Radio's
 * scheduler-dependent running-sum order (re-seeded every `chunk` outputs). */
int amo_frontend_running(const float *iq, uint64_t n, int spc, int use_pmf,
                         uint32_t chunk, float *bb, float *avg);
/* the same with the first re-seed after `first` outputs (0: after `chunk`) */
int amo_frontend_running2(const float *iq, uint64_t n, int spc, int use_pmf,
                          uint32_t chunk, uint32_t first, float *bb, float *avg);

/* a5-a9: greedy preamble scan over the two float streams the reference block
 * sees (in = bb, inavg = avg), canonical whole-stream semantics.
 * bursts: cap*240 floats, tags: cap entries.  Returns number of hits (may
 * exceed cap; only the first cap are stored). */
uint64_t amo_preamble_scan(const float *bb, const float *avg, uint64_t n,
                           int spc, float thr_db, uint64_t rate,
                           float *bursts, amo_tag *tags, uint64_t cap);

/* every first-stage candidate (preamble_impl.cc:172-179) refined on its own (:182-209), independent of the scan
 * order; coordinates are item counts incl. the block's history (like amo_tag.sample), positions k < k_limit */
uint64_t amo_candidates(const float *bb, const float *avg, uint64_t n, int spc, float thr_db, uint64_t k_limit,
                        uint64_t *pos, uint64_t *refined, uint8_t *valid, float *inavg_out, uint64_t cap);
/* the same for any sample rate (samples per chip = (float)rate / 2e6, the reference's own float arithmetic) */
uint64_t amo_candidates_r(const float *bb, const float *avg, uint64_t n, uint64_t rate, float thr_db, uint64_t k_limit,
                          uint64_t *pos, uint64_t *refined, uint8_t *valid, float *inavg_out, uint64_t cap);

/* a10-a12: slice one 240-chip burst.  Returns 1 if the packet is accepted
 * (the reference would post a message), 0 if it is dropped. */
int amo_slice(const float *burst, const amo_tag *tag, amo_packet *out);

/* a12: CRC-24 over nbytes bytes, poly 0xFFF409, init 0 (modes_crc.cc:55-63) */
uint32_t amo_crc24(const uint8_t *data, int nbytes);

/* whole path: interleaved IQ -> accepted packets.  Returns packet count (may
 * exceed cap).  n_tags (optional) receives the number of preamble hits. */
uint64_t amo_demod(const float *iq, uint64_t n, double rate, float thr_db,
                   int use_pmf, amo_packet *out, uint64_t cap,
                   uint64_t *n_tags);

/* a2 (optional, python/rx_path.py:39-41): dc_blocker_cc(100*spc, False) in front of |.|^2, canonical
 * summation order (see the .c file); out = 2n floats.  amo_demod2 = amo_demod with that option. */
int amo_dcblock(const float *iq, uint64_t n, int spc, float *out);
uint64_t amo_demod2(const float *iq, uint64_t n, double rate, float thr_db, int use_pmf, int use_dcblock,
                    amo_packet *out, uint64_t cap, uint64_t *n_tags);

/* rx_time stream tags (preamble_impl.cc:100-137,165-170): tags sorted by offset; the one in force for item
 * count k is the last with offset <= k (none: offset 0, time 0).  The restamp helpers recompute
 * secs/frac of packets / preamble tags from their .sample under a tag list. */
typedef struct amo_time_tag {
    uint64_t offset;       /* item count the time stamp belongs to            */
    uint64_t secs;
    double   frac;
} amo_time_tag;
void amo_timestamp(uint64_t k, uint64_t rate, const amo_time_tag *tt, uint64_t ntt, uint64_t *secs, double *frac);
void amo_restamp_packets(amo_packet *p, uint64_t n, uint64_t rate, const amo_time_tag *tt, uint64_t ntt);
void amo_restamp_tags(amo_tag *t, uint64_t n, uint64_t rate, const amo_time_tag *tt, uint64_t ntt);

/* slicer_impl.cc:186-192 message text incl. the sticky-precision quirk:
 * first != 0 -> reference level printed with 6 significant digits, else 10. */
int amo_format_message(const amo_packet *p, int first, char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
