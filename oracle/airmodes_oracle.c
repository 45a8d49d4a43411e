/*
 * airmodes_oracle.c -- CPU oracle (restatement) of the gr-air-modes rx_path hot path.
 *
 * TEST INFRASTRUCTURE ONLY -- see airmodes_oracle.h.  Never linked into, loaded
 * by, or used as a fallback for the product path.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).  Every
 * float expression below relies on IEEE-754 binary32/binary64 evaluation with no
 * contraction, exactly one rounding per written operation.
 *
 * Canonical semantics ("one infinitely long work() call over the whole stream"):
 * SURVEY.md Appendix D; the only free choice is the summation order inside the
 * two moving averages, fixed here as the CHIP-ALIGNED TWO-LEVEL ORDER described
 * at canonical_window_sums() and in DESIGN.md section 3.
 */
#include "airmodes_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHIPS_PER_AVG 48   /* rx_path.py:54  moving_average_ff(48*spc, ...)   */
#define BURST_CHIPS   240  /* preamble_impl.cc:219 / slicer_impl.cc:59        */

/* ---------------------------------------------------------------- a5 ---- */
/* preamble_impl.cc:67: powf(10., threshold_db/20.) -- the quotient is formed
 * in double, both powf arguments are then narrowed to float. */
float amo_threshold_lin(float thr_db)
{
    double q = (double)thr_db / 20.0;
    return powf(10.0f, (float)q);
}

/* ---------------------------------------------------------------- a1 ---- */
/* rx_path.py:38 complex_to_mag_squared: re*re and im*im each rounded to
 * float, then one rounded add (VOLK generic form; no fused multiply-add). */
void amo_mag2(const float *iq, uint64_t n, float *m)
{
    for (uint64_t i = 0; i < n; i++) {
        float re = iq[2 * i], im = iq[2 * i + 1];
        float rr = re * re;
        float ii = im * im;
        m[i] = rr + ii;
    }
}

/* ------------------------------------------------------------ a3 / a4 ---- */
/*
 * Canonical windowed sums (the documented order; GNU Radio's own running sum
 * depends on its scheduler, so one order has to be chosen -- DESIGN.md 3).
 *
 * All blocks are aligned to the absolute stream index (sample 0 = first
 * sample of the stream); samples before the stream start are zero.
 *
 * Level 1 -- within one chip (spc consecutive samples, chip q = n / spc):
 *     pre[n] = x[q*spc] + ... + x[n]              summed left  -> right
 *     suf[n] = x[n] + ... + x[q*spc + spc - 1]    summed right -> left
 *     tot[q] = pre[q*spc + spc - 1]
 * Pulse-matched filter, window W = spc (one chip):
 *     SUM_spc(x)[n] = pre[n]                       if n is the last sample of its chip
 *                   = suf[n - spc + 1] + pre[n]    otherwise   (suf of the previous chip)
 * Level 2 -- within one 48-chip block (block B = q / 48, j = q % 48):
 *     PT[q] = tot[48B] + ... + tot[q-1]            left  -> right, 0 when j == 0
 *     ST[q] = tot[q+1] + ... + tot[48B+47]         right -> left, 0 when j == 47
 *     PRE[n] = PT[q] + pre[n]        SUF[n] = suf[n] + ST[q]
 * Reference level, window L = 48*spc:
 *     SUM_L(x)[n] = PRE[n]                         if n is the last sample of its block
 *                 = SUF[n - L + 1] + PRE[n]        otherwise
 * Every '+' above is one binary32 rounding; adding the literal 0 is exact.
 */
static void chip_prefix_suffix(const float *x, uint64_t n, int spc,
                               float *pre, float *suf)
{
    for (uint64_t c0 = 0; c0 < n; c0 += (uint64_t)spc) {
        uint64_t c1 = c0 + (uint64_t)spc;
        if (c1 > n) c1 = n;             /* ragged last chip: missing samples = 0 */
        float acc = 0.0f;
        for (uint64_t i = c0; i < c1; i++) { acc = acc + x[i]; pre[i] = acc; }
        acc = 0.0f;
        for (uint64_t i = c1; i-- > c0;) { acc = acc + x[i]; suf[i] = acc; }
    }
}

static int moving_sum_chip(const float *x, uint64_t n, int spc, float scale,
                           float *out, float *pre, float *suf)
{
    chip_prefix_suffix(x, n, spc, pre, suf);
    for (uint64_t i = 0; i < n; i++) {
        float s;
        if ((i + 1) % (uint64_t)spc == 0 || i + 1 < (uint64_t)spc)
            s = pre[i];
        else
            s = suf[i - (uint64_t)spc + 1] + pre[i];
        out[i] = s * scale;
    }
    return 0;
}

/* window = `chips` chips, blocks of `chips` chips aligned to sample 0; out[i] = sum * scale when
 * divisor == 0, sum / divisor otherwise */
static int moving_sum_blockc(const float *x, uint64_t n, int spc, int chips, float scale, float divisor,
                             float *out, float *pre, float *suf)
{
    const uint64_t CH = (uint64_t)chips;
    const uint64_t L = CH * (uint64_t)spc;
    const uint64_t nchips = (n + (uint64_t)spc - 1) / (uint64_t)spc;
    const uint64_t nchips_pad = (nchips + CH - 1) / CH * CH;
    float *tot = (float *)calloc(nchips_pad ? nchips_pad : 1, sizeof(float));
    float *PT = (float *)calloc(nchips_pad ? nchips_pad : 1, sizeof(float));
    float *ST = (float *)calloc(nchips_pad ? nchips_pad : 1, sizeof(float));
    if (!tot || !PT || !ST) { free(tot); free(PT); free(ST); return -1; }

    chip_prefix_suffix(x, n, spc, pre, suf);
    for (uint64_t q = 0; q < nchips; q++) {
        uint64_t last = q * (uint64_t)spc + (uint64_t)spc - 1;
        if (last >= n) last = n - 1;
        tot[q] = pre[last];
    }
    for (uint64_t b = 0; b < nchips_pad; b += CH) {
        float acc = 0.0f;
        for (int j = 0; j < chips; j++) { PT[b + j] = acc; acc = acc + tot[b + j]; }
        acc = 0.0f;
        for (int j = chips - 1; j >= 0; j--) { ST[b + j] = acc; acc = acc + tot[b + j]; }
    }
    for (uint64_t i = 0; i < n; i++) {
        uint64_t q = i / (uint64_t)spc;
        float PRE = PT[q] + pre[i];
        float s;
        if ((i + 1) % L == 0 || i + 1 < L) {
            s = PRE;
        } else {
            uint64_t a = i - L + 1;
            float SUF = suf[a] + ST[a / (uint64_t)spc];
            s = SUF + PRE;
        }
        out[i] = (divisor != 0.0f) ? (s / divisor) : (s * scale);
    }
    free(tot); free(PT); free(ST);
    return 0;
}

static int moving_sum_block(const float *x, uint64_t n, int spc, float scale,
                            float *out, float *pre, float *suf)
{
    return moving_sum_blockc(x, n, spc, CHIPS_PER_AVG, scale, 0.0f, out, pre, suf);
}

/* a2: filter.dc_blocker_cc(100*spc, False) -- python/rx_path.py:39-41 (GNU Radio 3.8 gr-filter,
 * source not under /root/reference: PARITY UNPINNED).  Published algorithm (R. Lyons, "DC blocker
 * algorithms", the linear-phase form GNU Radio documents for long_form = False): two cascaded
 * D-sample moving averages and a (D-1)-sample delay,
 *     m1 = MA_D(x),  m2 = MA_D(m1),  y[n] = x[n - (D-1)] - m2[n],      x, m1 = 0 before the stream,
 * each moving average being sum / (float)D as in GNU Radio's moving_averager_c.  GNU Radio forms
 * the window sums with a recursive running sum whose rounding depends on the whole history; as
 * for a3/a4 the order is fixed here instead: the chip-aligned two-level order above with blocks
 * of 100 chips, I and Q treated as two real streams.  Output is interleaved like the input. */
#define CHIPS_PER_DCBLOCK 100
int amo_dcblock(const float *iq, uint64_t n, int spc, float *out)
{
    if (spc < 1) return -1;
    if (n == 0) return 0;
    const uint64_t D = (uint64_t)CHIPS_PER_DCBLOCK * (uint64_t)spc;
    float *x = (float *)malloc(n * sizeof(float));
    float *m1 = (float *)malloc(n * sizeof(float));
    float *m2 = (float *)malloc(n * sizeof(float));
    float *pre = (float *)malloc(n * sizeof(float));
    float *suf = (float *)malloc(n * sizeof(float));
    int rc = (x && m1 && m2 && pre && suf) ? 0 : -1;
    for (int c = 0; c < 2 && rc == 0; c++) {
        for (uint64_t i = 0; i < n; i++) x[i] = iq[2 * i + c];
        rc = moving_sum_blockc(x, n, spc, CHIPS_PER_DCBLOCK, 0.0f, (float)D, m1, pre, suf);
        if (rc == 0) rc = moving_sum_blockc(m1, n, spc, CHIPS_PER_DCBLOCK, 0.0f, (float)D, m2, pre, suf);
        for (uint64_t i = 0; i < n && rc == 0; i++) {
            const float d = (i >= D - 1) ? x[i - (D - 1)] : 0.0f;
            out[2 * i + c] = d - m2[i];
        }
    }
    free(x); free(m1); free(m2); free(pre); free(suf);
    return rc;
}

int amo_frontend(const float *iq, uint64_t n, int spc, int use_pmf,
                 float *bb, float *avg)
{
    if (spc < 1) return -1;
    if (n == 0) return 0;
    float *m = (float *)malloc(n * sizeof(float));
    float *pre = (float *)malloc(n * sizeof(float));
    float *suf = (float *)malloc(n * sizeof(float));
    if (!m || !pre || !suf) { free(m); free(pre); free(suf); return -1; }
    amo_mag2(iq, n, m);
    if (use_pmf) {
        /* rx_path.py:49 moving_average_ff(spc, 1.0/spc): the python double is
         * narrowed to the block's float scale. */
        float s1 = (float)(1.0 / (double)spc);
        moving_sum_chip(m, n, spc, s1, bb, pre, suf);
    } else {
        memcpy(bb, m, n * sizeof(float));
    }
    /* rx_path.py:54 moving_average_ff(48*spc, 1.0/(48*spc)) */
    float sL = (float)(1.0 / (double)(CHIPS_PER_AVG * spc));
    int rc = moving_sum_block(bb, n, spc, sL, avg, pre, suf);
    free(m); free(pre); free(suf);
    return rc;
}

/* GNU-Radio-like running sum (sum += new; out = sum*scale; sum -= old), re-seeded at the outputs
 * first, first + chunk, first + 2 chunk, ... (and at 0): the scheduler-dependent part of GNU Radio 3.8's
 * moving_average_ff (it re-seeds at the start of every work() call and every <= 4096 outputs inside one).
 * Sensitivity study only (tools/frontend_sensitivity.py). */
static void running_average(const float *x, uint64_t n, uint64_t len, float scale,
                            uint32_t chunk, uint32_t first, float *out)
{
    uint64_t o0 = 0;
    uint64_t o1 = (first > 0 && first < chunk) ? first : chunk;
    while (o0 < n) {
        if (o1 > n) o1 = n;
        float sum = 0.0f;
        for (uint64_t i = 0; i + 1 < len; i++) {
            int64_t idx = (int64_t)o0 - (int64_t)(len - 1) + (int64_t)i;
            sum = sum + (idx >= 0 ? x[idx] : 0.0f);
        }
        for (uint64_t o = o0; o < o1; o++) {
            sum = sum + x[o];
            out[o] = sum * scale;
            int64_t old = (int64_t)o - (int64_t)(len - 1);
            sum = sum - (old >= 0 ? x[old] : 0.0f);
        }
        o0 = o1;
        o1 = o0 + chunk;
    }
}

int amo_frontend_running2(const float *iq, uint64_t n, int spc, int use_pmf,
                          uint32_t chunk, uint32_t first, float *bb, float *avg)
{
    if (spc < 1 || chunk == 0) return -1;
    if (n == 0) return 0;
    float *m = (float *)malloc(n * sizeof(float));
    if (!m) return -1;
    amo_mag2(iq, n, m);
    if (use_pmf)
        running_average(m, n, (uint64_t)spc, (float)(1.0 / (double)spc), chunk, first, bb);
    else
        memcpy(bb, m, n * sizeof(float));
    running_average(bb, n, (uint64_t)CHIPS_PER_AVG * spc,
                    (float)(1.0 / (double)(CHIPS_PER_AVG * spc)), chunk, first, avg);
    free(m);
    return 0;
}

int amo_frontend_running(const float *iq, uint64_t n, int spc, int use_pmf,
                         uint32_t chunk, float *bb, float *avg)
{
    return amo_frontend_running2(iq, n, spc, use_pmf, chunk, 0, bb, avg);
}

/* ------------------------------------------------------------ a6 - a9 ---- */
/* The preamble block's geometry (preamble_impl.cc:56-63,150,158-162,185,205-208,212,220,237), with the reference's own
 * types: d_samples_per_chip is a FLOAT (channel_rate / d_chip_rate) and every use goes through a float product and an
 * int() truncation -- so a rate that is not a multiple of 2 MHz (5 Msps: 2.5 samples per chip) is a legal rate with its
 * own, slightly crooked, geometry (e.g. the correlation windows sit at multiples of int(2.5) = 2 samples while the pulse
 * offsets are int(2 * 2.5) = 5, int(7 * 2.5) = 17, int(9 * 2.5) = 22).  For whole samples per chip everything below
 * reduces to the familiar multiples of spc. */
typedef struct {
    int S;              /* int(d_samples_per_chip): granularity of ninputs (:150), correlation window (:90-98)     */
    int hist0;          /* history items in front of the stream: set_history(d_samples_per_symbol) - 1 (:62)      */
    int o1, o2, o3;     /* pulse offsets int(2 spc), int(7 spc), int(9 spc) (:158-162)                             */
    float spcf;         /* d_samples_per_chip: the late-peak loop runs while how_late < spcf (:192)                 */
    int za0, za1;       /* quiet zone 1: j = int(1.5 sps) .. j <= 3 sps (:205)                                      */
    int zb0, zb1;       /* quiet zone 2: j = int(5 sps) .. j <= 7.5 sps (:207)                                      */
    float Bf;           /* 240 * d_samples_per_chip: the room a burst needs (:212)                                  */
    int B;              /* ... and what consume_each() skips after a hit, as an int (:237)                          */
    int idx[BURST_CHIPS];   /* int(j * d_samples_per_chip): the sample of soft chip j (:220)                        */
} amo_geom;

static void geom_of(uint64_t rate_i, amo_geom *g)
{
    const float channel_rate = (float)(int)rate_i;               /* preamble::make(float channel_rate, ...)        */
    const float spcf = channel_rate / (float)2000000;            /* :57  float / int d_chip_rate                   */
    const float sps = spcf * 2;                                  /* :58                                             */
    g->spcf = spcf;
    g->S = (int)spcf;
    g->hist0 = (int)sps - 1;
    g->o1 = (int)(2 * spcf); g->o2 = (int)(7 * spcf); g->o3 = (int)(9 * spcf);
    g->za0 = (int)(1.5 * sps);                                   /* double product, truncated                      */
    g->za1 = (int)floorf(3 * sps);                               /* largest j with (float)j <= 3 * sps              */
    g->zb0 = (int)(5 * sps);                                     /* float product, truncated                       */
    g->zb1 = (int)floor(7.5 * sps);                              /* largest j with (double)j <= 7.5 * sps           */
    g->Bf = 240 * spcf;
    /* :237 consume_each(i + 240 * d_samples_per_chip): int + float, rounded to float, truncated -- i counts from the start
     * of the current general_work() window.  Where 240 * spcf is a whole number (every multiple of 2 MHz; 5, 6.25, 4.8, 3,
     * 13 Msps ...) the skip is that number for every i.  Where it is not (2.1 Msps: 251.99998) the float sum rounds to
     * i + 252 for every i >= 4 and to i + 251 below: the skip is taken at a representative window offset (i = 1024), i.e.
     * what the reference does for all but the first few items of a window (canonical choice). */
    g->B = (int)((float)1024 + g->Bf) - 1024;
    for (int j = 0; j < BURST_CHIPS; j++) g->idx[j] = (int)(j * spcf);
}

/* preamble_impl.cc:90-98: energy in the four preamble chips {0,2,7,9},
 * accumulated in double, chip-major then sample-major; samples_per_chip arrives as an int (:185-186). */
static double preamble_energy(const float *p, int spc)
{
    static const int pulse_chip[4] = {0, 2, 7, 9};
    double e = 0.0;
    for (int c = 0; c < 4; c++)
        for (int j = 0; j < spc; j++)
            e += p[pulse_chip[c] * spc + j];
    return e;
}

uint64_t amo_preamble_scan(const float *bb, const float *avg, uint64_t n,
                           int spc, float thr_db, uint64_t rate,
                           float *bursts, amo_tag *tags, uint64_t cap)
{
    if (spc < 1 || n == 0) return 0;
    amo_geom G;
    geom_of(rate, &G);
    if (G.S != spc) return 0;                 /* (the front end in front of this block runs at int(rate / 2e6): rx_path.py:35) */
    const uint64_t S = (uint64_t)G.S;
    const uint64_t hist = (uint64_t)G.hist0;  /* set_history(d_samples_per_symbol): preamble_impl.cc:62 */
    const uint64_t K = n + hist;              /* items the single work() call sees       */
    const uint64_t pad = 260 * (S + 1);       /* zeros beyond the end of the stream      */
    float *in = (float *)calloc(K + pad, sizeof(float));
    float *inavg = (float *)calloc(K + pad, sizeof(float));
    if (!in || !inavg) { free(in); free(inavg); return 0; }
    memcpy(in + hist, bb, n * sizeof(float));
    memcpy(inavg + hist, avg, n * sizeof(float));

    const float T = amo_threshold_lin(thr_db);
    /* preamble_impl.cc:150 */
    uint64_t ninputs = (K - K % S > S) ? K - K % S - S : 0;
    const uint64_t o1 = (uint64_t)G.o1, o2 = (uint64_t)G.o2, o3 = (uint64_t)G.o3;   /* :158-162 */
    uint64_t hits = 0;

    uint64_t k = 0;
    while (k < ninputs) {
        float thr = inavg[k] * T;                                  /* :173 */
        if (!(in[k] > thr)) { k++; continue; }                     /* :174 */
        if (in[k + 1] > in[k]) { k++; continue; }                  /* :175 */
        if (in[k + o1] < thr) { k++; continue; }                   /* :177 */
        if (in[k + o2] < thr) { k++; continue; }                   /* :178 */
        if (in[k + o3] < thr) { k++; continue; }                   /* :179 */

        /* :184-192 slide right while the 4-pulse energy still grows, while how_late < d_samples_per_chip */
        uint32_t how_late = 0;
        for (;;) {
            double now = preamble_energy(in + k, G.S);
            double nxt = preamble_energy(in + k + 1, G.S);
            int late = nxt > now;
            if (late) { k++; how_late++; }
            if (!(late && (float)how_late < G.spcf)) break;
        }

        /* :198-203 */
        float peaksum = in[k] + in[k + o1];
        peaksum = peaksum + in[k + o2];
        peaksum = peaksum + in[k + o3];
        float avgpeak = (float)((double)peaksum / 4.0);
        float space_thr = inavg[k] + (avgpeak - inavg[k]) / T;
        int valid = 1;
        for (int j = G.za0; j <= G.za1; j++)                       /* :205-206 */
            if (in[k + (uint64_t)j] > space_thr) valid = 0;
        for (int j = G.zb0; j <= G.zb1; j++)                       /* :207-208 */
            if (in[k + (uint64_t)j] > space_thr) valid = 0;
        if (!valid) { k++; continue; }                             /* :209 */

        /* :212 end of stream (the reference's `ninputs - i` is a signed int: a late-peak shift past ninputs counts as
         * no room too); the comparison is the reference's: int against the float 240 * d_samples_per_chip */
        if (k >= ninputs || (float)(ninputs - k) < G.Bf) break;

        if (hits < cap) {
            float *o = bursts + hits * BURST_CHIPS;
            for (int j = 0; j < BURST_CHIPS; j++)                  /* :219-221 */
                o[j] = in[k + (uint64_t)G.idx[j]] - inavg[k];
            amo_tag *t = &tags[hits];
            t->sample = k;                                         /* :224, history offset included */
            t->secs = k / rate;                                    /* :124 */
            t->frac = (double)(k % rate) / (double)rate;           /* :125 */
            if (t->frac > 1.0f) { t->frac -= 1.0f; t->secs += 1; } /* :129-132 */
            t->inavg = inavg[k];
            t->how_late = how_late;
        }
        hits++;
        k += (uint64_t)G.B;                                        /* :237 */
    }
    free(in); free(inavg);
    return hits;
}

/* Every position that passes the first-stage test (preamble_impl.cc:172-179), refined on its own: what the
 * late-peak search (:182-192) and the quiet-zone test (:198-209) make of it if the scan gets there -- independent
 * of the greedy order in which the reference visits candidates (stage-level parity of the GPU's candidate
 * records; the records of the candidates amo_preamble_scan does visit must agree with its tags).
 * Coordinates are item counts k of the preamble block (stream index + history, like amo_tag.sample); only
 * positions k < k_limit are reported.  Returns the number found (may exceed cap: then only cap are stored). */
uint64_t amo_candidates(const float *bb, const float *avg, uint64_t n, int spc, float thr_db, uint64_t k_limit,
                        uint64_t *pos, uint64_t *refined, uint8_t *valid, float *inavg_out, uint64_t cap)
{
    return amo_candidates_r(bb, avg, n, (uint64_t)spc * 2000000u, thr_db, k_limit, pos, refined, valid, inavg_out, cap);
}

uint64_t amo_candidates_r(const float *bb, const float *avg, uint64_t n, uint64_t rate, float thr_db, uint64_t k_limit,
                          uint64_t *pos, uint64_t *refined, uint8_t *valid, float *inavg_out, uint64_t cap)
{
    amo_geom G;
    geom_of(rate, &G);
    if (G.S < 1 || n == 0) return 0;
    const uint64_t S = (uint64_t)G.S;
    const uint64_t hist = (uint64_t)G.hist0;
    const uint64_t K = n + hist;
    const uint64_t pad = 260 * (S + 1);
    float *in = (float *)calloc(K + pad, sizeof(float));
    float *inavg = (float *)calloc(K + pad, sizeof(float));
    if (!in || !inavg) { free(in); free(inavg); return 0; }
    memcpy(in + hist, bb, n * sizeof(float));
    memcpy(inavg + hist, avg, n * sizeof(float));
    const float T = amo_threshold_lin(thr_db);
    uint64_t ninputs = (K - K % S > S) ? K - K % S - S : 0;
    if (ninputs > k_limit) ninputs = k_limit;
    const uint64_t o1 = (uint64_t)G.o1, o2 = (uint64_t)G.o2, o3 = (uint64_t)G.o3;
    uint64_t found = 0;
    for (uint64_t k0 = 0; k0 < ninputs; k0++) {
        float thr = inavg[k0] * T;
        if (!(in[k0] > thr)) continue;
        if (in[k0 + 1] > in[k0]) continue;
        if (in[k0 + o1] < thr) continue;
        if (in[k0 + o2] < thr) continue;
        if (in[k0 + o3] < thr) continue;
        uint64_t k = k0;
        uint32_t how_late = 0;
        for (;;) {
            double now = preamble_energy(in + k, G.S);
            double nxt = preamble_energy(in + k + 1, G.S);
            int late = nxt > now;
            if (late) { k++; how_late++; }
            if (!(late && (float)how_late < G.spcf)) break;
        }
        float peaksum = in[k] + in[k + o1];
        peaksum = peaksum + in[k + o2];
        peaksum = peaksum + in[k + o3];
        float avgpeak = (float)((double)peaksum / 4.0);
        float space_thr = inavg[k] + (avgpeak - inavg[k]) / T;
        int ok = 1;
        for (int j = G.za0; j <= G.za1; j++)
            if (in[k + (uint64_t)j] > space_thr) ok = 0;
        for (int j = G.zb0; j <= G.zb1; j++)
            if (in[k + (uint64_t)j] > space_thr) ok = 0;
        if (found < cap) {
            pos[found] = k0;
            refined[found] = k;
            valid[found] = (uint8_t)ok;
            inavg_out[found] = inavg[k];
        }
        found++;
    }
    free(in); free(inavg);
    return found;
}

/* ---------------------------------------------------------------- a12 ---- */
/* CRC-24, generator 0xFFF409 (x^24 + ... ), zero initial value, MSB first.
 * Stated bit-serially: the register holds msg(x)*x^24 mod G(x). */
uint32_t amo_crc24(const uint8_t *data, int nbytes)
{
    uint32_t reg = 0;
    for (int i = 0; i < nbytes; i++) {
        for (int b = 7; b >= 0; b--) {
            uint32_t inbit = (data[i] >> b) & 1u;
            uint32_t top = (reg >> 23) & 1u;
            reg = (reg << 1) & 0xFFFFFFu;
            if (top ^ inbit) reg ^= 0xFFF409u;
        }
    }
    return reg;
}

/* ---------------------------------------------------------- a10 / a11 ---- */
/* slicer_impl.cc:67-100.  Returns decision in bit 0, confidence in bit 1. */
static int chip_pair_slice(float c0, float c1, float ref)
{
    float hi = (float)((double)ref * 1.414);     /* :71 */
    float lo = (float)((double)ref * 0.707);     /* :72 */
    int in0 = (c0 > lo) && (c0 < hi);
    int in1 = (c1 > lo) && (c1 < hi);
    int decision, conf;
    if (in0 && !in1) { decision = 1; conf = 1; }
    else if (in1 && !in0) { decision = 0; conf = 1; }
    else if (in0 && in1) { decision = c0 > c1; conf = 0; }
    else {
        decision = c0 > c1;
        double half_lo = (double)lo * 0.5;       /* :92,:95 */
        float loser = decision ? c1 : c0;
        conf = ((double)loser < half_lo) ? 1 : 0;
    }
    return decision | (conf << 1);
}

int amo_slice(const float *b, const amo_tag *tag, amo_packet *out)
{
    amo_packet p;
    memset(&p, 0, sizeof(p));
    float s = b[0] + b[2];                       /* :128-131 */
    s = s + b[7];
    s = s + b[9];
    p.ref = (float)((double)s / 4.0);

    const float *d = b + 16;                     /* :133 */
    unsigned hdr = 0;
    for (int j = 0; j < 5; j++)                  /* :136-139 */
        if (chip_pair_slice(d[2 * j], d[2 * j + 1], p.ref) & 1) hdr |= 1u << (4 - j);
    int nbits = (hdr == 16 || hdr == 17 || hdr == 20 || hdr == 21) ? 112 : 56;  /* :140-142 */

    unsigned nlow = 0;
    for (int j = 0; j < nbits; j++) {            /* :146-159 */
        int r = chip_pair_slice(d[2 * j], d[2 * j + 1], p.ref);
        if (r & 1) p.data[j / 8] |= (uint8_t)(1u << (7 - (j % 8)));
        if (!(r & 2) && nlow < 24) nlow++;
    }
    int allzero = 1;
    for (int m = 0; m < 14; m++) if (p.data[m]) allzero = 0;
    if (allzero) return 0;                       /* :162-166 */

    unsigned mt = (p.data[0] >> 3) & 0x1F;       /* :168 */
    if (nbits == 56 && mt != 11 && nlow > 0) return 0;   /* :170 */
    if (mt == 11 && nlow >= 10) return 0;                /* :171 */

    int nbytes = nbits / 8;
    uint32_t syn = amo_crc24(p.data, nbytes - 3);        /* :173-177 */
    syn ^= ((uint32_t)p.data[nbytes - 3] << 16) | ((uint32_t)p.data[nbytes - 2] << 8) |
           (uint32_t)p.data[nbytes - 1];
    if (syn && (mt == 11 || mt == 17)) return 0;         /* :182 */

    p.nbytes = (uint8_t)nbytes;
    p.df = (uint8_t)mt;
    p.numlowconf = (uint8_t)nlow;
    p.crc = syn;
    p.sample = tag->sample;
    p.secs = tag->secs;
    p.frac = tag->frac;
    *out = p;
    return 1;
}

/* ------------------------------------------------------------ whole path ---- */
uint64_t amo_demod2(const float *iq, uint64_t n, double rate, float thr_db, int use_pmf, int use_dcblock,
                    amo_packet *out, uint64_t cap, uint64_t *n_tags)
{
    if (!use_dcblock) return amo_demod(iq, n, rate, thr_db, use_pmf, out, cap, n_tags);
    int spc = (int)(rate / 2e6);
    if (n_tags) *n_tags = 0;
    if (spc < 1 || n == 0) return 0;
    float *y = (float *)malloc(2 * n * sizeof(float));
    uint64_t npk = 0;
    if (y && amo_dcblock(iq, n, spc, y) == 0) npk = amo_demod(y, n, rate, thr_db, use_pmf, out, cap, n_tags);
    free(y);
    return npk;
}

uint64_t amo_demod(const float *iq, uint64_t n, double rate, float thr_db,
                   int use_pmf, amo_packet *out, uint64_t cap, uint64_t *n_tags)
{
    int spc = (int)(rate / 2e6);                 /* rx_path.py:35 */
    if (n_tags) *n_tags = 0;
    if (spc < 1 || n == 0) return 0;
    uint64_t rate_i = (uint64_t)(int)(float)rate;    /* preamble_impl.cc:60 (int d_sample_rate) */
    float *bb = (float *)malloc(n * sizeof(float));
    float *avg = (float *)malloc(n * sizeof(float));
    uint64_t tcap = n / (BURST_CHIPS * (uint64_t)spc) + 2;       /* (a burst spans >= 240 * int(samples per chip) items) */
    float *bursts = (float *)malloc(tcap * BURST_CHIPS * sizeof(float));
    amo_tag *tags = (amo_tag *)malloc(tcap * sizeof(amo_tag));
    uint64_t npk = 0;
    if (bb && avg && bursts && tags && amo_frontend(iq, n, spc, use_pmf, bb, avg) == 0) {
        uint64_t hits = amo_preamble_scan(bb, avg, n, spc, thr_db, rate_i, bursts, tags, tcap);
        if (n_tags) *n_tags = hits;
        for (uint64_t h = 0; h < hits && h < tcap; h++) {
            amo_packet p;
            if (amo_slice(bursts + h * BURST_CHIPS, &tags[h], &p)) {
                if (npk < cap) out[npk] = p;
                npk++;
            }
        }
    }
    free(bb); free(avg); free(bursts); free(tags);
    return npk;
}

/* rx_time stream tags (preamble_impl.cc:165-170 picks the latest tag, tag_to_timestamp :100-137
 * turns it into a time stamp).  Canonical rule: the tag in force for an item
 * count k is the last one whose offset is <= k; with no such tag the reference's default applies
 * (offset 0, time 0).  Which call's window first *sees* a tag depends on GNU Radio's scheduler in the
 * reference (a tag up to one buffer ahead of the preamble may already be in force there, and the
 * unsigned difference then wraps); this rule is the limit of small windows and is what the reference
 * computes whenever no preamble lies in the same window in front of a tag. */
void amo_timestamp(uint64_t k, uint64_t rate, const amo_time_tag *tt, uint64_t ntt, uint64_t *secs,
                   double *frac)
{
    uint64_t off = 0, whole = 0;
    double fr = 0.0;
    for (uint64_t i = 0; i < ntt; i++)
        if (tt[i].offset <= k) { off = tt[i].offset; whole = tt[i].secs; fr = tt[i].frac; }
    const uint64_t d = k - off;
    uint64_t s = whole + d / rate;                                  /* :124,127 */
    double f = fr + (double)(d % rate) / (double)rate;              /* :125,128 */
    if (f > 1.0f) { f -= 1.0f; s += 1; }                            /* :129-132 */
    *secs = s;
    *frac = f;
}

void amo_restamp_packets(amo_packet *p, uint64_t n, uint64_t rate, const amo_time_tag *tt, uint64_t ntt)
{
    for (uint64_t i = 0; i < n; i++) amo_timestamp(p[i].sample, rate, tt, ntt, &p[i].secs, &p[i].frac);
}

void amo_restamp_tags(amo_tag *t, uint64_t n, uint64_t rate, const amo_time_tag *tt, uint64_t ntt)
{
    for (uint64_t i = 0; i < n; i++) amo_timestamp(t[i].sample, rate, tt, ntt, &t[i].secs, &t[i].frac);
}

/* slicer_impl.cc:186-192.  The reference formats through a member
 * ostringstream whose precision is raised to 10 while printing the first
 * message's fractional timestamp and never lowered again, so the reference
 * level is printed %.6g in the first message and %.10g afterwards. */
int amo_format_message(const amo_packet *p, int first, char *buf, size_t cap)
{
    char tmp[160];
    int w = 0;
    for (int m = 0; m < p->nbytes; m++)
        w += snprintf(tmp + w, sizeof(tmp) - (size_t)w, "%02x", (unsigned)p->data[m]);
    w += snprintf(tmp + w, sizeof(tmp) - (size_t)w, " %06x %.*g %llu %.10g",
                  (unsigned)p->crc, first ? 6 : 10, (double)p->ref,
                  (unsigned long long)p->secs, p->frac);
    if ((size_t)w + 1 > cap) return -1;
    memcpy(buf, tmp, (size_t)w + 1);
    return w;
}
