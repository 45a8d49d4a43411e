// am_refine_seg.hip -- candidate list + bb rows + refinement in ONE launch at 32 samples per chip (64 Msps; round 6).
//
// Until round 5 three launches stood between the streaming front end and the greedy chain: am_k_gather_wg<1> listed the candidates of
// the bitmap and formed, from the samples, the bb rows around them (42 MB written per 64 M samples at the bench density), and
// am_k_refine_late read those rows back (50 MB) for the late-peak search (lib/preamble_impl.cc:90-98,182-192) and the quiet zones
// (:198-209).  Here a workgroup does all of it for its share of a front-end segment with the rows in LDS: nothing is written that is
// only read back, one launch and its gap are gone, and the refinement's chains of memory round trips (positions, eight bb samples
// per position, six row maxima, four partial rows, a galloping search for the successor) become LDS reads and popcounts.
//
//   * a front-end segment (am_k_fe3: 1 344 bitmap words at 64 M samples) is cut into `parts` equal shares, one workgroup each; a
//     workgroup adds up the counts of the segments before its own (wg_cnt) and the popcounts of the shares before its own -- its
//     place in the flat candidate list -- in the same memory round trip that brings its own words;
//   * the share's words (+ the 256 behind them: a hit's resume position lies at most 241 words on) stay in LDS with the exclusive
//     prefix of their popcounts: the flat index of a candidate, and the chain's successor -- the number of candidates below the
//     resume position -- are a table read and a popcount;
//   * the share is worked off in GROUPS of consecutive candidate words: as many as give at most RS_NT candidates and RS_R wanted
//     chips (a candidate in word w reads the chips w .. w + 16: bit b of word w is position 32 w + b - 288, array chip w - 9).  A
//     group's wanted chips are formed from IQ in the canonical order of DESIGN.md 3 -- 32 per wave, all of a batch's loads in flight
//     together, exactly as am_rows_segment32 did -- into rows that are CONSECUTIVE in LDS for consecutive chips of a run, so a
//     candidate addresses everything it reads from the row of its own chip.  The NEXT group's loads are issued before the current
//     group is refined;
//   * late-peak decisions once per position the group's candidates can reach, a quarter of a bitmap word per lane (exact difference
//     of the two sums; the reference's two sequential sums only for close calls), a word of decisions per bitmap word; then one lane
//     per candidate: shift (count of trailing ones), reference level (the one global load left), quiet zones from the row maxima and
//     four partial rows, the record, the successor.
// Every serial chain of LDS reads that a first form of this kernel had (prefix loops, bisections, a loop per late shift) is a wave
// ballot, a table or a bit count here: LDS latency (~100 cycles) times a few dozen steps, per group, was most of that form's time.
// Results are bit-identical to am_k_gather_wg<1> + am_k_refine_late (test builds keep those: AIRMODES_FUSED_REFINE=0) and to the
// oracle's candidate records (stage-level parity tests, every record).
#include "am_internal.h"
#include "am_fe_stream.h"

#include <stdio.h>

#include <algorithm>
#include <vector>

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

// keeps a loaded value where it is in the program (the compiler otherwise sinks loads to their first use, behind branches)
#if defined(__HIP_DEVICE_COMPILE__)
#define AM_PIN_RS(x) asm volatile("" : "+v"(x))
#define RS_FENCE() asm volatile("" ::: "memory")
// the thread index anew (nothing that follows from it is to live across loop iterations: hoisted out of the group loop those values --
// row addresses, piece offsets of nine rounds of loads -- cost ~70 spilled registers and a scratch round trip at every use)
#define RS_LAUNDER(x) asm volatile("" : "+v"(x))
#else
#define AM_PIN_RS(x) ((void)0)
#define RS_FENCE() ((void)0)
#define RS_LAUNDER(x) ((void)0)
#endif

#define RS_NT 256                         /* threads = candidates per group */
#define RS_NWV (RS_NT / AM_WAVE)
#ifndef RS_R
#define RS_R 128                          /* bb rows per group: 32 per wave and batch */
#endif
#define RS_XE 4                           /* rows for the chip before a run's first, per wave (a batch of 32 holds at most 3 run starts) */
#define RS_XS 36                          /* floats per LDS row: 32 + 4 pad (16-byte reads of consecutive rows hit all banks) */
#define RS_BBW 17                         /* chips a candidate reads from its own chip on */
#define RS_WPT 4                          /* bitmap words per thread and window */
#define RS_NWORD (RS_NT * RS_WPT)         /* words in LDS: a window of the share + the words behind it */
#define RS_AHEAD 256                      /* ... of which behind the window (a resume position lies <= 241 words on) */
#define RS_WIN (RS_NWORD - RS_AHEAD)      /* words of the share per window */
#define RS_NFW (RS_NWORD / 64 + 1)        /* 64-bit words of per-word flags */
#define RS_GW 255                         /* bitmap words per group at most (a word of late decisions per word + one) */
#ifndef RS_PW
#define RS_PW 672                         /* bitmap words per workgroup aimed at (am_k_fe3 at 64 M samples: 1 344 per segment, two shares) */
#endif
#define RS_MAXPARTS 8
#ifndef RS_WPS
#define RS_WPS 4                          /* waves per SIMD the kernel is compiled for */
#endif
static_assert(RS_R % 32 == 0 && RS_R / 32 == RS_NWV, "one batch of 32 rows per wave and group");
static_assert(RS_AHEAD >= AM_BURST + 2 && RS_WIN > 0 && RS_NFW <= AM_WAVE, "a hit's resume position must lie inside the words in LDS");
static_assert(RS_PW <= RS_WIN, "a share of RS_PW words is one window");

struct am_rseg_lds {
    __attribute__((aligned(16))) uint32_t W[RS_NWORD];   // bitmap words of the window (+ RS_AHEAD behind it)
    __attribute__((aligned(16))) float XM[RS_R * RS_XS];               // the group's rows: |.|^2, then bb
    __attribute__((aligned(16))) float XE[RS_NWV * RS_XE * RS_XS];     // per wave: |.|^2 of the chip before a run's first
    unsigned long long NZ[RS_NFW];        // word holds a candidate (own words only)
    unsigned long long FLW[RS_NFW];       // chip is wanted: a candidate in one of the 16 words before it or in its own
    uint32_t PFW[RS_NFW + 1];             // exclusive prefix of popcount(FLW)
    __attribute__((aligned(16))) float QMAX[RS_R * 8];   // largest bb of every four samples of a row
    float RMAX[RS_R];                     // largest bb of a row
    uint32_t POSL[RS_NT];                 // the group's candidates: position,
    uint32_t LATEW[RS_GW + 2];            // per word of the group (+ one): bit b = E(q + 1) > E(q) at the word's position b
    uint32_t ws[RS_NWV], red[3][RS_NWV];
    uint16_t P[RS_NWORD + 4];             // exclusive prefix of the words' popcounts
    uint16_t CHL[RS_WIN + 32];            // the window's wanted chips in ascending order
    uint16_t CR0[RS_NT];                  // ... row of a candidate's own chip
    uint16_t RC[RS_NWV * (32 + RS_XE)];   // per wave: chip (+ 1) whose samples row slot s takes (slots from cnt on: the chip before a run's first)
};

struct am_rseg_args {
    const uint32_t *bits, *wg_cnt;
    const float *wg_max;
    uint32_t nwg, words_per_wg, nwords, Mcap, vspan, nv, end_j, parts, pw;
    uint32_t n_long, words_short, vspan_short;   // segments n_long .. : words_short words, vspan_short array coordinates each (levelled front end)
    const float *iq;
    long long src_abs0, src_abs1, out_abs0;
    const float *avg_sparse;
    float s1, thr_lin;
    uint32_t *pos, *e, *tgt, *jump0, *total_out;
    float *inavg;
    uint8_t *valid;
    long long *prof;                      // profiling builds (-DRS_PROFILE): [grid][12] wall-clock ticks (10 ns) per phase, else null
};

// Profiling builds only (tools/build_variants.sh, -DRS_PROFILE): where a workgroup's time goes, stamped by thread 0
#if defined(RS_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define RS_STAMP(k) do { if (tid == 0) { const long long now__ = (long long)wall_clock64(); rsp[k] += now__ - rsl; rsl = now__; } } while (0)
#else
#define RS_STAMP(k) do { } while (0)
#endif

// a value that is the same in every lane, kept in a scalar register (what comes out of LDS is a vector register to the compiler)
#if defined(__HIP_DEVICE_COMPILE__)
#define RS_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#else
#define RS_UNI(x) ((uint32_t)(x))
#endif
// the value lane + 32 holds, for the lanes below 32 (the upper half gets something it does not use): one v_permlane32_swap_b32 on
// gfx950 where a shuffle is an LDS round trip (ds_bpermute)
__device__ __forceinline__ float rs_from_upper_half(float v, int lane)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // (inline assembly with both registers read-write: __builtin_amdgcn_permlane32_swap lets the allocator hand it a source register
    // whose value is still needed afterwards -- measured, build/t in round 6: 224 of 256 sums wrong -- the instruction overwrites both
    // operands; the s_nop pairs are the wait states the hazard recogniser would insert around an instruction it can see)
    (void)lane;
    unsigned a_ = __builtin_bit_cast(unsigned, v), b_ = a_;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a_), "+v"(b_));
    return __builtin_bit_cast(float, b_);
#else
    return __shfl(v, lane ^ 32, AM_WAVE);
#endif
}
// number of wanted chips below chip x of the window
__device__ __forceinline__ uint32_t rs_rank(const am_rseg_lds &L, uint32_t x)
{
    return L.PFW[x >> 6] + (uint32_t)__popcll(L.FLW[x >> 6] & ((1ull << (x & 63u)) - 1ull));
}

// the reference's sequential double-precision sum over the four pulses from offset o (0 .. 32) of row `row` on (preamble_impl.cc:91-98)
__device__ __attribute__((noinline)) double rs_energy(const float *XM, uint32_t row, uint32_t o)
{
    double e = 0.0;
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        const uint32_t r = row + (c == 0 ? 0u : (c == 1 ? 2u : (c == 2 ? 7u : 9u)));
#pragma unroll 1
        for (uint32_t j = 0; j < 32u; ++j) {
            const uint32_t idx = o + j;
            e += (double)XM[(r + (idx >> 5)) * RS_XS + (idx & 31u)];
        }
    }
    return e;
}

// the largest of the samples i >= o (from_o) or i <= o of a row (fmaxf skips a NaN exactly as `NaN > x` is false): the quad that
// holds sample o from the row itself, the other quads from the row's quad maxima -- three 16-byte reads instead of eight
__device__ __forceinline__ float rs_row_partial_max(const float *row, const float *qmax, int o, bool from_o)
{
    const int qo = o >> 2, ro = o & 3;
    const float4 u = reinterpret_cast<const float4 *>(row)[qo];
    const float4 q0 = reinterpret_cast<const float4 *>(qmax)[0], q1 = reinterpret_cast<const float4 *>(qmax)[1];
    const float NEG = -__builtin_inff();
    float m;
    if (from_o) {
        m = u.w;                                                  // sample 4 qo + 3 >= o always
        m = fmaxf(m, ro <= 2 ? u.z : NEG);
        m = fmaxf(m, ro <= 1 ? u.y : NEG);
        m = fmaxf(m, ro <= 0 ? u.x : NEG);
    } else {
        m = u.x;                                                  // sample 4 qo <= o always
        m = fmaxf(m, ro >= 1 ? u.y : NEG);
        m = fmaxf(m, ro >= 2 ? u.z : NEG);
        m = fmaxf(m, ro >= 3 ? u.w : NEG);
    }
    const float qv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const bool in = from_o ? (k > qo) : (k < qo);
        m = fmaxf(m, in ? qv[k] : NEG);
    }
    return m;
}

// a group of consecutive candidate words [ga, gb) of the window (uniform)
struct rs_group {
    uint32_t ga, gb;
    uint32_t pa;                          // candidates of the window before word ga
    uint32_t r_base, nrows, nc;           // wanted chips before chip ga / in [ga, gb + 15] / candidates
    bool inside;                          // every sample the rows need is present, 16-byte aligned
};

// a wave's batch of a group's rows: what a lane keeps between issuing the loads and using them
constexpr int RS_MAXR = (32 + RS_XE) * 16 / AM_WAVE;      // rounds of 64 16-byte pieces: 9
struct rs_batch {
    int cnt, srank, nrow;
    bool live, start;
    uint32_t ci;
    float4 v[RS_MAXR];
};

template <bool PMF>
__global__ void __launch_bounds__(RS_NT, RS_WPS) am_k_refine_seg(am_rseg_args a)
{
    constexpr int SPC = 32;
    __shared__ am_rseg_lds L;
    int tid_ = threadIdx.x;
    const int tid0 = tid_;
    const int wv = __builtin_amdgcn_readfirstlane(tid0 / AM_WAVE);
    int tid = tid0, lane = tid0 & (AM_WAVE - 1);
    const uint32_t g = blockIdx.x / a.parts, part = blockIdx.x % a.parts;
    const uint32_t seg_b = g < a.n_long ? g * a.words_per_wg : a.n_long * a.words_per_wg + (g - a.n_long) * a.words_short;
    const uint32_t seg_w = g < a.n_long ? a.words_per_wg : a.words_short;
    const uint32_t seg_e = (seg_b + seg_w < a.nwords) ? seg_b + seg_w : a.nwords;
    const uint32_t pb = seg_b + part * a.pw;                   // this workgroup's words: [pb, pe)
    const uint32_t pe = (pb + a.pw < seg_e) ? pb + a.pw : seg_e;
    const bool writes_total = blockIdx.x == 0;
    if (pb >= seg_e && !writes_total) return;                  // (uniform) nothing here
#if defined(RS_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    long long rsp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long rsl = (long long)wall_clock64();
    rsp[10] = rsl;
#endif

    // the first window's words go out together with the counts of the other segments and the words of the shares before this one:
    // one memory round trip in front of the arithmetic (unconditional loads from clamped indices: a load behind a branch is waited
    // for at the join)
    uint32_t wl[RS_WPT];
#pragma unroll
    for (int k = 0; k < RS_WPT; ++k) {
        const uint32_t x = pb + (uint32_t)tid + (uint32_t)k * RS_NT;
        wl[k] = a.bits[x < a.nwords ? x : a.nwords - 1u];
    }
    // (wave 0 adds up the segments' counts, wave 1 the words of this segment's shares before the own one -- the same sums in every
    // wave would be the same instructions four times over, and this kernel is bound by instruction issue)
    uint32_t before = 0, all = 0, pre = 0;
    if (wv == 0) {
        for (uint32_t k0 = (uint32_t)lane; k0 < a.nwg; k0 += 8u * AM_WAVE) {
            uint32_t v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t k = k0 + (uint32_t)j * AM_WAVE;
                v[j] = a.wg_cnt[k < a.nwg ? k : 0u];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t k = k0 + (uint32_t)j * AM_WAVE;
                if (k < a.nwg) { all += v[j]; if (k < g) before += v[j]; }
            }
        }
        for (int o = AM_WAVE / 2; o >= 1; o >>= 1) {
            before += (uint32_t)__shfl_xor((int)before, o, AM_WAVE);
            all += (uint32_t)__shfl_xor((int)all, o, AM_WAVE);
        }
        if (lane == 0) { L.red[0][0] = before; L.red[1][0] = all; }
    } else if (wv == 1) {
        const uint32_t npre = pb < seg_e ? pb - seg_b : 0u;    // words of this segment in front of the share
        for (uint32_t k0 = (uint32_t)lane; k0 < npre; k0 += 8u * AM_WAVE) {
            uint32_t v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t k = k0 + (uint32_t)j * AM_WAVE;
                v[j] = a.bits[seg_b + (k < npre ? k : 0u)];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (k0 + (uint32_t)j * AM_WAVE < npre) pre += (uint32_t)__popc(v[j]);
        }
        for (int o = AM_WAVE / 2; o >= 1; o >>= 1) pre += (uint32_t)__shfl_xor((int)pre, o, AM_WAVE);
        if (lane == 0) L.red[2][0] = pre;
    }
    fes_barrier();
    uint32_t cbase = RS_UNI(L.red[0][0] + L.red[2][0]);        // candidates before the current window
    const uint32_t total = RS_UNI(L.red[1][0]);                // ... in the whole scan
    const uint32_t M = total < a.Mcap ? total : a.Mcap;
    if (writes_total && tid == 0) *a.total_out = total;
    if (pb >= seg_e) return;                                   // (uniform)
    RS_STAMP(0);                                               // counts + first words

    const bool wide = (reinterpret_cast<uintptr_t>(a.iq) & 15u) == 0 && (((a.out_abs0 - a.src_abs0) & 1) == 0);   // (uniform)
    const float2 *iq2 = reinterpret_cast<const float2 *>(a.iq);

    for (uint32_t wa = pb; wa < pe; wa += RS_WIN) {            // windows of the share (one, unless the segments are very long)
        RS_LAUNDER(tid_); tid = tid_; lane = tid & (AM_WAVE - 1);
        const uint32_t nown = (pe - wa < RS_WIN) ? pe - wa : RS_WIN;          // the window's own words; behind them: look-ahead
        const long long A0 = a.out_abs0 + ((long long)wa - 9) * SPC;          // absolute index of the first sample of window chip 0
        if (wa != pb) {
            fes_barrier();                                     // (the tables of the window before are still being read)
#pragma unroll
            for (int k = 0; k < RS_WPT; ++k) {
                const uint32_t x = wa + (uint32_t)tid + (uint32_t)k * RS_NT;
                wl[k] = a.bits[x < a.nwords ? x : a.nwords - 1u];
            }
        }
        // ---- the window's tables ------------------------------------------------------------------------------------------------
#pragma unroll
        for (int k = 0; k < RS_WPT; ++k) {
            const uint32_t i = (uint32_t)tid + (uint32_t)k * RS_NT;
            AM_PIN_RS(wl[k]);
            const uint32_t w = (wa + i < a.nwords) ? wl[k] : 0u;
            L.W[i] = w;
            const unsigned long long m = __ballot(w != 0u && i < nown);
            if (lane == 0) L.NZ[(uint32_t)wv + (uint32_t)k * RS_NWV] = m;
        }
        if (tid == 0) L.NZ[RS_NFW - 1] = 0ull;
        fes_barrier();
        uint32_t c4[RS_WPT], csum = 0, cincl;
        {
            // a thread takes RS_WPT consecutive words (one 16-byte read)
            const uint4 u = *reinterpret_cast<const uint4 *>(&L.W[(uint32_t)tid * RS_WPT]);
            static_assert(RS_WPT == 4, "four words per thread");
            c4[0] = (uint32_t)__popc(u.x); c4[1] = (uint32_t)__popc(u.y); c4[2] = (uint32_t)__popc(u.z); c4[3] = (uint32_t)__popc(u.w);
            csum = c4[0] + c4[1] + c4[2] + c4[3];
            cincl = csum;
            for (int d = 1; d < AM_WAVE; d <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)cincl, d, AM_WAVE);
                if (lane >= d) cincl += up;
            }
            if (lane == AM_WAVE - 1) L.ws[wv] = cincl;
            // wanted chips: a word with a candidate flags its chip and the 16 behind it (dilation by 16 bits across the 64-bit
            // words); their prefix counts by one wave scan
            if (wv == 0) {
                unsigned long long f = 0ull;
                if (lane < RS_NFW) {
                    const unsigned long long x = L.NZ[lane];
                    unsigned long long d = x | (x << 1);
                    d |= d << 2; d |= d << 4; d |= d << 8;            // shifts 0 .. 15
                    static_assert(RS_BBW == 17, "a candidate's chip and the 16 after it");
                    f = d | (x << 16);
                    const uint32_t hp = lane ? (uint32_t)(L.NZ[lane - 1] >> 48) : 0u;   // the previous word's last 16 chips reach into this one
                    if (hp) f |= (2ull << (31 - __clz((int)hp))) - 1ull;
                    L.FLW[lane] = f;
                }
                const uint32_t fc = (uint32_t)__popcll(f);
                uint32_t fi = fc;
                for (int d = 1; d < 32; d <<= 1) {
                    const uint32_t up = (uint32_t)__shfl_up((int)fi, d, AM_WAVE);
                    if (lane >= d) fi += up;
                }
                if (lane <= RS_NFW) L.PFW[lane] = fi - fc;            // (lanes >= RS_NFW hold no flags: lane RS_NFW's exclusive prefix is the total)
            }
        }
        fes_barrier();
        {
            uint32_t off = cincl - csum;
            for (int k = 0; k < wv; ++k) off += L.ws[k];
#pragma unroll
            for (int k = 0; k < RS_WPT; ++k) { L.P[(uint32_t)tid * RS_WPT + (uint32_t)k] = (uint16_t)off; off += c4[k]; }
            if (tid == RS_NT - 1) L.P[RS_NWORD] = (uint16_t)off;
            // the wanted chips in ascending order (a thread's four chips lie in one 64-bit flag word)
            const uint32_t x0 = (uint32_t)tid * RS_WPT;
            const unsigned long long f = L.FLW[x0 >> 6];
            uint32_t r = L.PFW[x0 >> 6] + (uint32_t)__popcll(f & ((1ull << (x0 & 63u)) - 1ull));
            const uint32_t fb = (uint32_t)(f >> (x0 & 63u)) & 15u;
#pragma unroll
            for (int k = 0; k < RS_WPT; ++k)
                if ((fb >> k) & 1u) { if (r < RS_WIN + 32u) L.CHL[r] = (uint16_t)(x0 + (uint32_t)k); ++r; }
        }
        fes_barrier();
        RS_STAMP(1);                                               // the window's tables

        // the next group of candidate words from word `cur` on (every wave works it out for itself: ballots, no barrier)
        auto choose = [&](uint32_t cur, rs_group &G) -> bool {
            RS_LAUNDER(tid_); tid = tid_; lane = tid & (AM_WAVE - 1);
            unsigned long long m = (lane < RS_NFW) ? L.NZ[lane < RS_NFW ? lane : 0] : 0ull;
            const uint32_t cw = cur >> 6;
            if ((uint32_t)lane < cw) m = 0ull;
            if ((uint32_t)lane == cw) m &= ~((1ull << (cur & 63u)) - 1ull);
            const unsigned long long bal = __ballot(m != 0ull);
            if (!bal) return false;                                // (uniform)
            const int l0 = __ffsll((long long)bal) - 1;
            const unsigned long long mm = __shfl(m, l0, AM_WAVE);
            G.ga = RS_UNI(64u * (uint32_t)l0 + (uint32_t)(__ffsll((long long)mm) - 1));
            G.pa = RS_UNI(L.P[G.ga]);
            G.r_base = RS_UNI(rs_rank(L, G.ga));
            // the longest run of words from ga with at most RS_NT candidates, RS_R wanted chips, RS_GW words (word ga alone: <= 32, 17,
            // 1): the conditions are monotone, 64 trial ends per round
            uint32_t lo = G.ga + 1u, hi = (nown < G.ga + RS_GW) ? nown : G.ga + RS_GW;      // ok(lo) holds; largest gb in [lo, hi] with ok(gb)
            while (hi > lo) {                                      // (uniform)
                const uint32_t step = (hi - lo + 63u) >> 6;
                uint32_t t = lo + ((uint32_t)lane + 1u) * step;
                t = t < hi ? t : hi;
                const bool ok = (uint32_t)L.P[t] - G.pa <= RS_NT && rs_rank(L, t + 16u) - G.r_base <= RS_R;
                const uint32_t nok = (uint32_t)__popcll(__ballot(ok));       // (monotone: the first nok trial ends hold)
                if (nok == AM_WAVE) { lo = hi; break; }            // (the last lane tried hi itself)
                uint32_t f = lo + (nok + 1u) * step;               // the first trial end that fails
                f = f < hi ? f : hi;
                lo = lo + nok * step;
                hi = f - 1u;
            }
            G.gb = RS_UNI(lo);
            // (chips the group's candidates read: [ga, gb + 15]; rank(gb + 16) counts the wanted ones among them)
            G.nrows = RS_UNI(rs_rank(L, G.gb + 16u) - G.r_base);
            G.nc = RS_UNI((uint32_t)L.P[G.gb] - G.pa);
            const long long Alo = A0 + ((long long)G.ga - 1) * SPC, Ahi = A0 + ((long long)G.gb + 16) * SPC;
            G.inside = wide && Alo >= a.src_abs0 && Ahi <= a.src_abs1;
            return true;
        };
        // a wave's batch of the group's rows -- the wanted chips of rank 32 wave .. + 31 --, first half: which chips, and their loads
        auto rows_issue = [&](const rs_group &G, rs_batch &B) {
            RS_LAUNDER(tid_); tid = tid_; lane = tid & (AM_WAVE - 1);
            const int j = lane & 31, half = lane >> 5;
            const uint32_t b0 = (uint32_t)wv * 32u;
            B.cnt = (G.nrows > b0) ? ((G.nrows - b0 < 32u) ? (int)(G.nrows - b0) : 32) : 0;      // (wave-uniform)
            B.live = j < B.cnt;
            B.start = false; B.srank = 0; B.nrow = 0; B.ci = 0;
            if (B.cnt == 0) return;                                // (wave-uniform)
            B.ci = L.CHL[G.r_base + b0 + (uint32_t)(B.live ? j : 0)];
            // runs: the chip before chip j is lane j - 1's own unless a run starts at j.  Wanted chips come in runs of 17 and more (only
            // the run a group begins in can be cut shorter): 32 consecutive ones hold at most three run starts
            const uint32_t cprev = (uint32_t)__shfl((int)B.ci, (lane + AM_WAVE - 1) & (AM_WAVE - 1), AM_WAVE);
            B.start = B.live && (j == 0 || cprev + 1u != B.ci);
            const uint32_t smask = (uint32_t)__ballot(B.start);
            B.srank = __popc(smask & ((1u << j) - 1u));
            B.srank = B.srank < RS_XE ? B.srank : RS_XE - 1;       // (never: see above)
            const int nstart = PMF ? __popc(smask) : 0;
            B.nrow = B.cnt + (nstart < RS_XE ? nstart : RS_XE);
            uint16_t *const RC = L.RC + wv * (32 + RS_XE);
            if (half == 0 && B.live) {
                RC[j] = (uint16_t)(B.ci + 1u);
                if (PMF && B.start) RC[B.cnt + B.srank] = (uint16_t)B.ci;     // (the chip before: ci - 1, stored + 1)
            }
            __builtin_amdgcn_wave_barrier();
            if (G.inside) {
                const long long Alo = A0 + ((long long)G.ga - 1) * SPC;
                unsigned long long gbase = reinterpret_cast<unsigned long long>(iq2 + (Alo - a.src_abs0));
                gbase = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(gbase >> 32)) << 32) |
                        (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)gbase);
                const int np = B.nrow * 16;
                // (straight-line: a lane without a piece loads piece 0 again)
#pragma unroll
                for (int r = 0; r < RS_MAXR; ++r) {
                    const int p = lane + AM_WAVE * r;
                    const int pc = p < np ? p : 0;
                    const unsigned off = ((unsigned)RC[pc >> 4] - G.ga) * (unsigned)(SPC * 8) + (unsigned)(pc & 15) * 16u;
                    B.v[r] = fes_gload16_cached_at(gbase, off);
                }
            }
        };
        // ... second half: |.|^2 into the rows, pulse-matched filter in the canonical order, the rows' maxima
        auto rows_finish = [&](const rs_group &G, rs_batch &B) {
            if (B.cnt == 0) return;                                // (wave-uniform)
            RS_LAUNDER(tid_); tid = tid_; lane = tid & (AM_WAVE - 1);
            const int j = lane & 31, half = lane >> 5;
            const uint32_t b0 = (uint32_t)wv * 32u;
            float *const XMw = L.XM + b0 * RS_XS;
            float *const XEw = L.XE + wv * (RS_XE * RS_XS);
            const uint16_t *const RC = L.RC + wv * (32 + RS_XE);
            const int np = B.nrow * 16;
            if (G.inside) {
#pragma unroll
                for (int r = 0; r < RS_MAXR; ++r) {
                    const int p = lane + AM_WAVE * r;
                    if (p < np) {
                        const fes_f2 q0 = fes_pk_mul(fes_mk2(B.v[r].x, B.v[r].y), fes_mk2(B.v[r].x, B.v[r].y));
                        const fes_f2 q1 = fes_pk_mul(fes_mk2(B.v[r].z, B.v[r].w), fes_mk2(B.v[r].z, B.v[r].w));
                        float2 mm;
                        mm.x = q0.x + q0.y;                           // a1: fl(fl(I*I) + fl(Q*Q))
                        mm.y = q1.x + q1.y;
                        const int s = p >> 4;
                        float *row = (s < B.cnt) ? XMw + s * RS_XS : XEw + (s - B.cnt) * RS_XS;
                        *reinterpret_cast<float2 *>(row + 2 * (p & 15)) = mm;
                    }
                }
            } else {
                // stream edges / unaligned input: one sample at a time, zeros outside the stream (rare: kept small)
#pragma unroll 1
                for (int p = lane; p < np; p += AM_WAVE) {
                    const int s = p >> 4;
                    const long long aa = A0 + ((long long)RC[s] - 1) * SPC + 2 * (p & 15);
                    float2 u0, u1;
                    u0.x = 0.0f; u0.y = 0.0f; u1 = u0;
                    if (aa >= a.src_abs0 && aa < a.src_abs1) u0 = iq2[aa - a.src_abs0];
                    if (aa + 1 >= a.src_abs0 && aa + 1 < a.src_abs1) u1 = iq2[aa + 1 - a.src_abs0];
                    const float r0 = u0.x * u0.x, i0 = u0.y * u0.y, r1 = u1.x * u1.x, i1 = u1.y * u1.y;
                    float2 mm;
                    mm.x = r0 + i0;
                    mm.y = r1 + i1;
                    float *row = (s < B.cnt) ? XMw + s * RS_XS : XEw + (s - B.cnt) * RS_XS;
                    *reinterpret_cast<float2 *>(row + 2 * (p & 15)) = mm;
                }
            }
            __builtin_amdgcn_wave_barrier();
            // positions beyond the end of the stream read as zero (the preamble view pads with zeros)
            const long long leftn = a.src_abs1 - (A0 + (long long)B.ci * SPC);
            const int nin = leftn >= SPC ? SPC : (leftn <= 0 ? 0 : (int)leftn);
            if (PMF) {
                // lane j: prefix sums of the own chip, left->right; lane j + 32: suffix sums of the chip before (its row reversed, so both
                // run the same chain); bb[i] = fl((suf[i + 1] + pre[i]) s1), the chip's last sample: pre alone
                float c[SPC];
                if (B.live) {
                    const float *rowp = half ? (B.start ? XEw + B.srank * RS_XS : XMw + (j - 1) * RS_XS) : XMw + j * RS_XS;
                    const float4 *row = reinterpret_cast<const float4 *>(rowp);
#pragma unroll
                    for (int k = 0; k < SPC / 4; ++k) {
                        const float4 u = row[half ? SPC / 4 - 1 - k : k];
                        c[4 * k] = half ? u.w : u.x; c[4 * k + 1] = half ? u.z : u.y; c[4 * k + 2] = half ? u.y : u.z; c[4 * k + 3] = half ? u.x : u.w;
                    }
                    float ap = 0.0f;
#pragma unroll
                    for (int i = 0; i < SPC; ++i) { ap = ap + c[i]; c[i] = ap; }       // lane j: pre[i]; lane j + 32: suf[31 - i]
                } else {
#pragma unroll
                    for (int i = 0; i < SPC; ++i) c[i] = 0.0f;
                }
                __builtin_amdgcn_wave_barrier();                      // (every lane has read its row)
                float4 *own = reinterpret_cast<float4 *>(XMw + j * RS_XS);
                float rmx = 0.0f;                                     // the row's largest value (fmaxf: a NaN is no sample above anything)
                const bool edge = __ballot(B.live && nin < SPC) != 0ull;             // (wave-uniform) a chip at the end of the stream
                const fes_f2 s2 = fes_mk2(a.s1, a.s1);
#pragma unroll
                for (int k = 0; k < SPC / 4; ++k) {
                    // suf[i + 1] + pre[i], the chip's last sample: pre alone (x + (-0) == x for every x, NaN included);
                    // two positions per instruction where the operands pair up (v_pk_add_f32 / v_pk_mul_f32: the scalar forms' rounding)
                    const float x0 = rs_from_upper_half(c[SPC - 2 - 4 * k], lane), x1 = rs_from_upper_half(c[SPC - 3 - 4 * k], lane);
                    const float x2 = rs_from_upper_half(c[SPC - 4 - 4 * k], lane);
                    const float x3 = (4 * k + 3 < SPC - 1) ? rs_from_upper_half(c[(4 * k + 3 < SPC - 1) ? SPC - 5 - 4 * k : 0], lane) : -0.0f;
                    fes_f2 o01 = fes_pk_mul(fes_pk_add(fes_mk2(x0, x1), fes_mk2(c[4 * k], c[4 * k + 1])), s2);
                    fes_f2 o23 = fes_pk_mul(fes_pk_add(fes_mk2(x2, x3), fes_mk2(c[4 * k + 2], c[4 * k + 3])), s2);
                    if (edge) {
                        if (4 * k >= nin) o01.x = 0.0f;
                        if (4 * k + 1 >= nin) o01.y = 0.0f;
                        if (4 * k + 2 >= nin) o23.x = 0.0f;
                        if (4 * k + 3 >= nin) o23.y = 0.0f;
                    }
                    const float qm = fmaxf(fmaxf(o01.x, o01.y), fmaxf(o23.x, o23.y));
                    rmx = fmaxf(rmx, qm);
                    if (B.live && half == 0) {
                        float4 o; o.x = o01.x; o.y = o01.y; o.z = o23.x; o.w = o23.y; own[k] = o;
                        L.QMAX[(b0 + (uint32_t)j) * 8u + (uint32_t)k] = qm;
                    }
#if defined(__HIP_DEVICE_COMPILE__)
                    __builtin_amdgcn_sched_barrier(0);                // (four positions at a time: hoisted, the 31 exchanges cost 31 registers more)
#endif
                }
                if (B.live && half == 0) L.RMAX[b0 + (uint32_t)j] = rmx;
            } else if (B.live && half == 0) {
                // no filter: bb is |.|^2 itself; only the end of the stream needs a hand
                float *own = XMw + j * RS_XS;
                for (int i = nin; i < SPC; ++i) own[i] = 0.0f;
                float rmx = 0.0f;
                for (int k = 0; k < SPC / 4; ++k) {
                    const float qm = fmaxf(fmaxf(own[4 * k], own[4 * k + 1]), fmaxf(own[4 * k + 2], own[4 * k + 3]));
                    L.QMAX[(b0 + (uint32_t)j) * 8u + (uint32_t)k] = qm;
                    rmx = fmaxf(rmx, qm);
                }
                L.RMAX[b0 + (uint32_t)j] = rmx;
            }
        };

        // ---- groups of consecutive candidate words ------------------------------------------------------------------------------
        rs_group G, Gn;
        bool have = choose(0u, G);
        RS_STAMP(2);
        while (have) {                                             // (uniform)
#if defined(RS_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
            if (tid == 0) { rsp[8] += 1; rsp[9] += G.nc; }
#endif
            {
                // (the next group's loads issued here, ahead of this group's refinement, were tried: nine 16-byte registers per lane held
                // across the refinement cost more in spills -- each right behind its load, a memory round trip apiece -- than the overlap
                // gained; the other workgroups of the CU are what a group's load latency hides behind)
                rs_batch B;
                rows_issue(G, B);
                rows_finish(G, B);
            }
            RS_STAMP(3);                                           // wave 0's rows
            fes_barrier();                                         // the group's rows are complete
            RS_STAMP(4);                                           // ... the other waves'
            const bool have_n = choose(G.gb, Gn);
            RS_STAMP(5);                                           // next group chosen
            RS_LAUNDER(tid_); tid = tid_; lane = tid & (AM_WAVE - 1);

            // ---- late-peak decisions + the candidate list: a quarter of a bitmap word per lane -----------------------------------
            // Word w of the group (ga <= w <= gb: the positions of word gb belong to candidates of word gb - 1) holds the positions
            // 32 w + b - 288; the ones some candidate of the group can shift to -- from a candidate's own position up to 31 on -- are
            //   D = (bits of w at and above its lowest candidate) | (bits of w below the highest candidate of word w - 1).
            // late(q) = E(q + 1) > E(q), all the late-peak search asks (preamble_impl.cc:184-192), from the EXACT difference of the two
            // sums (they share all but eight samples; am_k_refine_late has the error bound), the reference's own two sums for close calls.
            {
                // V bounds every sample the group's positions can see: the largest bb of the front-end workgroups whose segments they span
                float vb = 0.0f;
                {
                    const uint32_t p_lo = (wa + G.ga) * 32u, p_hi = (wa + G.gb) * 32u + 13u * (uint32_t)SPC;
                    // (front-end workgroup of an array coordinate: n_long segments of vspan coordinates, then shorter ones)
                    const uint32_t c_long = a.n_long * a.vspan;
                    const uint32_t q_lo = p_lo > 288u ? p_lo - 288u : 0u;
                    uint32_t v0 = q_lo < c_long ? q_lo / a.vspan : a.n_long + (q_lo - c_long) / a.vspan_short;
                    uint32_t v1 = p_hi < c_long ? p_hi / a.vspan : a.n_long + (p_hi - c_long) / a.vspan_short;
                    v0 = v0 < a.nv ? v0 : a.nv - 1u;
                    v1 = v1 < a.nv ? v1 : a.nv - 1u;
                    for (uint32_t v = v0; v <= v1; ++v) vb = fmaxf(vb, a.wg_max[v]);
                }
                const double bound = (double)vb * 0x1p-36;            // (+inf when a sample is not finite: nothing is decided by D)
                // (a quarter word per lane over the group's WANTED chips -- every word with a position to decide is one: it holds a
                // candidate or the word before it does --, not over all its words: most words between two bursts hold nothing, and the
                // chip's rank is its row)
                const uint32_t nunit = 4u * G.nrows;
                for (uint32_t u0 = 0; u0 < nunit; u0 += RS_NT) {      // (uniform)
                    const uint32_t u = u0 + (uint32_t)tid;
                    const bool inr = u < nunit;
                    const uint32_t row0 = inr ? (u >> 2) : 0u, qt = u & 3u;
                    const uint32_t w = L.CHL[G.r_base + row0];
                    const bool act = inr && w <= G.gb;
                    const uint32_t C = (act && w < G.gb) ? L.W[w] : 0u;
                    const uint32_t Cp = (act && w > G.ga) ? L.W[w - 1u] : 0u;
                    uint32_t D = C ? ~((C & (0u - C)) - 1u) : 0u;
                    if (Cp) D |= (1u << (31 - __clz((int)Cp))) - 1u;
                    const uint32_t Dq = (D >> (8u * qt)) & 0xFFu, Cq = (C >> (8u * qt)) & 0xFFu;
                    uint32_t lb = 0;
                    if (Dq | Cq) {
                        uint32_t idx = (uint32_t)L.P[w] - G.pa + (uint32_t)__popc(C & ((1u << (8u * qt)) - 1u));
                        if (Dq) {
                            // the quarter's eight positions four at a time: 8 reads of 16 bytes each; D = (x1 - x0) + (x3 - x2) + (x8 - x7) + (x10 - x9)
                            constexpr int XS4 = RS_XS / 4;
                            uint32_t closem = 0;
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                if (((Dq >> (4 * h)) & 15u) == 0u) continue;
                                const float4 *pr = reinterpret_cast<const float4 *>(L.XM + row0 * RS_XS + 8u * qt) + h;
                                double dd[4];
                                {
                                    const float4 lo4 = pr[0], hi4 = pr[XS4];
                                    dd[0] = (double)hi4.x - (double)lo4.x; dd[1] = (double)hi4.y - (double)lo4.y;
                                    dd[2] = (double)hi4.z - (double)lo4.z; dd[3] = (double)hi4.w - (double)lo4.w;
                                }
#pragma unroll
                                for (int c = 0; c < 3; ++c) {
                                    const int r = c == 0 ? 2 : (c == 1 ? 7 : 9);
                                    const float4 lo4 = pr[r * XS4], hi4 = pr[(r + 1) * XS4];
                                    dd[0] = dd[0] + ((double)hi4.x - (double)lo4.x); dd[1] = dd[1] + ((double)hi4.y - (double)lo4.y);
                                    dd[2] = dd[2] + ((double)hi4.z - (double)lo4.z); dd[3] = dd[3] + ((double)hi4.w - (double)lo4.w);
                                }
#pragma unroll
                                for (int b = 0; b < 4; ++b) {
                                    if (fabs(dd[b]) > bound) lb |= (dd[b] > 0.0) ? (1u << (4 * h + b)) : 0u;
                                    else closem |= 1u << (4 * h + b);
                                }
                            }
                            lb &= Dq;
                            closem &= Dq;
                            // (rare: exact ties of quantised or constant input, non-finite samples) the reference's two sequential sums
#pragma unroll 1
                            while (closem) {
                                const uint32_t b = (uint32_t)(__ffs((int)closem) - 1);
                                if (rs_energy(L.XM, row0, 8u * qt + b + 1u) > rs_energy(L.XM, row0, 8u * qt + b)) lb |= 1u << b;
                                closem &= closem - 1u;
                            }
                        }
                        uint32_t cq = Cq;
#pragma unroll 1
                        while (cq) {
                            const uint32_t b = (uint32_t)(__ffs((int)cq) - 1);
                            L.POSL[idx] = (wa + w) * 32u + 8u * qt + b - 288u;
                            L.CR0[idx] = (uint16_t)row0;
                            ++idx;
                            cq &= cq - 1u;
                        }
                    }
                    uint32_t word = lb << (8u * qt);
                    word |= (uint32_t)__shfl_xor((int)word, 1, AM_WAVE);
                    word |= (uint32_t)__shfl_xor((int)word, 2, AM_WAVE);
                    if (act && qt == 0u && (C | Cp) != 0u) L.LATEW[w - G.ga] = word;     // (every word a candidate of the group reads: its own and the next)
                }
            }
            fes_barrier();
            RS_STAMP(6);                                           // late decisions
            RS_LAUNDER(tid_); tid = tid_; lane = tid & (AM_WAVE - 1);
            // ---- one lane per candidate ---------------------------------------------------------------------------------------------
            if ((uint32_t)tid < G.nc) {
                const uint32_t i = (uint32_t)tid;
                const uint32_t jpos = L.POSL[i];
                const uint32_t jw = ((jpos + 288u) >> 5) - wa, jb = (jpos + 288u) & 31u;     // word of the window, bit
                uint32_t how_late;
                {
                    // consecutive late decisions from the own position on, at most 32 (how_late < samples per chip, :192)
                    const uint32_t l0 = L.LATEW[jw - G.ga], l1 = L.LATEW[jw - G.ga + 1u];
                    const uint32_t x = jb ? ((l0 >> jb) | (l1 << (32u - jb))) : l0;
                    how_late = (x == 0xFFFFFFFFu) ? 32u : (uint32_t)(__ffs((int)~x) - 1);
                }
                const uint32_t e = jpos + how_late;
                const float av = (e >= a.end_j) ? 0.0f : a.avg_sparse[e];   // beyond the end of the stream: 0
                const uint32_t ee = e + 288u;
                const int o = (int)(ee & 31u);
                const uint32_t rowe = (uint32_t)L.CR0[i] + ((ee >> 5) - (wa + jw));
                const float *pe_ = L.XM + rowe * RS_XS;
                const float p0 = pe_[o], p1 = pe_[2 * RS_XS + o], p2 = pe_[7 * RS_XS + o], p3 = pe_[9 * RS_XS + o];
                float ps = p0 + p1;                                       // quiet zones (preamble_impl.cc:198-209)
                ps = ps + p2;
                ps = ps + p3;
                const float avgpeak = (float)((double)ps / 4.0);
                const float sthr = av + (avgpeak - av) / a.thr_lin;
                // with e at offset o of chip c the zones [e + 96, e + 192] and [e + 320, e + 480] are: chip c + 3 from o on, chips c + 4,
                // c + 5 whole, chip c + 6 up to o; chip c + 10 from o on, chips c + 11 .. c + 14 whole, chip c + 15 up to o
                const float *mx = L.RMAX + rowe;
                bool hit = mx[4] > sthr || mx[5] > sthr || mx[11] > sthr || mx[12] > sthr || mx[13] > sthr || mx[14] > sthr;
                if (!hit) {
                    const float *qx = L.QMAX + rowe * 8u;
                    const float m1 = fmaxf(rs_row_partial_max(pe_ + 3 * RS_XS, qx + 3 * 8, o, true), rs_row_partial_max(pe_ + 10 * RS_XS, qx + 10 * 8, o, true));
                    const float m2 = fmaxf(rs_row_partial_max(pe_ + 6 * RS_XS, qx + 6 * 8, o, false), rs_row_partial_max(pe_ + 15 * RS_XS, qx + 15 * 8, o, false));
                    hit = m1 > sthr || m2 > sthr;
                }
                const bool ok = !hit;
                const uint32_t tg = ok ? (e + (uint32_t)(AM_BURST * SPC)) : (e + 1u);   // :237 / :209
                // the greedy chain's successor: the number of candidates below the resume position
                uint32_t succ;
                {
                    const uint32_t tw = (tg + 288u) >> 5, tb = (tg + 288u) & 31u;
                    if (tw >= a.nwords) succ = total;
                    else {
                        const uint32_t rel = tw - wa;                     // < RS_NWORD: the resume position lies <= 241 words behind the window's last
                        succ = cbase + (uint32_t)L.P[rel] + (uint32_t)__popc(L.W[rel] & ((1u << tb) - 1u));
                    }
                    succ = succ < M ? succ : M;
                }
                const uint32_t gi = cbase + G.pa + i;
                if (gi < a.Mcap) {
                    a.pos[gi] = jpos;
                    a.e[gi] = e;
                    a.inavg[gi] = av;
                    a.valid[gi] = ok ? 1 : 0;
                    a.tgt[gi] = tg;
                    a.jump0[gi] = succ;
                    if (gi == M - 1u) a.jump0[M] = M;
                }
            }
            fes_barrier();                                                // (the rows and the lists are rewritten by the next group)
            RS_STAMP(7);                                                  // candidates
            G = Gn;
            have = have_n;
        }
        cbase += (uint32_t)L.P[nown];
    }
#if defined(RS_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    if (a.prof && tid == 0) {
        rsp[11] = (long long)wall_clock64();
        for (int k = 0; k < 12; ++k) a.prof[(size_t)blockIdx.x * 12 + k] = rsp[k];
    }
#endif
}

hipError_t am_launch_refine_seg(const uint32_t *bits, const uint32_t *wg_cnt, const float *wg_max, uint32_t nwg, uint32_t n_long,
                                uint32_t words_per_wg, uint32_t words_per_step, uint32_t nwords, uint32_t Mcap, uint32_t lag, uint32_t wbits, uint32_t vspan, uint32_t nv,
                                const am_rows_args &rows, const float *avg_sparse, float thr_lin, uint32_t end_j, uint32_t *pos,
                                uint32_t *e, uint32_t *tgt, float *inavg, uint8_t *valid, uint32_t *jump0, uint32_t *total_out,
                                hipStream_t s)
{
    if (nwg == 0 || nwords == 0) return hipSuccess;
    // (am_k_fe3's bitmap: a word = one 32-sample chip, lag 288)
    if (wbits != 32 || lag != 288 || words_per_wg == 0 || !rows.iq || vspan == 0 || nv == 0) return hipErrorInvalidValue;
    am_rseg_args a;
    a.bits = bits; a.wg_cnt = wg_cnt; a.wg_max = wg_max; a.nwg = nwg; a.words_per_wg = words_per_wg; a.nwords = nwords; a.Mcap = Mcap;
    a.vspan = vspan; a.nv = nv; a.end_j = end_j; a.iq = rows.iq; a.src_abs0 = rows.src_abs0; a.src_abs1 = rows.src_abs1;
    a.out_abs0 = rows.out_abs0; a.avg_sparse = avg_sparse; a.s1 = rows.s1; a.thr_lin = thr_lin; a.pos = pos; a.e = e; a.tgt = tgt;
    a.jump0 = jump0; a.total_out = total_out; a.inavg = inavg; a.valid = valid;
    // levelled segments: the first n_long of words_per_wg words, the others one step shorter (n_long = 0 or >= nwg: all alike)
    a.n_long = (n_long == 0 || n_long > nwg) ? nwg : n_long;
    if (a.n_long < nwg && (words_per_step == 0 || words_per_step >= words_per_wg)) return hipErrorInvalidValue;
    a.words_short = a.n_long < nwg ? words_per_wg - words_per_step : words_per_wg;
    a.vspan_short = a.n_long < nwg ? (uint32_t)((unsigned long long)vspan * a.words_short / words_per_wg) : vspan;
    // shares of a segment: about RS_PW words each, at most RS_MAXPARTS (longer segments: several windows per share)
    a.parts = (words_per_wg + RS_PW - 1) / RS_PW;
    if (a.parts > RS_MAXPARTS) a.parts = RS_MAXPARTS;
    if (a.parts == 0) a.parts = 1;
    a.pw = (words_per_wg + a.parts - 1) / a.parts;
    const uint32_t grid = nwg * a.parts;
    a.prof = nullptr;
#if defined(RS_PROFILE)
    if (hipMalloc(reinterpret_cast<void **>(&a.prof), (size_t)grid * 12 * sizeof(long long)) != hipSuccess) a.prof = nullptr;
    if (a.prof) (void)hipMemsetAsync(a.prof, 0, (size_t)grid * 12 * sizeof(long long), s);
#endif
    if (rows.use_pmf) hipLaunchKernelGGL(am_k_refine_seg<true>, dim3(grid), dim3(RS_NT), 0, s, a);
    else hipLaunchKernelGGL(am_k_refine_seg<false>, dim3(grid), dim3(RS_NT), 0, s, a);
    const hipError_t lrc = hipGetLastError();
#if defined(RS_PROFILE)
    if (a.prof) {
        // blocking; prints where the workgroups' time went (10 ns ticks -> us) and the launch's timeline -- never in the default build
        std::vector<long long> h((size_t)grid * 12);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), a.prof, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        (void)hipFree(a.prof);
        static const char *names[8] = {"counts+words", "window tables", "first group", "rows finish (wave 0)", "wait other waves", "next group + issue", "late decisions", "candidates"};
        double acc[12] = {};
        long long first = 0, last = 0;
        std::vector<double> st, en, du;
        for (uint32_t b = 0; b < grid; ++b) if (h[(size_t)b * 12 + 11]) { if (!first || h[(size_t)b * 12 + 10] < first) first = h[(size_t)b * 12 + 10]; last = std::max(last, h[(size_t)b * 12 + 11]); }
        for (uint32_t b = 0; b < grid; ++b) {
            if (!h[(size_t)b * 12 + 11]) continue;
            for (int k = 0; k < 10; ++k) acc[k] += (double)h[(size_t)b * 12 + k];
            st.push_back((double)(h[(size_t)b * 12 + 10] - first) * 0.01); en.push_back((double)(h[(size_t)b * 12 + 11] - first) * 0.01); du.push_back(en.back() - st.back());
        }
        const double nw = (double)st.size();
        fprintf(stderr, "rseg: %u workgroups (%u shares per segment, %u words each), span %.1f us, %.2f groups and %.0f candidates per workgroup; us per workgroup:", (unsigned)st.size(), a.parts, a.pw, (double)(last - first) * 0.01,
                acc[8] / nw, acc[9] / nw);
        for (int k = 0; k < 8; ++k) fprintf(stderr, " %s %.2f", names[k], acc[k] / nw * 0.01);
        fprintf(stderr, "\n");
        auto pct = [&](std::vector<double> v, const char *name) {
            std::sort(v.begin(), v.end());
            const size_t n = v.size();
            if (n) fprintf(stderr, "rseg timeline %-10s us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f\n", name, v[0], v[n / 10], v[n / 2], v[(size_t)(n * 0.9)], v[n - 1]);
        };
        pct(st, "start"); pct(en, "end"); pct(du, "duration");
    }
#endif
    return lrc;
}
