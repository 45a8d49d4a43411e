// am_refine_seg.hip -- candidate list + bb rows + refinement in ONE launch at 32 samples per chip (64 Msps; round 6).
//
// Until round 5 three launches stood between the streaming front end and the greedy chain: am_k_gather_wg<1> listed the candidates of
// the bitmap and formed, from the samples, the bb rows around them (42 MB written per 64 M samples at the bench density), and
// am_k_refine_late read those rows back (50 MB) for the late-peak search (lib/preamble_impl.cc:90-98,182-192) and the quiet zones
// (:198-209).  Here one workgroup per front-end segment does all of it with the rows in LDS: nothing is written that is only read
// back, one launch and its gap are gone, and the refinement's chains of memory round trips (positions, eight bb samples per
// position, six row maxima, four partial rows, a galloping search for the successor) become LDS reads and popcounts.
//
//   * the segment's bitmap words (+ the 256 behind them: a hit's resume position lies at most 241 words on) stay in LDS with the
//     exclusive prefix of their popcounts: the flat index of a candidate, and the chain's successor -- the number of candidates
//     below the resume position -- are a table read and a popcount;
//   * the segment is worked off in GROUPS of consecutive candidate words: as many as give at most RS_NT candidates and RS_R wanted
//     chips (a candidate in word w reads the chips w .. w + 16: bit b of word w is position 32 w + b - 288, array chip w - 9).  A
//     group's wanted chips are formed from IQ in the canonical order of DESIGN.md 3 -- 32 per wave, all of a batch's loads in flight
//     together, exactly as am_rows_segment32 did -- into rows that are CONSECUTIVE in LDS for consecutive chips of a run, so a
//     candidate addresses everything it reads from the row of its own chip;
//   * late-peak decisions once per position the group's candidates can reach (exact difference of the two sums; the reference's two
//     sequential sums only for close calls), then one lane per candidate: shift, reference level (the one global load left),
//     quiet zones from the row maxima and four partial rows, the record, the successor.
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
#else
#define AM_PIN_RS(x) ((void)0)
#endif

#define RS_NT 256                         /* threads = candidates per group */
#define RS_NWV (RS_NT / AM_WAVE)
#ifndef RS_R
#define RS_R 128                          /* bb rows per group: 32 per wave and batch */
#endif
#define RS_XE 4                           /* rows for the chip before a run's first, per wave (a batch of 32 holds at most 3 run starts) */
#define RS_XS 36                          /* floats per LDS row: 32 + 4 pad (16-byte reads of consecutive rows hit all banks) */
#define RS_BBW 17                         /* chips a candidate reads from its own chip on */
#define RS_WPT 7                          /* bitmap words per thread and window */
#define RS_NWORD (RS_NT * RS_WPT)         /* words in LDS: a window of the segment + the words behind it */
#define RS_AHEAD 256                      /* ... of which behind the window (a resume position lies <= 241 words on) */
#define RS_WIN (RS_NWORD - RS_AHEAD)      /* words of the segment per window (am_k_fe3 at 64 M samples: 1 344 per segment, one window) */
#define RS_NFW (RS_NWORD / 64 + 1)        /* 64-bit words of per-word flags */
#ifndef RS_WPS
#define RS_WPS 4                          /* waves per SIMD the kernel is compiled for (<= 128 VGPRs; ~37 KB of LDS: four workgroups per CU) */
#endif
static_assert(RS_R % 32 == 0 && RS_R / 32 == RS_NWV, "one batch of 32 rows per wave and group");
static_assert(RS_AHEAD >= AM_BURST + 2 && RS_WIN > 0, "a hit's resume position must lie inside the words in LDS");

struct am_rseg_lds {
    uint32_t W[RS_NWORD];                 // bitmap words of the window (+ RS_AHEAD behind it)
    uint16_t P[RS_NWORD + 2];             // exclusive prefix of their popcounts
    unsigned long long NZ[RS_NFW];        // word holds a candidate (own words only)
    unsigned long long FLW[RS_NFW];       // chip is wanted: a candidate in one of the 16 words before it or in its own
    uint32_t PFW[RS_NFW + 1];             // exclusive prefix of popcount(FLW)
    __attribute__((aligned(16))) float XM[RS_R * RS_XS];               // the group's rows: |.|^2, then bb
    __attribute__((aligned(16))) float XE[RS_NWV * RS_XE * RS_XS];     // per wave: |.|^2 of the chip before a run's first
    float RMAX[RS_R];                     // largest bb of a row
    uint16_t RC[RS_NWV * (32 + RS_XE)];   // per wave: chip (+ 1) held by row slot s
    uint32_t POSL[RS_NT];                 // the group's candidates: position,
    uint16_t CR0[RS_NT];                  // ... row of its chip
    uint32_t coff[RS_NT + 1], clo[RS_NT]; // compact index of the first position candidate i owns (+ end) / that position
    uint32_t LATE[RS_NT];                 // one bit per owned position: E(q + 1) > E(q)
    uint32_t ws[RS_NWV], red[RS_NWV], red2[RS_NWV];
};

struct am_rseg_args {
    const uint32_t *bits, *wg_cnt;
    const float *wg_max;
    uint32_t nwg, words_per_wg, nwords, Mcap, vspan, nv, end_j;
    const float *iq;
    long long src_abs0, src_abs1, out_abs0;
    const float *avg_sparse;
    float s1, thr_lin;
    uint32_t *pos, *e, *tgt, *jump0, *total_out;
    float *inavg;
    uint8_t *valid;
    long long *prof;                      // profiling builds (-DRS_PROFILE): [nwg][12] wall-clock ticks (10 ns) per phase, else null
};

// Profiling builds only (tools/build_variants.sh, -DRS_PROFILE): where a workgroup's time goes, stamped by thread 0
#if defined(RS_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define RS_STAMP(k) do { if (tid == 0) { const long long now__ = (long long)wall_clock64(); rsp[k] += now__ - rsl; rsl = now__; } } while (0)
#else
#define RS_STAMP(k) do { } while (0)
#endif

// number of wanted chips below chip x of the window
__device__ __forceinline__ uint32_t rs_rank(const am_rseg_lds &L, uint32_t x)
{
    return L.PFW[x >> 6] + (uint32_t)__popcll(L.FLW[x >> 6] & ((1ull << (x & 63u)) - 1ull));
}

// the reference's sequential double-precision sum over the four pulses from offset o (0 .. 32) of row `row` on (preamble_impl.cc:91-98)
__device__ __forceinline__ double rs_energy(const float *XM, uint32_t row, uint32_t o)
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

// any of the samples i >= o (from_o) or i <= o of the rows ra, rb above thr
__device__ __forceinline__ bool rs_rows_partly_above(const float *ra, const float *rb, int o, bool from_o, float thr)
{
    bool hit = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float4 a = reinterpret_cast<const float4 *>(ra)[k], b = reinterpret_cast<const float4 *>(rb)[k];
        const float a4[4] = {a.x, a.y, a.z, a.w}, b4[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = 4 * k + q;
            const bool in = from_o ? (i >= o) : (i <= o);
            hit = hit || (in && (a4[q] > thr || b4[q] > thr));
        }
    }
    return hit;
}

template <bool PMF>
__global__ void __launch_bounds__(RS_NT, RS_WPS) am_k_refine_seg(am_rseg_args a)
{
    constexpr int SPC = 32;
    __shared__ am_rseg_lds L;
    const uint32_t g = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & (AM_WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid / AM_WAVE);
    const uint32_t w_begin = g * a.words_per_wg;
    const uint32_t w_end = (w_begin + a.words_per_wg < a.nwords) ? w_begin + a.words_per_wg : a.nwords;
#if defined(RS_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    long long rsp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long rsl = (long long)wall_clock64();
    rsp[10] = rsl;
#endif

    // the first window's words go out together with the counts of the other workgroups: one memory round trip in front of the
    // arithmetic (unconditional loads from clamped indices: a load behind a branch is waited for at the join)
    uint32_t wl[RS_WPT];
#pragma unroll
    for (int k = 0; k < RS_WPT; ++k) {
        const uint32_t x = w_begin + (uint32_t)tid + (uint32_t)k * RS_NT;
        wl[k] = a.bits[x < a.nwords ? x : a.nwords - 1u];
    }
    // where this workgroup's candidates start, and how many there are in all
    uint32_t before = 0, all = 0;
    for (uint32_t k0 = (uint32_t)tid; k0 < a.nwg; k0 += 8u * RS_NT) {
        uint32_t v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t k = k0 + (uint32_t)j * RS_NT;
            v[j] = a.wg_cnt[k < a.nwg ? k : 0u];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t k = k0 + (uint32_t)j * RS_NT;
            if (k < a.nwg) { all += v[j]; if (k < g) before += v[j]; }
        }
    }
    for (int o = AM_WAVE / 2; o >= 1; o >>= 1) {
        before += (uint32_t)__shfl_xor((int)before, o, AM_WAVE);
        all += (uint32_t)__shfl_xor((int)all, o, AM_WAVE);
    }
    if (lane == 0) { L.red[wv] = before; L.red2[wv] = all; }
    __syncthreads();
    uint32_t cbase = 0, total = 0;                             // candidates before the current window / in the whole scan
    for (int k = 0; k < RS_NWV; ++k) { cbase += L.red[k]; total += L.red2[k]; }
    const uint32_t M = total < a.Mcap ? total : a.Mcap;
    if (g == a.nwg - 1u && tid == 0) *a.total_out = total;

    RS_STAMP(0);                                               // counts + first words
    const bool wide = (reinterpret_cast<uintptr_t>(a.iq) & 15u) == 0 && (((a.out_abs0 - a.src_abs0) & 1) == 0);   // (uniform)
    const float2 *iq2 = reinterpret_cast<const float2 *>(a.iq);

    for (uint32_t wa = w_begin; wa < w_end; wa += RS_WIN) {    // windows of the segment (one, for am_k_fe3's segments at 64 M samples)
        const uint32_t nown = (w_end - wa < RS_WIN) ? w_end - wa : RS_WIN;      // the window's own words; behind them: look-ahead
        if (wa != w_begin) {
            __syncthreads();                                   // (the tables of the window before are still being read)
#pragma unroll
            for (int k = 0; k < RS_WPT; ++k) {
                const uint32_t x = wa + (uint32_t)tid + (uint32_t)k * RS_NT;
                wl[k] = a.bits[x < a.nwords ? x : a.nwords - 1u];
            }
        }
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
        __syncthreads();
        // exclusive prefix of the popcounts (a thread takes RS_WPT consecutive words: stride 7 words, no bank conflict)
        {
            uint32_t c[RS_WPT], sum = 0;
#pragma unroll
            for (int k = 0; k < RS_WPT; ++k) { c[k] = (uint32_t)__popc(L.W[(uint32_t)tid * RS_WPT + (uint32_t)k]); sum += c[k]; }
            uint32_t incl = sum;
            for (int d = 1; d < AM_WAVE; d <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, d, AM_WAVE);
                if (lane >= d) incl += up;
            }
            if (lane == AM_WAVE - 1) L.ws[wv] = incl;
            // wanted chips: a word with a candidate flags its chip and the 16 behind it (dilation by 16 bits across the 64-bit words)
            if (tid < RS_NFW) {
                const unsigned long long x = L.NZ[tid];
                unsigned long long d = x | (x << 1);
                d |= d << 2; d |= d << 4; d |= d << 8;                // shifts 0 .. 15
                static_assert(RS_BBW == 17, "a candidate's chip and the 16 after it");
                unsigned long long f = d | (x << 16);
                const uint32_t hp = tid ? (uint32_t)(L.NZ[tid - 1] >> 48) : 0u;    // the previous word's last 16 chips reach into this one
                if (hp) f |= (2ull << (31 - __clz((int)hp))) - 1ull;
                L.FLW[tid] = f;
            }
            __syncthreads();
            uint32_t off = incl - sum;
            for (int k = 0; k < wv; ++k) off += L.ws[k];
#pragma unroll
            for (int k = 0; k < RS_WPT; ++k) { L.P[(uint32_t)tid * RS_WPT + (uint32_t)k] = (uint16_t)off; off += c[k]; }
            if (tid == RS_NT - 1) L.P[RS_NWORD] = (uint16_t)off;
            if (tid <= RS_NFW) {
                uint32_t acc = 0;
                for (int k = 0; k < tid; ++k) acc += (uint32_t)__popcll(L.FLW[k]);
                L.PFW[tid] = acc;
            }
        }
        __syncthreads();

        RS_STAMP(1);                                               // the window's tables
        // ---- groups of consecutive candidate words ------------------------------------------------------------------------------
        uint32_t cur = 0;                                          // (uniform) first word of the window not yet worked off
        for (;;) {
            // first word with a candidate at or after cur
            uint32_t ga = nown;
            for (uint32_t j = cur >> 6; j < (nown + 63u) >> 6; ++j) {
                unsigned long long m = L.NZ[j];
                if (j == (cur >> 6)) m &= ~((1ull << (cur & 63u)) - 1ull);
                if (m) { ga = 64u * j + (uint32_t)(__ffsll((long long)m) - 1); break; }
            }
            if (ga >= nown) break;
            // the longest run of words from ga with at most RS_NT candidates and RS_R wanted chips (word ga alone: <= 32, 17)
            const uint32_t pa = L.P[ga], r_base = rs_rank(L, ga);
            uint32_t gb;
            {
                uint32_t lo = ga + 1u, hi = nown;                  // ok(lo) holds; largest gb in [lo, hi] with ok(gb)
                while (lo < hi) {
                    const uint32_t mid = (lo + hi + 1u) >> 1;
                    const bool ok = (uint32_t)L.P[mid] - pa <= RS_NT && rs_rank(L, mid + 16u) - r_base <= RS_R;
                    if (ok) lo = mid; else hi = mid - 1u;
                }
                gb = lo;
            }
            // (chips the group's candidates read: [ga, gb + 15]; rank(gb + 16) counts the wanted ones among them)
            const uint32_t nrows = rs_rank(L, gb + 16u) - r_base;
            const uint32_t nc = (uint32_t)L.P[gb] - pa;
            cur = gb;
            RS_STAMP(2);                                           // group chosen
#if defined(RS_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
            if (tid == 0) { rsp[8] += 1; rsp[9] += nc; }
#endif

            // ---- the group's rows from IQ (canonical order): wave v takes the wanted chips of rank 32 v .. 32 v + 31 --
            {
                const int j = lane & 31, half = lane >> 5;
                const uint32_t b0 = (uint32_t)wv * 32u;
                const int cnt = (nrows > b0) ? ((nrows - b0 < 32u) ? (int)(nrows - b0) : 32) : 0;     // (wave-uniform)
                const bool live = j < cnt;
                uint32_t ci;                                       // the lane's chip (window index)
                {
                    const uint32_t r = r_base + b0 + (uint32_t)(live ? j : 0);
                    uint32_t lo = 0, hi = RS_NFW;                  // last flag word with PFW <= r
                    while (hi - lo > 1u) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (L.PFW[mid] <= r) lo = mid; else hi = mid;
                    }
                    uint32_t rr = r - L.PFW[lo];
                    unsigned long long y = L.FLW[lo];
                    uint32_t p = 0;
#pragma unroll
                    for (int sh = 32; sh >= 1; sh >>= 1) {
                        const unsigned long long lowpart = y & ((1ull << sh) - 1ull);
                        const uint32_t c = (uint32_t)__popcll(lowpart);
                        if (rr >= c) { rr -= c; y >>= sh; p += (uint32_t)sh; } else y = lowpart;
                    }
                    ci = 64u * lo + p;
                }
                // runs: the chip before chip j is lane j - 1's own unless a run starts at j.  Wanted chips come in runs of 17 and more
                // (only the run a group begins in can be cut shorter): 32 consecutive ones hold at most three run starts
                const uint32_t cprev = (uint32_t)__shfl((int)ci, (lane + AM_WAVE - 1) & (AM_WAVE - 1), AM_WAVE);
                const bool start = live && (j == 0 || cprev + 1u != ci);
                const uint32_t smask = (uint32_t)__ballot(start);
                int srank = __popc(smask & ((1u << j) - 1u));
                srank = srank < RS_XE ? srank : RS_XE - 1;        // (never: see above)
                const int nstart = PMF ? __popc(smask) : 0;
                const int nrow = cnt + (nstart < RS_XE ? nstart : RS_XE);
                uint16_t *const RC = L.RC + wv * (32 + RS_XE);
                float *const XMw = L.XM + b0 * RS_XS;
                float *const XEw = L.XE + wv * (RS_XE * RS_XS);
                if (cnt > 0) {                                     // (wave-uniform: a wave without rows has nothing to load)
                if (half == 0 && live) {
                    RC[j] = (uint16_t)(ci + 1u);
                    if (PMF && start) RC[cnt + srank] = (uint16_t)ci;       // (the chip before: ci - 1, stored + 1)
                }
                __builtin_amdgcn_wave_barrier();
                // absolute index of the first sample of window chip x: out_abs0 + (wa + x - 9) * 32
                const long long A0 = a.out_abs0 + ((long long)wa - 9) * SPC;
                const long long Alo = A0 + ((long long)ga - 1) * SPC, Ahi = A0 + ((long long)gb + 16) * SPC;
                const bool inside = wide && Alo >= a.src_abs0 && Ahi <= a.src_abs1;      // (uniform) every sample present, 16-byte aligned
                const int np = nrow * 16;
                constexpr int MAXR = (32 + RS_XE) * 16 / AM_WAVE;                    // rounds of 64 pieces: 9
                if (inside) {
                    unsigned long long gbase = reinterpret_cast<unsigned long long>(iq2 + (Alo - a.src_abs0));
                    gbase = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(gbase >> 32)) << 32) |
                            (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)gbase);
                    float4 v[MAXR];
                    // (straight-line: a lane without a piece loads piece 0 again)
#pragma unroll
                    for (int r = 0; r < MAXR; ++r) {
                        const int p = lane + AM_WAVE * r;
                        const int pc = p < np ? p : 0;
                        const unsigned off = ((unsigned)RC[pc >> 4] - (unsigned)ga) * (unsigned)(SPC * 8) + (unsigned)(pc & 15) * 16u;
                        v[r] = fes_gload16_cached_at(gbase, off);
                    }
#pragma unroll
                    for (int r = 0; r < MAXR; ++r) {
                        const int p = lane + AM_WAVE * r;
                        if (p < np) {
                            const float r0 = v[r].x * v[r].x, i0 = v[r].y * v[r].y, r1 = v[r].z * v[r].z, i1 = v[r].w * v[r].w;
                            float2 mm;
                            mm.x = r0 + i0;                               // a1: fl(fl(I*I) + fl(Q*Q))
                            mm.y = r1 + i1;
                            const int s = p >> 4;
                            float *row = (s < cnt) ? XMw + s * RS_XS : XEw + (s - cnt) * RS_XS;
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
                        float *row = (s < cnt) ? XMw + s * RS_XS : XEw + (s - cnt) * RS_XS;
                        *reinterpret_cast<float2 *>(row + 2 * (p & 15)) = mm;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                // positions beyond the end of the stream read as zero (the preamble view pads with zeros)
                const long long leftn = a.src_abs1 - (A0 + (long long)ci * SPC);
                const int nin = leftn >= SPC ? SPC : (leftn <= 0 ? 0 : (int)leftn);
                if (PMF) {
                    // lane j: prefix sums of the own chip, left->right; lane j + 32: suffix sums of the chip before (its row reversed, so
                    // both run the same chain); bb[i] = fl((suf[i + 1] + pre[i]) s1), the chip's last sample: pre alone
                    float c[SPC];
                    if (live) {
                        const float *rowp = half ? (start ? XEw + srank * RS_XS : XMw + (j - 1) * RS_XS) : XMw + j * RS_XS;
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
#pragma unroll
                    for (int k = 0; k < SPC / 4; ++k) {
                        float o4[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int i = 4 * k + q;
                            float tt = c[i];
                            if (i < SPC - 1) tt = __shfl(c[(i < SPC - 1) ? SPC - 2 - i : 0], lane ^ 32, AM_WAVE) + c[i];   // suf[i + 1] + pre[i]
                            o4[q] = (i >= nin) ? 0.0f : tt * a.s1;
                            rmx = fmaxf(rmx, o4[q]);
                        }
                        if (live && half == 0) { float4 o; o.x = o4[0]; o.y = o4[1]; o.z = o4[2]; o.w = o4[3]; own[k] = o; }
#if defined(__HIP_DEVICE_COMPILE__)
                        __builtin_amdgcn_sched_barrier(0);                // (four positions at a time: hoisted, the 31 exchanges cost 31 registers more)
#endif
                    }
                    if (live && half == 0) L.RMAX[b0 + (uint32_t)j] = rmx;
                } else if (live && half == 0) {
                    // no filter: bb is |.|^2 itself; only the end of the stream needs a hand
                    float *own = XMw + j * RS_XS;
                    for (int i = nin; i < SPC; ++i) own[i] = 0.0f;
                    float rmx = 0.0f;
                    for (int i = 0; i < SPC; ++i) rmx = fmaxf(rmx, own[i]);
                    L.RMAX[b0 + (uint32_t)j] = rmx;
                }
                }
            }
            RS_STAMP(3);                                           // wave 0's rows
            // ---- the group's candidates: position and row of the own chip ---------------------------------------------------------
            for (uint32_t x = ga + (uint32_t)tid; x < gb; x += RS_NT) {
                uint32_t word = L.W[x];
                if (word) {
                    uint32_t idx = (uint32_t)L.P[x] - pa;
                    const uint32_t r0 = rs_rank(L, x) - r_base, p0 = (wa + x) * 32u - 288u;
                    while (word) {
                        const int b = __ffs((int)word) - 1;
                        L.POSL[idx] = p0 + (uint32_t)b;
                        L.CR0[idx] = (uint16_t)r0;
                        ++idx;
                        word &= word - 1u;
                    }
                }
            }
            L.LATE[tid] = 0u;
            __syncthreads();
            RS_STAMP(4);                                           // the other waves' rows, the candidate list

            // ---- late-peak decisions, once per position the group's candidates reach (am_k_refine_late, from LDS rows) -------------
            const uint32_t i = (uint32_t)tid;
            const bool live = i < nc;
            const uint32_t jpos = L.POSL[live ? i : 0u];
            uint32_t lo = jpos, d = 0;
            if (live) {
                if (i) { const uint32_t pv = L.POSL[i - 1u] + (uint32_t)SPC; lo = pv > jpos ? pv : jpos; }
                d = jpos + (uint32_t)SPC - lo;                            // >= 1: positions ascend strictly
            }
            {
                uint32_t incl = d;
                for (int o = 1; o < AM_WAVE; o <<= 1) {
                    const uint32_t up = (uint32_t)__shfl_up((int)incl, o, AM_WAVE);
                    if (lane >= o) incl += up;
                }
                if (lane == AM_WAVE - 1) L.ws[wv] = incl;
                __syncthreads();
                uint32_t off = incl - d;
                for (int k = 0; k < wv; ++k) off += L.ws[k];
                if (live) { L.coff[i] = off; L.clo[i] = lo; }
                if (i == nc - 1u) L.coff[nc] = off + d;
            }
            __syncthreads();
            const uint32_t kend = L.coff[nc];
            // V bounds every sample the group's positions can see: the largest bb of the front-end workgroups whose segments they span
            float vb = 0.0f;
            {
                uint32_t v0 = L.clo[0] / a.vspan, v1 = (L.clo[nc - 1u] + 13u * (uint32_t)SPC + 1u) / a.vspan;
                v0 = v0 < a.nv ? v0 : a.nv - 1u;
                v1 = v1 < a.nv ? v1 : a.nv - 1u;
                for (uint32_t v = v0; v <= v1; ++v) vb = fmaxf(vb, a.wg_max[v]);
            }
            const double bound = (double)vb * 0x1p-36;                    // (+inf when a sample is not finite: nothing is decided by D)
            RS_STAMP(5);                                           // positions laid out
            for (uint32_t k = (uint32_t)tid; k < kend; k += RS_NT) {
                uint32_t l = 0, h = nc;                                   // last candidate with coff <= k
                while (h - l > 1) {
                    const uint32_t mid = (l + h) >> 1;
                    if (L.coff[mid] <= k) l = mid; else h = mid;
                }
                const uint32_t q = L.clo[l] + (k - L.coff[l]);
                const uint32_t qq = q + 288u, o = qq & 31u;
                const uint32_t row = (uint32_t)L.CR0[l] + ((qq >> 5) - ((L.POSL[l] + 288u) >> 5));
                const float *p = L.XM + row * RS_XS + o;
                const float x0 = p[0], x1 = p[RS_XS], x2 = p[2 * RS_XS], x3 = p[3 * RS_XS];
                const float x4 = p[7 * RS_XS], x5 = p[8 * RS_XS], x6 = p[9 * RS_XS], x7 = p[10 * RS_XS];
                double dd = (double)x1 - (double)x0;
                dd = dd + ((double)x3 - (double)x2);
                dd = dd + ((double)x5 - (double)x4);
                dd = dd + ((double)x7 - (double)x6);
                bool late;
                if (fabs(dd) > bound) late = dd > 0.0;
                else late = rs_energy(L.XM, row, o + 1u) > rs_energy(L.XM, row, o);   // (rare: exact ties, non-finite samples)
                if (late) atomicOr(&L.LATE[k >> 5], 1u << (k & 31u));
            }
            __syncthreads();
            RS_STAMP(6);                                           // late decisions
            // ---- one lane per candidate ---------------------------------------------------------------------------------------------
            if (live) {
                int how_late = 0;
                {
                    const uint32_t kb = L.coff[i] - (lo - jpos);          // decision of position jpos (never below bit 0: see am_k_refine_late)
                    bool rising = true;
                    for (int k = 0; k < SPC && rising; ++k) {
                        const uint32_t kk = kb + (uint32_t)k;
                        if ((L.LATE[kk >> 5] >> (kk & 31u)) & 1u) how_late++; else rising = false;
                    }
                }
                const uint32_t e = jpos + (uint32_t)how_late;
                const float av = (e >= a.end_j) ? 0.0f : a.avg_sparse[e];   // beyond the end of the stream: 0
                const uint32_t ee = e + 288u;
                const int o = (int)(ee & 31u);
                const uint32_t rowe = (uint32_t)L.CR0[i] + ((ee >> 5) - ((jpos + 288u) >> 5));
                const float *pe = L.XM + rowe * RS_XS;
                const float p0 = pe[o], p1 = pe[2 * RS_XS + o], p2 = pe[7 * RS_XS + o], p3 = pe[9 * RS_XS + o];
                float ps = p0 + p1;                                       // quiet zones (preamble_impl.cc:198-209)
                ps = ps + p2;
                ps = ps + p3;
                const float avgpeak = (float)((double)ps / 4.0);
                const float sthr = av + (avgpeak - av) / a.thr_lin;
                // with e at offset o of chip c the zones [e + 96, e + 192] and [e + 320, e + 480] are: chip c + 3 from o on, chips c + 4,
                // c + 5 whole, chip c + 6 up to o; chip c + 10 from o on, chips c + 11 .. c + 14 whole, chip c + 15 up to o
                const float *mx = L.RMAX + rowe;
                bool hit = mx[4] > sthr || mx[5] > sthr || mx[11] > sthr || mx[12] > sthr || mx[13] > sthr || mx[14] > sthr;
                if (!hit) hit = rs_rows_partly_above(pe + 3 * RS_XS, pe + 10 * RS_XS, o, true, sthr);
                if (!hit) hit = rs_rows_partly_above(pe + 6 * RS_XS, pe + 15 * RS_XS, o, false, sthr);
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
                const uint32_t gi = cbase + pa + i;
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
            __syncthreads();                                              // (the rows and the lists are rewritten by the next group)
            RS_STAMP(7);                                           // candidates
        }
        cbase += (uint32_t)L.P[nown];
    }
#if defined(RS_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    if (a.prof && tid == 0) {
        rsp[11] = (long long)wall_clock64();
        for (int k = 0; k < 12; ++k) a.prof[(size_t)g * 12 + k] = rsp[k];
    }
#endif
}

hipError_t am_launch_refine_seg(const uint32_t *bits, const uint32_t *wg_cnt, const float *wg_max, uint32_t nwg, uint32_t words_per_wg,
                                uint32_t nwords, uint32_t Mcap, uint32_t lag, uint32_t wbits, uint32_t vspan, uint32_t nv,
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
    a.prof = nullptr;
#if defined(RS_PROFILE)
    if (hipMalloc(reinterpret_cast<void **>(&a.prof), (size_t)nwg * 12 * sizeof(long long)) != hipSuccess) a.prof = nullptr;
#endif
    if (rows.use_pmf) hipLaunchKernelGGL(am_k_refine_seg<true>, dim3(nwg), dim3(RS_NT), 0, s, a);
    else hipLaunchKernelGGL(am_k_refine_seg<false>, dim3(nwg), dim3(RS_NT), 0, s, a);
    const hipError_t lrc = hipGetLastError();
#if defined(RS_PROFILE)
    if (a.prof) {
        // blocking; prints where the workgroups' time went (10 ns ticks -> us) and the launch's timeline -- never in the default build
        std::vector<long long> h((size_t)nwg * 12);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), a.prof, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        (void)hipFree(a.prof);
        static const char *names[8] = {"counts+words", "window tables", "group choice", "rows (wave 0)", "other waves+list", "layout", "late decisions", "candidates"};
        double acc[12] = {};
        long long first = h[10], last = 0;
        std::vector<double> st(nwg), en(nwg), du(nwg);
        for (uint32_t b = 0; b < nwg; ++b) { first = std::min(first, h[(size_t)b * 12 + 10]); last = std::max(last, h[(size_t)b * 12 + 11]); }
        for (uint32_t b = 0; b < nwg; ++b) {
            for (int k = 0; k < 10; ++k) acc[k] += (double)h[(size_t)b * 12 + k];
            st[b] = (double)(h[(size_t)b * 12 + 10] - first) * 0.01; en[b] = (double)(h[(size_t)b * 12 + 11] - first) * 0.01; du[b] = en[b] - st[b];
        }
        fprintf(stderr, "rseg: %u workgroups, span %.1f us, %.2f groups and %.0f candidates per workgroup; us per workgroup:", nwg, (double)(last - first) * 0.01,
                acc[8] / nwg, acc[9] / nwg);
        for (int k = 0; k < 8; ++k) fprintf(stderr, " %s %.2f", names[k], acc[k] / nwg * 0.01);
        fprintf(stderr, "\n");
        auto pct = [&](std::vector<double> v, const char *name) {
            std::sort(v.begin(), v.end());
            const size_t n = v.size();
            fprintf(stderr, "rseg timeline %-10s us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f\n", name, v[0], v[n / 10], v[n / 2], v[(size_t)(n * 0.9)], v[n - 1]);
        };
        pct(st, "start"); pct(en, "end"); pct(du, "duration");
    }
#endif
    return lrc;
}
