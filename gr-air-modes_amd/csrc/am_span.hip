// am_span.hip -- barrier-free fused front end + detection + refinement for gfx950.
//
// One WAVE (a 64-thread workgroup, no __syncthreads anywhere) owns a contiguous span of
// 48-chip blocks of the stream and walks it block by block; lane l < 48 owns chip l of the
// current block and keeps its samples in registers.  Waves never wait for each other, so the
// HBM loads, LDS traffic and VALU work of the ~9 waves resident on a CU overlap freely.
//
// Per block b (software pipelined, everything in the canonical order of DESIGN.md section 3):
//   (i)   |iq|^2 of block b+1 (IQ was prefetched while block b-1 was refined) -> LDS ring slot,
//         pulse matched filter in registers (previous chip from the neighbour lane's row in LDS,
//         the block's first chip from a one-chip carry), bb(b+1) -> ring slot (+ a 17-chip
//         mirror so that block b's look-ahead is contiguous) and -> HBM (coalesced, from LDS)
//   (ii)  chip totals of block b, their sequential prefix/suffix inside the block, reference
//         level avg[n] in registers -- the window's older half comes from the SAME lane's
//         suffix sums of block b-1, carried in registers -- and the first-stage preamble test
//         (a6) against wide LDS loads of the chips 2, 7 and 9 ahead
//   (iii) IQ of block b+2 is requested (lands while (iv) runs)
//   (iv)  refinement of block b's candidates (a7, a8): late(q) = E(q+1) > E(q) once per
//         reachable position (one lane per position, E shared with the neighbour lane by
//         shuffle), shifts = trailing late bits, avg at the shifted start fetched from the
//         owning lane by shuffle, quiet zones from the ring -> candidate records
//
// LDS per wave: (2*48 + 17) chip rows + small bitmaps = ~17.5 KB -> 9 waves per CU.
// Reference: python/rx_path.py:38-54, lib/preamble_impl.cc:90-98,172-209.
#include "am_internal.h"

#include <stdlib.h>

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

// Lanes of one wave exchange data through LDS without a workgroup barrier: the hardware executes
// a wave's LDS instructions in order, so only the COMPILER must be kept from moving a lane's
// load above another lane's (same-instruction) store.  (tests/emu maps this to a wave rendezvous.)
#define AM_WAVE_SYNC()                                               \
    do {                                                             \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       \
        __builtin_amdgcn_wave_barrier();                             \
    } while (0)

#define SP_BLK AM_CHIPS_AVG          /* chips per block = lanes that own a chip */
#define SP_LOOK 17                   /* chips of look-ahead the refinement needs */
#define SP_ROWS (2 * SP_BLK + SP_LOOK)

struct am_span_args {
    const float *iq;
    long long src_abs0, src_abs1;   // absolute range of samples present in iq
    long long out_abs0;             // absolute index of bb[0] (multiple of 48*spc)
    long long out_n;                // outputs wanted
    float *bb;                      // dense pulse-matched power (read by burst extraction)
    uint32_t j0, j1;                // positions (array coordinates) whose preamble test is wanted
    uint32_t *seg_pos;              // per span: blocks_per_span * L candidate slots ...
    uint32_t *seg_e;
    float *seg_inavg;
    uint8_t *seg_valid;
    uint32_t *blk_cnt;              // candidates per span
    int nblocks;                    // blocks in [out_abs0, out_abs0 + out_n)
    int blocks_per_span;
    int nspans;
    int use_pmf;
    float s1, sL, thr_lin;
};

// chip row stride in LDS (floats): multiple of 4 (16-byte rows), lanes 16 apart on distinct banks
template <int SPC> struct sp_cfg { static constexpr int CS = SPC + 4; };
template <> struct sp_cfg<20> { static constexpr int CS = 20; };
template <> struct sp_cfg<10> { static constexpr int CS = 12; };

template <int N>
__device__ __forceinline__ void sp_row_load(const float *row, float (&v)[N])
{
    if constexpr (N % 4 == 0) {
#pragma unroll
        for (int k = 0; k < N / 4; ++k) {
            const float4 t = *reinterpret_cast<const float4 *>(row + 4 * k);
            v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < N / 2; ++k) {
            const float2 t = *reinterpret_cast<const float2 *>(row + 2 * k);
            v[2 * k] = t.x; v[2 * k + 1] = t.y;
        }
    }
}

template <int N>
__device__ __forceinline__ void sp_row_store(float *row, const float (&v)[N])
{
    if constexpr (N % 4 == 0) {
#pragma unroll
        for (int k = 0; k < N / 4; ++k) {
            float4 t;
            t.x = v[4 * k]; t.y = v[4 * k + 1]; t.z = v[4 * k + 2]; t.w = v[4 * k + 3];
            *reinterpret_cast<float4 *>(row + 4 * k) = t;
        }
    } else {
#pragma unroll
        for (int k = 0; k < N / 2; ++k) {
            float2 t;
            t.x = v[2 * k]; t.y = v[2 * k + 1];
            *reinterpret_cast<float2 *>(row + 2 * k) = t;
        }
    }
}

template <int SPC>
__global__ void __launch_bounds__(AM_WAVE, 2) am_k_span(am_span_args a)
{
    static_assert(SPC % 2 == 0, "even samples per chip");
    constexpr int CS = sp_cfg<SPC>::CS;
    constexpr int L = SP_BLK * SPC;                       // samples per block
    constexpr int NPAIR = L / 2;                          // 16-byte IQ pairs per block
    constexpr int NLD = (NPAIR + AM_WAVE - 1) / AM_WAVE;  // pairs per lane
    constexpr int NW = L / 32;                            // bitmap words per block
    static_assert(L % 32 == 0, "block is whole bitmap words");

    HIP_DYNAMIC_SHARED(float, smem);
    float *RING = smem;                                   // SP_ROWS rows of CS floats
    float *CARRY = RING + SP_ROWS * CS;                   // |iq|^2 of the last chip of the previous block
    float *TOTL = CARRY + SPC;                            // chip totals / their prefix / suffix   [3*48]
    uint32_t *BMW = reinterpret_cast<uint32_t *>(TOTL + 3 * SP_BLK);   // candidate bitmap [NW + 2]
    uint32_t *NL = BMW + (NW + 2);                        // positions whose energy is needed
    uint32_t *NLP = NL + (NW + 2);                        // exclusive prefix of their popcounts [NW + 3]
    uint32_t *LB = NLP + (NW + 3);                        // late flags

    const int lane = threadIdx.x;
    const bool own = lane < SP_BLK;                       // lanes 48..63 own no chip
    const int span = blockIdx.x;
    if (span >= a.nspans) return;
    const int b_lo = span * a.blocks_per_span;
    int b_hi = b_lo + a.blocks_per_span;
    if (b_hi > a.nblocks) b_hi = a.nblocks;
    if (b_lo >= b_hi) {
        if (lane == 0) a.blk_cnt[span] = 0;
        return;
    }
    const float2 *iq2 = reinterpret_cast<const float2 *>(a.iq);
    const float4 *iq4 = reinterpret_cast<const float4 *>(a.iq);
    const long long len = a.src_abs1 - a.src_abs0;
    // 16-byte loads need the block starts to sit on even sample offsets of a 16-byte aligned buffer
    const bool vec = (((a.out_abs0 - a.src_abs0) & 1) == 0) && ((reinterpret_cast<uintptr_t>(a.iq) & 15u) == 0) &&
                     len >= 2;
    const long long npf = len >> 1;

    // ---- helpers ------------------------------------------------------------------------------
    // request the IQ of block b (relative to out_abs0): branch-free clamped loads, zeroed later
    auto iq_request = [&](int b, float4 (&v)[NLD]) __attribute__((always_inline)) {
        const long long n0 = a.out_abs0 + (long long)b * L;        // absolute first sample
        const long long rel0 = n0 - a.src_abs0;
        if (vec) {
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                long long pi = (rel0 >> 1) + (lane + k * AM_WAVE);
                pi = pi < 0 ? 0 : (pi >= npf ? npf - 1 : pi);
                v[k] = iq4[pi];
            }
        } else {
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                long long i0 = rel0 + 2 * (lane + k * AM_WAVE), i1 = i0 + 1;
                i0 = i0 < 0 ? 0 : (i0 >= len ? len - 1 : i0);
                i1 = i1 < 0 ? 0 : (i1 >= len ? len - 1 : i1);
                float2 t0, t1;
                t0.x = 0.0f; t0.y = 0.0f; t1 = t0;
                if (len > 0) { t0 = iq2[i0]; t1 = iq2[i1]; }
                v[k].x = t0.x; v[k].y = t0.y; v[k].z = t1.x; v[k].w = t1.y;
            }
        }
    };
    // |.|^2 (a1) of the requested block into ring rows [row0, row0+48)
    auto stage_m = [&](int b, const float4 (&v)[NLD], int row0) __attribute__((always_inline)) {
        const long long n0 = a.out_abs0 + (long long)b * L;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int p = lane + k * AM_WAVE;
            if (p < NPAIR) {
                const long long n = n0 + 2 * p;
                const bool ok0 = n >= a.src_abs0 && n < a.src_abs1;
                const bool ok1 = n + 1 >= a.src_abs0 && n + 1 < a.src_abs1;
                float x0 = v[k].x, y0 = v[k].y;
                if (vec && ok0 && !ok1) {                              // odd stream length: last sample
                    const float2 t = iq2[n - a.src_abs0];
                    x0 = t.x; y0 = t.y;
                }
                const float r0 = x0 * x0, i0 = y0 * y0;
                const float r1 = v[k].z * v[k].z, i1 = v[k].w * v[k].w;
                float2 m;
                m.x = ok0 ? (r0 + i0) : 0.0f;                          // fl(fl(I*I) + fl(Q*Q))
                m.y = ok1 ? (r1 + i1) : 0.0f;
                const int s = 2 * p;                                   // sample inside the block
                const int c = s / SPC, o = s - c * SPC;
                *reinterpret_cast<float2 *>(&RING[(row0 + c) * CS + o]) = m;
            }
        }
    };
    // pulse matched filter (a3) for this lane's chip of the block staged in rows [row0, ..):
    // bbn = bb of the chip; also leaves bb in the ring (in place) and in the mirror rows
    auto pmf_block = [&](int b, int row0, float (&bbn)[SPC]) __attribute__((always_inline)) {
        float mc[SPC];
#pragma unroll
        for (int i = 0; i < SPC; ++i) mc[i] = 0.0f;
        AM_WAVE_SYNC();                                                 // staged |iq|^2 visible to all lanes
        if (own) sp_row_load<SPC>(&RING[(row0 + lane) * CS], mc);
        if (a.use_pmf) {
            float mp[SPC];
#pragma unroll
            for (int i = 0; i < SPC; ++i) mp[i] = 0.0f;
            if (own) {
                if (lane == 0) sp_row_load<SPC>(CARRY, mp);
                else sp_row_load<SPC>(&RING[(row0 + lane - 1) * CS], mp);
            }
            AM_WAVE_SYNC();                                             // every lane has read its neighbour's row
            if (lane == SP_BLK - 1) sp_row_store<SPC>(CARRY, mc);       // for the next block's first chip
            float acc = 0.0f;
#pragma unroll
            for (int i = SPC - 1; i >= 0; --i) { acc = acc + mp[i]; mp[i] = acc; }   // suffix sums of the previous chip
            acc = 0.0f;
#pragma unroll
            for (int i = 0; i < SPC; ++i) {
                acc = acc + mc[i];
                const float s = (i == SPC - 1) ? acc : (mp[(i + 1 < SPC) ? i + 1 : i] + acc);
                bbn[i] = s * a.s1;
            }
        } else {
#pragma unroll
            for (int i = 0; i < SPC; ++i) bbn[i] = mc[i];
        }
        // samples beyond the end of the stream read as zero (the preamble view pads with zeros)
        const long long c_abs = a.out_abs0 + (long long)b * L + (long long)lane * SPC;
#pragma unroll
        for (int i = 0; i < SPC; ++i) if (c_abs + i >= a.src_abs1) bbn[i] = 0.0f;
        AM_WAVE_SYNC();
        if (own) {
            sp_row_store<SPC>(&RING[(row0 + lane) * CS], bbn);
            if (row0 == 0 && lane < SP_LOOK) sp_row_store<SPC>(&RING[(2 * SP_BLK + lane) * CS], bbn);
        }
        AM_WAVE_SYNC();                                                 // bb rows visible to all lanes
    };
    // dense bb of block b: ring rows -> HBM, 16 bytes per lane, coalesced
    auto bb_store = [&](int b, int row0) __attribute__((always_inline)) {
        if (!a.bb) return;
        const long long o0 = (long long)b * L;
        constexpr int NQ = L / 4;
        for (int q4 = lane; q4 < NQ; q4 += AM_WAVE) {
            const int s = 4 * q4;
            const int c = s / SPC, o = s - c * SPC;
            const float *src = &RING[(row0 + c) * CS + o];
            const long long o_abs = o0 + s;
            if constexpr (SPC % 4 == 0) {
                const float4 t = *reinterpret_cast<const float4 *>(src);
                if (o_abs + 3 < a.out_n) *reinterpret_cast<float4 *>(&a.bb[o_abs]) = t;
                else {
                    if (o_abs < a.out_n) a.bb[o_abs] = t.x;
                    if (o_abs + 1 < a.out_n) a.bb[o_abs + 1] = t.y;
                    if (o_abs + 2 < a.out_n) a.bb[o_abs + 2] = t.z;
                }
            } else {
                // chips of 10 samples: a group of 4 may straddle two rows
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int sk = s + k;
                    const int ck = sk / SPC, ok = sk - ck * SPC;
                    if (o_abs + k < a.out_n) a.bb[o_abs + k] = RING[(row0 + ck) * CS + ok];
                }
            }
        }
    };
    // sample s (>= 0, relative to the start of the current block, up to 48+17 chips) in the ring
    auto ring_at = [&](int cur_row0, int s) __attribute__((always_inline)) -> float {
        const int c = s / SPC, o = s - c * SPC;
        return RING[(cur_row0 + c) * CS + o];
    };
    // 4-pulse energy at sample s of the current block (preamble_impl.cc:91-98: chips 0,2,7,9,
    // chip-major, double accumulation)
    auto energy = [&](int cur_row0, int s) __attribute__((always_inline)) {
        double e = 0.0;
        constexpr int chips[4] = {0, 2, 7, 9};
        const int c0 = s / SPC, o0 = s - c0 * SPC;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const float *r0 = &RING[(cur_row0 + c0 + chips[cc]) * CS];
            float t[SPC];
#pragma unroll
            for (int i = 0; i < SPC; ++i) {
                const int oi = o0 + i;
                t[i] = (oi < SPC) ? r0[oi] : r0[CS + oi - SPC];
            }
#pragma unroll
            for (int i = 0; i < SPC; ++i) e += (double)t[i];
        }
        return e;
    };
    // wave-exclusive scan of a per-lane count; *total = wave sum
    auto wave_excl = [&](uint32_t cnt, uint32_t *total) __attribute__((always_inline)) {
        uint32_t incl = cnt;
        for (int d = 1; d < AM_WAVE; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d, AM_WAVE);
            if (lane >= d) incl += up;
        }
        *total = (uint32_t)__shfl((int)incl, AM_WAVE - 1, AM_WAVE);
        return incl - cnt;
    };

    // ---- prologue: |iq|^2 of the chip before block b_lo-1 (carry for its first chip) -------------
    {
        const long long n0 = a.out_abs0 + (long long)(b_lo - 1) * L - SPC;
        float m = 0.0f;
        if (lane < SPC) {
            const long long n = n0 + lane;
            if (n >= a.src_abs0 && n < a.src_abs1) {
                const float2 t = iq2[n - a.src_abs0];
                const float rr = t.x * t.x, ii = t.y * t.y;
                m = rr + ii;
            }
            CARRY[lane] = m;
        }
        AM_WAVE_SYNC();
    }

    float scp[SPC];                                        // in-chip suffix sums of the same lane's chip, previous block
#pragma unroll
    for (int i = 0; i < SPC; ++i) scp[i] = 0.0f;
    float st_prev = 0.0f;                                  // ST of that chip inside the previous block
    float sl_prev = 0.0f;                                  // RTOT + ST of the NEXT chip of the previous block
    uint32_t seg_fill = 0;                                 // candidates emitted so far by this span
    const size_t seg0 = (size_t)span * (size_t)a.blocks_per_span * L;

    float4 iqv[NLD];
    iq_request(b_lo - 1, iqv);

    for (int b = b_lo - 2; b < b_hi; ++b) {
        const int cur_row0 = (b & 1) * SP_BLK;             // ring rows of block b
        const int nxt_row0 = ((b + 1) & 1) * SP_BLK;       // ring rows of block b+1
        // ---- (i) bb of block b+1 ------------------------------------------------------------------
        float bbn[SPC];
        stage_m(b + 1, iqv, nxt_row0);
        pmf_block(b + 1, nxt_row0, bbn);
        if (b + 1 >= b_lo && b + 1 < b_hi) bb_store(b + 1, nxt_row0);

        if (b >= b_lo - 1) {
            // ---- (ii) chip totals, block prefix/suffix (canonical level 2) -----------------------------
            float bbc[SPC];                                // bb of this lane's chip of block b (from the ring)
#pragma unroll
            for (int i = 0; i < SPC; ++i) bbc[i] = 0.0f;
            if (own) sp_row_load<SPC>(&RING[(cur_row0 + lane) * CS], bbc);
            float tot = 0.0f, rtot = 0.0f;
#pragma unroll
            for (int i = 0; i < SPC; ++i) tot = tot + bbc[i];
#pragma unroll
            for (int i = SPC - 1; i >= 0; --i) rtot = rtot + bbc[i];
            if (own) TOTL[lane] = tot;
            AM_WAVE_SYNC();
            if (lane < 2) {
                // lane 0: exclusive prefix left->right, lane 1: exclusive suffix right->left
                // (rolled: a 48-register batch here would push the whole kernel into spilling)
                float acc = 0.0f;
                float *dst = TOTL + SP_BLK * (1 + lane);
#pragma unroll 4
                for (int j = 0; j < SP_BLK; ++j) {
                    const int jj = lane == 0 ? j : SP_BLK - 1 - j;
                    const float v = TOTL[jj];
                    dst[jj] = acc;
                    acc = acc + v;
                }
            }
            AM_WAVE_SYNC();
            const float pt = own ? TOTL[SP_BLK + lane] : 0.0f;
            const float st = own ? TOTL[2 * SP_BLK + lane] : 0.0f;

            if (b >= b_lo) {
                // ---- reference level (a4) ---------------------------------------------------------------
                float avgv[SPC];
                {
                    float acc = 0.0f;
#pragma unroll
                    for (int i = 0; i < SPC; ++i) {
                        acc = acc + bbc[i];
                        const float PRE = pt + acc;
                        float s;
                        if (i == SPC - 1) s = (lane == SP_BLK - 1) ? PRE : (sl_prev + PRE);
                        else s = (scp[i + 1] + st_prev) + PRE;
                        avgv[i] = s * a.sL;
                    }
                }
                // ---- first-stage preamble test (a6), branch-free --------------------------------------------
                const uint32_t jb0 = (uint32_t)((long long)b * L);          // array coordinate of the block
                uint32_t cm = 0;                                             // SPC candidate bits of this chip
                {
                    const int lrow = own ? lane : 0;                          // idle lanes stay inside the ring
                    const float nxt = RING[(cur_row0 + lrow + 1) * CS];
#pragma unroll
                    for (int i = 0; i < SPC; ++i) {
                        const float x = bbc[i];
                        const float thr = avgv[i] * a.thr_lin;                 // preamble_impl.cc:173
                        const float nx = (i + 1 < SPC) ? bbc[(i + 1 < SPC) ? i + 1 : i] : nxt;
                        const uint32_t j = jb0 + (uint32_t)(lane * SPC + i);
                        const bool c = own && (x > thr) && !(nx > x) && j >= a.j0 && j < a.j1;   // :174, :175
                        cm |= (c ? 1u : 0u) << i;
                    }
                    constexpr int ahead[3] = {2, 7, 9};
#pragma unroll
                    for (int o = 0; o < 3; ++o) {
                        float t[SPC];
                        sp_row_load<SPC>(&RING[(cur_row0 + lrow + ahead[o]) * CS], t);
                        uint32_t below = 0;
#pragma unroll
                        for (int i = 0; i < SPC; ++i) below |= ((t[i] < avgv[i] * a.thr_lin) ? 1u : 0u) << i;   // :177-179
                        cm &= ~below;
                    }
                }
                // candidate bitmap of the block (positions = bits), via LDS words
                if (lane < NW + 2) { BMW[lane] = 0u; LB[lane] = 0u; }
                AM_WAVE_SYNC();
                if (cm) {
                    const int bit0 = lane * SPC;
                    const unsigned long long wide = (unsigned long long)cm << (bit0 & 31);
                    atomicOr(&BMW[bit0 >> 5], (uint32_t)wide);
                    if ((uint32_t)(wide >> 32)) atomicOr(&BMW[(bit0 >> 5) + 1], (uint32_t)(wide >> 32));
                }
                // ---- carry this block's in-chip suffix sums and block suffix to the next block (bbc dies here)
                const float st0 = __shfl(st, 0, AM_WAVE);                   // ST of the block's first chip
                const float sl1 = __shfl(rtot + st, 1, AM_WAVE);            // RTOT + ST of its second chip
                {
                    float acc = 0.0f;
#pragma unroll
                    for (int i = SPC - 1; i >= 0; --i) { acc = acc + bbc[i]; scp[i] = acc; }
                    st_prev = st;
                    sl_prev = __shfl_down(rtot + st, 1, AM_WAVE);
                }
                AM_WAVE_SYNC();
                // ---- (iii) IQ of block b+2 lands while the refinement runs --------------------------------
                if (b + 2 <= b_hi) iq_request(b + 2, iqv);

                // ordered candidate list of this block
                uint32_t ncand = 0;
                const uint32_t myw = (lane < NW) ? BMW[lane] : 0u;
                const uint32_t coff = wave_excl((uint32_t)__popcll((unsigned long long)myw), &ncand);
                if (ncand) {
                    {
                        uint32_t wbits = myw, o = seg_fill + coff;
                        while (wbits) {
                            const int bit = __ffsll((long long)wbits) - 1;
                            a.seg_pos[seg0 + o++] = jb0 + (uint32_t)(lane * 32 + bit);
                            wbits &= wbits - 1u;
                        }
                    }
                    // ---- (iv) refinement ---------------------------------------------------------------------
                    // energies are needed at positions [p, p+SPC] of every candidate p; late flags at
                    // [p, p+SPC-1].  NL = candidate bitmap dilated by SPC (towards higher positions).
                    uint32_t nneed = 0;
                    {
                        uint32_t d32 = 0;
                        if (lane < NW + 2) {
                            const unsigned long long hi = (lane < NW) ? BMW[lane] : 0u;
                            const unsigned long long lo = (lane > 0 && lane - 1 < NW) ? BMW[lane - 1] : 0u;
                            unsigned long long v = (hi << 32) | lo;
                            int covered = 1;
#pragma unroll
                            for (int stepw = 1; stepw * 2 <= SPC; stepw *= 2) { v |= v << stepw; covered = stepw * 2; }
                            if (covered < SPC) v |= v << (SPC - covered);      // shifts 0 .. SPC-1
                            v |= v << 1;                                       // ... and SPC: E(p + SPC)
                            d32 = (uint32_t)(v >> 32);
                            NL[lane] = d32;
                        }
                        const uint32_t off = wave_excl((uint32_t)__popcll((unsigned long long)d32), &nneed);
                        if (lane < NW + 2) NLP[lane] = off;
                        if (lane == 0) NLP[NW + 2] = nneed;
                        AM_WAVE_SYNC();
                        // late flags: rounds of 63 consecutive needed positions (+1 overlap lane)
                        for (uint32_t k0 = 0; k0 < nneed; k0 += AM_WAVE - 1) {
                            const uint32_t k = k0 + (uint32_t)lane;
                            const bool act = k < nneed;
                            int q = 0;
                            if (act) {
                                int lo = 0, hi = NW + 2;
                                while (hi - lo > 1) {
                                    const int mid = (lo + hi) >> 1;
                                    if (NLP[mid] <= k) lo = mid; else hi = mid;
                                }
                                uint32_t wbits = NL[lo];
                                for (uint32_t r = k - NLP[lo]; r > 0; --r) wbits &= wbits - 1u;
                                q = lo * 32 + (__ffsll((long long)wbits) - 1);
                            }
                            const double e = act ? energy(cur_row0, q) : 0.0;
                            const double en = __shfl_down(e, 1, AM_WAVE);
                            const int qn = __shfl_down(q, 1, AM_WAVE);
                            // lane 63 of a round is the overlap lane (its flag comes from the next round)
                            if (act && lane < AM_WAVE - 1 && k + 1 < nneed && qn == q + 1 && en > e)
                                atomicOr(&LB[q >> 5], 1u << (q & 31));
                        }
                        AM_WAVE_SYNC();
                    }
                    // per candidate: shifts, reference level at the shifted start, quiet zones
                    for (uint32_t c0 = 0; c0 < ncand; c0 += AM_WAVE) {
                        const uint32_t ci = c0 + (uint32_t)lane;
                        const bool act = ci < ncand;
                        int p = 0, how_late = 0;
                        if (act) {
                            // ci-th set bit of the block bitmap
                            // (prefix of candidate popcounts is recomputed cheaply: NW <= 48 words)
                            uint32_t acc = 0;
                            int w = 0;
                            for (; w < NW; ++w) {
                                const uint32_t pc = (uint32_t)__popcll((unsigned long long)BMW[w]);
                                if (acc + pc > ci) break;
                                acc += pc;
                            }
                            uint32_t wbits = BMW[w];
                            for (uint32_t r = ci - acc; r > 0; --r) wbits &= wbits - 1u;
                            p = w * 32 + (__ffsll((long long)wbits) - 1);
                            const int w0 = p >> 5;
                            const unsigned long long win =
                                (((unsigned long long)LB[w0 + 1] << 32) | LB[w0]) >> (p & 31);
                            how_late = __ffsll((long long)~win) - 1;
                            if (how_late > SPC) how_late = SPC;
                        }
                        const int se = p + how_late;                          // shifted start, block-relative
                        const int ce = se / SPC, oe = se - ce * SPC;          // chip (0..48) and offset
                        // reference level at e: held by the lane that owns chip ce (all lanes take part)
                        float av = 0.0f;
#pragma unroll
                        for (int o = 0; o < SPC; ++o) {
                            const float v = __shfl(avgv[o], ce < SP_BLK ? ce : 0, AM_WAVE);
                            if (o == oe) av = v;
                        }
                        if (act) {
                            if (ce >= SP_BLK) {
                                // e fell into the first chip of the next block: its window's older half is
                                // chip 0 of THIS block; its own block prefix is empty
                                float pc = 0.0f, sc = 0.0f;
#pragma unroll
                                for (int i = 0; i < SPC; ++i) {
                                    const float v = RING[(cur_row0 + SP_BLK) * CS + i];
                                    pc = (i <= oe) ? (pc + v) : pc;
                                }
                                float ssum;
                                if (oe == SPC - 1) {
                                    ssum = sl1 + pc;
                                } else {
#pragma unroll
                                    for (int i = SPC - 1; i >= 0; --i) {
                                        const float v = RING[(cur_row0 + 0) * CS + i];
                                        sc = (i > oe) ? (sc + v) : sc;
                                    }
                                    ssum = (sc + st0) + pc;
                                }
                                av = ssum * a.sL;
                            }
                            const long long e_abs = a.out_abs0 + (long long)b * L + se;
                            if (e_abs >= a.src_abs1) av = 0.0f;
                            // quiet zones (preamble_impl.cc:198-209)
                            const float p0 = ring_at(cur_row0, se), p1 = ring_at(cur_row0, se + 2 * SPC);
                            const float p2 = ring_at(cur_row0, se + 7 * SPC), p3 = ring_at(cur_row0, se + 9 * SPC);
                            float ps = p0 + p1;
                            ps = ps + p2;
                            ps = ps + p3;
                            const float avgpeak = (float)((double)ps / 4.0);
                            const float sthr = av + (avgpeak - av) / a.thr_lin;
                            bool bad = false;
                            constexpr int N1 = 3 * SPC + 1, N2 = 5 * SPC + 1;
                            constexpr int CH = 16;
                            for (int o = 0; o < N1 && !bad; o += CH) {
                                float t[CH];
#pragma unroll
                                for (int k = 0; k < CH; ++k) t[k] = ring_at(cur_row0, se + 3 * SPC + ((o + k < N1) ? o + k : N1 - 1));
#pragma unroll
                                for (int k = 0; k < CH; ++k) bad = bad || (t[k] > sthr);
                            }
                            for (int o = 0; o < N2 && !bad; o += CH) {
                                float t[CH];
#pragma unroll
                                for (int k = 0; k < CH; ++k) t[k] = ring_at(cur_row0, se + 10 * SPC + ((o + k < N2) ? o + k : N2 - 1));
#pragma unroll
                                for (int k = 0; k < CH; ++k) bad = bad || (t[k] > sthr);
                            }
                            const size_t so = seg0 + seg_fill + ci;
                            a.seg_e[so] = jb0 + (uint32_t)se;
                            a.seg_inavg[so] = av;
                            a.seg_valid[so] = bad ? 0 : 1;
                        }
                    }
                    seg_fill += ncand;
                }
            } else {
                // warm-up block: only its suffix sums are needed
                float acc = 0.0f;
#pragma unroll
                for (int i = SPC - 1; i >= 0; --i) { acc = acc + bbc[i]; scp[i] = acc; }
                st_prev = st;
                sl_prev = __shfl_down(rtot + st, 1, AM_WAVE);
                if (b + 2 <= b_hi) iq_request(b + 2, iqv);
            }
        } else {
            iq_request(b + 2, iqv);
        }
        AM_WAVE_SYNC();                                    // block b's rows may be overwritten from here on
    }
    if (lane == 0) a.blk_cnt[span] = seg_fill;
}

// span geometry: blocks per span and number of spans for nblocks blocks
static void span_geometry(long long nblocks, int *bps_out, int *nspans_out)
{
    // enough spans to fill the chip several times over (9 waves per CU), spans long enough to
    // amortise the two warm-up blocks each span recomputes
    long long target = 256 * 9 * 2;
    if (const char *x = getenv("AIRMODES_SPAN_TARGET")) target = atoi(x) > 0 ? atoi(x) : target;
    long long bps = (nblocks + target - 1) / target;
    if (bps < 8) bps = 8;
    if (const char *x = getenv("AIRMODES_SPAN_BLOCKS")) bps = atoi(x) > 0 ? atoi(x) : bps;
    *bps_out = (int)bps;
    *nspans_out = (int)((nblocks + bps - 1) / bps);
}

template <int SPC>
static hipError_t span_launch(am_span_args a, hipStream_t s, unsigned *nseg, unsigned *seg_stride)
{
    constexpr int CS = sp_cfg<SPC>::CS, L = SP_BLK * SPC, NW = L / 32;
    const size_t lds = ((size_t)SP_ROWS * CS + SPC + 3 * SP_BLK + (NW + 2) * 3 + (NW + 3) + 8) * sizeof(float);
    a.nblocks = (int)((a.out_n + L - 1) / L);
    span_geometry(a.nblocks, &a.blocks_per_span, &a.nspans);
    *nseg = (unsigned)a.nspans;
    *seg_stride = (unsigned)(a.blocks_per_span * L);
    if (a.nspans == 0) return hipSuccess;
    hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void *>(&am_k_span<SPC>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc != hipSuccess) return rc;
    hipLaunchKernelGGL((am_k_span<SPC>), dim3((unsigned)a.nspans), dim3(AM_WAVE), lds, s, a);
    return hipGetLastError();
}

// number of candidate slots (= positions) the span kernel needs for out_n outputs; 0 = unsupported spc
size_t am_span_slots(int spc, long long out_n, unsigned *nseg_max)
{
    switch (spc) {
    case 8: case 10: case 16: case 20: case 32: break;
    default: return 0;
    }
    const long long L = (long long)SP_BLK * spc;
    const long long nblocks = (out_n + L - 1) / L;
    int bps = 0, nspans = 0;
    span_geometry(nblocks, &bps, &nspans);
    if (nseg_max) *nseg_max = (unsigned)nspans;
    return (size_t)nspans * (size_t)bps * (size_t)L;
}

hipError_t am_launch_span(int spc, const float *iq, long long src_abs0, long long src_abs1, long long out_abs0,
                          long long out_n, float *bb, uint32_t j0, uint32_t j1, int use_pmf, float s1, float sL,
                          float thr_lin, uint32_t *seg_pos, uint32_t *seg_e, float *seg_inavg, uint8_t *seg_valid,
                          uint32_t *blk_cnt, unsigned *nseg, unsigned *seg_stride, hipStream_t s)
{
    am_span_args a;
    a.iq = iq; a.src_abs0 = src_abs0; a.src_abs1 = src_abs1; a.out_abs0 = out_abs0; a.out_n = out_n;
    a.bb = bb; a.j0 = j0; a.j1 = j1; a.seg_pos = seg_pos; a.seg_e = seg_e; a.seg_inavg = seg_inavg;
    a.seg_valid = seg_valid; a.blk_cnt = blk_cnt; a.nblocks = 0; a.blocks_per_span = 0; a.nspans = 0;
    a.use_pmf = use_pmf; a.s1 = s1; a.sL = sL; a.thr_lin = thr_lin;
    switch (spc) {
    case 8: return span_launch<8>(a, s, nseg, seg_stride);
    case 10: return span_launch<10>(a, s, nseg, seg_stride);
    case 16: return span_launch<16>(a, s, nseg, seg_stride);
    case 20: return span_launch<20>(a, s, nseg, seg_stride);
    case 32: return span_launch<32>(a, s, nseg, seg_stride);
    default: return hipErrorInvalidValue;
    }
}
