// am_fe4.hip -- the streaming fused front end (am_fe3.hip) for rates below 64 Msps: several chips per lane.
//
// am_k_fe3 gives a lane one 32-sample chip.  At 20 Msps a chip has 10 samples, at 10 Msps 5, at 4 Msps 2, at 2 Msps 1: a lane takes a UNIT
// of G consecutive chips instead (R = G * spc samples, even: 30 at 20 and 10 Msps, 32 at 4 Msps, 24 at 2 Msps), a 48-chip block is 48 / G lanes, and
// everything else keeps am_k_fe3's shape -- persistent workgroups of two waves walking a contiguous segment in steps of
// 96 units (128 where 48 / G divides 64: then every lane of a wave owns a unit, 20, 10 and 2 Msps), |.|^2 staged straight into LDS ring rows, phase A (pulse-matched filter, chip totals, sequential in-block
// scans) FE4 lag units ahead of phase B (reference level + first-stage test), a candidate bitmap (R bits per unit) and
// sparse bb / reference-level runs around candidates as the only outputs.  What changes with G > 1:
//   * the in-chip prefix / suffix chains restart at every chip of the unit; the chip before a unit's first chip belongs
//     to lane - 1 (its last chip's suffix sums come over by DPP), the chip before any other chip is the lane's own;
//   * the sequential in-block scans of the chip totals hop from lane to lane with G additions per hop
//     (x <- ((x(lane-1) + f0) + f1) + ...), restarted where a block starts (lane % (48 / G) == 0); the ring keeps, per
//     UNIT, the prefix entering it from the left, the suffix entering it from the right and RTOT + ST of its first chip
//     -- the per-chip values phase B needs are re-formed from those and the unit's own bb row (G additions);
//   * the pulses 2, 7 and 9 chips ahead of a sample lie in other units at compile-time offsets: element-wise LDS reads.
// Same canonical order, same results as am_k_fe2 / the oracle; burst extraction recomputes from IQ.
//
// Reference: python/rx_path.py:35-54 (spc = rate / 2e6, |.|^2, moving averages), lib/preamble_impl.cc:172-179 (test).
#include "am_internal.h"

#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <vector>

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#include "am_fe_cmpx.h"

#define FE4_NW 2                          /* waves per workgroup                                         */
#define FE4_NT (AM_WAVE * FE4_NW)
#ifndef FE4_WG_PER_CU
#define FE4_WG_PER_CU 6
#endif

struct am_fe4_args {
    const float *iq;
    long long src_abs0, src_abs1;         // absolute range of samples present in iq
    long long out_abs0;                   // absolute index of array coordinate 0 (multiple of 48*spc)
    long long out_n;                      // array coordinates with data
    float *bb_sparse;                     // bb runs around candidates (array coordinates)
    float *avg_sparse;                    // reference-level runs around candidates
    uint32_t j0, j1;                      // positions whose preamble test is wanted
    uint32_t *bits;                       // [nsteps * US] candidate words: bit b of word w = position w*R + b - lag
    uint32_t *seg_cnt;                    // [nsteps * 2] candidates per (step, wave); wave w = words LU w .. LU w + LU - 1
    float *wg_max;                        // [grid] largest bb a workgroup formed (+inf if one was not finite)
    unsigned nsteps, steps_per_wg;
    int raw_lo, raw_hi, test_lo, test_hi; // steps loaded without guards / tested without a range mask
    int use_pmf;
    float s1, sL, thr_lin;
};

template <int SPC, int G>
struct fe4_cfg {
    static constexpr int R = SPC * G;                                  // samples per unit
    // ring row stride in floats: even (rows are read 8 bytes at a time), at least R + 2, and not a multiple of 32 (consecutive
    // lanes read consecutive rows: a stride of 32 floats would put every lane on the same banks)
    static constexpr int RS = ((R + 2) % 32 == 0) ? R + 4 : R + 2;
    static constexpr int LPB = AM_CHIPS_AVG / G;                       // lanes (units) per 48-chip block
    // lanes of a wave that own a unit: a whole number of blocks -- all 64 where 48 / G divides 64 (G = 3, 24: the 20 Msps and
    // 2 Msps kernels), 48 otherwise
    static constexpr int LU = (AM_WAVE % LPB == 0) ? AM_WAVE : AM_CHIPS_AVG;
    static constexpr int US = LU * FE4_NW;                             // units per step
    // waves per SIMD the registers are budgeted for: 3 (<= 168 VGPRs, 6 workgroups per CU; the 2 Msps kernel needs fewer and
    // gets 8).  (4 for the 20 Msps kernel -- 128 VGPRs, 7 workgroups per CU, one step fewer per workgroup -- measured
    // 0.058 against 0.0556 ms: the tighter register budget costs more than the step.)
    static constexpr int MINW = 3;
    static constexpr unsigned long long LUMASK = (LU == 64) ? ~0ull : ((1ull << (LU & 63)) - 1ull);
    static constexpr int LAGU = 1 + 8 / G;                             // units phase B runs behind phase A
    static constexpr int NBU = (G - 1 + 16) / G;                       // units of bb kept after a candidate's unit
    static constexpr int CRU = US + LAGU + LPB + 1;                // ring capacity in units
    static constexpr int T = US * R;                               // samples per step
    static constexpr int PIECES = T / 2;                               // 16-byte pieces (2 samples) per step
    static constexpr int NLD = (PIECES + FE4_NT - 1) / FE4_NT;         // loads per thread and step
    static constexpr int LPR = R / 2;                                  // lanes that move one row (8 bytes each)
    static constexpr int RPI = AM_WAVE / LPR;                          // rows per wave instruction
    static_assert(AM_CHIPS_AVG % G == 0 && R % 2 == 0 && R <= 32 && RS % 2 == 0, "unit shape");
    static constexpr int LDS_FLOATS = CRU * RS + FE4_NW * 4 * RS + 3 * CRU + 2 * SPC + (FE4_NW - 1) * SPC + 2 + FE4_NW * 64 + 4 + 2 * FE4_NW;
};

template <int SPC, int G>
struct fe4_smem {
    float *X;                 // [CRU * RS] ring rows: |.|^2 of a step's units while it is staged, then bb
    float *UPT, *UST, *USL;   // [CRU] per unit: in-block prefix of the chip totals entering it from the left; suffix entering it
                              // from the right (= ST of its last chip); RTOT + ST of its first chip
    float *SBL;               // [2][SPC] in-chip suffix sums of |.|^2 of a step's last chip (by step parity)
    float *MLW;               // [NW-1][SPC] |.|^2 of the last chip of wave w's last unit (the chip before wave w+1's first)
    float *AVS;               // [NW][4 * RS] four units of reference level on their way out
    uint32_t *CARRY;          // [2] units at the start of the next step whose bb must be written (by step parity)
    uint32_t *TAB;            // [NW][64] lane of the r-th unit whose bb / reference level is written
    float *WMX;               // [NW] the waves' largest samples at the end
    uint32_t *WOV;            // [2][NW] units after a wave's last whose bb must be written (by step parity; LU = 64 only)
};

template <int SPC, int G>
__device__ __forceinline__ fe4_smem<SPC, G> fe4_smem_at(float *base)
{
    using C = fe4_cfg<SPC, G>;
    fe4_smem<SPC, G> L;
    L.X = base;                                                       // (the arrays read 8 bytes at a time first: even sizes)
    L.AVS = L.X + C::CRU * C::RS;
    L.UPT = L.AVS + FE4_NW * 4 * C::RS;
    L.UST = L.UPT + C::CRU;
    L.USL = L.UST + C::CRU;
    L.SBL = L.USL + C::CRU;
    L.MLW = L.SBL + 2 * SPC;
    L.CARRY = reinterpret_cast<uint32_t *>(L.MLW + (FE4_NW - 1) * SPC);
    L.TAB = L.CARRY + 2;
    L.WMX = reinterpret_cast<float *>(L.TAB + FE4_NW * 64);
    L.WOV = reinterpret_cast<uint32_t *>(L.WMX + 4);
    return L;
}

__device__ __forceinline__ void fe4_barrier()
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
}
// value of lane-1 (lane 0 of a wave gets `first`) / of lane+1 (lane 63 gets `last`)
__device__ __forceinline__ float fe4_from_prev_lane(float v, float first, int lane)
{
#if defined(__HIP_DEVICE_COMPILE__)
    (void)lane;
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, first), __builtin_bit_cast(int, v),
                                                                 0x138, 0xf, 0xf, false));
#else
    const float s = __shfl_up(v, 1, AM_WAVE);
    return lane == 0 ? first : s;
#endif
}
__device__ __forceinline__ float fe4_from_next_lane(float v, float last, int lane)
{
#if defined(__HIP_DEVICE_COMPILE__)
    (void)lane;
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, last), __builtin_bit_cast(int, v),
                                                                 0x130, 0xf, 0xf, false));
#else
    const float s = __shfl_down(v, 1, AM_WAVE);
    return lane == AM_WAVE - 1 ? last : s;
#endif
}
#if defined(__clang__)
typedef float fe4_f4 __attribute__((ext_vector_type(4)));
#endif
__device__ __forceinline__ float4 fe4_gload16(const void *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const fe4_f4 t = __builtin_nontemporal_load(reinterpret_cast<const fe4_f4 *>(p));
    float4 r; r.x = t.x; r.y = t.y; r.z = t.z; r.w = t.w;
    return r;
#else
    return *reinterpret_cast<const float4 *>(p);
#endif
}

template <int SPC, int G>
__device__ __forceinline__ int fe4_wrap_up(int s) { return s >= fe4_cfg<SPC, G>::CRU ? s - fe4_cfg<SPC, G>::CRU : s; }
template <int SPC, int G>
__device__ __forceinline__ int fe4_wrap_dn(int s) { return s < 0 ? s + fe4_cfg<SPC, G>::CRU : s; }

// |iq|^2 of one step into the ring rows of its units: piece p = tid + 128 j holds samples 2p, 2p+1 of the step
// = unit (2p) / R, offset (2p) % R (R is even: a piece never straddles two units).  GUARD: stream edges / unaligned
// input, one sample at a time, zeros outside the stream.  P0: pieces below it are not loaded (ring rebuild).
template <int SPC, int G, bool GUARD>
__device__ __forceinline__ void fe4_stage_step(const am_fe4_args &a, const fe4_smem<SPC, G> &L, long long A0, int slot0, int tid,
                                               int p0)
{
    using C = fe4_cfg<SPC, G>;
    auto put = [&](int p, float m0, float m1) __attribute__((always_inline)) {
        const int u = (2 * p) / C::R, o = (2 * p) % C::R;
        float2 mm; mm.x = m0; mm.y = m1;
        *reinterpret_cast<float2 *>(L.X + fe4_wrap_up<SPC, G>(slot0 + u) * C::RS + o) = mm;
        // the last chip of a wave's last unit a second time: the next wave needs it after the row holds bb
        if ((u + 1) % C::LU == 0 && u + 1 < C::US && o >= C::R - SPC - (SPC & 1)) {
            float *d = L.MLW + ((u + 1) / C::LU - 1) * SPC;
            const int k = o - (C::R - SPC);
            if (k >= 0) d[k] = m0;
            if (k + 1 >= 0 && k + 1 < SPC) d[k + 1] = m1;
        }
    };
    if constexpr (GUARD) {
        const float2 *iq2 = reinterpret_cast<const float2 *>(a.iq);
#pragma unroll 1
        for (int j = 0; j < C::NLD; ++j) {
            const int p = tid + FE4_NT * j;
            if (p >= C::PIECES) break;
            const long long n = A0 + 2 * (long long)p;
            float2 u0, u1;
            u0.x = 0.0f; u0.y = 0.0f; u1 = u0;
            if (n >= a.src_abs0 && n < a.src_abs1) u0 = iq2[n - a.src_abs0];
            if (n + 1 >= a.src_abs0 && n + 1 < a.src_abs1) u1 = iq2[n + 1 - a.src_abs0];
            const float r0 = u0.x * u0.x, i0 = u0.y * u0.y, r1 = u1.x * u1.x, i1 = u1.y * u1.y;
            put(p, r0 + i0, r1 + i1);
        }
    } else {
        const unsigned char *gb = reinterpret_cast<const unsigned char *>(a.iq) + (size_t)(A0 - a.src_abs0) * 8;
        float4 v[C::NLD];
#pragma unroll
        for (int j = 0; j < C::NLD; ++j) {
            const int p = tid + FE4_NT * j;
            if (p < C::PIECES && p >= p0) v[j] = fe4_gload16(gb + (size_t)p * 16u);
        }
#pragma unroll
        for (int j = 0; j < C::NLD; ++j) {
            const int p = tid + FE4_NT * j;
            if (p < C::PIECES && p >= p0) {
                const float r0 = v[j].x * v[j].x, i0 = v[j].y * v[j].y, r1 = v[j].z * v[j].z, i1 = v[j].w * v[j].w;
                put(p, r0 + i0, r1 + i1);                                 // a1: fl(fl(I*I) + fl(Q*Q))
            }
        }
    }
}

template <int SPC, int G>
__device__ __forceinline__ void fe4_step(const am_fe4_args &a, const fe4_smem<SPC, G> &L, const int step, const bool test,
                                         const int slot0, const int par, const bool edge, const int tid, float &mxrun,
                                         bool &badrun)
{
    using C = fe4_cfg<SPC, G>;
    constexpr int R = C::R, RS = C::RS, LPB = C::LPB, LAGU = C::LAGU;
    const int lane = tid & (AM_WAVE - 1), wv = tid / AM_WAVE;
    constexpr int LU = C::LU;
    const bool unit_thread = lane < LU;
    const int t = wv * LU + (unit_thread ? lane : LU - 1);    // unit of the step (spare lanes shadow the last one, never write)
    const long long A0 = a.out_abs0 + (long long)step * C::T;
    const int slotA = fe4_wrap_up<SPC, G>(slot0 + t);
    const bool do_pmf = a.use_pmf != 0 && SPC > 1;
    const int lb = lane % LPB;                                        // position of the unit inside its 48-chip block

    // ---- phase A ---------------------------------------------------------------------------------------------------
    float bb[R];
    {
        float m[R];
        const float2 *mp = reinterpret_cast<const float2 *>(L.X + slotA * RS);
#pragma unroll
        for (int k = 0; k < R / 2; ++k) { const float2 u = mp[k]; m[2 * k] = u.x; m[2 * k + 1] = u.y; }
        if (do_pmf) {
            // in-chip suffix sums (right -> left) and prefix sums (left -> right), restarted at every chip
            float sx[R], pp[R];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float as = 0.0f, ap = 0.0f;
#pragma unroll
                for (int i = 0; i < SPC; ++i) {
                    as = as + m[g * SPC + SPC - 1 - i]; sx[g * SPC + SPC - 1 - i] = as;
                    ap = ap + m[g * SPC + i]; pp[g * SPC + i] = ap;
                }
            }
            // the step's last chip hands its suffix sums to the next step's first chip
            if (tid == (FE4_NW - 1) * AM_WAVE + LU - 1) {
#pragma unroll
                for (int i = 0; i < SPC; ++i) L.SBL[par * SPC + i] = sx[(G - 1) * SPC + i];
            }
            // suffix sums of the chip before the unit's first: lane-1's last chip; lane 0 of wave 0: the previous step's
            // last chip (LDS); lane 0 of a later wave: recomputed from the staged |.|^2 of that chip
            float pv[SPC];
            {
                const float *src = (wv == 0) ? (L.SBL + (par ^ 1) * SPC) : (L.MLW + (wv - 1) * SPC);
#pragma unroll
                for (int i = 0; i < SPC; ++i) pv[i] = src[i];
                if (wv != 0) {                                        // (uniform)
                    float acc2 = 0.0f;
#pragma unroll
                    for (int i = SPC - 1; i >= 0; --i) { acc2 = acc2 + pv[i]; pv[i] = acc2; }
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int i = 0; i < SPC; ++i) {
                    const int j = g * SPC + i;
                    if (i == SPC - 1) bb[j] = pp[j] * a.s1;           // the window is the chip
                    else if (g == 0) bb[j] = (fe4_from_prev_lane(sx[(G - 1) * SPC + i + 1], pv[i + 1], lane) + pp[j]) * a.s1;
                    else bb[j] = (sx[(g - 1) * SPC + i + 1] + pp[j]) * a.s1;
                }
        } else {
#pragma unroll
            for (int j = 0; j < R; ++j) bb[j] = m[j];
        }
    }
    if (edge) {
        // positions beyond the end of the stream read as zero (the preamble view pads with zeros)
        const long long left = a.src_abs1 - (A0 + (long long)t * R);
        const int nin = left >= R ? R : (left <= 0 ? 0 : (int)left);
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (j >= nin) bb[j] = 0.0f;
    }
    {
        // chip totals left -> right (f) and right -> left (b); spare lanes contribute zeros to the scans
        float f[G], b0 = 0.0f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float ff = 0.0f, bk = 0.0f;
#pragma unroll
            for (int i = 0; i < SPC; ++i) { ff = ff + bb[g * SPC + i]; bk = bk + bb[g * SPC + SPC - 1 - i]; mxrun = fmaxf(mxrun, bb[g * SPC + i]); }
            if (!unit_thread) ff = 0.0f;
            f[g] = ff;
            if (g == 0) b0 = bk;
            badrun = badrun || !(ff < __builtin_inff());
        }
        // in-block scans of the chip totals, strictly sequential (canonical order), hopping from lane to lane: the value
        // entering a lane from the left is ((x(lane-1) + f0) + f1) + ... of lane-1, 0 where a block starts; a lane's value
        // is final after as many rounds as its position in the block and is recomputed identically afterwards
        float xin = 0.0f, yin = 0.0f;
#pragma unroll
        for (int r = 0; r < LPB - 1; ++r) {
            float xe = xin, ye = yin;
#pragma unroll
            for (int g = 0; g < G; ++g) { xe = xe + f[g]; ye = ye + f[G - 1 - g]; }
            const float xp = fe4_from_prev_lane(xe, 0.0f, lane), yn = fe4_from_next_lane(ye, 0.0f, lane);
            xin = (lb == 0) ? 0.0f : xp;
            yin = (lb == LPB - 1) ? 0.0f : yn;
        }
        if (unit_thread) {
            float st0 = yin;                                          // ST of the unit's first chip: the later chips of the unit, right -> left
#pragma unroll
            for (int g = G - 1; g >= 1; --g) st0 = st0 + f[g];
            L.UPT[slotA] = xin;
            L.UST[slotA] = yin;
            L.USL[slotA] = b0 + st0;
            float2 *xp = reinterpret_cast<float2 *>(L.X + slotA * RS);
#pragma unroll
            for (int k = 0; k < R / 2; ++k) { float2 u; u.x = bb[2 * k]; u.y = bb[2 * k + 1]; xp[k] = u; }
        }
    }
    fe4_barrier();                                                    // B3: ring and scans of this step complete
    if (!test) return;                                                // (uniform) ring rebuild only

    // ---- phase B on unit v = (this thread's phase-A unit) - LAGU --------------------------------------------------
    const int slotB = fe4_wrap_dn<SPC, G>(slotA - LAGU);
    const int slotS = fe4_wrap_dn<SPC, G>(slotB - LPB);               // the unit 48 chips back
    float x[R], avgv[R];
    {
        float sc[R];
        const float2 *xp = reinterpret_cast<const float2 *>(L.X + slotB * RS);
        const float2 *sp = reinterpret_cast<const float2 *>(L.X + slotS * RS);
#pragma unroll
        for (int k = 0; k < R / 2; ++k) { const float2 u = xp[k]; x[2 * k] = u.x; x[2 * k + 1] = u.y; }
#pragma unroll
        for (int k = 0; k < R / 2; ++k) { const float2 u = sp[k]; sc[2 * k] = u.x; sc[2 * k + 1] = u.y; }
        const float xinB = L.UPT[slotB], yinS = L.UST[slotS];
        const float sl_next = L.USL[fe4_wrap_up<SPC, G>(slotS + 1)];  // RTOT + ST of the chip after the back unit's last
        // position of the unit's first chip inside its block (phase B's unit is LAGU behind: (lb - LAGU) mod LPB)
        const int lbB = (lb + LPB * 8 - LAGU) % LPB;
        // back unit: forward chip totals (for ST), right -> left in-chip suffix sums (their first value is RTOT)
        float fS[G], stS[G], rtS[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float ff = 0.0f, as = 0.0f;
#pragma unroll
            for (int i = 0; i < SPC; ++i) ff = ff + sc[g * SPC + i];           // (before the row is overwritten with its suffix sums)
#pragma unroll
            for (int i = 0; i < SPC; ++i) { as = as + sc[g * SPC + SPC - 1 - i]; sc[g * SPC + SPC - 1 - i] = as; }
            fS[g] = ff; rtS[g] = as;
        }
        {
            float y = yinS;                                           // ST of the back unit's last chip
#pragma unroll
            for (int g = G - 1; g >= 0; --g) { stS[g] = y; y = y + fS[g]; }
        }
        // own unit: in-chip prefix sums, PT per chip
        float pt = xinB;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float ap = 0.0f;
            // (the block's last chip: lbB == LPB - 1 and g == G - 1)
            const bool blk_last = (lbB == LPB - 1) && (g == G - 1);
            const float suf_last = (g + 1 < G) ? (rtS[(g + 1 < G) ? g + 1 : g] + stS[(g + 1 < G) ? g + 1 : g]) : sl_next;
#pragma unroll
            for (int i = 0; i < SPC; ++i) {
                const int j = g * SPC + i;
                ap = ap + x[j];
                const float pre = pt + ap;
                float s;
                if (i == SPC - 1) s = blk_last ? pre : (suf_last + pre);
                else s = (sc[j + 1] + stS[g]) + pre;
                avgv[j] = s * a.sL;
            }
            pt = pt + ap;                                             // PT of the next chip: + this chip's forward total
        }
    }
    // array coordinate of x[0]
    const long long jstep = (long long)step * C::T - (long long)(LAGU * R);
    const long long jrun = jstep + (long long)t * R;
    uint32_t cm = 0u;
    {
        // first-stage test (preamble_impl.cc:172-179): the sample after, and the pulses 2, 7 and 9 chips on, lie in this or
        // later units at compile-time offsets
        auto ahead = [&](int j, int chips) __attribute__((always_inline)) {
            const int g = j / SPC + chips, i = j % SPC;
            const int d = g / G, gg = g % G;
            return d == 0 ? x[gg * SPC + i] : L.X[fe4_wrap_up<SPC, G>(slotB + d) * RS + gg * SPC + i];
        };
        const float nxt = L.X[fe4_wrap_up<SPC, G>(slotB + 1) * RS];
        // eight samples at a time (the partial results of more would not fit the scalar registers: they are lane masks)
#pragma unroll
        for (int h = 0; h < R; h += 8) {
            constexpr int CH = 8;
            float thr[CH], xs[CH + 1];
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const int j = (h + k < R) ? h + k : R - 1;
                thr[k] = avgv[j] * a.thr_lin;                            // :173
                xs[k] = x[j];
            }
            xs[CH] = (h + CH < R) ? x[(h + CH < R) ? h + CH : R - 1] : nxt;
            uint32_t part = 0u;
#if defined(FE2_CMPX)
            if (h + CH <= R) {
                fe2_peak8<0>(part, &xs[0], xs[CH], &thr[0]);
            } else
#endif
            {
#pragma unroll
                for (int k = 0; k < CH; ++k)
                    if (h + k < R) {
                        const float nx = (h + k + 1 < R) ? x[(h + k + 1 < R) ? h + k + 1 : R - 1] : nxt;
                        part |= ((xs[k] > thr[k]) & !(nx > xs[k])) ? (1u << k) : 0u;   // :174, :175
                    }
            }
            // the three later pulses must not be below the threshold (:177-179): one test on the smallest (a NaN pulse is not
            // below it); only where some lane has a survivor
            if (__ballot(part != 0u) != 0ull) {
                float wk[CH];
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const int j = (h + k < R) ? h + k : R - 1;
                    wk[k] = fminf(fminf(ahead(j, 2), ahead(j, 7)), ahead(j, 9));
                }
#if defined(FE2_CMPX)
                if (h + CH <= R) {
                    fe2_weak8<0>(part, &wk[0], &thr[0]);
                } else
#endif
                {
#pragma unroll
                    for (int k = 0; k < CH; ++k)
                        if (h + k < R && (wk[k] < thr[k])) part &= ~(1u << k);
                }
            }
            cm |= part << h;
        }
    }
    if (edge) {
        const long long lo = (long long)a.j0 - jrun, hi = (long long)a.j1 - jrun;
        uint32_t keep = 0u;
        if (hi > 0 && lo < R) {
            keep = hi >= 32 ? 0xFFFFFFFFu : ((1u << (int)hi) - 1u);
            if (lo > 0) keep &= ~((1u << (int)lo) - 1u);
        }
        cm &= keep;
    }
    if (!unit_thread) cm = 0u;
    if (unit_thread) a.bits[(size_t)step * C::US + t] = cm;
    uint32_t cnt = (uint32_t)__popcll((unsigned long long)cm);
    for (int o = 32; o >= 1; o >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, o, AM_WAVE);
    if (lane == 0) a.seg_cnt[(size_t)step * FE4_NW + wv] = cnt;
    const unsigned long long cand = __ballot(cm != 0u);               // bit l: unit 48 wave + l has a candidate

    // ---- sparse outputs (am_fe3.hip): rows are R floats, moved 8 bytes per lane, RPI rows per wave instruction ------------
    const long long lo64 = -jstep, hi64 = a.out_n - jstep;            // elements [lo, hi) of this step's coordinates exist
    const int lo = lo64 <= 0 ? 0 : (lo64 > 0x7FFFFFF ? 0x7FFFFFF : (int)lo64);
    const int hi = hi64 <= 0 ? 0 : (hi64 > 0x7FFFFFF ? 0x7FFFFFF : (int)hi64);
    const int sub = lane / C::LPR, piece = lane % C::LPR;
    auto put2 = [&](float *dst, int rel, float2 u) __attribute__((always_inline)) {
        if (!edge || (rel >= lo && rel + 2 <= hi)) *reinterpret_cast<float2 *>(dst + rel) = u;
        else {
            if (rel >= lo && rel < hi) dst[rel] = u.x;
            if (rel + 1 >= lo && rel + 1 < hi) dst[rel + 1] = u.y;
        }
    };
    {
        // reference level: the unit of a candidate and the one after it (a wave's lane 0 cannot see the unit before it: always)
        const unsigned long long wa = (cand | (cand << 1) | 1ull) & C::LUMASK;
        const int nav = __popcll(wa);
        uint32_t *tab = L.TAB + wv * AM_WAVE;
        float *avs = L.AVS + wv * (4 * RS);
        const bool mine = ((wa >> lane) & 1ull) != 0ull;
        const int my_rank = __popcll(wa & ((1ull << lane) - 1ull));
        if (mine) tab[my_rank] = (uint32_t)lane;
        float *const dst = a.avg_sparse + jstep;
        for (int r0 = 0; r0 < nav; r0 += 4) {                         // (uniform trip count)
            if (mine && my_rank >= r0 && my_rank < r0 + 4) {
                float2 *d = reinterpret_cast<float2 *>(avs + (my_rank - r0) * RS);
#pragma unroll
                for (int k = 0; k < R / 2; ++k) { float2 u; u.x = avgv[2 * k]; u.y = avgv[2 * k + 1]; d[k] = u; }
            }
            __builtin_amdgcn_wave_barrier();
            const int r = r0 + sub;
            if (sub < 4 && sub < C::RPI && r < nav) {
                const int tu = wv * LU + (int)tab[r];
                put2(dst, tu * R + 2 * piece, *reinterpret_cast<const float2 *>(avs + sub * RS + 2 * piece));
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    {
        // bb: the units from a candidate's on that hold the 17 chips from its chip on, copied from the ring
        unsigned long long need = cand;
#pragma unroll
        for (int k = 1; k <= C::NBU; ++k) need |= cand << k;
        if constexpr (LU == 64) {
            // every lane owns a unit: what reaches past the wave's last unit goes to the next wave (through LDS, behind a
            // barrier the reference-level rows above give slack to), or, from the last wave, to the next step
            uint32_t ov = 0u;
#pragma unroll
            for (int k = 1; k <= C::NBU; ++k) ov |= (uint32_t)(cand >> (64 - k));
            if (lane == 0) {
                if (wv == FE4_NW - 1) L.CARRY[par] = ov;
                else L.WOV[par * FE4_NW + wv] = ov;
            }
            fe4_barrier();
            if (wv == 0) need |= (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)L.CARRY[par ^ 1]);
            else need |= (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)L.WOV[par * FE4_NW + wv - 1]);
        } else {
            // (16 spare lanes: a wave reaches into the next wave's units itself)
            if (wv == 0) need |= (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)L.CARRY[par ^ 1]);
            if (wv == FE4_NW - 1) {
                if (lane == 0) L.CARRY[par] = (uint32_t)(need >> LU) & 0xFFFFu;
                need &= C::LUMASK;
            }
        }
        const int nflag = __popcll(need);
        uint32_t *tab = L.TAB + wv * AM_WAVE;
        __builtin_amdgcn_wave_barrier();                              // (the table is reused: the reference level's reads come first)
        if ((need >> lane) & 1ull) tab[__popcll(need & ((1ull << lane) - 1ull))] = (uint32_t)lane;
        __builtin_amdgcn_wave_barrier();
        float *const dst = a.bb_sparse + jstep;
        for (int r0 = 0; r0 < nflag; r0 += C::RPI) {                  // (uniform trip count)
            const int r = r0 + sub;
            if (sub < C::RPI && r < nflag) {
                const int tu = wv * LU + (int)tab[r];                 // test index of the unit
                const int slot = fe4_wrap_dn<SPC, G>(fe4_wrap_up<SPC, G>(slot0 + tu) - LAGU);
                put2(dst, tu * R + 2 * piece, *reinterpret_cast<const float2 *>(L.X + slot * RS + 2 * piece));
            }
        }
    }
}

template <int SPC, int G>
__global__ void __launch_bounds__(FE4_NT, (fe4_cfg<SPC, G>::MINW)) am_k_fe4(am_fe4_args a)
{
    using C = fe4_cfg<SPC, G>;
    HIP_DYNAMIC_SHARED(unsigned char, smem);
    const fe4_smem<SPC, G> L = fe4_smem_at<SPC, G>(reinterpret_cast<float *>(smem));
    const int tid0 = threadIdx.x;
    const int sb = (int)(blockIdx.x * a.steps_per_wg);
    if (sb >= (int)a.nsteps) return;
    const int se = (sb + (int)a.steps_per_wg < (int)a.nsteps) ? sb + (int)a.steps_per_wg : (int)a.nsteps;
    for (int i = tid0; i < C::LDS_FLOATS; i += FE4_NT) L.X[i] = 0.0f;
    fe4_barrier();
    if (tid0 < 2) L.CARRY[tid0] = 0xFFFFu;                            // the bb of a segment's first units is always written
    fe4_barrier();
    int slot0 = 0, par = 0;
    float mxrun = 0.0f;
    bool badrun = false;
    // the step before the segment only rebuilds the rings: its first tested unit is unit US - LAGU, whose reference level
    // reaches back LPB units
    constexpr int WARM_P0 = ((C::US - C::LAGU - C::LPB - 1) * C::R / 2 / FE4_NT) * FE4_NT;
    for (int step = sb - 1; step < se; ++step) {
        const bool test = step >= sb;
        const bool have = step >= a.raw_lo && step < a.raw_hi;
        const bool edge = !have || (test && !(step >= a.test_lo && step < a.test_hi));
        int tid = tid0;                                               // (nothing derived from the thread index lives across iterations: registers)
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(tid));
#endif
        if (have) fe4_stage_step<SPC, G, false>(a, L, a.out_abs0 + (long long)step * C::T, slot0, tid, test ? 0 : WARM_P0);
        else fe4_stage_step<SPC, G, true>(a, L, a.out_abs0 + (long long)step * C::T, slot0, tid, 0);
        fe4_barrier();                                                // B1: |.|^2 of this step staged
        fe4_step<SPC, G>(a, L, step, test, slot0, par, edge, tid, mxrun, badrun);
        slot0 = fe4_wrap_up<SPC, G>(slot0 + C::US);
        par ^= 1;
        fe4_barrier();                                                // B5: every ring read of this step done
    }
    {
        float wmx = mxrun;
        for (int o = 32; o >= 1; o >>= 1) wmx = fmaxf(wmx, __shfl_xor(wmx, o, AM_WAVE));
        const bool bad = __ballot(badrun) != 0ull;
        if ((tid0 & (AM_WAVE - 1)) == 0) L.WMX[tid0 / AM_WAVE] = bad ? __builtin_inff() : wmx;
        fe4_barrier();
        if (tid0 == 0) {
            float v = L.WMX[0];
            for (int w = 1; w < FE4_NW; ++w) v = fmaxf(v, L.WMX[w]);
            a.wg_max[blockIdx.x] = v;
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------------
// the specialisations that exist: chips per lane by samples per chip
static int fe4_g_of(int spc)
{
    switch (spc) {
    case 1: return 24;      //  2 Msps: 24 samples per lane
    case 2: return 16;      //  4 Msps: 32
    case 4: return 8;       //  8 Msps: 32
    case 5: return 6;       // 10 Msps: 30
    case 8: return 4;       // 16 Msps: 32
    case 10: return 3;      // 20 Msps: 30
    case 16: return 2;      // 32 Msps: 32
    case 20: return 1;      // 40 Msps: 20
    default: return 0;      // (64 Msps: am_k_fe3; everything else: the rate-generic kernels)
    }
}
int am_fe4_supported(int spc) { return fe4_g_of(spc) != 0 ? 1 : 0; }
unsigned am_fe4_unit(int spc) { return (unsigned)(spc * fe4_g_of(spc)); }                    // R: positions per bitmap word
unsigned am_fe4_words(int spc)                                                               // bitmap words (units) per step and wave
{
    const int g = fe4_g_of(spc);
    return g ? (unsigned)((AM_WAVE % (AM_CHIPS_AVG / g) == 0) ? AM_WAVE : AM_CHIPS_AVG) : 0u;
}
unsigned am_fe4_tile(int spc) { return (unsigned)FE4_NW * am_fe4_words(spc) * am_fe4_unit(spc); }   // positions per step
unsigned am_fe4_lag(int spc) { const int g = fe4_g_of(spc); return g ? (unsigned)((1 + 8 / g) * spc * g) : 0u; }
unsigned am_fe4_steps(long long out_n, int spc)
{
    const long long T = am_fe4_tile(spc);
    return T ? (unsigned)((out_n + am_fe4_lag(spc) + T - 1) / T) : 0u;
}

static long long fe4_floor_div(long long x, long long d) { return x >= 0 ? x / d : -((-x + d - 1) / d); }
static long long fe4_ceil_div(long long x, long long d) { return -fe4_floor_div(-x, d); }

template <int SPC, int G>
static hipError_t fe4_launch(am_fe4_args &a, unsigned *steps_per_wg, hipStream_t s)
{
    using C = fe4_cfg<SPC, G>;
    const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float);
    // workgroups per CU: what registers and LDS of THIS instantiation allow (6 to 8), asked of the runtime once per device
    static std::atomic<int> per_cu[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    int wpc = (dev >= 0 && dev < 64) ? per_cu[dev].load(std::memory_order_acquire) : 0;
    if (wpc <= 0) {
        hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void *>(&am_k_fe4<SPC, G>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (rc != hipSuccess) return rc;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(&am_k_fe4<SPC, G>), FE4_NT, lds) != hipSuccess ||
            nb <= 0)
            nb = FE4_WG_PER_CU;
        wpc = nb > 8 ? 8 : nb;
        if (dev >= 0 && dev < 64) per_cu[dev].store(wpc, std::memory_order_release);
    }
#if defined(AM_TEST_KNOBS)
    if (const char *e = getenv("AIRMODES_FE4_WGS_PER_CU"))
        if (atoi(e) > 0) wpc = atoi(e);
#endif
    const unsigned resident = (unsigned)(wpc * am_device_cus());
    unsigned spw = (a.nsteps + resident - 1) / resident;
    if (spw < 4) spw = 4;
    a.steps_per_wg = spw;
    *steps_per_wg = spw;
    const unsigned grid = (a.nsteps + spw - 1) / spw;
    hipLaunchKernelGGL((am_k_fe4<SPC, G>), dim3(grid), dim3(FE4_NT), lds, s, a);
    return hipGetLastError();
}

hipError_t am_launch_fe4(int spc, const float *iq, long long src_abs0, long long src_abs1, long long out_abs0, long long out_n,
                         float *bb_sparse, float *avg_sparse, uint32_t j0, uint32_t j1, int use_pmf, float s1, float sL,
                         float thr_lin, uint32_t *bits, uint32_t *seg_cnt, float *wg_max, unsigned *nsteps,
                         unsigned *steps_per_wg, hipStream_t s)
{
    if (!am_fe4_supported(spc)) return hipErrorInvalidValue;
    am_fe4_args a;
    a.iq = iq; a.src_abs0 = src_abs0; a.src_abs1 = src_abs1; a.out_abs0 = out_abs0; a.out_n = out_n;
    a.bb_sparse = bb_sparse; a.avg_sparse = avg_sparse; a.j0 = j0; a.j1 = j1; a.bits = bits; a.seg_cnt = seg_cnt; a.wg_max = wg_max;
    a.use_pmf = use_pmf ? 1 : 0; a.s1 = s1; a.sL = sL; a.thr_lin = thr_lin;
    a.nsteps = am_fe4_steps(out_n, spc);
    *nsteps = a.nsteps;
    *steps_per_wg = 1;
    if (a.nsteps == 0) return hipSuccess;
    const long long T = am_fe4_tile(spc), lag = am_fe4_lag(spc);
    const bool aligned = ((reinterpret_cast<uintptr_t>(iq) + (uintptr_t)(out_abs0 - src_abs0) * 8u) & 15u) == 0 && (T % 2) == 0;
    auto clampi = [](long long v) { return (int)(v < -4 ? -4 : (v > 0x7FFFFFF0ll ? 0x7FFFFFF0ll : v)); };
    a.raw_lo = clampi(fe4_ceil_div(src_abs0 - out_abs0, T));
    a.raw_hi = aligned ? clampi(fe4_floor_div(src_abs1 - out_abs0, T)) : a.raw_lo;
    const long long jhi = (long long)j1 < out_n ? (long long)j1 : out_n;
    a.test_lo = clampi(fe4_ceil_div((long long)j0 + lag, T));
    a.test_hi = clampi(fe4_floor_div(jhi + lag, T));
    switch (spc) {
    case 1: return fe4_launch<1, 24>(a, steps_per_wg, s);
    case 2: return fe4_launch<2, 16>(a, steps_per_wg, s);
    case 4: return fe4_launch<4, 8>(a, steps_per_wg, s);
    case 5: return fe4_launch<5, 6>(a, steps_per_wg, s);
    case 8: return fe4_launch<8, 4>(a, steps_per_wg, s);
    case 10: return fe4_launch<10, 3>(a, steps_per_wg, s);
    case 16: return fe4_launch<16, 2>(a, steps_per_wg, s);
    default: return fe4_launch<20, 1>(a, steps_per_wg, s);
    }
}
