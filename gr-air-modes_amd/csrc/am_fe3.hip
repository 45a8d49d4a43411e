// am_fe3.hip -- streaming fused front end + preamble detection for gfx950 at 32 samples per chip
// (64 Msps), the rate BASELINE.json's metric is quoted on.  Same results as am_k_fe2 / the oracle
// (canonical summation order), different machine mapping:
//
//   * PERSISTENT workgroups, three per CU (128 threads, ~50 KB of LDS each).  A workgroup owns a
//     contiguous segment of the stream and walks it in steps of 96 chips (two 48-chip blocks,
//     3072 samples).  What a step needs from the past -- the pulse-matched power bb of the last 57
//     chips and their per-chip sums -- stays in LDS rings, so nothing is loaded twice (the tile
//     kernel re-read a 49-chip halo per tile and could not shrink its tiles for that reason).
//   * RAW IQ ARRIVES BY LDS-DMA (global_load_lds_dwordx4): the 24 KB of step k+1 are in flight while
//     step k is processed; no VGPRs, no address arithmetic, no ds_write pass.  The DMA destination is
//     lane-linear, so the 16-byte pieces are permuted on the SOURCE side: thread t then reads its own
//     chip (16 pieces) from LDS without bank conflicts.
//   * thread = one chip (32 samples in registers).  The chip before it belongs to lane-1: its in-chip
//     suffix sums come over with DPP wave_shr:1 (no LDS traffic for the pulse-matched filter).
//   * phase B (reference level + first-stage test) runs 9 chips BEHIND phase A, so the pulses 2, 7 and 9
//     chips ahead are already in the ring: no right halo, no redundant arithmetic.
//   * outputs are sparse: one candidate bit per position (a dense bitmap, 1/64 of the input bytes) and,
//     only around candidates, the runs of bb (17 chips: what am_k_energy / am_k_cand read) and of the
//     reference level (2 chips).  The dense 4 B/sample bb array of the tile kernel was a third of its HBM
//     traffic; burst extraction now recomputes its 240 chip-spaced samples from IQ (am_kernels.hip).
//
// Reference: python/rx_path.py:38-54 (|.|^2, moving averages), lib/preamble_impl.cc:172-179 (test).
#include "am_internal.h"

#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <vector>

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#include "am_fe_cmpx.h"

// Two builds of the load path (FE3_DMA; measured side by side on MI355X):
//   0 (default): plain coalesced 16-byte loads, |.|^2 written straight into the ring slots of the new chips.  26 KB
//      of LDS per workgroup -> six workgroups (12 waves) per CU; a workgroup waits for its own loads, the other five
//      keep the CU busy.
//   1: raw IQ by LDS-DMA into a 24 KB staging buffer, one step ahead of the arithmetic (no wait, no VGPRs), but only
//      three workgroups (6 waves) per CU: the dependent chains of this arithmetic then leave the SIMDs idle.
#ifndef FE3_DMA
#define FE3_DMA 0
#endif
#if FE3_DMA
#define FE3_WPS 2                         /* launch bound, waves per SIMD */
#define FE3_WG_PER_CU 3
#else
#define FE3_WPS 3
#define FE3_WG_PER_CU 6
#endif
// tuning builds only (tools/build_variants.sh): FE3_ABLATE bit mask removes parts of the kernel -- results INVALID
#ifndef FE3_ABLATE
#define FE3_ABLATE 0
#endif
#define FE3_SPC 32
#define FE3_S 96                          /* chips per step: two 48-chip blocks                        */
#define FE3_NT 128                        /* 96 chip threads + 32 helpers (block scans, DMA issue)     */
#define FE3_T (FE3_S * FE3_SPC)           /* samples per step                                          */
#define FE3_LAG 9                         /* phase B runs this many chips behind phase A               */
#define FE3_CR 160                        /* ring capacity in chips: 96 new + 9 lag + 48 back + slack   */
#define FE3_XS 36                         /* floats per ring chip: 32 + 4 pad (16-byte reads of consecutive chips hit all banks) */
#define FE3_RAWB (FE3_T * 8)              /* raw bytes per step                                        */
#define FE3_BBW 17                        /* chips of bb kept after a candidate's chip (am_k_cand reads up to pos + 16*spc) */

struct am_fe3_args {
    const float *iq;
    long long src_abs0, src_abs1;         // absolute range of samples present in iq
    long long out_abs0;                   // absolute index of array coordinate 0 (multiple of 48*spc)
    long long out_n;                      // array coordinates with data
    float *bb_sparse;                     // bb runs around candidates (array coordinates)
    float *avg_sparse;                    // reference-level runs around candidates
    uint32_t j0, j1;                      // positions whose preamble test is wanted
    uint32_t *bits;                       // [nsteps * 96] candidate words: bit b of word w = position w*32 + b - 288
    uint32_t *seg_cnt;                    // [nsteps * 2] candidates per (step, wave)
    unsigned nsteps;                      // steps (= bitmap tiles) of the whole launch
    unsigned steps_per_wg;
    // steps whose raw samples are all present and 16-byte aligned arrive by DMA: [raw_lo, raw_hi); steps whose
    // tested positions are all wanted need no range mask: [test_lo, test_hi)   (host: fe3_ranges)
    int raw_lo, raw_hi, test_lo, test_hi;
    int use_pmf;
    float s1, sL, thr_lin;
    long long *prof;                      // profiling builds: [grid * 2 waves][12] cycles per phase, else null
};

// ---- small device helpers -------------------------------------------------------------------------
// ring slot arithmetic (operands within one ring length of the valid range)
__device__ __forceinline__ int fe3_wrap_up(int s) { return s >= FE3_CR ? s - FE3_CR : s; }     // s in [0, 2*CR)
__device__ __forceinline__ int fe3_wrap_dn(int s) { return s < 0 ? s + FE3_CR : s; }           // s in [-CR, CR)

// everything this workgroup wrote to LDS is visible to it after this; asynchronous LDS-DMA stays in flight
__device__ __forceinline__ void fe3_barrier()
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
}

// 16 bytes from global memory (gbase + voff + IMM) to LDS address (lds_base + IMM + lane * 16), asynchronously
// (counted by vmcnt): the instruction's immediate offset applies to BOTH addresses.  gbase and lds_base are
// wave-uniform (scalar registers), voff is the lane's byte offset.
template <int IMM>
__device__ __forceinline__ void fe3_dma16(const unsigned char *gbase, unsigned voff, unsigned lds_base,
                                          unsigned char *lds_generic, int lane)
{
#if defined(__HIP_DEVICE_COMPILE__)
    (void)lds_generic; (void)lane;
    // (s_nop 4: the scalar base may come straight from an SALU instruction the compiler placed in front of this
    // statement -- it does not know that a memory instruction reads it here)
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3 nt"
                 : : "v"(voff), "s"(gbase), "s"(lds_base), "n"(IMM) : "memory");
#else
    (void)lds_base;
    memcpy(lds_generic + IMM + (size_t)lane * 16, gbase + voff + IMM, 16);
#endif
}
__device__ __forceinline__ void fe3_dma_wait()
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// value of lane-1 (lane 0 of a wave keeps `old`): DPP wave_shr:1 on gfx9
__device__ __forceinline__ float fe3_from_prev_lane(float v, float old, int lane)
{
#if defined(__HIP_DEVICE_COMPILE__)
    (void)lane;
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                 0x138, 0xf, 0xf, false));
#else
    const float s = __shfl_up(v, 1, AM_WAVE);
    return lane == 0 ? old : s;
#endif
}

// Profiling builds only (-DFE3_PROFILE, tools/build_variants.sh): cycles per phase, summed over a workgroup's steps
// by lane 0 of each wave.  The default build contains none of it.
#if defined(FE3_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
struct fe3_prof { long long last; long long acc[12]; };
#define FE3_STAMP(k) do { const long long now__ = (long long)__builtin_readcyclecounter(); PR.acc[k] += now__ - PR.last; PR.last = now__; } while (0)
#else
struct fe3_prof { };
#define FE3_STAMP(k) do { } while (0)
#endif

struct fe3_smem {
    unsigned char *raw;       // [FE3_RAWB] DMA target, 16-byte pieces permuted per chip
    float *X;                 // [FE3_CR * FE3_XS] bb ring
    float *TOT, *RTOT, *PT, *ST;   // [FE3_CR] per-chip sums (left->right, right->left) and their in-block scans
    float *SB0;               // [2][32] in-chip suffix sums of |.|^2 of a step's last chip (by step parity)
    float *SB1;               // [32] DMA build: the same for chip 63 (last lane of wave 0); else |.|^2 of chip 63
    uint32_t *MASK;           // [2][4] chips with candidates, by step parity
};
#define FE3_LDS_BYTES ((FE3_DMA ? FE3_RAWB : 0) + FE3_CR * FE3_XS * 4 + 4 * FE3_CR * 4 + 3 * 32 * 4 + 8 * 4)

// what a thread keeps across steps
struct fe3_thread {
    unsigned raw_addr;        // byte offset of the own chip's staging row, already XORed with its swizzle
    unsigned dma_off[4];      // lane part of the DMA source offsets (wave-instruction j uses [j & 3])
};

// raw IQ of one step (first sample at byte address g0, wave-uniform) -> staging buffer: 24 KB = 24 wave-instructions
// of 1 KB, 12 per wave.  LDS piece p = 16 * chip + kk receives the chip's piece kk ^ (chip & 15): thread t then
// reads its 16 pieces with 16 consecutive lanes on 16 different bank groups.
__device__ __forceinline__ void fe3_issue_dma(const unsigned char *g0, const fe3_smem &L, const fe3_thread &T,
                                              unsigned raw_lds, int wv, int lane)
{
    wv = __builtin_amdgcn_readfirstlane(wv);                          // (bases go through scalar registers)
    const unsigned char *gw = g0 + (size_t)wv * 12288;                // this wave's 12 KB
    const unsigned lw = raw_lds + (unsigned)wv * 12288u;
    unsigned char *lg = L.raw + (size_t)wv * 12288;
#define FE3_DMA4(G)                                                                              \
    fe3_dma16<0>(gw + (G) * 4096, T.dma_off[0], lw + (G) * 4096u, lg + (G) * 4096, lane);        \
    fe3_dma16<1024>(gw + (G) * 4096, T.dma_off[1], lw + (G) * 4096u, lg + (G) * 4096, lane);     \
    fe3_dma16<2048>(gw + (G) * 4096, T.dma_off[2], lw + (G) * 4096u, lg + (G) * 4096, lane);     \
    fe3_dma16<3072>(gw + (G) * 4096, T.dma_off[3], lw + (G) * 4096u, lg + (G) * 4096, lane);
    FE3_DMA4(0)
    FE3_DMA4(1)
    FE3_DMA4(2)
#undef FE3_DMA4
}

// A step whose raw samples did not arrive by DMA (stream edges, unaligned input): the staging buffer is filled
// with guarded loads, zeros outside the stream, in the same permuted layout.
__device__ __forceinline__ void fe3_fill_raw_guarded(const am_fe3_args &a, const fe3_smem &L, long long A0, int tid)
{
    const float2 *iq2 = reinterpret_cast<const float2 *>(a.iq);
    for (int p = tid; p < FE3_T / 2; p += FE3_NT) {
        const int t = p >> 4, kk = p & 15;
        const int k = kk ^ (t & 15);
        const long long n = A0 + (long long)t * FE3_SPC + 2 * k;
        float4 v;
        v.x = 0.0f; v.y = 0.0f; v.z = 0.0f; v.w = 0.0f;
        if (n >= a.src_abs0 && n < a.src_abs1) { const float2 u = iq2[n - a.src_abs0]; v.x = u.x; v.y = u.y; }
        if (n + 1 >= a.src_abs0 && n + 1 < a.src_abs1) { const float2 u = iq2[n + 1 - a.src_abs0]; v.z = u.x; v.w = u.y; }
        *reinterpret_cast<float4 *>(L.raw + (size_t)p * 16) = v;
    }
}

#if !FE3_DMA
// |iq|^2 of one step straight into the ring slots of its chips (they hold chips nobody needs any more): piece
// p = tid + 128 j (16 bytes = samples 2k, 2k+1 of chip p >> 4, k = p & 15 = tid & 15) -> X[slot][2k .. 2k+1].
// Coalesced loads (consecutive lanes, consecutive pieces), 8-byte LDS stores (16 lanes = one chip's 128 bytes).
// chip 63's values are stored a second time (M63): the other wave needs them after chip 63's slot holds bb.
template <bool GUARD>
__device__ __forceinline__ void fe3_stage_step(const am_fe3_args &a, const fe3_smem &L, long long A0, int slot0, int tid)
{
    const int c0 = tid >> 4, k = tid & 15;
    if constexpr (GUARD) {
        // stream edges / unaligned input: one piece at a time, zeros outside the stream (rare: kept small)
        const float2 *iq2 = reinterpret_cast<const float2 *>(a.iq);
#pragma unroll 1
        for (int j = 0; j < 12; ++j) {
            const long long n = A0 + 2 * (long long)(tid + FE3_NT * j);
            float2 u0, u1;
            u0.x = 0.0f; u0.y = 0.0f; u1 = u0;
            if (n >= a.src_abs0 && n < a.src_abs1) u0 = iq2[n - a.src_abs0];
            if (n + 1 >= a.src_abs0 && n + 1 < a.src_abs1) u1 = iq2[n + 1 - a.src_abs0];
            const float r0 = u0.x * u0.x, i0 = u0.y * u0.y, r1 = u1.x * u1.x, i1 = u1.y * u1.y;
            float2 mm;
            mm.x = r0 + i0;
            mm.y = r1 + i1;
            const int slot = fe3_wrap_up(slot0 + c0 + 8 * j);
            *reinterpret_cast<float2 *>(L.X + slot * FE3_XS + 2 * k) = mm;
            if (j == 7 && c0 == 7) *reinterpret_cast<float2 *>(L.SB1 + 2 * k) = mm;
        }
        return;
    }
    // wave-uniform 64-bit base + 32-bit lane offset
    const unsigned char *gb = reinterpret_cast<const unsigned char *>(a.iq) + (size_t)(A0 - a.src_abs0) * 8;
    const unsigned off = (unsigned)tid * 16u;
    float4 v[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) v[j] = *reinterpret_cast<const float4 *>(gb + (off + (unsigned)j * (FE3_NT * 16u)));
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const float r0 = v[j].x * v[j].x, i0 = v[j].y * v[j].y, r1 = v[j].z * v[j].z, i1 = v[j].w * v[j].w;
        float2 mm;
        mm.x = r0 + i0;                                               // a1: fl(fl(I*I) + fl(Q*Q))
        mm.y = r1 + i1;
        const int slot = fe3_wrap_up(slot0 + c0 + 8 * j);
        *reinterpret_cast<float2 *>(L.X + slot * FE3_XS + 2 * k) = mm;
        if (j == 7 && c0 == 7) *reinterpret_cast<float2 *>(L.SB1 + 2 * k) = mm;   // chip 63
    }
}
#endif

// One step.
//   step     global step index (may be -1: history before the first wanted block)
//   test     false for a workgroup's first step (it only rebuilds the rings from the previous segment's tail)
//   slot0    ring slot of this step's chip 0
//   edge     (uniform) the step touches the end of the stream or positions that are not wanted
__device__ __forceinline__ void fe3_step(const am_fe3_args &a, const fe3_smem &L, const fe3_thread &T,
                                         const unsigned raw_lds, const int step, const bool test, const int slot0,
                                         const int par, const bool issue_next, const bool edge, fe3_prof &PR)
{
    constexpr int SPC = FE3_SPC;
    const int tid = threadIdx.x, lane = tid & (AM_WAVE - 1), wv = tid / AM_WAVE;
    const bool chip_thread = tid < FE3_S;
    const long long A0 = a.out_abs0 + (long long)step * FE3_T;      // absolute index of the step's first sample
    const int slotA = fe3_wrap_up(slot0 + tid);                       // (only meaningful for chip threads)
    const bool do_pmf = a.use_pmf != 0;

    // ---- phase A1: |iq|^2 of the own chip, in-chip suffix sums ---------------------------------------------------
    float m[SPC];
#if FE3_DMA
    // raw IQ from the staging buffer (piece k sits at raw_addr ^ (k << 4))
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float4 v = *reinterpret_cast<const float4 *>(L.raw + (T.raw_addr ^ (unsigned)(k << 4)));
        const float r0 = v.x * v.x, i0 = v.y * v.y, r1 = v.z * v.z, i1 = v.w * v.w;
        m[2 * k] = r0 + i0;                                           // a1: fl(fl(I*I) + fl(Q*Q))
        m[2 * k + 1] = r1 + i1;
    }
#else
    {
        const float4 *mp = reinterpret_cast<const float4 *>(L.X + slotA * FE3_XS);
#pragma unroll
        for (int k = 0; k < SPC / 4; ++k) {
            const float4 t = mp[k];
            m[4 * k] = t.x; m[4 * k + 1] = t.y; m[4 * k + 2] = t.z; m[4 * k + 3] = t.w;
        }
    }
#endif
    // suffix sums of the own chip (what the NEXT chip's filter needs): sx[i] = m[31] + ... + m[i], right->left
    float sx[SPC];
    if (do_pmf) {
        float acc = 0.0f;
#pragma unroll
        for (int i = SPC - 1; i >= 0; --i) { acc = acc + m[i]; sx[i] = acc; }
        // the step's last chip hands its sums to the next step's first chip (DMA build: and chip 63 to chip 64,
        // other wave; the plain-load build staged chip 63's |.|^2 for that)
        if (tid == FE3_S - 1 || (FE3_DMA && tid == AM_WAVE - 1)) {
            float4 *dst = reinterpret_cast<float4 *>((tid == FE3_S - 1) ? (L.SB0 + par * 32) : L.SB1);
#pragma unroll
            for (int k = 0; k < SPC / 4; ++k) {
                float4 t;
                t.x = sx[4 * k]; t.y = sx[4 * k + 1]; t.z = sx[4 * k + 2]; t.w = sx[4 * k + 3];
                dst[k] = t;
            }
        }
    }
    FE3_STAMP(1);
#if FE3_DMA
    fe3_barrier();                                                    // B2: staging buffer read by everyone
    FE3_STAMP(2);
    if (issue_next)
        fe3_issue_dma(reinterpret_cast<const unsigned char *>(a.iq) + (size_t)((A0 + FE3_T) - a.src_abs0) * 8, L, T,
                      raw_lds, wv, lane);
#endif
    float bb[SPC];
    if (do_pmf) {
        // suffix sums of the chip before: lane-1, except lane 0 of a wave (from LDS; every lane reads: a broadcast)
        float pv[SPC];
        {
            const float4 *src = reinterpret_cast<const float4 *>((wv == 0) ? (L.SB0 + (par ^ 1) * 32) : L.SB1);
#pragma unroll
            for (int k = 0; k < SPC / 4; ++k) {
                const float4 t = src[k];
                pv[4 * k] = t.x; pv[4 * k + 1] = t.y; pv[4 * k + 2] = t.z; pv[4 * k + 3] = t.w;
            }
#if !FE3_DMA
            if (wv != 0) {                                            // (uniform) |.|^2 of chip 63 -> its suffix sums
                float acc2 = 0.0f;
#pragma unroll
                for (int i = SPC - 1; i >= 0; --i) { acc2 = acc2 + pv[i]; pv[i] = acc2; }
            }
#endif
        }
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < SPC; ++i) {
            acc = acc + m[i];                                         // in-chip prefix, left->right
            if (i == SPC - 1) bb[i] = acc * a.s1;                     // the window is the chip
            else bb[i] = (fe3_from_prev_lane(sx[i + 1], pv[i + 1], lane) + acc) * a.s1;
        }
    } else {
#pragma unroll
        for (int i = 0; i < SPC; ++i) bb[i] = m[i];
    }
    if (edge) {
        // positions beyond the end of the stream read as zero (the preamble view pads with zeros)
        long long left = a.src_abs1 - (A0 + (long long)tid * SPC);      // samples of this chip inside the stream
        const int nin = left >= SPC ? SPC : (left <= 0 ? 0 : (int)left);
#pragma unroll
        for (int i = 0; i < SPC; ++i)
            if (i >= nin) bb[i] = 0.0f;
    }
    // ---- phase A2: chip totals in both directions, bb -> ring ------------------------------------------------
    if (chip_thread) {
        float f = 0.0f, b = 0.0f;
#pragma unroll
        for (int i = 0; i < SPC; ++i) f = f + bb[i];
#pragma unroll
        for (int i = SPC - 1; i >= 0; --i) b = b + bb[i];
        L.TOT[slotA] = f;
        L.RTOT[slotA] = b;
        float4 *xp = reinterpret_cast<float4 *>(L.X + slotA * FE3_XS);
#pragma unroll
        for (int k = 0; k < SPC / 4; ++k) {
            float4 t;
            t.x = bb[4 * k]; t.y = bb[4 * k + 1]; t.z = bb[4 * k + 2]; t.w = bb[4 * k + 3];
            xp[k] = t;
        }
    }
    FE3_STAMP(3);
    fe3_barrier();                                                    // B3: ring and totals of this step complete
    FE3_STAMP(4);

    // ---- in-block scans of the two new blocks (4 helper lanes: block x direction), canonical sequential order.
    // Blocks start at multiples of 16 slots, so groups of four consecutive chips never straddle the ring's end.
    if (!(FE3_ABLATE & 8) && tid >= FE3_S && tid < FE3_S + 4) {
        const int blk = (tid - FE3_S) >> 1;
        const int s0 = fe3_wrap_up(slot0 + blk * AM_CHIPS_AVG);
        float t[AM_CHIPS_AVG];
#pragma unroll
        for (int g = 0; g < AM_CHIPS_AVG / 4; ++g) {
            const float4 v = *reinterpret_cast<const float4 *>(L.TOT + fe3_wrap_up(s0 + 4 * g));
            t[4 * g] = v.x; t[4 * g + 1] = v.y; t[4 * g + 2] = v.z; t[4 * g + 3] = v.w;
        }
        float acc = 0.0f;
        float *dst = L.PT;
        if (tid & 1) {
            dst = L.ST;
#pragma unroll
            for (int j = AM_CHIPS_AVG - 1; j >= 0; --j) { const float v = t[j]; t[j] = acc; acc = acc + v; }
        } else {
#pragma unroll
            for (int j = 0; j < AM_CHIPS_AVG; ++j) { const float v = t[j]; t[j] = acc; acc = acc + v; }
        }
#pragma unroll
        for (int g = 0; g < AM_CHIPS_AVG / 4; ++g) {
            float4 v;
            v.x = t[4 * g]; v.y = t[4 * g + 1]; v.z = t[4 * g + 2]; v.w = t[4 * g + 3];
            *reinterpret_cast<float4 *>(dst + fe3_wrap_up(s0 + 4 * g)) = v;
        }
    }
    if (!test) {                                                      // (uniform) ring rebuild only
        fe3_barrier();                                                // B4
        return;
    }
    // phase B works on chip q = (this thread's phase-A chip) - 9
    const int slotB = fe3_wrap_dn(slotA - FE3_LAG);
    const int slotS = fe3_wrap_dn(slotB - AM_CHIPS_AVG);              // the chip 48 chips back
    float x[SPC], scv[SPC];
    float nxt;
    {
        const float4 *xp = reinterpret_cast<const float4 *>(L.X + slotB * FE3_XS);
        const float4 *sp = reinterpret_cast<const float4 *>(L.X + slotS * FE3_XS);
#pragma unroll
        for (int k = 0; k < SPC / 4; ++k) {
            const float4 t = xp[k];
            x[4 * k] = t.x; x[4 * k + 1] = t.y; x[4 * k + 2] = t.z; x[4 * k + 3] = t.w;
        }
#pragma unroll
        for (int k = 0; k < SPC / 4; ++k) {
            const float4 t = sp[k];
            scv[4 * k] = t.x; scv[4 * k + 1] = t.y; scv[4 * k + 2] = t.z; scv[4 * k + 3] = t.w;
        }
        nxt = L.X[fe3_wrap_up(slotB + 1) * FE3_XS];
        float acc = 0.0f;
#pragma unroll
        for (int i = SPC - 1; i >= 0; --i) { acc = acc + scv[i]; scv[i] = acc; }   // in-chip suffix sums, 48 chips back
    }
    FE3_STAMP(5);
    fe3_barrier();                                                    // B4: PT / ST of the new blocks
    FE3_STAMP(6);

    // ---- phase B: reference level (a4) + first-stage test (a6) ------------------------------------------------
    const int jb = (tid + AM_CHIPS_AVG - FE3_LAG) % AM_CHIPS_AVG;     // chip index inside its 48-chip block
    float avgv[SPC];
    {
        const int slotS1 = fe3_wrap_up(slotS + 1);
        const float pt = L.PT[slotB];
        const float st_a = L.ST[slotS];
        const float suf_last = L.RTOT[slotS1] + L.ST[slotS1];
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < SPC; ++i) {
            acc = acc + x[i];
            const float PRE = pt + acc;
            float s;
            if (i == SPC - 1) s = (jb == AM_CHIPS_AVG - 1) ? PRE : (suf_last + PRE);
            else s = (scv[i + 1] + st_a) + PRE;
            avgv[i] = s * a.sL;
        }
    }
    // array coordinate of x[0]
    const long long jrun = (long long)step * FE3_T + (long long)(tid * SPC - FE3_LAG * SPC);
    uint32_t cm = 0u;
    {
        constexpr int CH = 16;
        const int s2 = fe3_wrap_up(slotB + 2), s7 = fe3_wrap_up(slotB + 7), s9 = fe3_wrap_up(slotB + 9);
#if defined(FE2_CMPX)
        auto pass = [&](auto hc) __attribute__((always_inline)) {
            constexpr int H = decltype(hc)::value;
            float thr[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) thr[i] = avgv[H + i] * a.thr_lin;            // preamble_impl.cc:173
            uint32_t part = 0u;
            fe2_peak8<H>(part, &x[H], x[H + 8], &thr[0]);
            fe2_peak8<H + 8>(part, &x[H + 8], (H + 16 < SPC) ? x[(H + 16 < SPC) ? H + 16 : 0] : nxt, &thr[8]);
            // the three later pulses must not be below the threshold (:177-179): one test on the smallest
            // (v_min3 ignores a NaN operand exactly as `NaN < thr` is false); only where some lane has a survivor
            if (!(FE3_ABLATE & 2) && __ballot(part != 0u) != 0ull) {
                float t2[CH], t7[CH], t9[CH];
#pragma unroll
                for (int k = 0; k < CH / 4; ++k) {
                    const float4 u = reinterpret_cast<const float4 *>(L.X + s2 * FE3_XS + H)[k];
                    const float4 v = reinterpret_cast<const float4 *>(L.X + s7 * FE3_XS + H)[k];
                    const float4 w = reinterpret_cast<const float4 *>(L.X + s9 * FE3_XS + H)[k];
                    t2[4 * k] = u.x; t2[4 * k + 1] = u.y; t2[4 * k + 2] = u.z; t2[4 * k + 3] = u.w;
                    t7[4 * k] = v.x; t7[4 * k + 1] = v.y; t7[4 * k + 2] = v.z; t7[4 * k + 3] = v.w;
                    t9[4 * k] = w.x; t9[4 * k + 1] = w.y; t9[4 * k + 2] = w.z; t9[4 * k + 3] = w.w;
                }
#pragma unroll
                for (int i = 0; i < CH; ++i) t2[i] = fminf(fminf(t2[i], t7[i]), t9[i]);
                fe2_weak8<H>(part, &t2[0], &thr[0]);
                fe2_weak8<H + 8>(part, &t2[8], &thr[8]);
            }
            cm |= part;
        };
        if (!(FE3_ABLATE & 4)) {
        pass(std::integral_constant<int, 0>{});
        pass(std::integral_constant<int, CH>{});
        }
#else
        // (host build of the same source for the CPU-fiber tests: the predicate as plain C++)
#pragma unroll
        for (int h = 0; h < SPC; h += CH) {
            bool c[CH];
            float thr[CH];
            bool any = false;
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const float xv = x[h + i];
                thr[i] = avgv[h + i] * a.thr_lin;                        // preamble_impl.cc:173
                const float nx = (h + i + 1 < SPC) ? x[(h + i + 1 < SPC) ? h + i + 1 : h + i] : nxt;
                c[i] = (xv > thr[i]) & !(nx > xv);                       // :174, :175
                any = any | c[i];
            }
            if (__ballot(any) != 0ull) {
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const float weakest = fminf(fminf(L.X[s2 * FE3_XS + h + i], L.X[s7 * FE3_XS + h + i]),
                                                L.X[s9 * FE3_XS + h + i]);
                    c[i] = c[i] & !(weakest < thr[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) cm |= c[i] ? (1u << (h + i)) : 0u;
        }
#endif
    }
    if (edge) {
        // keep the positions in [j0, j1)
        const long long lo = (long long)a.j0 - jrun, hi = (long long)a.j1 - jrun;
        uint32_t keep = 0u;
        if (hi > 0 && lo < SPC) {
            keep = hi >= SPC ? 0xFFFFFFFFu : ((1u << (int)hi) - 1u);
            if (lo > 0) keep &= ~((1u << (int)lo) - 1u);
        }
        cm &= keep;
    }
    if (!chip_thread) cm = 0u;
    // candidate word, per-wave count, chips with candidates
    if (chip_thread) a.bits[(size_t)step * FE3_S + tid] = cm;
    {
        uint32_t cnt = (uint32_t)__popcll((unsigned long long)cm);
        for (int o = 32; o >= 1; o >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, o, AM_WAVE);
        const unsigned long long hm = __ballot(cm != 0u);
        if (lane == 0) {
            a.seg_cnt[(size_t)step * 2 + wv] = cnt;
            L.MASK[par * 4 + wv * 2] = (uint32_t)hm;
            if (wv == 0) L.MASK[par * 4 + 1] = (uint32_t)(hm >> 32);
        }
    }
    FE3_STAMP(7);
    fe3_barrier();                                                    // B5: chip masks of this step
    FE3_STAMP(8);
    // ---- sparse outputs: bb for the 17 chips from a candidate's chip on, avg for 2 -----------------------------------
    if (chip_thread && !(FE3_ABLATE & 1)) {
        // bit i of `win` = chip (tid - 31 + i) has a candidate, chips before this step come from the previous mask
        const uint32_t *cur = L.MASK + par * 4, *prv = L.MASK + (par ^ 1) * 4;
        const int w = tid >> 5, sh = tid & 31;
        const uint32_t hi = cur[w], lo = (w == 0) ? prv[2] : cur[w - 1];
        const unsigned long long both = ((unsigned long long)hi << 32) | lo;     // chips 32(w-1) .. 32(w+1)-1
        const uint32_t win = (uint32_t)(both >> (sh + 1));                       // bit 31 = own chip
        const bool want_bb = (win >> (32 - FE3_BBW)) != 0u;
        const bool want_avg = (win >> 30) != 0u;
        const bool inside = !edge || (jrun >= 0 && jrun + SPC <= a.out_n);
        if (want_bb && inside) {
            float4 *d = reinterpret_cast<float4 *>(a.bb_sparse + jrun);
#pragma unroll
            for (int k = 0; k < SPC / 4; ++k) {
                float4 t;
                t.x = x[4 * k]; t.y = x[4 * k + 1]; t.z = x[4 * k + 2]; t.w = x[4 * k + 3];
                d[k] = t;
            }
        }
        if (want_avg && inside) {
            float4 *d = reinterpret_cast<float4 *>(a.avg_sparse + jrun);
#pragma unroll
            for (int k = 0; k < SPC / 4; ++k) {
                float4 t;
                t.x = avgv[4 * k]; t.y = avgv[4 * k + 1]; t.z = avgv[4 * k + 2]; t.w = avgv[4 * k + 3];
                d[k] = t;
            }
        }
        if ((want_bb || want_avg) && !inside) {
            // ragged end of the stream: element by element
            const long long left = a.out_n - jrun;
            const int i1 = left >= SPC ? SPC : (left <= 0 ? 0 : (int)left);
            const int i0 = jrun >= 0 ? 0 : (jrun <= -SPC ? SPC : (int)(-jrun));
#pragma unroll
            for (int i = 0; i < SPC; ++i) {
                if (i >= i0 && i < i1) {
                    if (want_bb) a.bb_sparse[jrun + i] = x[i];
                    if (want_avg) a.avg_sparse[jrun + i] = avgv[i];
                }
            }
        }
    }
}

__global__ void __launch_bounds__(FE3_NT, FE3_WPS) am_k_fe3(am_fe3_args a)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem);
    fe3_smem L;
    L.raw = smem;
    L.X = reinterpret_cast<float *>(smem + (FE3_DMA ? FE3_RAWB : 0));
    L.TOT = L.X + FE3_CR * FE3_XS;
    L.RTOT = L.TOT + FE3_CR;
    L.PT = L.RTOT + FE3_CR;
    L.ST = L.PT + FE3_CR;
    L.SB0 = L.ST + FE3_CR;
    L.SB1 = L.SB0 + 64;
    L.MASK = reinterpret_cast<uint32_t *>(L.SB1 + 32);
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned raw_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
#else
    const unsigned raw_lds = 0;
#endif
    const int tid = threadIdx.x, lane = tid & (AM_WAVE - 1), wv = tid / AM_WAVE;
    const int sb = (int)(blockIdx.x * a.steps_per_wg);
    if (sb >= (int)a.nsteps) return;
    const int se = (sb + (int)a.steps_per_wg < (int)a.nsteps) ? sb + (int)a.steps_per_wg : (int)a.nsteps;

    fe3_thread T;
    {
        const int t = tid < FE3_S ? tid : 0;
        T.raw_addr = (unsigned)t * 256u + (((unsigned)t & 15u) << 4);          // = t*256 ^ swizzle (low 8 bits of t*256 are 0)
        // wave-instruction j = 4g + r of a wave moves LDS pieces [64 j', 64 j' + 64), j' = 12 wave + j: lane l is chip
        // 4 j' + (l >> 4), slot l & 15 of that chip, i.e. the chip's piece (l & 15) ^ ((4 r + (l >> 4)) & 15)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned k = ((unsigned)lane & 15u) ^ ((4u * (unsigned)r + ((unsigned)lane >> 4)) & 15u);
            T.dma_off[r] = ((unsigned)lane >> 4) * 256u + k * 16u;
        }
    }
    // rings start empty; the first step's chip 0 has no predecessor (its bb is never used); chips before the
    // segment count as "had candidates" so that the first 17 chips' bb is always written
    for (int i = tid; i < FE3_CR * FE3_XS + 4 * FE3_CR + 96; i += FE3_NT) L.X[i] = 0.0f;
    if (tid < 8) L.MASK[tid] = 0xFFFFFFFFu;
#if !FE3_DMA
    fe3_barrier();                                                    // (the first step stages into the ring right away)
#endif

    int step = sb - 1;                                                // the step before the segment rebuilds the rings
    bool fast = step >= a.raw_lo && step < a.raw_hi;
#if FE3_DMA
    if (fast)
        fe3_issue_dma(reinterpret_cast<const unsigned char *>(a.iq) + (size_t)((a.out_abs0 + (long long)step * FE3_T) - a.src_abs0) * 8,
                      L, T, raw_lds, wv, lane);
#else
    (void)raw_lds; (void)lane; (void)wv;
#endif
    int slot0 = 0, par = 0;
    fe3_prof PR;
#if defined(FE3_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    for (int k = 0; k < 12; ++k) PR.acc[k] = 0;
    PR.last = (long long)__builtin_readcyclecounter();
#endif
    for (; step < se; ++step) {
        const bool test = step >= sb;
        const bool next_fast = (step + 1 < se) && step + 1 >= a.raw_lo && step + 1 < a.raw_hi;
        const bool edge = !fast || (test && !(step >= a.test_lo && step < a.test_hi));
        FE3_STAMP(9);
#if FE3_DMA
        if (fast) fe3_dma_wait();
        else {
            fe3_barrier();                                            // (the staging buffer may still be read)
            fe3_fill_raw_guarded(a, L, a.out_abs0 + (long long)step * FE3_T, tid);
        }
#else
        if (fast) fe3_stage_step<false>(a, L, a.out_abs0 + (long long)step * FE3_T, slot0, tid);
        else fe3_stage_step<true>(a, L, a.out_abs0 + (long long)step * FE3_T, slot0, tid);
#endif
        FE3_STAMP(10);
        fe3_barrier();                                                // B1: raw of this step landed (all waves); LDS reuse
        FE3_STAMP(0);
        fe3_step(a, L, T, raw_lds, step, test, slot0, par, next_fast, edge, PR);
        fast = next_fast;
        slot0 = fe3_wrap_up(slot0 + FE3_S);
        par ^= 1;
    }
#if defined(FE3_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    if (a.prof && lane == 0)
        for (int k = 0; k < 12; ++k) a.prof[((size_t)blockIdx.x * 2 + wv) * 12 + k] = PR.acc[k];
#endif
}

// ---- host side ----------------------------------------------------------------------------------------
int am_fe3_supported(int spc) { return spc == FE3_SPC ? 1 : 0; }
unsigned am_fe3_tile(void) { return FE3_T; }
unsigned am_fe3_lag(void) { return FE3_LAG * FE3_SPC; }
unsigned am_fe3_steps(long long out_n) { return (unsigned)((out_n + FE3_LAG * FE3_SPC + FE3_T - 1) / FE3_T); }

static long long fe3_floor_div(long long x, long long d) { return x >= 0 ? x / d : -((-x + d - 1) / d); }
static long long fe3_ceil_div(long long x, long long d) { return -fe3_floor_div(-x, d); }

static int fe3_wgs_for_device()
{
    static int cached[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cached[dev] == 0) {
        hipDeviceProp_t prop;
        const int ncu = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                            ? prop.multiProcessorCount : 256;
        cached[dev] = FE3_WG_PER_CU * ncu;                            // resident workgroups (LDS)
    }
    return cached[dev];
}

hipError_t am_launch_fe3(const float *iq, long long src_abs0, long long src_abs1, long long out_abs0, long long out_n,
                         float *bb_sparse, float *avg_sparse, uint32_t j0, uint32_t j1, int use_pmf, float s1, float sL,
                         float thr_lin, uint32_t *bits, uint32_t *seg_cnt, unsigned *nsteps, hipStream_t s)
{
    am_fe3_args a;
    a.iq = iq; a.src_abs0 = src_abs0; a.src_abs1 = src_abs1; a.out_abs0 = out_abs0; a.out_n = out_n;
    a.bb_sparse = bb_sparse; a.avg_sparse = avg_sparse; a.j0 = j0; a.j1 = j1; a.bits = bits; a.seg_cnt = seg_cnt;
    a.use_pmf = (use_pmf && FE3_SPC > 1) ? 1 : 0; a.s1 = s1; a.sL = sL; a.thr_lin = thr_lin;
    a.nsteps = am_fe3_steps(out_n);
    *nsteps = a.nsteps;
    if (a.nsteps == 0) return hipSuccess;
    // steps served by DMA: samples [out_abs0 + k T, + T) inside [src_abs0, src_abs1), source 16-byte aligned (the
    // parity of the offset is the same for every step: T is even)
    const bool aligned = ((reinterpret_cast<uintptr_t>(iq) + (uintptr_t)(out_abs0 - src_abs0) * 8u) & 15u) == 0;
    auto clampi = [](long long v) { return (int)(v < -4 ? -4 : (v > 0x7FFFFFF0ll ? 0x7FFFFFF0ll : v)); };
    a.raw_lo = clampi(fe3_ceil_div(src_abs0 - out_abs0, FE3_T));
    a.raw_hi = aligned ? clampi(fe3_floor_div(src_abs1 - out_abs0, FE3_T)) : a.raw_lo;
    // steps whose tested positions [k T - 288, k T + T - 288) all lie in [j0, min(j1, out_n))
    const long long lag = (long long)FE3_LAG * FE3_SPC;
    const long long jhi = (long long)j1 < out_n ? (long long)j1 : out_n;
    a.test_lo = clampi(fe3_ceil_div((long long)j0 + lag, FE3_T));
    a.test_hi = clampi(fe3_floor_div(jhi + lag, FE3_T));
    // persistent workgroups: as many as are resident at once, each with a contiguous run of steps; short inputs
    // get at least 4 steps per workgroup (the ring rebuild costs one)
    const unsigned resident = (unsigned)fe3_wgs_for_device();
    unsigned spw = (a.nsteps + resident - 1) / resident;
    if (spw < 4) spw = 4;
    a.steps_per_wg = spw;
    const unsigned grid = (a.nsteps + spw - 1) / spw;
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
        hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void *>(&am_k_fe3),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)FE3_LDS_BYTES);
        if (rc != hipSuccess) return rc;
        if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
    a.prof = nullptr;
#if defined(FE3_PROFILE)
    // blocking; prints mean cycles per step and phase (wave 0 / wave 1) -- never in the default build
    if (hipMalloc(reinterpret_cast<void **>(&a.prof), (size_t)grid * 24 * sizeof(long long)) != hipSuccess) a.prof = nullptr;
#endif
    hipLaunchKernelGGL(am_k_fe3, dim3(grid), dim3(FE3_NT), FE3_LDS_BYTES, s, a);
    hipError_t lrc = hipGetLastError();
#if defined(FE3_PROFILE)
    if (a.prof) {
        std::vector<long long> h((size_t)grid * 24);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), a.prof, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        (void)hipFree(a.prof);
        static const char *names[12] = {"B1wait", "A1 raw+chains", "B2wait", "A2 pmf+totals+ring", "B3wait", "scan|B1 loads",
                                        "B4wait", "B2 avg+test", "B5wait", "sparse+loop", "dma wait", "-"};
        for (int w = 0; w < 2; ++w) {
            double acc[12] = {};
            for (unsigned b = 0; b < grid; ++b)
                for (int k = 0; k < 12; ++k) acc[k] += (double)h[((size_t)b * 2 + w) * 12 + k];
            const double steps = (double)grid * (double)(spw + 1);
            double tot = 0;
            for (int k = 0; k < 11; ++k) tot += acc[k];
            fprintf(stderr, "fe3 clocks/step wave %d (total %.0f):", w, tot / steps);
            // order of execution: 10 (dma wait) 0 (B1) 1 2 3 4 5 6 7 8 9
            static const int order[11] = {10, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9};
            for (int k = 0; k < 11; ++k) fprintf(stderr, " %s:%.0f", names[order[k]], acc[order[k]] / steps);
            fprintf(stderr, "\n");
        }
    }
#endif
    return lrc;
}
