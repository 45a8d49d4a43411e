// am_fe3.hip -- streaming fused front end + preamble detection for gfx950 at 32 samples per chip
// (64 Msps), the rate BASELINE.json's metric is quoted on.  Same results as am_k_fe2 / the oracle
// (canonical summation order), different machine mapping:
//
//   * PERSISTENT workgroups, six per CU (128 threads = 2 waves, 26 KB of LDS, 156 VGPRs: three waves per SIMD).  A workgroup owns a
//     contiguous segment of the stream and walks it in steps of 96 chips (two 48-chip blocks, 3072 samples).  What a
//     step needs from the past -- the pulse-matched power bb of the last 57 chips and their per-chip sums -- stays in
//     LDS rings, so nothing is loaded twice (the tile kernel re-read a 49-chip halo per tile and could not shrink its
//     tiles for that reason).
//   * raw IQ arrives by plain coalesced 16-byte loads with the streaming (nt) policy, 12 per thread, issued back to
//     back and waited for; |.|^2 goes straight into the ring slots of the step's chips.  No prefetch: while one
//     workgroup waits for its loads the other five of the CU compute.  (LDS-DMA staging and register prefetch -- also the
//     next step's loads under the current step's sparse outputs, which fit the registers -- were built and measured in
//     rounds 2, 4 and 5 (again once the bb rows had left the kernel), the default cache policy instead of nt in round 5:
//     DESIGN.md 5.1 / profiles/r4_valu / profiles/r5_fe64: none is faster; those variants are gone from the source.)
//   * thread = one chip (32 samples in registers).  The chip before it belongs to lane-1: its in-chip suffix sums come
//     over inside the addition (v_add_f32_dpp wave_ror:1); lane 63 stands in for the chip before the wave's first.
//   * THE KERNEL IS PRICED AGAINST HBM BUT WAS BOUND BY VALU ISSUE (a wave64 instruction holds the SIMD for four cycles;
//     ~85 % busy until round 4): every instruction per sample counts.  Address arithmetic is kept off the vector ALU (scalar
//     bases, immediate offsets, 24-bit multiplies, compare + select for the ring's wrap-around), products and sums of
//     neighbouring positions are packed (v_pk_mul_f32 / v_pk_add_f32: same rounding as the scalar forms).
//   * phase B (reference level + first-stage test) runs 9 chips BEHIND phase A, so the pulses 2, 7 and 9
//     chips ahead are already in the ring: no right halo, no redundant arithmetic.
//   * outputs are sparse: one candidate bit per position (a dense bitmap, 1/64 of the input bytes) and,
//     only around candidates, the runs of bb (17 chips: what am_k_cand_d reads) and of the reference level
//     (2 chips).  The dense 4 B/sample bb array of the tile kernel was a third of its HBM
//     traffic; burst extraction now recomputes its 240 chip-spaced samples from IQ (am_kernels.hip).
//     Plus one number per workgroup: the largest bb of its segment (+inf if any is not finite), the bound the
//     refinement's exact energy-difference test needs (am_k_cand_d).
//
// Reference: python/rx_path.py:38-54 (|.|^2, moving averages), lib/preamble_impl.cc:172-179 (test).
#include "am_fe_stream.h"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <type_traits>
#include <vector>

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#include "am_fe_cmpx.h"

// tuning builds only (tools/build_variants.sh): FE3_ABLATE bit mask removes parts of the kernel -- results INVALID
// pieces of the ring-rebuilding step that are not loaded (8 chips each): chips from FE3_S - FE3_LAG - 48 on are needed
#define FE3_WARM_J0 ((FE3_S - FE3_LAG - AM_CHIPS_AVG) / (FE3_NT / 16))
#ifndef FE3_ABLATE
#define FE3_ABLATE 0
#endif
#ifndef FE3_WPS
#define FE3_WPS 3                         /* launch bound, waves per SIMD (<= 168 VGPRs; it uses 156)  */
#endif
#ifndef FE3_WG_PER_CU
#define FE3_WG_PER_CU 6                   /* resident workgroups per CU (26 KB of LDS each: the limit; 2 waves of <= 168 VGPRs) */
#endif
#define FE3_SPC 32
#ifndef FE3_NW
#define FE3_NW 2                          /* waves per workgroup = 48-chip blocks per step              */
#endif
#define FE3_S (AM_CHIPS_AVG * FE3_NW)     /* chips per step: one 48-chip block per wave                 */
#define FE3_NT (AM_WAVE * FE3_NW)         /* lanes 0..47 of a wave = its block's chips; all threads stage the loads */
#define FE3_T (FE3_S * FE3_SPC)           /* samples per step                                          */
#define FE3_LAG 9                         /* phase B runs this many chips behind phase A               */
#ifndef FE3_CR_EXTRA
#define FE3_CR_EXTRA 0
#endif
#define FE3_CR (FE3_S + FE3_LAG + AM_CHIPS_AVG + 1 + FE3_CR_EXTRA)   /* ring capacity in chips: a step + lag + 48 back + 1 */
#define FE3_XS 36                         /* floats per ring chip: 32 + 4 pad (16-byte reads of consecutive chips hit all banks) */
#define FE3_BBW 17                        /* chips of bb kept from a candidate's chip on (am_k_cand reads up to pos + 16*spc) */

struct am_fe3_args {
    const float *iq;
    long long src_abs0, src_abs1;         // absolute range of samples present in iq
    long long out_abs0;                   // absolute index of array coordinate 0 (multiple of 48*spc)
    long long out_n;                      // array coordinates with data
    float *bb_sparse;                     // bb runs around candidates (array coordinates)
    float *avg_sparse;                    // reference-level runs around candidates
    uint32_t j0, j1;                      // positions whose preamble test is wanted
    uint32_t *bits;                       // [nsteps * 96] candidate words: bit b of word w = position w*32 + b - 288
    uint32_t *wg_cnt;                     // [grid] candidates a workgroup found (am_k_gather_wg lays the flat list out from these)
    float *wg_max;                        // [grid] largest bb a workgroup formed (+inf if one was not finite)
    unsigned nsteps;                      // steps (= bitmap tiles) of the whole launch
    unsigned steps_per_wg;                // ... of each of the first n_long workgroups; the others take one fewer (levelled segments)
    unsigned n_long;
    // steps whose raw samples are all present and 16-byte aligned are loaded without guards: [raw_lo, raw_hi);
    // steps whose tested positions are all wanted need no range mask: [test_lo, test_hi)
    int raw_lo, raw_hi, test_lo, test_hi;
    int use_pmf;
    float s1, sL, thr_lin;
    long long *prof;                      // profiling builds: [grid * 2 waves][12] cycles per phase, else null
};

// ---- small device helpers -------------------------------------------------------------------------
// ring slot arithmetic (operands within one ring length of the valid range)
__device__ __forceinline__ int fe3_wrap_up(int s) { return s >= FE3_CR ? s - FE3_CR : s; }     // s in [0, 2*CR)
__device__ __forceinline__ int fe3_wrap_dn(int s) { return s < 0 ? s + FE3_CR : s; }           // s in [-CR, CR)

// 16-byte store of a sparse output row piece
__device__ __forceinline__ void fe3_gstore16(float *p, const float4 &u) { *reinterpret_cast<float4 *>(p) = u; }

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
    float *X;                 // [FE3_CR * FE3_XS] ring: |.|^2 of a step's chips while it is staged, then bb
    float *RTOT, *PT, *ST;    // [FE3_CR] per chip: right->left sum of its bb; in-block exclusive prefix / suffix of the chip totals
    float *MP;                // [2][32] |.|^2 of a step's last chip (by step parity): the chip before the next step's first
    float *M47;               // [32] |.|^2 of chip 47 (the chip before wave 1's first)
    uint32_t *CARRY;          // [2] chips at the start of the next step whose bb must be written (bit mask, by step parity)
    uint32_t *TAB;            // [2][64] per wave: lane of the r-th chip whose bb / reference level is written
    float *AVS;               // [8 * FE3_XS] eight chips of wave 1's reference level on their way out (wave 0 parks its rows in the ring)
};
#define FE3_LDS_BYTES (FE3_CR * FE3_XS * 4 + (64 + 32 * (FE3_NW - 1)) * 4 + FE3_NW * 4 * FE3_XS * 4 + 3 * FE3_CR * 4 + 2 * 4 + FE3_NW * 64 * 4)

struct fe3_raw { float4 v[12]; };      // a thread's 12 pieces of a step's raw IQ (piece tid + 128 j)

// unguarded loads of a step (wave-uniform 64-bit base + 32-bit lane offset): issued back to back, consumed by fe3_store_step
template <int J0 = 0>
__device__ __forceinline__ void fe3_load_step(const am_fe3_args &a, long long A0, int tid, fe3_raw &r)
{
    const unsigned char *gb = reinterpret_cast<const unsigned char *>(a.iq) + (size_t)(A0 - a.src_abs0) * 8;
    const unsigned off = (unsigned)tid * 16u;
#pragma unroll
    for (int j = J0; j < 12; ++j) {
        // scalar base per pair of pieces + the lane's 32-bit offset (+ 2 KB as the instruction's immediate): the address
        // arithmetic stays on the scalar unit
        unsigned long long g = reinterpret_cast<unsigned long long>(gb + (size_t)(j & ~1) * (FE3_NT * 16u));
#if defined(__HIP_DEVICE_COMPILE__)
        asm("" : "+s"(g));
#endif
        r.v[j] = fes_gload16_at(g, off + (unsigned)(j & 1) * (FE3_NT * 16u));
    }
}

// where a thread's pieces go: piece j = samples 2k, 2k+1 (k = tid & 15) of the step's chip c0 + 8 j (c0 = tid >> 4), ring
// slot slot0 + c0 + 8 j, wrapped from piece jw on -- one compare and one select per piece, the rest is the
// instruction's immediate offset (the compiler's form of the same arithmetic was ~20 VALU instructions per piece, three of
// them quarter-rate multiplies: a third of the kernel's VALU time)
struct fe3_dest { float *row, *roww; int jw; bool last8; float *m47, *mp; };
static_assert(FE3_SPC == 32 && FE3_NT / 16 == 8 && AM_CHIPS_AVG % 8 == 0, "eight chips per round of pieces");
__device__ __forceinline__ fe3_dest fe3_dest_of(const fe3_smem &L, int slot0, int par, int tid)
{
    const int c0 = tid >> 4, k = tid & 15;
    const int sc = slot0 + c0;                                        // < FE3_CR + 8
    fe3_dest d;
    d.jw = (FE3_CR + 7 - sc) >> 3;                                    // slot0 + c0 + 8 j >= FE3_CR  <=>  j >= jw
    d.row = L.X + fes_mul24(sc, FE3_XS) + 2 * k;
    d.roww = d.row - FE3_CR * FE3_XS;
    d.last8 = c0 == 7;                                                // this thread's chips are 7, 15, ... : 47 and 95 among them
    d.m47 = L.M47 + 2 * k;
    d.mp = L.MP + par * 32 + 2 * k;
    return d;
}
__device__ __forceinline__ void fe3_store_piece(const fe3_dest &d, int j, float2 mm)
{
    float *dst = (j >= d.jw) ? d.roww : d.row;
    *reinterpret_cast<float2 *>(dst + j * (8 * FE3_XS)) = mm;
    // the chip before a later wave's first is stored a second time (M47: that wave needs it after the chip's own slot holds
    // bb), and so is the step's last chip (MP: the next step's wave 0 needs it)
    if ((8 * j + 8) % AM_CHIPS_AVG == 0 && d.last8) {
        if (8 * j + 8 < FE3_S) *reinterpret_cast<float2 *>(d.m47 + ((8 * j + 8) / AM_CHIPS_AVG - 1) * 32) = mm;
        else *reinterpret_cast<float2 *>(d.mp) = mm;
    }
}
template <int J0 = 0>
__device__ __forceinline__ void fe3_store_step(const fe3_smem &L, int slot0, int par, int tid, const fe3_raw &r)
{
    const fe3_dest d = fe3_dest_of(L, slot0, par, tid);
#pragma unroll
    for (int j = J0; j < 12; ++j) {
        const fes_f2 q0 = fes_pk_mul(fes_mk2(r.v[j].x, r.v[j].y), fes_mk2(r.v[j].x, r.v[j].y));
        const fes_f2 q1 = fes_pk_mul(fes_mk2(r.v[j].z, r.v[j].w), fes_mk2(r.v[j].z, r.v[j].w));
        float2 mm;
        mm.x = q0.x + q0.y;                                           // a1: fl(fl(I*I) + fl(Q*Q))
        mm.y = q1.x + q1.y;
        fe3_store_piece(d, j, mm);
    }
}

// |iq|^2 of one step straight into the ring slots of its chips (they hold chips nobody needs any more): piece
// p = tid + 128 j (16 bytes = samples 2k, 2k+1 of chip p >> 4, k = p & 15 = tid & 15) -> X[slot][2k .. 2k+1].
// Coalesced loads (consecutive lanes, consecutive pieces), 8-byte LDS stores (16 lanes = one chip's 128 bytes).
// J0 > 0: only pieces J0.. (chips 8 J0 ..) -- the step that rebuilds the rings needs its last 57 chips only
template <bool GUARD, int J0 = 0>
__device__ __forceinline__ void fe3_stage_step(const am_fe3_args &a, const fe3_smem &L, long long A0, int slot0, int par, int tid)
{
    if constexpr (GUARD) {
        // stream edges / unaligned input: one piece at a time, zeros outside the stream (rare: kept small)
        const float2 *iq2 = reinterpret_cast<const float2 *>(a.iq);
        const fe3_dest d = fe3_dest_of(L, slot0, par, tid);
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
            fe3_store_piece(d, j, mm);
        }
        return;
    }
    fe3_raw v;
    fe3_load_step<J0>(a, A0, tid, v);
    fe3_store_step<J0>(L, slot0, par, tid, v);
}

// One step (its |.|^2 is staged).
//   step     global step index (may be -1: history before the first wanted block)
//   test     false for a workgroup's first step (it only rebuilds the rings from the previous segment's tail)
//   slot0    ring slot of this step's chip 0
//   edge     (uniform) the step touches the end of the stream or positions that are not wanted
__device__ __forceinline__ void fe3_step(const am_fe3_args &a, const fe3_smem &L, const int step, const bool test,
                                         const int slot0, const int par, const bool edge, const int tid, float &mxrun,
                                         bool &badrun, uint32_t &ncand, fe3_prof &PR)
{
    constexpr int SPC = FE3_SPC;
    const int lane = tid & (AM_WAVE - 1);
#if defined(__HIP_DEVICE_COMPILE__)
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);          // (wave-uniform, and known to be: scalar arithmetic, uniform branches)
#else
    const int wv = tid >> 6;
#endif
    const bool chip_thread = lane < AM_CHIPS_AVG;
    const int lc = chip_thread ? lane : AM_CHIPS_AVG - 1;             // chip inside the wave's block (spare lanes shadow the last one, never write)
    const int t = fes_mul24(wv, AM_CHIPS_AVG) + lc;                   // chip of the step
    const long long A0 = a.out_abs0 + (long long)step * FE3_T;      // absolute index of the step's first sample
    const int slotA = fe3_wrap_up(slot0 + t);
    const int offA = fes_mul24(slotA, FE3_XS);
    const bool do_pmf = a.use_pmf != 0;

    // ---- phase A: pulse matched filter of the own chip, chip totals, in-block scans -------------------------------
    float bb[SPC];
    {
        float m[SPC];
        // lane 63 stands in for the chip before the wave's first one (wave 0: the previous step's last chip, wave 1: chip
        // 47 -- their |.|^2 was stored a second time, the ring slots hold bb by now): lane 0 takes its suffix sums from
        // there with the same lane rotation that hands every other lane its left neighbour's
        const float *mrow = L.X + offA;
        if (lane == AM_WAVE - 1) mrow = (wv == 0) ? (L.MP + (par ^ 1) * 32) : (L.M47 + (wv - 1) * 32);
        const float4 *mp = reinterpret_cast<const float4 *>(mrow);
#pragma unroll
        for (int k = 0; k < SPC / 4; ++k) {
            const float4 u = mp[k];
            m[4 * k] = u.x; m[4 * k + 1] = u.y; m[4 * k + 2] = u.z; m[4 * k + 3] = u.w;
        }
        if (do_pmf) {
            // suffix sums of the own chip (what the NEXT chip's filter needs): sx[i] = m[31] + ... + m[i], right->left
            // (the in-chip prefix sums pp[] run left->right through the same loop: two independent chains per iteration)
            float sx[SPC], pp[SPC];
            {
                float as = 0.0f, ap = 0.0f;
#pragma unroll
                for (int i = 0; i < SPC; ++i) {
                    as = as + m[SPC - 1 - i]; sx[SPC - 1 - i] = as;
                    ap = ap + m[i]; pp[i] = ap;
                }
            }
            float tt[SPC];
#pragma unroll
            for (int i = 0; i < SPC - 1; ++i) tt[i] = fes_from_prev_lane_ror(sx[i + 1], lane) + pp[i];
            tt[SPC - 1] = pp[SPC - 1];                                // the window is the chip
#pragma unroll
            for (int i = 0; i < SPC; i += 2) {
                const fes_f2 v = fes_pk_mul(fes_mk2(tt[i], tt[i + 1]), fes_mk2(a.s1, a.s1));
                bb[i] = v.x; bb[i + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int i = 0; i < SPC; ++i) bb[i] = m[i];
        }
    }
    if (edge) {
        // positions beyond the end of the stream read as zero (the preamble view pads with zeros)
        const long long left = a.src_abs1 - (A0 + (long long)t * SPC);    // samples of this chip inside the stream
        const int nin = left >= SPC ? SPC : (left <= 0 ? 0 : (int)left);
#pragma unroll
        for (int i = 0; i < SPC; ++i)
            if (i >= nin) bb[i] = 0.0f;
    }
    {
        // chip totals in both directions (canonical level-1 sums); spare lanes contribute zeros to the scans
        float f = 0.0f, b = 0.0f, mx = 0.0f;
#pragma unroll
        for (int i = 0; i < SPC; ++i) { f = f + bb[i]; b = b + bb[SPC - 1 - i]; mx = fmaxf(mx, bb[i]); }
        if (!chip_thread) { f = 0.0f; mx = 0.0f; }                    // (lane 63 formed something that is no chip's bb)
        mxrun = fmaxf(mxrun, mx);
        badrun = badrun || !(f < __builtin_inff());                   // a sample that is not finite makes its chip's total so (all terms >= 0 or NaN)
        // exclusive prefix / suffix of the 48 chip totals of this wave's block, strictly sequential (canonical
        // order): x <- x(lane-1) + f repeated 47 times leaves lane j with ((f0 + f1) + ...) + fj (a lane's value is
        // final after j rounds and is recomputed identically afterwards); the same right->left from lane 47
        float xs = f, ys = f;
        if (!(FE3_ABLATE & 8)) {
#pragma unroll
            for (int r = 0; r < AM_CHIPS_AVG - 1; ++r) {
                xs = fes_from_prev_lane(xs, 0.0f, lane) + f;
                ys = fes_from_next_lane(ys, 0.0f, lane) + f;
            }
        }
        const float pt = fes_from_prev_lane(xs, 0.0f, lane);
        const float st = fes_from_next_lane(ys, 0.0f, lane);
        if (chip_thread) {
            L.RTOT[slotA] = b;
            L.PT[slotA] = pt;
            L.ST[slotA] = st;
            float4 *xp = reinterpret_cast<float4 *>(L.X + offA);
#pragma unroll
            for (int k = 0; k < SPC / 4; ++k) {
                float4 u;
                u.x = bb[4 * k]; u.y = bb[4 * k + 1]; u.z = bb[4 * k + 2]; u.w = bb[4 * k + 3];
                xp[k] = u;
            }
        }
    }
    FE3_STAMP(1);
    fes_barrier();                                                    // B3: ring, totals and scans of this step complete
    FE3_STAMP(2);
    if (!test) return;                                                // (uniform) ring rebuild only

    // ---- phase B on chip q = (this thread's phase-A chip) - 9: reference level (a4) + first-stage test (a6) --------
    // (ring offsets by add / compare / select from the phase-A chip's: slot * 36 is a multiply only once per step)
    const bool lowA = slotA < FE3_LAG;
    const int slotB = slotA - FE3_LAG + (lowA ? FE3_CR : 0);
    const int offB = offA - FE3_LAG * FE3_XS + (lowA ? FE3_CR * FE3_XS : 0);
    const bool lowB = slotB < AM_CHIPS_AVG;
    const int slotS = slotB - AM_CHIPS_AVG + (lowB ? FE3_CR : 0);     // the chip 48 chips back
    const int offS = offB - AM_CHIPS_AVG * FE3_XS + (lowB ? FE3_CR * FE3_XS : 0);
    float x[SPC], avgv[SPC];
    float nxt;
    {
        float scv[SPC];
        const float4 *xp = reinterpret_cast<const float4 *>(L.X + offB);
        const float4 *sp = reinterpret_cast<const float4 *>(L.X + offS);
#pragma unroll
        for (int k = 0; k < SPC / 4; ++k) {
            const float4 u = xp[k];
            x[4 * k] = u.x; x[4 * k + 1] = u.y; x[4 * k + 2] = u.z; x[4 * k + 3] = u.w;
        }
#pragma unroll
        for (int k = 0; k < SPC / 4; ++k) {
            const float4 u = sp[k];
            scv[4 * k] = u.x; scv[4 * k + 1] = u.y; scv[4 * k + 2] = u.z; scv[4 * k + 3] = u.w;
        }
        const bool topB = slotB == FE3_CR - 1;
        nxt = L.X[offB + FE3_XS - (topB ? FE3_CR * FE3_XS : 0)];
        const int slotS1 = slotS + 1 - ((slotS == FE3_CR - 1) ? FE3_CR : 0);
        const float pt = L.PT[slotB];
        const float st_a = L.ST[slotS];
        const float suf_last = L.RTOT[slotS1] + L.ST[slotS1];
        const int jb = lc - FE3_LAG + ((lc < FE3_LAG) ? AM_CHIPS_AVG : 0);   // chip index inside its 48-chip block
        // in-chip suffix sums of the chip 48 back (right->left) and prefix sums of the own chip (left->right): two
        // independent chains per iteration
        float ap_[SPC];
        {
            float as = 0.0f, ap = 0.0f;
#pragma unroll
            for (int i = 0; i < SPC; ++i) {
                as = as + scv[SPC - 1 - i]; scv[SPC - 1 - i] = as;
                ap = ap + x[i]; ap_[i] = ap;
            }
        }
        // s[i] = (scv[i + 1] + st_a) + (pt + ap[i]); the chip's last position: (suf_last + pre), or pre alone where the
        // window ends with the block (x + (-0) == x for every x, NaN included) -- two positions per instruction
        const float q32 = (jb == AM_CHIPS_AVG - 1) ? -0.0f : suf_last;
#pragma unroll
        for (int i = 0; i < SPC; i += 2) {
            const fes_f2 pre = fes_pk_add(fes_mk2(pt, pt), fes_mk2(ap_[i], ap_[i + 1]));
            fes_f2 q;
            if (i + 2 < SPC) q = fes_pk_add(fes_mk2(scv[i + 1], scv[i + 2]), fes_mk2(st_a, st_a));
            else q = fes_mk2(scv[i + 1] + st_a, q32);
            const fes_f2 sv = fes_pk_add(q, pre);
            const fes_f2 av = fes_pk_mul(sv, fes_mk2(a.sL, a.sL));
            avgv[i] = av.x; avgv[i + 1] = av.y;
        }
    }
    // array coordinate of x[0]
    const long long jstep = (long long)step * FE3_T - (long long)(FE3_LAG * SPC);
    const long long jrun = jstep + (long long)(t * SPC);
    uint32_t cm = 0u;
    {
        constexpr int CH = 16;
        // the later pulses' chips: 2, 7 and 9 chips on = 7, 2 and 0 chips before the phase-A chip
        const int off9 = offA;
        const int off7 = offA - 2 * FE3_XS + ((slotA < 2) ? FE3_CR * FE3_XS : 0);
        const int off2 = offA - 7 * FE3_XS + ((slotA < 7) ? FE3_CR * FE3_XS : 0);
#if defined(FE2_CMPX)
        auto pass = [&](auto hc) __attribute__((always_inline)) {
            constexpr int H = decltype(hc)::value;
            float thr[CH];
#pragma unroll
            for (int i = 0; i < CH; i += 2) {
                const fes_f2 th = fes_pk_mul(fes_mk2(avgv[H + i], avgv[H + i + 1]), fes_mk2(a.thr_lin, a.thr_lin));   // preamble_impl.cc:173
                thr[i] = th.x; thr[i + 1] = th.y;
            }
            uint32_t part = 0u;
            fe2_peak8<H>(part, &x[H], x[H + 8], &thr[0]);
            fe2_peak8<H + 8>(part, &x[H + 8], (H + 16 < SPC) ? x[(H + 16 < SPC) ? H + 16 : 0] : nxt, &thr[8]);
            // the three later pulses must not be below the threshold (:177-179): one test on the smallest
            // (v_min3 ignores a NaN operand exactly as `NaN < thr` is false); only where some lane has a survivor
            if (!(FE3_ABLATE & 2) && __ballot(part != 0u) != 0ull) {
#pragma unroll
                for (int g = 0; g < CH; g += 8) {                     // (eight positions at a time: 24 registers in flight, not 48)
                    float t2[8], t7[8], t9[8];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const float4 u = reinterpret_cast<const float4 *>(L.X + off2 + H + g)[k];
                        const float4 v = reinterpret_cast<const float4 *>(L.X + off7 + H + g)[k];
                        const float4 w = reinterpret_cast<const float4 *>(L.X + off9 + H + g)[k];
                        t2[4 * k] = u.x; t2[4 * k + 1] = u.y; t2[4 * k + 2] = u.z; t2[4 * k + 3] = u.w;
                        t7[4 * k] = v.x; t7[4 * k + 1] = v.y; t7[4 * k + 2] = v.z; t7[4 * k + 3] = v.w;
                        t9[4 * k] = w.x; t9[4 * k + 1] = w.y; t9[4 * k + 2] = w.z; t9[4 * k + 3] = w.w;
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) t2[i] = fminf(fminf(t2[i], t7[i]), t9[i]);
                    if (g == 0) fe2_weak8<H>(part, &t2[0], &thr[0]);
                    else fe2_weak8<H + 8>(part, &t2[0], &thr[8]);
                }
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
                    const float weakest = fminf(fminf(L.X[off2 + h + i], L.X[off7 + h + i]), L.X[off9 + h + i]);
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
    // candidate word; the lane's running count (summed over the workgroup at the end)
    if (chip_thread) a.bits[(size_t)step * FE3_S + t] = cm;
    ncand += (uint32_t)__popcll((unsigned long long)cm);
    const unsigned long long cand = __ballot(cm != 0u);               // bit l: chip 48 wave + l has a candidate
    FE3_STAMP(3);
    if (FE3_ABLATE & 1) return;
    // ---- sparse outputs ---------------------------------------------------------------------------------------------
    // reference level: the chip of a candidate and the one after it (a wave's lane 0 cannot see the chip before it:
    // always).  The values exist only in registers: the flagged lanes park them in LDS rows and the wave writes them out
    // with coalesced stores like bb below, eight rows per instruction.  Where they park: wave 0 in the ring rows of the
    // chips 57 .. 10 before the step -- its own phase B (behind it in program order) was their last reader, wave 1 never
    // reads them, and the next step's staging overwrites them after barrier B5 -- all of its rows at once; wave 1 in the
    // eight rows of the small buffer, eight at a time.  (Round 4: both waves went through four buffer rows at a time,
    // 8 LDS writes with four active lanes per four rows: 8.5 us of the kernel at the stress density.)
    if (!(FE3_ABLATE & 16)) {
        static_assert(FE3_NW == 2 && FE3_CR >= FE3_S + FE3_LAG + AM_CHIPS_AVG + 1, "who may reuse which ring rows");
        const unsigned long long wa = (cand | (cand << 1) | 1ull) & ((1ull << AM_CHIPS_AVG) - 1ull);
        const int nav = __popcll(wa);
        uint32_t *tab = L.TAB + wv * AM_WAVE;
        const bool mine = ((wa >> lane) & 1ull) != 0ull;
        const int my_rank = __popcll(wa & ((1ull << lane) - 1ull));
        if (mine) tab[my_rank] = (uint32_t)lane;
        float *const dst = a.avg_sparse + jstep;
        const long long lo64 = -jstep, hi64 = a.out_n - jstep;
        const int lo = lo64 <= 0 ? 0 : (lo64 > 0x7FFFFFF ? 0x7FFFFFF : (int)lo64);
        const int hi = hi64 <= 0 ? 0 : (hi64 > 0x7FFFFFF ? 0x7FFFFFF : (int)hi64);
        const int sub = lane >> 3, piece = lane & 7;
        const int RB = (wv == 0) ? AM_CHIPS_AVG : 2 * 4;              // (uniform) rows parked per batch
        // row r of a batch: wave 0 -> ring slot of chip r - 57 of the step; wave 1 -> buffer row r
        auto park_row = [&](int r) __attribute__((always_inline)) -> float * {
            if (wv == 0) {
                int slot = slot0 - (FE3_LAG + AM_CHIPS_AVG) + r;      // >= -57, < FE3_CR
                slot += (slot < 0) ? FE3_CR : 0;
                return L.X + fes_mul24(slot, FE3_XS);
            }
            return L.AVS + fes_mul24(r, FE3_XS);
        };
        for (int b0 = 0; b0 < nav; b0 += RB) {                        // (uniform trip count: one batch in wave 0)
            if (mine && my_rank >= b0 && my_rank < b0 + RB) {
                float4 *d = reinterpret_cast<float4 *>(park_row(my_rank - b0));
#pragma unroll
                for (int k = 0; k < SPC / 4; ++k) {
                    float4 u;
                    u.x = avgv[4 * k]; u.y = avgv[4 * k + 1]; u.z = avgv[4 * k + 2]; u.w = avgv[4 * k + 3];
                    d[k] = u;
                }
            }
            __builtin_amdgcn_wave_barrier();
            const int bend = (b0 + RB < nav) ? b0 + RB : nav;
            for (int r0 = b0; r0 < bend; r0 += 8) {                   // (uniform)
                const int r = r0 + sub;
                if (r < bend) {
                    const int tc = fes_mul24(wv, AM_CHIPS_AVG) + (int)tab[r];
                    const float4 u = *reinterpret_cast<const float4 *>(park_row(r - b0) + 4 * piece);
                    const int rel = (tc << 5) + 4 * piece;
                    if (!edge || (rel >= lo && rel + 4 <= hi)) fe3_gstore16(dst + rel, u);
                    else {
                        if (rel >= lo && rel < hi) dst[rel] = u.x;
                        if (rel + 1 >= lo && rel + 1 < hi) dst[rel + 1] = u.y;
                        if (rel + 2 >= lo && rel + 2 < hi) dst[rel + 2] = u.z;
                        if (rel + 3 >= lo && rel + 3 < hi) dst[rel + 3] = u.w;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    // bb: the 17 chips from a candidate's chip on, copied from the ring (phase A is 9 chips ahead: chips up to test index
    // 104 are there) with fully coalesced stores: the flagged chips are ranked, and every wave-instruction moves eight of
    // them, 8 lanes x 16 bytes = one 128-byte line each.  (A lane storing its own chip's 128 bytes from registers issues
    // 8 stores that touch one line per lane: store-issue bound, measured 5x slower.)  A wave's mask covers 64 chips from
    // its first one: wave 0 thereby serves the first 16 chips of wave 1 where its own candidates reach; what wave 1's
    // candidates need beyond the step is handed to the next step's wave 0 (CARRY).
    // (round 5: at 64 Msps the launcher passes no bb array -- the rows are formed from the samples by the kernel that lists the
    // candidates, am_k_gather_wg<1>, am_kernels.hip: ~42 MB of stores per launch less here.  The block stays for callers that
    // still hand one in.)
    if (!(FE3_ABLATE & 32) && a.bb_sparse != nullptr) {                // (uniform)
        unsigned long long need = cand;                               // dilate by 16 chips to the right
        need |= need << 1; need |= need << 2; need |= need << 4; need |= need << 8;
        need |= cand << 16;
        if (wv == 0) need |= (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)L.CARRY[par ^ 1]);
        if (wv == FE3_NW - 1) {
            if (lane == 0) L.CARRY[par] = (uint32_t)(need >> AM_CHIPS_AVG) & 0xFFFFu;
            need &= (1ull << AM_CHIPS_AVG) - 1ull;
        }
        const int nflag = __popcll(need);
        uint32_t *tab = L.TAB + wv * AM_WAVE;
        __builtin_amdgcn_wave_barrier();                              // (the table is reused: the reference level's reads come first)
        if ((need >> lane) & 1ull) tab[__popcll(need & ((1ull << lane) - 1ull))] = (uint32_t)lane;
        __builtin_amdgcn_wave_barrier();                              // (one wave: its LDS accesses execute in order)
        float *const dst = a.bb_sparse + jstep;                       // array coordinate of test index 0 (may lie before the array)
        const long long lo64 = -jstep, hi64 = a.out_n - jstep;        // elements [lo, hi) of this step's coordinates exist
        const int lo = lo64 <= 0 ? 0 : (lo64 > 0x7FFFFFF ? 0x7FFFFFF : (int)lo64);
        const int hi = hi64 <= 0 ? 0 : (hi64 > 0x7FFFFFF ? 0x7FFFFFF : (int)hi64);
        const int sub = lane >> 3, piece = lane & 7;
        for (int r0 = 0; r0 < nflag; r0 += 8) {                       // (uniform trip count)
            const int r = r0 + sub;
            if (r < nflag) {
                const int tc = fes_mul24(wv, AM_CHIPS_AVG) + (int)tab[r];       // test index of the chip, <= 103
                int slot = slot0 + tc - FE3_LAG;                      // in [-9, 2 CR)
                slot += (slot < 0) ? FE3_CR : ((slot >= FE3_CR) ? -FE3_CR : 0);
                const float4 u = *reinterpret_cast<const float4 *>(L.X + fes_mul24(slot, FE3_XS) + 4 * piece);
                const int rel = (tc << 5) + 4 * piece;
                if (!edge || (rel >= lo && rel + 4 <= hi)) fe3_gstore16(dst + rel, u);
                else {
                    if (rel >= lo && rel < hi) dst[rel] = u.x;
                    if (rel + 1 >= lo && rel + 1 < hi) dst[rel + 1] = u.y;
                    if (rel + 2 >= lo && rel + 2 < hi) dst[rel + 2] = u.z;
                    if (rel + 3 >= lo && rel + 3 < hi) dst[rel + 3] = u.w;
                }
            }
        }
    }
}

__global__ void __launch_bounds__(FE3_NT, FE3_WPS) am_k_fe3(am_fe3_args a)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem);
    fe3_smem L;
    // (arrays read or written 16 bytes at a time first: their sizes are multiples of 16 bytes)
    L.X = reinterpret_cast<float *>(smem);
    L.MP = L.X + FE3_CR * FE3_XS;
    L.M47 = L.MP + 64;
    L.AVS = L.M47 + 32 * (FE3_NW - 1);
    L.RTOT = L.AVS + FE3_NW * 4 * FE3_XS;
    L.PT = L.RTOT + FE3_CR;
    L.ST = L.PT + FE3_CR;
    L.CARRY = reinterpret_cast<uint32_t *>(L.ST + FE3_CR);
    L.TAB = L.CARRY + 2;
    const int tid0 = threadIdx.x;
    // levelled segments (round 6): the first n_long workgroups take steps_per_wg steps, the others one fewer -- 1 536 segments of 13 or
    // 14 steps at 64 M samples instead of 1 489 of 14, so that no CU is left with five workgroups
    const bool longseg = blockIdx.x < a.n_long;
    const int sb = longseg ? (int)(blockIdx.x * a.steps_per_wg)
                           : (int)(a.n_long * a.steps_per_wg + (blockIdx.x - a.n_long) * (a.steps_per_wg - 1u));
    if (sb >= (int)a.nsteps) return;
    const int mine = (int)a.steps_per_wg - (longseg ? 0 : 1);
    const int se = (sb + mine < (int)a.nsteps) ? sb + mine : (int)a.nsteps;

    // rings start empty; the first step's chip 0 has no predecessor (its bb is never used); the bb of the first 16
    // chips of a segment is always written (the candidates of the previous segment's tail are not known here)
    for (int i = tid0; i < FE3_CR * FE3_XS + 64 + 32 * (FE3_NW - 1) + FE3_NW * 4 * FE3_XS + 3 * FE3_CR; i += FE3_NT) L.X[i] = 0.0f;
    if (tid0 < 2) L.CARRY[tid0] = 0xFFFFu;
    fes_barrier();                                                    // (the first step stages into the ring right away)

    int slot0 = 0, par = 0;
    fe3_prof PR;
#if defined(FE3_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    for (int k = 0; k < 12; ++k) PR.acc[k] = 0;
    PR.last = (long long)__builtin_readcyclecounter();
    PR.acc[8] = (long long)wall_clock64();                            // the workgroup's timeline (100 MHz): start,
    // where it runs: HW_ID (register 4: cu_id [11:8], sh_id [12], se_id [15:13]) and XCC_ID (register 20, [3:0])
    PR.acc[7] = (long long)(((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                            (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4));
#endif
    float mxrun = 0.0f;                                               // largest bb this thread has formed
    bool badrun = false;                                              // ... or one that is not finite
    uint32_t ncand = 0;                                               // candidates this thread's chips held
    for (int step = sb - 1; step < se; ++step) {                      // the step before the segment rebuilds the rings
        const bool test = step >= sb;
        const bool have = step >= a.raw_lo && step < a.raw_hi;        // the step's raw samples are all present and 16-byte aligned
        const bool edge = !have || (test && !(step >= a.test_lo && step < a.test_hi));
        int tid = tid0;                                               // (nothing that follows from the thread index is to live across iterations:
#if defined(__HIP_DEVICE_COMPILE__)                                   //  hoisted out of the loop those values cost 15 VGPRs of 168)
        asm volatile("" : "+v"(tid));
#endif
        fes_step_priority((unsigned)(step - sb + 1));                   // (am_fe_stream.h: the CU's workgroups end together)
        FE3_STAMP(4);
        // (the ring slots about to be staged were read by the previous step's phase B: its last barrier is behind us)
        // load, wait, stage.  The step before the segment only feeds the rings: the first chip tested is chip
        // FE3_S - FE3_LAG of it, whose reference level reaches back 47 chips -- chips below FE3_WARM_J0 * 8 stay zero
        if (have) {
            if (test) fe3_stage_step<false>(a, L, a.out_abs0 + (long long)step * FE3_T, slot0, par, tid);
            else fe3_stage_step<false, FE3_WARM_J0>(a, L, a.out_abs0 + (long long)step * FE3_T, slot0, par, tid);
        } else
            fe3_stage_step<true>(a, L, a.out_abs0 + (long long)step * FE3_T, slot0, par, tid);
        FE3_STAMP(5);
        fes_barrier();                                                // B1: |.|^2 of this step staged
        FE3_STAMP(0);
        fe3_step(a, L, step, test, slot0, par, edge, tid, mxrun, badrun, ncand, PR);
        slot0 = fe3_wrap_up(slot0 + FE3_S);
        par ^= 1;
        FE3_STAMP(6);
        fes_barrier();                                                // B5: every ring read of this step done
#if defined(FE3_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
        if (step == sb - 1) PR.acc[9] = (long long)wall_clock64();    // rings rebuilt,
        if (step == sb) PR.acc[11] = (long long)wall_clock64();       // first tested step done,
#endif
    }
    // the largest sample of the segment (with the chips the ring rebuild went through): +inf if one was not finite
    {
        float wmx = mxrun;
        for (int o = 32; o >= 1; o >>= 1) wmx = fmaxf(wmx, __shfl_xor(wmx, o, AM_WAVE));
        const bool bad = __ballot(badrun) != 0ull;
        uint32_t wcnt = ncand;
        for (int o = 32; o >= 1; o >>= 1) wcnt += (uint32_t)__shfl_xor((int)wcnt, o, AM_WAVE);
        if ((tid0 & (AM_WAVE - 1)) == 0) {
            L.MP[tid0 / AM_WAVE] = bad ? __builtin_inff() : wmx;
            L.TAB[tid0 / AM_WAVE] = wcnt;
        }
        fes_barrier();
        if (tid0 == 0) {
            float v = L.MP[0];
            uint32_t n = L.TAB[0];
            for (int w = 1; w < FE3_NW; ++w) { v = fmaxf(v, L.MP[w]); n += L.TAB[w]; }
            a.wg_max[blockIdx.x] = v;
            a.wg_cnt[blockIdx.x] = n;
        }
    }
#if defined(FE3_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    PR.acc[10] = (long long)wall_clock64();                           // end
    if (a.prof && (tid0 & (AM_WAVE - 1)) == 0)
        for (int k = 0; k < 12; ++k) a.prof[((size_t)blockIdx.x * FE3_NW + tid0 / AM_WAVE) * 12 + k] = PR.acc[k];
#endif
}

// ---- host side ----------------------------------------------------------------------------------------
unsigned am_fe3_tile(void) { return FE3_T; }
unsigned am_fe3_lag(void) { return FE3_LAG * FE3_SPC; }
unsigned am_fe3_waves(void) { return FE3_NW; }
unsigned am_fe3_steps(long long out_n) { return (unsigned)((out_n + FE3_LAG * FE3_SPC + FE3_T - 1) / FE3_T); }


static int fe3_wgs_for_device() { return FE3_WG_PER_CU * am_device_cus(); }   // resident workgroups

hipError_t am_launch_fe3(const float *iq, long long src_abs0, long long src_abs1, long long out_abs0, long long out_n,
                         float *bb_sparse, float *avg_sparse, uint32_t j0, uint32_t j1, int use_pmf, float s1, float sL,
                         float thr_lin, uint32_t *bits, uint32_t *wg_cnt, float *wg_max, unsigned *nsteps, unsigned *steps_per_wg,
                         hipStream_t s, int wgs_per_cu, unsigned *n_long)
{
    am_fe3_args a;
    a.iq = iq; a.src_abs0 = src_abs0; a.src_abs1 = src_abs1; a.out_abs0 = out_abs0; a.out_n = out_n;
    a.bb_sparse = bb_sparse; a.avg_sparse = avg_sparse; a.j0 = j0; a.j1 = j1; a.bits = bits; a.wg_cnt = wg_cnt; a.wg_max = wg_max;
    a.use_pmf = (use_pmf && FE3_SPC > 1) ? 1 : 0; a.s1 = s1; a.sL = sL; a.thr_lin = thr_lin;
    a.nsteps = am_fe3_steps(out_n);
    *nsteps = a.nsteps;
    *steps_per_wg = 1;
    if (a.nsteps == 0) return hipSuccess;
    // steps served by DMA: samples [out_abs0 + k T, + T) inside [src_abs0, src_abs1), source 16-byte aligned (the
    // parity of the offset is the same for every step: T is even)
    const bool aligned = ((reinterpret_cast<uintptr_t>(iq) + (uintptr_t)(out_abs0 - src_abs0) * 8u) & 15u) == 0;
    auto clampi = [](long long v) { return (int)(v < -4 ? -4 : (v > 0x7FFFFFF0ll ? 0x7FFFFFF0ll : v)); };
    a.raw_lo = clampi(fes_ceil_div(src_abs0 - out_abs0, FE3_T));
    a.raw_hi = aligned ? clampi(fes_floor_div(src_abs1 - out_abs0, FE3_T)) : a.raw_lo;
    // steps whose tested positions [k T - 288, k T + T - 288) all lie in [j0, min(j1, out_n))
    const long long lag = (long long)FE3_LAG * FE3_SPC;
    const long long jhi = (long long)j1 < out_n ? (long long)j1 : out_n;
    a.test_lo = clampi(fes_ceil_div((long long)j0 + lag, FE3_T));
    a.test_hi = clampi(fes_floor_div(jhi + lag, FE3_T));
    // persistent workgroups: as many as are resident at once, each with a contiguous run of steps; short inputs
    // get at least 4 steps per workgroup (the ring rebuild costs one)
    unsigned resident = (unsigned)fe3_wgs_for_device();
    if (wgs_per_cu > 0 && wgs_per_cu < FE3_WG_PER_CU) resident = (unsigned)(wgs_per_cu * am_device_cus());   // (am_pipe: room for other batches' tails)
#if defined(AM_TEST_KNOBS)
    if (const char *e = getenv("AIRMODES_FE3_WGS_PER_CU"))            // tuning: leave room on every CU for another batch's tail
        if (atoi(e) > 0) resident = (unsigned)(atoi(e) * am_device_cus());
#endif
    unsigned spw = (a.nsteps + resident - 1) / resident;
    if (spw < 4) spw = 4;
#ifdef FE3_FORCE_SPW
    spw = FE3_FORCE_SPW;                                              // tuning builds
#endif
    unsigned grid = (a.nsteps + spw - 1) / spw;
    a.n_long = grid;
    if (n_long) {
        // the caller can place segments of two lengths (am_k_refine_seg): as many workgroups as are resident (at least ~4 steps each),
        // the steps dealt out as evenly as they go
        unsigned G = a.nsteps / 4u;
        G = G < 1u ? 1u : (G > resident ? resident : G);
        const unsigned lo = a.nsteps / G, r = a.nsteps - lo * G;
        grid = G;
        if (r == 0) { spw = lo; a.n_long = G; }
        else { spw = lo + 1u; a.n_long = r; }
        *n_long = a.n_long;
    }
    a.steps_per_wg = spw;
    *steps_per_wg = spw;
    static std::atomic<bool> attr_done[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_done[dev].load(std::memory_order_acquire)) {
        hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void *>(&am_k_fe3),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)FE3_LDS_BYTES);
        if (rc != hipSuccess) return rc;
        if (dev >= 0 && dev < 64) attr_done[dev].store(true, std::memory_order_release);
    }
    a.prof = nullptr;
#if defined(FE3_PROFILE)
    // blocking; prints mean cycles per step and phase (wave 0 / wave 1) -- never in the default build
    if (hipMalloc(reinterpret_cast<void **>(&a.prof), (size_t)grid * FE3_NW * 12 * sizeof(long long)) != hipSuccess) a.prof = nullptr;
#endif
    hipLaunchKernelGGL(am_k_fe3, dim3(grid), dim3(FE3_NT), FE3_LDS_BYTES, s, a);
    hipError_t lrc = hipGetLastError();
#if defined(FE3_PROFILE)
    if (a.prof) {
        int occ = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void *>(&am_k_fe3), FE3_NT, FE3_LDS_BYTES);
        fprintf(stderr, "fe3: grid %u, %u steps per workgroup, %d bytes of LDS, runtime says %d workgroups per CU\n", grid, spw,
                (int)FE3_LDS_BYTES, occ);
        std::vector<long long> h((size_t)grid * FE3_NW * 12);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), a.prof, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        (void)hipFree(a.prof);
        static const char *names[12] = {"B1wait", "A pmf+totals+scans+ring", "B3wait", "B avg+test", "B5wait", "stage (wait loads, lds)",
                                        "next loads+sparse", "-", "-", "-", "-", "-"};
        for (int w = 0; w < FE3_NW; ++w) {
            double acc[12] = {};
            for (unsigned b = 0; b < grid; ++b)
                for (int k = 0; k < 12; ++k) acc[k] += (double)h[((size_t)b * FE3_NW + w) * 12 + k];
            const double steps = (double)grid * (double)(spw + 1);
            double tot = 0;
            for (int k = 0; k < 7; ++k) tot += acc[k];
            fprintf(stderr, "fe3 clocks/step wave %d (total %.0f):", w, tot / steps);
            static const int order[7] = {5, 0, 1, 2, 3, 6, 4};       // order of execution
            for (int k = 0; k < 7; ++k) fprintf(stderr, " %s:%.0f", names[order[k]], acc[order[k]] / steps);
            fprintf(stderr, "\n");
        }
        // the launch's timeline from wave 0 of every workgroup (wall clock, 10 ns ticks): when the workgroups start, when their rings
        // are rebuilt, when the first tested step is done, when they end -- in microseconds after the first start
        {
            std::vector<double> t0(grid), t1(grid), t2(grid), t3(grid);
            long long first = h[8];
            for (unsigned b = 0; b < grid; ++b) first = std::min(first, h[(size_t)b * FE3_NW * 12 + 8]);
            for (unsigned b = 0; b < grid; ++b) {
                const long long *q = &h[(size_t)b * FE3_NW * 12];
                t0[b] = (double)(q[8] - first) * 0.01; t1[b] = (double)(q[9] - first) * 0.01;
                t2[b] = (double)(q[11] - first) * 0.01; t3[b] = (double)(q[10] - first) * 0.01;
            }
            auto pct = [&](std::vector<double> v, const char *name) {
                std::sort(v.begin(), v.end());
                const size_t n = v.size();
                fprintf(stderr, "fe3 timeline %-22s us after the first start: min %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f\n", name, v[0],
                        v[n / 10], v[n / 2], v[(size_t)(n * 0.9)], v[(size_t)(n * 0.99)], v[n - 1]);
            };
            pct(t0, "start"); pct(t1, "rings rebuilt"); pct(t2, "first tested step done"); pct(t3, "end");
            std::vector<double> dur(grid), per(grid);
            for (unsigned b = 0; b < grid; ++b) { dur[b] = t3[b] - t0[b]; per[b] = (t3[b] - t2[b]) / (double)(spw > 1 ? spw - 1 : 1); }
            pct(dur, "duration");
            pct(per, "per step after the 1st");
            // by CU: do a CU's workgroups end together (the chip drains CU by CU) or one after the other (every CU is busy to the end)?
            {
                std::vector<std::pair<unsigned, unsigned>> key(grid);      // (cu key, workgroup)
                for (unsigned b = 0; b < grid; ++b) {
                    const unsigned long long w = (unsigned long long)h[(size_t)b * FE3_NW * 12 + 7];
                    const unsigned hw = (unsigned)w, xcc = (unsigned)(w >> 32) & 15u;
                    key[b] = {(xcc << 12) | (((hw >> 13) & 7u) << 8) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u), b};
                }
                std::sort(key.begin(), key.end());
                std::vector<double> cu_first, cu_last, cu_n, rank_end[8];
                for (size_t i = 0; i < key.size();) {
                    size_t j = i;
                    std::vector<std::pair<double, double>> se;          // (start, end) of the CU's workgroups
                    while (j < key.size() && key[j].first == key[i].first) { se.push_back({t0[key[j].second], t3[key[j].second]}); j++; }
                    std::sort(se.begin(), se.end());
                    double lo = se[0].second, hi = se[0].second;
                    for (size_t k = 0; k < se.size(); ++k) {
                        lo = std::min(lo, se[k].second); hi = std::max(hi, se[k].second);
                        if (k < 8) rank_end[k].push_back(se[k].second);
                    }
                    cu_first.push_back(lo); cu_last.push_back(hi); cu_n.push_back((double)se.size());
                    i = j;
                }
                fprintf(stderr, "fe3 timeline: %zu CUs seen\n", cu_n.size());
                pct(cu_n, "workgroups per CU");
                pct(cu_first, "a CU's first end");
                pct(cu_last, "a CU's last end");
                for (int k = 0; k < 8; ++k)
                    if (!rank_end[k].empty()) {
                        char nm[64];
                        snprintf(nm, sizeof nm, "end of a CU's %d. started", k + 1);
                        pct(rank_end[k], nm);
                    }
            }
            // by XCD (workgroups are dealt round-robin over the eight XCDs)
            for (int x = 0; x < 8; ++x) {
                double e = 0, m = 0; unsigned n = 0;
                for (unsigned b = (unsigned)x; b < grid; b += 8) { e += t3[b]; m = std::max(m, t3[b]); n++; }
                fprintf(stderr, "fe3 timeline xcd %d: mean end %.1f last end %.1f (%u workgroups)\n", x, n ? e / n : 0.0, m, n);
            }
        }
    }
#endif
    return lrc;
}
