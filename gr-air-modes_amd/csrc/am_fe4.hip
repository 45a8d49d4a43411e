// am_fe4.hip -- the streaming fused front end + first-stage preamble detection for gfx950, every rate the library
// specialises: 2, 4, 8, 10, 16, 20, 32, 40 and 64 Msps (the rate BASELINE.json's metric is quoted on).
//
// One template, am_k_fe4<SPC, G, NW>: SPC samples per chip, a lane owns a UNIT of G consecutive chips (R = G * SPC samples
// in registers, even: 32 at 64 Msps with G = 1, 30 at 20 and 10 Msps, 24 at 2 Msps), NW waves per workgroup.
//
//   * PERSISTENT workgroups: a workgroup owns a contiguous segment of the stream and walks it in steps of US units.
//     What a step needs from the past -- the pulse-matched power bb of the last LAGU + 48 chips' worth of units and
//     three scalars per unit -- stays in LDS rings, so nothing is loaded twice; a segment costs one extra step (no test,
//     no output) that rebuilds the rings from its predecessor's tail.
//   * raw IQ arrives by plain coalesced 16-byte loads with the streaming (nt) policy, issued back to back and waited
//     for; |.|^2 goes straight into the ring rows of the step's units.  No prefetch: while one workgroup waits for its
//     loads the others of the CU compute (LDS-DMA staging and register prefetch cost the occupancy this
//     arithmetic needs).
//   * phase A, per lane: pulse-matched filter of the unit's chips from in-chip prefix / suffix sums (the chip before a
//     unit's first chip belongs to lane - 1: DPP wave_shr:1), chip totals, and the strictly sequential in-block scans of
//     the 48 chip totals (canonical order) as hops from lane to lane, G additions per hop.
//   * phase B (reference level + first-stage test) runs LAGU units BEHIND phase A, so the pulses 2, 7 and 9 chips ahead
//     are already in the ring: no right halo, no redundant arithmetic.
//   * outputs are sparse: one candidate bit per position (a dense bitmap, 1/64 of the input bytes), a count per (step,
//     wave), and, only around candidates, rows of bb (17 chips from a candidate's chip on: what am_k_cand reads) and of
//     the reference level (the unit of a candidate and the next); plus one number per workgroup: the largest bb of its
//     segment (+inf if one was not finite), the bound the refinement's exact energy-difference test needs.
//
// LANES THAT OWN A UNIT.  A 48-chip block is LPB = 48 / G lanes.  Where a step of NW * 64 units is a whole number of blocks,
// all 64 lanes of every wave own a unit (LU = 64): 64 Msps with three waves (192 chips = 4 blocks), 20, 10 and 2 Msps
// with two; otherwise 48 lanes do and 16 only help with the loads.  With LU = 64 and LPB not a divisor of 64 (64 Msps:
// blocks start at lanes 0 and 48 of wave 0, 32 of wave 1, 16 of wave 2) a block STRADDLES two waves, and the sequential
// scans cross the wave boundary.  No extra workgroup barrier is spent on that:
//   - before barrier B3 every wave scans what it can on its own (forward: blocks that start in it; backward: blocks
//     that end in it), and leaves in LDS the forward value that leaves its lane 63 and, for the lanes whose block ends
//     in the next wave, their chip totals;
//   - after B3 wave w finishes (a) the forward scan of its own leading lanes from wave w - 1's value and (b) the backward
//     scan of wave w - 1's trailing lanes from its own lane 0 -- and every value formed there is read, 9 chips later in
//     phase B, by wave w itself (checked at compile time), so program order inside the wave is all the ordering needed.
//
// Same canonical order, same results as the tile kernel am_k_fe2 / the oracle; burst extraction
// recomputes its samples from IQ (am_kernels.hip).
//
// Reference: python/rx_path.py:35-54 (spc = rate / 2e6, |.|^2, moving averages), lib/preamble_impl.cc:172-179 (test).
#include "am_fe_stream.h"

#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <vector>

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#include "am_fe_cmpx.h"

#ifndef FE4_WG_PER_CU
#define FE4_WG_PER_CU 6                   /* fallback when the occupancy query fails */
#endif
// tuning builds only (tools/build_variants.sh): FE4_ABLATE bit mask removes parts of the kernel -- results INVALID
//   1: no sparse outputs at all   16: no reference-level rows   32: no bb rows
#ifndef FE4_ABLATE
#define FE4_ABLATE 0
#endif
#ifndef FE4_NW64
#define FE4_NW64 3                        /* waves per workgroup at 64 Msps */
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
    uint32_t *wg_cnt;                     // [grid] candidates a workgroup found (am_k_gather_wg lays the flat list out from these)
    float *wg_max;                        // [grid] largest bb a workgroup formed (+inf if one was not finite)
    unsigned nsteps, steps_per_wg;
    int raw_lo, raw_hi, test_lo, test_hi; // steps loaded without guards / tested without a range mask
    int use_pmf;
    float s1, sL, thr_lin;
    long long *prof;                      // profiling builds: [grid * NW][8] cycles per phase, else null
};

template <int SPC, int G, int NWV>
struct fe4_cfg {
    static constexpr int NW = NWV;                                     // waves per workgroup
    static constexpr int NT = AM_WAVE * NW;
    static constexpr int R = SPC * G;                                  // samples per unit
    // ring row stride in floats: even (rows are read 8 bytes at a time), at least R + 2, and not a multiple of 32 (consecutive
    // lanes read consecutive rows: a stride of 32 floats would put every lane on the same banks)
    static constexpr int RS = ((R + 2) % 32 == 0) ? R + 4 : R + 2;
    static constexpr int LPB = AM_CHIPS_AVG / G;                       // lanes (units) per 48-chip block
    // lanes of a wave that own a unit: all 64 where a step of NW x 64 units is a whole number of blocks, 48 otherwise
    static constexpr int LU = ((AM_WAVE * NW) % LPB == 0) ? AM_WAVE : AM_CHIPS_AVG;
    // ... and then blocks cross wave boundaries unless a wave is a whole number of blocks itself
    static constexpr bool STRADDLE = (LU == AM_WAVE) && (AM_WAVE % LPB != 0);
    static constexpr int US = LU * NW;                                 // units per step
    // waves per SIMD the registers are budgeted for: 3 (<= 168 VGPRs; the 2 Msps kernel needs fewer and gets 8 workgroups).
    // (4 for the 20 Msps kernel -- 128 VGPRs, 7 workgroups per CU, one step fewer per workgroup -- measured 0.058 against
    // 0.0556 ms: the tighter register budget costs more than the step.)
    static constexpr int MINW = 3;
    static constexpr unsigned long long LUMASK = (LU == 64) ? ~0ull : ((1ull << (LU & 63)) - 1ull);
    static constexpr int LAGU = 1 + 8 / G;                             // units phase B runs behind phase A
    static constexpr int NBU = (G - 1 + 16) / G;                       // units of bb kept after a candidate's unit
    static constexpr int CRU = US + LAGU + LPB + 1;                    // ring capacity in units
    static constexpr int T = US * R;                                   // samples per step
    static constexpr int PIECES = T / 2;                               // 16-byte pieces (2 samples) per step
    static constexpr int NLD = (PIECES + NT - 1) / NT;                 // loads per thread and step
    static constexpr int LPR = R / 2;                                  // lanes that move one row (8 bytes each)
    static constexpr int RPI = AM_WAVE / LPR;                          // rows per wave instruction
    static constexpr int UPR = NT / LPR;                               // staging: rows (units) per round of the workgroup's threads
    static constexpr int ROUNDS = (US + UPR - 1) / UPR;                // staging: rounds (= loads per thread) per step
    static_assert(AM_CHIPS_AVG % G == 0 && R % 2 == 0 && R <= 32 && RS % 2 == 0, "unit shape");
    // straddling blocks (see the head of the file).  Wave w's first s0(w) lanes continue a block of wave w - 1; its last
    // t0(w) lanes belong to a block that ends in wave w + 1.  (A step is a whole number of blocks: s0(0) = t0(NW - 1) = 0.)
    static constexpr int s0_of(int w) { return (LPB - (AM_WAVE * w) % LPB) % LPB; }
    static constexpr int t0_of(int w) { return (AM_WAVE * (w + 1)) % LPB; }
    static constexpr int max_s0() { int m = 0; for (int w = 0; w < NW; ++w) m = s0_of(w) > m ? s0_of(w) : m; return m; }
    static constexpr int max_t0() { int m = 0; for (int w = 0; w < NW; ++w) m = t0_of(w) > m ? t0_of(w) : m; return m; }
    static constexpr int MAXS0 = STRADDLE ? max_s0() : 0;
    static constexpr int MAXT0 = STRADDLE ? max_t0() : 0;
    // what wave w completes after B3 is read by wave w alone in the same step (and by nobody in a ring-rebuilding step):
    //   prefix of unit u in [64 w, 64 w + s0): read by the lane whose phase-A unit is u + LAGU;
    //   suffix values of unit u in [64 w - t0(w-1), 64 w): read by the lanes whose phase-A units are u + LPB + LAGU - 1, u + LPB + LAGU
    static_assert(!STRADDLE || (LPB - 1 + LAGU < AM_WAVE && LPB + LAGU <= AM_WAVE), "continued scans must be consumed by the wave that forms them");
    static constexpr int LDS_FLOATS = CRU * RS + NW * 4 * RS + 3 * CRU + 2 * SPC + (NW - 1) * SPC + 2 + NW * 64 + 4 + 2 * NW +
                                      (STRADDLE ? NW + (NW - 1) * MAXT0 * (G + 1) : 0);
};

template <int SPC, int G, int NW>
struct fe4_smem {
    float *X;                 // [CRU * RS] ring rows: |.|^2 of a step's units while it is staged, then bb
    float *UPT, *UST, *USL;   // [CRU] per unit: in-block prefix of the chip totals entering it from the left; suffix entering it
                              // from the right (= ST of its last chip); RTOT + ST of its first chip
    float *SBL;               // [2][SPC] in-chip suffix sums of |.|^2 of a step's last chip (by step parity)
    float *MLW;               // [NW-1][SPC] |.|^2 of the last chip of wave w's last unit (the chip before wave w+1's first)
    float *AVS;               // [NW][4 * RS] units of reference level on their way out (four per wave; with two waves and wave 0
                              // parking in the ring, all eight rows are wave 1's)
    uint32_t *CARRY;          // [2] units at the start of the next step whose bb must be written (by step parity)
    uint32_t *TAB;            // [NW][64] lane of the r-th unit whose bb / reference level is written
    float *WMX;               // [NW] the waves' largest samples at the end
    uint32_t *WOV;            // [2][NW] units after a wave's last whose bb must be written (by step parity; LU = 64 only)
    float *CF;                // [NW] straddling blocks: the forward scan value leaving a wave's lane 63
    float *FT;                // [NW-1][MAXT0][G + 1] ... chip totals (G) and the first chip's right->left total of a wave's trailing lanes
};

template <int SPC, int G, int NW>
__device__ __forceinline__ fe4_smem<SPC, G, NW> fe4_smem_at(float *base)
{
    using C = fe4_cfg<SPC, G, NW>;
    fe4_smem<SPC, G, NW> L;
    L.X = base;                                                       // (the arrays read 8 bytes at a time first: even sizes)
    L.AVS = L.X + C::CRU * C::RS;
    L.UPT = L.AVS + NW * 4 * C::RS;
    L.UST = L.UPT + C::CRU;
    L.USL = L.UST + C::CRU;
    L.SBL = L.USL + C::CRU;
    L.MLW = L.SBL + 2 * SPC;
    L.CARRY = reinterpret_cast<uint32_t *>(L.MLW + (NW - 1) * SPC);
    L.TAB = L.CARRY + 2;
    L.WMX = reinterpret_cast<float *>(L.TAB + NW * 64);
    L.WOV = reinterpret_cast<uint32_t *>(L.WMX + 4);
    L.CF = reinterpret_cast<float *>(L.WOV + 2 * NW);
    L.FT = L.CF + NW;
    return L;
}

// Profiling builds only (-DFE4_PROFILE, tools/build_variants.sh): cycles per phase, summed over a workgroup's steps
// by lane 0 of each wave.  The default build contains none of it.
#if defined(FE4_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
struct fe4_prof { long long last; long long acc[8]; };
#define FE4_STAMP(k) do { const long long now__ = (long long)__builtin_readcyclecounter(); PR.acc[k] += now__ - PR.last; PR.last = now__; } while (0)
#else
struct fe4_prof { };
#define FE4_STAMP(k) do { } while (0)
#endif

template <class C>
__device__ __forceinline__ int fe4_wrap_up(int s) { return s >= C::CRU ? s - C::CRU : s; }
template <class C>
__device__ __forceinline__ int fe4_wrap_dn(int s) { return s < 0 ? s + C::CRU : s; }

// |iq|^2 of one step into the ring rows of its units.  GUARD: stream edges / unaligned input, one sample at a time, zeros
// outside the stream: piece p = tid + NT j holds samples 2p, 2p+1 of the step = unit (2p) / R, offset (2p) % R (R is even: a
// piece never straddles two units).
template <int SPC, int G, int NW>
__device__ __forceinline__ void fe4_put_piece(const fe4_smem<SPC, G, NW> &L, int slot0, int u, int o, float m0, float m1)
{
    using C = fe4_cfg<SPC, G, NW>;
    float2 mm; mm.x = m0; mm.y = m1;
    *reinterpret_cast<float2 *>(L.X + fe4_wrap_up<C>(slot0 + u) * C::RS + o) = mm;
    // the last chip of a wave's last unit a second time: the next wave needs it after the row holds bb
    if ((u + 1) % C::LU == 0 && u + 1 < C::US && o >= C::R - SPC - (SPC & 1)) {
        float *d = L.MLW + ((u + 1) / C::LU - 1) * SPC;
        const int k = o - (C::R - SPC);
        if (k >= 0) d[k] = m0;
        if (k + 1 >= 0 && k + 1 < SPC) d[k + 1] = m1;
    }
}
template <int SPC, int G, int NW>
__device__ __forceinline__ void fe4_stage_guarded(const am_fe4_args &a, const fe4_smem<SPC, G, NW> &L, long long A0, int slot0, int tid)
{
    using C = fe4_cfg<SPC, G, NW>;
    const float2 *iq2 = reinterpret_cast<const float2 *>(a.iq);
#pragma unroll 1
    for (int j = 0; j < C::NLD; ++j) {
        const int p = tid + C::NT * j;
        if (p >= C::PIECES) break;
        const long long n = A0 + 2 * (long long)p;
        float2 u0, u1;
        u0.x = 0.0f; u0.y = 0.0f; u1 = u0;
        if (n >= a.src_abs0 && n < a.src_abs1) u0 = iq2[n - a.src_abs0];
        if (n + 1 >= a.src_abs0 && n + 1 < a.src_abs1) u1 = iq2[n + 1 - a.src_abs0];
        const float r0 = u0.x * u0.x, i0 = u0.y * u0.y, r1 = u1.x * u1.x, i1 = u1.y * u1.y;
        fe4_put_piece<SPC, G, NW>(L, slot0, (2 * p) / C::R, (2 * p) % C::R, r0 + i0, r1 + i1);
    }
}
// The same for a step whose samples are all present and aligned, in ROUNDS of whole rows: R / 2 lanes move one row, a
// round is as many rows as the workgroup's threads hold (UPR; at 30 or 24 samples per unit eight threads of 128 sit
// out), so a thread's unit advances by UPR per round and its place in the row never changes: one compare and one select
// per piece for the ring's wrap-around, the rest is the instructions' immediate offsets.  (The division and the remainder by
// R per piece, the 32-bit multiply by the row stride -- quarter rate -- and the select chains around them were a quarter of
// the kernel's VALU time, which is what bounds it: DESIGN.md 5.1a.)  J0: rounds below it are not loaded (ring rebuild).
template <int SPC, int G, int NW, int J0>
__device__ __forceinline__ void fe4_stage_rows(const am_fe4_args &a, const fe4_smem<SPC, G, NW> &L, long long A0, int slot0, int tid)
{
    using C = fe4_cfg<SPC, G, NW>;
    constexpr int LPR = C::LPR, UPR = C::UPR, USED = UPR * LPR, ROUNDS = C::ROUNDS;
    if (USED < C::NT && tid >= USED) return;
    const int c0 = fes_div_small<LPR>(tid);                           // unit inside the round
    const int k = tid - fes_mul24(c0, LPR);                           // piece inside the row
    const int sc = slot0 + c0;
    float *const row = L.X + fes_mul24(sc, C::RS) + 2 * k;            // round 0's destination
    float *const roww = row - C::CRU * C::RS;
    const unsigned long long gb = reinterpret_cast<unsigned long long>(a.iq) + (unsigned long long)(A0 - a.src_abs0) * 8ull;
    const unsigned off = (unsigned)tid * 16u;
    float4 v[ROUNDS];
#pragma unroll
    for (int j = J0; j < ROUNDS; ++j) {
        if (UPR * (j + 1) > C::US && c0 + UPR * j >= C::US) continue; // (the last round may be short)
        unsigned long long g = gb + (unsigned long long)(j & ~1) * (USED * 16u);
#if defined(__HIP_DEVICE_COMPILE__)
        asm("" : "+s"(g));                                            // (scalar base per pair of rounds + the lane's offset)
#endif
        v[j] = fes_gload16_at(g, off + (unsigned)(j & 1) * (USED * 16u));
    }
#pragma unroll
    for (int j = J0; j < ROUNDS; ++j) {
        if (UPR * (j + 1) > C::US && c0 + UPR * j >= C::US) continue;
        const fes_f2 q0 = fes_pk_mul(fes_mk2(v[j].x, v[j].y), fes_mk2(v[j].x, v[j].y));
        const fes_f2 q1 = fes_pk_mul(fes_mk2(v[j].z, v[j].w), fes_mk2(v[j].z, v[j].w));
        float2 mm;
        mm.x = q0.x + q0.y;                                           // a1: fl(fl(I*I) + fl(Q*Q))
        mm.y = q1.x + q1.y;
        float *dst = (sc >= C::CRU - UPR * j) ? roww : row;
        *reinterpret_cast<float2 *>(dst + j * (UPR * C::RS)) = mm;
        // the last chip of a wave's last unit a second time: the next wave needs it after the row holds bb
#pragma unroll
        for (int m = 1; m < NW; ++m) {
            constexpr int LUc = C::LU;
            const int cstar = LUc * m - 1 - UPR * j;                  // (compile time after unrolling)
            if (cstar >= 0 && cstar < UPR && c0 == cstar && 2 * k >= C::R - SPC - (SPC & 1)) {
                float *d = L.MLW + (m - 1) * SPC;
                const int kk = 2 * k - (C::R - SPC);
                if (kk >= 0) d[kk] = mm.x;
                if (kk + 1 >= 0 && kk + 1 < SPC) d[kk + 1] = mm.y;
            }
        }
    }
}

// One step (its |.|^2 is staged).
//   step     global step index (may be -1: history before the first wanted block)
//   test     false for a workgroup's first step (it only rebuilds the rings from the previous segment's tail)
//   slot0    ring slot of this step's unit 0
//   edge     (uniform) the step touches the end of the stream or positions that are not wanted
template <int SPC, int G, int NW>
__device__ __forceinline__ void fe4_step(const am_fe4_args &a, const fe4_smem<SPC, G, NW> &L, const int step, const bool test,
                                         const int slot0, const int par, const bool edge, const int tid, float &mxrun,
                                         bool &badrun, uint32_t &ncand, fe4_prof &PR)
{
    using C = fe4_cfg<SPC, G, NW>;
    constexpr int R = C::R, RS = C::RS, LPB = C::LPB, LAGU = C::LAGU;
    const int lane = tid & (AM_WAVE - 1);
#if defined(__HIP_DEVICE_COMPILE__)
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);          // (wave-uniform, and known to be)
#else
    const int wv = tid >> 6;
#endif
    constexpr int LU = C::LU;
    const bool unit_thread = lane < LU;
    const int t = fes_mul24(wv, LU) + (unit_thread ? lane : LU - 1);    // unit of the step (spare lanes shadow the last one, never write)
    const long long A0 = a.out_abs0 + (long long)step * C::T;
    const int slotA = fe4_wrap_up<C>(slot0 + t);
    const bool do_pmf = a.use_pmf != 0 && SPC > 1;
    // position of the unit inside its 48-chip block (a step is a whole number of blocks; with 48 lanes per wave so is a wave)
    const int lb0 = (LU == AM_WAVE ? ((wv << 6) + lane) : lane);
    const int lb = lb0 - fes_mul24(fes_div_small<LPB>(lb0), LPB);
    // straddling blocks: this wave's leading lanes that continue a block of the wave before / trailing lanes whose block
    // ends in the next wave (wave-uniform)
    const int s0 = C::STRADDLE ? (LPB - (AM_WAVE * wv) % LPB) % LPB : 0;
    const int t0 = C::STRADDLE ? (AM_WAVE * (wv + 1)) % LPB : 0;

    // ---- phase A ---------------------------------------------------------------------------------------------------
    float bb[R];
    {
        float m[R];
        const float2 *mp = reinterpret_cast<const float2 *>(L.X + fes_mul24(slotA, RS));
#pragma unroll
        for (int k = 0; k < R / 2; ++k) { const float2 u = mp[k]; m[2 * k] = u.x; m[2 * k + 1] = u.y; }
        if (do_pmf) {
            // in-chip suffix sums (right -> left) and prefix sums (left -> right), restarted at every chip
            // (two independent chains per iteration)
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
            if (tid == (NW - 1) * AM_WAVE + LU - 1) {
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
            float tt[R];
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int i = 0; i < SPC; ++i) {
                    const int j = g * SPC + i;
                    if (i == SPC - 1) tt[j] = pp[j];                  // the window is the chip
                    else if (g == 0) tt[j] = fes_from_prev_lane(sx[(G - 1) * SPC + i + 1], pv[i + 1], lane) + pp[j];
                    else tt[j] = sx[(g - 1) * SPC + i + 1] + pp[j];
                }
#pragma unroll
            for (int j = 0; j < R; j += 2) {                          // (two products per instruction)
                const fes_f2 v = fes_pk_mul(fes_mk2(tt[j], tt[j + 1]), fes_mk2(a.s1, a.s1));
                bb[j] = v.x; bb[j + 1] = v.y;
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
    // chip totals left -> right (f) and right -> left (first chip: b0); spare lanes contribute zeros to the scans
    float f[G], b0 = 0.0f, xin = 0.0f, yin = 0.0f;
    {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float ff = 0.0f, bk = 0.0f;
#pragma unroll
            for (int i = 0; i < SPC; ++i) { ff = ff + bb[g * SPC + i]; bk = bk + bb[g * SPC + SPC - 1 - i]; mxrun = fmaxf(mxrun, bb[g * SPC + i]); }
            if (!unit_thread) ff = 0.0f;
            f[g] = ff;
            if (g == 0) b0 = bk;
            badrun = badrun || !(ff < __builtin_inff());              // a sample that is not finite makes its chip's total so (all terms >= 0 or NaN)
        }
        // in-block scans of the chip totals, strictly sequential (canonical order), hopping from lane to lane: the value
        // entering a lane from the left is ((x(lane-1) + f0) + f1) + ... of lane-1, 0 where a block starts; a lane's value
        // is final after as many rounds as its position in the block and is recomputed identically afterwards.
        // (Straddling blocks: a wave's leading s0 lanes get a forward value, its trailing t0 lanes a backward value that
        // is not theirs -- those are formed after B3, below, and not stored here.)
#pragma unroll
        for (int r = 0; r < LPB - 1; ++r) {
            float xe = xin, ye = yin;
#pragma unroll
            for (int g = 0; g < G; ++g) { xe = xe + f[g]; ye = ye + f[G - 1 - g]; }
            if constexpr (LPB == 16 && LU == AM_WAVE) {
                // (a block is a row of 16 lanes: the DPP row shift zero-fills where a block starts / ends)
                xin = fes_from_prev_lane_row16(xe, lane);
                yin = fes_from_next_lane_row16(ye, lane);
            } else {
                const float xp = fes_from_prev_lane(xe, 0.0f, lane), yn = fes_from_next_lane(ye, 0.0f, lane);
                xin = (lb == 0) ? 0.0f : xp;
                yin = (lb == LPB - 1) ? 0.0f : yn;
            }
        }
        if (unit_thread) {
            float st0 = yin;                                          // ST of the unit's first chip: the later chips of the unit, right -> left
#pragma unroll
            for (int g = G - 1; g >= 1; --g) st0 = st0 + f[g];
            if (!C::STRADDLE || lane >= s0) L.UPT[slotA] = xin;
            if (!C::STRADDLE || lane < AM_WAVE - t0) {
                L.UST[slotA] = yin;
                L.USL[slotA] = b0 + st0;
            }
            float2 *xp = reinterpret_cast<float2 *>(L.X + fes_mul24(slotA, RS));
#pragma unroll
            for (int k = 0; k < R / 2; ++k) { float2 u; u.x = bb[2 * k]; u.y = bb[2 * k + 1]; xp[k] = u; }
        }
        if constexpr (C::STRADDLE) {
            if (test) {                                               // (uniform; nothing of a ring-rebuilding step's continued scans is ever read)
                // what the next wave needs to finish the scans: the forward value leaving lane 63, and the totals of the
                // lanes whose block ends there
                if (lane == AM_WAVE - 1 && wv + 1 < NW) {
                    float xe = xin;
#pragma unroll
                    for (int g = 0; g < G; ++g) xe = xe + f[g];
                    L.CF[wv] = xe;
                }
                if (lane >= AM_WAVE - t0) {                           // (t0 = 0 in the last wave)
                    float *d = L.FT + ((size_t)wv * C::MAXT0 + (size_t)(lane - (AM_WAVE - C::MAXT0))) * (G + 1);
#pragma unroll
                    for (int g = 0; g < G; ++g) d[g] = f[g];
                    d[G] = b0;
                }
            }
        }
    }
    FE4_STAMP(1);
    fes_barrier();                                                    // B3: ring and scans of this step complete
    FE4_STAMP(2);
    if (!test) return;                                                // (uniform) ring rebuild only

    if constexpr (C::STRADDLE) {
        if (wv > 0) {                                                 // (uniform)
            const int t0p = (AM_WAVE * wv) % LPB;                     // trailing lanes of the wave before whose block ends here
            // (a) forward scan of the own leading lanes, entered with what left the wave before
            // (b) backward scan of the trailing lanes of the wave before, entered with what leaves the own lane 0 leftwards
            float fT[G], b0T = 0.0f;
#pragma unroll
            for (int g = 0; g < G; ++g) fT[g] = 0.0f;
            if (lane >= AM_WAVE - t0p) {
                const float *d = L.FT + ((size_t)(wv - 1) * C::MAXT0 + (size_t)(lane - (AM_WAVE - C::MAXT0))) * (G + 1);
#pragma unroll
                for (int g = 0; g < G; ++g) fT[g] = d[g];
                b0T = d[G];
            }
            const float cf = L.CF[wv - 1];
            // what leaves lane 0 to the left: its unit's totals on top of what entered it from the right (lane 0's block ends
            // in this wave: its backward value is final)
            float cb;
            {
                float ye = yin;
#pragma unroll
                for (int g = 0; g < G; ++g) ye = ye + f[G - 1 - g];
                cb = __shfl(ye, 0, AM_WAVE);
            }
            float xh = 0.0f, yh = 0.0f;
            constexpr int ROUNDS = C::MAXS0 > C::MAXT0 ? C::MAXS0 : C::MAXT0;
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                float xe = xh, ye = yh;
#pragma unroll
                for (int g = 0; g < G; ++g) { xe = xe + f[g]; ye = ye + fT[G - 1 - g]; }
                xh = fes_from_prev_lane(xe, cf, lane);
                yh = fes_from_next_lane(ye, cb, lane);
            }
            if (lane < s0) L.UPT[slotA] = xh;
            if (lane >= AM_WAVE - t0p) {
                const int slotP = fe4_wrap_dn<C>(slotA - AM_WAVE);    // the unit of the same lane in the wave before
                float st0 = yh;
#pragma unroll
                for (int g = G - 1; g >= 1; --g) st0 = st0 + fT[g];
                L.UST[slotP] = yh;
                L.USL[slotP] = b0T + st0;
            }
        }
        __builtin_amdgcn_wave_barrier();                              // (one wave: its LDS accesses execute in order)
    }

    // ---- phase B on unit v = (this thread's phase-A unit) - LAGU --------------------------------------------------
    const int slotB = fe4_wrap_dn<C>(slotA - LAGU);
    const int slotS = fe4_wrap_dn<C>(slotB - LPB);                    // the unit 48 chips back
    float x[R], avgv[R];
    {
        float sc[R];
        const float2 *xp = reinterpret_cast<const float2 *>(L.X + fes_mul24(slotB, RS));
        const float2 *sp = reinterpret_cast<const float2 *>(L.X + fes_mul24(slotS, RS));
#pragma unroll
        for (int k = 0; k < R / 2; ++k) { const float2 u = xp[k]; x[2 * k] = u.x; x[2 * k + 1] = u.y; }
#pragma unroll
        for (int k = 0; k < R / 2; ++k) { const float2 u = sp[k]; sc[2 * k] = u.x; sc[2 * k + 1] = u.y; }
        const float xinB = L.UPT[slotB], yinS = L.UST[slotS];
        const float sl_next = L.USL[fe4_wrap_up<C>(slotS + 1)];       // RTOT + ST of the chip after the back unit's last
        // position of the unit's first chip inside its block (phase B's unit is LAGU behind: (lb - LAGU) mod LPB)
        const int lbB = lb - (LAGU % LPB) + ((lb < LAGU % LPB) ? LPB : 0);
        // back unit: forward chip totals (for ST), right -> left in-chip suffix sums (their first value is RTOT)
        float fS[G], stS[G], rtS[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float ff = 0.0f, as = 0.0f;
            if constexpr (G > 1) {
#pragma unroll
                for (int i = 0; i < SPC; ++i) ff = ff + sc[g * SPC + i];       // (before the row is overwritten with its suffix sums)
            }
#pragma unroll
            for (int i = 0; i < SPC; ++i) { as = as + sc[g * SPC + SPC - 1 - i]; sc[g * SPC + SPC - 1 - i] = as; }
            fS[g] = ff; rtS[g] = as;
        }
        {
            float y = yinS;                                           // ST of the back unit's last chip
#pragma unroll
            for (int g = G - 1; g >= 0; --g) { stS[g] = y; if (g > 0) y = y + fS[g]; }
        }
        // own unit: in-chip prefix sums, PT per chip
        float pt = xinB;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float ap = 0.0f;
            // (the block's last chip: lbB == LPB - 1 and g == G - 1)
            const bool blk_last = (lbB == LPB - 1) && (g == G - 1);
            const float suf_last = (g + 1 < G) ? (rtS[(g + 1 < G) ? g + 1 : g] + stS[(g + 1 < G) ? g + 1 : g]) : sl_next;
            if constexpr (SPC % 2 == 0) {
                // two positions per instruction; the chip's last position: (suf_last + pre), or pre alone where the window ends
                // with the block (x + (-0) == x for every x, NaN included)
                float apv[SPC];
#pragma unroll
                for (int i = 0; i < SPC; ++i) { ap = ap + x[g * SPC + i]; apv[i] = ap; }
                const float q_last = blk_last ? -0.0f : suf_last;
#pragma unroll
                for (int i = 0; i < SPC; i += 2) {
                    const int j = g * SPC + i;
                    const fes_f2 pre = fes_pk_add(fes_mk2(pt, pt), fes_mk2(apv[i], apv[i + 1]));
                    fes_f2 q;
                    if (i + 2 < SPC) q = fes_pk_add(fes_mk2(sc[j + 1], sc[j + 2]), fes_mk2(stS[g], stS[g]));
                    else q = fes_mk2(sc[j + 1] + stS[g], q_last);
                    const fes_f2 av = fes_pk_mul(fes_pk_add(q, pre), fes_mk2(a.sL, a.sL));
                    avgv[j] = av.x; avgv[j + 1] = av.y;
                }
            } else {
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
            return d == 0 ? x[gg * SPC + i] : L.X[fes_mul24(fe4_wrap_up<C>(slotB + d), RS) + gg * SPC + i];
        };
        const float nxt = L.X[fes_mul24(fe4_wrap_up<C>(slotB + 1), RS)];
        // eight samples at a time (the partial results of more would not fit the scalar registers: they are lane masks)
#pragma unroll
        for (int h = 0; h < R; h += 8) {
            constexpr int CH = 8;
            float thr[CH], xs[CH + 1];
#pragma unroll
            for (int k = 0; k < CH; k += 2) {                         // (R is even: a pair is inside the unit or beyond it)
                const int j = (h + k < R) ? h + k : R - 2;
                const fes_f2 th = fes_pk_mul(fes_mk2(avgv[j], avgv[j + 1]), fes_mk2(a.thr_lin, a.thr_lin));   // :173
                thr[k] = th.x; thr[k + 1] = th.y;
                xs[k] = x[j]; xs[k + 1] = x[j + 1];
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
                if constexpr (G == 1 && SPC % 8 == 0) {
                    // (one chip per unit: the pulses are the same offsets of the rows 2, 7 and 9 units on -- 8-byte reads)
                    const float2 *q2 = reinterpret_cast<const float2 *>(L.X + fes_mul24(fe4_wrap_up<C>(slotB + 2), RS) + h);
                    const float2 *q7 = reinterpret_cast<const float2 *>(L.X + fes_mul24(fe4_wrap_up<C>(slotB + 7), RS) + h);
                    const float2 *q9 = reinterpret_cast<const float2 *>(L.X + fes_mul24(fe4_wrap_up<C>(slotB + 9), RS) + h);
#pragma unroll
                    for (int k = 0; k < CH / 2; ++k) {
                        const float2 u = q2[k], v = q7[k], w = q9[k];
                        wk[2 * k] = fminf(fminf(u.x, v.x), w.x);
                        wk[2 * k + 1] = fminf(fminf(u.y, v.y), w.y);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        const int j = (h + k < R) ? h + k : R - 1;
                        wk[k] = fminf(fminf(ahead(j, 2), ahead(j, 7)), ahead(j, 9));
                    }
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
    ncand += (uint32_t)__popcll((unsigned long long)cm);              // (summed over the workgroup at the end)
    const unsigned long long cand = __ballot(cm != 0u);               // bit l: unit LU wave + l has a candidate
    FE4_STAMP(3);
    if (FE4_ABLATE & 1) return;

    // ---- sparse outputs: rows are R floats, moved 8 bytes per lane, RPI rows per wave instruction -------------------
    // (coalesced: the flagged units are ranked with popcounts, and every wave-instruction moves RPI of them, LPR lanes x
    // 8 bytes = one row each.  A lane storing its own unit's row from registers issues R / 2 stores that touch one line
    // per lane: store-issue bound, measured 5x slower.)
    const long long lo64 = -jstep, hi64 = a.out_n - jstep;            // elements [lo, hi) of this step's coordinates exist
    const int lo = lo64 <= 0 ? 0 : (lo64 > 0x7FFFFFF ? 0x7FFFFFF : (int)lo64);
    const int hi = hi64 <= 0 ? 0 : (hi64 > 0x7FFFFFF ? 0x7FFFFFF : (int)hi64);
    const int sub = fes_div_small<C::LPR>(lane), piece = lane - fes_mul24(sub, C::LPR);
    auto put2 = [&](float *dst, int rel, float2 u) __attribute__((always_inline)) {
        if (!edge || (rel >= lo && rel + 2 <= hi)) *reinterpret_cast<float2 *>(dst + rel) = u;
        else {
            if (rel >= lo && rel < hi) dst[rel] = u.x;
            if (rel + 1 >= lo && rel + 1 < hi) dst[rel + 1] = u.y;
        }
    };
    if (!(FE4_ABLATE & 16)) {
        // reference level: the unit of a candidate and the one after it (a wave's lane 0 cannot see the unit before it: always).
        // The values exist only in registers: the flagged lanes park them in LDS rows, a batch at a time, and the wave writes
        // the batch out RPI rows per instruction.  Where they park (as in am_k_fe3): wave 0 in the ring rows of the LPB units
        // before unit -LAGU of the step -- its own phase B, behind it in program order, was their last reader (the rows 48
        // chips back of its units), no other wave reads them, the next step's staging overwrites them after barrier B5;
        // the other waves in the small buffer (with two waves, wave 1 has all eight of its rows).
        constexpr bool PARK_RING = LPB >= 8;                          // (2 .. 8 Msps: a block is 2 .. 6 units, the buffer is larger)
        constexpr int BUF_ROWS = (PARK_RING && NW == 2) ? 8 : 4;
        const unsigned long long wa = (cand | (cand << 1) | 1ull) & C::LUMASK;
        const int nav = __popcll(wa);
        uint32_t *tab = L.TAB + wv * AM_WAVE;
        const bool ring = PARK_RING && wv == 0;                       // (uniform)
        const int RB = ring ? LPB : BUF_ROWS;
        const bool mine = ((wa >> lane) & 1ull) != 0ull;
        const int my_rank = __popcll(wa & ((1ull << lane) - 1ull));
        if (mine) tab[my_rank] = (uint32_t)lane;
        float *const dst = a.avg_sparse + jstep;
        auto park_row = [&](int r) __attribute__((always_inline)) -> float * {
            if (ring) return L.X + fes_mul24(fe4_wrap_dn<C>(slot0 - (LAGU + LPB) + r), RS);
            return L.AVS + fes_mul24(((PARK_RING && NW == 2) ? 0 : wv * 4) + r, RS);
        };
        for (int b0 = 0; b0 < nav; b0 += RB) {                        // (uniform trip count)
            if (mine && my_rank >= b0 && my_rank < b0 + RB) {
                float2 *d = reinterpret_cast<float2 *>(park_row(my_rank - b0));
#pragma unroll
                for (int k = 0; k < R / 2; ++k) { float2 u; u.x = avgv[2 * k]; u.y = avgv[2 * k + 1]; d[k] = u; }
            }
            __builtin_amdgcn_wave_barrier();
            const int bend = (b0 + RB < nav) ? b0 + RB : nav;
            for (int r0 = b0; r0 < bend; r0 += C::RPI) {              // (uniform)
                const int r = r0 + sub;
                if (sub < C::RPI && r < bend) {
                    const int tu = fes_mul24(wv, LU) + (int)tab[r];
                    put2(dst, fes_mul24(tu, R) + 2 * piece, *reinterpret_cast<const float2 *>(park_row(r - b0) + 2 * piece));
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (!(FE4_ABLATE & 32) && a.bb_sparse != nullptr) {                // (uniform; null: am_k_gather_wg forms the rows from the samples)
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
                if (wv == NW - 1) L.CARRY[par] = ov;
                else L.WOV[par * NW + wv] = ov;
            }
            fes_barrier();
            if (wv == 0) need |= (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)L.CARRY[par ^ 1]);
            else need |= (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)L.WOV[par * NW + wv - 1]);
        } else {
            // (16 spare lanes: a wave reaches into the next wave's units itself)
            if (wv == 0) need |= (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)L.CARRY[par ^ 1]);
            if (wv == NW - 1) {
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
                const int tu = fes_mul24(wv, LU) + (int)tab[r];       // test index of the unit
                const int slot = fe4_wrap_dn<C>(fe4_wrap_up<C>(slot0 + tu) - LAGU);
                put2(dst, fes_mul24(tu, R) + 2 * piece, *reinterpret_cast<const float2 *>(L.X + fes_mul24(slot, RS) + 2 * piece));
            }
        }
    }
}

template <int SPC, int G, int NW>
__global__ void __launch_bounds__(AM_WAVE * NW, (fe4_cfg<SPC, G, NW>::MINW)) am_k_fe4(am_fe4_args a)
{
    using C = fe4_cfg<SPC, G, NW>;
    HIP_DYNAMIC_SHARED(unsigned char, smem);
    const fe4_smem<SPC, G, NW> L = fe4_smem_at<SPC, G, NW>(reinterpret_cast<float *>(smem));
    const int tid0 = threadIdx.x;
    const int sb = (int)(blockIdx.x * a.steps_per_wg);
    if (sb >= (int)a.nsteps) return;
    const int se = (sb + (int)a.steps_per_wg < (int)a.nsteps) ? sb + (int)a.steps_per_wg : (int)a.nsteps;
    // rings start empty; the first step's unit 0 has no predecessor (its bb is never used)
    for (int i = tid0; i < C::LDS_FLOATS; i += C::NT) L.X[i] = 0.0f;
    fes_barrier();
    if (tid0 < 2) L.CARRY[tid0] = 0xFFFFu;                            // the bb of a segment's first units is always written
    fes_barrier();
    int slot0 = 0, par = 0;
    fe4_prof PR;
#if defined(FE4_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    for (int k = 0; k < 8; ++k) PR.acc[k] = 0;
    PR.last = (long long)__builtin_readcyclecounter();
#endif
    float mxrun = 0.0f;                                               // largest bb this thread has formed
    bool badrun = false;                                              // ... or one that is not finite
    uint32_t ncand = 0;                                               // candidates this thread's units held
    // the step before the segment only rebuilds the rings: its first tested unit is unit US - LAGU, whose reference level
    // reaches back LPB units
    constexpr int WARM_J0 = (C::US - C::LAGU - C::LPB - 1) / C::UPR;
    for (int step = sb - 1; step < se; ++step) {
        const bool test = step >= sb;
        const bool have = step >= a.raw_lo && step < a.raw_hi;        // the step's raw samples are all present and 16-byte aligned
        const bool edge = !have || (test && !(step >= a.test_lo && step < a.test_hi));
        int tid = tid0;                                               // (nothing derived from the thread index lives across iterations: registers)
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(tid));
#endif
        fes_step_priority((unsigned)(step - sb + 1));                   // (am_fe_stream.h: the CU's workgroups end together)
        FE4_STAMP(4);
        if (have) {
            if (test) fe4_stage_rows<SPC, G, NW, 0>(a, L, a.out_abs0 + (long long)step * C::T, slot0, tid);
            else fe4_stage_rows<SPC, G, NW, WARM_J0>(a, L, a.out_abs0 + (long long)step * C::T, slot0, tid);
        } else
            fe4_stage_guarded<SPC, G, NW>(a, L, a.out_abs0 + (long long)step * C::T, slot0, tid);
        FE4_STAMP(5);
        fes_barrier();                                                // B1: |.|^2 of this step staged
        FE4_STAMP(0);
        fe4_step<SPC, G, NW>(a, L, step, test, slot0, par, edge, tid, mxrun, badrun, ncand, PR);
        slot0 = fe4_wrap_up<C>(slot0 + C::US);
        par ^= 1;
        FE4_STAMP(6);
        fes_barrier();                                                // B5: every ring read of this step done
    }
    // the largest sample of the segment (with the units the ring rebuild went through): +inf if one was not finite
    {
        float wmx = mxrun;
        for (int o = 32; o >= 1; o >>= 1) wmx = fmaxf(wmx, __shfl_xor(wmx, o, AM_WAVE));
        const bool bad = __ballot(badrun) != 0ull;
        uint32_t wcnt = ncand;
        for (int o = 32; o >= 1; o >>= 1) wcnt += (uint32_t)__shfl_xor((int)wcnt, o, AM_WAVE);
        if ((tid0 & (AM_WAVE - 1)) == 0) {
            L.WMX[tid0 / AM_WAVE] = bad ? __builtin_inff() : wmx;
            L.TAB[tid0 / AM_WAVE] = wcnt;
        }
        fes_barrier();
        if (tid0 == 0) {
            float v = L.WMX[0];
            uint32_t n = L.TAB[0];
            for (int w = 1; w < NW; ++w) { v = fmaxf(v, L.WMX[w]); n += L.TAB[w]; }
            a.wg_max[blockIdx.x] = v;
            a.wg_cnt[blockIdx.x] = n;
        }
    }
#if defined(FE4_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    if (a.prof && (tid0 & (AM_WAVE - 1)) == 0)
        for (int k = 0; k < 8; ++k) a.prof[((size_t)blockIdx.x * NW + tid0 / AM_WAVE) * 8 + k] = PR.acc[k];
#endif
}

// ---- host side ----------------------------------------------------------------------------------------
// 64 Msps runs am_k_fe3 (am_fe3.hip): the same machine with one chip per lane, two waves of 48 chip lanes and 16-byte LDS rows.
// am_k_fe4<32, 1, NW> computes the same results and was measured against it in round 4 (profiles/r4_fe64: 8-byte rows cost
// 20 %, and three waves of 64 lanes -- no idle lane, 4 blocks per step -- another 3 %); tuning builds (-DFE4_64MSPS) still
// run it.
#if defined(FE4_64MSPS)
#define FE4_USE_FE3(spc) false
#else
#define FE4_USE_FE3(spc) ((spc) == 32)
#endif
// the specialisations that exist: chips per lane and waves per workgroup by samples per chip
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
    case 32: return 1;      // 64 Msps: 32
    default: return 0;      // (everything else: the rate-generic kernels)
    }
}
static int fe4_nw_of(int spc) { return spc == 32 ? FE4_NW64 : 2; }
int am_fe4_supported(int spc) { return fe4_g_of(spc) != 0 ? 1 : 0; }
unsigned am_fe4_waves(int spc)                                                               // segments (waves) per step
{
    if (FE4_USE_FE3(spc)) return am_fe3_waves();
    return fe4_g_of(spc) ? (unsigned)fe4_nw_of(spc) : 0u;
}
unsigned am_fe4_unit(int spc) { return (unsigned)(spc * fe4_g_of(spc)); }                    // R: positions per bitmap word
unsigned am_fe4_words(int spc)                                                               // bitmap words (units) per step and wave
{
    if (FE4_USE_FE3(spc)) return AM_CHIPS_AVG;
    const int g = fe4_g_of(spc);
    return g ? (unsigned)(((AM_WAVE * fe4_nw_of(spc)) % (AM_CHIPS_AVG / g) == 0) ? AM_WAVE : AM_CHIPS_AVG) : 0u;
}
unsigned am_fe4_tile(int spc) { return am_fe4_waves(spc) * am_fe4_words(spc) * am_fe4_unit(spc); }   // positions per step
unsigned am_fe4_lag(int spc) { const int g = fe4_g_of(spc); return g ? (unsigned)((1 + 8 / g) * spc * g) : 0u; }
unsigned am_fe4_steps(long long out_n, int spc)
{
    const long long T = am_fe4_tile(spc);
    return T ? (unsigned)((out_n + am_fe4_lag(spc) + T - 1) / T) : 0u;
}

template <int SPC, int G, int NW>
static hipError_t fe4_launch(am_fe4_args &a, unsigned *steps_per_wg, hipStream_t s, int wgs_per_cu)
{
    using C = fe4_cfg<SPC, G, NW>;
    const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float);
    // workgroups per CU: what registers and LDS of THIS instantiation allow (4 at 64 Msps -- 40 KB of LDS, three waves --, 6
    // to 8 below), asked of the runtime once per device
    static std::atomic<int> per_cu[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    int wpc = (dev >= 0 && dev < 64) ? per_cu[dev].load(std::memory_order_acquire) : 0;
    if (wpc <= 0) {
        hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void *>(&am_k_fe4<SPC, G, NW>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (rc != hipSuccess) return rc;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(&am_k_fe4<SPC, G, NW>), C::NT, lds) != hipSuccess ||
            nb <= 0)
            nb = FE4_WG_PER_CU;
        wpc = nb > 8 ? 8 : nb;
        if (dev >= 0 && dev < 64) per_cu[dev].store(wpc, std::memory_order_release);
    }
#if defined(AM_TEST_KNOBS)
    if (const char *e = getenv("AIRMODES_FE4_WGS_PER_CU"))
        if (atoi(e) > 0) wpc = atoi(e);
#endif
    if (wgs_per_cu > 0 && wgs_per_cu < wpc) wpc = wgs_per_cu;         // (am_pipe: room on every CU for other batches' tails, as am_launch_fe3)
    const unsigned resident = (unsigned)(wpc * am_device_cus());
    unsigned spw = (a.nsteps + resident - 1) / resident;
    if (spw < 4) spw = 4;
#ifdef FE4_FORCE_SPW
    spw = FE4_FORCE_SPW;                                              // tuning builds
#endif
    a.steps_per_wg = spw;
    *steps_per_wg = spw;
    const unsigned grid = (a.nsteps + spw - 1) / spw;
    a.prof = nullptr;
#if defined(FE4_PROFILE)
    // blocking; prints mean cycles per step and phase -- never in the default build
    if (hipMalloc(reinterpret_cast<void **>(&a.prof), (size_t)grid * NW * 8 * sizeof(long long)) != hipSuccess) a.prof = nullptr;
#endif
    hipLaunchKernelGGL((am_k_fe4<SPC, G, NW>), dim3(grid), dim3(C::NT), lds, s, a);
    hipError_t lrc = hipGetLastError();
#if defined(FE4_PROFILE)
    if (a.prof) {
        fprintf(stderr, "fe4<%d,%d,%d>: grid %u, %u steps per workgroup, %d bytes of LDS, %d workgroups per CU\n", SPC, G, NW, grid, spw,
                (int)lds, wpc);
        std::vector<long long> h((size_t)grid * NW * 8);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), a.prof, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        (void)hipFree(a.prof);
        static const char *names[8] = {"B1wait", "A pmf+totals+scans+ring", "B3wait", "B scans' rest+avg+test", "B5wait", "stage (wait loads, lds)",
                                       "sparse", "-"};
        for (int w = 0; w < NW; ++w) {
            double acc[8] = {};
            for (unsigned b = 0; b < grid; ++b)
                for (int k = 0; k < 8; ++k) acc[k] += (double)h[((size_t)b * NW + w) * 8 + k];
            const double steps = (double)grid * (double)(spw + 1);
            double tot = 0;
            for (int k = 0; k < 7; ++k) tot += acc[k];
            fprintf(stderr, "fe4 clocks/step wave %d (total %.0f):", w, tot / steps);
            static const int order[7] = {5, 0, 1, 2, 3, 6, 4};       // order of execution
            for (int k = 0; k < 7; ++k) fprintf(stderr, " %s:%.0f", names[order[k]], acc[order[k]] / steps);
            fprintf(stderr, "\n");
        }
    }
#endif
    return lrc;
}

hipError_t am_launch_fe4(int spc, const float *iq, long long src_abs0, long long src_abs1, long long out_abs0, long long out_n,
                         float *bb_sparse, float *avg_sparse, uint32_t j0, uint32_t j1, int use_pmf, float s1, float sL,
                         float thr_lin, uint32_t *bits, uint32_t *wg_cnt, float *wg_max, unsigned *nsteps,
                         unsigned *steps_per_wg, hipStream_t s, int wgs_per_cu, unsigned *n_long)
{
    if (!am_fe4_supported(spc)) return hipErrorInvalidValue;
    if (FE4_USE_FE3(spc))
        return am_launch_fe3(iq, src_abs0, src_abs1, out_abs0, out_n, bb_sparse, avg_sparse, j0, j1, use_pmf, s1, sL, thr_lin, bits,
                             wg_cnt, wg_max, nsteps, steps_per_wg, s, wgs_per_cu, n_long);
    am_fe4_args a;
    a.iq = iq; a.src_abs0 = src_abs0; a.src_abs1 = src_abs1; a.out_abs0 = out_abs0; a.out_n = out_n;
    a.bb_sparse = bb_sparse; a.avg_sparse = avg_sparse; a.j0 = j0; a.j1 = j1; a.bits = bits; a.wg_cnt = wg_cnt; a.wg_max = wg_max;
    a.use_pmf = use_pmf ? 1 : 0; a.s1 = s1; a.sL = sL; a.thr_lin = thr_lin;
    a.nsteps = am_fe4_steps(out_n, spc);
    a.prof = nullptr;
    *nsteps = a.nsteps;
    *steps_per_wg = 1;
    if (a.nsteps == 0) return hipSuccess;
    const long long T = am_fe4_tile(spc), lag = am_fe4_lag(spc);
    // steps loaded without guards: samples [out_abs0 + k T, + T) inside [src_abs0, src_abs1), source 16-byte aligned (the
    // parity of the offset is the same for every step: T is even)
    const bool aligned = ((reinterpret_cast<uintptr_t>(iq) + (uintptr_t)(out_abs0 - src_abs0) * 8u) & 15u) == 0 && (T % 2) == 0;
    auto clampi = [](long long v) { return (int)(v < -4 ? -4 : (v > 0x7FFFFFF0ll ? 0x7FFFFFF0ll : v)); };
    a.raw_lo = clampi(fes_ceil_div(src_abs0 - out_abs0, T));
    a.raw_hi = aligned ? clampi(fes_floor_div(src_abs1 - out_abs0, T)) : a.raw_lo;
    // steps whose tested positions [k T - lag, k T + T - lag) all lie in [j0, min(j1, out_n))
    const long long jhi = (long long)j1 < out_n ? (long long)j1 : out_n;
    a.test_lo = clampi(fes_ceil_div((long long)j0 + lag, T));
    a.test_hi = clampi(fes_floor_div(jhi + lag, T));
    switch (spc) {
    case 1: return fe4_launch<1, 24, 2>(a, steps_per_wg, s, wgs_per_cu);
    case 2: return fe4_launch<2, 16, 2>(a, steps_per_wg, s, wgs_per_cu);
    case 4: return fe4_launch<4, 8, 2>(a, steps_per_wg, s, wgs_per_cu);
    case 5: return fe4_launch<5, 6, 2>(a, steps_per_wg, s, wgs_per_cu);
    case 8: return fe4_launch<8, 4, 2>(a, steps_per_wg, s, wgs_per_cu);
    case 10: return fe4_launch<10, 3, 2>(a, steps_per_wg, s, wgs_per_cu);
    case 16: return fe4_launch<16, 2, 2>(a, steps_per_wg, s, wgs_per_cu);
    case 20: return fe4_launch<20, 1, 2>(a, steps_per_wg, s, wgs_per_cu);
#if defined(FE4_64MSPS)
    default: return fe4_launch<32, 1, FE4_NW64>(a, steps_per_wg, s, wgs_per_cu);
#else
    default: return hipErrorInvalidValue;
#endif
    }
}
