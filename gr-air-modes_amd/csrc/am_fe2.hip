// am_fe2.hip -- fused front end + preamble detection for gfx950, specialised per samples-per-chip.
//
// One workgroup (768 threads = 12 waves, one per CU) owns a tile of T = 768 * R samples; every thread
// owns a RUN of R = SPC * CPT consecutive samples (CPT whole chips) and keeps it in registers:
//
//   P1  IQ (HBM, 16 B per lane, coalesced) -> |.|^2 -> LDS X            (tile + halos)
//   P2  pulse matched filter: thread reads chip(q-1), its own chips from X, forms the in-chip
//       suffix/prefix sums in registers, bb -> registers; barrier; bb -> X in place
//   P3  chip totals (left->right and right->left) -> small LDS arrays
//   P4  per 48-chip block: exclusive prefix / suffix of chip totals (sequential, canonical order)
//   P5  reference level avg[n] in registers + first-stage preamble test (a6), branch-free from
//       wide LDS loads of the runs 2, 7 and 9 chips ahead -> candidate bitmap
//   P6  bb out (coalesced, from X), ordered candidate list (bitmap + block scan)
//       and avg[] for the runs that hold or follow a candidate (all the refinement kernels
//       am_k_energy / am_k_cand in am_kernels.hip need)
//
// LDS holds ONE float per sample (X) plus per-chip side arrays (about 120 KB at 64 Msps).  The left
// halo is one 48-chip block + one chip (6 % at 64 Msps) and is served from L2 because consecutive
// tiles are mapped to the same XCD; the right halo is the 9 chips the test looks ahead.
// Summation order = DESIGN.md section 3 (identical to the generic kernel am_k_frontend and to the
// oracle).  Reference: python/rx_path.py:38-54, lib/preamble_impl.cc:172-179.
#include "am_internal.h"

#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <type_traits>
#include <vector>

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#ifndef FE2_NT
#define FE2_NT 768                          /* threads per workgroup (tuning builds may override) */
#endif
#ifndef FE2_WPS
#define FE2_WPS 3                          /* launch bound: waves per SIMD */
#endif
#ifndef FE2_NO_PREFIX_CSE
#define FE2_NO_PREFIX_CSE 0                 /* 1: P5 recomputes the in-chip prefix sums (register diet, see P5) */
#endif
#define FE2_RH_CHIPS 9                       /* the first-stage test looks 9 chips (+ 1 sample) ahead */
#define FE2_LH_CHIPS (AM_CHIPS_AVG + 1)      /* 48-chip block + one chip                 */
#define FE2_HALO_THREADS (AM_CHIPS_AVG + FE2_RH_CHIPS)

// LDS padding: 4 spare words per 32 keep 16-byte alignment for ds_read_b128 while spreading
// lanes that walk 32-float runs over all banks.
__device__ __forceinline__ int fe2_pidx(int i) { return i + ((i >> 5) << 2); }
__host__ __device__ constexpr int fe2_padn(int n) { return n + ((n >> 5) << 2) + 8; }
__host__ __device__ constexpr int fe2_lhp(int lh) { return (lh + 1 + 31) & ~31; }

// INGROUP: the caller guarantees that the N words lie inside one 32-word padding group, so the
// padded index is one computation plus constant offsets (the per-word form costs ~3 VALU each).
template <int N, bool ALIGNED, bool INGROUP = false>
__device__ __forceinline__ void fe2_lds_load(const float *X, int base, float (&v)[N])
{
    const float *P = X + (INGROUP ? fe2_pidx(base) : 0);
    if constexpr (ALIGNED && (N % 4 == 0)) {
#pragma unroll
        for (int k = 0; k < N / 4; ++k) {
            const float4 t = *reinterpret_cast<const float4 *>(INGROUP ? &P[4 * k] : &X[fe2_pidx(base + 4 * k)]);
            v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = INGROUP ? P[k] : X[fe2_pidx(base + k)];
    }
}

template <int N, bool ALIGNED, bool INGROUP = false>
__device__ __forceinline__ void fe2_lds_store(float *X, int base, const float (&v)[N])
{
    float *P = X + (INGROUP ? fe2_pidx(base) : 0);
    if constexpr (ALIGNED && (N % 4 == 0)) {
#pragma unroll
        for (int k = 0; k < N / 4; ++k) {
            float4 t;
            t.x = v[4 * k]; t.y = v[4 * k + 1]; t.z = v[4 * k + 2]; t.w = v[4 * k + 3];
            *reinterpret_cast<float4 *>(INGROUP ? &P[4 * k] : &X[fe2_pidx(base + 4 * k)]) = t;
        }
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) (INGROUP ? P[k] : X[fe2_pidx(base + k)]) = v[k];
    }
}

// bb of one chip from |.|^2 of the previous chip (mp) and of this chip (mc):
//   bb[i] = fl( (suf_prev[i+1] + pre[i]) * s1 ), last sample: pre only     
// (in place: mp is turned into its own suffix sums, mc from |iq|^2 into bb -- register diet)
template <int SPC>
__device__ __forceinline__ void fe2_pmf_chip(float (&mp)[SPC], float *mc, float s1)
{
    float acc = 0.0f;
#pragma unroll
    for (int i = SPC - 1; i >= 0; --i) { acc = acc + mp[i]; mp[i] = acc; }
    acc = 0.0f;
#pragma unroll
    for (int i = 0; i < SPC; ++i) {
        acc = acc + mc[i];
        const float s = (i == SPC - 1) ? acc : (mp[(i + 1 < SPC) ? i + 1 : i] + acc);
        mc[i] = s * s1;
    }
}

// tile -> global, 16 bytes per lane (the output arrays are 16-byte aligned and tiles start at
// multiples of 4 samples); the ragged end of the stream falls back to scalar stores
template <int T, bool EDGE>
__device__ __forceinline__ void fe2_store_tile(const float *X, int lhp, float *dst, long long o0, long long out_n,
                                               int tid)
{
    static_assert(T % 4 == 0, "tile length");
    if constexpr (!EDGE) {
        float4 *d4 = reinterpret_cast<float4 *>(dst + o0);       // uniform base, 32-bit lane offsets
        static_assert((T / 4) % FE2_NT == 0, "whole rounds");
        const float *P = X + fe2_pidx(lhp + 4 * tid);           // lhp and 4*768 are multiples of 32
        constexpr int KSTEP = 4 * FE2_NT + (((4 * FE2_NT) >> 5) << 2);
#pragma unroll
        for (int k = 0; k < T / 4 / FE2_NT; ++k)
            d4[(unsigned)(tid + k * FE2_NT)] = *reinterpret_cast<const float4 *>(&P[k * KSTEP]);
        return;
    }
#pragma unroll 4
    for (int i4 = tid; i4 < T / 4; i4 += FE2_NT) {
        const long long o = o0 + 4 * i4;
        const float4 t = *reinterpret_cast<const float4 *>(&X[fe2_pidx(lhp + 4 * i4)]);
        if (o + 3 < out_n) {
            *reinterpret_cast<float4 *>(&dst[o]) = t;
        } else {
            if (o < out_n) dst[o] = t.x;
            if (o + 1 < out_n) dst[o + 1] = t.y;
            if (o + 2 < out_n) dst[o + 2] = t.z;
        }
    }
}

// Profiling builds only (-DFE2_PROFILING=1, tools/build_variants.sh): phase stamps (lane 0 of wave 0 and of wave 5),
// phase ablation, start-up stagger and LDS reservation from the environment.  The default build contains none of
// it: no environment variable can change what the kernel computes.
#ifndef FE2_PROFILING
#define FE2_PROFILING 0
#endif
#if FE2_PROFILING
#define FE2_ABL(a) ((a).ablate)
#define FE2_STAMP(k)                                                                            \
    do {                                                                                       \
        if (a.clk && (tid == 0 || tid == 5 * AM_WAVE))                                          \
            a.clk[(size_t)tile * 32 + (tid ? 16 : 0) + (k)] = (long long)clock64();             \
    } while (0)
#else
#define FE2_ABL(a) 0u
#define FE2_STAMP(k) do { } while (0)
#endif

#include "am_fe_cmpx.h"

struct am_fe2_args {
    const float *iq;
    long long src_abs0, src_abs1;   // absolute range of samples present in iq
    long long out_abs0;             // absolute index of bb[0] (multiple of 48*spc)
    long long out_n;                // outputs wanted
    float *bb;                      // dense pulse-matched power (read by burst extraction); may be null
    float *avg;                     // dense reference level; only written when non-null (block-level API)
    float *avg_sparse;              // avg runs around candidates only (read by am_k_cand); may be null
    uint32_t j0, j1;                // positions (array coordinates) whose preamble test is wanted
    uint32_t *seg_pos;              // per tile: up to T candidate positions, in order
    uint32_t *blk_cnt;              // ntiles
    unsigned ntiles;
    int use_pmf;
    float s1, sL, thr_lin;
    unsigned ablate;                // profiling only (AIRMODES_FE2_ABLATE): skip phases, results invalid
    unsigned stagger, stagger_n;    // start-up delay (shader clocks) spread over the first stagger_n workgroups
    long long *clk;                 // profiling only (AIRMODES_FE2_CLOCK): per tile, 2 waves x 16 phase stamps
};

// One tile.  EDGE = false is the interior fast path: the tile, its halos and every position it
// tests lie inside the stream and inside the wanted output range, so no per-sample bounds
// predicate, clamp or 64-bit address survives (they were ~60 % of the VALU instructions, and
// this kernel is VALU-issue bound once its loads are batched).  EDGE = true keeps all of them
// and serves the first / last tiles of a stream and unaligned inputs.
template <int SPC, int CPT, bool EDGE>
__device__ __forceinline__ void fe2_tile(const am_fe2_args &a, const unsigned tile, float *smem)
{
    constexpr int R = SPC * CPT;                 // samples per thread
    constexpr int T = FE2_NT * R;                // samples per tile
    constexpr int LH = FE2_LH_CHIPS * SPC;
    constexpr int LHP = fe2_lhp(LH);             // tile starts 32-aligned in LDS, >= 1 spare slot before the halo
    constexpr int RH = FE2_RH_CHIPS * SPC;
    constexpr int W = LH + T + RH;
    constexpr int NCH = FE2_LH_CHIPS + FE2_NT * CPT + FE2_RH_CHIPS;   // chips resident
    constexpr int NBLK = 1 + (FE2_NT * CPT) / AM_CHIPS_AVG;           // 48-chip blocks incl. halo block
    constexpr bool RUN_AL = (R % 4 == 0);
    constexpr bool CHIP_AL = (SPC % 4 == 0);     // chips start on 16-byte LDS boundaries (LHP - LH is then a multiple of 4)
    static_assert(SPC % 4 != 0 || (LHP - LH) % 4 == 0, "chip alignment");
    constexpr bool SHIFT_AL = RUN_AL && ((2 * SPC) % 4 == 0) && ((7 * SPC) % 4 == 0) && ((9 * SPC) % 4 == 0);
    // padded-index shortcuts (LHP is a multiple of 32; chips sit at LHP - 49*SPC + c*SPC)
    constexpr bool RUN_IG = (32 % R == 0);       // a run lies inside one padding group
    constexpr bool CHIP_IG = (32 % SPC == 0);    // so does a chip
    constexpr int NWORDS = (T + 31) / 32;
    constexpr int WPT = (NWORDS + FE2_NT - 1) / FE2_NT;               // bitmap words per thread
    static_assert((FE2_NT * CPT) % AM_CHIPS_AVG == 0, "tile must be whole 48-chip blocks");
    static_assert(FE2_HALO_THREADS <= AM_WAVE, "halo chips are handled by one wave, one per lane");

    float *X = smem;                                        // [LHP + T + RH] padded
    float *TOT = X + fe2_padn(LHP + T + RH);                // chip totals, left->right   [NCH]
    float *RTOT = TOT + NCH;                                // chip totals, right->left   [NCH]
    float *PT = RTOT + NCH;
    float *ST = PT + NCH;
    uint32_t *BM = reinterpret_cast<uint32_t *>(ST + NCH);  // candidate bitmap [NWORDS]
    uint32_t *WS = BM + NWORDS;                             // wave sums for the block scan
    uint32_t *RUNANY = WS + 16;                             // per thread: does its run hold a candidate

    const int tid = threadIdx.x;
    const long long tile0 = a.out_abs0 + (long long)tile * T;
    const long long x0 = tile0 - LH;                        // absolute index of logical LDS index LHP-LH
    const long long rel0 = x0 - a.src_abs0;                 // the same as an offset into a.iq

    FE2_STAMP(0);
    // ---- P1: IQ -> |.|^2 -> X ---------------------------------------------------------------
    // All of a thread's loads are issued back to back before the first use (one HBM round trip
    // per tile instead of one per load).
    if constexpr (!EDGE) {
        // 16 bytes per lane from a uniform base: pair p holds samples 2p-sh, 2p-sh+1 of the window
        // (sh = 1 when the window starts on an odd sample: it is then loaded from one sample early)
        constexpr int NPI = (W + 2) / 2;
        constexpr int NP = (NPI + FE2_NT - 1) / FE2_NT;
        const int sh = (int)(rel0 & 1);
        const float4 *base = reinterpret_cast<const float4 *>(a.iq) + ((rel0 - sh) >> 1);
        float4 v[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            unsigned p = (unsigned)(tid + k * FE2_NT);
            if ((k + 1) * FE2_NT > NPI) p = p < (unsigned)NPI ? p : (unsigned)(NPI - 1);   // last round only
            v[k] = base[p];
        }
        // LDS slot of pair 0; pair p + 768 k lands 1536 k samples = 48 k padding groups further
        const int li0 = LHP - LH - sh + 2 * tid;
        const int q0 = fe2_pidx(li0), q1 = fe2_pidx(li0 + 1);
        constexpr int KSTEP = 2 * FE2_NT + (((2 * FE2_NT) >> 5) << 2);
        static_assert((2 * FE2_NT) % 32 == 0, "pair rounds keep the padding phase");
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const float r0 = v[k].x * v[k].x, i0 = v[k].y * v[k].y;
            const float r1 = v[k].z * v[k].z, i1 = v[k].w * v[k].w;
            if ((k + 1) * FE2_NT <= NPI || tid + k * FE2_NT < NPI) {
                X[q0 + k * KSTEP] = r0 + i0;                             // a1: fl(fl(I*I) + fl(Q*Q))
                X[q1 + k * KSTEP] = r1 + i1;
            }
        }
        for (int w = tid; w < NWORDS; w += FE2_NT) BM[w] = 0u;
    } else {
        // Loads are unconditional from a clamped (always valid) address so that they stay in one
        // basic block, and out-of-stream lanes are zeroed afterwards: a branch around a load would
        // force an s_waitcnt at its join.
        constexpr int NPAIR = (W + 1) / 2;
        constexpr int NP = (NPAIR + FE2_NT - 1) / FE2_NT;
        const float2 *iq2 = reinterpret_cast<const float2 *>(a.iq);
        const long long len = a.src_abs1 - a.src_abs0;
        const bool vec = (rel0 & 1) == 0 && (reinterpret_cast<uintptr_t>(a.iq) & 15u) == 0 && len >= 2;
        if (FE2_ABL(a) & 16u) {
            for (int i = tid; i < fe2_padn(LHP + T + RH); i += FE2_NT) X[i] = 1.0f;
        } else if (vec) {
            const float4 *iq4 = reinterpret_cast<const float4 *>(a.iq);
            const long long npf = len >> 1;                      // whole 16-byte pairs in the stream
            float4 v[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                long long pi = (rel0 >> 1) + (tid + k * FE2_NT);
                pi = pi < 0 ? 0 : (pi >= npf ? npf - 1 : pi);
                v[k] = iq4[pi];
            }
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int p = tid + k * FE2_NT;
                const long long n = x0 + 2 * p;
                if (p < NPAIR) {
                    const bool ok = n >= a.src_abs0 && n + 1 < a.src_abs1;
                    const float r0 = v[k].x * v[k].x, i0 = v[k].y * v[k].y;
                    const float r1 = v[k].z * v[k].z, i1 = v[k].w * v[k].w;
                    float m0 = ok ? (r0 + i0) : 0.0f;            // a1: fl(fl(I*I) + fl(Q*Q))
                    const float m1 = ok ? (r1 + i1) : 0.0f;
                    if (!ok && n == a.src_abs1 - 1 && n >= a.src_abs0) {   // odd stream length: last sample
                        const float2 t = iq2[n - a.src_abs0];
                        const float rr = t.x * t.x, ii = t.y * t.y;
                        m0 = rr + ii;
                    }
                    const int li = LHP - LH + 2 * p;
                    X[fe2_pidx(li)] = m0;
                    if (2 * p + 1 < W) X[fe2_pidx(li + 1)] = m1;
                }
            }
        } else {
            float2 v0[NP], v1[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                long long i0 = rel0 + 2 * (tid + k * FE2_NT), i1 = i0 + 1;
                i0 = i0 < 0 ? 0 : (i0 >= len ? len - 1 : i0);
                i1 = i1 < 0 ? 0 : (i1 >= len ? len - 1 : i1);
                v0[k].x = 0.0f; v0[k].y = 0.0f; v1[k] = v0[k];
                if (len > 0) { v0[k] = iq2[i0]; v1[k] = iq2[i1]; }     // uniform condition
            }
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int p = tid + k * FE2_NT;
                const long long n = x0 + 2 * p;
                if (p < NPAIR) {
                    const bool ok0 = n >= a.src_abs0 && n < a.src_abs1;
                    const bool ok1 = n + 1 >= a.src_abs0 && n + 1 < a.src_abs1;
                    const float r0 = v0[k].x * v0[k].x, q0 = v0[k].y * v0[k].y;
                    const float r1 = v1[k].x * v1[k].x, q1 = v1[k].y * v1[k].y;
                    const int li = LHP - LH + 2 * p;
                    X[fe2_pidx(li)] = ok0 ? (r0 + q0) : 0.0f;
                    if (2 * p + 1 < W) X[fe2_pidx(li + 1)] = ok1 ? (r1 + q1) : 0.0f;
                }
            }
        }
        for (int w = tid; w < NWORDS; w += FE2_NT) BM[w] = 0u;
    }
    FE2_STAMP(1);
    __syncthreads();
    FE2_STAMP(2);

    // logical LDS index of chip c (c = 0 is the extra halo chip, tile chips start at 49)
    auto chip_base = [](int c) __attribute__((always_inline)) { return LHP - LH + c * SPC; };
    const int c0 = FE2_LH_CHIPS + tid * CPT;               // first chip of this thread's run
    const int run_base = LHP + tid * R;                    // == chip_base(c0)

    float bbv[R];                                          // this thread's run of bb
    // halo chip handled additionally by the lanes of wave 0: 48 left-halo chips, 9 right-halo chips
    const int hq = (tid < AM_CHIPS_AVG) ? (1 + tid) : (FE2_LH_CHIPS + FE2_NT * CPT + (tid - AM_CHIPS_AVG));
    const bool has_halo = tid < FE2_HALO_THREADS;
    // absolute index of the last valid sample + 1, as a logical LDS index (bb beyond it reads 0)
    const long long end_li = a.src_abs1 - x0 + (LHP - LH);

    // ---- P2: pulse matched filter (a3), registers; P3: chip totals -----------------------------
    // halo chip first (its temporaries die before the run's registers come alive)
    const bool do_pmf = a.use_pmf && SPC > 1 && !(FE2_ABL(a) & 1u);
    float hb[SPC] = {};
    if (has_halo) {
        fe2_lds_load<SPC, CHIP_AL, CHIP_IG>(X, chip_base(hq), hb);
        if (do_pmf) {
            float hp[SPC];
            fe2_lds_load<SPC, CHIP_AL, CHIP_IG>(X, chip_base(hq) - SPC, hp);
            fe2_pmf_chip<SPC>(hp, &hb[0], a.s1);
        }
    }
    fe2_lds_load<R, RUN_AL, RUN_IG>(X, run_base, bbv);             // |.|^2 of the run
    if (do_pmf) {
        float mp[SPC];
        fe2_lds_load<SPC, CHIP_AL && RUN_AL, CHIP_IG>(X, run_base - SPC, mp);
        // last chip first: chip k still needs the raw |iq|^2 of chip k-1
#pragma unroll
        for (int k = CPT - 1; k >= 0; --k) {
            if (k == 0) {
                fe2_pmf_chip<SPC>(mp, &bbv[0], a.s1);
            } else {
                float prev[SPC];
#pragma unroll
                for (int i = 0; i < SPC; ++i) prev[i] = bbv[(k - 1) * SPC + i];
                fe2_pmf_chip<SPC>(prev, &bbv[k * SPC], a.s1);
            }
        }
    }
    if constexpr (EDGE) {
        // samples beyond the end of the stream read as zero (preamble view pads with zeros)
#pragma unroll
        for (int i = 0; i < R; ++i) if (run_base + i >= end_li) bbv[i] = 0.0f;
        if (has_halo) {
#pragma unroll
            for (int i = 0; i < SPC; ++i) if (chip_base(hq) + i >= end_li) hb[i] = 0.0f;
        }
    }
    // chip totals in both directions (canonical level-1 sums)
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        float f = 0.0f, b = 0.0f;
#pragma unroll
        for (int i = 0; i < SPC; ++i) f = f + bbv[k * SPC + i];
#pragma unroll
        for (int i = SPC - 1; i >= 0; --i) b = b + bbv[k * SPC + i];
        TOT[c0 + k] = f;
        RTOT[c0 + k] = b;
    }
    if (has_halo) {
        float f = 0.0f, b = 0.0f;
#pragma unroll
        for (int i = 0; i < SPC; ++i) f = f + hb[i];
#pragma unroll
        for (int i = SPC - 1; i >= 0; --i) b = b + hb[i];
        TOT[hq] = f;
        RTOT[hq] = b;
    }
    FE2_STAMP(3);
    __syncthreads();                                       // everyone has read its |.|^2
    FE2_STAMP(4);
    fe2_lds_store<R, RUN_AL, RUN_IG>(X, run_base, bbv);
    if (has_halo) fe2_lds_store<SPC, CHIP_AL, CHIP_IG>(X, chip_base(hq), hb);

    // ---- P4: exclusive prefix / suffix of chip totals inside each 48-chip block ----------------
    // (the 48 totals are fetched in one batch; the additions keep the canonical sequential order)
    if (!(FE2_ABL(a) & 2u))
    for (int idx = tid; idx < 2 * NBLK; idx += FE2_NT) {
        const int qb = 1 + AM_CHIPS_AVG * (idx >> 1);
        float t[AM_CHIPS_AVG];
#pragma unroll
        for (int j = 0; j < AM_CHIPS_AVG; ++j) t[j] = TOT[qb + j];
        float acc = 0.0f;
        if (idx & 1) {
#pragma unroll
            for (int j = AM_CHIPS_AVG - 1; j >= 0; --j) { const float v = t[j]; t[j] = acc; acc = acc + v; }
#pragma unroll
            for (int j = 0; j < AM_CHIPS_AVG; ++j) ST[qb + j] = t[j];
        } else {
#pragma unroll
            for (int j = 0; j < AM_CHIPS_AVG; ++j) { const float v = t[j]; t[j] = acc; acc = acc + v; }
#pragma unroll
            for (int j = 0; j < AM_CHIPS_AVG; ++j) PT[qb + j] = t[j];
        }
    }
    FE2_STAMP(5);
    __syncthreads();
    FE2_STAMP(6);

    const uint32_t jt0 = (uint32_t)(tile0 - a.out_abs0);   // array coordinate of the tile start
    // bb out (coalesced, from X) as soon as it is complete in LDS: the writes drain under the reference
    // level / detection / list phases instead of holding the finished workgroup's resources
    if (a.bb && !(FE2_ABL(a) & 4u)) fe2_store_tile<T, EDGE>(X, LHP, a.bb, (long long)jt0, a.out_n, tid);

    // ---- P5: reference level (a4) + first-stage preamble test (a6) ------------------------------
#if defined(__HIP_DEVICE_COMPILE__) && FE2_NO_PREFIX_CSE
    // The in-chip prefix sums below repeat the left-to-right chain of the chip totals (P3).  Left alone,
    // the compiler keeps all R partial sums of that chain alive across three barriers (168 VGPRs at
    // 64 Msps; 72 spilled registers when the build is limited to 128) to save R additions here; making
    // the values opaque ends those live ranges (143 VGPRs; 6 spilled dwords at 128).
#pragma unroll
    for (int i = 0; i < R; ++i) asm volatile("" : "+v"(bbv[i]));
#endif
    float avgv[R];
    if (FE2_ABL(a) & 2u) {
#pragma unroll
        for (int i = 0; i < R; ++i) avgv[i] = bbv[i];
    } else
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int q = c0 + k;
        const int jb = (tid * CPT + k) % AM_CHIPS_AVG;     // chip index inside its 48-chip block
        // in-chip suffix sums of the chip 48 chips back
        float scv[SPC];
        {
            fe2_lds_load<SPC, CHIP_AL && RUN_AL, CHIP_IG>(X, chip_base(q - AM_CHIPS_AVG), scv);
            float acc = 0.0f;
#pragma unroll
            for (int i = SPC - 1; i >= 0; --i) { acc = acc + scv[i]; scv[i] = acc; }
        }
        const float pt = PT[q];
        const float st_a = ST[q - AM_CHIPS_AVG];
        const float suf_last = RTOT[q - AM_CHIPS_AVG + 1] + ST[q - AM_CHIPS_AVG + 1];
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < SPC; ++i) {
            acc = acc + bbv[k * SPC + i];
            const float PRE = pt + acc;
            float s;
            if (i == SPC - 1) s = (jb == AM_CHIPS_AVG - 1) ? PRE : (suf_last + PRE);
            else s = (scv[i + 1] + st_a) + PRE;
            avgv[k * SPC + i] = s * a.sL;
        }
    }
    if (a.avg && !(FE2_ABL(a) & 8u)) {
        // block-level API only (am_frontend_work): dense reference level, one 4*R-byte run per lane
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const long long o = (long long)jt0 + tid * R + i;
            if (!EDGE || o < a.out_n) a.avg[o] = avgv[i];
        }
    }
    FE2_STAMP(7);
    // a6, branch-free.  Each test is a per-sample bool (a lane mask in scalar registers, combined
    // on the scalar unit); the pulses 2, 7 and 9 chips ahead are the same run shifted, fetched with
    // wide LDS loads.  The result is folded into one bit per sample at the end.
    static_assert(R <= 32, "one candidate word per thread");
    uint32_t cm = 0u;
    if (!(FE2_ABL(a) & 64u)) {
        constexpr int CH = (R % 16 == 0) ? 16 : R;           // samples per pass (scalar register budget)
        static_assert(R % CH == 0, "passes tile the run");
        const float nxt = X[fe2_pidx(run_base + R)];
#if defined(FE2_CMPX)
        if constexpr (!EDGE && (CH % 8 == 0)) {       // interior tiles: EXEC-narrowing compares (fe2_peak8)
            static_assert(CH <= 16 && R <= 2 * CH, "at most two passes of at most two blocks of eight");
            auto pass = [&](auto hc) __attribute__((always_inline)) {
                constexpr int H = decltype(hc)::value;
                float thr[CH];
#pragma unroll
                for (int i = 0; i < CH; ++i) thr[i] = avgv[H + i] * a.thr_lin;        // preamble_impl.cc:173
                uint32_t part = 0u;
                fe2_peak8<H>(part, &bbv[H], (H + 8 < R) ? bbv[(H + 8 < R) ? H + 8 : 0] : nxt, &thr[0]);
                if constexpr (CH > 8)
                    fe2_peak8<H + 8>(part, &bbv[H + 8], (H + 16 < R) ? bbv[(H + 16 < R) ? H + 16 : 0] : nxt, &thr[8]);
                // later pulses (:177-179), only where some lane still has a survivor (see below)
                if (__ballot(part != 0u) != 0ull) {
                    constexpr bool AL = SHIFT_AL && (CH % 4 == 0), IG = (SPC % CH == 0) && (32 % CH == 0);
                    float t2[CH], t7[CH], t9[CH];
                    fe2_lds_load<CH, AL, IG>(X, run_base + 2 * SPC + H, t2);
                    fe2_lds_load<CH, AL, IG>(X, run_base + 7 * SPC + H, t7);
                    fe2_lds_load<CH, AL, IG>(X, run_base + 9 * SPC + H, t9);
#pragma unroll
                    for (int i = 0; i < CH; ++i) t2[i] = fminf(fminf(t2[i], t7[i]), t9[i]);
                    fe2_weak8<H>(part, &t2[0], &thr[0]);
                    if constexpr (CH > 8) fe2_weak8<H + 8>(part, &t2[8], &thr[8]);
                }
                cm |= part;
            };
            pass(std::integral_constant<int, 0>{});
            if constexpr (R > CH) pass(std::integral_constant<int, CH>{});
        } else
#endif
#pragma unroll
        for (int h = 0; h < R; h += CH) {
            bool c[CH];
            float thr[CH];
            bool any = false;
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const float x = bbv[h + i];
                thr[i] = avgv[h + i] * a.thr_lin;                        // preamble_impl.cc:173
                const float nx = (h + i + 1 < R) ? bbv[(h + i + 1 < R) ? h + i + 1 : h + i] : nxt;
                c[i] = (x > thr[i]) & !(nx > x);                         // :174, :175
                if constexpr (EDGE) {
                    const uint32_t j = jt0 + (uint32_t)(tid * R + h + i);
                    c[i] = c[i] & (j >= a.j0) & (j < a.j1);
                }
                any = any | c[i];
            }
            // The three later pulses must not be below the threshold (:177-179): one test on the
            // smallest of them.  fminf ignores a NaN operand exactly as `NaN < thr` is false, so
            // this is the same predicate.  Wave-uniform early out: in quiet stretches no lane has
            // a survivor of the first test.
            if (__ballot(any) != 0ull) {
                constexpr bool AL = SHIFT_AL && (CH % 4 == 0), IG = (SPC % CH == 0) && (32 % CH == 0);
                float t2[CH], t7[CH], t9[CH];
                fe2_lds_load<CH, AL, IG>(X, run_base + 2 * SPC + h, t2);
                fe2_lds_load<CH, AL, IG>(X, run_base + 7 * SPC + h, t7);
                fe2_lds_load<CH, AL, IG>(X, run_base + 9 * SPC + h, t9);
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const float weakest = fminf(fminf(t2[i], t7[i]), t9[i]);
                    c[i] = c[i] & !(weakest < thr[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) cm |= c[i] ? (1u << (h + i)) : 0u;
        }
        if constexpr (R == 32) {
            BM[tid] = cm;
        } else {
            if (cm) {
                const int bit0 = tid * R;                       // first sample this word describes
                const unsigned long long wide = (unsigned long long)cm << (bit0 & 31);
                atomicOr(&BM[bit0 >> 5], (uint32_t)wide);
                if ((uint32_t)(wide >> 32)) atomicOr(&BM[(bit0 >> 5) + 1], (uint32_t)(wide >> 32));
            }
        }
        RUNANY[tid] = cm ? 1u : 0u;
    }
    FE2_STAMP(8);
    __syncthreads();
    FE2_STAMP(9);

    // the refinement kernels need avg[e] for e in a candidate's run or the next one;
    // write those runs only (plus the tile's first run, for candidates at the end of the previous tile)
    if (a.avg_sparse && !(FE2_ABL(a) & 64u)) {
        const bool need = tid == 0 || RUNANY[tid] != 0u || RUNANY[tid - 1] != 0u;
        if (need) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const long long o = (long long)jt0 + tid * R + i;
                if (!EDGE || o < a.out_n) a.avg_sparse[o] = avgv[i];
            }
        }
    }

    FE2_STAMP(10);
    // ---- P6: ordered candidate list ------------------------------------------------------------------
    FE2_STAMP(11);
    uint32_t total = 0;
    uint32_t *seg = a.seg_pos + (size_t)tile * T;
    {
        uint32_t words[WPT];
        uint32_t cnt = 0;
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            const int w = tid * WPT + k;
            words[k] = (w < NWORDS) ? BM[w] : 0u;
            cnt += (uint32_t)__popcll((unsigned long long)words[k]);
        }
        // block-exclusive scan of cnt: wave scan by shuffles, then wave totals through LDS
        const int lane = tid & (AM_WAVE - 1), wv = tid / AM_WAVE;
        uint32_t incl = cnt;
        for (int d = 1; d < AM_WAVE; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d, AM_WAVE);
            if (lane >= d) incl += up;
        }
        if (lane == AM_WAVE - 1) WS[wv] = incl;
        __syncthreads();
        uint32_t off = incl - cnt;
        for (int k = 0; k < FE2_NT / AM_WAVE; ++k) {
            if (k < wv) off += WS[k];
            total += WS[k];
        }
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            uint32_t wbits = words[k];
            const uint32_t jw = jt0 + (uint32_t)((tid * WPT + k) * 32);
            while (wbits) {
                const int b = __ffsll((long long)wbits) - 1;
                seg[off++] = jw + (uint32_t)b;
                wbits &= wbits - 1u;
            }
        }
        if (tid == 0) a.blk_cnt[tile] = total;
    }
    FE2_STAMP(12);
}

template <int SPC, int CPT>
__global__ void __launch_bounds__(FE2_NT, FE2_WPS) am_k_fe2(am_fe2_args a)
{
    constexpr int R = SPC * CPT, T = FE2_NT * R, LH = FE2_LH_CHIPS * SPC, RH = FE2_RH_CHIPS * SPC;
    constexpr int W = LH + T + RH;
    HIP_DYNAMIC_SHARED(float, smem);
    // XCD-aware tile order: workgroup b runs on XCD b % 8; give each XCD a contiguous range of
    // tiles so that neighbouring tiles (which share the halo) share an L2.  The order is reversed so
    // that the (slower) edge tiles at the end of the stream are dispatched first, not last.
    const unsigned nb = a.ntiles;
    const unsigned per = (nb + 7u) / 8u;
    const unsigned fwd = (blockIdx.x % 8u) * per + blockIdx.x / 8u;
    if (fwd >= nb) return;                                  // whole workgroup (uniform)
    const unsigned tile = nb - 1u - fwd;
    if (FE2_ABL(a) & 1024u) { if (threadIdx.x == 0) a.blk_cnt[tile] = 0; return; }
    if (a.stagger && blockIdx.x < a.stagger_n) {
        // first round of workgroups (one per CU): spread their start over `stagger` clocks so that the
        // CUs do not all load, and then all compute, at the same time
        const long long until = (long long)clock64() + (long long)(((blockIdx.x * 2654435769u) >> 8) % a.stagger);
        while ((long long)clock64() < until) __builtin_amdgcn_s_sleep(32);
    }
    const long long jt0 = (long long)tile * T;              // array coordinate of the tile start
    const long long rel0 = a.out_abs0 + jt0 - LH - a.src_abs0;
    const bool interior = rel0 >= 1 && rel0 + W + 2 <= a.src_abs1 - a.src_abs0 &&
                          (reinterpret_cast<uintptr_t>(a.iq) & 15u) == 0 && jt0 >= (long long)a.j0 &&
                          jt0 + T <= (long long)a.j1 && jt0 + T <= a.out_n && !(FE2_ABL(a) & (16u | 2048u));
    if (interior) fe2_tile<SPC, CPT, false>(a, tile, smem);
    else fe2_tile<SPC, CPT, true>(a, tile, smem);
}

static int ncu_for_stagger()
{
    static std::atomic<int> ncu_cache{0};
    int ncu = ncu_cache.load(std::memory_order_relaxed);
    if (ncu == 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
               prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        ncu_cache.store(ncu, std::memory_order_relaxed);
    }
    return ncu;
}

template <int SPC, int CPT>
static hipError_t fe2_launch(const am_fe2_args &a_in, hipStream_t s, unsigned *ntiles, unsigned *tile_len)
{
    constexpr int R = SPC * CPT, T = FE2_NT * R, LH = FE2_LH_CHIPS * SPC, LHP = fe2_lhp(LH);
    constexpr int RH = FE2_RH_CHIPS * SPC;
    constexpr int NCH = FE2_LH_CHIPS + FE2_NT * CPT + FE2_RH_CHIPS;
    constexpr int NWORDS = (T + 31) / 32;
    const size_t lds = ((size_t)fe2_padn(LHP + T + RH) + (size_t)4 * NCH + NWORDS + 16 + 16 + FE2_NT) *
                       sizeof(float);
    am_fe2_args a = a_in;
    size_t lds_req = lds;
#if FE2_PROFILING
    if (const char *x = getenv("AIRMODES_FE2_LDS_EXTRA")) lds_req += (size_t)atoi(x);   // occupancy experiments
#endif
    a.ntiles = (unsigned)((a.out_n + T - 1) / T);
    *ntiles = a.ntiles;
    *tile_len = T;
    if (a.ntiles == 0) return hipSuccess;
    // (once per device and LDS size: the call is not free and this launch is on the critical path)
    // (contexts may be driven from several host threads: the cache is atomic, setting the attribute twice is harmless)
    static std::atomic<size_t> attr_lds[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || attr_lds[dev].load(std::memory_order_acquire) != lds_req) {
        hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void *>(&am_k_fe2<SPC, CPT>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_req);
        if (rc != hipSuccess) return rc;
        if (dev >= 0 && dev < 64) attr_lds[dev].store(lds_req, std::memory_order_release);
    }
    const unsigned grid = ((a.ntiles + 7u) / 8u) * 8u;     // whole XCD rounds (extra groups exit)
    // Workgroups that start together stay in lockstep (same work per tile): the whole chip would load
    // (HBM saturated), then compute (HBM idle), in turns.  The first round of workgroups -- one per CU
    // -- therefore starts spread over about one tile time (measured optimum: 0.85-0.95 tile times,
    // -12 % kernel time at 64 Msps, -9 % at 20 Msps, nothing at 2-4 Msps where tiles are compute bound);
    // later workgroups inherit the offsets because each starts when its CU becomes free.
    a.stagger = (SPC >= 8) ? 1300u * (unsigned)R : 0u;
    a.stagger_n = (unsigned)ncu_for_stagger();
#if FE2_PROFILING
    if (const char *x = getenv("AIRMODES_FE2_STAGGER")) a.stagger = (unsigned)atoi(x);
#endif
    // profiling only: per-phase clock stamps, printed as average cycles between stamps (blocking)
#if FE2_PROFILING
    static const bool want_clk = getenv("AIRMODES_FE2_CLOCK") != nullptr;
#else
    const bool want_clk = false;
#endif
    a.clk = nullptr;
    if (want_clk && hipMalloc(reinterpret_cast<void **>(&a.clk), (size_t)a.ntiles * 32 * sizeof(long long)) != hipSuccess)
        a.clk = nullptr;
    hipLaunchKernelGGL((am_k_fe2<SPC, CPT>), dim3(grid), dim3(FE2_NT), lds_req, s, a);
    hipError_t lrc = hipGetLastError();
    if (a.clk) {
        std::vector<long long> h((size_t)a.ntiles * 32);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), a.clk, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        (void)hipFree(a.clk);
        for (int w = 0; w < 2; ++w) {
            double acc[13] = {};
            unsigned cnt = 0;
            for (unsigned t = 1; t + 2 < a.ntiles; ++t, ++cnt)       // interior tiles
                for (int k = 1; k <= 12; ++k) acc[k] += (double)(h[t * 32 + w * 16 + k] - h[t * 32 + w * 16 + k - 1]);
            fprintf(stderr, "fe2 clock wave %d:", w ? 5 : 0);
            for (int k = 1; k <= 12; ++k) fprintf(stderr, " %d:%.0f", k, cnt ? acc[k] / cnt : 0.0);
            fprintf(stderr, "\n");
        }
    }
    return lrc;
}

// Geometry only (buffer sizing): tile length for a supported spc, 0 if there is no specialisation.
unsigned am_fe2_tile(int spc)
{
    switch (spc) {
    case 1: return FE2_NT * 8;
    case 2: return FE2_NT * 8;
    case 4: return FE2_NT * 16;
    case 5: return FE2_NT * 20;
    case 8: return FE2_NT * 16;
    case 10: return FE2_NT * 20;
    case 16: return FE2_NT * 32;
    case 20: return FE2_NT * 20;
    case 32: return FE2_NT * 32;
    default: return 0;
    }
}

hipError_t am_launch_fe2(int spc, const float *iq, long long src_abs0, long long src_abs1, long long out_abs0,
                         long long out_n, float *bb, float *avg, uint32_t j0, uint32_t j1, int use_pmf, float s1,
                         float sL, float thr_lin, uint32_t *seg_pos, float *avg_sparse, uint32_t *blk_cnt,
                         unsigned *ntiles, unsigned *tile_len, hipStream_t s)
{
    am_fe2_args a;
    a.avg_sparse = avg_sparse;
    a.iq = iq; a.src_abs0 = src_abs0; a.src_abs1 = src_abs1; a.out_abs0 = out_abs0; a.out_n = out_n;
    a.bb = bb; a.avg = avg; a.j0 = j0; a.j1 = j1; a.seg_pos = seg_pos; a.blk_cnt = blk_cnt; a.ntiles = 0;
    a.use_pmf = use_pmf; a.s1 = s1; a.sL = sL; a.thr_lin = thr_lin;
    {
#if FE2_PROFILING
        const char *ab = getenv("AIRMODES_FE2_ABLATE");
        a.ablate = ab ? (unsigned)atoi(ab) : 0u;
#else
        a.ablate = 0u;
#endif
    }
    switch (spc) {
    // (SPC, chips per thread): run = SPC*CPT samples per thread, chosen so that the per-chip
    // side arrays and the sample array together stay <= 80 KB of LDS (two workgroups per CU)
    case 1: return fe2_launch<1, 8>(a, s, ntiles, tile_len);
    case 2: return fe2_launch<2, 4>(a, s, ntiles, tile_len);
    case 4: return fe2_launch<4, 4>(a, s, ntiles, tile_len);
    case 5: return fe2_launch<5, 4>(a, s, ntiles, tile_len);
    case 8: return fe2_launch<8, 2>(a, s, ntiles, tile_len);
    case 10: return fe2_launch<10, 2>(a, s, ntiles, tile_len);
    case 16: return fe2_launch<16, 2>(a, s, ntiles, tile_len);
    case 20: return fe2_launch<20, 1>(a, s, ntiles, tile_len);
    case 32: return fe2_launch<32, 1>(a, s, ntiles, tile_len);
    default: return hipErrorInvalidValue;
    }
}
