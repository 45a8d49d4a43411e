// am_fe2.hip -- fused front end + preamble detection for gfx950, specialised per samples-per-chip.
//
// One workgroup (384 threads = 6 waves) owns a tile of T = 384 * R samples; every thread owns a
// RUN of R = SPC * CPT consecutive samples (CPT whole chips) and keeps it in registers:
//
//   P1  IQ (HBM, 16 B per lane, coalesced) -> |.|^2 -> LDS X            (tile + halos)
//   P2  pulse matched filter: thread reads chip(q-1), its own chips from X, forms the in-chip
//       suffix/prefix sums in registers, bb -> registers; barrier; bb -> X in place
//   P3  chip totals (left->right and right->left) -> small LDS arrays
//   P4  per 48-chip block: exclusive prefix / suffix of chip totals (sequential, canonical order)
//   P5  reference level avg[n] in registers + first-stage preamble test (a6) -> LDS bitmap
//   P6  bb (coalesced, from X), ordered candidate list (bitmap + block scan), avg (staged via X)
//
// LDS holds ONE float per sample (X), so a 12 K-sample tile fits twice per CU; the left halo is
// one 48-chip block + one chip (13 % at 64 Msps) and is served from L2 because consecutive
// tiles are mapped to the same XCD.  Summation order = DESIGN.md section 3 (identical to the
// generic kernel am_k_frontend and to the oracle).  Reference: python/rx_path.py:38-54,
// lib/preamble_impl.cc:172-179.
#include "am_internal.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#define FE2_NT 384
#define FE2_RH_CHIPS 10                      /* detection looks ahead 9 chips + 1 sample */
#define FE2_LH_CHIPS (AM_CHIPS_AVG + 1)      /* 48-chip block + one chip                 */
#define FE2_HALO_THREADS (AM_CHIPS_AVG + FE2_RH_CHIPS)

// LDS padding: 4 spare words per 32 keep 16-byte alignment for ds_read_b128 while spreading
// lanes that walk 32-float runs over all banks.
__device__ __forceinline__ int fe2_pidx(int i) { return i + ((i >> 5) << 2); }
__host__ __device__ constexpr int fe2_padn(int n) { return n + ((n >> 5) << 2) + 8; }

template <int N, bool ALIGNED>
__device__ __forceinline__ void fe2_lds_load(const float *X, int base, float (&v)[N])
{
    if constexpr (ALIGNED && (N % 4 == 0)) {
#pragma unroll
        for (int k = 0; k < N / 4; ++k) {
            const float4 t = *reinterpret_cast<const float4 *>(&X[fe2_pidx(base + 4 * k)]);
            v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = X[fe2_pidx(base + k)];
    }
}

template <int N, bool ALIGNED>
__device__ __forceinline__ void fe2_lds_store(float *X, int base, const float (&v)[N])
{
    if constexpr (ALIGNED && (N % 4 == 0)) {
#pragma unroll
        for (int k = 0; k < N / 4; ++k) {
            float4 t;
            t.x = v[4 * k]; t.y = v[4 * k + 1]; t.z = v[4 * k + 2]; t.w = v[4 * k + 3];
            *reinterpret_cast<float4 *>(&X[fe2_pidx(base + 4 * k)]) = t;
        }
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) X[fe2_pidx(base + k)] = v[k];
    }
}

// bb of one chip from |.|^2 of the previous chip (mp) and of this chip (mc):
//   bb[i] = fl( (suf_prev[i+1] + pre[i]) * s1 ), last sample: pre only     
template <int SPC>
__device__ __forceinline__ void fe2_pmf_chip(const float (&mp)[SPC], const float *mc, float s1, float *out)
{
    float suf[SPC];
    float acc = 0.0f;
#pragma unroll
    for (int i = SPC - 1; i >= 0; --i) { acc = acc + mp[i]; suf[i] = acc; }
    acc = 0.0f;
#pragma unroll
    for (int i = 0; i < SPC; ++i) {
        acc = acc + mc[i];
        const float s = (i == SPC - 1) ? acc : (suf[i + 1] + acc);
        out[i] = s * s1;
    }
}

struct am_fe2_args {
    const float *iq;
    long long src_abs0, src_abs1;   // absolute range of samples present in iq
    long long out_abs0;             // absolute index of bb[0]/avg[0] (multiple of 48*spc)
    long long out_n;                // outputs wanted
    float *bb;
    float *avg;
    uint32_t j0, j1;                // positions (array coordinates) whose preamble test is wanted
    uint32_t *cand_seg;             // ntiles * T
    uint32_t *blk_cnt;              // ntiles
    unsigned ntiles;
    int use_pmf;
    float s1, sL, thr_lin;
};

template <int SPC, int CPT>
__global__ void __launch_bounds__(FE2_NT) am_k_fe2(am_fe2_args a)
{
    constexpr int R = SPC * CPT;                 // samples per thread
    constexpr int T = FE2_NT * R;                // samples per tile
    constexpr int LH = FE2_LH_CHIPS * SPC;
    constexpr int LHP = (LH + 31) & ~31;         // tile starts 32-aligned in LDS
    constexpr int RH = FE2_RH_CHIPS * SPC;
    constexpr int NCH = FE2_LH_CHIPS + FE2_NT * CPT + FE2_RH_CHIPS;   // chips resident
    constexpr int NBLK = 1 + (FE2_NT * CPT) / AM_CHIPS_AVG;           // 48-chip blocks incl. halo block
    constexpr bool RUN_AL = (R % 4 == 0);
    constexpr bool CHIP_AL = (SPC % 4 == 0);
    constexpr int NWORDS = (T + 31) / 32;
    constexpr int WPT = (NWORDS + FE2_NT - 1) / FE2_NT;               // bitmap words per thread
    static_assert((FE2_NT * CPT) % AM_CHIPS_AVG == 0, "tile must be whole 48-chip blocks");

    HIP_DYNAMIC_SHARED(float, smem);
    float *X = smem;                                        // [LHP + T + RH] padded
    float *HB = X + fe2_padn(LHP + T + RH);                 // halo bb staging: 58 chips
    float *TOT = HB + FE2_HALO_THREADS * SPC;               // chip totals, left->right   [NCH]
    float *RTOT = TOT + NCH;                                // chip totals, right->left   [NCH]
    float *PT = RTOT + NCH;
    float *ST = PT + NCH;
    uint32_t *BM = reinterpret_cast<uint32_t *>(ST + NCH);  // candidate bitmap [NWORDS]
    uint32_t *WS = BM + NWORDS;                             // wave sums for the block scan

    const int tid = threadIdx.x;
    // XCD-aware tile order: workgroup b runs on XCD b % 8; give each XCD a contiguous range of
    // tiles so that neighbouring tiles (which share the halo) share an L2.
    const unsigned nb = a.ntiles;
    const unsigned per = (nb + 7u) / 8u;
    const unsigned tile = (blockIdx.x % 8u) * per + blockIdx.x / 8u;
    if (tile >= nb) return;                                 // whole workgroup (uniform)
    const long long tile0 = a.out_abs0 + (long long)tile * T;
    const long long x0 = tile0 - LH;                        // absolute index of logical LDS index LHP-LH

    // ---- P1: IQ -> |.|^2 -> X ---------------------------------------------------------------
    {
        constexpr int W = LH + T + RH;
        const float2 *iq2 = reinterpret_cast<const float2 *>(a.iq);
        const long long rel0 = x0 - a.src_abs0;
        if ((rel0 & 1) == 0 && (reinterpret_cast<uintptr_t>(a.iq) & 15u) == 0) {
            // two complex samples (16 bytes) per lane per load
            const float4 *iq4 = reinterpret_cast<const float4 *>(a.iq);
            for (int p = tid; p < W / 2; p += FE2_NT) {
                const long long n = x0 + 2 * p;
                float m0 = 0.0f, m1 = 0.0f;
                if (n >= a.src_abs0 && n + 1 < a.src_abs1) {
                    const float4 v = iq4[(n - a.src_abs0) >> 1];
                    const float r0 = v.x * v.x, i0 = v.y * v.y, r1 = v.z * v.z, i1 = v.w * v.w;
                    m0 = r0 + i0;
                    m1 = r1 + i1;
                } else {
                    if (n >= a.src_abs0 && n < a.src_abs1) {
                        const float2 v = iq2[n - a.src_abs0];
                        const float rr = v.x * v.x, ii = v.y * v.y;
                        m0 = rr + ii;
                    }
                    if (n + 1 >= a.src_abs0 && n + 1 < a.src_abs1) {
                        const float2 v = iq2[n + 1 - a.src_abs0];
                        const float rr = v.x * v.x, ii = v.y * v.y;
                        m1 = rr + ii;
                    }
                }
                const int li = LHP - LH + 2 * p;
                X[fe2_pidx(li)] = m0;
                X[fe2_pidx(li + 1)] = m1;
            }
            if ((W & 1) && tid == 0) {
                const long long n = x0 + (W - 1);
                float m = 0.0f;
                if (n >= a.src_abs0 && n < a.src_abs1) {
                    const float2 v = iq2[n - a.src_abs0];
                    const float rr = v.x * v.x, ii = v.y * v.y;
                    m = rr + ii;
                }
                X[fe2_pidx(LHP - LH + W - 1)] = m;
            }
        } else {
            for (int i = tid; i < W; i += FE2_NT) {
                const long long n = x0 + i;
                float m = 0.0f;
                if (n >= a.src_abs0 && n < a.src_abs1) {
                    const float2 v = iq2[n - a.src_abs0];
                    const float rr = v.x * v.x, ii = v.y * v.y;
                    m = rr + ii;
                }
                X[fe2_pidx(LHP - LH + i)] = m;
            }
        }
        for (int w = tid; w < NWORDS; w += FE2_NT) BM[w] = 0u;
    }
    __syncthreads();

    // logical LDS index of chip c (c = 0 is the extra halo chip, tile chips start at 49)
    auto chip_base = [](int c) { return LHP - LH + c * SPC; };
    const int c0 = FE2_LH_CHIPS + tid * CPT;               // first chip of this thread's run
    const int run_base = LHP + tid * R;                    // == chip_base(c0)

    float bbv[R];                                          // this thread's run of bb
    // halo chip handled additionally by threads 0..57: 48 left-halo chips, 10 right-halo chips
    const int hq = (tid < AM_CHIPS_AVG) ? (1 + tid) : (FE2_LH_CHIPS + FE2_NT * CPT + (tid - AM_CHIPS_AVG));
    const bool has_halo = tid < FE2_HALO_THREADS;
    // absolute index of the last valid sample + 1, as a logical LDS index (bb beyond it reads 0)
    const long long end_li = a.src_abs1 - x0 + (LHP - LH);

    // ---- P2: pulse matched filter (a3), registers; P3: chip totals -----------------------------
    fe2_lds_load<R, RUN_AL>(X, run_base, bbv);             // |.|^2 of the run
    float hb[SPC] = {};
    if (has_halo) fe2_lds_load<SPC, false>(X, chip_base(hq), hb);
    if (a.use_pmf && SPC > 1) {
        float mp[SPC];
        fe2_lds_load<SPC, CHIP_AL && RUN_AL>(X, run_base - SPC, mp);
        float out[R];
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            if (k == 0) {
                fe2_pmf_chip<SPC>(mp, &bbv[0], a.s1, &out[0]);
            } else {
                float prev[SPC];
#pragma unroll
                for (int i = 0; i < SPC; ++i) prev[i] = bbv[(k - 1) * SPC + i];
                fe2_pmf_chip<SPC>(prev, &bbv[k * SPC], a.s1, &out[k * SPC]);
            }
        }
#pragma unroll
        for (int i = 0; i < R; ++i) bbv[i] = out[i];
        if (has_halo) {
            float hp[SPC], ho[SPC];
            fe2_lds_load<SPC, false>(X, chip_base(hq) - SPC, hp);
            fe2_pmf_chip<SPC>(hp, &hb[0], a.s1, &ho[0]);
#pragma unroll
            for (int i = 0; i < SPC; ++i) hb[i] = ho[i];
        }
    }
    // samples beyond the end of the stream read as zero (preamble view pads with zeros)
#pragma unroll
    for (int i = 0; i < R; ++i) if (run_base + i >= end_li) bbv[i] = 0.0f;
    if (has_halo) {
#pragma unroll
        for (int i = 0; i < SPC; ++i) if (chip_base(hq) + i >= end_li) hb[i] = 0.0f;
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
    __syncthreads();                                       // everyone has read its |.|^2
    fe2_lds_store<R, RUN_AL>(X, run_base, bbv);
    if (has_halo) fe2_lds_store<SPC, false>(X, chip_base(hq), hb);

    // ---- P4: exclusive prefix / suffix of chip totals inside each 48-chip block ----------------
    for (int idx = tid; idx < 2 * NBLK; idx += FE2_NT) {
        const int qb = 1 + AM_CHIPS_AVG * (idx >> 1);
        float acc = 0.0f;
        if (idx & 1) {
            for (int j = AM_CHIPS_AVG - 1; j >= 0; --j) { ST[qb + j] = acc; acc = acc + TOT[qb + j]; }
        } else {
            for (int j = 0; j < AM_CHIPS_AVG; ++j) { PT[qb + j] = acc; acc = acc + TOT[qb + j]; }
        }
    }
    __syncthreads();

    // ---- P5: reference level (a4) + first-stage preamble test (a6) ------------------------------
    float avgv[R];
    const uint32_t jt0 = (uint32_t)(tile0 - a.out_abs0);   // array coordinate of the tile start
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int q = c0 + k;
        const int jb = (tid * CPT + k) % AM_CHIPS_AVG;     // chip index inside its 48-chip block
        // in-chip suffix sums of the chip 48 chips back
        float scv[SPC];
        {
            float pv[SPC];
            fe2_lds_load<SPC, CHIP_AL && RUN_AL>(X, chip_base(q - AM_CHIPS_AVG), pv);
            float acc = 0.0f;
#pragma unroll
            for (int i = SPC - 1; i >= 0; --i) { acc = acc + pv[i]; scv[i] = acc; }
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
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const float x = bbv[i];
        const float thr = avgv[i] * a.thr_lin;                           // preamble_impl.cc:173
        if (x > thr) {                                                   // :174
            const uint32_t j = jt0 + (uint32_t)(tid * R + i);
            if (j >= a.j0 && j < a.j1) {
                const int li = run_base + i;
                const float nx = (i + 1 < R) ? bbv[(i + 1 < R) ? i + 1 : i] : X[fe2_pidx(li + 1)];
                if (!(nx > x) &&                                         // :175
                    !(X[fe2_pidx(li + 2 * SPC)] < thr) && !(X[fe2_pidx(li + 7 * SPC)] < thr) &&
                    !(X[fe2_pidx(li + 9 * SPC)] < thr))                  // :177-179
                    atomicOr(&BM[(tid * R + i) >> 5], 1u << ((tid * R + i) & 31));
            }
        }
    }
    __syncthreads();

    // ---- P6a: bb out (coalesced), ordered candidate list ------------------------------------------
    for (int i = tid; i < T; i += FE2_NT) {
        const long long o = (long long)jt0 + i;
        if (o < a.out_n) a.bb[o] = X[fe2_pidx(LHP + i)];
    }
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
        uint32_t off = incl - cnt, total = 0;
        for (int k = 0; k < FE2_NT / AM_WAVE; ++k) {
            if (k < wv) off += WS[k];
            total += WS[k];
        }
        uint32_t *seg = a.cand_seg + (size_t)tile * T;
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
    __syncthreads();                                       // X (bb) has been written out

    // ---- P6b: avg out, staged through X for coalesced stores ---------------------------------------
    fe2_lds_store<R, RUN_AL>(X, run_base, avgv);
    __syncthreads();
    for (int i = tid; i < T; i += FE2_NT) {
        const long long o = (long long)jt0 + i;
        if (o < a.out_n) a.avg[o] = X[fe2_pidx(LHP + i)];
    }
}

template <int SPC, int CPT>
static hipError_t fe2_launch(const am_fe2_args &a_in, hipStream_t s, unsigned *ntiles, unsigned *tile_len)
{
    constexpr int R = SPC * CPT, T = FE2_NT * R, LH = FE2_LH_CHIPS * SPC, LHP = (LH + 31) & ~31;
    constexpr int RH = FE2_RH_CHIPS * SPC;
    constexpr int NCH = FE2_LH_CHIPS + FE2_NT * CPT + FE2_RH_CHIPS;
    constexpr int NWORDS = (T + 31) / 32;
    const size_t lds = ((size_t)fe2_padn(LHP + T + RH) + (size_t)FE2_HALO_THREADS * SPC + (size_t)4 * NCH +
                        NWORDS + 16) * sizeof(float);
    am_fe2_args a = a_in;
    a.ntiles = (unsigned)((a.out_n + T - 1) / T);
    *ntiles = a.ntiles;
    *tile_len = T;
    if (a.ntiles == 0) return hipSuccess;
    hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void *>(&am_k_fe2<SPC, CPT>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc != hipSuccess) return rc;
    const unsigned grid = ((a.ntiles + 7u) / 8u) * 8u;     // whole XCD rounds (extra groups exit)
    hipLaunchKernelGGL((am_k_fe2<SPC, CPT>), dim3(grid), dim3(FE2_NT), lds, s, a);
    return hipGetLastError();
}

// Geometry only (buffer sizing): tile length for a supported spc, 0 if there is no specialisation.
unsigned am_fe2_tile(int spc)
{
    switch (spc) {
    case 1: return FE2_NT * 8;
    case 2: return FE2_NT * 16;
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
                         float sL, float thr_lin, uint32_t *cand_seg, uint32_t *blk_cnt, unsigned *ntiles,
                         unsigned *tile_len, hipStream_t s)
{
    am_fe2_args a;
    a.iq = iq; a.src_abs0 = src_abs0; a.src_abs1 = src_abs1; a.out_abs0 = out_abs0; a.out_n = out_n;
    a.bb = bb; a.avg = avg; a.j0 = j0; a.j1 = j1; a.cand_seg = cand_seg; a.blk_cnt = blk_cnt; a.ntiles = 0;
    a.use_pmf = use_pmf; a.s1 = s1; a.sL = sL; a.thr_lin = thr_lin;
    switch (spc) {
    // (SPC, chips per thread): run = SPC*CPT samples per thread, chosen so that the per-chip
    // side arrays and the sample array together stay <= 80 KB of LDS (two workgroups per CU)
    case 1: return fe2_launch<1, 8>(a, s, ntiles, tile_len);
    case 2: return fe2_launch<2, 8>(a, s, ntiles, tile_len);
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
