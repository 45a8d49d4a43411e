// am_kernels.hip -- gfx950 (MI355X, wave64) kernels of the Mode-S receive hot path.
//
// Everything here is HBM-bound streaming / sparse integer work: no MFMA.  Floating point
// follows the reference expression by expression (one IEEE rounding per written operation);
// the translation unit is built with -ffp-contract=off and contraction is also disabled by
// pragma so that  re*re + im*im,  sum*scale,  avg*thr  are never fused.
//
// Reference behaviour implemented (paths relative to the gr-air-modes tree):
//   front end        python/rx_path.py:38,49,54  (|.|^2, moving_average_ff(spc), (48*spc))
//   detect/refine    lib/preamble_impl.cc:172-209
//   greedy chain     lib/preamble_impl.cc:172,209,237  (scan order / skip semantics)
//   extract + tag    lib/preamble_impl.cc:100-137,219-232
//   slicer + CRC     lib/slicer_impl.cc:67-182, lib/modes_crc.cc:38-63
// The summation order inside the two moving averages is the chip-aligned two-level order
// of DESIGN.md section 3 (GNU Radio's own order depends on its scheduler).
#include <atomic>
#include <string.h>

#include "am_internal.h"
#include "am_fe_stream.h"

// keeps a loaded value where it is in the program (the compiler otherwise sinks loads to their first use, behind branches)
#if defined(__HIP_DEVICE_COMPILE__)
#define AM_PIN_U32(x) asm volatile("" : "+v"(x))
#else
#define AM_PIN_U32(x) ((void)0)
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

// LDS index padding: one spare word per 32 so that lanes walking their own chip (stride spc
// words, spc = 32 at 64 Msps) hit different banks.
#define PIDX(i) ((i) + ((i) >> 5))
#define PADN(n) ((n) + ((n) >> 5) + 1)

// ------------------------------------------------------------------------------------------
// Front end: one workgroup = one tile of `tile` outputs (tile is a multiple of the 48-chip
// block, tiles are aligned to the absolute sample index).  LDS holds the tile plus a left
// halo of one 48-chip block (for the reference-level window) plus one chip (for the pulse
// matched filter window):  X = |iq|^2 then bb,  S = in-chip suffix sums,  A = avg.
// ------------------------------------------------------------------------------------------
size_t am_fe_lds_bytes(int spc, int tile)
{
    const int L = AM_CHIPS_AVG * spc;
    const int W = L + spc + tile;
    size_t floats = (size_t)2 * PADN(W) + PADN(tile) + (size_t)3 * (W / spc + 1);
    return floats * sizeof(float);
}

int am_fe_pick_tile(int spc)
{
    const int L = AM_CHIPS_AVG * spc;
    int best = 0;
    for (int t = L; t <= (1 << 20); t += L) {
        if (am_fe_lds_bytes(spc, t) > AM_FE_LDS_BUDGET) break;
        best = t;
    }
    if (best == 0 && am_fe_lds_bytes(spc, L) <= 160 * 1024) best = L;
    return best;
}

__global__ void __launch_bounds__(AM_FE_THREADS) am_k_frontend(am_fe_args a)
{
    HIP_DYNAMIC_SHARED(float, smem);
    const int spc = a.spc;
    const int L = AM_CHIPS_AVG * spc;
    const int LH = L + spc;          // left halo: one block of bb + one chip of |iq|^2
    const int T = a.tile;
    const int W = LH + T;
    const int nch = W / spc;         // chips resident in LDS (chip 0 = the extra halo chip)
    float *X = smem;
    float *S = X + PADN(W);
    float *A = S + PADN(W);
    float *TOT = A + PADN(T);
    float *PT = TOT + (nch + 1);
    float *ST = PT + (nch + 1);
    const int tid = threadIdx.x;
    const int nt = blockDim.x;
    const long long tile0 = a.out_abs0 + (long long)blockIdx.x * T;
    const long long x0 = tile0 - LH;

    // P1: coalesced IQ load, a1: m = fl(fl(I*I) + fl(Q*Q)); zeros outside the stream
    const float2 *iq2 = reinterpret_cast<const float2 *>(a.iq);
    for (int i = tid; i < W; i += nt) {
        const long long n = x0 + i;
        float m = 0.0f;
        if (n >= a.src_abs0 && n < a.src_abs1) {
            const float2 v = iq2[n - a.src_abs0];
            const float rr = v.x * v.x;
            const float ii = v.y * v.y;
            m = rr + ii;
        }
        X[PIDX(i)] = m;
    }
    __syncthreads();

    // P2 (a3): pulse matched filter, window = one chip:
    //   bb[n] = fl( (suf_prevchip[n-spc+1] + pre_chip[n]) * s1 ),  last sample of a chip: pre only
    if (a.use_pmf && spc > 1) {
        for (int q = tid; q < nch; q += nt) {
            const int b = q * spc;
            float acc = 0.0f;
            for (int i = spc - 1; i >= 0; --i) {
                acc = acc + X[PIDX(b + i)];
                S[PIDX(b + i)] = acc;
            }
        }
        __syncthreads();
        for (int q = 1 + tid; q < nch; q += nt) {
            const int b = q * spc;
            float acc = 0.0f;
            for (int i = 0; i < spc; ++i) {
                acc = acc + X[PIDX(b + i)];
                const float s = (i == spc - 1) ? acc : (S[PIDX(b - spc + i + 1)] + acc);
                X[PIDX(b + i)] = s * a.s1;
            }
        }
        __syncthreads();
    }

    // P3: chip totals (left->right) and in-chip suffix sums (right->left) of bb
    for (int q = 1 + tid; q < nch; q += nt) {
        const int b = q * spc;
        float acc = 0.0f;
        for (int i = 0; i < spc; ++i) acc = acc + X[PIDX(b + i)];
        TOT[q] = acc;
        acc = 0.0f;
        for (int i = spc - 1; i >= 0; --i) {
            acc = acc + X[PIDX(b + i)];
            S[PIDX(b + i)] = acc;
        }
    }
    __syncthreads();

    // P4: per 48-chip block, exclusive prefix (PT) and exclusive suffix (ST) of chip totals
    const int nblk = 1 + T / L;
    for (int idx = tid; idx < 2 * nblk; idx += nt) {
        const int qb = 1 + AM_CHIPS_AVG * (idx >> 1);
        float acc = 0.0f;
        if (idx & 1) {
            for (int j = AM_CHIPS_AVG - 1; j >= 0; --j) { ST[qb + j] = acc; acc = acc + TOT[qb + j]; }
        } else {
            for (int j = 0; j < AM_CHIPS_AVG; ++j) { PT[qb + j] = acc; acc = acc + TOT[qb + j]; }
        }
    }
    __syncthreads();

    // P5 (a4): avg[n] = fl( (SUF[n-L+1] + PRE[n]) * sL ),  last sample of a block: PRE only
    for (int q = 1 + AM_CHIPS_AVG + tid; q < nch; q += nt) {
        const int b = q * spc;
        const int j = (q - 1) % AM_CHIPS_AVG;
        const float pt = PT[q];
        float acc = 0.0f;
        for (int i = 0; i < spc; ++i) {
            acc = acc + X[PIDX(b + i)];
            const float PRE = pt + acc;
            float s;
            if (j == AM_CHIPS_AVG - 1 && i == spc - 1) {
                s = PRE;
            } else {
                const int al = b + i - L + 1;
                const int qa = (i == spc - 1) ? (q - AM_CHIPS_AVG + 1) : (q - AM_CHIPS_AVG);
                const float SUF = S[PIDX(al)] + ST[qa];
                s = SUF + PRE;
            }
            A[PIDX(b + i - LH)] = s * a.sL;
        }
    }
    __syncthreads();

    // P6: coalesced stores
    for (int i = tid; i < T; i += nt) {
        const long long o = tile0 + i - a.out_abs0;
        if (o < a.out_n) {
            a.bb[o] = X[PIDX(LH + i)];
            a.avg[o] = A[PIDX(i)];
        }
    }
}

hipError_t am_launch_frontend(const am_fe_args &a, hipStream_t s)
{
    if (a.out_n <= 0) return hipSuccess;
    const size_t lds = am_fe_lds_bytes(a.spc, a.tile);
    const unsigned grid = (unsigned)((a.out_n + a.tile - 1) / a.tile);
    hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void *>(am_k_frontend),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc != hipSuccess) return rc;
    hipLaunchKernelGGL(am_k_frontend, dim3(grid), dim3(AM_FE_THREADS), lds, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Detection (a6): first-stage predicate at every position, compacted in position order with
// wave ballots.  Block b owns positions [j0 + b*2048, +2048) and a 2048-entry output segment,
// so the segment can never overflow.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(AM_DET_THREADS)
am_k_detect(const float *__restrict__ bb, const float *__restrict__ avg, uint32_t j0, uint32_t j1,
            uint32_t o1, uint32_t o2, uint32_t o3, float thr_lin, uint32_t *__restrict__ cand_seg,
            uint32_t *__restrict__ blk_cnt)
{
    __shared__ uint32_t wl[AM_DET_THREADS / AM_WAVE][AM_DET_PER_THREAD * AM_WAVE];
    __shared__ uint32_t wc[AM_DET_THREADS / AM_WAVE];
    const int lane = threadIdx.x & (AM_WAVE - 1);
    const int w = threadIdx.x / AM_WAVE;
    const uint32_t base = j0 + blockIdx.x * AM_DET_PER_BLOCK + w * (AM_DET_PER_THREAD * AM_WAVE);
    uint32_t cnt = 0;                                     // (o1, o2, o3: int(2 spc), int(7 spc), int(9 spc), preamble_impl.cc:158-162)
    for (int it = 0; it < AM_DET_PER_THREAD; ++it) {
        const uint32_t j = base + it * AM_WAVE + lane;
        bool c = false;
        if (j < j1) {
            const float x = bb[j];
            const float thr = avg[j] * thr_lin;             // preamble_impl.cc:173
            if (x > thr) {                                  // :174
                if (!(bb[j + 1] > x) &&                     // :175
                    !(bb[j + o1] < thr) && !(bb[j + o2] < thr) && !(bb[j + o3] < thr))   // :177-179
                    c = true;
            }
        }
        const unsigned long long m = __ballot(c);
        if (c) wl[w][cnt + __popcll(m & ((1ull << lane) - 1ull))] = j;
        cnt += (uint32_t)__popcll(m);
    }
    if (lane == 0) wc[w] = cnt;
    __syncthreads();
    uint32_t off = 0, total = 0;
    for (int k = 0; k < AM_DET_THREADS / AM_WAVE; ++k) {
        if (k < w) off += wc[k];
        total += wc[k];
    }
    uint32_t *seg = cand_seg + (size_t)blockIdx.x * AM_DET_PER_BLOCK;
    for (uint32_t i = lane; i < cnt; i += AM_WAVE) seg[off + i] = wl[w][i];
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = total;
}

hipError_t am_launch_detect(const float *bb, const float *avg, uint32_t j0, uint32_t j1, const am_geom &g,
                            float thr_lin, uint32_t *cand_seg, uint32_t *blk_cnt, uint32_t nblk,
                            hipStream_t s)
{
    if (nblk == 0) return hipSuccess;
    hipLaunchKernelGGL(am_k_detect, dim3(nblk), dim3(AM_DET_THREADS), 0, s, bb, avg, j0, j1, (uint32_t)g.o1, (uint32_t)g.o2,
                       (uint32_t)g.o3, thr_lin, cand_seg, blk_cnt);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Exclusive scan of block counts (single workgroup; the list is short: one entry per 2048
// positions or per 2048 flags).
// ------------------------------------------------------------------------------------------
#define AM_SCAN_THREADS 1024
__global__ void __launch_bounds__(AM_SCAN_THREADS)
am_k_scan_u32(const uint32_t *__restrict__ cnt, uint32_t *__restrict__ off, uint32_t n)
{
    // one workgroup: thread t owns the contiguous slice [t*per, (t+1)*per); slice sums are scanned
    // with wave shuffles + one LDS hop; then every thread writes its slice's exclusive offsets
    __shared__ uint32_t ws[AM_SCAN_THREADS / AM_WAVE];
    const int tid = threadIdx.x, lane = tid & (AM_WAVE - 1), wv = tid / AM_WAVE;
    const uint32_t per = (n + AM_SCAN_THREADS - 1) / AM_SCAN_THREADS;
    const uint32_t lo = (uint32_t)tid * per;
    const uint32_t hi = lo + per < n ? lo + per : n;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += cnt[i];
    uint32_t incl = sum;
    for (int d = 1; d < AM_WAVE; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d, AM_WAVE);
        if (lane >= d) incl += up;
    }
    if (lane == AM_WAVE - 1) ws[wv] = incl;
    __syncthreads();
    uint32_t base = incl - sum, total = 0;
    for (int k = 0; k < AM_SCAN_THREADS / AM_WAVE; ++k) { if (k < wv) base += ws[k]; total += ws[k]; }
    for (uint32_t i = lo; i < hi; ++i) { const uint32_t v = cnt[i]; off[i] = base; base += v; }
    if (tid == 0) off[n] = total;
}

hipError_t am_launch_scan_u32(const uint32_t *cnt, uint32_t *off, uint32_t n, hipStream_t s)
{
    hipLaunchKernelGGL(am_k_scan_u32, dim3(1), dim3(AM_SCAN_THREADS), 0, s, cnt, off, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Refinement (a7, a8): per candidate, the 4-pulse energy (double precision, chip-major order
// as in the reference) at the late shifts it actually visits, then the quiet-zone scan.
// ------------------------------------------------------------------------------------------
// four floats at a 4-byte aligned address in one 16-byte global load (gfx950 allows unaligned
// vector memory access; the packed type tells the compiler not to assume 16-byte alignment)
struct __attribute__((packed, aligned(4))) am_f4u { float v[4]; };

// 16 values are put in flight before they are consumed; the additions keep the reference's
// strict left-to-right double-precision order (preamble_impl.cc:91-98).
__device__ __forceinline__ double am_energy_chip(const float *__restrict__ p, int spc, double e)
{
    int i = 0;
    for (; i + 16 <= spc; i += 16) {
        am_f4u t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = *reinterpret_cast<const am_f4u *>(p + i + 4 * k);
#pragma unroll
        for (int k = 0; k < 16; ++k) e += (double)t[k >> 2].v[k & 3];
    }
    for (; i + 4 <= spc; i += 4) {
        float t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = p[i + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) e += (double)t[k];
    }
    for (; i < spc; ++i) e += (double)p[i];
    return e;
}

// (Putting two pulses' loads in flight at once -- two memory round trips per energy instead of eight -- made
// am_k_energy slower, 31 -> 37 us at the bench density: the kernel is bound by the 256 double-precision
// convert + add instructions per position, and the extra registers cost occupancy.)
__device__ __forceinline__ double am_preamble_energy(const float *__restrict__ p, int spc)
{
    double e = 0.0;
    e = am_energy_chip(p, spc, e);
    e = am_energy_chip(p + 2 * spc, spc, e);
    e = am_energy_chip(p + 7 * spc, spc, e);
    e = am_energy_chip(p + 9 * spc, spc, e);
    return e;
}

// any(z[0..n) > thr), 16 loads in flight per step
__device__ __forceinline__ bool am_any_above(const float *__restrict__ z, int n, float thr)
{
    int o = 0;
    for (; o + 16 <= n; o += 16) {
        am_f4u t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = *reinterpret_cast<const am_f4u *>(z + o + 4 * k);
        bool hit = false;
#pragma unroll
        for (int k = 0; k < 16; ++k) hit = hit || (t[k >> 2].v[k & 3] > thr);
        if (hit) return true;
    }
    for (; o < n; ++o) if (z[o] > thr) return true;
    return false;
}

// both quiet zones of a candidate (any sample above thr in z1[0..n1) or z2[0..n2)), 32 samples of each per memory
// round trip while both last, then the rest: 6 round trips for the 97 + 161 samples at 32 samples per chip, not 18
__device__ __forceinline__ bool am_any_above2(const float *__restrict__ z1, int n1, const float *__restrict__ z2, int n2,
                                              float thr)
{
    int o = 0;
    for (; o + 32 <= n1 && o + 32 <= n2; o += 32) {
        am_f4u ta[8], tb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) ta[k] = *reinterpret_cast<const am_f4u *>(z1 + o + 4 * k);
#pragma unroll
        for (int k = 0; k < 8; ++k) tb[k] = *reinterpret_cast<const am_f4u *>(z2 + o + 4 * k);
        bool hit = false;
#pragma unroll
        for (int k = 0; k < 32; ++k) hit = hit || (ta[k >> 2].v[k & 3] > thr) || (tb[k >> 2].v[k & 3] > thr);
        if (hit) return true;
    }
    return am_any_above(z1 + o, n1 - o, thr) || am_any_above(z2 + o, n2 - o, thr);
}

__global__ void __launch_bounds__(256)
am_k_refine(const float *__restrict__ bb, const float *__restrict__ avg, am_geom G, float thr_lin,
            const uint32_t *__restrict__ cand_seg, uint32_t seg_stride,
            const uint32_t *__restrict__ blk_off, uint32_t nblk, uint32_t M,
            uint32_t *__restrict__ pos, uint32_t *__restrict__ eo,
            uint32_t *__restrict__ tgt, float *__restrict__ inavg, uint8_t *__restrict__ valid)
{
    // One lane per candidate.  The reference recomputes both energies on every pass of its
    // do-while (preamble_impl.cc:184-192); the "now" energy of pass k+1 is the "late" energy of
    // pass k (same samples, same order, same double roundings), so it is carried over.
    // (Rate-generic: the geometry comes in the reference's own float arithmetic, am_geom.)
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= M) return;
    // segment that holds flat candidate g: last b with blk_off[b] <= g
    uint32_t lo = 0, hi = nblk;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (blk_off[mid] <= g) lo = mid; else hi = mid;
    }
    const uint32_t j = cand_seg[(size_t)lo * seg_stride + (g - blk_off[lo])];
    int how_late = 0;
    double e_now = am_preamble_energy(bb + j, G.S);
    for (;;) {
        const double e_next = am_preamble_energy(bb + j + how_late + 1, G.S);
        const bool late = e_next > e_now;
        if (late) { how_late++; e_now = e_next; }
        if (!(late && how_late < G.late_max)) break;            // :192  how_late < d_samples_per_chip
    }
    const uint32_t e = j + (uint32_t)how_late;
    // quiet zones (preamble_impl.cc:198-209)
    const float p0 = bb[e], p1 = bb[e + G.o1], p2 = bb[e + G.o2], p3 = bb[e + G.o3];
    const float av = avg[e];
    float ps = p0 + p1;
    ps = ps + p2;
    ps = ps + p3;
    const float avgpeak = (float)((double)ps / 4.0);
    const float sthr = av + (avgpeak - av) / thr_lin;
    const bool ok = !am_any_above(bb + e + G.za0, G.za1 - G.za0 + 1, sthr) &&      // offsets int(1.5 sps) .. 3 sps
                    !am_any_above(bb + e + G.zb0, G.zb1 - G.zb0 + 1, sthr);        // offsets int(5 sps) .. 7.5 sps
    pos[g] = j;
    eo[g] = e;
    inavg[g] = av;
    valid[g] = ok ? 1 : 0;
    tgt[g] = ok ? (e + (uint32_t)G.B) : (e + 1u);   // :237 / :209
}

hipError_t am_launch_refine(const float *bb, const float *avg, const am_geom &g, float thr_lin,
                            const uint32_t *cand_seg, uint32_t seg_stride, const uint32_t *blk_off,
                            uint32_t nblk, uint32_t M, uint32_t *pos, uint32_t *e, uint32_t *tgt,
                            float *inavg, uint8_t *valid, hipStream_t s)
{
    if (M == 0) return hipSuccess;
    hipLaunchKernelGGL(am_k_refine, dim3((M + 255) / 256), dim3(256), 0, s, bb, avg, g, thr_lin, cand_seg,
                       seg_stride, blk_off, nblk, M, pos, e, tgt, inavg, valid);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Split refinement (used after the fused front end).  A lane-per-candidate late-peak search makes
// every wave as slow as its slowest lane (3 % of the candidates slide the full spc steps) and
// recomputes the same energies for neighbouring candidates.  Instead the energies E(q) are
// computed ONCE per reachable position: candidate c (positions sorted) contributes the
// d[c] = min(spc+1, pos[c]-pos[c-1]) positions (pos[c]+spc-d[c], pos[c]+spc] that its
// predecessors did not already cover; an exclusive scan of d[] lays them out contiguously, so
// position pos[c]+s sits at index off[c]+d[c]-1-spc+s.  am_k_energy fills that array with one
// lane per position (consecutive lanes = consecutive positions: coalesced loads); am_k_cand then
// walks at most spc comparisons per candidate and does the quiet-zone test.
// ------------------------------------------------------------------------------------------
#define AM_SCAN_BLK 2048

// Candidate count: `cap` from the host (exact, or a capacity when the host speculates to avoid a
// round trip) clipped by the device-side count when one is given.
__device__ __forceinline__ uint32_t am_count(uint32_t cap, const uint32_t *__restrict__ Mp)
{
    if (!Mp) return cap;
    const uint32_t m = *Mp;
    return m < cap ? m : cap;
}

// Flat candidate positions from the streaming front ends' bitmap (am_fe3.hip / am_fe4.hip), one workgroup per FRONT-END
// workgroup (round 4).  Word w, bit b = array coordinate wbits * w + b - lag (wbits = 32 at 64 Msps, the unit length of
// am_k_fe4 otherwise).  Front-end workgroup g tested the words [g * words_per_wg, (g + 1) * words_per_wg) and left their
// candidate count in wg_cnt[g]: this workgroup's part of pos[] starts at wg_cnt[0] + ... + wg_cnt[g - 1], which it adds up
// itself (a few KB of counts) -- no scan launch in front of it, no chain between workgroups, nothing to wait for.  (Round 3:
// a chained scan of 41 664 per-(step, wave) counts, then one wave per 48-word segment: 6.3 + 14.9 us at the bench density.)
// A thread takes four consecutive words per round (one 16-byte load); the round's counts are scanned in the workgroup.
// Entries at or beyond Mcap (a capacity launch that was too small: the scan is redone) are dropped; *total_out = the
// number of candidates there are.
// ---- bb rows around candidates, rebuilt from IQ (round 5; 64 Msps) ------------------------------------------------------------
// Until round 4 am_k_fe3 wrote the pulse-matched power bb of the 17 chips from every candidate's chip on: ~42 MB of stores per
// 64 M-sample launch at the bench density, 12-15 us of a kernel that moves bytes at the rate the part can move them (DESIGN.md
// 5.1).  Those rows are formed HERE instead, by the workgroup that lists the segment's candidates anyway: it knows which chips the
// refinement will read (a candidate in bitmap word w -> chips w - 9 .. w + 7 of the array: bit b of word w is position
// 32 w + b - 288), loads exactly their samples (and the chip before each run: the filter's window reaches back one chip) and
// repeats phase A of am_k_fe3 in the canonical order: in-chip suffix sums of the chip before right->left, prefix
// sums of the own chip left->right, bb[n] = fl((suf + pre) s1); the chip's last sample: pre alone.  Same bits as the front
// end's ring rows (stage-level device tests compare every candidate record with the arrays NaN-poisoned).
// Machine mapping: the wanted chips of the segment are taken 32 at a time, DENSELY (a wave's lanes hold 32 wanted chips wherever
// they lie; the first version walked stripes of 32 consecutive chips of which 8 were wanted: 41 us of instruction issue at the
// bench density).  Every wave works on its own batches, no workgroup barrier: 16 lanes load one chip's 256 bytes (coalesced), all of
// a batch's loads in flight together; |.|^2 goes to the wave's LDS rows (stride 36 floats: 16-byte reads of consecutive rows hit
// all banks); lane j forms the prefix sums of chip j's row, lane j + 32 the suffix sums of the chip before it (the row of lane
// j - 1 where the wanted chips are neighbours -- they come in runs of 17 and more --, an extra row at the start of a run), lane j
// fetches across the wave what it needs and writes the row back in place; the rows leave eight per store instruction
// (8 lanes x 16 bytes = one 128-byte line each).
#define AM_ROWS_XS 36                     /* floats per LDS row: 32 + 4 pad */
#define AM_ROWS_BBW 17                    /* chips of bb the refinement reads from a candidate's chip on */
#define AM_ROWS_MAXW 2048                 /* bitmap words per pass of a workgroup (am_k_fe3: 14 x 96 words per workgroup) */
#ifndef AM_ROWS_BATCH
#define AM_ROWS_BATCH 32                  /* wanted chips a wave takes at a time (<= 32: lane j and lane j + 32 share a chip) */
#endif
#define AM_ROWS_XSTART 8                  /* rows for the chip before a run's first (a batch holds at most 3 run starts: see below) */
#define AM_ROWS_SLOTS (AM_ROWS_BATCH + AM_ROWS_XSTART)
#define AM_ROWS_NFL ((AM_ROWS_MAXW + 16) / 64 + 2)
#ifndef AM_ROWS_WPS
#define AM_ROWS_WPS 6                     /* waves per SIMD the gather + rows kernel is compiled for: all of am_k_fe3's 1 489 segments resident at once (67 VGPRs, 24 KB of LDS) */
#endif

// rows of the chips [w_begin - 9, w_end - 9) that some candidate in the words [w_begin - 16, w_end) asks for.  All threads of the
// workgroup call it (three barriers in front, none after).  NZ / FL: AM_ROWS_NFL words each; PFX: AM_ROWS_NFL + 1; XR: AM_ROWS_SLOTS
// rows per wave; RC: AM_ROWS_SLOTS entries per wave.
// the words [w_begin - 16, w_end) as the flags want them: word index i = (wave + 4 q) * 64 + lane of a 256-thread workgroup
#define AM_ROWS_NCH ((AM_ROWS_NFL + 3) / 4)                  /* chunks of 64 words per wave, four waves: 9 */
struct am_rows_words { uint32_t w[AM_ROWS_NCH]; };
// (two halves: the loads -- unconditional, from clamped indices, all in flight together -- and, where the caller has something else
// to wait for first, the masking.  Written as `valid ? bits[w] : 0` each load sat behind its own branch and was waited for on the
// spot -- the compiler keeps only "word != 0" --: nine serial memory round trips)
__device__ __forceinline__ void am_rows_issue_words(const uint32_t *__restrict__ bits, uint32_t w_begin, uint32_t w_end, am_rows_words &x)
{
    const int lane = threadIdx.x & (AM_WAVE - 1), wv = threadIdx.x / AM_WAVE, nwv = blockDim.x / AM_WAVE;
#pragma unroll
    for (int q = 0; q < AM_ROWS_NCH; ++q) {
        const uint32_t i = ((uint32_t)wv + (uint32_t)q * (uint32_t)nwv) * AM_WAVE + (uint32_t)lane;
        const long long w = (long long)w_begin - 16 + (long long)i;
        const long long wc = w < 0 ? 0 : (w >= (long long)w_end ? (long long)w_end - 1 : w);
        x.w[q] = bits[wc];
    }
}
__device__ __forceinline__ void am_rows_finish_words(uint32_t w_begin, uint32_t w_end, am_rows_words &x)
{
    const int lane = threadIdx.x & (AM_WAVE - 1), wv = threadIdx.x / AM_WAVE, nwv = blockDim.x / AM_WAVE;
    const uint32_t nidx = w_end - w_begin + 16u;
#pragma unroll
    for (int q = 0; q < AM_ROWS_NCH; ++q) {
        AM_PIN_U32(x.w[q]);
        const uint32_t i = ((uint32_t)wv + (uint32_t)q * (uint32_t)nwv) * AM_WAVE + (uint32_t)lane;
        const long long w = (long long)w_begin - 16 + (long long)i;
        if (!(i < nidx && w >= 0)) x.w[q] = 0u;
    }
}
__device__ __forceinline__ void am_rows_load_words(const uint32_t *__restrict__ bits, uint32_t w_begin, uint32_t w_end, am_rows_words &x)
{
    am_rows_issue_words(bits, w_begin, w_end, x);
    am_rows_finish_words(w_begin, w_end, x);
}

template <bool PMF>
__device__ __forceinline__ void am_rows_segment32(const am_rows_args &ra, const uint32_t *__restrict__ bits, uint32_t w_begin,
                                                  uint32_t w_end, unsigned long long *NZ, unsigned long long *FL, uint32_t *PFX,
                                                  float *XR_all, uint16_t *RC_all, const am_rows_words &early, bool use_early)
{
    constexpr int SPC = 32;
    const int tid = threadIdx.x, nwv = blockDim.x / AM_WAVE;
    const int lane = tid & (AM_WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(tid / AM_WAVE);     // (wave-uniform, and known to be: scalar arithmetic, uniform branches)
    const uint32_t nw = w_end - w_begin;                              // <= AM_ROWS_MAXW
    const uint32_t nidx = nw + 16u;                                   // index i <-> word w_begin - 16 + i <-> array chip w_begin - 25 + i
    const uint32_t n64 = (nidx + 63u) >> 6;
    // which words hold a candidate (one bit per word; a wave's ballot is 64 of them).  The loads of all of a wave's chunks go
    // out together (a load and its ballot per iteration were six serial memory round trips: ~12 us of this kernel) -- for the
    // workgroup's first pass at the very start of the kernel, under the listing of the candidates (`early`)
    {
        am_rows_words x = early;
        if (!use_early) am_rows_load_words(bits, w_begin, w_end, x);   // (uniform; a workgroup's later passes: test builds only)
#pragma unroll
        for (int q = 0; q < AM_ROWS_NCH; ++q) {
            const uint32_t c = (uint32_t)wv + (uint32_t)q * (uint32_t)nwv;
            const unsigned long long m = __ballot(x.w[q] != 0u);
            if (lane == 0 && c < n64) NZ[c] = m;
        }
    }
    __syncthreads();
    // a word with a candidate flags its chip and the 16 after it: dilation by 16 bits across the 64-bit words; kept: the chips
    // of THIS segment (indices 16 .. nidx - 1: what lies before belongs to the workgroup before)
    for (uint32_t j = (uint32_t)tid; j < n64 + 1u; j += blockDim.x) {
        unsigned long long f = 0ull;
        if (j < n64) {
            const unsigned long long x = NZ[j];
            unsigned long long d = x | (x << 1);
            d |= d << 2; d |= d << 4; d |= d << 8;                    // shifts 0 .. 15
            static_assert(AM_ROWS_BBW == 17, "a candidate's chip and the 16 after it");
            f = d | (x << 16);
            const uint32_t hp = j ? (uint32_t)(NZ[j - 1u] >> 48) : 0u;      // the previous word's last 16 words reach into this one
            if (hp) f |= (2ull << (31 - __clz((int)hp))) - 1ull;
            if (j == 0) f &= ~0xFFFFull;
            const uint32_t lim = nidx - 64u * j;                      // indices of this word below nidx
            if (lim < 64u) f &= (1ull << lim) - 1ull;
        }
        FL[j] = f;
    }
    __syncthreads();
    if ((uint32_t)tid <= n64) {                                       // (<= 34 words: every entry adds up the words before it, reads in flight together)
        uint32_t acc = 0;
        for (uint32_t k = 0; k < (uint32_t)tid; ++k) acc += (uint32_t)__popcll(FL[k]);
        PFX[tid] = acc;
    }
    __syncthreads();
    const uint32_t nf = PFX[n64];                                     // wanted chips of the segment
    float *const XR = XR_all + wv * (AM_ROWS_SLOTS * AM_ROWS_XS);
    uint16_t *const RC = RC_all + wv * AM_ROWS_SLOTS;                 // index i of the chip in row slot s
    const bool wide = (reinterpret_cast<uintptr_t>(ra.iq) & 15u) == 0 && (((ra.out_abs0 - ra.src_abs0) & 1) == 0);   // (uniform)
    const float2 *iq2 = reinterpret_cast<const float2 *>(ra.iq);
    // index 15 = the chip before the segment's first: the lowest one a row can ask for
    const long long Abase = ra.out_abs0 + ((long long)w_begin - 10) * SPC;
    const bool inside = wide && Abase >= ra.src_abs0 && Abase + (long long)(nw + 1u) * SPC <= ra.src_abs1;   // (uniform) every sample present, 16-byte aligned
    unsigned long long gb = reinterpret_cast<unsigned long long>(iq2 + (inside ? Abase - ra.src_abs0 : 0));
    gb = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(gb >> 32)) << 32) |
         (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)gb);   // (wave-uniform: the base stays in scalar registers)
    constexpr int MAXR = AM_ROWS_SLOTS * 16 / AM_WAVE;                // rounds of 64 pieces that cover all row slots: 10
    int lane_ = lane;
    for (uint32_t b0 = (uint32_t)wv * AM_ROWS_BATCH; b0 < nf; b0 += (uint32_t)nwv * AM_ROWS_BATCH) {          // (wave-uniform)
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(lane_));                               // (nothing derived from the lane index is to live across batches: hoisted,
#endif                                                                //  those addresses cost 40 registers and as many spills)
        const int lane = lane_;
        const int j = lane & 31, half = lane >> 5;
        const int cnt = (nf - b0 < AM_ROWS_BATCH) ? (int)(nf - b0) : AM_ROWS_BATCH;
        const bool live = j < cnt;
        // the (b0 + j)-th wanted chip: the word by bisection of the prefix counts, the bit by bisection of the word
        uint32_t ci;
        {
            const uint32_t r = b0 + (uint32_t)(live ? j : 0);
            uint32_t lo = 0, hi = n64;                                // last word with PFX <= r
            while (hi - lo > 1u) {
                const uint32_t mid = (lo + hi) >> 1;
                if (PFX[mid] <= r) lo = mid; else hi = mid;
            }
            uint32_t rr = r - PFX[lo];
            unsigned long long y = FL[lo];
            uint32_t pos = 0;
#pragma unroll
            for (int sh = 32; sh >= 1; sh >>= 1) {
                const unsigned long long lowpart = y & ((1ull << sh) - 1ull);
                const uint32_t c = (uint32_t)__popcll(lowpart);
                if (rr >= c) { rr -= c; y >>= sh; pos += (uint32_t)sh; } else y = lowpart;
            }
            ci = 64u * lo + pos;                                      // >= 16
        }
        // runs: the chip before chip j is lane j - 1's own chip unless a run starts at j.  Wanted chips come in runs of 17 and
        // more (every candidate word flags 17 consecutive chips; only a segment's first run can be cut shorter), so 32 consecutive
        // ones hold at most three run starts: the batch's first, the end of a cut first run, and one more 17 further on.
        const uint32_t cprev = (uint32_t)__shfl((int)ci, (lane + AM_WAVE - 1) & (AM_WAVE - 1), AM_WAVE);
        const bool start = live && (j == 0 || cprev + 1u != ci);
        const uint32_t smask = (uint32_t)__ballot(start);            // (lanes 0..31; 32..63 hold the same chips)
        int srank = __popc(smask & ((1u << j) - 1u));
        srank = srank < AM_ROWS_XSTART ? srank : AM_ROWS_XSTART - 1;  // (never: see above)
        const int nstart = PMF ? __popc(smask) : 0;
        const int nrow = cnt + (nstart < AM_ROWS_XSTART ? nstart : AM_ROWS_XSTART);
        __builtin_amdgcn_wave_barrier();                              // (RC / XR: the batch before is done with them)
        if (half == 0 && live) {
            RC[j] = (uint16_t)ci;
            if (PMF && start) RC[cnt + srank] = (uint16_t)(ci - 1u);
        }
        __builtin_amdgcn_wave_barrier();
        // |.|^2 of the rows: all loads of the batch in flight together
        const int np = nrow * 16;
        if (inside) {
            float4 v[MAXR];
            // (straight-line: a lane without a piece loads piece 0 again -- with a branch per round the register allocator
            // spilled the first loads right behind their issue, one memory round trip each)
#pragma unroll
            for (int r = 0; r < MAXR; ++r) {
                const int p = lane + AM_WAVE * r;
                const int pc = p < np ? p : 0;
                const unsigned off = ((unsigned)RC[pc >> 4] - 15u) * (unsigned)(SPC * 8) + (unsigned)(pc & 15) * 16u;
                v[r] = fes_gload16_cached_at(gb, off);
            }
#pragma unroll
            for (int r = 0; r < MAXR; ++r) {
                const int p = lane + AM_WAVE * r;
                if (p < np) {
                    const float r0 = v[r].x * v[r].x, i0 = v[r].y * v[r].y, r1 = v[r].z * v[r].z, i1 = v[r].w * v[r].w;
                    float2 mm;
                    mm.x = r0 + i0;                                   // a1: fl(fl(I*I) + fl(Q*Q))
                    mm.y = r1 + i1;
                    *reinterpret_cast<float2 *>(XR + (p >> 4) * AM_ROWS_XS + 2 * (p & 15)) = mm;
                }
            }
        } else {
            // stream edges / unaligned input: one sample at a time, zeros outside the stream (rare: kept small)
#pragma unroll 1
            for (int p = lane; p < np; p += AM_WAVE) {
                const long long a = Abase + ((long long)RC[p >> 4] - 15) * SPC + 2 * (p & 15);   // absolute index of the piece's first sample
                float2 u0, u1;
                u0.x = 0.0f; u0.y = 0.0f; u1 = u0;
                if (a >= ra.src_abs0 && a < ra.src_abs1) u0 = iq2[a - ra.src_abs0];
                if (a + 1 >= ra.src_abs0 && a + 1 < ra.src_abs1) u1 = iq2[a + 1 - ra.src_abs0];
                const float r0 = u0.x * u0.x, i0 = u0.y * u0.y, r1 = u1.x * u1.x, i1 = u1.y * u1.y;
                float2 mm;
                mm.x = r0 + i0;
                mm.y = r1 + i1;
                *reinterpret_cast<float2 *>(XR + (p >> 4) * AM_ROWS_XS + 2 * (p & 15)) = mm;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // positions beyond the end of the stream read as zero (the preamble view pads with zeros)
        const long long leftn = ra.src_abs1 - (Abase + ((long long)ci - 15) * SPC);
        const int nin = leftn >= SPC ? SPC : (leftn <= 0 ? 0 : (int)leftn);
        if (PMF) {
            // lane j: prefix sums of the own chip (row j), left->right; lane j + 32: suffix sums of the chip before (the row
            // reversed, so both run the same chain); bb[i] = fl((suf[i + 1] + pre[i]) s1)
            float c[SPC];
            if (live) {
                const int slot = half ? (start ? cnt + srank : j - 1) : j;
                const float4 *row = reinterpret_cast<const float4 *>(XR + slot * AM_ROWS_XS);
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
            __builtin_amdgcn_wave_barrier();                          // (every lane has read its row)
            float4 *own = reinterpret_cast<float4 *>(XR + j * AM_ROWS_XS);
            float rmx = 0.0f;                                         // the row's largest value (fmaxf: a NaN is no sample above anything)
#pragma unroll
            for (int k = 0; k < SPC / 4; ++k) {
                float o4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = 4 * k + q;
                    float tt = c[i];                                  // the chip's last sample: the window is the chip
                    if (i < SPC - 1) tt = __shfl(c[(i < SPC - 1) ? SPC - 2 - i : 0], lane ^ 32, AM_WAVE) + c[i];   // suf[i + 1] + pre[i]
                    o4[q] = (i >= nin) ? 0.0f : tt * ra.s1;
                    rmx = fmaxf(rmx, o4[q]);
                }
                if (live && half == 0) { float4 o; o.x = o4[0]; o.y = o4[1]; o.z = o4[2]; o.w = o4[3]; own[k] = o; }
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_sched_barrier(0);                    // (four positions at a time: hoisted, the 31 exchanges cost 31 registers more)
#endif
            }
            // the row's maximum beside the row: am_k_refine_late judges a chip that lies inside a quiet zone as a whole by it
            const long long chipa = (long long)w_begin - 25 + (long long)ci;          // array chip
            if (live && half == 0 && ra.bb_max && chipa >= 0 && chipa * SPC < ra.out_n) ra.bb_max[chipa] = rmx;
        } else if (live && half == 0) {
            // no filter: bb is |.|^2 itself; only the end of the stream needs a hand
            float *own = XR + j * AM_ROWS_XS;
            for (int i = nin; i < SPC; ++i) own[i] = 0.0f;
            float rmx = 0.0f;
            for (int i = 0; i < SPC; ++i) rmx = fmaxf(rmx, own[i]);
            const long long chipa = (long long)w_begin - 25 + (long long)ci;
            if (ra.bb_max && chipa >= 0 && chipa * SPC < ra.out_n) ra.bb_max[chipa] = rmx;
        }
        __builtin_amdgcn_wave_barrier();
        // the rows leave eight per store instruction
        const int sub = lane >> 3, piece = lane & 7;
        for (int r0 = 0; r0 < cnt; r0 += 8) {                         // (uniform trip count)
            const int r = r0 + sub;
            if (r < cnt) {
                const float4 u = *reinterpret_cast<const float4 *>(XR + r * AM_ROWS_XS + 4 * piece);
                const long long rel = ((long long)w_begin - 25 + (long long)RC[r]) * SPC + 4 * piece;   // array coordinate
                if (rel >= 0 && rel + 4 <= ra.out_n) *reinterpret_cast<float4 *>(ra.bb_sparse + rel) = u;
                else {
                    if (rel >= 0 && rel < ra.out_n) ra.bb_sparse[rel] = u.x;
                    if (rel + 1 >= 0 && rel + 1 < ra.out_n) ra.bb_sparse[rel + 1] = u.y;
                    if (rel + 2 >= 0 && rel + 2 < ra.out_n) ra.bb_sparse[rel + 2] = u.z;
                    if (rel + 3 >= 0 && rel + 3 < ra.out_n) ra.bb_sparse[rel + 3] = u.w;
                }
            }
        }
    }
}

template <int ROWS>       // 0: candidates only; 1 / 2: + the bb rows around them from IQ at 32 samples per chip (filter on / off)
__global__ void __launch_bounds__(256, (ROWS ? AM_ROWS_WPS : 8))       // (with rows: five waves per SIMD, <= 102 VGPRs; left alone the compiler took 162)
am_k_gather_wg(const uint32_t *__restrict__ bits, const uint32_t *__restrict__ wg_cnt, uint32_t nwg, uint32_t words_per_wg,
               uint32_t nwords, uint32_t Mcap, uint32_t lag, uint32_t wbits, uint32_t *__restrict__ pos,
               uint32_t *__restrict__ total_out, am_rows_args ra)
{
    __shared__ uint32_t ws[256 / AM_WAVE];
    __shared__ uint32_t red[256 / AM_WAVE];
    const uint32_t g = blockIdx.x;
    const int lane = threadIdx.x & (AM_WAVE - 1), wv = threadIdx.x / AM_WAVE;
    const uint32_t w_begin = g * words_per_wg;
    const uint32_t w_end = (w_begin + words_per_wg < nwords) ? w_begin + words_per_wg : nwords;
    // (uniform) 16-byte loads: the segment starts on a 16-byte boundary and is a whole number of four-word groups (always, for the
    // streaming front ends' bitmaps: 96 or 128 words per step)
    const bool vec = (reinterpret_cast<uintptr_t>(bits + w_begin) & 15u) == 0 && ((w_end - w_begin) & 3u) == 0u;
    // The load and, apart from it, what depends on it.  The 16-byte load is UNCONDITIONAL (a lane without a group loads the
    // segment's first one again; a segment that does not qualify loads from the 16-byte boundary below it and throws the result
    // away): round 5 found that written as `if (vec && inside) load16; else guarded loads` each of the two rounds below was waited
    // for on the spot (the join of the two paths does not know which loads are in flight) -- the "one memory round trip in front of
    // the arithmetic" of round 4 was three in the machine code.
    auto issue4 = [&](uint32_t w0) __attribute__((always_inline)) -> uint4 {
        const uint32_t wc = (w0 + 4u <= w_end) ? w0 : w_begin;
        return *reinterpret_cast<const uint4 *>(reinterpret_cast<uintptr_t>(bits + wc) & ~(uintptr_t)15);
    };
    auto finish4 = [&](uint32_t w0, uint4 t, uint32_t *x) __attribute__((always_inline)) {
        AM_PIN_U32(t.x); AM_PIN_U32(t.y); AM_PIN_U32(t.z); AM_PIN_U32(t.w);
        const bool in = w0 + 4u <= w_end;
        x[0] = in ? t.x : 0u; x[1] = in ? t.y : 0u; x[2] = in ? t.z : 0u; x[3] = in ? t.w : 0u;
        if (!vec) {                                           // (uniform; never for the streaming front ends' bitmaps)
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = (w0 + (uint32_t)k < w_end) ? bits[w0 + (uint32_t)k] : 0u;
        }
    };
    // the first two rounds' words (all of them at 64 Msps) go out together with the counts of the workgroups before this one:
    // one memory round trip in front of the arithmetic, not three
    uint32_t xa[4], xb[4];
    const uint4 ta = issue4(w_begin + 4u * threadIdx.x);
    const uint4 tb = issue4(w_begin + 4u * (threadIdx.x + blockDim.x));
    // (with rows: the words once more, laid out as the rows' chip flags want them -- the same cache lines, one more round of loads
    // under this one instead of a memory round trip of its own behind the listing)
    am_rows_words early;
    if constexpr (ROWS != 0)
        am_rows_issue_words(bits, w_begin, (w_begin + AM_ROWS_MAXW < w_end) ? w_begin + AM_ROWS_MAXW : w_end, early);
    // where this workgroup's candidates start: the counts of the workgroups before it, eight per thread and round trip (a plain
    // `acc += wg_cnt[k]` loop waits for every load before it issues the next: six serial round trips at 64 Msps)
    uint32_t acc = 0;
    for (uint32_t k0 = threadIdx.x; k0 < g; k0 += 8u * blockDim.x) {
        uint32_t v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t k = k0 + (uint32_t)j * blockDim.x;
            v[j] = wg_cnt[k < g ? k : 0u];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            AM_PIN_U32(v[j]);
            if (k0 + (uint32_t)j * blockDim.x < g) acc += v[j];
        }
    }
    for (int o = AM_WAVE / 2; o >= 1; o >>= 1) acc += (uint32_t)__shfl_xor((int)acc, o, AM_WAVE);
    if (lane == 0) red[wv] = acc;
    finish4(w_begin + 4u * threadIdx.x, ta, xa);              // (arrived with the counts)
    finish4(w_begin + 4u * (threadIdx.x + blockDim.x), tb, xb);
    if constexpr (ROWS != 0)
        am_rows_finish_words(w_begin, (w_begin + AM_ROWS_MAXW < w_end) ? w_begin + AM_ROWS_MAXW : w_end, early);
    __syncthreads();
    uint32_t run = 0;
    for (int k = 0; k < 256 / AM_WAVE; ++k) run += red[k];
    uint32_t round = 0;
    for (uint32_t c0 = w_begin; c0 < w_end; c0 += 4u * blockDim.x, ++round) {
        const uint32_t w0 = c0 + 4u * threadIdx.x;
        uint32_t x[4];
        if (round == 0) { x[0] = xa[0]; x[1] = xa[1]; x[2] = xa[2]; x[3] = xa[3]; }
        else if (round == 1) { x[0] = xb[0]; x[1] = xb[1]; x[2] = xb[2]; x[3] = xb[3]; }
        else finish4(w0, issue4(w0), x);
        const uint32_t c = (uint32_t)(__popc(x[0]) + __popc(x[1]) + __popc(x[2]) + __popc(x[3]));
        uint32_t incl = c;
        for (int d = 1; d < AM_WAVE; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d, AM_WAVE);
            if (lane >= d) incl += up;
        }
        __syncthreads();                                      // (ws may still be read from the round before)
        if (lane == AM_WAVE - 1) ws[wv] = incl;
        __syncthreads();
        uint32_t idx = run + incl - c, tot = 0;
        for (int k = 0; k < 256 / AM_WAVE; ++k) { if (k < wv) idx += ws[k]; tot += ws[k]; }
        run += tot;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t word = x[k];
            const uint32_t p0 = (w0 + (uint32_t)k) * wbits - lag;
            while (word) {
                const int b = __ffs((int)word) - 1;
                if (idx < Mcap) pos[idx] = p0 + (uint32_t)b;
                ++idx;
                word &= word - 1u;
            }
        }
    }
    if (g == nwg - 1u && threadIdx.x == 0) *total_out = run;
    if constexpr (ROWS != 0) {
        __shared__ unsigned long long NZ[AM_ROWS_NFL];
        __shared__ unsigned long long FL[AM_ROWS_NFL];
        __shared__ uint32_t PFX[AM_ROWS_NFL + 1];
        __shared__ __attribute__((aligned(16))) float XR[(256 / AM_WAVE) * AM_ROWS_SLOTS * AM_ROWS_XS];
        __shared__ uint16_t RC[(256 / AM_WAVE) * AM_ROWS_SLOTS];
        // (one pass at am_k_fe3's 14 x 96 words per workgroup: its words were loaded at the start of the kernel; more only in the
        // test builds that run 64 Msps through am_k_fe4<32, 1, 3>)
        am_rows_segment32<ROWS == 1>(ra, bits, w_begin, (w_begin + AM_ROWS_MAXW < w_end) ? w_begin + AM_ROWS_MAXW : w_end, NZ, FL, PFX,
                                     XR, RC, early, true);
        for (uint32_t wa = w_begin + AM_ROWS_MAXW; wa < w_end; wa += AM_ROWS_MAXW) {
            __syncthreads();                                          // (the flags of the pass before are still being read)
            am_rows_words none;
#pragma unroll
            for (int q = 0; q < AM_ROWS_NCH; ++q) none.w[q] = 0u;
            am_rows_segment32<ROWS == 1>(ra, bits, wa, (wa + AM_ROWS_MAXW < w_end) ? wa + AM_ROWS_MAXW : w_end, NZ, FL, PFX, XR, RC,
                                         none, false);
        }
    }
}

hipError_t am_launch_gather_wg(const uint32_t *bits, const uint32_t *wg_cnt, uint32_t nwg, uint32_t words_per_wg,
                               uint32_t nwords, uint32_t Mcap, uint32_t lag, uint32_t wbits, uint32_t *pos,
                               uint32_t *total_out, hipStream_t s, const am_rows_args *rows)
{
    if (nwg == 0) return hipSuccess;
    if (wbits == 0 || wbits > 32 || words_per_wg == 0) return hipErrorInvalidValue;
    am_rows_args ra;
    memset(&ra, 0, sizeof(ra));
    if (rows && rows->iq) {
#if defined(AM_TEST_KNOBS)
        // TEST BUILDS ONLY since round 6 (AIRMODES_FUSED_REFINE=0): round 5's arrangement -- this kernel forms the bb rows around the
        // candidates from the samples and writes them, am_k_refine_late reads them back.  The product library runs am_k_refine_seg
        // (am_refine_seg.hip: the rows never leave LDS) and does not instantiate these two.
        // (the rows are formed for am_k_fe3's bitmap: a word = one 32-sample chip, lag 288)
        if (wbits != 32 || lag != 288 || !rows->bb_sparse) return hipErrorInvalidValue;
        ra = *rows;
        if (ra.use_pmf)
            hipLaunchKernelGGL(am_k_gather_wg<1>, dim3(nwg), dim3(256), 0, s, bits, wg_cnt, nwg, words_per_wg, nwords, Mcap, lag,
                               wbits, pos, total_out, ra);
        else
            hipLaunchKernelGGL(am_k_gather_wg<2>, dim3(nwg), dim3(256), 0, s, bits, wg_cnt, nwg, words_per_wg, nwords, Mcap, lag,
                               wbits, pos, total_out, ra);
        return hipGetLastError();
#else
        return hipErrorInvalidValue;                          // (the product path forms the rows in am_k_refine_seg)
#endif
    }
    hipLaunchKernelGGL(am_k_gather_wg<0>, dim3(nwg), dim3(256), 0, s, bits, wg_cnt, nwg, words_per_wg, nwords, Mcap, lag, wbits, pos,
                       total_out, ra);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Exclusive prefix of one value per workgroup inside a single launch (a chained scan in its plainest form).
// A workgroup first DRAWS ITS PLACE in the chain (am_chain_place: an atomic counter tagged with the launch's epoch), then
// publishes (epoch, value) in slots[place] and adds up the values of the places before it, waiting for the ones that have
// not published yet.  Whoever holds a lower place has drawn it, i.e. is running: the wait cannot deadlock, whatever
// order the hardware dispatches workgroups in and wherever it puts them (MI355X_MICROARCH.md: dispatch order and placement
// are undefined).  Round 3 took blockIdx.x as the place below 512 workgroups -- an argument about dispatch order, and a
// __builtin_trap() behind it; the ticket costs one same-address atomic per workgroup (~11 ns each, ~200 workgroups).
// `epoch` differs from launch to launch (the host counts), so neither slots nor the counter are ever reset.
// The word carries its own payload, so the atomics are relaxed (device scope): release / acquire would write back
// and invalidate the whole L2 of the XCD at every step (measured: the marking kernel 18 -> 35 us).
// Should a place never be published after all (a workgroup that died), the wait gives up after AM_CHAIN_SPIN_MAX polls
// (~2 s): am_chain_prefix returns AM_CHAIN_FAIL to every thread, the caller skips its stores and raises the context's
// error word, and the host returns AM_EHIP from the call -- an error at the C ABI, not a hang and not a trap (a trap
// takes the whole process down, every other context with it).
// All threads of the workgroup call these (they synchronise); red / tick = LDS scratch.
// What the chain replaces: a one-workgroup scan launch between producer and consumer, 4.7 us each.
// ------------------------------------------------------------------------------------------
#ifndef AM_CHAIN_SPIN_MAX
#define AM_CHAIN_SPIN_MAX (1u << 25)
#endif
#define AM_CHAIN_FAIL 0xFFFFFFFFu
__device__ __forceinline__ uint32_t am_chain_prefix(unsigned long long *slots, uint32_t b, uint32_t epoch, uint32_t mine,
                                                    uint32_t *red)
{
    if (threadIdx.x == 0)
        __hip_atomic_store(&slots[b], ((unsigned long long)epoch << 32) | (unsigned long long)mine, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    uint32_t acc = 0;
    bool failed = false;
    for (uint32_t k = threadIdx.x; k < b && !failed; k += blockDim.x) {
        unsigned long long v = __hip_atomic_load(&slots[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (unsigned spins = 0; (uint32_t)(v >> 32) != epoch; ++spins) {
            if (spins == AM_CHAIN_SPIN_MAX) { failed = true; break; }
            __builtin_amdgcn_s_sleep(1);
            v = __hip_atomic_load(&slots[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        acc += (uint32_t)v;
    }
    for (int o = AM_WAVE / 2; o >= 1; o >>= 1) acc += (uint32_t)__shfl_xor((int)acc, o, AM_WAVE);
    const bool wave_failed = __ballot(failed) != 0ull;
    const int nw = (int)(blockDim.x / AM_WAVE);
    __syncthreads();                                          // (red may still be read from an earlier use)
    if ((threadIdx.x & (AM_WAVE - 1)) == 0) red[threadIdx.x / AM_WAVE] = wave_failed ? AM_CHAIN_FAIL : acc;
    __syncthreads();
    uint32_t tot = 0;
    bool any_failed = false;
    for (int k = 0; k < nw; ++k) { any_failed = any_failed || red[k] == AM_CHAIN_FAIL; tot += red[k]; }
    return any_failed ? AM_CHAIN_FAIL : tot;                  // (a sum of candidate counts never reaches 2^32 - 1)
}

// The workgroup's place in the chain: a ticket drawn at its start, so that everything it will wait for is already running.
// The counter only ever counts up: EVERY workgroup of every launch draws exactly one ticket, so the host knows the value the
// counter has when a launch starts (`base`: the grids of the launches before it, modulo 2^32) and a place is ticket - base.
// One plain fetch-add per workgroup.  (A compare-and-swap loop that re-tagged the counter with the launch's epoch -- no
// host bookkeeping -- made ~250 workgroups retry against each other: the marking kernel 16 -> 260 us, measured.)
// All threads call it (it synchronises); tick = LDS scratch word.
__device__ __forceinline__ uint32_t am_chain_place(uint32_t *ticket, uint32_t base, uint32_t *tick)
{
    if (threadIdx.x == 0) *tick = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base;
    __syncthreads();
    return *tick;
}

__device__ __forceinline__ uint32_t am_lower_bound(const uint32_t *__restrict__ pos, uint32_t lo,
                                                   uint32_t hi, uint32_t key)
{
    // first index in [lo, hi) with pos[idx] >= key (hi if none); gallop then bisect
    uint32_t stepw = 1, l = lo, h = lo;
    while (h < hi && pos[h] < key) { l = h + 1; h = (h + stepw < hi) ? h + stepw : hi; stepw <<= 1; }
    if (h > hi) h = hi;
    while (l < h) {
        const uint32_t mid = l + ((h - l) >> 1);
        if (pos[mid] < key) l = mid + 1; else h = mid;
    }
    return l;
}

#if AM_WITH_TILE_KERNEL
#include "am_split_refine.inc"      // (test builds only: the split refinement behind the tile kernel)
#endif
// The refinement behind the streaming front ends in ONE launch (round 4): late-peak decisions and the per-candidate test
// of am_k_energy (late mode) + am_k_cand, per group of AM_RCB consecutive candidates.  The decisions never leave LDS, and
// with them went the global compact layout: no distance-to-the-predecessor array, no scan of it (two launches fewer, and
// am_k_gather_bits no longer writes 4 bytes per candidate).
//   * candidate i of the group owns the positions [lo_i, pos_i + spc), lo_i = max(pos_i, pos_{i-1} + spc) -- what its
//     predecessors in the group do not cover (the group's first candidate: all spc of them; a neighbouring group may
//     decide up to spc - 1 positions a second time, identically); an LDS scan of the counts lays them out contiguously,
//     position pos_i + s at index coff[i] - (lo_i - pos_i) + s;
//   * one lane per position: late[k] = (E(q+1) > E(q)).  The late-peak search only needs the SIGN of E(q+1) - E(q), and
//     the two sums share all but eight samples: D = sum over the four pulses of bb[q + c spc + spc] - bb[q + c spc].  Both
//     reference sums have non-negative terms not above V, so each carries a rounding error below (4 spc - 1) 2^-53 * 4 spc V
//     <= 2^-39 V for spc <= 32; D is formed here in double precision with an error below 2^-47 V.  Hence whenever
//     |D| > 2^-36 V the sign of D IS the reference's comparison; only closer calls (exact ties of quantised or constant
//     input; non-finite samples: V = +inf) repeat the reference's two sequential sums (preamble_impl.cc:91-98).  V comes from
//     the front end: the largest bb of the workgroup segments the samples lie in (vmax[array coordinate / vspan]);
//   * one lane per candidate: the late-peak search over those bytes (preamble_impl.cc:184-192), the quiet zones
//     (:198-209), the record, and the greedy chain's successor.
#define AM_RCB 256                  /* candidates per workgroup */
#define AM_RPL 2                    /* positions per lane and round */
// any sample of the two 32-sample chips at ra_ / rb_ (16-byte aligned rows) above thr, among the samples i >= o (from_o) or i <= o.
// Half a row of each at a time (eight 16-byte loads in flight: 32 registers -- whole rows cost the kernel its eighth wave per SIMD),
// and only the halves that hold wanted samples
__device__ __forceinline__ bool am_rows_partly_above(const float *__restrict__ ra_, const float *__restrict__ rb_, int o, bool from_o,
                                                     float thr)
{
    bool hit = false;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        // samples 16 h .. 16 h + 15: wanted ones among them?  (from_o: i >= o -> the upper half always, the lower one if o < 16;
        // else i <= o -> the lower half always, the upper one if o >= 16)
        const bool some = from_o ? (h == 1 || o < 16) : (h == 0 || o >= 16);
        if (some && !hit) {
            float4 ta[4], tb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) ta[k] = reinterpret_cast<const float4 *>(ra_)[4 * h + k];
#pragma unroll
            for (int k = 0; k < 4; ++k) tb[k] = reinterpret_cast<const float4 *>(rb_)[4 * h + k];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a4[4] = {ta[k].x, ta[k].y, ta[k].z, ta[k].w}, b4[4] = {tb[k].x, tb[k].y, tb[k].z, tb[k].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = 16 * h + 4 * k + q;
                    const bool in = from_o ? (i >= o) : (i <= o);
                    hit = hit || (in && (a4[q] > thr || b4[q] > thr));
                }
            }
        }
    }
    return hit;
}

__global__ void __launch_bounds__(256, 8)                    // (64 VGPRs: all of a bench scan's ~1 950 workgroups resident at once)
am_k_refine_late(const float *__restrict__ bb, const float *__restrict__ avg_sparse, const uint32_t *__restrict__ pos,
                 uint32_t Mcap, int spc, float thr_lin, uint32_t end_j, uint32_t *__restrict__ eo,
                 uint32_t *__restrict__ tgt, float *__restrict__ inavg, uint8_t *__restrict__ valid,
                 uint32_t *__restrict__ jump0, const uint32_t *__restrict__ Mp, const float *__restrict__ vmax,
                 uint32_t vspan, uint32_t nv, const float *__restrict__ bb_max)
{
    const uint32_t M = am_count(Mcap, Mp);
    __shared__ uint32_t coff[AM_RCB + 1];   // compact index of the first position candidate i owns (+ end)
    __shared__ uint32_t clo[AM_RCB];        // that position
    __shared__ uint32_t ws[AM_RCB / AM_WAVE];
    __shared__ uint8_t LATE[AM_RCB * 32];   // (spc <= 32: the launcher checks)
    const uint32_t c0 = blockIdx.x * AM_RCB;
    if (c0 >= M) return;                                      // (uniform)
    const uint32_t nc = (M - c0 < AM_RCB) ? M - c0 : AM_RCB;
    const uint32_t i = threadIdx.x;
    const int lane = (int)(i & (AM_WAVE - 1)), wv = (int)(i / AM_WAVE);
    const bool live = i < nc;
    const uint32_t g = c0 + (live ? i : 0u);
    const uint32_t j = pos[g];
    uint32_t pvw = pos[g ? g - 1u : 0u];                      // (unconditional, in flight with pos[g]: behind `if (live && i)` it was a round trip of its own)
    AM_PIN_U32(pvw);
    uint32_t lo = j, d = 0;
    if (live) {
        if (i) { const uint32_t pv = pvw + (uint32_t)spc; lo = pv > j ? pv : j; }
        d = j + (uint32_t)spc - lo;                           // >= 1: positions ascend strictly
    }
    {
        uint32_t incl = d;
        for (int o = 1; o < AM_WAVE; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, o, AM_WAVE);
            if (lane >= o) incl += up;
        }
        if (lane == AM_WAVE - 1) ws[wv] = incl;
        __syncthreads();
        uint32_t off = incl - d;
        for (int k = 0; k < wv; ++k) off += ws[k];
        if (live) { coff[i] = off; clo[i] = lo; }
        if (i == nc - 1u) coff[nc] = off + d;
    }
    __syncthreads();
    const uint32_t kend = coff[nc];
    // V bounds every sample the group's positions can see: the largest bb of the front-end workgroups whose segments they
    // span (usually one; a larger V only sends more close calls to the exact sums)
    float vb = 0.0f;
    {
        uint32_t v0 = clo[0] / vspan, v1 = (clo[nc - 1u] + 13u * (uint32_t)spc + 1u) / vspan;
        v0 = v0 < nv ? v0 : nv - 1u;
        v1 = v1 < nv ? v1 : nv - 1u;
        for (uint32_t v = v0; v <= v1; ++v) vb = fmaxf(vb, vmax[v]);      // (every lane the same words)
    }
    const double bound = (double)vb * 0x1p-36;                 // (+inf when a sample is not finite: nothing is decided by D)
    // AM_RPL positions per lane and round, their loads in flight together (a group has ~950 positions; the first version staged
    // each 256-position pass in LDS behind barriers: a dependent round trip per pass.  Four per lane cost the registers of a
    // sixth resident workgroup per CU: 34 -> 46 us at the bench density, measured; two do not)
    for (uint32_t k0 = 0; k0 < kend; k0 += AM_RPL * blockDim.x) {
        uint32_t q[AM_RPL];
        bool has[AM_RPL];
#pragma unroll
        for (int u = 0; u < AM_RPL; ++u) {
            const uint32_t k = k0 + (uint32_t)u * blockDim.x + threadIdx.x;
            has[u] = k < kend;
            uint32_t l = 0, h = nc;                            // last candidate with coff <= k
            const uint32_t kk = has[u] ? k : kend - 1u;
            while (h - l > 1) {
                const uint32_t mid = (l + h) >> 1;
                if (coff[mid] <= kk) l = mid; else h = mid;
            }
            q[u] = clo[l] + (kk - coff[l]);
        }
        float x[AM_RPL][8];
#pragma unroll
        for (int u = 0; u < AM_RPL; ++u) {
            const float *p = bb + q[u];                        // (a lane without a position reads the group's last one again)
            x[u][0] = p[0]; x[u][1] = p[spc]; x[u][2] = p[2 * spc]; x[u][3] = p[3 * spc];
            x[u][4] = p[7 * spc]; x[u][5] = p[8 * spc]; x[u][6] = p[9 * spc]; x[u][7] = p[10 * spc];
        }
        bool close = false;
#pragma unroll
        for (int u = 0; u < AM_RPL; ++u) {
            double dd = (double)x[u][1] - (double)x[u][0];
            dd = dd + ((double)x[u][3] - (double)x[u][2]);
            dd = dd + ((double)x[u][5] - (double)x[u][4]);
            dd = dd + ((double)x[u][7] - (double)x[u][6]);
            const bool far = fabs(dd) > bound;
            close = close || (has[u] && !far);
            if (has[u] && far) LATE[k0 + (uint32_t)u * blockDim.x + threadIdx.x] = dd > 0.0 ? 1 : 0;
        }
        if (close) {
            // (rare: exact ties of quantised or constant input, non-finite samples) the reference's two sequential sums
#pragma unroll 1
            for (int u = 0; u < AM_RPL; ++u) {
                if (!has[u]) continue;
                double dd = (double)x[u][1] - (double)x[u][0];
                dd = dd + ((double)x[u][3] - (double)x[u][2]);
                dd = dd + ((double)x[u][5] - (double)x[u][4]);
                dd = dd + ((double)x[u][7] - (double)x[u][6]);
                if (!(fabs(dd) > bound))
                    LATE[k0 + (uint32_t)u * blockDim.x + threadIdx.x] =
                        am_preamble_energy(bb + q[u] + 1u, spc) > am_preamble_energy(bb + q[u], spc) ? 1 : 0;
            }
        }
    }
    __syncthreads();
    // ---- one lane per candidate -------------------------------------------------------------------------------
    int how_late = 0;
    if (live) {
        const uint8_t *Lb = LATE + (coff[i] - (lo - j));       // decision of position j (see above: never below LATE[0])
        bool rising = true;
        for (int k = 0; k < spc && rising; ++k) {
            if (Lb[k]) how_late++; else rising = false;
        }
    }
    const uint32_t e = j + (uint32_t)how_late;
    // quiet zones (preamble_impl.cc:198-209)
    const float p0 = bb[e], p1 = bb[e + 2 * spc], p2 = bb[e + 7 * spc], p3 = bb[e + 9 * spc];
    const float av = (e >= end_j) ? 0.0f : avg_sparse[e];    // beyond the end of the stream: 0
    // (the maxima of the quiet zones' whole chips ride along with the four pulses: they depend on e alone)
    float mz[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (bb_max) {
        const uint32_t c = e >> 5;
        const uint32_t cz[6] = {c + 4u, c + 5u, c + 11u, c + 12u, c + 13u, c + 14u};
#pragma unroll
        for (int k = 0; k < 6; ++k) mz[k] = ((unsigned long long)cz[k] * 32ull < (unsigned long long)end_j) ? bb_max[cz[k]] : 0.0f;
    }
    float ps = p0 + p1;
    ps = ps + p2;
    ps = ps + p3;
    const float avgpeak = (float)((double)ps / 4.0);
    const float sthr = av + (avgpeak - av) / thr_lin;
    bool ok;
    if (bb_max) {
        // (uniform) 32 samples per chip, rows and their maxima from am_k_gather_wg<1> (round 5).  With e at offset o of chip c the
        // zones [e + 96, e + 192] and [e + 320, e + 480] are: chip c + 3 from o on, chips c + 4, c + 5 whole, chip c + 6 up to o;
        // chip c + 10 from o on, chips c + 11 .. c + 14 whole, chip c + 15 up to o.  A whole chip holds a sample above the limit
        // exactly when its maximum is above it (fmaxf skips a NaN as `NaN > x` is false): six numbers in one round trip decide
        // most candidates; the four partial chips follow as whole aligned rows, two at a time, masked by the sample index.
        // Chips beyond the data read as zeros, like the array's padding.
        const uint32_t c = e >> 5;
        const int o = (int)(e & 31u);
        bool hit = false;
#pragma unroll
        for (int k = 0; k < 6; ++k) hit = hit || (mz[k] > sthr);
        if (!hit) hit = am_rows_partly_above(bb + (size_t)(c + 3u) * 32u, bb + (size_t)(c + 10u) * 32u, o, true, sthr);
        if (!hit) hit = am_rows_partly_above(bb + (size_t)(c + 6u) * 32u, bb + (size_t)(c + 15u) * 32u, o, false, sthr);
        ok = live && !hit;
    } else
        ok = live && !am_any_above2(bb + e + 3 * spc, 3 * spc + 1,                    // offsets 3spc .. 6spc
                                    bb + e + 10 * spc, 5 * spc + 1, sthr);            // offsets 10spc .. 15spc
    if (!live) return;
    eo[g] = e;
    inavg[g] = av;
    valid[g] = ok ? 1 : 0;
    const uint32_t tg = ok ? (e + (uint32_t)(AM_BURST * spc)) : (e + 1u);   // :237 / :209
    tgt[g] = tg;
    // the greedy chain's successor: first candidate at or after the resume position
    jump0[g] = am_lower_bound(pos, g + 1u, M, tg);
    if (g == M - 1u) jump0[M] = M;
}

hipError_t am_launch_refine_late(const float *bb, const float *avg_sparse, const uint32_t *pos, uint32_t M, int spc,
                                 float thr_lin, uint32_t end_j, uint32_t *e, uint32_t *tgt, float *inavg, uint8_t *valid,
                                 uint32_t *jump0, hipStream_t s, const uint32_t *Mp, const float *vmax, uint32_t vspan,
                                 uint32_t nv, const float *bb_max)
{
    if (M == 0) return hipSuccess;
    if (spc > 32 || spc < 1 || !vmax || vspan == 0 || nv == 0 || !jump0) return hipErrorInvalidValue;   // (the rounding bound is stated for 4 spc <= 128 terms)
    if (bb_max && (spc != 32 || (reinterpret_cast<uintptr_t>(bb) & 15u) != 0)) return hipErrorInvalidValue;   // (rows of 32 samples, 16-byte aligned)
    hipLaunchKernelGGL(am_k_refine_late, dim3((M + AM_RCB - 1) / AM_RCB), dim3(AM_RCB), 0, s, bb, avg_sparse, pos, M, spc, thr_lin,
                       end_j, e, tgt, inavg, valid, jump0, Mp, vmax, vspan, nv, bb_max);
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------
// Greedy chain.  The reference scan visits candidates in position order; after visiting c it
// resumes at tgt[c], so the next visited candidate is succ(c) = first candidate with
// pos >= tgt[c] (node M is the end sentinel), and the visited set is the orbit of the root (the
// first candidate at or after the position the scan starts from) under succ.  A successor is never
// far away (a jump spans at most 241*spc + 1 positions), so the candidates are cut into blocks of
// AM_CB nodes and
//   1. am_k_cblk_exit : per block, in LDS, pointer jumping on the block's successor array: for
//      EVERY node the first node of its orbit beyond the block (and, for the time-sharded path,
//      the last one inside it); the exits of the block's first AM_CB_HEADW nodes -- its "head",
//      where a jump from an earlier block lands -- also go to a dense table;
//   2. am_k_cblk_walk : one workgroup copies the head table to LDS, finds the root and walks block
//      to block (one LDS read per block; global memory only for the root and for an unusually
//      long head): the node at which the scan enters each block;
//   3. am_k_cblk_mark : per block, in LDS: log2(AM_CB) levels of 2^l-hop tables, top-down marking
//      from the block's entry node, visited[] out.
// Exact for any input, O(M) work, a handful of dependent global loads in all.  Step 1 does not
// depend on where the scan starts, so a time shard runs it before its entry position is known and
// derives its exit table (am_k_cblk_exit_table) from the same arrays.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
am_k_chain_succ(const uint32_t *__restrict__ pos, const uint32_t *__restrict__ tgt, uint32_t Mcap,
                uint32_t *__restrict__ jump0, const uint32_t *__restrict__ Mp)
{
    const uint32_t M = am_count(Mcap, Mp);
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < M) jump0[g] = am_lower_bound(pos, g + 1, M, tgt[g]);
    if (g == M) jump0[M] = M;
}

#define AM_CB 2048                  /* nodes per block */
#define AM_CB_THREADS 256
#define AM_CB_PER (AM_CB / AM_CB_THREADS)
#define AM_CB_LEVELS 11             /* 2^11 = AM_CB */
#ifndef AM_CB_HEADW
#define AM_CB_HEADW 256             /* head nodes per block the walk keeps in LDS (16-bit links) ... */
#endif
#define AM_CB_HEADCAP 65534         /* ... as long as all of them fit (slots are 16-bit, two values reserved; 128 KB of links + the group tables) */
#define AM_CB_NONE 0xFFFFFFFFu
#define AM_CB_END 0xFFFFu            /* link: the orbit leaves the candidate list */
#define AM_CB_OUT 0xFFFEu            /* link: it lands beyond the next block's head (resolved through exitnode[]) */
#ifndef AM_CB_GROUP
#define AM_CB_GROUP 16              /* blocks per group of the two-level walk */
#endif

__global__ void __launch_bounds__(AM_CB_THREADS)
am_k_cblk_exit(const uint32_t *__restrict__ jump0, uint32_t Mcap, uint32_t headw, uint32_t *__restrict__ exitnode,
               uint32_t *__restrict__ lastnode, uint16_t *__restrict__ headlink, const uint32_t *__restrict__ Mp)
{
    __shared__ uint32_t e[2][AM_CB];           // orbit node 2^r hops ahead, clipped to the first one outside
    __shared__ uint32_t l[2][AM_CB];           // the orbit node just before it (always inside the block)
    const uint32_t M = am_count(Mcap, Mp);
    const uint32_t base = blockIdx.x * AM_CB;
    if (base >= M) {                                          // (capacity launch: nothing here)
        for (uint32_t i = threadIdx.x; i < headw; i += blockDim.x) headlink[(size_t)blockIdx.x * headw + i] = (uint16_t)AM_CB_END;
        return;
    }
    const uint32_t end = (base + AM_CB < M) ? base + AM_CB : M;
    const uint32_t n = end - base;
    {
        // (unconditional loads from clamped indices: behind `i < n` every load sat in its own branch and waited for the one
        // before it -- eight serial memory round trips)
        uint32_t j[AM_CB_PER];
#pragma unroll
        for (int k = 0; k < AM_CB_PER; ++k) {
            const uint32_t i = threadIdx.x + k * AM_CB_THREADS;
            j[k] = jump0[base + (i < n ? i : n - 1u)];
        }
#pragma unroll
        for (int k = 0; k < AM_CB_PER; ++k) {
            const uint32_t i = threadIdx.x + k * AM_CB_THREADS;
            e[0][i] = (i < n) ? j[k] : end;
            l[0][i] = base + i;
        }
    }
    __syncthreads();
    int cur = 0;
    for (int r = 0; r < AM_CB_LEVELS; ++r) {                 // successors strictly increase: <= 2^11 hops inside
        // (the thread's eight nodes side by side: all first reads, then all dependent reads, then the stores -- written
        // as one read-read-store per node the compiler keeps them in program order, LDS stores may alias LDS loads.
        // Eight CONSECUTIVE nodes per thread -- own entries as 16-byte accesses, 16 LDS instructions per level instead of
        // 40 -- was measured in round 3: this kernel 9.0 -> 10-11 us (a lane stride of 64 bytes is a 4-way bank conflict),
        // the marking kernel 19.5 -> 16.5-17.2 us with the same layout: not the instruction count, dropped.)
        uint32_t t[AM_CB_PER], ne[AM_CB_PER], nl[AM_CB_PER];
#pragma unroll
        for (int k = 0; k < AM_CB_PER; ++k) t[k] = e[cur][threadIdx.x + k * AM_CB_THREADS];
#pragma unroll
        for (int k = 0; k < AM_CB_PER; ++k) {
            const uint32_t i = threadIdx.x + k * AM_CB_THREADS;
            const bool in = t[k] < end;
            const uint32_t j = in ? t[k] - base : i;
            ne[k] = in ? e[cur][j] : t[k];
            nl[k] = l[cur][j];                               // (not in: the node's own entry)
        }
#pragma unroll
        for (int k = 0; k < AM_CB_PER; ++k) {
            const uint32_t i = threadIdx.x + k * AM_CB_THREADS;
            e[cur ^ 1][i] = ne[k];
            l[cur ^ 1][i] = nl[k];
        }
        cur ^= 1;
        __syncthreads();
    }
    for (int k = 0; k < AM_CB_PER; ++k) {
        const uint32_t i = threadIdx.x + k * AM_CB_THREADS;
        if (i < n) {
            exitnode[base + i] = e[cur][i];
            if (lastnode) lastnode[base + i] = l[cur][i];
        }
        if (i < headw) {
            // the head's exits as links to the next table slot (slot = block * headw + index in block; headw = 2^k)
            const uint32_t g = (i < n) ? e[cur][i] : M, kb = g / AM_CB, ki = g % AM_CB;
            headlink[(size_t)blockIdx.x * headw + i] = (uint16_t)(g >= M ? AM_CB_END : (ki < headw ? kb * headw + ki : AM_CB_OUT));
        }
    }
}

// entry[b] = node at which the scan that starts at position cur0 enters block b, AM_CB_NONE if it
// jumps over the block.  scalars[0] = cur0 (the emit kernel raises it to the resume position),
// scalars[1] = 0.
// The walk itself is sequential (one hop per block), so its inner loop is made as short as it can
// be: the head table is turned into "next table slot" links (16 bit: slot = block * headw + index),
// and a hop is one dependent LDS read plus one LDS write that records the block's entry; node
// numbers, global memory and the entry array only appear outside that loop.

#if defined(__HIP_DEVICE_COMPILE__)
#define AM_PIN_U4(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w))
#else
#define AM_PIN_U4(v) ((void)0)
#endif
// head table (16-bit links, written by am_k_cblk_exit) -> LDS, 16 bytes per load, all of a thread's loads in
// flight at once (eight)
__device__ __forceinline__ void am_cblk_load_links(uint16_t *lnk, const uint16_t *__restrict__ headlink, uint32_t total)
{
    const uint32_t nvec = total / 8u;                        // (the table starts 16-byte aligned in the scratch buffer)
    const uint4 *src = reinterpret_cast<const uint4 *>(headlink);
    uint4 *dst = reinterpret_cast<uint4 *>(lnk);
    for (uint32_t f0 = threadIdx.x; f0 < nvec; f0 += 8u * blockDim.x) {
        uint4 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t f = f0 + (uint32_t)k * blockDim.x;
            t[k] = src[f < nvec ? f : nvec - 1u];
        }
        // (all eight loads go out before the first store waits for its own: without the pin the compiler sinks each
        // load to its store, eight memory round trips instead of one)
#pragma unroll
        for (int k = 0; k < 8; ++k) AM_PIN_U4(t[k]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t f = f0 + (uint32_t)k * blockDim.x;
            if (f < nvec) dst[f] = t[k];
        }
    }
    for (uint32_t f = nvec * 8u + threadIdx.x; f < total; f += blockDim.x) lnk[f] = headlink[f];
}

#if defined(AM_WALK_DEBUG)
// tuning builds only: what the walk does (printed by its lane 0, on the device or in the CPU emulation)
#include <stdio.h>
#define AM_WALK_COUNT(k) (wdbg[k]++)
#define AM_WALK_ITER(k) (wdbg[2 + (k)]++)
#define AM_WALK_CLOCK(k) (wclk[k] = (long long)__builtin_readcyclecounter())
#define AM_WALK_REPORT() printf("walk: nblk %u headw %u M %u root %u | global hops %d + %d, group jumps %d, plain hops %d, step-3 hops (lane 0) %d | cycles: load+root %lld, step 1 %lld, step 2 %lld, step 3 %lld\n", nblk, headw, M, root_s, wdbg[0], wdbg[1], wdbg[2], wdbg[3], wdbg[4], wclk[1] - wclk[0], wclk[2] - wclk[1], wclk[3] - wclk[2], wclk[4] - wclk[3])
#else
#define AM_WALK_COUNT(k) ((void)0)
#define AM_WALK_ITER(k) ((void)0)
#define AM_WALK_CLOCK(k) ((void)0)
#define AM_WALK_REPORT() ((void)0)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define AM_KEEP_VGPR(x) asm volatile("" : "+v"(x))
#else
#define AM_KEEP_VGPR(x) ((void)0)
#endif

// Time shards, device-side composition of the exit tables (am_shard_entry2 on the host does the same): msgs = `world`
// messages of AM_SHARD_MSG_HEADER + cap entries each: entry 0 = {count, overflow}, entry 1 = {where the scan left that
// rank's chunk in the step BEFORE, -}, then the table.  The scan enters chunk 0 where it left the last chunk of the step
// before (0 at the start of a stream); chunk r is entered at `cur`; the first candidate of chunk r at or after cur says
// where the scan leaves the chunk.  Returns the absolute position at which the scan enters chunk `rank`, *exit_out = where
// it leaves it; *bad = 1 if some table did not fit its message or some scan met more candidates than its capacity (the
// caller then repeats the step on the synchronous path: every rank alike).
// ONE WAVE calls it (all 64 lanes).  The chain over the chunks is sequential (where chunk r is entered depends on where chunk
// r - 1 was left), but inside a table the first entry at or after `cur` is found 64 entries per memory round trip (positions
// ascend: a ballot) -- one thread stepping through a table entry by entry paid a dependent load per entry, and chunk 7 of 8
// waits for seven tables.
__device__ __forceinline__ uint64_t am_shard_entry_wave(const am_shard_exit *__restrict__ msgs, uint32_t world, uint32_t rank,
                                                        uint32_t cap, int lane, uint32_t *bad_out, uint64_t *exit_out,
                                                        const uint64_t *__restrict__ cur_in = nullptr, uint64_t *carry_out = nullptr)
{
    const size_t stride = (size_t)cap + AM_SHARD_MSG_HEADER;
    // where the scan left the last chunk of the step before: the last rank's header -- or (chunks of ONE stream in flight on one GPU:
    // am_spipe; steps of the time-sharded receiver in flight: am_shard_resolve_submit) a device word, read NOW: the scan of this
    // chunk was enqueued before the word was written
    uint64_t cur = cur_in ? *cur_in : msgs[(size_t)(world - 1u) * stride + 1u].pos;             // (every lane reads the same word)
    uint64_t entry = cur, leave = cur;
    uint32_t bad = 0;
    // carry_out: the composition runs through ALL chunks of the step, not only up to the own one: where the scan leaves the step's
    // last chunk is what the next step starts from -- every rank works it out for itself, from the same tables (round 6: the carry no
    // longer travels in the next step's message, which is written before this step is resolved when steps are in flight)
    const uint32_t r_end = carry_out ? world : rank + 1u;
    for (uint32_t r = 0; r < r_end; ++r) {
        const am_shard_exit *m = msgs + (size_t)r * stride;
        const uint64_t n = m[0].pos;
        if (n > cap || m[0].exit != 0) { bad = 1; break; }
        if (r == rank) entry = cur;
        const am_shard_exit *t = m + AM_SHARD_MSG_HEADER;
        for (uint64_t i0 = 0; i0 < n; i0 += AM_WAVE) {                       // (uniform)
            const uint64_t i = i0 + (uint64_t)lane;
            const bool here = i < n && t[i].pos >= cur;
            const unsigned long long hit = __ballot(here);
            if (hit) {
                const uint64_t k = i0 + (uint64_t)(__ffsll((long long)hit) - 1);
                const uint64_t ex = t[k].exit;                                // (every lane reads the same entry)
                cur = ex > cur ? ex : cur;
                break;
            }
        }                                                                     // (no candidate left: the scan passes through)
        if (r == rank) leave = cur;
    }
    for (uint32_t r = r_end + (uint32_t)lane; r < world && !bad; r += AM_WAVE)   // (every rank must take the same decision)
        if (msgs[(size_t)r * stride].pos > cap || msgs[(size_t)r * stride].exit != 0) bad = 1;
    *bad_out = __ballot(bad != 0u) != 0ull ? 1u : 0u;
    *exit_out = leave;
    if (carry_out && lane == 0 && !*bad_out) *carry_out = cur;
    return entry;
}

// entry[b] = node at which the scan that starts at position cur0 enters block b, AM_CB_NONE if it
// jumps over the block.  scalars[0] = cur0 (the emit kernel raises it to the resume position),
// scalars[1] = 0.
// The walk is one hop per block (a dependent LDS read each), so it is done on two levels: the blocks are cut into
// groups of AM_CB_GROUP; (1) in parallel, from EVERY head slot of every group's first block, the slot at which the
// orbit leaves the group (<= AM_CB_GROUP hops each); (2) one lane goes from group to group through that table;
// (3) in parallel, one lane per group repeats the hops inside its group from the slot (2) found and records the
// entry node of each block.  Global memory only for the root, for a link that lands beyond a head, and for an
// unusually long head (ki >= headw): those go hop by hop in step (2).
__global__ void __launch_bounds__(1024)
am_k_cblk_walk(const uint32_t *__restrict__ pos, const uint32_t *__restrict__ exitnode,
               const uint16_t *__restrict__ headlink, uint32_t Mcap, uint32_t nblk, uint32_t headw, uint32_t cur0_host,
               uint32_t *__restrict__ entry, uint32_t *__restrict__ scalars, const uint32_t *__restrict__ Mp,
               am_entry_src es)
{
    const uint32_t M = am_count(Mcap, Mp);
    // time shards: the position at which the scan enters this chunk is composed here, from everybody's exit tables, by the
    // first wave -- while the other waves fetch the link table (a launch of its own cost 4.4 us + the gap behind it)
    __shared__ uint32_t cur0_s;
    if (es.msgs) {
        if (threadIdx.x < AM_WAVE) {
            uint32_t bad = 0;
            uint64_t leave = 0;
            const uint64_t cur = am_shard_entry_wave(es.msgs, es.world, es.rank, es.cap, (int)threadIdx.x, &bad, &leave, es.cur_in, es.carry_out);
            if (threadIdx.x == 0) {
                uint64_t rel = cur > es.base_abs ? cur - es.base_abs : 0;
                if (rel > 0xFFFFFFF0ull) rel = 0xFFFFFFF0ull;
                cur0_s = (uint32_t)rel;
                es.flags[0] = bad;                           // (written either way: nobody has to clear it first)
                if (!bad) *es.exit_out = leave;              // where the scan leaves this chunk: the next step's message carries it
            }
        }
    } else if (threadIdx.x == 0)
        cur0_s = cur0_host;
#if defined(AM_WALK_DEBUG)
    int wdbg[5] = {0, 0, 0, 0, 0};
    long long wclk[5] = {0, 0, 0, 0, 0};
    AM_WALK_CLOCK(0);
#endif
    HIP_DYNAMIC_SHARED(uint16_t, lnk);         // [nblk * headw (+pad to 8)] links | [nblk] entry index | [ngrp * headw] group exits | [ngrp] group entry slots
    const uint32_t total = nblk * headw;
    const uint32_t ngrp = (nblk + AM_CB_GROUP - 1u) / AM_CB_GROUP;
    uint16_t *ent = lnk + ((total + 7u) & ~7u);
    uint16_t *gex = ent + nblk;
    uint16_t *gin = gex + ngrp * headw;
    __shared__ uint32_t seg, root_s;
    // root = first candidate with pos >= cur0, in two parallel rounds (two dependent loads in all): which of 1024
    // equal segments holds it, then which node of that segment.  The first round's loads go out together with the
    // table's.
    const uint32_t stride = (M + blockDim.x - 1u) / blockDim.x;
    const uint32_t lo = threadIdx.x * stride;
    const uint32_t hi = (lo + stride < M) ? lo + stride : M;
    uint32_t p_hi = 0, p_lo = 0;
    if (stride && lo < M) { p_hi = pos[hi - 1u]; p_lo = lo ? pos[lo - 1u] : 0u; }
    if (threadIdx.x == 0) { seg = M; root_s = M; }
    am_cblk_load_links(lnk, headlink, total);
    for (uint32_t bb = threadIdx.x; bb < nblk; bb += blockDim.x) ent[bb] = (uint16_t)AM_CB_END;
    for (uint32_t gg = threadIdx.x; gg < ngrp; gg += blockDim.x) gin[gg] = (uint16_t)AM_CB_END;
    __syncthreads();
    const uint32_t cur0 = cur0_s;
    if (threadIdx.x == 0) { scalars[0] = cur0; scalars[1] = 0u; }
    if (stride && lo < M && p_hi >= cur0 && (lo == 0 || p_lo < cur0)) seg = lo;
    __syncthreads();
    for (uint32_t g = seg + threadIdx.x; g < M && g < seg + stride; g += blockDim.x)
        if (pos[g] >= cur0 && (g == 0 || pos[g - 1u] < cur0)) root_s = g;
    const uint32_t hs = headw ? (uint32_t)(31 - __clz((int)headw)) : 0u, hm = headw - 1u;   // headw = 2^hs
    AM_WALK_CLOCK(1);
    // (1) group exits: the slot reached from head slot h of group gi's first block once the orbit is past the group,
    // or the slot inside the group whose link is not a slot (END / OUT)
    // (one walker at a time per thread: four side by side were tried, 50 % slower)
    for (uint32_t idx = threadIdx.x; idx < ngrp * headw; idx += blockDim.x) {
        const uint32_t gi = idx >> hs, h = idx & hm;
        const uint32_t limit = (gi + 1u) * AM_CB_GROUP;      // first block of the next group
        uint32_t slot = ((gi * AM_CB_GROUP) << hs) + h;
        for (int hop = 0; hop < AM_CB_GROUP; ++hop) {        // (a link leads to a later block: <= AM_CB_GROUP hops)
            const uint32_t nx = lnk[slot];
            if (nx >= AM_CB_OUT) break;
            slot = nx;
            if ((slot >> hs) >= limit) break;
        }
        gex[idx] = (uint16_t)slot;
    }
    __syncthreads();
    AM_WALK_CLOCK(2);
    // (2) one lane, group to group
    if (threadIdx.x == 0) {
        uint32_t g = root_s;                                 // node the orbit is at
        while (g < M) {
            const uint32_t kb = g / AM_CB, ki = g % AM_CB;
            if (ki >= headw) {                               // root / long head: the node itself is the entry; one global hop
                ent[kb] = (uint16_t)ki;
                g = exitnode[g];
                AM_WALK_COUNT(0);
                continue;
            }
            uint32_t slot = (kb << hs) + ki;
            bool done = false;
            for (;;) {
                const uint32_t b = slot >> hs;
                uint32_t nx;
                if (b % AM_CB_GROUP == 0u) {                 // a group's first block: through the group in one step
                    const uint32_t gi = b / AM_CB_GROUP;
                    gin[gi] = (uint16_t)slot;                // (step 3 records the entries from here)
                    const uint32_t r = gex[(gi << hs) + (slot & hm)];
                    AM_WALK_ITER(0);
                    if ((r >> hs) >= (gi + 1u) * AM_CB_GROUP) { slot = r; continue; }
                    slot = r;                                // stuck inside the group: its link is END or OUT
                    nx = lnk[slot];
                } else {                                     // (after a global hop: plain hops to the next group)
                    ent[b] = (uint16_t)(slot & hm);
                    nx = lnk[slot];
                    AM_WALK_ITER(1);
                    if (nx < AM_CB_OUT) { slot = nx; continue; }
                }
                if (nx == AM_CB_END) { done = true; break; }
                // lands beyond a head: one hop through global memory
                g = exitnode[(slot >> hs) * AM_CB + (slot & hm)];
                AM_WALK_COUNT(1);
                break;
            }
            if (done) break;
        }
    }
    __syncthreads();
    AM_WALK_CLOCK(3);
    // (3) the entries inside every group the orbit entered at its first block
    for (uint32_t gi = threadIdx.x; gi < ngrp; gi += blockDim.x) {
        uint32_t slot = gin[gi];
        if (slot == AM_CB_END) continue;
        const uint32_t limit = (gi + 1u) * AM_CB_GROUP;
        for (int hop = 0; hop <= AM_CB_GROUP; ++hop) {
            if ((slot >> hs) >= limit) break;
            ent[slot >> hs] = (uint16_t)(slot & hm);
            AM_WALK_ITER(2);
            const uint32_t nx = lnk[slot];
            if (nx >= AM_CB_OUT) break;
            slot = nx;
        }
    }
    __syncthreads();
    AM_WALK_CLOCK(4);
    if (threadIdx.x == 0) AM_WALK_REPORT();
    for (uint32_t bb = threadIdx.x; bb < nblk; bb += blockDim.x) {
        const uint32_t ki = ent[bb];
        entry[bb] = (ki == AM_CB_END) ? AM_CB_NONE : bb * AM_CB + ki;
    }
}

// Exit table of a time chunk (am_shard_scan): for each of the chunk's first n candidates, the scan
// position after the chunk's LAST visited candidate if the scan enters at that candidate.
__global__ void __launch_bounds__(1024)
am_k_cblk_exit_table(const uint32_t *__restrict__ pos, const uint32_t *__restrict__ tgt,
                     const uint32_t *__restrict__ exitnode, const uint32_t *__restrict__ lastnode,
                     const uint16_t *__restrict__ headlink, uint32_t Mcap, uint32_t nblk, uint32_t headw, uint32_t n,
                     uint32_t lead_end, uint64_t base_abs, am_shard_exit *__restrict__ table,
                     const uint32_t *__restrict__ Mp, am_shard_exit *__restrict__ header, const uint64_t *__restrict__ carry)
{
    const uint32_t M = am_count(Mcap, Mp);
    // header (device-side exchange of the tables): pos = number of entries = up to and including the first candidate at or
    // past lead_end, all min(n, M) of them if there is none.  Positions ascend: exactly one thread qualifies.
    // header->exit = 1: the scan met more candidates than the capacity it was launched for (its table is built on an
    // incomplete list: every rank must repeat the step, and every rank learns it from here)
    if (header) {
        const uint32_t nr = M < n ? M : n;
        const uint64_t over = (Mp && *Mp > Mcap) ? 1u : 0u;
        if (threadIdx.x == 0 && carry) { header[1].pos = *carry; header[1].exit = 0; }
        if (nr == 0) {
            if (threadIdx.x == 0) { header->pos = 0; header->exit = over; }
        } else
            for (uint32_t i = threadIdx.x; i < nr; i += blockDim.x) {
                const bool term = pos[i] >= lead_end && (i == 0 || pos[i - 1u] < lead_end);
                const bool tail = i == nr - 1u && pos[i] < lead_end;
                // (tail with candidates left beyond the table's n entries: the table does not fit -- count n + 1 says so)
                if (term || tail) { header->pos = term ? i + 1u : (M > n ? n + 1u : nr); header->exit = over; }
            }
    }
    HIP_DYNAMIC_SHARED(uint16_t, lnk);         // [nblk * headw (+pad to 8)] links | [ngrp * headw] group exits
    const uint32_t total = nblk * headw;
    const uint32_t ngrp = (nblk + AM_CB_GROUP - 1u) / AM_CB_GROUP;
    uint16_t *gex = lnk + ((total + 7u) & ~7u);
    am_cblk_load_links(lnk, headlink, total);
    __syncthreads();
    const uint32_t hs = headw ? (uint32_t)(31 - __clz((int)headw)) : 0u, hm = headw - 1u;   // headw = 2^hs
    // group exits, as in am_k_cblk_walk: the slot reached from head slot h of group gi's first block once the orbit is past
    // the group, or the slot inside the group whose link is not a slot (END / OUT).  Every entry's orbit runs to the end of
    // the chunk: block by block that was a dependent LDS read per block (189 at the bench density), through this table it
    // is at most AM_CB_GROUP - 1 hops to the next group's first block and one read per group from there.
    for (uint32_t idx = threadIdx.x; idx < ngrp * headw; idx += blockDim.x) {
        const uint32_t gi = idx >> hs, h = idx & hm;
        const uint32_t limit = (gi + 1u) * AM_CB_GROUP;      // first block of the next group
        uint32_t slot = ((gi * AM_CB_GROUP) << hs) + h;
        for (int hop = 0; hop < AM_CB_GROUP; ++hop) {        // (a link leads to a later block: <= AM_CB_GROUP hops)
            const uint32_t nx = lnk[slot];
            if (nx >= AM_CB_OUT) break;
            slot = nx;
            if ((slot >> hs) >= limit) break;
        }
        gex[idx] = (uint16_t)slot;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        am_shard_exit t;
        if (i >= M) {                                        // (capacity launch) end marker: "no candidate here"
            if (i == M) {                                    // (the reader stops at the first one)
                t.pos = ~(uint64_t)0;
                t.exit = 0;
                table[i] = t;
            }
            continue;
        }
        // the table ends with the first candidate at or past lead_end: later entries are never read
        if (i != 0 && pos[i - 1u] >= lead_end) continue;
        t.pos = base_abs + pos[i];
        t.exit = 0;
        {
            uint32_t g = i, ent = i;                         // ent: entry node of the last block the orbit touches
            while (g < M) {
                ent = g;
                const uint32_t kb = g / AM_CB, ki = g % AM_CB;
                if (ki >= headw) { g = __builtin_nontemporal_load(&exitnode[g]); continue; }
                uint32_t slot = (kb << hs) + ki, nx;
                for (;;) {
                    const uint32_t b = slot >> hs;
                    if (b % AM_CB_GROUP == 0u) {                             // a group's first block: through the group in one read
                        const uint32_t gi = b / AM_CB_GROUP;
                        const uint32_t r = gex[(gi << hs) + (slot & hm)];
                        slot = r;
                        if ((r >> hs) >= (gi + 1u) * AM_CB_GROUP) continue;  // past the group
                        nx = lnk[slot];                                      // stuck inside the group: its link is END or OUT
                        break;
                    }
                    nx = lnk[slot];                                          // (plain hops to the next group's first block)
                    if (nx >= AM_CB_OUT) break;
                    slot = nx;
                }
                ent = (slot >> hs) * AM_CB + (slot & hm);
                if (nx == AM_CB_END) break;
                g = __builtin_nontemporal_load(&exitnode[ent]);
            }
            t.exit = base_abs + tgt[lastnode[ent]];
        }
        table[i] = t;
    }
}

// What the scan does with the candidates it visits (preamble_impl.cc:209-237).
struct am_emit_args {
    const uint8_t *valid;
    const uint32_t *pos, *e, *tgt;
    const float *inavg;             // reference level of every candidate (goes into the hit records)
    uint32_t emit_max;              // room rule (:212): a valid hit too close to the end of the stream is not
                                    // emitted (and nothing after it can be)
    uint32_t own_lo, own_hi;        // first-stage positions this GPU's time chunk owns (everything on one GPU)
    uint4 *emit_idx;                // out: the candidates to extract, in position order: {candidate index, first-stage position,
                                    // refined position, reference level (bits)} -- everything the extraction kernels need of a hit
                                    // in ONE load (round 5: index -> e / inavg / pos was a dependent memory round trip per hit)
    uint32_t *n_out;                // out: how many
    unsigned long long *slots;      // chained scan of the per-block hit counts (am_chain_prefix)
    uint32_t epoch;
    uint32_t *ticket;               // ... and the counter its places are drawn from (am_chain_place)
    uint32_t ticket_base;
    uint32_t *scalars;              // [0]: raised to the largest resume target of a visited candidate
    int want_resume;
};

__global__ void __launch_bounds__(AM_CB_THREADS)
am_k_cblk_mark(const uint32_t *__restrict__ jump0, const uint32_t *__restrict__ entry, uint32_t Mcap,
               am_emit_args ea, const uint32_t *__restrict__ Mp)
{
    __shared__ uint16_t J[AM_CB_LEVELS][AM_CB];
    __shared__ uint8_t V[AM_CB];
    __shared__ uint32_t wc[AM_CB_PER][AM_CB_THREADS / AM_WAVE], wmax[AM_CB_THREADS / AM_WAVE];
    __shared__ uint32_t red[AM_CB_THREADS / AM_WAVE];
    __shared__ uint32_t tick;
#if defined(AM_MARK_PROF)
    long long mp[8]; int mpn = 0;
#define AM_MSTAMP() do { __syncthreads(); mp[mpn++] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
    AM_MSTAMP();
#else
#define AM_MSTAMP() ((void)0)
#endif
    const uint32_t M = am_count(Mcap, Mp);
    const uint32_t blk = am_chain_place(ea.ticket, ea.ticket_base, &tick);   // block of candidates = place in the chain
    const uint32_t base = blk * AM_CB;
    const uint32_t ent = (base < M) ? entry[blk] : AM_CB_NONE;
    const int lane = threadIdx.x & (AM_WAVE - 1), w = threadIdx.x / AM_WAVE;
    uint32_t embits = 0, tmax = 0;                            // bit k: node threadIdx.x + k * AM_CB_THREADS is a hit
    uint32_t hp[AM_CB_PER], he[AM_CB_PER];                    // position / refined position / reference level of the thread's nodes
    float hav[AM_CB_PER];                                     // (what a hit's record carries; loaded with the flags below)
#pragma unroll
    for (int k = 0; k < AM_CB_PER; ++k) { hp[k] = 0u; he[k] = 0u; hav[k] = 0.0f; }
    if (ent != AM_CB_NONE) {                                  // (uniform; otherwise the scan jumps over this block or nothing is here)
        const uint32_t end = (base + AM_CB < M) ? base + AM_CB : M;
        const uint32_t n = end - base;
        const uint16_t OUT = (uint16_t)AM_CB;                // "leaves the block"
        {
            uint32_t j[AM_CB_PER];                            // (unconditional loads from clamped indices, all in flight at once)
#pragma unroll
            for (int k = 0; k < AM_CB_PER; ++k) {
                const uint32_t i = threadIdx.x + k * AM_CB_THREADS;
                j[k] = jump0[base + (i < n ? i : n - 1u)];
            }
#pragma unroll
            for (int k = 0; k < AM_CB_PER; ++k) {
                const uint32_t i = threadIdx.x + k * AM_CB_THREADS;
                J[0][i] = (i < n && j[k] < end) ? (uint16_t)(j[k] - base) : OUT;
                V[i] = 0;
            }
        }
        __syncthreads();
        // links 2^l hops ahead, l = 1 .. : only as far as the orbit of the entry node needs them -- once J[l][entry] leaves the
        // block, the orbit has at most 2^l nodes here
        // (the thread's eight nodes side by side, as in am_k_cblk_exit: reads, dependent reads, stores)
        AM_MSTAMP();                                          // 1: successors loaded
        const uint32_t ent_l = ent - base;
        int nlev = 1;                                         // levels J[0 .. nlev) exist
        while (nlev < AM_CB_LEVELS && J[nlev - 1][ent_l] != OUT) {               // (uniform: every thread reads the same word)
            const int l = nlev;
            uint16_t t[AM_CB_PER], u[AM_CB_PER];
#pragma unroll
            for (int k = 0; k < AM_CB_PER; ++k) t[k] = J[l - 1][threadIdx.x + k * AM_CB_THREADS];
#pragma unroll
            for (int k = 0; k < AM_CB_PER; ++k) u[k] = J[l - 1][(t[k] == OUT) ? 0 : t[k]];
#pragma unroll
            for (int k = 0; k < AM_CB_PER; ++k) J[l][threadIdx.x + k * AM_CB_THREADS] = (t[k] == OUT) ? OUT : u[k];
            ++nlev;
            __syncthreads();
        }
        AM_MSTAMP();                                          // 2: levels built
        // The orbit's q-th node (q = 0: the entry) is reached by the hops of q's binary digits, in any order (they are powers
        // of one map): thread t enumerates q = t, t + 256, ... on its own -- up to 11 dependent LDS reads each, eight chains
        // side by side, no barrier in between -- and marks what it reaches.  (Round 2 pushed the marks down level by level, 11
        // more barrier-separated passes over the block: the same 4.5 us -- both are bound by the random 16-bit LDS gathers,
        // 88 per thread.  Phase clocks of a -DAM_MARK_PROF build, us: successors 1.1, levels 3.0, orbit 4.5, the visited
        // nodes' records 1.4, counts 0.7, waiting for the blocks before 0.4-3.4, index stores 2.)
        {
            uint32_t x[AM_CB_PER];
#pragma unroll
            for (int k = 0; k < AM_CB_PER; ++k) {
                const uint32_t q = threadIdx.x + k * AM_CB_THREADS;
                x[k] = (nlev < AM_CB_LEVELS && (q >> nlev) != 0u) ? (uint32_t)OUT : ent_l;   // (beyond the orbit's length)
            }
            for (int l = 0; l < nlev; ++l) {
#pragma unroll
                for (int k = 0; k < AM_CB_PER; ++k) {
                    const uint32_t q = threadIdx.x + k * AM_CB_THREADS;
                    const uint32_t nx = J[l][x[k] == OUT ? 0u : x[k]];
                    if ((q >> l) & 1u) x[k] = (x[k] == OUT) ? (uint32_t)OUT : nx;
                }
            }
#pragma unroll
            for (int k = 0; k < AM_CB_PER; ++k)
                if (x[k] != OUT) V[x[k]] = 1;
        }
        __syncthreads();
        AM_MSTAMP();                                          // 3: orbit marked
        // which visited nodes are hits, and where the scan resumes after everything visited here: the largest
        // target (only needed when the stream continues)
        // (all of a thread's loads go out together, visited or not: written as one node after the other -- position, then
        // `valid`, then `e` behind the short-circuit, eight nodes in turn -- the kernel spent most of its time in up to 24
        // serial memory round trips per thread)
        bool vis[AM_CB_PER];
        uint32_t p[AM_CB_PER], ee[AM_CB_PER], tg[AM_CB_PER];
        uint8_t va[AM_CB_PER];
#pragma unroll
        for (int k = 0; k < AM_CB_PER; ++k) {
            const uint32_t i = threadIdx.x + k * AM_CB_THREADS;
            vis[k] = i < n && V[i] != 0;
        }
#pragma unroll
        for (int k = 0; k < AM_CB_PER; ++k) {
            const uint32_t i = threadIdx.x + k * AM_CB_THREADS;
            const uint32_t g = base + (i < n ? i : n - 1u);   // (a node of the block in any case: no branch around the loads)
            p[k] = ea.pos[g];
            va[k] = ea.valid[g];
            ee[k] = ea.e[g];
            tg[k] = ea.tgt[g];
            hav[k] = ea.inavg[g];
        }
#pragma unroll
        for (int k = 0; k < AM_CB_PER; ++k) {
            const bool em = vis[k] && va[k] != 0 && ee[k] <= ea.emit_max && p[k] >= ea.own_lo && p[k] < ea.own_hi;
            if (vis[k] && ea.want_resume) tmax = tg[k] > tmax ? tg[k] : tmax;
            if (em) embits |= 1u << k;
            hp[k] = p[k]; he[k] = ee[k];
        }
    }
    AM_MSTAMP();                                              // 4 (or 1): hits known
    // ordered compaction of the hits, in the same launch: counts per (round k, wave) -> this block's total -> the
    // totals of the blocks before it (am_chain_prefix) -> every hit's index in emit_idx[].  Node order is
    // (k, wave, lane).
    for (int k = 0; k < AM_CB_PER; ++k) {
        const unsigned long long m = __ballot((embits >> k) & 1u);
        if (lane == 0) wc[k][w] = (uint32_t)__popcll(m);
    }
    for (int o = 32; o >= 1; o >>= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)tmax, o, AM_WAVE);
        tmax = other > tmax ? other : tmax;
    }
    if (lane == 0) wmax[w] = tmax;
    __syncthreads();
    uint32_t tot = 0;
    for (int k = 0; k < AM_CB_PER; ++k)
        for (int q = 0; q < AM_CB_THREADS / AM_WAVE; ++q) tot += wc[k][q];
    // Same-address atomics serialise (~11 ns each): one per workgroup.
    if (threadIdx.x == 0 && ea.want_resume) {
        uint32_t m = 0;
        for (int k = 0; k < AM_CB_THREADS / AM_WAVE; ++k) m = wmax[k] > m ? wmax[k] : m;
        if (m) atomicMax(&ea.scalars[0], m);
    }
    AM_MSTAMP();
    const uint32_t before = am_chain_prefix(ea.slots, blk, ea.epoch, tot, red);
    AM_MSTAMP();                                              // chain prefix
    if (before == AM_CHAIN_FAIL) {                            // (uniform) a place was never published: no stores, the error word, no hits
        if (threadIdx.x == 0) { ea.scalars[9] = 1u; if (blk == gridDim.x - 1) *ea.n_out = 0u; }
        return;
    }
    uint32_t off = before;
#pragma unroll
    for (int k = 0; k < AM_CB_PER; ++k) {
        const bool em = ((embits >> k) & 1u) != 0u;
        const unsigned long long m = __ballot(em);
        uint32_t o = off;
        for (int q = 0; q < w; ++q) o += wc[k][q];
        if (em) {
            const uint32_t g = base + threadIdx.x + (uint32_t)k * AM_CB_THREADS;
            uint4 rec;
            rec.x = g; rec.y = hp[k]; rec.z = he[k]; rec.w = __float_as_uint(hav[k]);
            ea.emit_idx[o + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = rec;
        }
        for (int q = 0; q < AM_CB_THREADS / AM_WAVE; ++q) off += wc[k][q];
    }
    if (blk == gridDim.x - 1 && threadIdx.x == 0) *ea.n_out = before + tot;
#if defined(AM_MARK_PROF)
    AM_MSTAMP();
    if (threadIdx.x == 0 && (blk == 0 || blk == gridDim.x / 2 || blk == gridDim.x - 1 || blk == M / AM_CB)) {
        printf("mark blk %u/%u ent %d:", blk, gridDim.x, ent != AM_CB_NONE);
        for (int k = 1; k < mpn; ++k) printf(" %lld", mp[k] - mp[k - 1]);
        printf("  (10 ns units; start %lld)\n", mp[0] % 1000000);
    }
#endif
}

// compute units of the current device (cached per device: a process may hold contexts on several)
int am_device_cus(void)
{
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        hipDeviceProp_t prop;
        n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

static inline unsigned am_grid(uint64_t n, unsigned block) { return (unsigned)((n + block - 1) / block); }

#define AM_CB_WALK_LDS (152 * 1024)  /* dynamic LDS the walk kernels may use */
// dynamic LDS of am_k_cblk_walk for nblk blocks with w-slot heads: links | entry index per block | group exits | group entries
static size_t am_chain_walk_lds_bytes(uint64_t nblk, uint64_t w)
{
    const uint64_t ngrp = (nblk + AM_CB_GROUP - 1) / AM_CB_GROUP;
    return (size_t)((nblk * w + 8 + nblk + ngrp * w + ngrp + 2) * sizeof(uint16_t));
}
static uint32_t am_chain_headw(uint32_t nblk)
{
    // the widest head (a power of two) whose table fits: slots are 16-bit with two values reserved, and the walk's
    // other tables share the LDS;  0: every step of the walk reads global memory
    uint32_t w = AM_CB_HEADW;
    while (w >= 1 && ((uint64_t)nblk * w > AM_CB_HEADCAP || am_chain_walk_lds_bytes(nblk, w) > AM_CB_WALK_LDS)) w >>= 1;
    return w;
}

struct am_chain_layout {
    uint32_t nblk, headw;
    size_t off_last, off_entry, off_head, words;    // offsets (in words) into the scratch buffer
};
static am_chain_layout am_chain_layout_of(uint32_t M)
{
    am_chain_layout L;
    L.nblk = (M + AM_CB - 1) / AM_CB;
    L.headw = am_chain_headw(L.nblk);
    L.off_last = (size_t)M + 1;                     // exitnode[M+1] | lastnode[M+1] | entry[nblk+8] | headexit
    L.off_entry = L.off_last + (size_t)M + 1;
    L.off_head = (L.off_entry + L.nblk + 8 + 3) & ~(size_t)3;     // (16-byte aligned: the walk copies it 16 bytes at a time)
    L.words = L.off_head + (size_t)L.nblk * L.headw + 8;
    return L;
}
size_t am_chain_scratch_bytes(uint32_t M) { return am_chain_layout_of(M).words * sizeof(uint32_t); }

// the walk kernels may need more than the default 64 KB of dynamic LDS: raised once per device and kernel
// (`done` is the caller's per-kernel table; a process may hold contexts on several devices)
static hipError_t am_chain_walk_lds(const void *kernel, std::atomic<bool> (&done)[64])
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && done[dev]) return hipSuccess;
    hipError_t rc = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)AM_CB_WALK_LDS);
    if (rc == hipSuccess && dev >= 0 && dev < 64) done[dev] = true;
    return rc;
}

// step 1 (independent of where the scan starts): successor array + per-block exits
hipError_t am_launch_chain_prepare(const uint32_t *pos, const uint32_t *tgt, uint32_t M, uint32_t *jump0,
                                   uint32_t *scratch, int want_last, hipStream_t s, const uint32_t *Mp, int have_succ)
{
    if (M == 0) return hipSuccess;
    const am_chain_layout L = am_chain_layout_of(M);
    if (!have_succ)                                           // (am_k_cand already wrote the successors)
    hipLaunchKernelGGL(am_k_chain_succ, dim3(am_grid((uint64_t)M + 1, 256)), dim3(256), 0, s, pos, tgt, M, jump0, Mp);
    hipLaunchKernelGGL(am_k_cblk_exit, dim3(L.nblk), dim3(AM_CB_THREADS), 0, s, jump0, M, L.headw, scratch,
                       want_last ? scratch + L.off_last : nullptr, reinterpret_cast<uint16_t *>(scratch + L.off_head), Mp);
    return hipGetLastError();
}

// steps 2 + 3: which candidates the scan that starts at position cur0 visits, and what it does with them:
// emit_idx[0 .. *n_out) = the hits in position order, scalars[0] = resume position
hipError_t am_launch_chain_visit(const uint32_t *pos, const uint32_t *jump0, uint32_t M, uint32_t cur0,
                                 uint32_t *scratch, const uint8_t *valid, const uint32_t *e, const uint32_t *tgt,
                                 uint32_t emit_max, uint32_t own_lo, uint32_t own_hi, uint4 *emit_idx, uint32_t *n_out,
                                 unsigned long long *slots, uint32_t epoch, uint32_t *ticket, uint32_t *ticket_base,
                                 uint32_t *scalars, int want_resume, hipStream_t s, const uint32_t *Mp, const am_entry_src *entry_src,
                                 const float *inavg, hipEvent_t after_walk)
{
    if (M == 0) return hipSuccess;
    const am_chain_layout L = am_chain_layout_of(M);
    static std::atomic<bool> attr_set[64];
    if (hipError_t rc = am_chain_walk_lds(reinterpret_cast<const void *>(&am_k_cblk_walk), attr_set); rc != hipSuccess)
        return rc;
    const size_t lds = am_chain_walk_lds_bytes(L.nblk, L.headw);
    if (lds > AM_CB_WALK_LDS) return hipErrorInvalidValue;   // (more than ~75 000 blocks of 2048 candidates in one scan)
    am_entry_src es;
    if (entry_src) es = *entry_src;
    else { es.msgs = nullptr; es.world = 0; es.rank = 0; es.cap = 0; es.base_abs = 0; es.flags = nullptr; es.exit_out = nullptr; es.cur_in = nullptr; }
    hipLaunchKernelGGL(am_k_cblk_walk, dim3(1), dim3(1024), lds, s, pos, scratch, reinterpret_cast<const uint16_t *>(scratch + L.off_head), M, L.nblk,
                       L.headw, cur0, scratch + L.off_entry, scalars, Mp, es);
    am_emit_args ea;
    ea.valid = valid; ea.pos = pos; ea.e = e; ea.tgt = tgt; ea.inavg = inavg; ea.emit_max = emit_max; ea.own_lo = own_lo;
    ea.own_hi = own_hi; ea.emit_idx = emit_idx; ea.n_out = n_out; ea.slots = slots; ea.epoch = epoch; ea.scalars = scalars;
    ea.want_resume = want_resume;
    ea.ticket = ticket; ea.ticket_base = *ticket_base;
    if (hipError_t rc = hipGetLastError(); rc != hipSuccess) return rc;   // (the walk's launch)
    // (am_spipe: the walk has written where the scan leaves this chunk -- all the next chunk's resolve step waits for)
    if (after_walk) if (hipError_t rc = hipEventRecord(after_walk, s); rc != hipSuccess) return rc;
    hipLaunchKernelGGL(am_k_cblk_mark, dim3(L.nblk), dim3(AM_CB_THREADS), 0, s, jump0, scratch + L.off_entry, M, ea,
                       Mp);
    const hipError_t rc = hipGetLastError();
    if (rc == hipSuccess) *ticket_base += L.nblk;            // (every workgroup of a launch that happened draws one)
    return rc;
}

// exit table of a time chunk for its first n candidates (needs am_launch_chain_prepare(want_last = 1))
hipError_t am_launch_chain_exit_table(const uint32_t *pos, const uint32_t *tgt, uint32_t M, uint32_t n,
                                      uint32_t lead_end, uint32_t *scratch, uint64_t base_abs, am_shard_exit *table,
                                      hipStream_t s, const uint32_t *Mp, am_shard_exit *header, const uint64_t *carry)
{
    if (n == 0 || M == 0) return hipSuccess;
    const am_chain_layout L = am_chain_layout_of(M);
    static std::atomic<bool> attr_set[64];
    if (hipError_t rc = am_chain_walk_lds(reinterpret_cast<const void *>(&am_k_cblk_exit_table), attr_set);
        rc != hipSuccess)
        return rc;
    const size_t lds = am_chain_walk_lds_bytes(L.nblk, L.headw);          // (links + group exits: the walk's layout holds both)
    if (lds > AM_CB_WALK_LDS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(am_k_cblk_exit_table, dim3(1), dim3(1024), lds, s, pos, tgt, scratch, scratch + L.off_last,
                       reinterpret_cast<const uint16_t *>(scratch + L.off_head), M, L.nblk, L.headw, n, lead_end, base_abs, table, Mp, header, carry);
    return hipGetLastError();
}

// The composition as a launch of its own (am_shard_entry_wave): where nothing is left to slice -- otherwise the block walk
// composes the entry itself.  Writes the array coordinate at which the scan enters chunk `rank` (cur0_out), where it leaves
// it (*exit_out) and flags[0] = 1 if some table did not fit its message.  With msgs == null (a chunk without a single
// candidate before the first exchange): the header of the own message only -- {0, 0}, {carry, 0}.
__global__ void __launch_bounds__(AM_WAVE)
am_k_shard_entry(const am_shard_exit *__restrict__ msgs, uint32_t world, uint32_t rank, uint32_t cap,
                 uint64_t base_abs, uint32_t *__restrict__ cur0_out, uint32_t *__restrict__ flags,
                 uint64_t *__restrict__ exit_out, am_shard_exit *__restrict__ header, const uint64_t *__restrict__ carry,
                 const uint64_t *__restrict__ cur_in, uint64_t *__restrict__ carry_out)
{
    if (blockIdx.x != 0) return;
    if (header) {
        if (threadIdx.x == 0) { header[0].pos = 0; header[0].exit = 0; header[1].pos = *carry; header[1].exit = 0; }
        return;
    }
    uint32_t bad = 0;
    uint64_t leave = 0;
    const uint64_t cur = am_shard_entry_wave(msgs, world, rank, cap, (int)(threadIdx.x & (AM_WAVE - 1)), &bad, &leave, cur_in, carry_out);
    if (threadIdx.x != 0) return;
    uint64_t rel = cur > base_abs ? cur - base_abs : 0;
    if (rel > 0xFFFFFFF0ull) rel = 0xFFFFFFF0ull;
    *cur0_out = (uint32_t)rel;
    flags[0] = bad;
    if (!bad) *exit_out = leave;
}

hipError_t am_launch_shard_entry(const am_shard_exit *msgs, uint32_t world, uint32_t rank, uint32_t cap, uint64_t base_abs,
                                 uint32_t *cur0_out, uint32_t *flags, uint64_t *exit_out, hipStream_t s, const uint64_t *cur_in,
                                 uint64_t *carry_out)
{
    hipLaunchKernelGGL(am_k_shard_entry, dim3(1), dim3(AM_WAVE), 0, s, msgs, world, rank, cap, base_abs, cur0_out, flags, exit_out,
                       (am_shard_exit *)nullptr, (const uint64_t *)nullptr, cur_in, carry_out);
    return hipGetLastError();
}
hipError_t am_launch_shard_header(am_shard_exit *header, const uint64_t *carry, hipStream_t s)
{
    hipLaunchKernelGGL(am_k_shard_entry, dim3(1), dim3(AM_WAVE), 0, s, (const am_shard_exit *)nullptr, 0u, 0u, 0u, (uint64_t)0,
                       (uint32_t *)nullptr, (uint32_t *)nullptr, (uint64_t *)nullptr, header, carry, (const uint64_t *)nullptr,
                       (uint64_t *)nullptr);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Burst extraction + tag (a9): one wave per emitted preamble (am_k_extract_slice, after the slicer below).
//   out[c] = in[e + c*spc] - inavg[e]          preamble_impl.cc:219-221
//   timestamp from the absolute item count     preamble_impl.cc:100-137, relative to the rx_time
//   tag in force at that item (tt[0..ntt) ascending; none: offset 0, time 0 = a file source)
// ------------------------------------------------------------------------------------------
// the "preamble_found" tag of item count `sample` (= stream index + history - 1, as the preamble block
// numbers its items): time stamp relative to the rx_time tag in force    preamble_impl.cc:100-137
__device__ __forceinline__ am_tag am_make_tag(uint64_t sample, uint64_t rate, const am_time_tag *__restrict__ tt,
                                          uint32_t ntt)
{
    am_tag t;
    t.sample = sample;
    uint64_t off = 0, whole = 0;                                // :103-108 no tag yet
    double fr = 0.0;
    uint32_t lo = 0, hi = ntt;                                  // last tag with offset <= sample
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tt[mid].offset <= t.sample) lo = mid + 1; else hi = mid;
    }
    if (lo) { off = tt[lo - 1].offset; whole = tt[lo - 1].secs; fr = tt[lo - 1].frac; }   // :110-111
    const uint64_t d = t.sample - off;
    t.secs = whole + d / rate;                                  // :124,127
    t.frac = fr + (double)(d % rate) / (double)rate;            // :125,128
    if (t.frac > 1.0f) { t.frac -= 1.0f; t.secs += 1; }         // :129-132
    t.inavg = 0.0f;
    t.how_late = 0;
    return t;
}

// ------------------------------------------------------------------------------------------
// Slicer + CRC (a10-a12): one wave per burst, lane l slices bits l and l+64.
// The syndrome is formed as the XOR of x^(nbits-1-j) mod G over the set bits j (CRC is
// linear), reduced across the wave with xor-shuffles.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int am_chip_pair_slice(float c0, float c1, float lo, float hi, double half_lo)
{
    // slicer_impl.cc:74-98; returns decision | confidence << 1
    const bool in0 = (c0 > lo) && (c0 < hi);
    const bool in1 = (c1 > lo) && (c1 < hi);
    int dec, conf;
    if (in0 && !in1) { dec = 1; conf = 1; }
    else if (in1 && !in0) { dec = 0; conf = 1; }
    else if (in0 && in1) { dec = (c0 > c1) ? 1 : 0; conf = 0; }
    else {
        dec = (c0 > c1) ? 1 : 0;
        const float loser = dec ? c1 : c0;
        conf = ((double)loser < half_lo) ? 1 : 0;
    }
    return dec | (conf << 1);
}

__device__ __forceinline__ uint32_t am_bitrev8(uint32_t v)
{
    v = ((v & 0xF0u) >> 4) | ((v & 0x0Fu) << 4);
    v = ((v & 0xCCu) >> 2) | ((v & 0x33u) << 2);
    v = ((v & 0xAAu) >> 1) | ((v & 0x55u) << 1);
    return v;
}

// the slicer's long / short decision from the first five data bits of burst b (what am_slice_wave derives from its
// ballots: slicer_impl.cc:128-140): callers that form the soft chips themselves skip chips 128.. of a short packet
__device__ __forceinline__ bool am_burst_is_long(const float *b)
{
    float s = b[0] + b[2];
    s = s + b[7];
    s = s + b[9];
    const float ref = (float)((double)s / 4.0);
    const float hi = (float)((double)ref * 1.414);
    const float lo = (float)((double)ref * 0.707);
    const double half_lo = (double)lo * 0.5;
    uint32_t hdr = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) hdr = (hdr << 1) | (uint32_t)(am_chip_pair_slice(b[16 + 2 * k], b[17 + 2 * k], lo, hi, half_lo) & 1);
    return hdr == 16 || hdr == 17 || hdr == 20 || hdr == 21;
}

// one wave slices burst b (240 soft chips, global memory or LDS); lane 0 files the packet under index i
__device__ __forceinline__ void am_slice_wave(const float *b, const am_tag &t, uint32_t i, int lane,
                                              const uint32_t *__restrict__ crc_pow, am_packet *__restrict__ packets)
{
    float s = b[0] + b[2];                                // slicer_impl.cc:128-131
    s = s + b[7];
    s = s + b[9];
    const float ref = (float)((double)s / 4.0);
    const float hi = (float)((double)ref * 1.414);        // :71
    const float lo = (float)((double)ref * 0.707);        // :72
    const double half_lo = (double)lo * 0.5;              // :92,:95
    const float *d = b + 16;                              // :133
    const int r0 = am_chip_pair_slice(d[2 * lane], d[2 * lane + 1], lo, hi, half_lo);
    int r1 = 2;                                           // bits >= 112 do not exist
    if (lane < 112 - 64) r1 = am_chip_pair_slice(d[2 * (lane + 64)], d[2 * (lane + 64) + 1], lo, hi, half_lo);
    unsigned long long m0 = __ballot(r0 & 1);
    unsigned long long m1 = __ballot(r1 & 1);
    unsigned long long l0 = __ballot(!(r0 & 2));
    unsigned long long l1 = __ballot(!(r1 & 2));
    const uint32_t hdr = am_bitrev8((uint32_t)(m0 & 0x1Full)) >> 3;       // :135-139
    const bool longpkt = (hdr == 16 || hdr == 17 || hdr == 20 || hdr == 21);   // :140
    const int nbits = longpkt ? 112 : 56;
    if (!longpkt) {
        m0 &= (1ull << 56) - 1ull; m1 = 0ull;
        l0 &= (1ull << 56) - 1ull; l1 = 0ull;
    }
    int nlow = __popcll(l0) + __popcll(l1);
    if (nlow > 24) nlow = 24;                             // :157 (counter saturates)
    // syndrome
    uint32_t part = 0;
    if ((m0 >> lane) & 1ull) part ^= crc_pow[nbits - 1 - lane];
    if (lane < 48 && ((m1 >> lane) & 1ull)) part ^= crc_pow[nbits - 1 - (lane + 64)];
    for (int o = 32; o >= 1; o >>= 1) part ^= (uint32_t)__shfl_xor((int)part, o, AM_WAVE);
    if (lane == 0) {
        am_packet p;
        for (int k = 0; k < 8; ++k) p.data[k] = (uint8_t)am_bitrev8((uint32_t)((m0 >> (8 * k)) & 0xFFull));
        for (int k = 0; k < 6; ++k) p.data[8 + k] = (uint8_t)am_bitrev8((uint32_t)((m1 >> (8 * k)) & 0xFFull));
        const uint32_t mt = (uint32_t)(p.data[0] >> 3) & 0x1Fu;              // :168
        bool ok = (m0 | m1) != 0ull;                                         // :162-166
        if (!longpkt && mt != 11 && nlow > 0) ok = false;                    // :170
        if (mt == 11 && nlow >= 10) ok = false;                              // :171
        if (part != 0 && (mt == 11 || mt == 17)) ok = false;                 // :182
        p.nbytes = (uint8_t)(nbits / 8);
        p.df = (uint8_t)mt;
        p.numlowconf = (uint8_t)nlow;
        p.reserved[0] = ok ? 1 : 0;
        p.reserved[1] = 0;
        p.reserved[2] = 0;
        p.crc = part;
        p.ref = ref;
        p.reserved2 = 0;
        if (!ok) {
            // never handed out: only the flag travels (the packet array is usually pinned host memory)
            packets[i].reserved[0] = 0;
            return;
        }
        p.sample = t.sample;
        p.secs = t.secs;
        p.frac = t.frac;
        packets[i] = p;
    }
}

__global__ void __launch_bounds__(256)
am_k_slice(const float *__restrict__ bursts, const am_tag *__restrict__ tags, const uint32_t *__restrict__ n_ptr,
           const uint32_t *__restrict__ crc_pow, am_packet *__restrict__ packets,
           const uint32_t *__restrict__ scalars, uint32_t *__restrict__ host_out, const uint32_t *__restrict__ Mp)
{
    const int lane = threadIdx.x & (AM_WAVE - 1);
    const uint32_t i = blockIdx.x * (blockDim.x / AM_WAVE) + threadIdx.x / AM_WAVE;
    // hit count and scan resume position go straight to pinned host memory (no extra copies)
    if (host_out && blockIdx.x == 0 && threadIdx.x == 0) {
        host_out[0] = *n_ptr;
        host_out[1] = scalars[0];
        host_out[2] = Mp ? *Mp : 0u;                          // actual candidate count (speculative launches)
    }
    if (i >= *n_ptr) return;                              // wave-uniform; device-side burst count
    const am_tag t = tags[i];                             // (lane 0 uses it)
    am_slice_wave(bursts + (size_t)i * AM_BURST, t, i, lane, crc_pow, packets);
}

// Extraction and slicing in one launch (the scan paths: every extracted burst is sliced right away).  One
// wave per hit: the 240 soft chips go through LDS instead of a bursts[] round trip through memory; bursts_out
// and tags_out are written only when the caller wants them (block-level API), packets as am_k_slice does.
__global__ void __launch_bounds__(256)
am_k_extract_slice(const float *__restrict__ bb, const float *__restrict__ inavg, int spc,
                   const int *__restrict__ chip_idx, int hist0,
                   const uint4 *__restrict__ emit_idx, const uint32_t *__restrict__ n_ptr,
                   const uint32_t *__restrict__ pos, const uint32_t *__restrict__ eo, uint64_t base_abs,
                   long long e_off, uint64_t rate, const am_time_tag *__restrict__ tt, uint32_t ntt,
                   float *__restrict__ bursts_out, am_tag *__restrict__ tags_out,
                   const uint32_t *__restrict__ crc_pow, am_packet *__restrict__ packets,
                   const uint32_t *__restrict__ scalars, uint32_t *__restrict__ host_out,
                   const uint32_t *__restrict__ Mp)
{
    __shared__ float sb[256 / AM_WAVE][AM_BURST];
    const int lane = threadIdx.x & (AM_WAVE - 1), wv = threadIdx.x / AM_WAVE;
    const uint32_t i = blockIdx.x * (blockDim.x / AM_WAVE) + wv;
    if (host_out && blockIdx.x == 0 && threadIdx.x == 0) {
        host_out[0] = *n_ptr;
        host_out[1] = scalars[0];
        host_out[2] = Mp ? *Mp : 0u;                          // actual candidate count (speculative launches)
        host_out[5] = scalars[9];                             // a chained scan of this step gave up (am_chain_prefix)
    }
    if (i >= *n_ptr) return;                              // wave-uniform; device-side hit count
    const uint4 rec = emit_idx[i];                          // {candidate, first-stage position, refined position, reference level}
    const uint32_t g = rec.x;
    const uint32_t e = rec.z;
    const size_t ei = (size_t)((long long)e + e_off);      // index of e in this GPU's bb/avg
    const float av = __uint_as_float(rec.w);                // reference level at the shifted start
    for (int c = lane; c < AM_BURST; c += AM_WAVE) {
        const float v = bb[ei + (size_t)(chip_idx ? chip_idx[c] : c * spc)] - av;    // preamble_impl.cc:219-221: in[i + int(j * spc)]
        sb[wv][c] = v;
        if (bursts_out) bursts_out[(size_t)i * AM_BURST + c] = v;
    }
    am_tag t = am_make_tag(base_abs + e + (uint64_t)hist0, rate, tt, ntt);
    t.inavg = av;
    t.how_late = e - pos[g];
    if (tags_out && lane == 0) tags_out[i] = t;
    __builtin_amdgcn_wave_barrier();                       // the wave's own LDS writes, in order, before its reads
    am_slice_wave(sb[wv], t, i, lane, crc_pow, packets);
}

hipError_t am_launch_extract_slice(const float *bb, const float *inavg, int spc, const int *chip_idx, int hist0,
                                   const uint4 *emit_idx,
                                   const uint32_t *n_ptr, uint32_t n_max, const uint32_t *pos, const uint32_t *e,
                                   uint64_t base_abs, long long e_off, uint64_t rate, const am_time_tag *tt,
                                   uint32_t ntt, float *bursts_out, am_tag *tags_out, const uint32_t *crc_pow,
                                   am_packet *packets, const uint32_t *scalars, uint32_t *host_out, hipStream_t s,
                                   const uint32_t *Mp)
{
    if (n_max == 0) return hipSuccess;
    hipLaunchKernelGGL(am_k_extract_slice, dim3(am_grid(n_max, 4)), dim3(256), 0, s, bb, inavg, spc, chip_idx, hist0, emit_idx, n_ptr,
                       pos, e, base_abs, e_off, rate, tt, ntt, bursts_out, tags_out, crc_pow, packets, scalars,
                       host_out, Mp);
    return hipGetLastError();
}

// Extraction + slicing when bb exists only around the candidates (streaming front end): the 240 chip-spaced
// samples of a hit are recomputed from IQ in the canonical order: sample n of chip q (offset i) is
//   bb[n] = fl( (suf + pre) * s1 ),  pre = m[q*spc] + ... + m[n] left->right,
//                                    suf = m[q*spc-1] + ... + m[n-spc+1] right->left (absent for i = spc-1),
// m = |iq|^2, zero outside the stream; bb itself reads zero beyond the end of the stream.
// Lane c of a workgroup forms sample c: the spc samples its filter window spans are its own 8*spc contiguous bytes
// of IQ, loaded with 16-byte loads that are all in flight at once and summed in registers -- all 240 samples of a
// burst sit at the same offset i inside their chips, so the split of the window into pre and suf is uniform.
// The loads are what the kernel costs (ablation builds: 55 us at the bench density, 11 us without them; staging
// the window through LDS with coalesced loads -- one hit per workgroup or grid-stride, whole or half bursts, per
// wave with a row transposition --, slicing hit k under the loads of hit k + 1, packets through device memory:
// 51-68 us, every one of them: profiles/r2_final/README.md), so it reads fewer bytes where it can: chips 0..127
// first (preamble + 56 bits), the slicer's long / short decision from the first five bits, and chips 128..239
// only for a long packet.
// A workgroup takes hits grid-stride over the device-side hit count: the launch is sized for the chip, not for
// the bound on the number of hits.
#if defined(AM_XPROF)
// tuning builds only: cycles per phase of the extraction kernel, summed over workgroups (am_debug_xprof reads them)
__device__ unsigned long long am_xprof_acc[8];
__device__ long long am_xprof_log[4096][4];      // per workgroup of the last launch: start, end (100 MHz real time), hits, longest hit (cycles)
#define AM_XSTAMP(k) do { if (tid == 0) { const long long now__ = (long long)__builtin_readcyclecounter(); atomicAdd(&am_xprof_acc[k], (unsigned long long)(now__ - xlast)); xlast = now__; } } while (0)
extern "C" int am_debug_xprof(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(am_xprof_acc), sizeof(am_xprof_acc)) != hipSuccess) return -1;
    {
        static long long h[4096][4];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(am_xprof_log), sizeof(h)) == hipSuccess) {
            long long t0 = 0x7fffffffffffffffll, t1 = 0, lmax = 0, hmax = 0; int n = 0; double life = 0, hits = 0;
            for (int b = 0; b < 4096; ++b) if (h[b][2] > 0) {
                if (h[b][0] < t0) t0 = h[b][0];
                if (h[b][1] > t1) t1 = h[b][1];
                if (h[b][1] - h[b][0] > lmax) lmax = h[b][1] - h[b][0];
                if (h[b][3] > hmax) hmax = h[b][3];
                life += (double)(h[b][1] - h[b][0]); hits += (double)h[b][2]; ++n;
            }
            if (n) fprintf(stderr, "xprof last launch: %d workgroups, %.0f hits, span %.2f us, workgroup life mean %.2f max %.2f us, longest hit %lld cycles\n",
                           n, hits, (double)(t1 - t0) / 100.0, life / n / 100.0, (double)lmax / 100.0, hmax);
        }
    }
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(am_xprof_acc), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define AM_XSTAMP(k) ((void)0)
#endif

// |.|^2 of absolute sample w, zero outside the source
__device__ __forceinline__ float am_mag_guarded(const float2 *__restrict__ iq2, long long src_abs0, long long src_abs1, long long w)
{
    float mm = 0.0f;
    if (w >= src_abs0 && w < src_abs1) { const float2 t = iq2[w - src_abs0]; const float r = t.x * t.x, q = t.y * t.y; mm = r + q; }
    return mm;
}

// soft chip c of the burst whose first sample has absolute index ae (bb; the reference level comes off later).
// inside (uniform): every lane's 16-byte loads lie inside the source -- all but the hits at the two ends of the stream
template <int SPC>
__device__ __forceinline__ float am_soft_chip_iq(const float *__restrict__ iq, long long src_abs0, long long src_abs1,
                                                 bool pmf, float s1, long long ae, int c, bool inside)
{
    const float2 *iq2 = reinterpret_cast<const float2 *>(iq);
    const long long n = ae + (long long)c * SPC;                              // the sample this lane forms
    if (!pmf) {
        float v = 0.0f;
        if (n >= src_abs0 && n < src_abs1) {
            const float2 t = iq2[n - src_abs0];
            const float r = t.x * t.x, q = t.y * t.y;
            v = r + q;
        }
        return v;
    }
    const int ii = (int)(ae % SPC);                                           // offset inside the chip, the same for all 240 samples
    float v;
    if (inside) {
        // window: samples n - SPC + 1 .. n; M[k] = |.|^2 of sample w0 + k, w0 = the window's first sample rounded
        // down to an even offset into iq (`odd`: uniform, the lanes are SPC samples apart)
        const long long wfirst = n - (SPC - 1);
        const int odd = (int)((wfirst - src_abs0) & 1);
        const float4 *src = reinterpret_cast<const float4 *>(iq) + ((wfirst - odd - src_abs0) >> 1);
        float4 t[SPC / 2 + 1];
#pragma unroll
        for (int k = 0; k < SPC / 2 + 1; ++k) t[k] = src[k];
        float M[SPC + 2];
#pragma unroll
        for (int k = 0; k < SPC / 2 + 1; ++k) {
            const float r0 = t[k].x * t[k].x, q0 = t[k].y * t[k].y, r1 = t[k].z * t[k].z, q1 = t[k].w * t[k].w;
            M[2 * k] = r0 + q0;
            M[2 * k + 1] = r1 + q1;
        }
        // m[w] = M[w + odd], as a bit select (a plain ?: becomes an indexed read of M[] through scratch memory)
        float m[SPC];
        const uint32_t omask = 0u - (uint32_t)odd;
#pragma unroll
        for (int w = 0; w < SPC; ++w)
            m[w] = __uint_as_float((__float_as_uint(M[w + 1]) & omask) | (__float_as_uint(M[w]) & ~omask));
        const int wb = SPC - 1 - ii;                                          // window index of the chip's first sample
        float pre = 0.0f, suf = 0.0f;
#pragma unroll
        for (int w = 0; w < SPC; ++w) {
            if (w >= wb) pre = pre + m[w];                                    // (uniform conditions)
            if (SPC - 1 - w < wb) suf = suf + m[SPC - 1 - w];
        }
        v = (ii == SPC - 1) ? pre * s1 : (suf + pre) * s1;
    } else {
        // (rare: one sample at a time, rolled loops, the canonical order directly)
        float pre = 0.0f, suf = 0.0f;
#pragma unroll 1
        for (long long w = n - ii; w <= n; ++w) pre = pre + am_mag_guarded(iq2, src_abs0, src_abs1, w);
#pragma unroll 1
        for (long long w = n - ii - 1; w >= n - SPC + 1; --w) suf = suf + am_mag_guarded(iq2, src_abs0, src_abs1, w);
        v = (ii == SPC - 1) ? pre * s1 : (suf + pre) * s1;
    }
    if (n >= src_abs1) v = 0.0f;                                              // bb reads zero beyond the end of the stream
    return v;
}

// 64 Msps, a hit whose samples are all inside the source (round 4, late): the windows of a burst's soft chips tile its
// samples -- chip c is the 32 samples that end at ae + 32 c -- so the workgroup loads them as ONE coalesced run (a wave's load is
// 1 KB of consecutive bytes, not 64 requests 256 bytes apart: the texture addressers were busy 55 % of this kernel's time),
// leaves |.|^2 in LDS rows of one window each (stride 36 floats: 16-byte row reads of consecutive lanes hit all banks), and a
// lane reads its window back 16 bytes at a time.  NS samples from absolute sample B0 (NS a multiple of 512: 256 threads x 2).
// E[(s >> 5) * 36 + (s & 31)] = |.|^2 of sample B0 + s.  The source is read 16 bytes at a time: from the even sample at or
// before B0 (one more piece, by thread 0, where B0 is odd).
#define AM_XROW 36
// (Round 5: the loads of a burst's chips 128..239 issued together with those of chips 0..127 and held in registers until the first
// five bits say whether the packet is a long one -- no second memory round trip behind the decision -- measured 43.8 against 40.0 us
// at the bench density, 20.9 against 18.7 at 2 000 bursts/s: 28 more registers and 47 % more bytes for every short packet cost more
// than the round trip; profiles/r5_fe64.  Not kept.)
template <int NS>
__device__ __forceinline__ void am_stage_energies32(const float *__restrict__ iq, long long src_abs0, long long B0, float *E, int tid)
{
    static_assert(NS % 512 == 0, "whole rounds of 256 threads x 2 samples");
    constexpr int ROUNDS = NS / 512;
    const int odd = (int)((B0 - src_abs0) & 1);
    const float4 *src = reinterpret_cast<const float4 *>(iq) + ((B0 - odd - src_abs0) >> 1);
    const int i0 = 2 * tid - odd, i1 = i0 + 1;                        // samples of the thread's piece in round 0 (i0 = -1: not wanted)
    float *const e0 = E + (i0 >> 5) * AM_XROW + (i0 & 31);            // (arithmetic shift: -1 -> row -1, column 31: right from round 1 on)
    float *const e1 = E + (i1 >> 5) * AM_XROW + (i1 & 31);
    float4 v[ROUNDS], vx;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) v[r] = src[tid + 256 * r];
    vx.x = vx.y = vx.z = vx.w = 0.0f;
    if (odd && tid == 0) vx = src[NS / 2];                            // (the last sample's piece)
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const float a = v[r].x * v[r].x, b = v[r].y * v[r].y, c = v[r].z * v[r].z, d = v[r].w * v[r].w;
        if (r > 0 || i0 >= 0) e0[r * 16 * AM_XROW] = a + b;           // a1: fl(fl(I*I) + fl(Q*Q))
        e1[r * 16 * AM_XROW] = c + d;
    }
    if (odd && tid == 0) { const float a = vx.x * vx.x, b = vx.y * vx.y; E[((NS - 1) >> 5) * AM_XROW + 31] = a + b; }
}
// soft chip of the window in LDS row `row` (bb; the reference level comes off later): ii = offset of the burst's first sample
// inside its canonical chip, the same for all of a burst's windows
__device__ __forceinline__ float am_soft_chip_row32(const float *E, int row, int ii, float s1)
{
    constexpr int SPC = 32;
    float m[SPC];
    const float4 *rp = reinterpret_cast<const float4 *>(E + row * AM_XROW);
#pragma unroll
    for (int k = 0; k < SPC / 4; ++k) { const float4 u = rp[k]; m[4 * k] = u.x; m[4 * k + 1] = u.y; m[4 * k + 2] = u.z; m[4 * k + 3] = u.w; }
    const int wb = SPC - 1 - ii;                                      // window index of the chip's first sample
    float pre = 0.0f, suf = 0.0f;
#pragma unroll
    for (int w = 0; w < SPC; ++w) {
        if (w >= wb) pre = pre + m[w];                                // (uniform conditions)
        if (SPC - 1 - w < wb) suf = suf + m[SPC - 1 - w];
    }
    return (ii == SPC - 1) ? pre * s1 : (suf + pre) * s1;
}

#ifndef AM_XS_WPS
#define AM_XS_WPS 8                       /* waves per SIMD the extraction kernel is compiled for (64 VGPRs, no spills: eight workgroups per CU, 2 048 hits in flight) */
#endif
template <int SPC>
__global__ void __launch_bounds__(256, AM_XS_WPS)
am_k_extract_slice_iq(const float *__restrict__ iq, long long src_abs0, long long src_abs1, int use_pmf, float s1,
                      const float *__restrict__ inavg, const uint4 *__restrict__ emit_idx,
                      const uint32_t *__restrict__ n_ptr, const uint32_t *__restrict__ pos,
                      const uint32_t *__restrict__ eo, uint64_t base_abs, uint64_t rate,
                      const am_time_tag *__restrict__ tt, uint32_t ntt, float *__restrict__ bursts_out,
                      am_tag *__restrict__ tags_out, const uint32_t *__restrict__ crc_pow,
                      am_packet *__restrict__ packets, const uint32_t *__restrict__ scalars,
                      uint32_t *__restrict__ host_out, const uint32_t *__restrict__ Mp)
{
    // (any SPC: a lane's window of SPC samples starts at either parity -- SPC / 2 + 1 16-byte loads cover it; 1 sample per chip:
    // no filter, the launcher passes use_pmf = 0)
    constexpr int HEAD = 128;                                 // chips of a short packet: 16 + 2 * 56
    __shared__ float sb[AM_BURST];
    __shared__ float stg[SPC == 32 ? HEAD * AM_XROW : 1];     // 64 Msps: |.|^2 of 128 windows (am_stage_energies32)
    const int tid = threadIdx.x, lane = tid & (AM_WAVE - 1);
#if defined(AM_XPROF)
    long long xlast = (long long)__builtin_readcyclecounter();
    const long long xstart = (long long)__builtin_amdgcn_s_memrealtime();
    long long xhits = 0, xlong = 0;
#endif
    const uint32_t nhit = *n_ptr;                             // device-side hit count
    if (host_out && blockIdx.x == 0 && threadIdx.x == 0) {
        host_out[0] = nhit;
        host_out[1] = scalars[0];
        host_out[2] = Mp ? *Mp : 0u;
        host_out[5] = scalars[9];                             // a chained scan of this step gave up (am_chain_prefix)
    }
    const bool pmf = use_pmf != 0;
    const bool wide = (reinterpret_cast<uintptr_t>(iq) & 15u) == 0;
    // (Giving every XCD one contiguous eighth of the hits, so that the overlapping windows of neighbouring hits meet
    // in one L2, changed nothing: 56.3 against 56.7 us.  Budgeting the 64 Msps instantiation for six workgroups per CU
    // instead of five -- 80 VGPRs, 9 of them spilled -- 59 against 55 us, and 30-37 against 27 at 2 000 bursts/s.  Ten workgroups of two waves per CU instead of five of four --
    // twice as many hits in flight, the second 112 chips by the same threads: 62.8 us.)
    // (Taking a hit's candidate index, refined position, reference level and first-stage position one hit AHEAD, so that
    // those two dependent round trips ride along with the current hit's sample loads, changed nothing: 54.8 against 54.7 us.)
    // (A wave priority that falls with the turn of this loop, as in the streaming front ends -- am_fe_stream.h -- changed nothing here:
    // 37.7 against 37.6 us, profiles/r5_prio.  Eight workgroups of ~10 short turns each do not run apart the way six long-lived ones do.)
    for (uint32_t i = blockIdx.x; i < nhit; i += gridDim.x) {                 // (uniform)
        const uint4 rec = emit_idx[i];                                        // one load per hit (am_k_cblk_mark filed everything)
        const uint32_t e = rec.z;
        const float av = __uint_as_float(rec.w);
        const long long ae = (long long)base_abs + (long long)e;             // absolute index of the burst's first sample
        // (uniform) every lane's loads inside the source: all but the hits at the two ends of the stream
        const bool inside = pmf && wide && ae - SPC >= src_abs0 && ae + (long long)(AM_BURST - 1) * SPC + 2 < src_abs1;
#if defined(AM_XPROF)
        const long long xh0 = (long long)__builtin_readcyclecounter();
#endif
        AM_XSTAMP(0);
        const bool staged = SPC == 32 && inside;                              // (uniform) coalesced loads through LDS
        const int ii = (int)(ae % SPC);
        if constexpr (SPC == 32) {
            if (staged) {
                am_stage_energies32<HEAD * 32>(iq, src_abs0, ae - (SPC - 1), stg, tid);
                __syncthreads();
            }
        }
        if (tid < HEAD) {
            float v;
            if (staged) v = am_soft_chip_row32(stg, tid, ii, s1) - av;
            else v = am_soft_chip_iq<SPC>(iq, src_abs0, src_abs1, pmf, s1, ae, tid, inside) - av;   // preamble_impl.cc:219-221
            sb[tid] = v;
            if (bursts_out) bursts_out[(size_t)i * AM_BURST + tid] = v;
        }
        __syncthreads();
        AM_XSTAMP(1);
        const bool all = bursts_out != nullptr || am_burst_is_long(sb);       // (uniform)
        if constexpr (SPC == 32) {
            if (staged && all) {                                              // (the rows' readers are behind the barrier above)
                am_stage_energies32<(AM_BURST - HEAD) * 32>(iq, src_abs0, ae - (SPC - 1) + (long long)HEAD * SPC, stg, tid);
                __syncthreads();
            }
        }
        if (tid >= HEAD && tid < AM_BURST) {
            float v = 0.0f;                                                   // (never looked at in a short packet)
            if (all) {
                if (staged) v = am_soft_chip_row32(stg, tid - HEAD, ii, s1) - av;
                else v = am_soft_chip_iq<SPC>(iq, src_abs0, src_abs1, pmf, s1, ae, tid, inside) - av;
                if (bursts_out) bursts_out[(size_t)i * AM_BURST + tid] = v;
            }
            sb[tid] = v;
        }
        __syncthreads();
        AM_XSTAMP(2);
        if (tid < AM_WAVE) {
            am_tag t = am_make_tag(base_abs + e + (uint64_t)(2 * SPC - 1), rate, tt, ntt);
            t.inavg = av;
            t.how_late = e - rec.y;
            if (tags_out && tid == 0) tags_out[i] = t;
            am_slice_wave(sb, t, i, lane, crc_pow, packets);
        }
        __syncthreads();                                          // (sb is rewritten by the next hit)
        AM_XSTAMP(3);
#if defined(AM_XPROF)
        { const long long d = (long long)__builtin_readcyclecounter() - xh0; if (d > xlong) xlong = d; ++xhits; }
#endif
    }
#if defined(AM_XPROF)
    if (tid == 0 && blockIdx.x < 4096) {
        am_xprof_log[blockIdx.x][0] = xstart; am_xprof_log[blockIdx.x][1] = (long long)__builtin_amdgcn_s_memrealtime();
        am_xprof_log[blockIdx.x][2] = xhits; am_xprof_log[blockIdx.x][3] = xlong;
    }
#endif
}

hipError_t am_launch_extract_slice_iq(const float *iq, long long src_abs0, long long src_abs1, int use_pmf, float s1,
                                      const float *inavg, int spc, const uint4 *emit_idx, const uint32_t *n_ptr,
                                      uint32_t n_max, const uint32_t *pos, const uint32_t *e, uint64_t base_abs,
                                      uint64_t rate, const am_time_tag *tt, uint32_t ntt, float *bursts_out,
                                      am_tag *tags_out, const uint32_t *crc_pow, am_packet *packets,
                                      const uint32_t *scalars, uint32_t *host_out, hipStream_t s, const uint32_t *Mp)
{
    if (n_max == 0) return hipSuccess;
    // workgroups of 256 threads per CU: what the instantiation's registers allow (five at 32 samples per chip, where a lane
    // holds a 34-sample window; eight at one or two samples per chip, where the kernel is a chain of memory round trips per
    // hit and more hits in flight is all that helps), asked of the runtime once per device and instantiation
    static std::atomic<int> per_cu[64][9];
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto resident_for = [&](const void *kernel, int slot) -> uint32_t {
        int w = (dev >= 0 && dev < 64) ? per_cu[dev][slot].load(std::memory_order_acquire) : 0;
        if (w <= 0) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&w, kernel, 256, 0) != hipSuccess || w <= 0) w = 5;
            w = w > 8 ? 8 : w;
            if (dev >= 0 && dev < 64) per_cu[dev][slot].store(w, std::memory_order_release);
        }
        return (uint32_t)w * (uint32_t)am_device_cus();
    };
#define AM_XS_IQ(S, SLOT)                                                                                                 \
    do {                                                                                                                  \
        const uint32_t resident = resident_for(reinterpret_cast<const void *>(&am_k_extract_slice_iq<S>), SLOT);         \
        const uint32_t grid = n_max < resident ? n_max : resident;                                                        \
        hipLaunchKernelGGL((am_k_extract_slice_iq<S>), dim3(grid), dim3(256), 0, s, iq, src_abs0, src_abs1, use_pmf, s1, \
                           inavg, emit_idx, n_ptr, pos, e, base_abs, rate, tt, ntt, bursts_out, tags_out, crc_pow,      \
                           packets, scalars, host_out, Mp);                                                              \
    } while (0)
    if (spc == 1) use_pmf = 0;                               // (a one-sample window is the sample itself: s1 = 1)
    switch (spc) {                                           // (the rates the streaming front ends serve)
    case 32: AM_XS_IQ(32, 0); break;
    case 20: AM_XS_IQ(20, 1); break;
    case 16: AM_XS_IQ(16, 2); break;
    case 10: AM_XS_IQ(10, 3); break;
    case 8: AM_XS_IQ(8, 4); break;
    case 4: AM_XS_IQ(4, 5); break;
    case 5: AM_XS_IQ(5, 8); break;
    case 2: AM_XS_IQ(2, 6); break;
    case 1: AM_XS_IQ(1, 7); break;
    default: return hipErrorInvalidValue;
    }
#undef AM_XS_IQ
    return hipGetLastError();
}

// Completion ticket: the last launch of a scan.  The host polls the pinned word instead of asking the
// runtime (whose completion path costs tens of microseconds per scan).
// (count_src / count_dst: a device-side count to hand to the host along with the ticket, or null)
// (Round 4: the ticket handed out by the last workgroup of the extraction kernel instead -- every workgroup counts itself
// done with an atomic, the last one stores the ticket.  The count has to be ordered behind the workgroup's packet stores: with
// a system-scope fence per workgroup the extraction kernel went 55 -> 111 us, with a device-scope one 55 -> 159 us -- on this
// part either one writes back an XCD's L2, 1 300 times over.  The launch of its own, 4.2 us + the gap, stays.)
// (hipStreamWriteValue32 in its place -- a queue packet instead of a dispatch -- measured the same step time, 64 and 2 Msps.)
__global__ void am_k_ticket(uint32_t *host_word, uint32_t seq, const uint32_t *count_src, uint32_t *count_dst,
                            const unsigned long long *word_src, unsigned long long *word_dst)
{
    if (count_src) *count_dst = *count_src;
    if (word_src) *word_dst = *word_src;                      // (am_spipe: where the scan left the chunk, for the host's books)
    __hip_atomic_store(host_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t am_launch_ticket(uint32_t *host_word, uint32_t seq, hipStream_t s, const uint32_t *count_src,
                            uint32_t *count_dst, const uint64_t *word_src, uint64_t *word_dst)
{
    hipLaunchKernelGGL(am_k_ticket, dim3(1), dim3(1), 0, s, host_word, seq, count_src, count_dst,
                       reinterpret_cast<const unsigned long long *>(word_src), reinterpret_cast<unsigned long long *>(word_dst));
    return hipGetLastError();
}

hipError_t am_launch_slice(const float *bursts, const am_tag *tags, const uint32_t *n_ptr, uint32_t n_max,
                           const uint32_t *crc_pow, am_packet *packets, const uint32_t *scalars,
                           uint32_t *host_out, hipStream_t s, const uint32_t *Mp)
{
    if (n_max == 0) return hipSuccess;
    hipLaunchKernelGGL(am_k_slice, dim3(am_grid(n_max, 4)), dim3(256), 0, s, bursts, tags, n_ptr, crc_pow, packets,
                       scalars, host_out, Mp);
    return hipGetLastError();
}
