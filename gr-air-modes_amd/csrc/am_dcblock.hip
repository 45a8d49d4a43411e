// am_dcblock.hip -- a2: the optional DC blocker in front of the path (python/rx_path.py:39-41,
// filter.dc_blocker_cc(100*spc, False)).  GNU Radio 3.8's gr-filter is not under /root/reference:
// parity unpinned; the published linear-phase form is
//     m1 = MA_D(x),  m2 = MA_D(m1),  y[n] = x[n - (D-1)] - m2[n],   D = 100*spc,  MA = sum / (float)D,
// with x and m1 zero before the stream.  GNU Radio accumulates its window sums recursively (rounding
// depends on the whole history); here the order is the chip-aligned two-level order of DESIGN.md
// section 3 with blocks of 100 chips, I and Q as two real streams -- the same definition as
// oracle/airmodes_oracle.c:amo_dcblock, bit for bit.  Off by default (python/radio.py:118); two
// extra passes over the samples when on.
#include "am_internal.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#define DC_THREADS 256
#define DC_PIDX(i) ((i) + ((i) >> 5))
#define DC_PADN(n) ((n) + ((n) >> 5) + 1)

struct am_ma_args {
    const float *in;                 // interleaved complex; zero outside [in_abs0, in_abs1)
    long long in_abs0, in_abs1;
    float *out;                      // interleaved complex, out[0] is sample out_abs0
    long long out_abs0, out_n;
    const float *xdel;               // null: out = sum / D;  else out = xdel[n - (D-1)] - sum / D
    long long xdel_abs0, xdel_abs1;
    int spc, chips, tile;            // tile: outputs per workgroup, a multiple of D = chips*spc
    float divisor;
};

static size_t dc_lds_bytes(int spc, int chips, int tile)
{
    const int W = chips * spc + tile;
    return ((size_t)2 * DC_PADN(W) + (size_t)3 * (W / spc + 2)) * sizeof(float);
}

// One workgroup = `tile` outputs of one component (blockIdx.y), aligned to the absolute block grid.
__global__ void __launch_bounds__(DC_THREADS) am_k_mavg(am_ma_args a)
{
    HIP_DYNAMIC_SHARED(float, smem);
    const int spc = a.spc, CH = a.chips;
    const int D = CH * spc;
    const int T = a.tile;
    const int W = D + T;                       // left halo: one block
    const int nch = W / spc;
    float *X = smem;                           // inputs, then unchanged
    float *S = X + DC_PADN(W);                 // in-chip suffix sums
    float *TOT = S + DC_PADN(W);
    float *PT = TOT + (nch + 1);
    float *ST = PT + (nch + 1);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int comp = blockIdx.y;
    const long long first_blk = (a.out_abs0 / D) * (long long)D;
    const long long tb = first_blk + (long long)blockIdx.x * T;      // first output of this tile
    const long long x0 = tb - D;

    for (int i = tid; i < W; i += nt) {
        const long long n = x0 + i;
        X[DC_PIDX(i)] = (n >= a.in_abs0 && n < a.in_abs1) ? a.in[2 * (n - a.in_abs0) + comp] : 0.0f;
    }
    __syncthreads();
    // chip totals (left->right) and in-chip suffix sums (right->left)
    for (int q = tid; q < nch; q += nt) {
        const int b = q * spc;
        float acc = 0.0f;
        for (int i = 0; i < spc; ++i) acc = acc + X[DC_PIDX(b + i)];
        TOT[q] = acc;
        acc = 0.0f;
        for (int i = spc - 1; i >= 0; --i) { acc = acc + X[DC_PIDX(b + i)]; S[DC_PIDX(b + i)] = acc; }
    }
    __syncthreads();
    // per block: exclusive prefix / suffix of chip totals (sequential, canonical order)
    const int nblk = W / D;
    for (int idx = tid; idx < 2 * nblk; idx += nt) {
        const int qb = CH * (idx >> 1);
        float acc = 0.0f;
        if (idx & 1) {
            for (int j = CH - 1; j >= 0; --j) { ST[qb + j] = acc; acc = acc + TOT[qb + j]; }
        } else {
            for (int j = 0; j < CH; ++j) { PT[qb + j] = acc; acc = acc + TOT[qb + j]; }
        }
    }
    __syncthreads();
    // window sums for the tile's chips (chip CH is the first chip of the tile)
    for (int q = CH + tid; q < nch; q += nt) {
        const int b = q * spc;
        const int j = q % CH;
        const float pt = PT[q];
        float acc = 0.0f;
        for (int i = 0; i < spc; ++i) {
            acc = acc + X[DC_PIDX(b + i)];
            const float PRE = pt + acc;
            const long long n = x0 + b + i;
            float s;
            if ((j == CH - 1 && i == spc - 1) || n + 1 < D) {
                s = PRE;                        // last sample of a block, or the window starts before sample 0
            } else {
                const int al = b + i - D + 1;
                const float SUF = S[DC_PIDX(al)] + ST[al / spc];
                s = SUF + PRE;
            }
            const long long o = n - a.out_abs0;
            if (o >= 0 && o < a.out_n) {
                float v = s / a.divisor;
                if (a.xdel) {
                    const long long k = n - (D - 1);
                    const float d = (k >= a.xdel_abs0 && k < a.xdel_abs1) ? a.xdel[2 * (k - a.xdel_abs0) + comp] : 0.0f;
                    v = d - v;
                }
                a.out[2 * o + comp] = v;
            }
        }
    }
}

static int dc_pick_tile(int spc, int chips)
{
    const int D = chips * spc;
    int k = 4096 / D;
    if (k < 1) k = 1;
    while (k > 1 && dc_lds_bytes(spc, chips, k * D) > 96 * 1024) --k;
    return k * D;
}

static hipError_t dc_launch(const am_ma_args &a, hipStream_t s)
{
    if (a.out_n <= 0) return hipSuccess;
    const size_t lds = dc_lds_bytes(a.spc, a.chips, a.tile);
    if (lds > 160 * 1024 - 256) return hipErrorInvalidValue;
    hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void *>(am_k_mavg),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc != hipSuccess) return rc;
    const long long D = (long long)a.chips * a.spc;
    const long long first_blk = (a.out_abs0 / D) * D;
    const unsigned grid = (unsigned)((a.out_abs0 + a.out_n - first_blk + a.tile - 1) / a.tile);
    hipLaunchKernelGGL(am_k_mavg, dim3(grid, 2), dim3(DC_THREADS), lds, s, a);
    return hipGetLastError();
}

// y[y_abs0 .. y_abs0 + y_n) from the raw samples [raw_abs0, raw_abs1) (which must reach back to
// max(0, y_abs0 - 2*(D-1))); m1 is scratch for y_n + D - 1 complex samples.
hipError_t am_launch_dcblock(const float *raw, long long raw_abs0, long long raw_abs1, long long y_abs0, long long y_n,
                             int spc, float *m1, float *y, hipStream_t s)
{
    if (y_n <= 0) return hipSuccess;
    const int chips = AM_DC_CHIPS;
    const long long D = (long long)chips * spc;
    const long long m_abs0 = y_abs0 > D - 1 ? y_abs0 - (D - 1) : 0;
    am_ma_args a;
    a.spc = spc; a.chips = chips; a.tile = dc_pick_tile(spc, chips); a.divisor = (float)D;
    a.in = raw; a.in_abs0 = raw_abs0; a.in_abs1 = raw_abs1;
    a.out = m1; a.out_abs0 = m_abs0; a.out_n = y_abs0 + y_n - m_abs0;
    a.xdel = nullptr; a.xdel_abs0 = a.xdel_abs1 = 0;
    hipError_t rc = dc_launch(a, s);
    if (rc != hipSuccess) return rc;
    a.in = m1; a.in_abs0 = m_abs0; a.in_abs1 = y_abs0 + y_n;
    a.out = y; a.out_abs0 = y_abs0; a.out_n = y_n;
    a.xdel = raw; a.xdel_abs0 = raw_abs0; a.xdel_abs1 = raw_abs1;
    return dc_launch(a, s);
}

// raw samples the blocker needs before the first output it is asked for
unsigned long long am_dcblock_history(int spc) { return 2ull * ((unsigned long long)AM_DC_CHIPS * spc - 1ull); }
