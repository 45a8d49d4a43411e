// am_resample.hip -- polyphase arbitrary-ratio interpolator in front of the receive path.
//
// python/radio.py:49-53: input slower than 4 Msps goes through pfb.arb_resampler_ccf(4e6 / rate) before rx_path.
// GNU Radio's block and its tap design are not in the reference tree (PARITY UNPINNED); the DEFINITION of this stage
// is air_modes/resample.py (32 phases x 8 taps, Kaiser-windowed sinc, linear interpolation between neighbouring
// phases), whose arithmetic is written out operation by operation so that this kernel can repeat it bit for bit:
//
//   output m of a block:  t = pos + step * m          (double; step = 1 / ratio, pos = read position carried between calls)
//                         i0 = floor(t), frac = (t - i0) * 32, p = min(floor(frac), 31), a = frac - p
//                         y0 = sum over q = 0..7, in that order, of x[i0 - q] * taps[p][q]      (re and im apart, double)
//                         y1 = the same one phase on (p + 1 = 32: phase 0 of the next sample)
//                         y  = float((1 - a) * y0 + a * y1)
//
// every product and every sum one IEEE double rounding (contraction off).  The taps come from the caller (numpy's
// kaiser / sinc are not reproduced here).  The host part -- how many outputs a block yields, the carried position and
// the last 8 input samples -- repeats resample.py's scalar double arithmetic, including its blocks of 2^17 inputs.
#include "am_internal.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <new>

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#define AM_RS_NPHASE 32
#define AM_RS_TAPS 8
#define AM_RS_BLOCK (1u << 17)            /* inputs per block, as resample.py */

struct am_resampler {
    int device = 0;
    hipStream_t stream = nullptr;
    double ratio = 1.0, step = 1.0, pos = 0.0;
    float hist[2 * AM_RS_TAPS] = {};      // the last 8 input samples of the calls so far (I, Q)
    double *taps_dev = nullptr;           // [32][8]
    float *in_dev = nullptr, *out_dev = nullptr;
    size_t in_cap = 0, out_cap = 0;       // floats
    uint64_t out_n = 0;                   // complex samples in out_dev after the last call
    char err[160] = "";
};

// buf = 8 history samples + the block's inputs (x[j] = buf[8 + j]); one thread per output
__global__ void __launch_bounds__(256)
am_k_resample(const float2 *__restrict__ buf, const double *__restrict__ taps, double pos, double step, uint32_t m_cnt,
              float2 *__restrict__ out)
{
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= m_cnt) return;
    const double t = pos + step * (double)m;
    const double fl = floor(t);
    const long long i0 = (long long)fl;
    const double frac = (t - fl) * (double)AM_RS_NPHASE;
    int p = (int)floor(frac);
    if (p > AM_RS_NPHASE - 1) p = AM_RS_NPHASE - 1;
    const double a = frac - (double)p;
    const int wrap = (p + 1 >= AM_RS_NPHASE) ? 1 : 0;
    const double *t0 = taps + p * AM_RS_TAPS, *t1 = taps + (wrap ? 0 : p + 1) * AM_RS_TAPS;
    const float2 *w0 = buf + (AM_RS_TAPS + i0), *w1 = w0 + wrap;
    double y0r = 0.0, y0i = 0.0, y1r = 0.0, y1i = 0.0;
#pragma unroll
    for (int q = 0; q < AM_RS_TAPS; ++q) {
        const float2 u = w0[-q], v = w1[-q];
        const double c0 = t0[q], c1 = t1[q];
        y0r = y0r + (double)u.x * c0;
        y0i = y0i + (double)u.y * c0;
        y1r = y1r + (double)v.x * c1;
        y1i = y1i + (double)v.y * c1;
    }
    const double b = 1.0 - a;
    float2 y;
    y.x = (float)(b * y0r + a * y1r);
    y.y = (float)(b * y0i + a * y1i);
    out[m] = y;
}

namespace {

int rs_fail(am_resampler *h, int code, const char *what, hipError_t rc = hipSuccess)
{
    if (h) {
        if (rc != hipSuccess) snprintf(h->err, sizeof(h->err), "%s: %s", what, hipGetErrorString(rc));
        else snprintf(h->err, sizeof(h->err), "%s", what);
    }
    return code;
}

int rs_ensure(am_resampler *h, float **p, size_t *cap, size_t floats)
{
    if (*p && *cap >= floats) return AM_OK;
    float *q = nullptr;
    const size_t want = floats + floats / 4 + 1024;
    hipError_t rc = hipMalloc(reinterpret_cast<void **>(&q), want * sizeof(float));
    if (rc != hipSuccess) return rs_fail(h, AM_ENOMEM, "hipMalloc", rc);
    if (*p) {
        // (the output buffer keeps what earlier blocks of this call wrote)
        if (p == &h->out_dev && h->out_n) (void)hipMemcpyAsync(q, *p, h->out_n * 2 * sizeof(float), hipMemcpyDeviceToDevice, h->stream);
        (void)hipStreamSynchronize(h->stream);
        (void)hipFree(*p);
    }
    *p = q;
    *cap = want;
    return AM_OK;
}

} // namespace

extern "C" {

am_resampler *am_resampler_create(int device, double ratio, const double *taps, int *err)
{
    int code = AM_OK;
    am_resampler *h = nullptr;
    do {
        if (!(ratio >= 1.0) || !taps) { code = AM_EINVAL; break; }
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { code = AM_ENODEV; break; }
        h = new (std::nothrow) am_resampler();
        if (!h) { code = AM_ENOMEM; break; }
        if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
        if (device >= ndev) { code = AM_ENODEV; break; }
        h->device = device;
        h->ratio = ratio;
        h->step = 1.0 / ratio;
        if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&h->stream) != hipSuccess) { code = AM_EHIP; break; }
        if (hipMalloc(reinterpret_cast<void **>(&h->taps_dev), AM_RS_NPHASE * AM_RS_TAPS * sizeof(double)) != hipSuccess) { code = AM_ENOMEM; break; }
        if (hipMemcpy(h->taps_dev, taps, AM_RS_NPHASE * AM_RS_TAPS * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { code = AM_EHIP; break; }
    } while (0);
    if (code != AM_OK && h) { am_resampler_destroy(h); h = nullptr; }
    if (err) *err = code;
    return h;
}

void am_resampler_destroy(am_resampler *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->taps_dev) (void)hipFree(h->taps_dev);
    if (h->in_dev) (void)hipFree(h->in_dev);
    if (h->out_dev) (void)hipFree(h->out_dev);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char *am_resampler_last_error(const am_resampler *h) { return h ? h->err : "null resampler"; }
const float *am_resampler_device_output(const am_resampler *h) { return h ? h->out_dev : nullptr; }

int am_resampler_reset(am_resampler *h)
{
    if (!h) return AM_EINVAL;
    h->pos = 0.0;
    memset(h->hist, 0, sizeof(h->hist));
    h->out_n = 0;
    return AM_OK;
}

int am_resampler_work(am_resampler *h, const float *iq, uint64_t n, uint32_t flags, float *out, uint64_t cap, uint64_t *n_out)
{
    if (!h) return AM_EINVAL;
    if (n_out) *n_out = 0;
    h->out_n = 0;
    if (n && !iq) return rs_fail(h, AM_EINVAL, "null input");
    if (n > ((uint64_t)1 << 31)) return rs_fail(h, AM_EINVAL, "chunk larger than 2^31 samples");
    if (hipSetDevice(h->device) != hipSuccess) return rs_fail(h, AM_EHIP, "hipSetDevice");
    const bool dev_in = (flags & AM_F_DEVICE_IN) != 0;
    const hipMemcpyKind in_kind = dev_in ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    const double step = h->step;
    for (uint64_t o = 0; o < n; o += AM_RS_BLOCK) {
        const uint64_t n_in = (n - o < AM_RS_BLOCK) ? n - o : AM_RS_BLOCK;
        // buf = history + block
        if (int rc = rs_ensure(h, &h->in_dev, &h->in_cap, (size_t)(AM_RS_TAPS + n_in) * 2); rc != AM_OK) return rc;
        hipError_t e = hipMemcpyAsync(h->in_dev, h->hist, sizeof(h->hist), hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(h->in_dev + 2 * AM_RS_TAPS, iq + 2 * o, (size_t)n_in * 2 * sizeof(float), in_kind, h->stream);
        if (e != hipSuccess) return rs_fail(h, AM_EHIP, "hipMemcpyAsync", e);
        // outputs that need nothing beyond x[n_in - 1]: t = pos + m step < n_in - 1   (resample.py, same double arithmetic)
        const double span = (double)(n_in - 1) - h->pos;
        const uint64_t m_cnt = span > 0.0 ? (uint64_t)ceil(span / step) : 0;
        if (m_cnt) {
            if (int rc = rs_ensure(h, &h->out_dev, &h->out_cap, (size_t)(h->out_n + m_cnt) * 2); rc != AM_OK) return rc;
            hipLaunchKernelGGL(am_k_resample, dim3((unsigned)((m_cnt + 255) / 256)), dim3(256), 0, h->stream,
                               reinterpret_cast<const float2 *>(h->in_dev), h->taps_dev, h->pos, step, (uint32_t)m_cnt,
                               reinterpret_cast<float2 *>(h->out_dev) + h->out_n);
            if (hipGetLastError() != hipSuccess) return rs_fail(h, AM_EHIP, "am_k_resample launch");
            const double t_last = h->pos + step * (double)(m_cnt - 1);
            h->pos = t_last + step - (double)n_in;
            h->out_n += m_cnt;
        } else
            h->pos -= (double)n_in;
        // the block's last 8 samples (with the history when it is shorter) become the history
        float nh[2 * AM_RS_TAPS];                              // buf[-8:] (part history, part block when the block is shorter)
        e = hipMemcpyAsync(nh, h->in_dev + 2 * n_in, sizeof(nh), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) return rs_fail(h, AM_EHIP, "history copy", e);
        memcpy(h->hist, nh, sizeof(nh));
    }
    if (n_out) *n_out = h->out_n;
    if (out) {
        if (h->out_n > cap) return rs_fail(h, AM_ECAPACITY, "output array too small");
        if (h->out_n) {
            const hipMemcpyKind k = (flags & AM_F_DEVICE_OUT) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
            hipError_t e = hipMemcpyAsync(out, h->out_dev, (size_t)h->out_n * 2 * sizeof(float), k, h->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) return rs_fail(h, AM_EHIP, "output copy", e);
        }
    } else if (hipStreamSynchronize(h->stream) != hipSuccess)
        return rs_fail(h, AM_EHIP, "hipStreamSynchronize");
    return AM_OK;
}

} // extern "C"

/* ---- pinned staging: host samples on their way to the device, several buffers in flight --------------------------
 * A file (or socket) reader fills slot k's pinned buffer and starts its copy while the receive path still works on
 * slot k-1's samples on the device: the PCIe transfer of one chunk overlaps the scan of the one before it. */
#define AM_UP_MAXSLOTS 8
struct am_uploader {
    int device = 0, nslots = 0;
    uint64_t cap = 0;                     // complex samples per slot
    float *host[AM_UP_MAXSLOTS] = {};
    float *dev[AM_UP_MAXSLOTS] = {};
    hipEvent_t done[AM_UP_MAXSLOTS] = {};
    hipStream_t stream = nullptr;
};

extern "C" {

am_uploader *am_uploader_create(int device, uint64_t capacity_complex, int nslots, int *err)
{
    int code = AM_OK;
    am_uploader *u = nullptr;
    do {
        if (nslots < 1 || nslots > AM_UP_MAXSLOTS || capacity_complex == 0) { code = AM_EINVAL; break; }
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { code = AM_ENODEV; break; }
        u = new (std::nothrow) am_uploader();
        if (!u) { code = AM_ENOMEM; break; }
        if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
        if (device >= ndev) { code = AM_ENODEV; break; }
        u->device = device; u->nslots = nslots; u->cap = capacity_complex;
        if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&u->stream) != hipSuccess) { code = AM_EHIP; break; }
        const size_t bytes = (size_t)capacity_complex * 2 * sizeof(float);
        for (int k = 0; k < nslots && code == AM_OK; ++k) {
            if (hipHostMalloc(reinterpret_cast<void **>(&u->host[k]), bytes, hipHostMallocDefault) != hipSuccess) code = AM_ENOMEM;
            else if (hipMalloc(reinterpret_cast<void **>(&u->dev[k]), bytes) != hipSuccess) code = AM_ENOMEM;
            else if (hipEventCreateWithFlags(&u->done[k], hipEventDisableTiming) != hipSuccess) code = AM_EHIP;
        }
    } while (0);
    if (code != AM_OK && u) { am_uploader_destroy(u); u = nullptr; }
    if (err) *err = code;
    return u;
}

void am_uploader_destroy(am_uploader *u)
{
    if (!u) return;
    (void)hipSetDevice(u->device);
    if (u->stream) (void)hipStreamSynchronize(u->stream);
    for (int k = 0; k < AM_UP_MAXSLOTS; ++k) {
        if (u->host[k]) (void)hipHostFree(u->host[k]);
        if (u->dev[k]) (void)hipFree(u->dev[k]);
        if (u->done[k]) (void)hipEventDestroy(u->done[k]);
    }
    if (u->stream) (void)hipStreamDestroy(u->stream);
    delete u;
}

float *am_uploader_host(am_uploader *u, int slot) { return (u && slot >= 0 && slot < u->nslots) ? u->host[slot] : nullptr; }

int am_uploader_start(am_uploader *u, int slot, uint64_t n_complex)
{
    if (!u || slot < 0 || slot >= u->nslots || n_complex > u->cap) return AM_EINVAL;
    if (hipSetDevice(u->device) != hipSuccess) return AM_EHIP;
    if (n_complex && hipMemcpyAsync(u->dev[slot], u->host[slot], (size_t)n_complex * 2 * sizeof(float), hipMemcpyHostToDevice,
                                    u->stream) != hipSuccess)
        return AM_EHIP;
    return hipEventRecord(u->done[slot], u->stream) == hipSuccess ? AM_OK : AM_EHIP;
}

const float *am_uploader_wait(am_uploader *u, int slot)
{
    if (!u || slot < 0 || slot >= u->nslots) return nullptr;
    if (hipSetDevice(u->device) != hipSuccess || hipEventSynchronize(u->done[slot]) != hipSuccess) return nullptr;
    return u->dev[slot];
}

} // extern "C"
