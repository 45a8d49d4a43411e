"""Arbitrary-ratio resampler in front of the receive path for input below 4 Msps.

The reference's radio front end resamples anything slower than 4 Msps to 4 Msps before rx_path
(python/radio.py:49-53: pfb.arb_resampler_ccf(4e6 / rate), GNU Radio's polyphase arbitrary resampler with
its default taps).  GNU Radio is not part of this tree and its tap design is not reproducible here: PARITY UNPINNED by
construction -- against the reference this stage is compared by packet recall, not bits.  This module is the
DEFINITION of the stage: `arb_resampler` (numpy) writes the arithmetic out operation by operation, in a fixed order,
one IEEE double rounding each; `gpu_resampler` is the same thing on the GPU (csrc/am_resample.hip, C ABI
am_resampler_*), bit for bit (tests/test_resample.py), and what modes_rx uses: its output stays on the device and goes
straight into the receive path.

Design: 32-phase polyphase interpolator, prototype = Kaiser-windowed sinc (beta 5.0, 8 taps per phase,
cut-off at 0.6 of the input rate: a Mode-S pulse at 2 Msps is ONE sample wide, its spectrum reaches the input's
Nyquist frequency; a cut-off below it smears the pulses into the quiet zones the preamble test checks and costs
packets -- measured on a seeded capture: cut-off 0.45: 159 of the 289 frames the direct 2 Msps path decodes,
0.5: 233, 0.6: 284; GNU Radio's default taps for this block also pass up to 0.4 and stop at 0.6),
output sample m taken at input time m / ratio from the two neighbouring
phases with linear interpolation between them (what pfb.arb_resampler does with its derivative filter, to
first order).  Unity DC gain.  Streaming: the last taps-per-phase input samples and the fractional read
position are carried from one call to the next, so chunking does not change the output.
"""
import numpy as np

NPHASE = 32
TAPS_PER_PHASE = 8


def design_taps(nphase=NPHASE, per_phase=TAPS_PER_PHASE, cutoff=0.6, beta=5.0):
    """Prototype low-pass at nphase x the input rate, split into phases: taps[p, k]."""
    n = nphase * per_phase
    t = (np.arange(n) - (n - 1) / 2.0) / nphase               # in input samples
    h = 2 * cutoff * np.sinc(2 * cutoff * t) * np.kaiser(n, beta)
    h *= nphase / h.sum()                                      # unity gain per phase
    return np.ascontiguousarray(h.reshape(per_phase, nphase).T.astype(np.float64))   # [phase, tap]


class arb_resampler(object):
    """y = resample(x, ratio) with ratio = f_out / f_in >= 1 (the reference only interpolates: radio.py:49).

    Output sample m is the input signal at time m / ratio, delayed by `delay` input samples (the prototype's
    group delay): y = sum_q x[i - q] * taps[p][q] with i + p / 32 the read position, linearly interpolated
    towards the next phase."""

    def __init__(self, ratio):
        if not ratio >= 1.0:
            raise ValueError("ratio must be >= 1 (interpolation)")
        self.ratio = float(ratio)
        self.taps = design_taps()                              # [phase, tap]; tap q applies to the sample q behind the newest
        self._hist = np.zeros(TAPS_PER_PHASE, np.complex128)   # the last T input samples of the previous calls
        self._pos = 0.0                                        # read position of the next output relative to the next call's x[0]
        self.delay = (NPHASE * TAPS_PER_PHASE - 1) / (2.0 * NPHASE)

    def work(self, x):
        x = np.asarray(x)
        if x.size > (1 << 17):                                 # (bounded temporaries; the carried state makes this exact)
            return np.concatenate([self.work(x[o:o + (1 << 17)]) for o in range(0, x.size, 1 << 17)])
        x = x.astype(np.complex128)
        T = TAPS_PER_PHASE
        n_in = x.size
        if n_in == 0:
            return np.zeros(0, np.complex64)
        buf = np.concatenate([self._hist, x])                  # x[j] = buf[T + j]
        step = 1.0 / self.ratio
        # outputs that need nothing beyond x[n_in - 1] (the next phase of the last one may need x[floor(t) + 1])
        span = (n_in - 1) - self._pos
        m_cnt = int(np.ceil(span / step)) if span > 0 else 0   # t = pos + m step < n_in - 1
        if m_cnt > 0:
            t = self._pos + step * np.arange(m_cnt)
            i0 = np.floor(t).astype(np.int64)
            frac = (t - i0) * NPHASE
            p = np.minimum(np.floor(frac).astype(np.int64), NPHASE - 1)
            a = frac - p
            wrap = (p + 1) >= NPHASE
            p1 = np.where(wrap, 0, p + 1)
            re, im = np.ascontiguousarray(buf.real), np.ascontiguousarray(buf.imag)
            y0r = np.zeros(m_cnt); y0i = np.zeros(m_cnt); y1r = np.zeros(m_cnt); y1i = np.zeros(m_cnt)
            for q in range(T):                                 # tap by tap, in this order: product, then sum (two roundings)
                k0 = T + i0 - q
                k1 = k0 + wrap
                c0, c1 = self.taps[p, q], self.taps[p1, q]
                y0r = y0r + re[k0] * c0
                y0i = y0i + im[k0] * c0
                y1r = y1r + re[k1] * c1
                y1i = y1i + im[k1] * c1
            b = 1.0 - a
            y = np.empty(m_cnt, np.complex64)
            y.real = (b * y0r + a * y1r).astype(np.float32)
            y.imag = (b * y0i + a * y1i).astype(np.float32)
            self._pos = t[-1] + step - n_in
        else:
            y = np.zeros(0, np.complex64)
            self._pos -= n_in
        self._hist = buf[-T:]
        return y


class gpu_resampler(object):
    """arb_resampler on the GPU (csrc/am_resample.hip), bit-identical to it.  work() returns host samples;
    work_device() leaves them on the device: (device pointer to interleaved float32 I,Q, number of complex samples),
    valid until the next call -- what rx_path.work_device / am_process_iq(AM_F_DEVICE_IN) take."""

    def __init__(self, ratio, device=-1, lib=None):
        import ctypes as C
        from . import _capi
        self._C, self._capi = C, _capi
        self.lib = lib or _capi.default_library()
        L = self.lib.L
        vp, u64 = C.c_void_p, C.c_uint64
        L.am_resampler_create.restype = vp
        L.am_resampler_create.argtypes = [C.c_int, C.c_double, vp, C.POINTER(C.c_int)]
        L.am_resampler_destroy.argtypes = [vp]
        L.am_resampler_reset.argtypes = [vp]
        L.am_resampler_work.argtypes = [vp, vp, u64, C.c_uint32, vp, u64, C.POINTER(u64)]
        L.am_resampler_device_output.restype = vp
        L.am_resampler_device_output.argtypes = [vp]
        L.am_resampler_last_error.restype = C.c_char_p
        L.am_resampler_last_error.argtypes = [vp]
        self.ratio = float(ratio)
        self.taps = design_taps()
        self.delay = (NPHASE * TAPS_PER_PHASE - 1) / (2.0 * NPHASE)
        err = C.c_int(0)
        self._h = L.am_resampler_create(int(device), self.ratio, self.taps.ctypes.data, C.byref(err))
        if not self._h:
            raise _capi.AirModesError(err.value, "am_resampler_create failed (no HIP device?)")

    def close(self):
        if getattr(self, "_h", None):
            self.lib.L.am_resampler_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise self._capi.AirModesError(rc, self.lib.L.am_resampler_last_error(self._h).decode())

    def _run(self, x, out_ptr, cap):
        f = self._capi._iq_f32(x)
        n = f.size // 2
        got = self._C.c_uint64(0)
        self._chk(self.lib.L.am_resampler_work(self._h, f.ctypes.data if n else None, n, 0, out_ptr, cap, self._C.byref(got)))
        return int(got.value)

    def work(self, x):
        n = np.asarray(x).size
        out = np.zeros(int(n * self.ratio) + 16, np.complex64)
        m = self._run(x, out.ctypes.data, out.size)
        return out[:m].copy()

    def work_device(self, x):
        m = self._run(x, None, 0)
        return int(self.lib.L.am_resampler_device_output(self._h) or 0), m

    def work_device_in(self, dev_ptr, n_complex):
        """Input already on the device (interleaved float32 at dev_ptr); output left on the device as in work_device."""
        got = self._C.c_uint64(0)
        self._chk(self.lib.L.am_resampler_work(self._h, int(dev_ptr) if n_complex else None, int(n_complex),
                                               self._capi.AM_F_DEVICE_IN, None, 0, self._C.byref(got)))
        return int(self.lib.L.am_resampler_device_output(self._h) or 0), int(got.value)
