"""Arbitrary-ratio resampler in front of the receive path for input below 4 Msps.

The reference's radio front end resamples anything slower than 4 Msps to 4 Msps before rx_path
(python/radio.py:49-53: pfb.arb_resampler_ccf(4e6 / rate), GNU Radio's polyphase arbitrary resampler with
its default taps).  GNU Radio is not part of this tree, its tap design is not reproducible here, and this
stage sits in front of the hot path at a few Msps: it runs on the host (numpy), with documented taps of its
own.  PARITY UNPINNED by construction -- tests compare packet recall, not bits.

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
            q = np.arange(T)[None, :]
            win0 = buf[(T + i0)[:, None] - q]
            y0 = np.einsum("mq,mq->m", win0, self.taps[p])
            wrap = (p + 1) >= NPHASE
            win1 = buf[(T + i0 + wrap)[:, None] - q]
            y1 = np.einsum("mq,mq->m", win1, self.taps[np.where(wrap, 0, p + 1)])
            y = ((1.0 - a) * y0 + a * y1).astype(np.complex64)
            self._pos = t[-1] + step - n_in
        else:
            y = np.zeros(0, np.complex64)
            self._pos -= n_in
        self._hist = buf[-T:]
        return y
