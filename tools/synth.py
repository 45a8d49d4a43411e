"""Seeded synthetic Mode-S captures (the reference ships no recording -- SURVEY.md 8d).

Workload generator for tests/ and bench.py; not part of the demodulator itself.

Burst model: 1090ES pulse-position modulation, 0.5 us chips, preamble pulses at chips
{0,2,7,9}, 56/112 data bits from chip 16 (bit 1 = pulse in the first half of the bit
period).  Each burst gets its own amplitude (uniform in dB SNR), carrier phase, carrier
frequency offset and fractional-sample start (linear-interpolated edges).  Background is
complex AWGN.  numpy PCG64, fully determined by (seed, arguments).
"""
import numpy as np

CRC_POLY = 0xFFF409


def _crc24(data: bytes) -> int:
    reg = 0
    for byte in data:
        for b in range(7, -1, -1):
            top = (reg >> 23) & 1
            reg = (reg << 1) & 0xFFFFFF
            if top ^ ((byte >> b) & 1):
                reg ^= CRC_POLY
    return reg


def make_frame(rng, df: int) -> bytes:
    """A frame of downlink format `df` with a parity field the reference accepts:
    DF11/DF17 carry a clean PI (syndrome 0); the others overlay a random ICAO address."""
    long_ = df in (16, 17, 18, 19, 20, 21) or df >= 24          # on the air (the reference slices 18 / 19 / 24.. as short)
    nbytes = 14 if long_ else 7
    body = bytearray(rng.integers(0, 256, nbytes - 3, dtype=np.uint8).tobytes())
    body[0] = ((df & 0x1F) << 3) | (body[0] & 0x07)
    par = _crc24(bytes(body))
    if df not in (11, 17):
        par ^= int(rng.integers(1, 1 << 24))
    return bytes(body) + par.to_bytes(3, "big")


def frame_chips(frame: bytes) -> np.ndarray:
    """0/1 chip sequence (2 Mchip/s) of one burst: 16 preamble chips + 2 per bit."""
    bits = np.unpackbits(np.frombuffer(frame, np.uint8))
    chips = np.zeros(16 + 2 * bits.size, np.float32)
    chips[[0, 2, 7, 9]] = 1.0
    chips[16 + 2 * np.arange(bits.size) + (1 - bits)] = 1.0
    return chips


DF_MIX = ((17, 0.60), (11, 0.15), (0, 0.05), (4, 0.05), (5, 0.05), (20, 0.05), (21, 0.05))
# the formats whose LENGTH the reference decides oddly (lib/slicer_impl.cc:140: long iff DF in {16, 17, 20, 21}): DF16 is
# sliced as 112 bits, DF18 / DF19 / DF24 -- long on the air -- as 56; for the tests that aim at the framer (the seeded captures
# of bench.py and of the golden files keep DF_MIX)
DF_MIX_EDGE = ((17, 0.20), (11, 0.15), (16, 0.15), (18, 0.15), (19, 0.10), (24, 0.10), (0, 0.05), (20, 0.05), (21, 0.05))


def synth_capture(rate, n, lam, seed, sigma=0.01, snr_db=(10.0, 35.0), cfo_hz=50e3,
                  overlap_frac=0.02, df_mix=DF_MIX, dtype=np.complex64):
    """n complex samples at `rate` (multiple of 2 MHz) with Poisson(lam per second) bursts.

    Returns (iq complex64[n], truth) where truth is a list of dicts
    {start (float sample), frame (hex), snr_db}."""
    rng = np.random.default_rng(np.random.PCG64(seed))
    spc = int(round(rate / 2e6))
    whole = spc >= 1 and abs(rate - 2e6 * spc) < 1e-6       # whole samples per chip (else: the chips are area-sampled)
    spcf = rate / 2e6
    assert spcf >= 1.0, "rate must be at least 2 MHz"
    iq = np.empty(n, np.complex64)
    v = iq.view(np.float32)
    chunk = 1 << 24
    for o in range(0, 2 * n, chunk):
        e = min(o + chunk, 2 * n)
        v[o:e] = rng.standard_normal(e - o, dtype=np.float32) * np.float32(sigma)

    dur = n / rate
    nb = int(rng.poisson(lam * dur)) if lam > 0 else 0
    starts = np.sort(rng.uniform(0, max(n - 1, 1), nb))
    n_ov = int(round(overlap_frac * nb))
    if n_ov:
        # deliberate collisions: a second burst starting inside an earlier one
        pick = rng.choice(nb, n_ov, replace=False)
        starts = np.sort(np.concatenate([starts, starts[pick] + rng.uniform(4 * spc, 200 * spc, n_ov)]))
    dfs = np.array([d for d, _ in df_mix])
    probs = np.array([p for _, p in df_mix], np.float64)
    probs /= probs.sum()
    truth = []
    for t0 in starts:
        df = int(rng.choice(dfs, p=probs))
        frame = make_frame(rng, df)
        snr = float(rng.uniform(*snr_db))
        amp = np.sqrt(2.0) * sigma * 10.0 ** (snr / 20.0)
        phase = float(rng.uniform(0, 2 * np.pi))
        cfo = float(rng.uniform(-cfo_hz, cfo_hz))
        i0 = int(np.floor(t0))
        fr = float(t0 - i0)
        if whole:
            env = np.repeat(frame_chips(frame), spc)
            env = np.concatenate([env, [0.0]]) * (1.0 - fr) + np.concatenate([[0.0], env]) * fr
        else:
            # a rate that is not a multiple of 2 MHz: sample m of the burst covers the chip times [(m - fr) / spcf,
            # (m + 1 - fr) / spcf); its amplitude is the mean of the chip waveform over that interval
            chips = frame_chips(frame).astype(np.float64)
            cum = np.concatenate([[0.0], np.cumsum(chips)])           # integral of the waveform up to whole chips
            m = np.arange(int(np.ceil(chips.size * spcf)) + 2, dtype=np.float64)
            def integral(x):
                x = np.clip(x, 0.0, float(chips.size))
                q = np.minimum(x.astype(np.int64), chips.size - 1)
                return cum[q] + chips[q] * (x - q)
            env = (integral((m + 1.0 - fr) / spcf) - integral((m - fr) / spcf)) * spcf
        i1 = min(i0 + env.size, n)
        if i1 <= i0:
            continue
        k = np.arange(i0, i1)
        rot = np.exp(1j * (2 * np.pi * cfo * (k - i0) / rate + phase))
        iq[i0:i1] += (amp * env[:i1 - i0] * rot).astype(np.complex64)
        truth.append({"start": float(t0), "frame": frame.hex(), "snr_db": snr})
    return iq.astype(dtype, copy=False), truth


# The configurations SURVEY.md 8(d) / BASELINE.md 3 name.
CONFIGS = {
    # name: (rate, seconds, lambda per second, seed)
    "2msps": (2e6, 10.0, 500.0, 1090),
    "64msps": (64e6, 1.0, 20000.0, 6400),
    "20msps": (20e6, 1.0, 5000.0, 2000),
}


def config_capture(name, seed_offset=0, seconds=None):
    rate, secs, lam, seed = CONFIGS[name]
    if seconds is not None:
        secs = seconds
    n = int(round(rate * secs))
    return rate, synth_capture(rate, n, lam, seed + seed_offset)
