"""input below 4 Msps is resampled to 4 Msps in front of the path (python/radio.py:49-53).  GNU
Radio's polyphase resampler and its taps are not available here, so this stage is PARITY-UNPINNED: the tests
check the interpolator itself (accuracy, chunk invariance) and compare packet RECALL with and without it."""
import io

import numpy as np
import pytest

import synth
from air_modes.resample import arb_resampler, gpu_resampler


@pytest.mark.parametrize("ratio", [2.0, 4e6 / 2.4e6, 1.25, 4e6 / 3.2e6])
def test_interpolator_accuracy_and_chunk_invariance(ratio):
    n, f = 30000, 0.07
    x = np.exp(2j * np.pi * f * np.arange(n)).astype(np.complex64)
    r = arb_resampler(ratio)
    y = r.work(x)
    m = np.arange(y.size)
    ref = np.exp(2j * np.pi * f * (m / ratio - r.delay))
    assert abs(y.size - n * ratio) <= 2
    assert np.abs(y[300:-300] - ref[300:-300]).max() < 5e-3          # in-band tone: amplitude and phase
    r2 = arb_resampler(ratio)
    y2 = np.concatenate([r2.work(c) for c in np.array_split(x, 41)] + [r2.work(x[:0])])
    assert y2.size == y.size and np.array_equal(y2, y)               # the carried state makes chunking invisible
    lo = np.exp(2j * np.pi * 0.15 * np.arange(n)).astype(np.complex64)
    img = arb_resampler(ratio).work(lo)                              # the image of the input spectrum (at 1 - 0.15) is suppressed
    spec = np.abs(np.fft.fft(img[1000:1000 + 8192] * np.hanning(8192)))
    k_sig = int(round(0.15 / ratio * 8192))
    k_img = int(round((1.0 - 0.15) / ratio * 8192))
    if k_img < 4096:                                                 # (representable at the output rate)
        assert spec[k_img - 3:k_img + 4].max() < 0.02 * spec[k_sig - 3:k_sig + 4].max()


def test_packet_recall_with_and_without_resampling(oracle_mod):
    """A 2 Msps capture demodulated directly (1 sample per chip) and after x2 interpolation (2 samples per chip, what
    modes_rx does): the interpolated path must not lose packets -- it decodes at least 85 % of what the direct path
    decodes, on frames whose payload matches the injected truth."""
    rate = 2e6
    iq, truth = synth.synth_capture(rate, 1500000, 700.0, seed=2024, overlap_frac=0.0)
    sent = {t["frame"] for t in truth}
    direct = {bytes(p["data"][:p["nbytes"]]).hex() for p in oracle_mod.demod(iq, rate, 7.0, True)} & sent
    up = arb_resampler(2.0).work(iq)
    inter = {bytes(p["data"][:p["nbytes"]]).hex() for p in oracle_mod.demod(up, 4e6, 7.0, True)} & sent
    assert len(direct) > 200
    assert len(inter) >= 0.85 * len(direct), (len(inter), len(direct), len(sent))


def test_modes_rx_resamples_below_4msps(emu_lib, oracle_mod, tmp_path, monkeypatch):
    """The command line at 2 Msps: messages == the oracle's at 4 Msps on the interpolated stream."""
    from conftest import EMU_LIB
    from air_modes import modes_rx
    rate = 2e6
    iq, _ = synth.synth_capture(rate, 300000, 600.0, seed=77)
    path = tmp_path / "cap2.cf32"
    np.asarray(iq, dtype=np.complex64).tofile(path)
    monkeypatch.setenv("AIRMODES_HIP_LIB", EMU_LIB)
    raw = io.StringIO()
    assert modes_rx.main(["-s", str(path), "-r", "2e6", "--raw", "--chunk", "70001"], out=raw) == 0
    up = arb_resampler(2.0).work(iq)
    want = oracle_mod.format_messages(oracle_mod.demod(up, 4e6, 7.0, True), 4e6)
    assert raw.getvalue().splitlines() == want and len(want) > 10


def _check_gpu_resampler(lib, n=400001):
    """csrc/am_resample.hip against its definition (resample.arb_resampler): the same bits, whatever the chunking."""
    rng = np.random.default_rng(5)
    x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.exp(rng.uniform(-12, 2, n))).astype(np.complex64)
    for ratio in (2.0, 4e6 / 2.4e6, 1.25, 1.0):
        a, g = arb_resampler(ratio), gpu_resampler(ratio, lib=lib)
        cuts = [0, 3, 11, 70001, 131072 + 5, 300000, n]
        for c0, c1 in zip(cuts[:-1], cuts[1:]):
            ya, yg = a.work(x[c0:c1]), g.work(x[c0:c1])
            assert ya.size == yg.size and np.array_equal(ya.view(np.uint32), yg.view(np.uint32)), (ratio, c0)
        g.close()


def test_gpu_resampler_matches_its_definition_emulated(emu_lib):
    _check_gpu_resampler(emu_lib, n=200001)


@pytest.mark.gpu
def test_gpu_resampler_matches_its_definition(hip_lib):
    _check_gpu_resampler(hip_lib)


@pytest.mark.gpu
def test_resampled_recall_on_device(hip_lib, oracle_mod):
    """2 Msps capture -> x2 on the GPU -> receive path on the GPU (samples never leave the device): the packets are the
    oracle's on the interpolated stream, and at least 85 % of what the direct 2 Msps path decodes."""
    from air_modes import _capi
    rate = 2e6
    iq, truth = synth.synth_capture(rate, 1500000, 700.0, seed=2024, overlap_frac=0.0)
    sent = {t["frame"] for t in truth}
    rs = gpu_resampler(2.0, lib=hip_lib)
    ctx = _capi.Context(4e6, 7.0, True, lib=hip_lib)
    got = []
    cuts = [0, 400001, 900000, len(iq)]
    for c0, c1 in zip(cuts[:-1], cuts[1:]):
        ptr, m = rs.work_device(iq[c0:c1])
        got.append(ctx.process_iq_device(ptr, m, flush=(c1 == len(iq))))
    got = np.concatenate(got)
    want = oracle_mod.demod(arb_resampler(2.0).work(iq), 4e6, 7.0, True)
    assert np.array_equal(got, want)
    direct = {bytes(p["data"][:p["nbytes"]]).hex() for p in oracle_mod.demod(iq, rate, 7.0, True)} & sent
    inter = {bytes(p["data"][:p["nbytes"]]).hex() for p in got} & sent
    assert len(direct) > 200 and len(inter) >= 0.85 * len(direct)
