"""Randomised differential test ON THE DEVICE: what tools/fuzz_emu.py draws for the CPU emulation, against the
real library on a real MI355X -- the DPP rotations, EXEC-narrowing compares, packed fp32, v_permlane32_swap, nt loads and wave
priorities of the streaming kernels exist only there, and fixed seeds are not a fuzz.

Every run draws NEW cases (hypothesis, not derandomised; a failure prints the falsifying example, which `@example` or
`--hypothesis-seed` reproduces) and is boxed to ~60 s: once the box is spent the remaining examples return at once.

Per case: a rate from 2 .. 64 Msps (whole and fractional samples per chip), a seeded capture with optional NaN / inf / denormal /
huge stretches, threshold and filter settings, then
  * the stream cut at random places (1-sample pieces, unaligned cuts) through am_process_iq == the oracle, packets as bytes;
  * bb / reference level of the block-level front end == the oracle's (u32);
  * random rx_time tags handed over with their chunks == the oracle;
  * the capture as K independent streams (empty ones, stubs) in ONE scan == the oracle on every stream;
  * every first-stage candidate record + tags + bursts of the production scan == the oracle, and the reference's own C++ where
    oracle/_ref travelled (check_production_stages, with_ref), whole-chip rates.
"""
import time

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import oracle
import parity_common as pc
import synth
from air_modes import _capi

pytestmark = pytest.mark.gpu

RATES = (2e6, 4e6, 5e6, 6.25e6, 8e6, 10e6, 16e6, 20e6, 32e6, 40e6, 64e6)
BOX_SECONDS = 60.0
_t0 = [None]
_ran = [0]


def same(a, b):
    return len(a) == len(b) and np.ascontiguousarray(a).tobytes() == np.ascontiguousarray(b).tobytes()


@settings(max_examples=100000, deadline=None, derandomize=False, database=None,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large, HealthCheck.filter_too_much,
                                 HealthCheck.function_scoped_fixture])
@given(st.data())
def test_random_cases_on_the_device(hip_lib, data):
    if _t0[0] is None:
        _t0[0] = time.monotonic()
    if time.monotonic() - _t0[0] > BOX_SECONDS:
        return                                                 # the box is spent: the remaining examples cost nothing
    lib = hip_lib
    draw = data.draw
    rate = draw(st.sampled_from(RATES), label="rate")
    spc = int(rate / 2e6)
    whole = float(rate) == 2e6 * spc
    n = draw(st.integers(20000 * spc, 45000 * spc), label="n")
    lam = draw(st.sampled_from((300.0, 3000.0, 20000.0, 60000.0)), label="lambda")
    thr = draw(st.sampled_from((2.0, 5.0, 7.0, 10.0)), label="threshold")
    pmf = draw(st.booleans(), label="pmf") or draw(st.booleans(), label="pmf2")
    seed = draw(st.integers(1, (1 << 30) - 1), label="seed")
    iq, _ = synth.synth_capture(rate, n, lam, seed)
    iq = np.array(iq)
    if draw(st.integers(0, 2), label="nonfinite") == 0:
        k = draw(st.integers(0, n - 700), label="where")
        iq[k:k + 200] *= np.complex64(1e-22)                   # denormals
        iq[k + 300] = np.complex64(complex(np.nan, 1.0))
        iq[k + 400] = np.complex64(complex(np.inf, 0.0))
        iq[k + 500:k + 520] *= np.complex64(1e18)              # overflow to inf in |.|^2
    with np.errstate(all="ignore"):
        want = oracle.demod(iq, rate, thr, pmf)

        # 1. the stream in random pieces (1-sample pieces and unaligned cuts among them)
        ncut = draw(st.integers(0, 5), label="ncut")
        cuts = sorted(set(draw(st.lists(st.integers(1, n - 1), min_size=ncut, max_size=ncut), label="cuts")))
        if cuts and draw(st.booleans(), label="one_sample_piece") and cuts[0] + 1 < n:
            cuts = sorted(set(cuts + [cuts[0] + 1]))
        ctx = _capi.Context(rate, thr, pmf, lib=lib)
        parts = [ctx.process_iq(iq[a:b], flush=(b == n)) for a, b in zip([0] + cuts, cuts + [n])]
        assert same(np.concatenate(parts), want), "chunked stream differs: %d vs %d packets" % (sum(map(len, parts)), len(want))

        # 2. the front end's dense outputs
        m = min(n, 12000 * spc)
        bb, avg = ctx.frontend_work(iq[:m])
        obb, oavg = oracle.frontend(iq[:m], spc, pmf)
        assert np.array_equal(pc.u32(bb), pc.u32(obb)) and np.array_equal(pc.u32(avg), pc.u32(oavg)), "bb / reference level differ"

        # 3. rx_time tags arriving with their chunks
        ntag = draw(st.integers(0, 3), label="ntag")
        if ntag:
            offs = sorted(draw(st.lists(st.integers(0, n - 1), min_size=ntag, max_size=ntag), label="tag_offsets"))
            rx = [(int(o), int(draw(st.integers(0, 2_000_000_000), label="secs")),
                   float(draw(st.sampled_from((0.0, 0.125, 0.5, 0.9999995, 0.75)), label="frac"))) for o in offs]
            want_t = oracle.demod(iq, rate, thr, pmf, rx_time=rx)
            ctx.reset()
            edges = [0] + cuts + [n]
            got = []
            for a, b in zip(edges[:-1], edges[1:]):
                for tag in rx:
                    if a <= tag[0] < b:
                        ctx.set_rx_time(*tag)
                got.append(ctx.process_iq(iq[a:b], flush=(b == n)))
            assert same(np.concatenate(got), want_t), "stream with rx_time tags differs"
        ctx.reset()

        # 4. K independent streams in one scan
        J = draw(st.integers(2, 5), label="streams")
        jc = sorted(draw(st.lists(st.integers(0, n), min_size=J - 1, max_size=J - 1), label="stream_cuts"))
        pieces = [iq[a:b] for a, b in zip([0] + jc, jc + [n])]
        if draw(st.booleans(), label="stub"):
            pieces.insert(draw(st.integers(0, len(pieces)), label="stub_at"), iq[:draw(st.integers(0, 200), label="stub_len")])
        buf, lens = ctx.multi_pack(pieces)
        got_k = ctx.process_multi(buf, lens, zero_gaps=draw(st.booleans(), label="zero_gaps"))
        for j, (g, x) in enumerate(zip(got_k, pieces)):
            assert same(g, oracle.demod(x, rate, thr, pmf)), "stream %d of %d in one scan differs" % (j, len(pieces))
        ctx.close()

        # 5. every candidate record, tags and bursts of the production scan; the reference's own C++ where it travelled
        if whole and draw(st.booleans(), label="stage_level"):
            pc.check_production_stages(lib, rate, n, lam, seed, thr=thr, pmf=pmf, iq=iq, with_ref=True)
    _ran[0] += 1


def test_the_box_was_used(hip_lib):
    """(runs after the fuzz in file order) a box that ran no case at all -- an import error swallowed, a library that refuses every
    call -- would be a silent pass"""
    assert _ran[0] >= 5, "only %d random cases ran inside the %g s box" % (_ran[0], BOX_SECONDS)
