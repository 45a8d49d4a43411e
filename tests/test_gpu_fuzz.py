"""Randomised differential test ON THE DEVICE: what tools/fuzz_emu.py draws for the CPU emulation, against the
real library on a real MI355X -- the DPP rotations, EXEC-narrowing compares, packed fp32, v_permlane32_swap, nt loads and wave
priorities of the streaming kernels exist only there, and fixed seeds are not a fuzz.

Every run draws NEW cases (a seed from the operating system, printed with the case number and every draw when a case fails;
AIRMODES_FUZZ_SEED=<seed> repeats the run) for ~60 s.  (A hypothesis @given was the first form: its engine calls a test that
stops drawing once a time box is spent "flaky data generation" -- a plain seeded loop keeps the box honest.)

Per case: a rate from 2 .. 64 Msps (whole and fractional samples per chip), a seeded capture with optional NaN / inf / denormal /
huge stretches, threshold and filter settings, then
  * the stream cut at random places (1-sample pieces, unaligned cuts) through am_process_iq == the oracle, packets as bytes;
  * bb / reference level of the block-level front end == the oracle's (u32);
  * random rx_time tags handed over with their chunks == the oracle;
  * the capture as K independent streams (empty ones, stubs) in ONE scan == the oracle on every stream;
  * every first-stage candidate record + tags + bursts of the production scan == the oracle, and the reference's own C++ where
    oracle/_ref travelled (check_production_stages, with_ref), whole-chip rates;
  * the stream over W ranks' device tables with steps in flight (run_stream_shards_in_flight) == the oracle.
"""
import os
import time

import numpy as np
import pytest

import oracle
import parity_common as pc
import synth
from air_modes import _capi

pytestmark = pytest.mark.gpu

RATES = (2e6, 4e6, 5e6, 6.25e6, 8e6, 10e6, 16e6, 20e6, 32e6, 40e6, 64e6)
BOX_SECONDS = 60.0
_ran = [0]


def same(a, b):
    return len(a) == len(b) and np.ascontiguousarray(a).tobytes() == np.ascontiguousarray(b).tobytes()


class _Draws(object):
    """numpy draws with a log (what hypothesis' data.draw(..., label=...) gave the first form of this test)"""

    def __init__(self, rng):
        self.rng, self.log = rng, []

    def note(self, label, v):
        self.log.append((label, v))
        return v

    def choice(self, seq, label):
        return self.note(label, seq[int(self.rng.integers(0, len(seq)))])

    def integer(self, lo, hi, label):
        return self.note(label, int(self.rng.integers(lo, hi + 1)))

    def boolean(self, label):
        return self.note(label, bool(self.rng.integers(0, 2)))

    def integers(self, lo, hi, count, label):
        return self.note(label, [int(x) for x in self.rng.integers(lo, hi + 1, count)])


def test_random_cases_on_the_device(hip_lib):
    seed0 = int(os.environ.get("AIRMODES_FUZZ_SEED", "0")) or int.from_bytes(os.urandom(8), "little")
    t0 = time.monotonic()
    case = 0
    while time.monotonic() - t0 < BOX_SECONDS:
        d = _Draws(np.random.default_rng([seed0, case]))
        try:
            _one_case(hip_lib, d)
        except Exception as ex:
            raise AssertionError("device fuzz: case %d of AIRMODES_FUZZ_SEED=%d failed: %s\ndraws: %s" % (case, seed0, ex, d.log)) from ex
        case += 1
        _ran[0] += 1


def _one_case(hip_lib, d):
    lib = hip_lib
    rate = d.choice(RATES, "rate")
    spc = int(rate / 2e6)
    whole = float(rate) == 2e6 * spc
    n = d.integer(20000 * spc, 45000 * spc, "n")
    lam = d.choice((300.0, 3000.0, 20000.0, 60000.0), "lambda")
    thr = d.choice((2.0, 5.0, 7.0, 10.0), "threshold")
    pmf = d.integer(0, 3, "pmf") != 0
    seed = d.integer(1, (1 << 30) - 1, "seed")
    iq, _ = synth.synth_capture(rate, n, lam, seed)
    iq = np.array(iq)
    if d.integer(0, 2, "nonfinite") == 0:
        k = d.integer(0, n - 700, "where")
        iq[k:k + 200] *= np.complex64(1e-22)                   # denormals
        iq[k + 300] = np.complex64(complex(np.nan, 1.0))
        iq[k + 400] = np.complex64(complex(np.inf, 0.0))
        iq[k + 500:k + 520] *= np.complex64(1e18)              # overflow to inf in |.|^2
    with np.errstate(all="ignore"):
        want = oracle.demod(iq, rate, thr, pmf)

        # 1. the stream in random pieces (1-sample pieces and unaligned cuts among them)
        ncut = d.integer(0, 5, "ncut")
        cuts = sorted(set(d.integers(1, n - 1, ncut, "cuts")))
        if cuts and d.boolean("one_sample_piece") and cuts[0] + 1 < n:
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
        ntag = d.integer(0, 3, "ntag")
        if ntag:
            offs = sorted(d.integers(0, n - 1, ntag, "tag_offsets"))
            rx = [(int(o), d.integer(0, 2_000_000_000, "secs"),
                   float(d.choice((0.0, 0.125, 0.5, 0.9999995, 0.75), "frac"))) for o in offs]
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
        J = d.integer(2, 5, "streams")
        jc = sorted(d.integers(0, n, J - 1, "stream_cuts"))
        pieces = [iq[a:b] for a, b in zip([0] + jc, jc + [n])]
        if d.boolean("stub"):
            pieces.insert(d.integer(0, len(pieces), "stub_at"), iq[:d.integer(0, 200, "stub_len")])
        buf, lens = ctx.multi_pack(pieces)
        got_k = ctx.process_multi(buf, lens, zero_gaps=d.boolean("zero_gaps"))
        for j, (g, x) in enumerate(zip(got_k, pieces)):
            assert same(g, oracle.demod(x, rate, thr, pmf)), "stream %d of %d in one scan differs" % (j, len(pieces))
        ctx.close()

        # 5. every candidate record, tags and bursts of the production scan; the reference's own C++ where it travelled
        if whole and d.boolean("stage_level"):
            pc.check_production_stages(lib, rate, n, lam, seed, thr=thr, pmf=pmf, iq=iq, with_ref=True)

        # 6. the stream over W ranks' device tables with steps in flight (am_shard_resolve_submit / _collect, a carry word per rank)
        if d.boolean("shards_in_flight"):
            W, K = d.integer(1, 4, "ranks"), d.integer(2, 4, "steps")
            m = n // (W * K)
            if m > 344 * (spc + 1):
                part = iq[:m * W * K]
                got, _ = pc.run_stream_shards_in_flight(lib, rate, part, W, K, thr, pmf, small_cap=d.choice((1, 8, 512), "message_entries"))
                assert same(got, want if m * W * K == n else oracle.demod(part, rate, thr, pmf)), "shards with steps in flight differ"


def test_the_box_was_used(hip_lib):
    """(runs after the fuzz in file order) a box that ran no case at all -- an import error swallowed, a library that refuses every
    call -- would be a silent pass"""
    print("device fuzz: %d random cases inside the %g s box" % (_ran[0], BOX_SECONDS))
    assert _ran[0] >= 5, "only %d random cases ran inside the %g s box" % (_ran[0], BOX_SECONDS)
