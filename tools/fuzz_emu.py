#!/usr/bin/env python3
"""Randomised differential campaign, CPU only: the product sources compiled against the fiber emulation (tests/emu,
both builds: the plain one and the one with 4-slot block heads / ticket-ordered scans) against the oracle on random
rates, burst densities, thresholds, filter settings, chunk cuts and shard counts.  Test infrastructure; prints one
line per case and stops at the first disagreement.

    python tools/fuzz_emu.py [--cases 40] [--seed 1]
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("gr-air-modes_amd", "tests", "oracle", "tools"):
    sys.path.insert(0, os.path.join(ROOT, p))

import numpy as np  # noqa: E402

import oracle  # noqa: E402
import parity_common as pc  # noqa: E402
import synth  # noqa: E402
from air_modes import _capi  # noqa: E402


def same(a, b):
    """packet lists equal as bytes (a NaN reference level is equal to itself here)"""
    return len(a) == len(b) and np.ascontiguousarray(a).tobytes() == np.ascontiguousarray(b).tobytes()


def run_chunked(lib, rate, iq, edges, thr, pmf, dc):
    ctx = _capi.Context(rate, thr, pmf, use_dcblock=dc, lib=lib)
    parts = [ctx.process_iq(iq[a:b], flush=(b == len(iq))) for a, b in zip(edges[:-1], edges[1:])]
    ctx.close()
    return np.concatenate(parts) if parts else np.zeros(0, _capi.PACKET_DTYPE)


def run_sharded(lib, rate, iq, G, thr, pmf, dc):
    n = len(iq)
    ctxs = [_capi.Context(rate, thr, pmf, use_dcblock=dc, lib=lib) for _ in range(G)]
    hl, hr = ctxs[0].shard_halo()
    bounds = [(g * n) // G for g in range(G + 1)]
    tables = [ctxs[g].shard_scan(iq[max(0, bounds[g] - hl):min(n, bounds[g + 1] + hr)], bounds[g], bounds[g + 1], n)
              for g in range(G)]
    entry = _capi.shard_entries(lib, tables, bounds[:-1])
    got = np.concatenate([ctxs[g].shard_resolve(int(entry[g])) for g in range(G)])
    for c in ctxs:
        c.close()
    return got


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    emu = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-s", "-C", emu])
    subprocess.check_call(["make", "-s", "-C", emu, "libairmodes_emu_rare.so"])
    # private copies: a campaign runs for an hour or two, and a rebuild of tests/emu in the meantime must not pull the file out from
    # under it
    import shutil
    import tempfile
    priv = tempfile.mkdtemp(prefix="fuzz_emu_")
    for f in ("libairmodes_emu.so", "libairmodes_emu_rare.so"):
        shutil.copy(os.path.join(emu, f), os.path.join(priv, f))
    libs = [_capi.Library(os.path.join(priv, "libairmodes_emu.so")), _capi.Library(os.path.join(priv, "libairmodes_emu_rare.so"))]
    rng = np.random.default_rng(args.seed)
    rates = (2e6, 2e6, 4e6, 8e6, 10e6, 10e6, 16e6, 20e6, 20e6, 32e6, 40e6, 64e6, 64e6,
             3e6, 4.8e6, 5e6, 6.25e6, 7e6, 13e6, 25.5e6)       # (and rates that are not multiples of 2 MHz)
    for case in range(args.cases):
        rate = float(rng.choice(rates))
        spc = int(rate / 2e6)
        n = int(rng.integers(30000 * spc, 90000 * spc))
        lam = float(rng.choice((300.0, 3000.0, 20000.0, 60000.0)))
        thr = float(rng.choice((2.0, 5.0, 7.0, 10.0)))
        pmf = bool(rng.integers(0, 4))
        dc = bool(rng.integers(0, 5) == 0)                   # the optional DC blocker in front of the path
        seed = int(rng.integers(1, 1 << 30))
        iq, _ = synth.synth_capture(rate, n, lam, seed)
        kind = int(rng.integers(0, 4))
        if kind == 0:                                        # some non-finite / tiny / huge samples
            k = int(rng.integers(0, n - 600))
            iq[k:k + 200] *= np.complex64(1e-22)
            iq[k + 300] = np.complex64(complex(np.nan, 1.0))
            iq[k + 400] = np.complex64(complex(np.inf, 0.0))
            iq[k + 500:k + 520] *= np.complex64(1e18)
        with np.errstate(all="ignore"):
            want = oracle.demod(iq, rate, thr, pmf, use_dcblock=dc)
        lib = libs[case % 2]
        cuts = sorted(set(int(x) for x in rng.integers(1, n, int(rng.integers(0, 4)))))
        edges = [0] + cuts + [n]
        got = run_chunked(lib, rate, iq, edges, thr, pmf, dc)
        assert same(got, want), "case %d: chunked result differs (%d vs %d packets)" % (case, len(got), len(want))
        G = int(rng.integers(2, 5))
        halo = 244 * (spc + 1) + 2 + (200 * spc if dc else 0)
        if n // G > halo:
            got = run_sharded(lib, rate, iq, G, thr, pmf, dc)
            assert same(got, want), "case %d: sharded result differs (%d vs %d packets)" % (case, len(got), len(want))
        # ... and as a RECEIVER: K steps of W chunks, the scan position crossing chunks and steps
        W, K = int(rng.integers(1, 4)), int(rng.integers(2, 4))
        m = n // (W * K)
        if m > 344 * (spc + 1) + (200 * spc if dc else 0):
            got = pc.run_stream_shards(lib, rate, iq[:m * W * K], W, K, thr, pmf, dc)
            with np.errstate(all="ignore"):
                want2 = want if m * W * K == n else oracle.demod(iq[:m * W * K], rate, thr, pmf, use_dcblock=dc)
            assert same(got, want2), "case %d: streamed shards differ (%d vs %d packets, W=%d K=%d)" % (case, len(got), len(want2), W, K)
            # ... with steps in flight and the tables on the device (am_shard_resolve_submit / _collect, a carry word per rank)
            got, _ = pc.run_stream_shards_in_flight(lib, rate, iq[:m * W * K], W, K, thr, pmf, dc, small_cap=int(rng.choice([1, 8, 512])))
            assert same(got, want2), "case %d: shards with steps in flight differ (%d vs %d packets, W=%d K=%d)" % (case, len(got), len(want2), W, K)
        if not dc:
            # K streams in one scan (am_process_multi): the capture cut into independent streams, an empty one and a stub among them
            J = int(rng.integers(2, 6))
            jc = sorted(int(x) for x in rng.integers(0, n + 1, J - 1))
            pieces = [iq[a:b] for a, b in zip([0] + jc, jc + [n])]
            if rng.integers(0, 2):
                pieces.insert(int(rng.integers(0, len(pieces) + 1)), iq[:int(rng.integers(0, 200))])
            ctx = _capi.Context(rate, thr, pmf, lib=lib)
            buf, lens = ctx.multi_pack(pieces)
            got_k = ctx.process_multi(buf, lens, zero_gaps=bool(rng.integers(0, 2)))
            for j, (g, x) in enumerate(zip(got_k, pieces)):
                with np.errstate(all="ignore"):
                    w = oracle.demod(x, rate, thr, pmf)
                assert same(g, w), "case %d: stream %d of %d in one scan differs (%d vs %d packets)" % (case, j, len(pieces), len(g), len(w))
            # the preamble block as a stream (am_preamble_stream): random pieces of its two inputs against one work() over them
            bb, avg = ctx.frontend_work(iq)
            ctx.reset()
            wb, wt = ctx.preamble_work(bb, avg)
            ctx.reset()
            pc_ = sorted(int(x) for x in rng.integers(0, n + 1, int(rng.integers(1, 7))))
            gb, gt = [], []
            for a, b in zip([0] + pc_, pc_ + [n]):
                b_, t_ = ctx.preamble_stream(bb[a:b], avg[a:b], flush=False)
                gb.append(b_); gt.append(t_)
            b_, t_ = ctx.preamble_stream(bb[:0], avg[:0], flush=True)
            gb.append(b_); gt.append(t_)
            assert np.concatenate(gt).tobytes() == wt.tobytes() and np.concatenate(gb).tobytes() == wb.tobytes(), \
                "case %d: the preamble block as a stream differs (cuts %s)" % (case, pc_)
            ctx.close()
        # ONE continuing stream with several of its chunks in flight (am_spipe, round 6): random chunk sizes, contiguous or scattered
        # buffers, against the oracle over the whole stream (a capture of its own: the helper draws it from the seed)
        if n > 12 * (344 * (spc + 1) + (200 * spc if dc else 0)):
            with np.errstate(all="ignore"):
                pc.check_stream_pipe(lib, rate, n, lam, seed, depth=int(rng.integers(1, 5)), thr=thr, pmf=pmf, dcblock=dc,
                                     contiguous=bool(rng.integers(0, 2)))
        print("case %3d ok: %5.0f Msps n=%8d lambda=%6.0f thr=%4.1f pmf=%d dc=%d kind=%d cuts=%s shards=%d packets=%d (%s build)"
              % (case, rate / 1e6, n, lam, thr, pmf, dc, kind, cuts, G, len(want), "rare" if case % 2 else "plain"), flush=True)


if __name__ == "__main__":
    main()
