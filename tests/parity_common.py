"""Shared parity checks: the same assertions run against the CPU-fiber emulation of the
kernels (no GPU, `-m "not gpu"`) and against the real HIP library (`-m gpu`)."""
import glob
import os

import numpy as np

import oracle
import synth
from air_modes import _capi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def u32(a):
    return np.ascontiguousarray(a).view(np.uint32)


def golden_cases():
    return sorted(glob.glob(os.path.join(GOLDEN, "g_*.npz")))


def load_golden(path):
    z = np.load(path)
    return dict(iq=z["iq"], rate=float(z["rate"]), thr=float(z["thr_db"]), pmf=bool(int(z["use_pmf"])),
                ref_bursts=z["ref_bursts"], ref_sample=z["ref_tag_sample"], ref_secs=z["ref_tag_secs"],
                ref_frac=z["ref_tag_frac"], ref_msgs=[str(m) for m in z["ref_msgs"]],
                bb_sha=str(z["bb_sha256"]), avg_sha=str(z["avg_sha256"]))


def preamble_geometry(rate, n):
    """(history items in front of the stream, ninputs, room a burst needs) of the reference's preamble block for a stream
    of n samples, in its own float arithmetic (lib/preamble_impl.cc:57-62,150,212) -- for the bounds of the candidate check."""
    spcf = np.float32(np.float32(int(rate)) / np.float32(2000000))
    hist = int(np.float32(spcf * np.float32(2))) - 1
    S = int(spcf)
    K = n + hist
    ninputs = K - K % S - S
    room = int(np.ceil(np.float32(np.float32(240) * spcf)))
    return hist, ninputs, room


def messages(lib, packets):
    return [lib.format_message(packets[i], i == 0) for i in range(len(packets))]


def check_golden(lib, path):
    """Product path vs what the REFERENCE's own C++ produced (tests/golden)."""
    g = load_golden(path)
    spc = int(g["rate"] / 2e6)
    ctx = _capi.Context(g["rate"], g["thr"], g["pmf"], lib=lib)
    bb, avg = ctx.frontend_work(g["iq"])
    obb, oavg = oracle.frontend(g["iq"], spc, g["pmf"])
    assert np.array_equal(u32(bb), u32(obb)) and np.array_equal(u32(avg), u32(oavg))
    bursts, tags = ctx.preamble_work(bb, avg)
    assert len(tags) == len(g["ref_sample"])
    assert np.array_equal(tags["sample"], g["ref_sample"])
    assert np.array_equal(tags["secs"], g["ref_secs"])
    assert np.array_equal(tags["frac"], g["ref_frac"])
    assert np.array_equal(u32(bursts), u32(g["ref_bursts"]))
    pk = ctx.slicer_work(bursts, tags)
    assert messages(lib, pk) == g["ref_msgs"]
    pk2 = ctx.process_iq(g["iq"], flush=True)
    assert messages(lib, pk2) == g["ref_msgs"]
    ctx.close()


def check_production_stages(lib, rate, n, lam, seed, thr=7.0, pmf=True, want_fe=None, iq=None, with_ref=False, chunks=None):
    """Stage-level parity of the kernels am_process_iq really runs: the record of EVERY
    first-stage candidate (position = the first-stage bitmap, refined position, quiet-zone outcome, reference level) against
    the oracle's refinement of every position that passes the first-stage test; the bursts + tags the scan hands its slicer
    (AM_F_KEEP_TAGS) against the oracle's preamble scan and, where oracle/_ref exists, the reference's own C++."""
    spc = int(rate / 2e6)
    if iq is None:
        iq, _ = synth.synth_capture(rate, n, lam, seed)
    n = len(iq)
    ctx = _capi.Context(rate, thr, pmf, lib=lib)
    if chunks:
        got_b, got_t, pk = [], [], []
        cuts = [0] + list(chunks) + [n]
        for a, b in zip(cuts[:-1], cuts[1:]):
            pk.append(ctx.process_iq(iq[a:b], flush=(b == n), keep_tags=True))
            bb_, tt_ = ctx.fetch_tags()
            got_b.append(bb_); got_t.append(tt_)
        bursts, tags, whole = np.concatenate(got_b), np.concatenate(got_t), np.concatenate(pk)
    else:
        whole = ctx.process_iq(iq, flush=True, keep_tags=True)
        bursts, tags = ctx.fetch_tags()
    if want_fe is not None:
        assert ctx.last_frontend() == want_fe, "front end %d ran, expected %d" % (ctx.last_frontend(), want_fe)
    obb, oavg = oracle.frontend(iq, spc, pmf)
    ob, ot = oracle.preamble_scan(obb, oavg, spc, thr, rate)
    assert len(tags) == len(ot), "tag count differs: %d vs %d" % (len(tags), len(ot))
    # (records compared as bytes: a NaN reference level -- non-finite samples in the fuzz -- is equal to itself here)
    assert np.ascontiguousarray(tags).tobytes() == np.ascontiguousarray(ot).tobytes(), "tags differ"
    assert np.array_equal(u32(bursts), u32(ob)), "bursts differ"
    want, ntags = oracle.demod(iq, rate, thr, pmf, return_tags=True)
    assert ntags == len(tags)
    assert np.ascontiguousarray(whole).tobytes() == np.ascontiguousarray(want).tobytes(), "packets differ"
    if with_ref and oracle.have_ref():
        rb, rt, _, keep = oracle.ref_preamble_slicer(obb, oavg, spc, thr, rate)
        rb, rt = rb[keep], rt[keep]
        assert np.array_equal(tags["sample"], rt["sample"]) and np.array_equal(tags["secs"], rt["secs"])
        assert np.array_equal(tags["frac"], rt["frac"]) and np.array_equal(u32(bursts), u32(rb)), "differs from the reference's C++"
    if not chunks:
        # every first-stage candidate of the (single) scan.  The GPU tests positions up to the last one a burst can
        # start at (end-of-stream rule, preamble_impl.cc:150,212): item counts k <= ninputs - 240 spc
        hist, ninputs, room = preamble_geometry(rate, n)
        pos, ref_, val, iav = ctx.fetch_candidates()
        opos, oref, oval, oiav = oracle.candidates(obb, oavg, spc, thr, k_limit=ninputs - room + 1, rate=rate)
        assert len(pos) == len(opos), "candidate count differs: %d vs %d" % (len(pos), len(opos))
        assert np.array_equal(pos + np.uint64(hist), opos), "first-stage positions differ"
        assert np.array_equal(ref_ + np.uint64(hist), oref), "refined positions differ"
        assert np.array_equal(val, oval), "quiet-zone outcomes differ"
        ok = oval == 1
        assert np.array_equal(u32(iav[ok]), u32(oiav[ok])), "reference levels at valid candidates differ"
    ctx.close()
    return len(tags)


def check_stages(lib, rate, n, lam, seed, thr=7.0, pmf=True, dcblock=False, dc_offset=0.0):
    """Stage-by-stage and end-to-end, product path vs oracle, on a seeded capture."""
    spc = int(rate / 2e6)
    iq, _ = synth.synth_capture(rate, n, lam, seed)
    if dc_offset:
        iq = (iq + np.complex64(dc_offset * (1 + 0.6j))).astype(np.complex64)
    ctx = _capi.Context(rate, thr, pmf, use_dcblock=dcblock, lib=lib)
    bb, avg = ctx.frontend_work(iq)
    obb, oavg = oracle.frontend(oracle.dcblock(iq, spc) if dcblock else iq, spc, pmf)
    assert np.array_equal(u32(bb), u32(obb)), "bb differs"
    assert np.array_equal(u32(avg), u32(oavg)), "avg differs"
    bursts, tags = ctx.preamble_work(obb, oavg)
    ob, ot = oracle.preamble_scan(obb, oavg, spc, thr, rate)
    assert len(tags) == len(ot) and np.array_equal(tags, ot), "tags differ"
    assert np.array_equal(u32(bursts), u32(ob)), "bursts differ"
    pk = ctx.slicer_work(ob, ot)
    opk = oracle.slice_bursts(ob, ot)
    assert np.array_equal(pk, opk), "slicer differs"
    whole = ctx.process_iq(iq, flush=True)
    want = oracle.demod(iq, rate, thr, pmf, use_dcblock=dcblock)
    assert np.array_equal(whole, want), "end-to-end differs"
    assert messages(lib, whole) == oracle.format_messages(want)
    ctx.close()
    return len(want)


def check_chunked(lib, rate, iq, edges, thr=7.0, pmf=True, want=None, dcblock=False):
    """Results must not depend on how the stream is cut into am_process_iq calls."""
    if want is None:
        want = oracle.demod(iq, rate, thr, pmf, use_dcblock=dcblock)
    ctx = _capi.Context(rate, thr, pmf, use_dcblock=dcblock, lib=lib)
    parts = []
    n = len(iq)
    edges = [0] + [e for e in edges if 0 < e < n] + [n]
    for a, b in zip(edges[:-1], edges[1:]):
        parts.append(ctx.process_iq(iq[a:b], flush=(b == n)))
    got = np.concatenate(parts) if parts else np.zeros(0, _capi.PACKET_DTYPE)
    ctx.close()
    assert np.array_equal(got, want), "chunked result differs (%d vs %d packets)" % (len(got), len(want))
    return len(got)


def check_sharded(lib, rate, iq, G, thr=7.0, pmf=True, want=None, ctxs=None, dcblock=False):
    """Time-sharded operation (G chunks, exit-table exchange) == whole-stream result.  `ctxs`: reuse
    these contexts (one per chunk, as a receiver that steps over many batches does) and keep them open."""
    n = len(iq)
    if want is None:
        want = oracle.demod(iq, rate, thr, pmf, use_dcblock=dcblock)
    keep = ctxs is not None
    if not keep:
        ctxs = [_capi.Context(rate, thr, pmf, use_dcblock=dcblock, lib=lib) for _ in range(G)]
    hl, hr = ctxs[0].shard_halo()
    bounds = [(g * n) // G for g in range(G + 1)]
    tables = []
    for g in range(G):
        a, b = bounds[g], bounds[g + 1]
        tables.append(ctxs[g].shard_scan(iq[max(0, a - hl):min(n, b + hr)], a, b, n))
    entry = _capi.shard_entries(lib, tables, bounds[:-1])
    got = np.concatenate([ctxs[g].shard_resolve(int(entry[g])) for g in range(G)])
    if not keep:
        for c in ctxs:
            c.close()
    assert np.array_equal(got, want), "sharded result differs (%d vs %d packets)" % (len(got), len(want))
    return len(got)


def run_stream_shards(lib, rate, iq, W, K, thr=7.0, pmf=True, dcblock=False):
    """The time-sharded RECEIVER through the C ABI alone (what air_modes/sharded.py does over torch.distributed, on its
    synchronous path): K steps of W chunks of m samples; rank r of step k decides the positions [k W m + r m - H, k W m +
    (r + 1) m - H) from its own samples + the tail in front of them (AM_F_MORE), the last step ends the stream; the scan
    position crosses chunks and steps through am_shard_entry2.  Returns the packets in (step, rank) order."""
    m = len(iq) // (W * K)
    assert m * W * K == len(iq)
    ctxs = [_capi.Context(rate, thr, pmf, use_dcblock=dcblock, lib=lib) for _ in range(W)]
    hl, H = ctxs[0].shard_halo()
    assert m >= hl + H, "chunk shorter than what a rank needs in front of it"
    cur, out = 0, []
    for k in range(K):
        S0, flush = k * W * m, k == K - 1
        total = S0 + W * m
        tables = []
        for r in range(W):
            a0 = 0 if (k == 0 and r == 0) else S0 + r * m - H
            a1 = total if (flush and r == W - 1) else S0 + (r + 1) * m - H
            lo, hi = max(0, a0 - hl), S0 + (r + 1) * m              # the samples rank r has
            tables.append(ctxs[r].shard_scan(iq[lo:hi], a0, a1, total, more=not flush))
        entry, leave = _capi.shard_entries(lib, tables, cur_in=cur, with_exits=True)
        for r in range(W):
            out.append(ctxs[r].shard_resolve(int(entry[r])))
        cur = int(leave[W - 1])
    for c in ctxs:
        c.close()
    return np.concatenate(out)


def run_stream_shards_in_flight(lib, rate, iq, W, K, thr=7.0, pmf=True, dcblock=False, small_cap=512, rx_time=()):
    """The same receiver with STEPS IN FLIGHT and the tables on the device, W ranks in ONE process: what
    air_modes/sharded.py::PipelinedShardedReceiver does over torch.distributed, through the C ABI alone.  Every rank has two
    contexts (steps alternate) and a carry word of its own; a step's messages lie side by side in one buffer (the all-gather is
    the layout); step k + 1 is scanned on every rank BEFORE step k is resolved; am_shard_resolve_submit composes the entry through
    all ranks' tables from the rank's carry word (cur_in) and leaves the step's last exit there (carry_out).  A flagged step (redo)
    is repeated on the synchronous path by every rank, with its successor already scanned.  Returns (packets in (step, rank)
    order, steps redone)."""
    m = len(iq) // (W * K)
    assert m * W * K == len(iq)
    emulated = bool(getattr(lib, "emulated", False))
    ctxs = [[_capi.Context(rate, thr, pmf, use_dcblock=dcblock, device=(-1 if emulated else 0), lib=lib) for _ in range(2)] for _ in range(W)]
    for pair in ctxs:
        for c in pair:
            for tag in rx_time:
                c.set_rx_time(*tag)
    hl, H = ctxs[0][0].shard_halo()
    assert m >= hl + H
    words = 2 * (_capi.SHARD_MSG_HEADER + small_cap)
    f32 = np.ascontiguousarray(iq).view(np.float32)
    if emulated:
        dev_iq, base = f32, f32.ctypes.data
        msgs = [np.zeros(W * words, np.int64) for _ in range(2)]
        carry = [np.zeros(2, np.int64) for _ in range(W)]
        ptr = lambda a: a.ctypes.data
        rd = lambda a: int(np.int64(a[0]).astype(np.uint64))

        def wr(a, v):
            a[0] = np.uint64(v).astype(np.int64)
    else:
        import torch
        dev_iq = torch.from_numpy(f32.copy()).cuda()
        base = dev_iq.data_ptr()
        msgs = [torch.zeros(W * words, dtype=torch.int64, device="cuda") for _ in range(2)]
        carry = [torch.zeros(2, dtype=torch.int64, device="cuda") for _ in range(W)]
        ptr = lambda a: a.data_ptr()
        rd = lambda a: int(np.int64(a[0].item()).astype(np.uint64))

        def wr(a, v):
            a[0] = int(np.uint64(v).astype(np.int64))

    def geometry(k, r):
        S0, flush = k * W * m, k == K - 1
        total = S0 + W * m
        a0 = 0 if (k == 0 and r == 0) else S0 + r * m - H
        a1 = total if (flush and r == W - 1) else S0 + (r + 1) * m - H
        return a0, a1, total, max(0, a0 - hl), S0 + (r + 1) * m, flush

    def scan(k):
        for r in range(W):
            a0, a1, total, lo, hi, flush = geometry(k, r)
            ctxs[r][k % 2].shard_scan_async(base + 8 * lo, a0, a1, total, ptr(msgs[k % 2]) + 8 * r * words, small_cap, device_in=True, more=not flush)

    out, redone = [], 0
    scan(0)
    for k in range(K):
        if k + 1 < K:
            scan(k + 1)                                      # ... before step k is resolved
        if not emulated:
            torch.cuda.synchronize()                         # (every rank has streams of its own here: the tables of all are complete)
        for r in range(W):
            ctxs[r][k % 2].shard_resolve_submit(ptr(msgs[k % 2]), W, r, small_cap, cur_in_ptr=ptr(carry[r]), carry_out_ptr=ptr(carry[r]))
        got = [ctxs[r][k % 2].shard_resolve_collect(capacity=max(64, m // 2000 + 64)) for r in range(W)]
        flags = [g[1] for g in got]
        assert len(set(flags)) == 1, "the ranks disagree about repeating step %d: %s" % (k, flags)
        if flags[0]:
            redone += 1
            cur = rd(carry[0])
            assert all(rd(carry[r]) == cur for r in range(W)), "a flagged step moved a carry word"
            tables = []
            for r in range(W):
                a0, a1, total, lo, hi, flush = geometry(k, r)
                tables.append(ctxs[r][k % 2].shard_scan(iq[lo:hi], a0, a1, total, more=not flush))
            entry, leave = _capi.shard_entries(lib, tables, cur_in=cur, with_exits=True)
            got = [(ctxs[r][k % 2].shard_resolve(int(entry[r])), False) for r in range(W)]
            for r in range(W):
                wr(carry[r], int(leave[W - 1]))
        out.extend(g[0] for g in got)
    for pair in ctxs:
        for c in pair:
            c.close()
    return np.concatenate(out), redone


def check_stream_sharded(lib, rate, iq, W, K, thr=7.0, pmf=True, want=None, dcblock=False):
    if want is None:
        want = oracle.demod(iq, rate, thr, pmf, use_dcblock=dcblock)
    got = run_stream_shards(lib, rate, iq, W, K, thr, pmf, dcblock)
    assert np.array_equal(got, want), "streamed shards differ (%d vs %d packets)" % (len(got), len(want))
    return len(got)


def rx_time_golden_cases():
    return sorted(glob.glob(os.path.join(GOLDEN, "rxtime_*.npz")))


def check_preamble_stream(lib, rate, n, lam, seed, thr=7.0, pmf=True, trials=4):
    """The preamble block as a stream (am_preamble_stream, preamble.general_work): consecutive calls on random pieces
    of the block's two input streams -- pieces shorter than a burst, empty ones, cuts inside preambles -- give the
    bursts and tags of ONE work() over the whole streams (= the oracle's scan), "rx_time" tags included; a flush
    ends the stream and the next one starts at item 0; reset() drops a half-fed stream."""
    import air_modes
    spc = max(int(rate / 2e6), 1)
    iq, _ = synth.synth_capture(rate, n, lam, seed)
    whole = float(rate) == 2e6 * spc                 # (the oracle's scan is pinned at whole samples per chip)
    if whole:
        obb, oavg = oracle.frontend(iq, spc, pmf)
    else:
        fe = _capi.Context(rate, thr, pmf, lib=lib)
        obb, oavg = fe.frontend_work(iq)
        fe.close()
    rx = [(0, 1600000000, 0.125), (n // 3 + 5, 1600000007, 0.9999995), (2 * n // 3, 1700000000, 0.75)]
    blk = air_modes.preamble(rate, thr, lib=lib)
    wb, wt = blk.work(obb, oavg, rx_time=rx)
    if whole:
        ob, ot = oracle.preamble_scan(obb, oavg, spc, thr, rate, rx_time=rx)
        assert np.array_equal(wt, ot) and np.array_equal(u32(wb), u32(ob))
    assert len(wt) > 3
    rng = np.random.default_rng(seed)
    hold = 244 * spc
    for t in range(trials):
        blk.reset()
        k = int(rng.integers(2, 12))
        cuts = sorted(int(c) for c in rng.integers(0, n + 1, k))
        if t == 0 and len(wt):                       # cuts inside a hit's preamble and inside its look-ahead
            s0 = int(wt["sample"][len(wt) // 2])
            cuts = sorted(cuts + [max(s0 - 3, 0), s0 + 2 * spc, min(s0 + hold - 1, n), min(s0 + hold, n), min(s0 + hold + 1, n)])
        if t == 1:
            cuts = sorted(cuts + [cuts[0], cuts[0] + min(7, n - cuts[0])])      # an empty piece and a 7-item one
        edges = [0] + cuts + [n]
        gb, gt = [], []
        for a, b in zip(edges[:-1], edges[1:]):
            tg = [x for x in rx if a <= x[0] < b] if b > a else []
            if a == b == 0:
                tg = []
            b_, t_ = blk.general_work(obb[a:b], oavg[a:b], rx_time=tg, flush=False)
            gb.append(b_); gt.append(t_)
        b_, t_ = blk.general_work(obb[:0], oavg[:0], flush=True)
        gb.append(b_); gt.append(t_)
        early = sum(len(x) for x in gt[:-1])
        gt = np.concatenate(gt); gb = np.concatenate(gb)
        assert early >= len(gt) - 2 - (n - edges[-2]) // (240 * spc), "hits were held back until the flush"
        assert np.array_equal(gt, wt), "streamed tags differ (cuts %s)" % (cuts,)
        assert np.array_equal(u32(gb), u32(wb)), "streamed bursts differ"
    # after the flush the stream starts over: the same input again gives the same hits (item 0 again, no old tags)
    b1, t1 = blk.general_work(obb, oavg, flush=True)
    b0, t0 = blk.work(obb, oavg)
    assert np.array_equal(t1, t0) and np.array_equal(u32(b1), u32(b0))
    # a half-fed stream is dropped by reset()
    blk.general_work(obb[: n // 2], oavg[: n // 2])
    blk.reset()
    b1, t1 = blk.general_work(obb, oavg, flush=True)
    assert np.array_equal(t1, t0)
    # arrays too small: AM_ECAPACITY says how many, the hits wait for am_fetch_tags
    import ctypes as C
    cx = blk._ctx
    got = C.c_uint64(0)
    a_, b_ = np.ascontiguousarray(obb, np.float32), np.ascontiguousarray(oavg, np.float32)
    rc = lib.L.am_preamble_stream(cx._h, a_.ctypes.data, b_.ctypes.data, n, _capi.AM_F_FLUSH, None, None, 0, C.byref(got))
    assert rc == _capi.AM_ECAPACITY and got.value == len(t0)
    fb, ft = cx.fetch_tags()
    assert np.array_equal(ft, t0) and np.array_equal(u32(fb), u32(b0))
    return len(wt)


def check_multi_streams(lib, rate, lengths, lam, seed, thr=7.0, pmf=True):
    """K independent streams in ONE scan (am_process_multi) = K calls of process_iq(stream, flush=True), bit for bit, and = the
    oracle on every stream: streams of different lengths, an empty one, one shorter than a burst, one that is all noise, bursts that
    end exactly at a stream's end / start at its first sample; gaps zeroed by the caller and by the library."""
    spc = max(int(rate / 2e6), 1)
    rng = np.random.default_rng(seed)
    streams = []
    for j, n in enumerate(lengths):
        if n < 300 * spc:
            streams.append((rng.standard_normal(2 * n).astype(np.float32) * 0.01).view(np.complex64))
            continue
        iq, _ = synth.synth_capture(rate, n, lam if j != 2 else 0.0, seed + 17 * j)
        iq = np.array(iq)
        # a strong burst whose first pulse is the stream's first sample, and one that ends j chips before the stream does (around
        # the end-of-buffer rule: some of these are emitted, some are not -- and their tails reach into the gap)
        frame = synth.make_frame(rng, 17)
        b = (np.repeat(synth.frame_chips(frame), spc) * np.float32(0.4)).astype(np.complex64)
        if len(b) < n // 4:
            iq[:len(b)] += b
            e1 = n - j * spc - (j % 2)
            iq[e1 - len(b):e1] += b * np.complex64(1j)
        streams.append(iq)
    ctx = _capi.Context(rate, thr, pmf, lib=lib)
    want = [ctx.process_iq(x, flush=True) for x in streams]
    whole = float(rate) == 2e6 * spc
    if whole:
        for x, w in zip(streams, want):
            assert np.array_equal(oracle.demod(x, rate, thr, pmf), w)
    buf, n = ctx.multi_pack(streams)
    got = ctx.process_multi(buf, n)
    assert len(got) == len(streams)
    for j, (g, w) in enumerate(zip(got, want)):
        assert g.tobytes() == w.tobytes(), "stream %d of %d differs: %d vs %d packets" % (j, len(streams), len(g), len(w))
    # garbage between the streams, zeroed by the library
    off, total = ctx.multi_layout(n)
    dirty = buf.copy()
    for j in range(len(streams) - 1):
        a, b_ = int(off[j] + n[j]), int(off[j + 1])
        dirty[2 * a:2 * b_] = 3.0
    got2 = ctx.process_multi(dirty, n, zero_gaps=True)
    assert all(g.tobytes() == w.tobytes() for g, w in zip(got2, want))
    # a too small receive array: nothing is lost; the context is a plain single-stream receiver again afterwards
    got3 = ctx.process_multi(buf, n, capacity=1)
    assert all(g.tobytes() == w.tobytes() for g, w in zip(got3, want))
    assert ctx.process_iq(streams[0], flush=True).tobytes() == want[0].tobytes()
    # the two halves (am_submit_multi / am_collect / am_multi_counts), two contexts used alternately by this one thread
    ctx2 = _capi.Context(rate, thr, pmf, lib=lib)
    half = max(1, len(streams) // 2)
    bufa, na = ctx.multi_pack(streams[:half])
    bufb, nb_ = ctx2.multi_pack(streams[half:] or streams[:1])
    ctx.submit_multi(bufa, na)
    ctx2.submit_multi(bufb, nb_)
    try:
        ctx.submit_multi(bufa, na)                          # not collected yet
        raise AssertionError("second submit accepted")
    except _capi.AirModesError:
        pass
    ga = ctx.collect_multi(capacity=1)                      # (too small: am_fetch_packets serves it)
    ctx.submit_multi(bufa, na)
    gb = ctx2.collect_multi()
    ga2 = ctx.collect_multi()
    wb = want[half:] or want[:1]
    assert all(g.tobytes() == w.tobytes() for g, w in zip(ga, want[:half])) and len(ga) == half
    assert all(g.tobytes() == w.tobytes() for g, w in zip(ga2, want[:half]))
    assert all(g.tobytes() == w.tobytes() for g, w in zip(gb, wb)) and len(gb) == len(wb)
    assert ctx.process_iq(streams[0], flush=True).tobytes() == want[0].tobytes()
    ctx2.close()
    # scans of K streams in flight behind ONE handle (am_pipe_submit_multi / am_pipe_collect / am_pipe_multi_counts): the two
    # layouts alternately, three in flight, mixed with a plain single-stream batch
    emulated = bool(getattr(lib, "emulated", False))
    if emulated:
        keep = [np.ascontiguousarray(bufa), np.ascontiguousarray(bufb), np.ascontiguousarray(streams[0]).view(np.float32)]
        ptrs = [k.ctypes.data for k in keep]
    else:
        import torch
        keep = [torch.from_numpy(np.ascontiguousarray(x).view(np.float32).copy()).cuda() for x in (bufa, bufb, streams[0])]
        ptrs = [k.data_ptr() for k in keep]
    pipe = _capi.Pipe(rate, thr, pmf, device=(-1 if emulated else 0), depth=3, lib=lib)
    order = [0, 1, 2, 0, 1, 1, 0]
    outs = []
    for i, which in enumerate(order):
        if pipe.in_flight() == pipe.depth():
            outs.append(pipe.collect_multi() if isinstance(pipe._held[0], tuple) else [pipe.collect()])
        if which == 2:
            pipe.submit_device(ptrs[2], len(streams[0]))
        else:
            pipe.submit_multi_device(ptrs[which], (na, nb_)[which])
    while pipe.in_flight():
        outs.append(pipe.collect_multi() if isinstance(pipe._held[0], tuple) else [pipe.collect()])
    assert len(outs) == len(order)
    for which, got_ in zip(order, outs):
        w_ = (want[:half], wb, want[:1])[which]
        assert len(got_) == len(w_) and all(g.tobytes() == w.tobytes() for g, w in zip(got_, w_)), "scan of layout %d in the pipe differs" % which
    pipe.close()
    ctx.close()
    return sum(len(w) for w in want)


def check_streamed_preamble_many_rx_time_tags(lib):
    """A long block-level stream with more "rx_time" tags than the context's table holds at once (4096): tags no future hit can
    refer to are dropped as the scan moves on (am_set_rx_time prunes by the streamed block's position too); the oracle, which takes
    all of them at once, is the witness."""
    import air_modes
    rate, n, spc = 2e6, 160000, 1
    iq, _ = synth.synth_capture(rate, n, 4000.0, 321)
    bb, avg = oracle.frontend(iq, spc, True)
    rx = [(k * 32, 1600000000 + k, 0.25) for k in range(n // 32)]         # 5 000 tags
    assert len(rx) > 4096
    ob, ot = oracle.preamble_scan(bb, avg, spc, 7.0, rate, rx_time=rx)
    blk = air_modes.preamble(rate, 7.0, lib=lib)
    gt = []
    step = 4000
    for a in range(0, n, step):
        b = min(n, a + step)
        _, t = blk.general_work(bb[a:b], avg[a:b], rx_time=[x for x in rx if a <= x[0] < b])
        gt.append(t)
    _, t = blk.general_work(bb[:0], avg[:0], flush=True)
    gt.append(t)
    gt = np.concatenate(gt)
    assert len(gt) > 100 and np.array_equal(gt, ot)
    return len(gt)


def load_rx_time_golden(path):
    z = np.load(path)
    rx = [(int(o), int(s_), float(f)) for o, s_, f in zip(z["rx_offset"], z["rx_secs"], z["rx_frac"])]
    return dict(iq=z["iq"], rate=float(z["rate"]), thr=float(z["thr_db"]), rx=rx, ref_sample=z["ref_tag_sample"],
                ref_secs=z["ref_tag_secs"], ref_frac=z["ref_tag_frac"], ref_msgs=[str(m) for m in z["ref_msgs"]])


def check_rx_time_golden(lib, path):
    """Product path under "rx_time" tags vs what the REFERENCE's own preamble_impl.cc (tag_to_timestamp)
    and slicer_impl.cc produced (tests/golden/rxtime_*.npz, tools/gen_golden.py rx_time)."""
    g = load_rx_time_golden(path)
    spc = int(g["rate"] / 2e6)
    ctx = _capi.Context(g["rate"], g["thr"], True, lib=lib)
    bb, avg = ctx.frontend_work(g["iq"])
    for tag in g["rx"]:
        ctx.set_rx_time(*tag)
    _, tags = ctx.preamble_work(bb, avg)
    assert np.array_equal(tags["sample"], g["ref_sample"])
    assert np.array_equal(tags["secs"], g["ref_secs"]) and np.array_equal(tags["frac"], g["ref_frac"])
    ctx.reset()
    n = len(g["iq"])
    parts = []
    for a, b in [(0, n // 2), (n // 2, n)]:
        for tag in g["rx"]:
            if a <= tag[0] < b:
                ctx.set_rx_time(*tag)
        parts.append(ctx.process_iq(g["iq"][a:b], flush=(b == n)))
    assert messages(lib, np.concatenate(parts)) == g["ref_msgs"]
    ctx.close()


def check_rx_time(lib, rate, n, lam, seed, thr=7.0, pmf=True, G=3):
    """"rx_time" stream tags (lib/preamble_impl.cc:100-137,165-170): a tag at the stream start, one that
    makes the fractional part roll over, two on the same item (the later wins), one on the very last
    items; block level, streaming (tags handed over with the chunk they fall into, and all in advance)
    and time-sharded.  Returns the number of distinct whole seconds seen (the tags must matter)."""
    spc = int(rate / 2e6)
    iq, _ = synth.synth_capture(rate, n, lam, seed)
    rx = [(0, 1600000000, 0.125), (n // 3 + 5, 1600000007, 0.9999995), (2 * n // 3, 3, 0.5),
          (2 * n // 3, 1700000000, 0.75), (n - 100 * spc, 9, 0.0)]
    want = oracle.demod(iq, rate, thr, pmf, rx_time=rx)
    plain = oracle.demod(iq, rate, thr, pmf)
    assert np.array_equal(want["sample"], plain["sample"]) and not np.array_equal(want["secs"], plain["secs"])
    # block level
    ctx = _capi.Context(rate, thr, pmf, lib=lib)
    obb, oavg = oracle.frontend(iq, spc, pmf)
    for tag in rx:
        ctx.set_rx_time(*tag)
    bursts, tags = ctx.preamble_work(obb, oavg)
    ob, ot = oracle.preamble_scan(obb, oavg, spc, thr, rate, rx_time=rx)
    assert np.array_equal(tags, ot), "tags differ"
    ctx.reset()                                              # drops the tags with the stream state
    assert np.array_equal(ctx.process_iq(iq, flush=True), plain)
    # streaming: each tag arrives with its chunk
    ctx.reset()
    edges = [0, n // 5, n // 3 + 5, n // 2, 2 * n // 3 + 7, n]
    parts = []
    for a, b in zip(edges[:-1], edges[1:]):
        for tag in rx:
            if a <= tag[0] < b:
                ctx.set_rx_time(*tag)
        parts.append(ctx.process_iq(iq[a:b], flush=(b == n)))
    assert np.array_equal(np.concatenate(parts), want), "streaming with rx_time differs"
    # streaming: all tags known in advance, other chunking
    ctx.reset()
    for tag in rx:
        ctx.set_rx_time(*tag)
    try:                                                     # offsets must not go backwards
        ctx.set_rx_time(5, 1, 0.0)
        raise AssertionError("backwards rx_time accepted")
    except _capi.AirModesError:
        pass
    got = np.concatenate([ctx.process_iq(iq[:n // 7]), ctx.process_iq(iq[n // 7:], flush=True)])
    assert np.array_equal(got, want)
    assert messages(lib, got) == oracle.format_messages(want)
    assert np.array_equal(ctx.process_iq(iq, flush=True), plain)      # the end of a stream drops its tags
    ctx.close()
    # time-sharded: every rank's context holds the same tags
    ctxs = [_capi.Context(rate, thr, pmf, lib=lib) for _ in range(G)]
    for c in ctxs:
        for tag in rx:
            c.set_rx_time(*tag)
    check_sharded(lib, rate, iq, G, thr, pmf, want=want, ctxs=ctxs)
    for c in ctxs:
        c.close()
    return len(np.unique(want["secs"]))


def edge_inputs(rate, seed=5):
    """Edge cases: empty, shorter than one burst, exactly at the room limit, all zeros,
    huge / tiny / non-finite samples."""
    spc = int(rate / 2e6)
    rng = np.random.default_rng(seed)
    base, _ = synth.synth_capture(rate, 6000 * spc, 3e6 / (240 * spc) * 0.3 * spc, seed)
    cases = {
        "empty": np.zeros(0, np.complex64),
        "one_sample": base[:1],
        "short": base[:100 * spc],
        "just_under_burst": base[:242 * spc - 1],
        "zeros": np.zeros(3000 * spc, np.complex64),
        "normal": base,
        "ragged_len": base[:5000 * spc + (spc // 2 + 1)],
    }
    big = base * np.complex64(3e18)          # |.|^2 overflows to inf in places
    cases["overflow"] = big
    tiny = base * np.complex64(1e-22)        # |.|^2 is denormal / underflows
    cases["denormal"] = tiny
    nan = base.copy()
    idx = rng.integers(0, len(nan), 25)
    nan[idx] = np.complex64(complex(float("nan"), 1.0))
    nan[rng.integers(0, len(nan), 25)] = np.complex64(complex(float("inf"), -1.0))
    cases["nan_inf"] = nan
    return cases


def nonfinite_stream(rate, n, seed=17, lam=15000.0):
    """A stream many tiles / steps long with NaN, +-inf, huge and denormal samples sprinkled over it:
    interior tiles of the fused kernels (their fast bodies, the EXEC-narrowing compares) see them too."""
    iq, _ = synth.synth_capture(rate, n, lam, seed)
    iq = iq.copy()
    rng = np.random.default_rng(seed)
    k = max(8, n // 4000)
    iq[rng.integers(0, n, k)] = np.complex64(complex(float("nan"), 0.5))
    iq[rng.integers(0, n, k)] = np.complex64(complex(float("inf"), -1.0))
    iq[rng.integers(0, n, k)] = np.complex64(complex(-2.0, float("-inf")))
    iq[rng.integers(0, n, k)] *= np.complex64(3e18)        # |.|^2 overflows
    idx = rng.integers(0, n - 600, k)
    for i in idx[:k // 2]:
        iq[i:i + 500] *= np.complex64(1e-22)               # denormal stretches
    return iq


def check_front_ends_agree(lib, rate, iq, monkeypatch, thr=7.0, pmf=True, expect_streaming=True):
    """The streaming kernel (default where it exists) and the tile kernel (AIRMODES_FE=2) against the oracle."""
    want = oracle.demod(iq, rate, thr, pmf)
    ctx = _capi.Context(rate, thr, pmf, lib=lib)
    got = ctx.process_iq(iq, flush=True)
    fe = ctx.last_frontend()
    ctx.close()
    if got.tobytes() != want.tobytes():                    # (bytes: a NaN reference level is equal to itself here)
        m = min(len(got), len(want))
        d = [i for i in range(m) if got[i].tobytes() != want[i].tobytes()]
        raise AssertionError("default front end (%d) differs from the oracle: %d vs %d packets, first differences at %r: "
                             "want %r got %r" % (fe, len(got), len(want), d[:4], [want[i] for i in d[:2]], [got[i] for i in d[:2]]))
    if expect_streaming:
        assert fe == 3, "the streaming front end did not run (front end %d)" % fe
    monkeypatch.setenv("AIRMODES_FE", "2")
    ctx = _capi.Context(rate, thr, pmf, lib=lib)
    got2 = ctx.process_iq(iq, flush=True)
    fe2 = ctx.last_frontend()
    ctx.close()
    monkeypatch.delenv("AIRMODES_FE")
    assert fe2 == 2 and got2.tobytes() == want.tobytes(), "tile kernel differs from the oracle"
    return len(want)


# ---- crafted bursts for the slicer / framer / CRC (lib/slicer_impl.cc:67-100, 128-182) ------------------------------
EDGE_DFS = (0, 4, 5, 11, 16, 17, 18, 19, 20, 21, 24, 27, 31)


def edge_bursts(n, seed):
    """n bursts of 240 soft chips + tags, aimed at the reference slicer's odd rules: every downlink format incl. the ones it
    takes for short although they are long on the air (DF18 / 19 / 24..31, :140) and DF16 long; chips exactly AT the 3 dB
    limits `lo` / `hi` (strict comparisons, :74-75) and at half the lower limit (:92,:95: a double comparison), one float
    next to them on either side; 0, 9 and 10 low-confidence bits on DF11 (:171), one on the other short formats (:170), 24
    and more on long ones (the list saturates, :157); all-zero payloads (:162-166); non-finite and negative chips; valid
    and damaged parity fields.  Reference levels over six decades (the message prints them: iostream %g formatting)."""
    import synth
    rng = np.random.default_rng(seed)
    bursts = np.zeros((n, 240), np.float32)
    tags = np.zeros(n, np.dtype([("sample", "<u8"), ("secs", "<u8"), ("frac", "<f8"), ("inavg", "<f4"), ("how_late", "<u4")]))
    tags["secs"] = rng.integers(0, 1 << 33, n)
    tags["frac"] = rng.random(n)
    tags["sample"] = np.arange(n, dtype=np.uint64) * 4801
    f32 = np.float32
    for i in range(n):
        kind = int(rng.integers(0, 12))
        scale = f32(10.0 ** rng.uniform(-4, 2))
        pre = (f32(1.0) + rng.uniform(-0.2, 0.2, 4).astype(f32)) * scale
        b = bursts[i]
        b[:16] = (rng.uniform(0.0, 0.1, 16).astype(f32) * scale)
        b[[0, 2, 7, 9]] = pre
        # the reference level exactly as slicer_impl.cc:128-131 forms it, and the limits of :71-72
        ref = f32(np.float64(f32(f32(f32(pre[0] + pre[1]) + pre[2]) + pre[3])) / 4.0)
        hi = f32(np.float64(ref) * 1.414)
        lo = f32(np.float64(ref) * 0.707)
        half = f32(np.float64(lo) * 0.5)
        df = int(EDGE_DFS[int(rng.integers(0, len(EDGE_DFS)))])
        frame = bytearray(synth.make_frame(rng, df if df in (16, 17, 20, 21) else (df if df < 24 else 0)))
        if len(frame) == 7:
            frame += bytes(rng.integers(0, 256, 7, dtype=np.uint8).tobytes())   # what follows a "short" frame on the air
        frame[0] = ((df & 0x1F) << 3) | (frame[0] & 7)
        if df not in (16, 17, 20, 21) and rng.random() < 0.7:
            # a parity field the reference can check for the length IT assumes (56 bits)
            par = synth._crc24(bytes(frame[:4]))
            if df != 11 or rng.random() < 0.3:
                par ^= int(rng.integers(0, 1 << 24)) if df != 11 else int(rng.integers(0, 2))
            frame[4:7] = par.to_bytes(3, "big")
        if kind == 0:
            frame = bytearray(14)                                  # all zero: tossed whatever else holds (:162-166)
        bits = np.unpackbits(np.frombuffer(bytes(frame), np.uint8))
        strong = ref * (f32(1.0) + rng.uniform(-0.15, 0.15, 112).astype(f32))
        weak = half * rng.uniform(0.0, 0.9, 112).astype(f32)
        c0 = np.where(bits == 1, strong, weak).astype(f32)
        c1 = np.where(bits == 1, weak, strong).astype(f32)
        palette = np.array([lo, hi, np.nextafter(lo, f32(np.inf)), np.nextafter(lo, f32(-np.inf)), np.nextafter(hi, f32(np.inf)),
                            np.nextafter(hi, f32(-np.inf)), half, np.nextafter(half, f32(np.inf)), np.nextafter(half, f32(-np.inf)),
                            f32(0.0), f32(-0.0), -ref, ref, f32(2.0) * ref, f32(np.inf), f32(-np.inf), f32(np.nan),
                            f32(1e-40), f32(3e38)], np.float32)
        nbits_ref = 112 if df in (16, 17, 20, 21) else 56

        def lowconf(j):                                            # both chips inside the limits: decided, not trusted (:84-87)
            a, d = ref * f32(1.05), ref * f32(0.95)
            c0[j], c1[j] = (a, d) if bits[j] else (d, a)
        if kind in (1, 2, 3):                                      # a chosen number of low-confidence bits
            want = {1: int(rng.integers(1, 3)), 2: int(rng.choice([8, 9, 10, 11])), 3: int(rng.integers(22, 40))}[kind]
            for j in rng.choice(np.arange(5, nbits_ref), min(want, nbits_ref - 5), replace=False):
                lowconf(int(j))
        elif kind in (4, 5, 6):                                    # chips at / next to the limits, a few or many
            m = {4: 2, 5: 12, 6: 60}[kind]
            js = rng.integers(0, 112, m)
            c0[js] = palette[rng.integers(0, len(palette), m)]
            js = rng.integers(0, 112, m)
            c1[js] = palette[rng.integers(0, len(palette), m)]
        elif kind == 7:                                            # the header bits themselves on the limits
            js = rng.integers(0, 5, 3)
            c0[js] = palette[rng.integers(0, 9, 3)]
            c1[js] = palette[rng.integers(0, 9, 3)]
        elif kind == 8:                                            # noise-like: neither chip near the reference
            c0 = (rng.uniform(0, 3, 112).astype(f32) * ref).astype(f32)
            c1 = (rng.uniform(0, 3, 112).astype(f32) * ref).astype(f32)
        b[16::2] = c0
        b[17::2] = c1
        if kind == 9:                                              # a non-finite or zero reference level
            b[[0, 2, 7, 9][int(rng.integers(0, 4))]] = [f32(np.inf), f32(np.nan), f32(0.0), f32(-1.0) * scale][int(rng.integers(0, 4))]
    return bursts, tags


def check_slicer_edge_vectors(lib, n, seed, with_ref=False):
    """pc.edge_bursts through the block-level slicer of the library (am_slicer_work -> am_k_slice / am_slice_wave) against the
    oracle and, where oracle/_ref exists, the reference's own slicer_impl::work."""
    bursts, tags = edge_bursts(n, seed)
    ctx = _capi.Context(4e6, 7.0, True, lib=lib)
    got = ctx.slicer_work(bursts, tags)
    want = oracle.slice_bursts(bursts, tags)
    assert got.tobytes() == want.tobytes(), "packets differ"       # (bytes: a NaN reference level must compare equal to itself)
    texts = lib.format_messages(got, True)
    assert texts == oracle.format_messages(want)
    if with_ref and oracle.have_ref():
        rtext, acc = oracle.ref_slice_bursts(bursts, tags)
        assert texts == rtext and int(acc.sum()) == len(got)
    return len(got)


def check_framer_edge_formats(lib, rate, n, lam, seed, want_fe=None):
    """DF16 / 18 / 19 / 24 on the air (synth.DF_MIX_EDGE) through the PRODUCTION path -- front end, refinement, chain,
    am_k_extract_slice_iq + am_slice_wave -- at stage level: tags, bursts, packets against the oracle and the reference's C++."""
    iq, truth = synth.synth_capture(rate, n, lam, seed, df_mix=synth.DF_MIX_EDGE, snr_db=(14.0, 35.0))
    npk = check_production_stages(lib, rate, n, lam, seed, iq=iq, with_ref=True, want_fe=want_fe)
    want = oracle.demod(iq, rate, 7.0, True)
    dfs = set(int(d) for d in want["df"])
    assert {16, 18, 19, 24} <= dfs, dfs
    assert set(want["nbytes"][np.isin(want["df"], (18, 19, 24))]) == {7} and set(want["nbytes"][want["df"] == 16]) == {14}
    return npk


def check_stream_pipe(lib, rate, n, lam, seed, depth=3, thr=7.0, pmf=True, dcblock=False, contiguous=True, min_chunk=None, rx_time=None,
                      device=None):
    """ONE continuing stream with `depth` consecutive chunks in flight (am_spipe: VERDICT r5 #3, lib/preamble_impl.cc:139-246 is a
    streaming block): random chunk sizes; the stream contiguous in memory, or every chunk in a buffer of its own with room for the
    previous chunk's tail in front of it; packets of all chunks == the oracle over the WHOLE stream (item counts and time stamps
    continue), == am_process_iq over the same cuts.  device: a torch device (the chunks then live in device memory); None: the
    emulation, where device memory is host memory.  Returns (packets, chunks redone on the synchronous path)."""
    rng = np.random.default_rng(seed)
    iq, _ = synth.synth_capture(rate, n, lam, seed)
    pipe = _capi.StreamPipe(rate, thr, pmf, use_dcblock=dcblock, depth=depth, lib=lib, device=(-1 if device is None else device.index or 0))
    front = pipe.front()
    lo = max(min_chunk or 0, front + 1)
    cuts, at = [], 0
    while n - at > 2 * lo + 2:
        step = int(rng.integers(lo, max(lo + 1, min(n - at - lo, 6 * lo))))
        at += step
        cuts.append(at)
    edges = [0] + cuts + [n]
    want = oracle.demod(iq, rate, thr, pmf, use_dcblock=dcblock, rx_time=rx_time)
    for tag in rx_time or []:
        pipe.set_rx_time(*tag)
    f32 = iq.view(np.float32)
    keep = []
    if device is None:
        if contiguous:
            base = np.ascontiguousarray(f32)
            keep.append(base)
            chunks = [(base.ctypes.data + 8 * a, b - a) for a, b in zip(edges[:-1], edges[1:])]
        else:
            chunks = []
            for a, b in zip(edges[:-1], edges[1:]):
                buf = np.full(2 * (front + (b - a)), np.float32(7.0))      # (garbage in front: the library puts the tail there)
                buf[2 * front:] = f32[2 * a:2 * b]
                keep.append(buf)
                chunks.append((buf.ctypes.data + 8 * front, b - a))
    else:
        import torch
        if contiguous:
            base = torch.from_numpy(np.ascontiguousarray(f32)).to(device)
            keep.append(base)
            chunks = [(base.data_ptr() + 8 * a, b - a) for a, b in zip(edges[:-1], edges[1:])]
        else:
            chunks = []
            for a, b in zip(edges[:-1], edges[1:]):
                buf = torch.full((2 * (front + (b - a)),), 7.0, dtype=torch.float32, device=device)
                buf[2 * front:].copy_(torch.from_numpy(f32[2 * a:2 * b].copy()))
                keep.append(buf)
                chunks.append((buf.data_ptr() + 8 * front, b - a))
        torch.cuda.synchronize()
    got = pipe.run(chunks)
    assert len(got) == len(chunks)
    allp = np.concatenate(got) if got else np.zeros(0, _capi.PACKET_DTYPE)
    assert np.array_equal(allp, want), "stream pipe differs from the oracle over the whole stream: %d vs %d packets (%d chunks, depth %d)" % (
        len(allp), len(want), len(chunks), depth)
    # the same cuts through am_process_iq, chunk by chunk: the same packets in the same chunks?  (am_process_iq holds the last 244
    # samples per chip of a chunk back, exactly as a pipe chunk leaves them to its successor)
    ctx = _capi.Context(rate, thr, pmf, use_dcblock=dcblock, lib=lib, device=(-1 if device is None else device.index or 0))
    for tag in rx_time or []:
        ctx.set_rx_time(*tag)
    parts = [ctx.process_iq(iq[a:b], flush=(b == n)) for a, b in zip(edges[:-1], edges[1:])]
    assert np.array_equal(np.concatenate(parts), want)
    ctx.close()
    # a second stream through the same pipe: everything starts over at sample 0 (the end of a stream drops its rx_time tags)
    for tag in rx_time or []:
        pipe.set_rx_time(*tag)
    got2 = pipe.run(chunks)
    assert np.array_equal(np.concatenate(got2), want), "the second stream through the pipe differs"
    redone = pipe.redone()
    pipe.close()
    del keep
    return allp, redone
