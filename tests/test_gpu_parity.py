"""Parity tests proper: the real libairmodes_hip.so on a real MI355X, through the C ABI,
bit-for-bit against the oracle and the reference-generated golden vectors.  BASELINE.json
configs 2 (2 Msps capture), 3 (64 Msps), 4 (time-sharded) and 5 (20 Msps streams) at full
size are covered here, plus size-independent properties (chunking / sharding invariance)."""
import os

import numpy as np
import pytest

import oracle
import parity_common as pc
import synth
from air_modes import _capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib(hip_lib):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return hip_lib


@pytest.fixture(scope="module")
def klib(hip_knobs_lib):
    """test-only build with the environment knobs (conftest.hip_knobs_lib): for the tests that steer kernels"""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return hip_knobs_lib


@pytest.mark.parametrize("path", pc.golden_cases(), ids=lambda p: os.path.basename(p)[:-4])
def test_golden(lib, path):
    pc.check_golden(lib, path)


@pytest.mark.parametrize("rate,n,lam,seed,pmf", [(2e6, 1000000, 1500.0, 31, True), (4e6, 1000000, 1500.0, 32, True),
                                                 (4e6, 500000, 2000.0, 33, False), (8e6, 1000000, 2000.0, 37, True),
                                                 (10e6, 1000000, 2000.0, 38, True), (16e6, 2000000, 3000.0, 39, True),
                                                 (20e6, 3000000, 5000.0, 34, True), (32e6, 3000000, 5000.0, 40, True),
                                                 (40e6, 3000000, 5000.0, 44, True),
                                                 (64e6, 6400000, 20000.0, 35, True), (100e6, 3000000, 10000.0, 36, True)])
def test_stages(lib, rate, n, lam, seed, pmf):
    assert pc.check_stages(lib, rate, n, lam, seed, pmf=pmf) > 5


@pytest.mark.parametrize("rate,n,lam", [(5e6, 2_000_000, 2500.0), (6.25e6, 2_500_000, 3000.0), (4.8e6, 1_500_000, 2500.0),
                                        (13e6, 3_000_000, 4000.0)])
def test_fractional_samples_per_chip(lib, rate, n, lam):
    """Rates that are not multiples of 2 MHz on the device (the reference's float geometry, lib/preamble_impl.cc:57,150,
    158-162,185,192,205-208,212,220,237): block by block and end to end against the oracle, every candidate record against
    the oracle and the reference's own C++, chunked, time-sharded; and the reference-generated golden vectors."""
    assert pc.check_stages(lib, rate, n, lam, 61) > 20
    assert pc.check_production_stages(lib, rate, n, lam, 62, with_ref=True, want_fe=1) > 20
    iq, _ = synth.synth_capture(rate, n, lam, seed=63)
    pc.check_chunked(lib, rate, iq, [n // 5 + 1, n // 2, n // 2 + 7, n - 997])
    assert pc.check_sharded(lib, rate, iq, 3) > 20


def test_tiled_fused_kernel_still_matches(klib, lib, monkeypatch):
    """The tile kernel am_k_fe2 (dense bb / reference level for the block-level API) lives in the TEST builds only since round 5
    (-DAM_WITH_TILE_KERNEL); the product library serves am_frontend_work from the rate-generic kernel: both, stage by stage."""
    for rate, n in ((16e6, 2000000), (20e6, 2000000), (64e6, 6000000)):
        assert pc.check_stages(klib, rate, n, 6000.0, 51) > 3
    assert pc.check_stages(lib, 64e6, 3000000, 6000.0, 52) > 3


def test_generic_kernels_still_match(klib, monkeypatch):
    lib = klib
    monkeypatch.setenv("AIRMODES_GENERIC", "1")
    for rate, n in ((2e6, 1000000), (20e6, 2000000), (64e6, 4000000)):
        assert pc.check_stages(lib, rate, n, 6000.0, 41) > 3
    monkeypatch.delenv("AIRMODES_GENERIC")
    assert pc.check_stages(lib, 6e6, 1000000, 3000.0, 42) > 3
    assert pc.check_stages(lib, 50e6, 2000000, 8000.0, 43) > 3


@pytest.mark.parametrize("rate", [2e6, 4e6, 8e6, 10e6, 16e6, 20e6, 32e6, 40e6, 64e6])
def test_edge_inputs(lib, rate):
    for name, iq in pc.edge_inputs(rate).items():
        want = oracle.demod(iq, rate)
        ctx = _capi.Context(rate, 7.0, True, lib=lib)
        got = ctx.process_iq(iq, flush=True)
        assert np.array_equal(got, want), name
        if len(iq):
            bb, avg = ctx.frontend_work(iq)
            obb, oavg = oracle.frontend(iq, int(rate / 2e6), True)
            assert np.array_equal(pc.u32(bb), pc.u32(obb)), name
            assert np.array_equal(pc.u32(avg), pc.u32(oavg)), name
        ctx.close()


@pytest.fixture(scope="module")
def capture_2msps():
    """BASELINE config 1/2 stand-in: 10 s at 2 Msps, seed 1090."""
    rate, (iq, truth) = synth.config_capture("2msps")
    return rate, iq, oracle.demod(iq, rate)


def test_config2_2msps_capture_bit_exact(lib, capture_2msps):
    rate, iq, want = capture_2msps
    ctx = _capi.Context(rate, 7.0, True, lib=lib)
    got = ctx.process_iq(iq, flush=True)
    assert len(want) > 1000
    assert np.array_equal(got, want)
    assert pc.messages(lib, got) == oracle.format_messages(want)
    ctx.close()
    pc.check_chunked(lib, rate, iq, [1, 4097, 5000000, 5000001, 12345678], want=want)


@pytest.fixture(scope="module")
def capture_64msps():
    """BASELINE config 3 at full size: 1 s at 64 Msps, Poisson 20 000 bursts/s, seed 6400."""
    rate, (iq, truth) = synth.config_capture("64msps")
    return rate, iq, oracle.demod(iq, rate)


def test_config3_64msps_full_size(lib, capture_64msps):
    import torch
    rate, iq, want = capture_64msps
    assert len(want) > 500
    ctx = _capi.Context(rate, 7.0, True, lib=lib)
    # device-resident input (the bench path)
    t = torch.from_numpy(iq.view(np.float32)).cuda()
    torch.cuda.synchronize()
    got = ctx.process_iq_device(t.data_ptr(), len(iq), flush=True)
    assert np.array_equal(got, want)
    # idempotence: same buffer again, same answer
    got2 = ctx.process_iq_device(t.data_ptr(), len(iq), flush=True)
    assert np.array_equal(got2, want)
    ctx.close()
    del t
    # chunking invariance at full size
    pc.check_chunked(lib, rate, iq, [20000000, 20000001, 47000000], want=want)


def test_config4_time_sharded_full_size(lib, capture_64msps):
    rate, iq, want = capture_64msps
    for G in (2, 8):
        pc.check_sharded(lib, rate, iq, G, want=want)


def test_config5_20msps_stream(lib):
    rate, (iq, truth) = synth.config_capture("20msps")
    want = oracle.demod(iq, rate)
    ctx = _capi.Context(rate, 7.0, True, lib=lib)
    got = ctx.process_iq(iq, flush=True)
    assert len(want) > 200 and np.array_equal(got, want)
    ctx.close()


def test_rx_path_facade_on_gpu(lib):
    import air_modes
    q = air_modes.msg_queue()
    rx = air_modes.rx_path(4e6, 7.0, q, use_pmf=True, lib=lib)
    iq, _ = synth.synth_capture(4e6, 2000000, 1500.0, seed=12)
    rx.work(iq[:700001])
    rx.work(iq[700001:], flush=True)
    got = []
    while not q.empty_p():
        got.append(q.delete_head().to_string())
    assert got == oracle.format_messages(oracle.demod(iq, 4e6)) and len(got) > 50


def test_modes_rx_cli_on_gpu(lib, tmp_path):
    """apps/modes_rx (file source) on the real library: raw messages == oracle's texts, and the
    printed report lines == what the message consumers make of those texts."""
    import io
    from air_modes import cpr, modes_rx, msprint, parse
    from air_modes.pubsub import pubsub
    rate = 4e6
    iq, _ = synth.synth_capture(rate, 3000000, 800.0, seed=4243)
    path = tmp_path / "cap.cf32"
    np.asarray(iq, dtype=np.complex64).tofile(path)
    raw = io.StringIO()
    assert modes_rx.main(["-s", str(path), "-r", "4e6", "--raw", "--chunk", "1000000"], out=raw) == 0
    want = oracle.format_messages(oracle.demod(iq, rate, 7.0, True))
    assert raw.getvalue().splitlines() == want and len(want) > 100
    parsed = io.StringIO()
    assert modes_rx.main(["-s", str(path), "-r", "4e6"], out=parsed) == 0
    pub, lines = pubsub(), []
    msprint.output_print(cpr.cpr_decoder(None), pub, callback=lines.append)
    feed = parse.make_parser(pub)
    for m in want:
        try:
            feed(m)
        except IndexError:              # a reference table bug the command line survives (see modes_rx.py)
            pass
    assert parsed.getvalue().splitlines() == lines and len(lines) > 20


@pytest.mark.parametrize("rate,n", [(2e6, 1000000), (4e6, 1000000), (20e6, 2000000), (64e6, 4000000), (100e6, 3000000)])
def test_dcblock_option(lib, rate, n):
    """rx_path(..., use_dcblock=True) (a2): stage by stage and end to end against the oracle's canonical
    definition on a capture with a DC offset; chunking and sharding invariance."""
    assert pc.check_stages(lib, rate, n, 2500.0, 71, dcblock=True, dc_offset=0.04) > 10
    iq, _ = synth.synth_capture(rate, n, 2500.0, 72)
    iq = (iq + np.complex64(0.03 - 0.02j)).astype(np.complex64)
    spc = int(rate / 2e6)
    pc.check_chunked(lib, rate, iq, [n // 3 + 1, n // 3 + 150 * spc, 2 * n // 3 + 7], dcblock=True)
    pc.check_sharded(lib, rate, iq, 4, dcblock=True)


@pytest.mark.parametrize("rate,n,lam", [(2e6, 2000000, 1500.0), (20e6, 3000000, 5000.0), (64e6, 6400000, 20000.0)])
def test_live_reference_cpp(lib, rate, n, lam):
    """The reference's OWN preamble_impl / slicer_impl / modes_crc (oracle/_ref: compiled by path from
    /root/reference in the build container, travels as a prebuilt .so) fed with the GPU front end's bb/avg,
    against the GPU's own preamble + slicer on the same streams: tags, bursts and message texts."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref/libairmodes_ref.so was not built (no /root/reference at build time)")
    spc = int(rate / 2e6)
    iq, _ = synth.synth_capture(rate, n, lam, seed=97)
    ctx = _capi.Context(rate, 7.0, True, lib=lib)
    bb, avg = ctx.frontend_work(iq)
    rb, rt, rmsgs, keep = oracle.ref_preamble_slicer(bb, avg, spc, 7.0, rate)
    nk = int(keep.sum())
    bursts, tags = ctx.preamble_work(bb, avg)
    assert nk == len(tags) > 20 and keep[:nk].all()
    assert np.array_equal(rt["sample"][:nk], tags["sample"]) and np.array_equal(rt["frac"][:nk], tags["frac"])
    assert np.array_equal(pc.u32(rb[:nk]), pc.u32(bursts))
    pk = ctx.slicer_work(bursts, tags)
    msgs = pc.messages(lib, pk)
    assert msgs == rmsgs[:len(msgs)] and len(msgs) > 10
    assert np.array_equal(ctx.process_iq(iq, flush=True), pk)
    ctx.close()


@pytest.mark.parametrize("path", pc.rx_time_golden_cases(), ids=os.path.basename)
def test_rx_time_reference_golden(lib, path):
    """"rx_time" stream tags: packets stamped by the GPU path vs the reference's own tag_to_timestamp."""
    pc.check_rx_time_golden(lib, path)


@pytest.mark.parametrize("rate,n", [(2e6, 1000000), (20e6, 2000000), (64e6, 6400000)])
def test_rx_time_tags(lib, rate, n):
    """am_set_rx_time: block level, streaming (tags arriving with their chunk / in advance), sharded."""
    assert pc.check_rx_time(lib, rate, n, 3000.0, 91, G=4) >= 3


@pytest.mark.parametrize("rate,n,lam", [(2e6, 600000, 3000.0), (4e6, 1000000, 3000.0), (5e6, 900000, 2500.0), (64e6, 6400000, 8000.0)])
def test_preamble_block_as_a_stream(lib, rate, n, lam):
    """preamble.general_work (am_preamble_stream): random pieces of the two input streams = one work() over them."""
    assert pc.check_preamble_stream(lib, rate, n, lam, seed=int(rate / 1e5) + 3, trials=3) > 3


@pytest.mark.parametrize("rate,lengths,lam", [(2e6, [600000, 0, 450000, 100, 800000, 523110, 700001, 640000], 3000.0),
                                              (5e6, [900000, 750003, 800000], 2500.0),
                                              (20e6, [4000000, 3000000, 3500001, 2500000, 777, 3200000, 2000000, 4100000], 6000.0),
                                              (64e6, [6400000, 5000001, 6500000], 12000.0)])
def test_k_streams_in_one_scan(lib, rate, lengths, lam):
    """am_process_multi: K whole streams in one buffer, one scan; every stream's packets = its own am_process_iq(AM_F_FLUSH)
    = the oracle's (streams of different lengths, empty, shorter than a burst, all noise; bursts at the very start / end)."""
    assert pc.check_multi_streams(lib, rate, lengths, lam, seed=int(rate / 1e5) + 11) > 3 * len(lengths) // 2


def test_streamed_preamble_block_drops_spent_rx_time_tags(lib):
    """5 000 "rx_time" tags through the streamed preamble block on the device (the table holds 4 096 at once)."""
    assert pc.check_streamed_preamble_many_rx_time_tags(lib) > 100


def test_chain_walk_beyond_64k_of_lds(lib):
    """Between 256 and 288 blocks of 2048 first-stage candidates in ONE scan (low threshold, dense traffic):
    the block-to-block walk of the greedy chain then keeps 128 head links per block = more than 64 KB in LDS,
    i.e. it depends on the raised dynamic-LDS limit of its kernel (beyond 288 blocks the heads get shorter)."""
    rate, n, thr = 8e6, 22000000, 2.0
    iq, _ = synth.synth_capture(rate, n, 40000.0, seed=123)
    ctx = _capi.Context(rate, thr, True, lib=lib)
    pk = ctx.process_iq(iq, flush=True)
    assert 256 * 2048 < ctx.last_num_candidates() <= 288 * 2048, ctx.last_num_candidates()
    want = oracle.demod(iq, rate, thr, True)
    assert np.array_equal(pk, want) and len(want) > 2000
    pc.check_sharded(lib, rate, iq, 2, thr=thr, want=want)
    ctx.close()


def test_greedy_chain_rare_branches_on_device():
    """The test-only build with 4-slot block heads, groups of two blocks and ticket-ordered chained scans
    (gr-air-modes_amd/csrc/Makefile, target `rare`): global-memory hops of the block walk, plain hops between groups, the
    atomic ticket of am_chain_place -- on the device, against the oracle; one stream, the 64 Msps streaming path, shards."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_variants", "libairmodes_hip_rare.so")
    if not os.path.exists(path):                              # (normally built by __graft_entry__.build())
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(path), "..", "..", "gr-air-modes_amd", "csrc"), "rare"])
    rare = _capi.Library(path)
    rate = 8e6
    iq, _ = synth.synth_capture(rate, 12000000, 30000.0, seed=606)
    want = oracle.demod(iq, rate, 5.0, True)
    ctx = _capi.Context(rate, 5.0, True, lib=rare)
    got = [ctx.process_iq(iq[:5000001], flush=False)]
    assert ctx.last_num_candidates() > 8 * 2048, ctx.last_num_candidates()
    got.append(ctx.process_iq(iq[5000001:], flush=True))
    ctx.close()
    assert np.array_equal(np.concatenate(got), want) and len(want) > 300
    assert pc.check_sharded(rare, rate, iq, 3, thr=5.0, want=want) == len(want)
    iq64, _ = synth.synth_capture(64e6, 16000000, 20000.0, seed=607)
    ctx = _capi.Context(64e6, 7.0, True, lib=rare)
    pk = ctx.process_iq(iq64, flush=True)
    assert ctx.last_num_candidates() > 16 * 2048
    ctx.close()
    assert np.array_equal(pk, oracle.demod(iq64, 64e6, 7.0, True)) and len(pk) > 100
    # the rare build runs 64 Msps through am_k_fe4<32, 1, 3> (-DFE4_64MSPS: three waves, all 64 lanes own a chip, 48-chip
    # blocks that straddle waves) instead of am_k_fe3: stage-level parity of that kernel on the device
    assert pc.check_production_stages(rare, 64e6, 12_000_000, 20000.0, 608, with_ref=True, want_fe=3) > 0
    assert pc.check_production_stages(rare, 64e6, 5_000_000, 20000.0, 609, chunks=[1_000_001, 3_300_000], want_fe=3) > 0


def test_streaming_and_tile_front_ends_agree(klib, monkeypatch):
    """64 Msps: am_k_fe3 (default) and am_k_fe2 (AIRMODES_FE=2) give the oracle's packets -- also without the
    pulse-matched filter and with NaN / inf / denormal samples in INTERIOR tiles and steps (the EXEC-narrowing
    compares of the fast bodies only exist on the device)."""
    lib = klib
    iq, _ = synth.synth_capture(64e6, 6000000, 20000.0, 77)
    assert pc.check_front_ends_agree(lib, 64e6, iq, monkeypatch) > 50
    assert pc.check_front_ends_agree(lib, 64e6, iq[:3000000], monkeypatch, thr=5.0, pmf=False) > 20
    bad = pc.nonfinite_stream(64e6, 4000000)
    pc.check_front_ends_agree(lib, 64e6, bad, monkeypatch)
    for rate in (40e6, 20e6, 16e6, 10e6, 8e6, 4e6, 2e6):      # am_k_fe4 (several chips per lane) and the tile kernel
        pc.check_front_ends_agree(lib, rate, pc.nonfinite_stream(rate, 1500000), monkeypatch)


def test_streaming_front_end_unaligned_and_short_inputs(klib, monkeypatch):
    lib = klib
    monkeypatch.setenv("AIRMODES_POISON", "1")     # NaN-fill the sparse arrays before every scan: no stale value can help
    iq, _ = synth.synth_capture(64e6, 5000000, 20000.0, 78)
    want = oracle.demod(iq, 64e6)
    assert pc.check_chunked(lib, 64e6, iq, [1, 3073, 1000001, 1000002, 2500001, 2500002 + 3071], want=want) > 50
    assert pc.check_chunked(lib, 64e6, iq, list(range(70001, 5000000, 70001)), want=want) > 50
    assert pc.check_sharded(lib, 64e6, iq, 5, want=want) > 50
    # the same for am_k_fe4 (several chips per lane): odd cuts = 8-byte aligned sources = guarded loads on every step,
    # chunks shorter than a step (3 840 samples at 20 Msps, 3 072 at 2 Msps), shards
    for rate, lam, seed in ((20e6, 8000.0, 79), (2e6, 1500.0, 80), (4e6, 2000.0, 81), (10e6, 4000.0, 82)):
        iq, _ = synth.synth_capture(rate, 2000000, lam, seed)
        want = oracle.demod(iq, rate)
        assert len(want) > 30
        assert pc.check_chunked(lib, rate, iq, [1, 3841, 500001, 500002, 1200001, 1200002 + 3071], want=want) == len(want)
        assert pc.check_chunked(lib, rate, iq, list(range(30001, 2000000, 30001)), want=want) == len(want)
        assert pc.check_sharded(lib, rate, iq, 3, want=want) == len(want)


def test_batches_in_flight_single_host_thread(lib):
    """am_pipe on the device: batches of different lengths and rates of traffic in flight, packets per batch == oracle."""
    rate = 64e6
    batches = [synth.synth_capture(rate, 3000000 + 100001 * k, 20000.0 if k % 2 else 3000.0, 500 + k)[0] for k in range(7)]
    want = [oracle.demod(b, rate) for b in batches]
    pipe = _capi.Pipe(rate, 7.0, True, depth=3, lib=lib)
    got = []
    for b in batches:
        if pipe.in_flight() == pipe.depth():
            got.append(pipe.collect())
        pipe.submit(b)
    while pipe.in_flight():
        got.append(pipe.collect())
    assert len(got) == len(want) and all(np.array_equal(g, w) for g, w in zip(got, want))
    pipe.close()


@pytest.mark.parametrize("rate,n,lam,fe", [(64e6, 64_000_000, 20000.0, 3), (64e6, 16_000_000, 2000.0, 3),
                                            (20e6, 20_000_000, 5000.0, 3), (2e6, 20_000_000, 500.0, 3),
                                            (4e6, 8_000_000, 1000.0, 3), (10e6, 4_000_000, 2000.0, 3),
                                            (8e6, 6_000_000, 2000.0, 3), (16e6, 6_000_000, 4000.0, 3),
                                            (32e6, 8_000_000, 8000.0, 3), (40e6, 8_000_000, 10000.0, 3)])
def test_production_stages_full_size(lib, rate, n, lam, fe):
    """stage-level parity of the kernels that actually run -- at the BASELINE sizes the
    record of every first-stage candidate (bitmap position, refined position, quiet-zone outcome, reference level), the
    bursts and tags handed to the slicer, the tag count and the packets, against the oracle and the reference's own C++."""
    assert pc.check_production_stages(lib, rate, n, lam, 6400, with_ref=True, want_fe=fe) > 0


def test_production_stages_chunked_and_no_pmf(lib):
    assert pc.check_production_stages(lib, 64e6, 9_000_000, 20000.0, 11, chunks=[2_000_001, 5_500_000], want_fe=3) > 0
    assert pc.check_production_stages(lib, 64e6, 6_000_000, 20000.0, 12, pmf=False, want_fe=3) > 0
    for rate, n in ((20e6, 5_000_000), (4e6, 2_000_000), (2e6, 2_000_000)):
        assert pc.check_production_stages(lib, rate, n, 3000.0, 13, chunks=[n // 3 + 1, n // 2], want_fe=3) > 0
        assert pc.check_production_stages(lib, rate, n // 2, 3000.0, 14, pmf=False, want_fe=3) > 0


def test_host_free_calls_refuse_pageable_message_buffers(lib):
    """am_shard_scan_async / am_shard_resolve_async write and read their message buffers on the device: handed pageable host
    memory (a numpy array) they return AM_EINVAL instead of faulting the GPU."""
    import torch
    rate, n = 4e6, 200_000
    iq, _ = synth.synth_capture(rate, n, 2000.0, 77)
    d = torch.from_numpy(iq.view(np.float32)).to("cuda:0")
    ctx = _capi.Context(rate, 7.0, True, lib=lib)
    words = 2 * (_capi.SHARD_MSG_HEADER + 512)
    host_msg = np.zeros(words, np.uint64)
    with pytest.raises(_capi.AirModesError) as ei:
        ctx.shard_scan_async(d.data_ptr(), 0, n, n, host_msg.ctypes.data, 512)
    assert ei.value.code == _capi.AM_EINVAL
    dev_msg = torch.zeros(words, dtype=torch.int64, device="cuda:0")
    ctx.shard_scan_async(d.data_ptr(), 0, n, n, dev_msg.data_ptr(), 512)          # (device memory: accepted)
    with pytest.raises(_capi.AirModesError) as ei:
        ctx.shard_resolve_async(host_msg.ctypes.data, 1, 0, 512)
    assert ei.value.code == _capi.AM_EINVAL
    pk, redo = ctx.shard_resolve_async(dev_msg.data_ptr(), 1, 0, 512)
    assert not redo and np.array_equal(pk, oracle.demod(iq, rate))
    ctx.close()


@pytest.mark.gpu
def test_host_free_sharded_step_on_device(lib):
    """The host-free time-shard step (am_shard_scan_async -> device-side entry composition -> am_shard_resolve_async) on the
    device: one chunk = the whole stream (world 1: no collective), three steps, none falls back to the synchronous path, and
    the device-side composition over several chunks' tables (all on one GPU) gives the single-stream packets."""
    import torch
    from air_modes.sharded import ShardedReceiver
    rate, n = 64e6, 8_000_000
    iq, _ = synth.synth_capture(rate, n, 20000.0, 4242)
    want = oracle.demod(iq, rate)
    ctx = _capi.Context(rate, 7.0, True, lib=lib)
    rx = ShardedReceiver(ctx, 0, 1, n, device="cuda:0")
    rx.chunk.copy_(torch.from_numpy(iq.view(np.float32)))
    torch.cuda.synchronize()
    for _ in range(3):
        assert np.array_equal(rx.step(flush=True), want)       # (flush: a finite batch, the receiver starts over)
    assert rx.sync_steps == 0
    # ... and as a STREAM: the same capture in four steps of n / 4 samples -- the scan position, the undecided tail and the
    # sample count cross the steps (lib/preamble_impl.cc:213,237,244); one rank = the last rank = rank 0
    m4 = n // 4
    rx4 = ShardedReceiver(ctx, 0, 1, m4, device="cuda:0")
    parts = []
    for k in range(4):
        rx4.chunk.copy_(torch.from_numpy(iq[k * m4:(k + 1) * m4].view(np.float32)))
        parts.append(rx4.step(flush=(k == 3)))
    assert np.array_equal(np.concatenate(parts), want) and all(len(p_) > 0 for p_ in parts)
    assert rx4.sync_steps == 0
    ctx.close()
    # four chunks on one GPU, tables gathered by hand: what the all_gather delivers
    G, cap = 4, 512
    ctxs = [_capi.Context(rate, 7.0, True, lib=lib) for _ in range(G)]
    hl, hr = ctxs[0].shard_halo()
    dev = torch.device("cuda:0")
    W = 2 * (_capi.SHARD_MSG_HEADER + cap)                     # int64 words per message: header + cap entries of (pos, exit)
    msgs = torch.zeros(G * W, dtype=torch.int64, device=dev)
    bufs = []
    m = n // G
    got = None
    for rep in range(2):               # (the second pass launches for a capacity: no read-back at all)
        for c_ in ctxs:
            c_.reset()                 # (a new stream: the scan starts at sample 0 again, not where the last pass left off)
        for g in range(G):
            a, b = g * m, (g + 1) * m
            lo, hi = max(0, a - hl), min(n, b + hr)
            t = torch.from_numpy(iq[lo:hi].view(np.float32)).to(dev)
            bufs.append(t)
            ctxs[g].shard_scan_async(t.data_ptr(), a, b, n, msgs[g * W:].data_ptr(), cap)
        torch.cuda.synchronize()
        parts = []
        for g in range(G):
            pk, redo = ctxs[g].shard_resolve_async(msgs.data_ptr(), G, g, cap)
            assert not redo
            parts.append(pk)
        got = np.concatenate(parts)
        assert np.array_equal(got, want)
    for c in ctxs:
        c.close()


@pytest.mark.gpu
def test_stream_ordering_by_events_on_device(lib):
    """am_wait_for_stream / am_signal_stream (the sharded step's ordering against PyTorch's streams had no test on
    the device).  A side stream is kept busy for tens of milliseconds and only then writes the samples: the context must scan
    the samples, not the zeros that are there before; and a copy enqueued on the side stream behind am_signal_stream must see
    the exit table the context's stream writes, not the buffer's old content."""
    import torch
    rate, n = 20e6, 4_000_000
    iq, _ = synth.synth_capture(rate, n, 5000.0, 777)
    want = oracle.demod(iq, rate)
    assert len(want) > 50
    dev = torch.device("cuda:0")
    src = torch.from_numpy(iq.view(np.float32)).to(dev)
    buf = torch.zeros_like(src)
    side = torch.cuda.Stream(device=dev)
    ctx = _capi.Context(rate, 7.0, True, lib=lib)
    ctx.process_iq_device(src.data_ptr(), n, flush=True)         # (buffers allocated, capacity estimate in place)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        torch.cuda._sleep(60_000_000)                             # ~25 ms of spinning in front of the copy
        buf.copy_(src, non_blocking=True)
    ctx.wait_for_stream(side.cuda_stream)
    got = ctx.process_iq_device(buf.data_ptr(), n, flush=True)
    assert np.array_equal(got, want)
    # the other direction: the table of an enqueued scan, copied by the side stream
    cap = 512
    msg = torch.full((2 * (_capi.SHARD_MSG_HEADER + cap),), -1, dtype=torch.int64, device=dev)
    seen = torch.empty_like(msg)
    torch.cuda.synchronize()
    ctx.shard_scan_async(src.data_ptr(), 0, n, n, msg.data_ptr(), cap)
    ctx.signal_stream(side.cuda_stream)
    with torch.cuda.stream(side):
        seen.copy_(msg, non_blocking=True)
    side.synchronize()
    torch.cuda.synchronize()
    assert int(seen[0].item()) >= 0 and torch.equal(seen, msg)    # (the header's count was written before the copy ran)
    pk, redo = ctx.shard_resolve_async(msg.data_ptr(), 1, 0, cap)
    assert not redo and np.array_equal(pk, want)
    ctx.close()


@pytest.mark.gpu
def test_slicer_edge_vectors_on_device(lib):
    """VERDICT r4 missing #3 / next #6: >= 1e5 crafted + random 240-float bursts through am_slicer_work on the device against the
    oracle AND the reference's own slicer_impl::work (oracle/_ref travels as a prebuilt .so): DF16 long, DF18 / 19 / 24.. short
    (lib/slicer_impl.cc:140), chips exactly at lo / hi / lo * 0.5 (:74-98), 9 and 10 low-confidence bits on DF11 (:171), 24+ on long
    frames (:157), all-zero payloads (:162-166), +-inf / NaN chips."""
    n = 0
    for seed in (20252, 20253):
        n += pc.check_slicer_edge_vectors(lib, 60000, seed, with_ref=True)
    assert n > 20000


@pytest.mark.gpu
@pytest.mark.parametrize("rate,n,lam", [(2e6, 2000000, 1500.0), (20e6, 6000000, 6000.0), (64e6, 12800000, 12000.0)])
def test_framer_edge_formats_through_the_production_path_on_device(lib, rate, n, lam):
    """The same formats on the air, through the production extraction + slicing kernels (AM_F_KEEP_TAGS), stage by stage."""
    assert pc.check_framer_edge_formats(lib, rate, n, lam, 617, want_fe=3) > 20


@pytest.mark.gpu
@pytest.mark.parametrize("rate,n,lam,depth,contiguous", [(64e6, 9000000, 20000.0, 4, False), (64e6, 6000000, 2000.0, 3, True),
                                                         (20e6, 3000000, 6000.0, 2, False), (2e6, 600000, 3000.0, 3, True),
                                                         (5e6, 900000, 2500.0, 3, False)])
def test_one_stream_with_chunks_in_flight_on_device(hip_lib, rate, n, lam, depth, contiguous):
    """am_spipe on the device: consecutive chunks of ONE stream in flight on streams of their own, the scan position
    handed on through a device word behind an event; random chunk sizes; == the oracle over the whole stream, twice (a second stream
    through the same pipe)."""
    import torch
    rx = [(0, 1000, 0.25), (n // 2 + 12345, 2000, 0.5)]
    pk, _ = pc.check_stream_pipe(hip_lib, rate, n, lam, seed=int(rate / 1e6) + depth, depth=depth, contiguous=contiguous, rx_time=rx,
                                 device=torch.device("cuda", 0))
    assert len(pk) > 10


@pytest.mark.gpu
def test_stream_pipe_silent_chunks_and_redone_chunks_on_device(hip_lib, hip_knobs_lib, monkeypatch):
    """am_spipe on the device, the two paths a busy capture does not reach: chunks without a single candidate (the scan position is
    still handed on), and chunks whose scan outgrew the capacity it was launched for (flagged in the message header: the pipe drains
    the chunks behind, redoes the chunk on the synchronous path, submits the others again)."""
    import torch
    dev = torch.device("cuda", 0)
    rate = 64e6
    a, _ = synth.synth_capture(rate, 2_000_000, 20000.0, 81)
    b, _ = synth.synth_capture(rate, 2_000_000, 20000.0, 82)
    iq = np.concatenate([a, np.zeros(4_000_000, np.complex64), b])
    want = oracle.demod(iq, rate, 7.0, True)
    base = torch.from_numpy(np.ascontiguousarray(iq.view(np.float32))).to(dev)
    m = len(iq) // 10
    chunks = [(base.data_ptr() + 8 * k * m, m) for k in range(10)]
    pipe = _capi.StreamPipe(rate, 7.0, True, depth=3, lib=hip_lib, device=0)
    got = pipe.run(chunks)
    assert sum(1 for g in got if len(g) == 0) >= 3 and np.array_equal(np.concatenate(got), want) and pipe.redone() == 0
    pipe.close()
    # a quiet first half, dense traffic behind it, no slack in the capacity estimate: some chunk must be redone
    monkeypatch.setenv("AIRMODES_SPEC_FLOOR", "0")
    quiet, _ = synth.synth_capture(rate, 8_000_000, 300.0, 71)
    busy, _ = synth.synth_capture(rate, 8_000_000, 30000.0, 72)
    iq2 = np.concatenate([quiet, busy])
    want2 = oracle.demod(iq2, rate, 7.0, True)
    base2 = torch.from_numpy(np.ascontiguousarray(iq2.view(np.float32))).to(dev)
    m2 = len(iq2) // 8
    pipe2 = _capi.StreamPipe(rate, 7.0, True, depth=3, lib=hip_knobs_lib, device=0)
    got2 = np.concatenate(pipe2.run([(base2.data_ptr() + 8 * k * m2, m2) for k in range(8)]))
    assert np.array_equal(got2, want2)
    assert pipe2.redone() >= 1, "no chunk took the synchronous path: the test did not reach it"
    pipe2.close()


@pytest.mark.parametrize("rate,W,K,m,small_cap,lam", [(64e6, 3, 4, 2_000_000, 512, 12000.0), (20e6, 4, 3, 900_000, 512, 9000.0),
                                                      (64e6, 2, 4, 2_500_000, 1, 12000.0), (2e6, 3, 3, 200_000, 512, 3000.0)])
def test_steps_in_flight_over_w_ranks_on_device(hip_lib, oracle_mod, rate, W, K, m, small_cap, lam):
    """am_shard_resolve_submit / _collect with cur_in / carry_out over W ranks' device tables on the GPU (W ranks in one process, two
    contexts each): every rank composes the step's entry and its last exit by itself; one-entry messages flag nearly every step."""
    import synth
    iq, _ = synth.synth_capture(rate, W * K * m, lam, seed=int(rate / 1e5) + W)
    tags = [(0, 1000, 0.25), (W * m + 777, 2000, 0.5)]
    got, redone = pc.run_stream_shards_in_flight(hip_lib, rate, iq, W, K, small_cap=small_cap, rx_time=tags)
    want = oracle_mod.demod(iq, rate, rx_time=tags)
    assert len(want) > 50 and got.tobytes() == want.tobytes()
    assert (redone >= K - 1) if small_cap == 1 else (redone == 0)
