"""Host-side checks that need no GPU: the C-ABI library loads and exports every symbol the
public header declares; the reference-shaped Python surface behaves."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import conftest


def header_symbols(name="airmodes_hip.h"):
    with open(os.path.join(conftest.ROOT, "include", name)) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(am_[a-z0-9_]+)\s*\(", text)))


# diagnostics and test hooks: exported, declared apart from the drop-in surface (include/airmodes_hip_debug.h)
DEBUG_SYMBOLS = ["am_fetch_candidates", "am_is_emulated", "am_last_frontend", "am_last_num_candidates", "am_last_timing",
                 "am_shard_get_exit", "am_shard_set_exit"]


def test_library_builds_loads_and_exports_all_symbols():
    subprocess.check_call(["make", "-s", "-C", os.path.join(conftest.ROOT, "gr-air-modes_amd", "csrc")])
    lib = ctypes.CDLL(conftest.HIP_LIB)
    main, debug = header_symbols(), header_symbols("airmodes_hip_debug.h")
    assert debug == DEBUG_SYMBOLS and not set(main) & set(debug)      # both lists, as the headers state them
    syms = sorted(main + debug)
    assert len(main) >= 20
    for s in syms:
        assert hasattr(lib, s), "libairmodes_hip.so does not export %s" % s
    lib.am_abi_version.restype = ctypes.c_uint32
    assert lib.am_abi_version() == 5
    assert lib.am_is_emulated() == 0
    # -fvisibility=hidden + AM_API: the shared object defines the header's entry points and NOTHING else (the reference
    # hides what is not AIR_MODES_API: CMakeLists.txt:65, include/gr_air_modes/api.h:27-31)
    out = subprocess.check_output(["nm", "-D", "--defined-only", conftest.HIP_LIB], text=True)
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.split() and line.split()[-2] in "TWVDBRi")
    exported = [e for e in exported if not e.startswith(("_init", "_fini", "__hip_", "_edata", "_end", "__bss_start"))]
    assert exported == syms, (sorted(set(exported) - set(syms)), sorted(set(syms) - set(exported)))
    # host-only helpers are callable without a GPU
    lib.am_crc24.restype = ctypes.c_uint32
    b = bytes.fromhex("8D4840D6202CC371C32CE0")
    assert lib.am_crc24(b, len(b)) == 0x576098


def test_no_gpu_means_loud_failure():
    """Without a HIP device the product path must raise, not fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from air_modes import _capi
    lib = _capi.Library(conftest.HIP_LIB)
    with pytest.raises(_capi.AirModesError):
        _capi.Context(4e6, 7.0, True, lib=lib)


def test_missing_library_is_an_error(tmp_path):
    from air_modes import _capi
    with pytest.raises(OSError):
        _capi.Library(str(tmp_path / "nope.so"))


def test_packet_layout_matches_header():
    from air_modes import _capi
    d = _capi.PACKET_DTYPE
    assert d.itemsize == 56
    assert [d.fields[k][1] for k in ("data", "nbytes", "df", "numlowconf", "crc", "ref", "sample", "secs", "frac")] == \
        [0, 14, 15, 16, 20, 24, 32, 40, 48]
    assert _capi.TAG_DTYPE.itemsize == 32 and _capi.EXIT_DTYPE.itemsize == 16


def test_msg_queue_semantics():
    import air_modes
    q = air_modes.msg_queue()
    assert q.empty_p() and q.count() == 0 and q.delete_head_nowait() is None
    q.handle(air_modes.message.make_from_string("a b c"))
    q.insert_tail(air_modes.message.make_from_string("d"))
    assert q.count() == 2 and not q.empty_p()
    assert q.delete_head().to_string() == "a b c"
    assert q.delete_head_nowait().to_string() == "d"
    q.insert_tail(air_modes.message("x"))
    q.flush()
    assert q.empty_p()


def test_rx_path_surface_via_emulation(emu_lib):
    """rx_path(rate, threshold, queue, use_pmf, use_dcblock) + setters, messages on the queue
    in the reference's text format incl. the first-message precision quirk."""
    import air_modes
    import oracle
    import synth
    q = air_modes.msg_queue()
    rx = air_modes.rx_path(4e6, 7.0, q, use_pmf=True, use_dcblock=False, lib=emu_lib)
    assert rx.get_threshold() == 7.0 and rx.get_pmf(None) is True
    rx.set_pmf(False)                        # no-op, like the reference
    assert rx.get_pmf(None) is True
    iq, _ = synth.synth_capture(4e6, 150000, 2500.0, seed=12)
    rx.work(iq[:40000])
    rx.work(iq[40000:], flush=True)
    got = []
    while not q.empty_p():
        got.append(q.delete_head().to_string())
    want = oracle.format_messages(oracle.demod(iq, 4e6))
    assert got == want and len(got) > 5
    first, second = got[0].split(), got[1].split()
    assert len(first) == 5 and len(first[0]) in (14, 28) and len(first[1]) == 6
    assert len(first[2].replace(".", "").replace("-", "").lstrip("0")) <= 6      # %.6g
    assert rx.packets == len(got) and rx.samples == len(iq)
    # block-level objects
    pre = air_modes.preamble(4e6, 7.0, lib=emu_lib)
    assert pre.get_rate() == 4e6 and pre.get_threshold() == 7.0
    bb, avg = oracle.frontend(iq, 2, True)
    bursts, tags = pre.work(bb, avg)
    q2 = air_modes.msg_queue()
    sl = air_modes.slicer(q2, lib=emu_lib)
    pk = sl.work(bursts, tags)
    assert q2.count() == len(pk) == len(got)


def test_rx_time_through_the_host_mirror(emu_lib):
    """rx_path.work(..., rx_time=[(offset, secs, frac)]) and preamble.work(..., rx_time=...): what a live
    source's "rx_time" stream tags do to the message timestamps (lib/preamble_impl.cc:100-137,165-170)."""
    import air_modes
    import oracle
    import synth
    rate = 4e6
    iq, _ = synth.synth_capture(rate, 160000, 2500.0, seed=14)
    rx_tags = [(0, 1700000000, 0.5), (90000, 1700000123, 0.25)]
    q = air_modes.msg_queue()
    rx = air_modes.rx_path(rate, 7.0, q, use_pmf=True, lib=emu_lib)
    rx.work(iq[:50000], rx_time=rx_tags[:1])
    rx.work(iq[50000:120000], rx_time=rx_tags[1:])
    rx.work(iq[120000:], flush=True)
    got = []
    while not q.empty_p():
        got.append(q.delete_head().to_string())
    want = oracle.format_messages(oracle.demod(iq, rate, rx_time=rx_tags))
    assert got == want and len(got) > 5
    assert {m.split()[3] for m in got} >= {"1700000000", "1700000123"}
    # a second stream through the same object starts without tags again (flush ended the first one)
    rx.work(iq, flush=True)
    again = []
    while not q.empty_p():
        again.append(q.delete_head().to_string())
    plain = oracle.format_messages(oracle.demod(iq, rate), first=False)
    assert again == plain
    # block level
    pre = air_modes.preamble(rate, 7.0, lib=emu_lib)
    bb, avg = oracle.frontend(iq, 2, True)
    _, tags = pre.work(bb, avg, rx_time=rx_tags)
    _, otags = oracle.preamble_scan(bb, avg, 2, 7.0, rate, rx_time=rx_tags)
    assert np.array_equal(tags, otags)
    _, tags0 = pre.work(bb, avg)
    assert np.array_equal(tags0, oracle.preamble_scan(bb, avg, 2, 7.0, rate)[1])


def test_timing_accessors(emu_lib):
    """am_last_timing with either pointer NULL (bench.py reads the dominant-kernel time only: that never
    waits for the call's trailing event)."""
    import synth
    from air_modes import _capi
    ctx = _capi.Context(4e6, 7.0, True, lib=emu_lib)
    iq, _ = synth.synth_capture(4e6, 60000, 2500.0, seed=15)
    ctx.process_iq(iq, flush=True)
    assert ctx.last_dom_ms() >= 0.0
    total, dom = ctx.last_timing()
    assert total >= 0.0 and dom >= 0.0
    ctx.close()


def test_format_messages_batch_equals_one_by_one():
    """am_format_messages: a batch per call (lib/slicer_impl.cc:186-194: the precision quirk belongs to the stream's first message)."""
    from air_modes import _capi
    lib = _capi.Library(conftest.HIP_LIB)
    rng = np.random.default_rng(5)
    pk = np.zeros(300, _capi.PACKET_DTYPE)
    pk["data"] = rng.integers(0, 256, (300, 14), dtype=np.uint8)
    pk["nbytes"] = np.where(rng.random(300) < 0.5, 7, 14)
    pk["crc"] = rng.integers(0, 1 << 24, 300)
    pk["ref"] = rng.random(300).astype(np.float32) * np.float32(3.0)
    pk["secs"] = rng.integers(0, 1 << 40, 300)
    pk["frac"] = rng.random(300)
    for first in (True, False):
        one = [lib.format_message(pk[i], first and i == 0) for i in range(len(pk))]
        assert lib.format_messages(pk, first) == one
    assert lib.format_messages(pk[:0], True) == []
    # capacity: nothing beyond cap is written, the bytes needed come back
    buf = ctypes.create_string_buffer(64)
    offs = np.zeros(4, np.uint64)
    need = ctypes.c_uint64(0)
    rc = lib.L.am_format_messages(pk.ctypes.data, 3, 1, ctypes.addressof(buf), 40, offs.ctypes.data, ctypes.byref(need))
    assert rc == -5 and need.value == sum(len(t) + 1 for t in lib.format_messages(pk[:3], True)) and buf.raw[40:] == b"\0" * 24


def test_format_messages_longest_texts_fit_after_one_retry():
    """a text can take up to ~91 bytes (two %.10g numbers of 16 characters, an 11-digit count of seconds); the
    facade's first buffer is sized for typical texts and the library's `need` pays for the rest."""
    from air_modes import _capi
    lib = _capi.Library(conftest.HIP_LIB)
    p = np.zeros(40, _capi.PACKET_DTYPE)
    p["data"][:] = 0xAB
    p["nbytes"] = 14
    p["crc"] = 0xABCDEF
    p["ref"] = np.float32(-1.23e-05)
    p["secs"] = 12345678901
    p["frac"] = 9.5e-05
    texts = lib.format_messages(p, first=False)
    assert len(texts) == 40 and all(t == texts[0] for t in texts)
    assert texts[0] == lib.format_message(p[0], False)
    assert (len(texts[0]) + 1) * 40 > 72 * 40 + 32            # (the first buffer was too small: the retry with `need` ran)
