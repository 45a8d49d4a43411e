"""ctypes front door to the CPU oracle (oracle/liboracle.so) and, where it has been
built, to the reference's own C++ compiled by path (oracle/_ref/libairmodes_ref.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_PATH = os.path.join(HERE, "_ref", "libairmodes_ref.so")

PACKET_DTYPE = np.dtype([
    ("data", "u1", 14), ("nbytes", "u1"), ("df", "u1"), ("numlowconf", "u1"),
    ("reserved", "u1", 3), ("crc", "<u4"), ("ref", "<f4"), ("reserved2", "<u4"),
    ("sample", "<u8"), ("secs", "<u8"), ("frac", "<f8")])
TAG_DTYPE = np.dtype([("sample", "<u8"), ("secs", "<u8"), ("frac", "<f8"),
                      ("inavg", "<f4"), ("how_late", "<u4")])
assert PACKET_DTYPE.itemsize == 56 and TAG_DTYPE.itemsize == 32

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(LIB_PATH) or (
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "airmodes_oracle.c"))):
        subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])
    if os.path.isdir("/root/reference/lib") and (force or not os.path.exists(REF_PATH)):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.amo_threshold_lin.restype = C.c_float
        L.amo_threshold_lin.argtypes = [C.c_float]
        L.amo_mag2.argtypes = [_f32p, C.c_uint64, _f32p]
        L.amo_frontend.argtypes = [_f32p, C.c_uint64, C.c_int, C.c_int, _f32p, _f32p]
        L.amo_frontend_running.argtypes = [_f32p, C.c_uint64, C.c_int, C.c_int, C.c_uint32, _f32p, _f32p]
        L.amo_frontend_running2.argtypes = [_f32p, C.c_uint64, C.c_int, C.c_int, C.c_uint32, C.c_uint32, _f32p, _f32p]
        L.amo_preamble_scan.restype = C.c_uint64
        L.amo_preamble_scan.argtypes = [_f32p, _f32p, C.c_uint64, C.c_int, C.c_float, C.c_uint64,
                                        _f32p, C.c_void_p, C.c_uint64]
        L.amo_candidates_r.restype = C.c_uint64
        L.amo_candidates_r.argtypes = [_f32p, _f32p, C.c_uint64, C.c_uint64, C.c_float, C.c_uint64, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_uint64]
        L.amo_candidates.restype = C.c_uint64
        L.amo_candidates.argtypes = [_f32p, _f32p, C.c_uint64, C.c_int, C.c_float, C.c_uint64, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_uint64]
        L.amo_slice.argtypes = [_f32p, C.c_void_p, C.c_void_p]
        L.amo_crc24.restype = C.c_uint32
        L.amo_crc24.argtypes = [_u8p, C.c_int]
        L.amo_dcblock.argtypes = [_f32p, C.c_uint64, C.c_int, _f32p]
        L.amo_demod2.restype = C.c_uint64
        L.amo_demod2.argtypes = [_f32p, C.c_uint64, C.c_double, C.c_float, C.c_int, C.c_int, C.c_void_p,
                                 C.c_uint64, C.POINTER(C.c_uint64)]
        L.amo_demod.restype = C.c_uint64
        L.amo_demod.argtypes = [_f32p, C.c_uint64, C.c_double, C.c_float, C.c_int, C.c_void_p,
                                C.c_uint64, C.POINTER(C.c_uint64)]
        L.amo_format_message.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
        L.amo_restamp_packets.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64]
        L.amo_restamp_tags.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64]
        L.amo_restamp_packets.restype = None
        L.amo_restamp_tags.restype = None
        _lib = L
    return _lib


def have_ref():
    return os.path.exists(REF_PATH)


def ref():
    global _ref
    if _ref is None:
        R = C.CDLL(REF_PATH)
        R.ref_crc24.restype = C.c_uint32
        R.ref_crc24.argtypes = [_u8p, C.c_int]
        R.ref_preamble_slicer.argtypes = [
            _f32p, _f32p, C.c_uint64, C.c_float, C.c_float, C.c_uint64, _f32p,
            np.ctypeslib.ndpointer(np.uint64), np.ctypeslib.ndpointer(np.float64),
            np.ctypeslib.ndpointer(np.uint64), C.c_uint64, C.POINTER(C.c_uint64),
            C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        R.ref_preamble_slicer_tt.argtypes = R.ref_preamble_slicer.argtypes + [
            C.c_uint64, np.ctypeslib.ndpointer(np.uint64), np.ctypeslib.ndpointer(np.uint64),
            np.ctypeslib.ndpointer(np.float64)]
        R.ref_slicer.argtypes = [_f32p, C.c_uint64, np.ctypeslib.ndpointer(np.uint64), np.ctypeslib.ndpointer(np.float64),
                                 np.ctypeslib.ndpointer(np.uint8), C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64),
                                 C.POINTER(C.c_uint64)]
        _ref = R
    return _ref


def ref_slice_bursts(bursts, tags):
    """The reference's OWN slicer (lib/slicer_impl.cc:102-198, compiled by path into oracle/_ref) over caller-made bursts
    [n, 240] tagged with (secs, frac) from `tags`.  Returns (message texts of the accepted bursts in order, accepted[n] bool)."""
    b = np.ascontiguousarray(bursts, np.float32).reshape(-1)
    n = len(tags)
    assert b.size == n * 240
    secs = np.ascontiguousarray(tags["secs"], np.uint64)
    frac = np.ascontiguousarray(tags["frac"], np.float64)
    acc = np.zeros(n, np.uint8)
    mcap = n * 96 + 64
    msgs = C.create_string_buffer(mcap)
    mlen, nmsg = C.c_uint64(0), C.c_uint64(0)
    rc = ref().ref_slicer(b, n, secs, frac, acc, msgs, mcap, C.byref(mlen), C.byref(nmsg))
    if rc != 0:
        raise RuntimeError("ref_slicer rc=%d" % rc)
    text = msgs.raw[:mlen.value].decode().split("\n")[:-1]
    assert len(text) == nmsg.value == int(acc.sum())
    return text, acc.astype(bool)


def as_iq_f32(iq):
    """complex64 (n,) or float32 (2n,) -> contiguous interleaved float32 (2n,)."""
    a = np.ascontiguousarray(iq)
    if np.iscomplexobj(a):
        a = a.astype(np.complex64, copy=False).view(np.float32)
    return np.ascontiguousarray(a, dtype=np.float32).reshape(-1)


def threshold_lin(thr_db):
    return float(lib().amo_threshold_lin(thr_db))


def mag2(iq):
    f = as_iq_f32(iq)
    out = np.empty(f.size // 2, np.float32)
    lib().amo_mag2(f, out.size, out)
    return out


def frontend(iq, spc, use_pmf=True, running_chunk=None, running_first=0):
    """bb, avg in the canonical summation order; with running_chunk: GNU Radio's running sum instead, re-seeded
    every running_chunk outputs (the first time after running_first outputs) -- sensitivity study only."""
    f = as_iq_f32(iq)
    n = f.size // 2
    bb = np.empty(n, np.float32)
    avg = np.empty(n, np.float32)
    if running_chunk:
        rc = lib().amo_frontend_running2(f, n, spc, int(use_pmf), running_chunk, int(running_first), bb, avg)
    else:
        rc = lib().amo_frontend(f, n, spc, int(use_pmf), bb, avg)
    if rc != 0:
        raise RuntimeError("amo_frontend failed")
    return bb, avg


TIME_TAG_DTYPE = np.dtype([("offset", "<u8"), ("secs", "<u8"), ("frac", "<f8")])


def time_tags(rx_time):
    """[(offset, secs, frac), ...] -> amo_time_tag array sorted by offset (stable: a later entry with the
    same offset wins, as tstamp_tags.back() does in preamble_impl.cc:168-170)."""
    tt = np.zeros(len(rx_time), TIME_TAG_DTYPE)
    for i, (o, s_, f) in enumerate(rx_time):
        tt[i] = (o, s_, f)
    return tt[np.argsort(tt["offset"], kind="stable")]


def restamp(records, rate, rx_time):
    """Recompute secs/frac of packets (PACKET_DTYPE) or preamble tags (TAG_DTYPE) under rx_time tags."""
    if rx_time is None or len(records) == 0:
        return records
    tt = time_tags(rx_time)
    records = np.ascontiguousarray(records)
    fn = lib().amo_restamp_packets if records.dtype == PACKET_DTYPE else lib().amo_restamp_tags
    fn(records.ctypes.data, len(records), int(rate), tt.ctypes.data, len(tt))
    return records


def preamble_scan(bb, avg, spc, thr_db, rate, rx_time=None):
    n = bb.size
    cap = n // (240 * spc) + 2
    bursts = np.zeros((cap, 240), np.float32)
    tags = np.zeros(cap, TAG_DTYPE)
    hits = lib().amo_preamble_scan(np.ascontiguousarray(bb, np.float32), np.ascontiguousarray(avg, np.float32),
                                   n, spc, thr_db, int(rate), bursts.reshape(-1), tags.ctypes.data, cap)
    assert hits <= cap
    return bursts[:hits], restamp(tags[:hits], rate, rx_time)


def candidates(bb, avg, spc, thr_db, k_limit=None, rate=None):
    """Every first-stage candidate refined on its own: (pos, refined, valid, inavg) in item counts of the preamble
    block (stream index + history: 2*spc - 1 for whole samples per chip), positions below k_limit.  rate: the sample
    rate when it is not spc * 2 MHz (fractional samples per chip)."""
    rate_i = int(rate) if rate is not None else int(spc) * 2000000
    n = bb.size
    bb = np.ascontiguousarray(bb, np.float32)
    avg = np.ascontiguousarray(avg, np.float32)
    lim = (1 << 62) if k_limit is None else int(k_limit)
    cap = 1 << 16
    while True:
        pos = np.zeros(cap, np.uint64)
        ref_ = np.zeros(cap, np.uint64)
        val = np.zeros(cap, np.uint8)
        iav = np.zeros(cap, np.float32)
        m = lib().amo_candidates_r(bb, avg, n, rate_i, thr_db, lim, pos.ctypes.data, ref_.ctypes.data, val.ctypes.data,
                                   iav.ctypes.data, cap)
        if m <= cap:
            return pos[:m], ref_[:m], val[:m], iav[:m]
        cap = int(m)


def slice_bursts(bursts, tags):
    out = np.zeros(len(tags), PACKET_DTYPE)
    n = 0
    one = np.zeros(1, PACKET_DTYPE)
    for i in range(len(tags)):
        b = np.ascontiguousarray(bursts[i], np.float32)
        t = tags[i:i + 1].copy()
        if lib().amo_slice(b, t.ctypes.data, one.ctypes.data):
            out[n] = one[0]
            n += 1
    return out[:n]


def crc24(data):
    d = np.frombuffer(bytes(data), np.uint8).copy()
    return int(lib().amo_crc24(d, d.size))


def dcblock(iq, spc):
    """a2: the optional DC blocker in front of the path; returns complex64."""
    f = as_iq_f32(iq)
    out = np.empty_like(f)
    if lib().amo_dcblock(f, f.size // 2, spc, out) != 0:
        raise RuntimeError("amo_dcblock failed")
    return out.view(np.complex64)


def demod(iq, rate, thr_db=7.0, use_pmf=True, return_tags=False, use_dcblock=False, rx_time=None):
    f = as_iq_f32(iq)
    n = f.size // 2
    spc = max(int(rate / 2e6), 1)
    cap = n // (240 * spc) + 2
    out = np.zeros(cap, PACKET_DTYPE)
    ntags = C.c_uint64(0)
    npk = lib().amo_demod2(f, n, float(rate), thr_db, int(use_pmf), int(use_dcblock), out.ctypes.data, cap,
                           C.byref(ntags))
    assert npk <= cap
    pk = restamp(out[:npk], rate, rx_time)
    return (pk, int(ntags.value)) if return_tags else pk


def format_messages(packets, first=True):
    """Message texts as the reference's slicer would post them, in order."""
    buf = C.create_string_buffer(200)
    msgs = []
    for i in range(len(packets)):
        p = packets[i:i + 1].copy()
        w = lib().amo_format_message(p.ctypes.data, int(first and i == 0), buf, 200)
        assert w > 0
        msgs.append(buf.value.decode())
    return msgs


def ref_preamble_slicer(bb, avg, spc, thr_db, rate, rx_time=None):
    """The reference's OWN preamble_impl + slicer_impl (+modes_crc) on the two float
    streams; canonical end-of-stream rule applied here.  Returns (bursts, tags, msgs, keep).
    rx_time = [(offset, secs, frac), ...]: "rx_time" stream tags on the preamble block's input (the
    driver ends scheduler windows at the tags, see ref_driver.cc; needs silence in front of each tag)."""
    if rx_time is not None:
        # the hits do not depend on the time tags: item counts come from an untagged run
        b0, t0, _, keep0 = ref_preamble_slicer(bb, avg, spc, thr_db, rate)
    n = bb.size
    pad = 600 * (spc + 1)
    cap = (n + pad) // (240 * spc) + 4
    bursts = np.zeros((cap, 240), np.float32)
    secs = np.zeros(cap, np.uint64)
    frac = np.zeros(cap, np.float64)
    item = np.zeros(cap, np.uint64)
    ntags = C.c_uint64(0)
    mcap = cap * 96 + 64
    msgs = C.create_string_buffer(mcap)
    mlen = C.c_uint64(0)
    nmsg = C.c_uint64(0)
    args = [np.ascontiguousarray(bb, np.float32), np.ascontiguousarray(avg, np.float32),
            n, float(rate), thr_db, pad, bursts.reshape(-1), secs, frac, item, cap,
            C.byref(ntags), msgs, mcap, C.byref(mlen), C.byref(nmsg)]
    if rx_time is None:
        rc = ref().ref_preamble_slicer(*args)
    else:
        tt = time_tags(rx_time)
        rc = ref().ref_preamble_slicer_tt(*args, len(tt), np.ascontiguousarray(tt["offset"]),
                                          np.ascontiguousarray(tt["secs"]), np.ascontiguousarray(tt["frac"]))
    if rc != 0:
        raise RuntimeError("ref_preamble_slicer rc=%d" % rc)
    nt = int(ntags.value)
    assert nt <= cap
    tags = np.zeros(nt, TAG_DTYPE)
    tags["secs"] = secs[:nt]
    tags["frac"] = frac[:nt]
    r = int(rate)
    text = msgs.raw[:mlen.value].decode().split("\n")[:-1]
    assert len(text) == nmsg.value
    if rx_time is not None:
        assert nt == len(t0) and np.array_equal(bursts[:nt], b0)
        tags["sample"] = t0["sample"]
        return bursts[:nt], tags, text, keep0
    tags["sample"] = secs[:nt] * np.uint64(r) + np.rint(frac[:nt] * r).astype(np.uint64)
    # canonical end-of-stream rule: hits need 240 * d_samples_per_chip items of room -- in the
    # reference's own float arithmetic (lib/preamble_impl.cc:57-62,150,212)
    spcf = np.float32(np.float32(int(rate)) / np.float32(2000000))
    hist0 = int(np.float32(spcf * np.float32(2))) - 1
    K = n + hist0
    ninputs = K - K % spc - spc
    room = (ninputs - tags["sample"].astype(np.int64)).astype(np.float32)
    keep = ~(room < np.float32(np.float32(240) * spcf))
    return bursts[:nt], tags, text, keep
