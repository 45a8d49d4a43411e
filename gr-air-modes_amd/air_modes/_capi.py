"""ctypes binding of include/airmodes_hip.h (libairmodes_hip.so).

This is the thin host-language shim north_star asks for: Python calls the HIP path through
the C ABI and nothing else.  There is NO CPU fallback here: if the library is missing or no
HIP device is usable, loading / context creation raises.
"""
import collections
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(os.path.dirname(_HERE), "csrc", "libairmodes_hip.so")

AM_F_DEVICE_IN = 0x1
AM_F_FLUSH = 0x2
AM_F_ZERO_GAPS = 0x20
AM_F_DEVICE_OUT = 0x4
AM_F_KEEP_TAGS = 0x8
AM_F_MORE = 0x10
ABI_VERSION = 5
SHARD_MSG_HEADER = 2          # header entries of a device-side exit-table message (am_shard_scan_async)

AM_OK, AM_EINVAL, AM_ENODEV, AM_ENOMEM, AM_EHIP, AM_ECAPACITY, AM_ENOTSUP = 0, -1, -2, -3, -4, -5, -6

PACKET_DTYPE = np.dtype([
    ("data", "u1", 14), ("nbytes", "u1"), ("df", "u1"), ("numlowconf", "u1"),
    ("reserved", "u1", 3), ("crc", "<u4"), ("ref", "<f4"), ("reserved2", "<u4"),
    ("sample", "<u8"), ("secs", "<u8"), ("frac", "<f8")])
TAG_DTYPE = np.dtype([("sample", "<u8"), ("secs", "<u8"), ("frac", "<f8"),
                      ("inavg", "<f4"), ("how_late", "<u4")])
EXIT_DTYPE = np.dtype([("pos", "<u8"), ("exit", "<u8")])
CAND_DTYPE = EXIT_DTYPE   # (name kept for older imports)
assert PACKET_DTYPE.itemsize == 56 and TAG_DTYPE.itemsize == 32 and CAND_DTYPE.itemsize == 16


class AirModesError(RuntimeError):
    def __init__(self, code, text):
        RuntimeError.__init__(self, "airmodes_hip error %d: %s" % (code, text))
        self.code = code


def _share_hip_runtime_with_torch():
    """One process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own
    libamdhip64.so (same soname as /opt/rocm's); if libairmodes_hip.so were loaded first it
    would bind to the system copy, a later `import torch` would bring up a second runtime and
    one of the two would see no device.  So when torch is installed but not yet imported, its
    runtime library is loaded first (without importing torch) and ours binds to it by soname."""
    import sys
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


class Library(object):
    """One loaded libairmodes_hip.so with typed entry points."""

    def __init__(self, path=None):
        path = path or os.environ.get("AIRMODES_HIP_LIB") or DEFAULT_LIB
        if not os.path.exists(path):
            raise OSError("libairmodes_hip.so not found at %s -- build it with "
                          "`python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % path)
        self.path = path
        _share_hip_runtime_with_torch()
        L = C.CDLL(path)
        vp, u64, u32, f32, f64, ci = C.c_void_p, C.c_uint64, C.c_uint32, C.c_float, C.c_double, C.c_int
        pu64 = C.POINTER(C.c_uint64)
        L.am_abi_version.restype = u32
        L.am_create.restype = vp
        L.am_create.argtypes = [ci, f64, f32, ci, ci, C.POINTER(ci)]
        L.am_destroy.argtypes = [vp]
        L.am_set_rate.argtypes = [vp, f64]
        L.am_set_threshold.argtypes = [vp, f32]
        L.am_set_rx_time.argtypes = [vp, u64, u64, f64]
        L.am_get_rate.restype = f64
        L.am_get_rate.argtypes = [vp]
        L.am_get_threshold.restype = f32
        L.am_get_threshold.argtypes = [vp]
        L.am_get_pmf.argtypes = [vp]
        L.am_reset.argtypes = [vp]
        L.am_process_iq.argtypes = [vp, vp, u64, u32, vp, u64, pu64]
        L.am_fetch_packets.argtypes = [vp, vp, u64, pu64]
        L.am_last_num_tags.restype = u64
        L.am_last_num_tags.argtypes = [vp]
        L.am_frontend_work.argtypes = [vp, vp, u64, u32, vp, vp]
        L.am_preamble_work.argtypes = [vp, vp, vp, u64, u32, vp, vp, u64, pu64]
        L.am_multi_layout.argtypes = [vp, u32, vp, vp, pu64]
        L.am_process_multi.argtypes = [vp, vp, u32, vp, u32, vp, u64, vp, pu64]
        L.am_submit_multi.argtypes = [vp, vp, u32, vp, u32]
        L.am_multi_counts.argtypes = [vp, vp, u32]
        L.am_preamble_stream.argtypes = [vp, vp, vp, u64, u32, vp, vp, u64, pu64]
        L.am_slicer_work.argtypes = [vp, vp, vp, u64, u32, vp, u64, pu64]
        L.am_crc24.restype = u32
        L.am_crc24.argtypes = [vp, ci]
        L.am_format_message.argtypes = [vp, ci, C.c_char_p, C.c_size_t]
        L.am_format_messages.argtypes = [vp, u64, ci, vp, C.c_size_t, vp, pu64]
        L.am_is_emulated.restype = ci
        L.am_shard_halo.argtypes = [vp, pu64, pu64]
        L.am_shard_scan.argtypes = [vp, vp, u64, u64, u64, u32, vp, u64, pu64]
        L.am_shard_entry.argtypes = [vp, vp, vp, u32, vp]
        L.am_shard_resolve.argtypes = [vp, u64, vp, u64, pu64]
        L.am_shard_entry2.argtypes = [vp, vp, u32, u64, vp, vp]
        L.am_shard_get_exit.argtypes = [vp, pu64]
        L.am_shard_set_exit.argtypes = [vp, u64]
        L.am_shard_keep_tail.argtypes = [vp, vp, vp, u64]
        L.am_stream_copy.argtypes = [vp, vp, vp, u64]
        L.am_last_error.restype = C.c_char_p
        L.am_last_error.argtypes = [vp]
        L.am_last_timing.argtypes = [vp, C.POINTER(f32), C.POINTER(f32)]
        L.am_last_num_candidates.restype = C.c_longlong
        L.am_last_num_candidates.argtypes = [vp]
        L.am_set_stream.argtypes = [vp, vp]
        L.am_wait_for_stream.argtypes = [vp, vp]
        L.am_signal_stream.argtypes = [vp, vp]
        L.am_shard_scan_async.argtypes = [vp, vp, u64, u64, u64, u32, vp, u64]
        L.am_shard_resolve_async.argtypes = [vp, vp, u32, u32, u64, vp, u64, pu64, C.POINTER(ci)]
        L.am_submit_iq.argtypes = [vp, vp, C.c_uint64, C.c_uint32]
        L.am_collect.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
        L.am_pipe_create.restype = vp
        L.am_pipe_create.argtypes = [C.c_int, C.c_double, C.c_float, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.am_pipe_destroy.argtypes = [vp]
        L.am_pipe_submit.argtypes = [vp, vp, C.c_uint64, C.c_uint32]
        L.am_pipe_collect.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
        L.am_pipe_submit_multi.argtypes = [vp, vp, u32, vp, u32]
        L.am_pipe_multi_counts.argtypes = [vp, vp, u32]
        L.am_pipe_in_flight.argtypes = [vp]
        L.am_pipe_depth.argtypes = [vp]
        L.am_pipe_last_error.restype = C.c_char_p
        L.am_pipe_last_error.argtypes = [vp]
        L.am_pipe_last_kernel_ms.restype = C.c_float
        L.am_pipe_last_kernel_ms.argtypes = [vp]
        L.am_shard_resolve_submit.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint64, vp, vp]
        L.am_shard_resolve_collect.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        L.am_spipe_create.restype = vp
        L.am_spipe_create.argtypes = [C.c_int, C.c_double, C.c_float, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.am_spipe_destroy.argtypes = [vp]
        L.am_spipe_submit.argtypes = [vp, vp, C.c_uint64, C.c_uint32]
        L.am_spipe_collect.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
        L.am_spipe_in_flight.argtypes = [vp]
        L.am_spipe_depth.argtypes = [vp]
        L.am_spipe_front.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.am_spipe_set_rx_time.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_double]
        L.am_spipe_redone.restype = C.c_uint64
        L.am_spipe_redone.argtypes = [vp]
        L.am_spipe_last_error.restype = C.c_char_p
        L.am_spipe_last_error.argtypes = [vp]
        L.am_spipe_last_kernel_ms.restype = C.c_float
        L.am_spipe_last_kernel_ms.argtypes = [vp]
        L.am_fetch_tags.argtypes = [vp, vp, vp, u64, pu64]
        L.am_fetch_candidates.argtypes = [vp, vp, vp, vp, vp, u64, pu64]
        L.am_last_frontend.restype = C.c_int
        L.am_last_frontend.argtypes = [vp]
        self.L = L
        if L.am_abi_version() != ABI_VERSION:
            raise OSError("ABI version mismatch in %s" % path)
        self.emulated = bool(L.am_is_emulated())      # the test-only CPU build of the same sources (tests/emu)

    # host-side helpers that need no context
    def crc24(self, data):
        d = np.frombuffer(bytes(data), np.uint8).copy()
        return int(self.L.am_crc24(d.ctypes.data, d.size))

    def format_message(self, packet, first):
        p = np.asarray(packet, PACKET_DTYPE).reshape(1).copy()
        buf = C.create_string_buffer(200)
        w = self.L.am_format_message(p.ctypes.data, int(bool(first)), buf, 200)
        if w < 0:
            raise AirModesError(w, "am_format_message")
        return buf.value.decode()


    def format_messages(self, packets, first):
        """The texts of a batch of accepted packets in ONE call (lib/slicer_impl.cc:186-194); `first`: the precision quirk
        of the stream's very first message applies to packets[0]."""
        p = np.ascontiguousarray(np.asarray(packets, PACKET_DTYPE).reshape(-1))
        n = int(p.size)
        if n == 0:
            return []
        cap = 72 * n + 32                                  # (typical texts take 55-65 bytes; the longest possible one 96)
        offs = np.zeros(n + 1, np.uint64)
        need = C.c_uint64(0)
        for _ in range(2):
            buf = C.create_string_buffer(cap)
            rc = self.L.am_format_messages(p.ctypes.data, n, int(bool(first)), C.addressof(buf), cap, offs.ctypes.data,
                                           C.byref(need))
            if rc != AM_ECAPACITY:
                break
            cap = int(need.value)                          # the library says what it takes: once more with exactly that
        if rc < 0:
            raise AirModesError(rc, "am_format_messages")
        raw = buf.raw[:int(offs[n])]
        return [t.decode() for t in raw.split(b"\0")[:n]]


_default = None


def default_library():
    global _default
    if _default is None:
        _default = Library()
    return _default


def _iq_f32(iq):
    a = np.ascontiguousarray(iq)
    if np.iscomplexobj(a):
        a = a.astype(np.complex64, copy=False).view(np.float32)
    return np.ascontiguousarray(a, dtype=np.float32).reshape(-1)


class Context(object):
    """am_ctx wrapper: one receive path (one stream) on one GPU."""

    def __init__(self, rate, threshold_db=7.0, use_pmf=True, use_dcblock=False, device=-1, lib=None):
        self.lib = lib or default_library()
        err = C.c_int(0)
        self._h = self.lib.L.am_create(int(device), float(rate), float(threshold_db), int(bool(use_pmf)),
                                       int(bool(use_dcblock)), C.byref(err))
        if not self._h:
            raise AirModesError(err.value, self.lib.L.am_last_error(None).decode())

    def close(self):
        if getattr(self, "_h", None):
            self.lib.L.am_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != AM_OK:
            raise AirModesError(rc, self.lib.L.am_last_error(self._h).decode())

    # setters / getters mirroring gr::air_modes::preamble and rx_path
    def set_rate(self, rate):
        self._chk(self.lib.L.am_set_rate(self._h, float(rate)))

    def set_threshold(self, thr_db):
        self._chk(self.lib.L.am_set_threshold(self._h, float(thr_db)))

    def set_rx_time(self, offset, secs, frac):
        """The "rx_time" stream tag of a live source (lib/preamble_impl.cc:165-170): from item `offset`
        on, packets are stamped (secs, frac) + (item - offset) / rate."""
        self._chk(self.lib.L.am_set_rx_time(self._h, int(offset), int(secs), float(frac)))

    def get_rate(self):
        return float(self.lib.L.am_get_rate(self._h))

    def get_threshold(self):
        return float(self.lib.L.am_get_threshold(self._h))

    def get_pmf(self):
        return bool(self.lib.L.am_get_pmf(self._h))

    def reset(self):
        self._chk(self.lib.L.am_reset(self._h))

    def last_num_tags(self):
        return int(self.lib.L.am_last_num_tags(self._h))

    def last_num_candidates(self):
        return int(self.lib.L.am_last_num_candidates(self._h))

    def set_stream(self, hip_stream):
        """Device work of this context goes to the caller's HIP stream (raw handle, e.g.
        torch.cuda.current_stream().cuda_stream); None / 0: the context's own stream again."""
        self._chk(self.lib.L.am_set_stream(self._h, C.c_void_p(int(hip_stream or 0))))

    def wait_for_stream(self, hip_stream):
        """The context's stream waits (on the device) for what is enqueued on hip_stream so far (0 / None: the default stream)."""
        self._chk(self.lib.L.am_wait_for_stream(self._h, C.c_void_p(int(hip_stream) if hip_stream else None)))

    def signal_stream(self, hip_stream):
        """hip_stream (0 / None: the default stream) waits, on the device, for what the context has enqueued so far."""
        self._chk(self.lib.L.am_signal_stream(self._h, C.c_void_p(int(hip_stream) if hip_stream else None)))

    def shard_scan_async(self, dev_ptr, abs_start, abs_end, total_n, msg_ptr, msg_cap, device_in=True, more=False):
        """msg_ptr: SHARD_MSG_HEADER + msg_cap entries of 16 bytes.  more: the stream goes on beyond total_n (AM_F_MORE)."""
        self._chk(self.lib.L.am_shard_scan_async(self._h, C.c_void_p(int(dev_ptr)), int(abs_start), int(abs_end), int(total_n),
                                                 (AM_F_DEVICE_IN if device_in else 0) | (AM_F_MORE if more else 0),
                                                 C.c_void_p(int(msg_ptr)), int(msg_cap)))

    def shard_get_exit(self):
        """Where the scan left this context's chunk in its last resolved time-shard step (0: none yet)."""
        v = C.c_uint64(0)
        self._chk(self.lib.L.am_shard_get_exit(self._h, C.byref(v)))
        return int(v.value)

    def shard_set_exit(self, pos):
        self._chk(self.lib.L.am_shard_set_exit(self._h, int(pos)))

    def shard_keep_tail(self, dst_ptr, src_ptr, nbytes):
        """Every following resolve step copies nbytes (device to device) from src to dst behind its slicing, before it completes."""
        self._chk(self.lib.L.am_shard_keep_tail(self._h, C.c_void_p(int(dst_ptr) or None), C.c_void_p(int(src_ptr) or None), int(nbytes)))

    def stream_copy(self, dst_ptr, src_ptr, nbytes):
        """A device-to-device copy on the context's stream (ordered with its scans)."""
        self._chk(self.lib.L.am_stream_copy(self._h, C.c_void_p(int(dst_ptr)), C.c_void_p(int(src_ptr)), int(nbytes)))

    def shard_resolve_async(self, msgs_ptr, world, rank, msg_cap, capacity=4096):
        """-> (packets, redo).  redo: nothing was delivered, repeat the step on the synchronous path."""
        out = self._receive_buffer(int(capacity))
        got, redo = C.c_uint64(0), C.c_int(0)
        rc = self.lib.L.am_shard_resolve_async(self._h, C.c_void_p(int(msgs_ptr)), int(world), int(rank), int(msg_cap),
                                               out.ctypes.data, len(out), C.byref(got), C.byref(redo))
        if rc == AM_ECAPACITY:
            return self._fetch(int(got.value)), False
        self._chk(rc)
        return self._received(out, got.value), bool(redo.value)


    def shard_resolve_submit(self, msgs_ptr, world, rank, msg_cap, cur_in_ptr=0, carry_out_ptr=0):
        """The resolve step, enqueued only (steps of the time-sharded receiver in flight); shard_resolve_collect completes it."""
        self._chk(self.lib.L.am_shard_resolve_submit(self._h, C.c_void_p(int(msgs_ptr)), int(world), int(rank), int(msg_cap),
                                                     C.c_void_p(int(cur_in_ptr) or None), C.c_void_p(int(carry_out_ptr) or None)))

    def shard_resolve_collect(self, capacity=4096):
        """-> (packets, redo) of the submitted resolve step."""
        out = self._receive_buffer(int(capacity))
        got, redo = C.c_uint64(0), C.c_int(0)
        rc = self.lib.L.am_shard_resolve_collect(self._h, out.ctypes.data, len(out), C.byref(got), C.byref(redo))
        if rc == AM_ECAPACITY:
            return self._fetch(int(got.value)), False
        self._chk(rc)
        return self._received(out, got.value), bool(redo.value)
    def last_frontend(self):
        """3 = streaming kernel, 2 = tile kernel, 1 = rate-generic kernels, 0 = no scan yet (diagnostic)."""
        return int(self.lib.L.am_last_frontend(self._h))

    def last_timing(self):
        """(whole call, dominant kernel) device milliseconds of the last call."""
        a, b = C.c_float(0), C.c_float(0)
        self._chk(self.lib.L.am_last_timing(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def last_dom_ms(self):
        """Dominant-kernel milliseconds only (never waits for the call's trailing event)."""
        b = C.c_float(0)
        self._chk(self.lib.L.am_last_timing(self._h, None, C.byref(b)))
        return b.value

    def _fetch(self, need):
        out = np.zeros(need, PACKET_DTYPE)
        n = C.c_uint64(0)
        self._chk(self.lib.L.am_fetch_packets(self._h, out.ctypes.data, need, C.byref(n)))
        return out[:n.value]

    def process_iq(self, iq, flush=False, capacity=None, keep_tags=False):
        """Host IQ (complex64 or interleaved float32) -> accepted packets of this chunk.  keep_tags: the bursts and
        tags of the chunk's preamble hits stay available for fetch_tags()."""
        f = _iq_f32(iq)
        n = f.size // 2
        return self._process(f.ctypes.data if n else None, n,
                             (AM_F_FLUSH if flush else 0) | (AM_F_KEEP_TAGS if keep_tags else 0), capacity)

    def process_iq_device(self, dev_ptr, n_complex, flush=False, capacity=None, keep_tags=False):
        """Device-resident interleaved float32 IQ (e.g. torch tensor .data_ptr())."""
        return self._process(int(dev_ptr), int(n_complex), AM_F_DEVICE_IN | (AM_F_FLUSH if flush else 0) |
                             (AM_F_KEEP_TAGS if keep_tags else 0), capacity)

    # K independent streams in one scan
    def multi_layout(self, lengths):
        """(offsets, total) in complex samples: where K whole streams of the given lengths lie in the ONE buffer
        process_multi scans (zeros between them)."""
        n = np.ascontiguousarray(lengths, np.uint64)
        off = np.zeros(n.size, np.uint64)
        total = C.c_uint64(0)
        self._chk(self.lib.L.am_multi_layout(self._h, n.size, n.ctypes.data, off.ctypes.data, C.byref(total)))
        return off, int(total.value)

    def multi_pack(self, streams):
        """Host helper: K streams (complex64 / interleaved float32) -> (buffer in the layout, lengths)."""
        fs = [_iq_f32(x) for x in streams]
        n = np.array([f.size // 2 for f in fs], np.uint64)
        off, total = self.multi_layout(n)
        buf = np.zeros(2 * total, np.float32)
        for f, o in zip(fs, off):
            buf[2 * int(o): 2 * int(o) + f.size] = f
        return buf, n

    def process_multi(self, buf, lengths, device_ptr=None, zero_gaps=False, capacity=None):
        """One scan over K whole streams (am_process_multi): list of K packet arrays, each what
        process_iq(stream, flush=True) gives.  buf: the packed host buffer, or device_ptr = the same on the device."""
        n = np.ascontiguousarray(lengths, np.uint64)
        _, total = self.multi_layout(n)
        flags = AM_F_ZERO_GAPS if zero_gaps else 0
        if device_ptr is not None:
            ptr, flags = int(device_ptr), flags | AM_F_DEVICE_IN
        else:
            f = _iq_f32(buf)
            assert f.size >= 2 * total
            ptr = f.ctypes.data if total else None
        cap = int(capacity) if capacity is not None else max(64, total // 2000 + 64)
        out = self._receive_buffer(cap)
        got = C.c_uint64(0)
        cnt = np.zeros(n.size, np.uint64)
        rc = self.lib.L.am_process_multi(self._h, ptr, n.size, n.ctypes.data, flags, out.ctypes.data, cap, cnt.ctypes.data,
                                         C.byref(got))
        if rc == AM_ECAPACITY:
            pk = self._fetch(int(got.value))
        else:
            self._chk(rc)
            pk = self._received(out, got.value)
        assert int(cnt.sum()) == len(pk)
        edges = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        return [pk[edges[j]:edges[j + 1]] for j in range(n.size)]

    def submit_multi(self, buf, lengths, device_ptr=None, zero_gaps=False):
        """First half of process_multi (am_submit_multi): the scan is enqueued, nothing is waited for."""
        n = np.ascontiguousarray(lengths, np.uint64)
        flags = AM_F_ZERO_GAPS if zero_gaps else 0
        if device_ptr is not None:
            ptr, flags = int(device_ptr), flags | AM_F_DEVICE_IN
            self._held_multi = None
        else:
            f = _iq_f32(buf)
            ptr = f.ctypes.data if f.size else None
            self._held_multi = f                      # the samples must stay valid until collect_multi
        self._chk(self.lib.L.am_submit_multi(self._h, ptr, n.size, n.ctypes.data, flags))
        self._multi_k = n.size

    def collect_multi(self, capacity=None):
        """Second half (am_collect + am_multi_counts): the K packet arrays of the scan submitted last."""
        cap = int(capacity) if capacity is not None else max(4096, len(getattr(self, "_rxbuf", ())))
        out = self._receive_buffer(cap)
        got = C.c_uint64(0)
        rc = self.lib.L.am_collect(self._h, out.ctypes.data, cap, C.byref(got))
        if rc == AM_ECAPACITY:
            pk = self._fetch(int(got.value))
            if capacity is None:
                self._rxbuf = np.zeros(int(got.value) + 1024, PACKET_DTYPE)
        else:
            self._chk(rc)
            pk = self._received(out, got.value)
        self._held_multi = None
        cnt = np.zeros(self._multi_k, np.uint64)
        self._chk(self.lib.L.am_multi_counts(self._h, cnt.ctypes.data, cnt.size))
        edges = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        assert int(edges[-1]) == len(pk)
        return [pk[edges[j]:edges[j + 1]] for j in range(cnt.size)]

    def fetch_tags(self):
        """(bursts [n, 240] float32, tags) of the last process_iq(..., keep_tags=True) call: what the preamble block
        hands the slicer (lib/preamble_impl.cc:219-232), from the kernels that produced the call's packets."""
        n = C.c_uint64(0)
        rc = self.lib.L.am_fetch_tags(self._h, None, None, 0, C.byref(n))
        if rc != AM_ECAPACITY:
            self._chk(rc)
        m = int(n.value)
        bursts = np.zeros((m, 240), np.float32)
        tags = np.zeros(m, TAG_DTYPE)
        if m:
            self._chk(self.lib.L.am_fetch_tags(self._h, bursts.ctypes.data, tags.ctypes.data, m, C.byref(n)))
        return bursts, tags

    def fetch_candidates(self):
        """(pos, refined, valid, inavg) of EVERY first-stage candidate of the last scan, absolute stream indices
        (diagnostic, stage-level parity tests)."""
        n = C.c_uint64(0)
        rc = self.lib.L.am_fetch_candidates(self._h, None, None, None, None, 0, C.byref(n))
        if rc != AM_ECAPACITY:
            self._chk(rc)
        m = int(n.value)
        pos = np.zeros(m, np.uint64)
        ref = np.zeros(m, np.uint64)
        val = np.zeros(m, np.uint8)
        iav = np.zeros(m, np.float32)
        if m:
            self._chk(self.lib.L.am_fetch_candidates(self._h, pos.ctypes.data, ref.ctypes.data, val.ctypes.data,
                                                     iav.ctypes.data, m, C.byref(n)))
        return pos, ref, val, iav

    def _process(self, ptr, n, flags, capacity):
        cap = int(capacity) if capacity is not None else max(64, n // 2000 + 64)
        out = self._receive_buffer(cap)
        got = C.c_uint64(0)
        rc = self.lib.L.am_process_iq(self._h, ptr, n, flags, out.ctypes.data, cap, C.byref(got))
        if rc == AM_ECAPACITY:
            return self._fetch(int(got.value))
        self._chk(rc)
        return self._received(out, got.value)

    def _receive_buffer(self, cap):
        """One reusable receive buffer per context (a fresh megabyte-sized array per call costs an
        mmap/munmap pair); the caller gets a right-sized copy (_received)."""
        out = getattr(self, "_rxbuf", None)
        if out is None or len(out) < cap:
            out = self._rxbuf = np.zeros(cap, PACKET_DTYPE)
        return out

    @staticmethod
    def _received(out, count):
        # copied as raw bytes: numpy copies a structured array with sub-array fields element by element
        nbytes = int(count) * PACKET_DTYPE.itemsize
        return out.view(np.uint8)[:nbytes].copy().view(PACKET_DTYPE)

    # block-level entry points
    def frontend_work(self, iq):
        f = _iq_f32(iq)
        n = f.size // 2
        bb = np.empty(n, np.float32)
        avg = np.empty(n, np.float32)
        self._chk(self.lib.L.am_frontend_work(self._h, f.ctypes.data if n else None, n, 0,
                                              bb.ctypes.data, avg.ctypes.data))
        return bb, avg

    def preamble_work(self, in_, inavg):
        a = np.ascontiguousarray(in_, np.float32)
        b = np.ascontiguousarray(inavg, np.float32)
        assert a.size == b.size
        n = a.size
        spc = max(int(self.get_rate() / 2e6), 1)
        cap = n // (240 * spc) + 2
        bursts = np.zeros((cap, 240), np.float32)
        tags = np.zeros(cap, TAG_DTYPE)
        got = C.c_uint64(0)
        self._chk(self.lib.L.am_preamble_work(self._h, a.ctypes.data, b.ctypes.data, n, 0, bursts.ctypes.data,
                                              tags.ctypes.data, cap, C.byref(got)))
        return bursts[:got.value], tags[:got.value]

    def preamble_stream(self, in_, inavg, flush=False):
        """The next items of the preamble block's two input streams (am_preamble_stream): this call's hits."""
        a = np.ascontiguousarray(in_, np.float32)
        b = np.ascontiguousarray(inavg, np.float32)
        assert a.size == b.size
        n = a.size
        spc = max(int(self.get_rate() / 2e6), 1)
        cap = n // (240 * spc) + 4
        bursts = np.zeros((cap, 240), np.float32)
        tags = np.zeros(cap, TAG_DTYPE)
        got = C.c_uint64(0)
        self._chk(self.lib.L.am_preamble_stream(self._h, a.ctypes.data if n else None, b.ctypes.data if n else None, n,
                                                AM_F_FLUSH if flush else 0, bursts.ctypes.data, tags.ctypes.data, cap,
                                                C.byref(got)))
        return bursts[:got.value], tags[:got.value]

    def slicer_work(self, bursts, tags):
        b = np.ascontiguousarray(bursts, np.float32).reshape(-1, 240)
        t = np.ascontiguousarray(tags, TAG_DTYPE)
        nb = t.size
        out = np.zeros(max(nb, 1), PACKET_DTYPE)
        got = C.c_uint64(0)
        self._chk(self.lib.L.am_slicer_work(self._h, b.ctypes.data if nb else None, t.ctypes.data if nb else None,
                                            nb, 0, out.ctypes.data, out.size, C.byref(got)))
        return out[:got.value]

    # time-sharded operation
    def shard_halo(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._chk(self.lib.L.am_shard_halo(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def shard_scan(self, iq_with_halo, abs_start, abs_end, total_n, device_ptr=None, more=False):
        """Scan chunk [abs_start, abs_end) of a stream of total_n samples; iq_with_halo covers
        [abs_start - left, abs_end + right) clipped to the stream.  Returns the chunk's exit table.
        more: the stream goes on beyond total_n (AM_F_MORE: no end-of-stream rule)."""
        flags = AM_F_MORE if more else 0
        if device_ptr is not None:
            ptr, flags = int(device_ptr), flags | AM_F_DEVICE_IN
        else:
            f = _iq_f32(iq_with_halo)
            ptr = f.ctypes.data if f.size else None
        cap = 1024
        while True:
            tab = np.zeros(cap, EXIT_DTYPE)
            got = C.c_uint64(0)
            rc = self.lib.L.am_shard_scan(self._h, ptr, abs_start, abs_end, total_n, flags, tab.ctypes.data, cap,
                                          C.byref(got))
            if rc == AM_ECAPACITY:
                cap = int(got.value)
                continue
            self._chk(rc)
            return tab[:got.value]

    def shard_resolve(self, cur_in, capacity=None):
        cap = int(capacity) if capacity is not None else 4096
        out = self._receive_buffer(cap)
        got = C.c_uint64(0)
        rc = self.lib.L.am_shard_resolve(self._h, int(cur_in), out.ctypes.data, cap, C.byref(got))
        if rc == AM_ECAPACITY:
            return self._fetch(int(got.value))
        self._chk(rc)
        return self._received(out, got.value)


class Pipe(object):
    """am_pipe wrapper: `depth` contexts used round-robin by one host thread; independent batches (each a whole
    stream) are submitted and collected in order, the tail of one overlapping the front end of the next."""

    def __init__(self, rate, threshold_db=7.0, use_pmf=True, use_dcblock=False, device=-1, depth=3, lib=None):
        self.lib = lib or default_library()
        err = C.c_int(0)
        self._h = self.lib.L.am_pipe_create(int(device), float(rate), float(threshold_db), int(bool(use_pmf)),
                                            int(bool(use_dcblock)), int(depth), C.byref(err))
        if not self._h:
            raise AirModesError(err.value, self.lib.L.am_last_error(None).decode())
        self._out = np.zeros(4096, PACKET_DTYPE)
        self._held = collections.deque()      # host batches in flight

    def close(self):
        if getattr(self, "_h", None):
            self.lib.L.am_pipe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def in_flight(self):
        return int(self.lib.L.am_pipe_in_flight(self._h))

    def depth(self):
        return int(self.lib.L.am_pipe_depth(self._h))

    def _chk(self, rc):
        if rc != AM_OK:
            raise AirModesError(rc, self.lib.L.am_pipe_last_error(self._h).decode())

    def submit(self, iq):
        """A batch in host memory.  The samples must stay valid until the batch is collected: the (possibly converted)
        array is kept referenced here until then."""
        f = _iq_f32(iq)
        self._chk(self.lib.L.am_pipe_submit(self._h, f.ctypes.data if f.size else None, f.size // 2, AM_F_FLUSH))
        self._held.append(f)

    def submit_device(self, ptr, n):
        self._chk(self.lib.L.am_pipe_submit(self._h, int(ptr), int(n), AM_F_FLUSH | AM_F_DEVICE_IN))
        self._held.append(None)

    def collect(self):
        got = C.c_uint64(0)
        rc = self.lib.L.am_pipe_collect(self._h, self._out.ctypes.data, len(self._out), C.byref(got))
        if rc == AM_ECAPACITY:
            self._out = np.zeros(int(got.value) + 1024, PACKET_DTYPE)
            rc = self.lib.L.am_pipe_collect(self._h, self._out.ctypes.data, len(self._out), C.byref(got))
        if rc != AM_EINVAL or self.in_flight() != len(self._held):   # (the oldest batch left the pipe, done or failed)
            if self._held:
                self._held.popleft()
        self._chk(rc)
        return Context._received(self._out, got.value)

    def submit_multi_device(self, ptr, lengths, zero_gaps=False):
        """K whole streams, packed on the device (Context.multi_pack's layout), as ONE scan in flight (am_pipe_submit_multi)."""
        n = np.ascontiguousarray(lengths, np.uint64)
        flags = AM_F_DEVICE_IN | (AM_F_ZERO_GAPS if zero_gaps else 0)
        self._chk(self.lib.L.am_pipe_submit_multi(self._h, int(ptr), n.size, n.ctypes.data, flags))
        self._held.append(("multi", n.size))

    def collect_multi(self):
        """The oldest scan in flight, submitted by submit_multi_device: K packet arrays."""
        k = self._held[0][1] if self._held and isinstance(self._held[0], tuple) else 0
        pk = self.collect()
        cnt = np.zeros(k, np.uint64)
        self._chk(self.lib.L.am_pipe_multi_counts(self._h, cnt.ctypes.data, cnt.size))
        edges = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        assert int(edges[-1]) == len(pk)
        return [pk[edges[j]:edges[j + 1]] for j in range(k)]

    def last_kernel_ms(self):
        return float(self.lib.L.am_pipe_last_kernel_ms(self._h))


class StreamPipe(object):
    """am_spipe wrapper: ONE continuing stream with `depth` of its consecutive chunks in flight on one GPU (lib/preamble_impl.cc:
    139-246 is a streaming block).  Chunks are DEVICE pointers; every chunk but a stream's first needs front() samples of room in
    front of it in the same allocation (the library copies the previous chunk's tail there unless the stream is contiguous in
    memory); a chunk and the one submitted before it stay valid until it is collected.  Packets of all chunks in order == one
    process_iq over the whole stream."""

    def __init__(self, rate, threshold_db=7.0, use_pmf=True, use_dcblock=False, device=-1, depth=3, lib=None):
        self.lib = lib or default_library()
        err = C.c_int(0)
        self._h = self.lib.L.am_spipe_create(int(device), float(rate), float(threshold_db), int(bool(use_pmf)),
                                             int(bool(use_dcblock)), int(depth), C.byref(err))
        if not self._h:
            raise AirModesError(err.value, self.lib.L.am_last_error(None).decode())
        self._out = np.zeros(4096, PACKET_DTYPE)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.L.am_spipe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != AM_OK:
            raise AirModesError(rc, self.lib.L.am_spipe_last_error(self._h).decode())

    def in_flight(self):
        return int(self.lib.L.am_spipe_in_flight(self._h))

    def depth(self):
        return int(self.lib.L.am_spipe_depth(self._h))

    def front(self):
        f = C.c_uint64(0)
        self._chk(self.lib.L.am_spipe_front(self._h, C.byref(f)))
        return int(f.value)

    def redone(self):
        return int(self.lib.L.am_spipe_redone(self._h))

    def set_rx_time(self, offset, secs, frac):
        self._chk(self.lib.L.am_spipe_set_rx_time(self._h, int(offset), int(secs), float(frac)))

    def submit_device(self, ptr, n, flush=False):
        self._chk(self.lib.L.am_spipe_submit(self._h, int(ptr), int(n), AM_F_DEVICE_IN | (AM_F_FLUSH if flush else 0)))

    def collect(self):
        got = C.c_uint64(0)
        rc = self.lib.L.am_spipe_collect(self._h, self._out.ctypes.data, len(self._out), C.byref(got))
        if rc == AM_ECAPACITY:
            self._out = np.zeros(int(got.value) + 1024, PACKET_DTYPE)
            rc = self.lib.L.am_spipe_collect(self._h, self._out.ctypes.data, len(self._out), C.byref(got))
        self._chk(rc)
        return Context._received(self._out, got.value)

    def last_kernel_ms(self):
        return float(self.lib.L.am_spipe_last_kernel_ms(self._h))

    def run(self, chunks):
        """chunks: iterable of (device pointer, samples); the last one ends the stream.  Keeps depth() chunks in flight; returns
        the packets of every chunk, in order."""
        chunks = list(chunks)
        out = []
        for k, (ptr, n) in enumerate(chunks):
            if self.in_flight() == self.depth():
                out.append(self.collect())
            self.submit_device(ptr, n, flush=(k == len(chunks) - 1))
        while self.in_flight():
            out.append(self.collect())
        return out


class Uploader(object):
    """am_uploader wrapper: `nslots` pinned host buffers with device twins and a copy stream.  buffer(slot) is a numpy
    float32 view of the pinned memory (fill it -- e.g. file.readinto -- then start(slot, n)); wait(slot) blocks until the
    samples are on the device and returns the device pointer.  The copy of one chunk overlaps the scan of the one before."""

    def __init__(self, capacity_complex, nslots=2, device=-1, lib=None):
        self.lib = lib or default_library()
        L = self.lib.L
        L.am_uploader_create.restype = C.c_void_p
        L.am_uploader_create.argtypes = [C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_int)]
        L.am_uploader_destroy.argtypes = [C.c_void_p]
        L.am_uploader_host.restype = C.c_void_p
        L.am_uploader_host.argtypes = [C.c_void_p, C.c_int]
        L.am_uploader_start.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
        L.am_uploader_wait.restype = C.c_void_p
        L.am_uploader_wait.argtypes = [C.c_void_p, C.c_int]
        err = C.c_int(0)
        self.capacity, self.nslots = int(capacity_complex), int(nslots)
        self._h = L.am_uploader_create(int(device), self.capacity, self.nslots, C.byref(err))
        if not self._h:
            raise AirModesError(err.value, "am_uploader_create failed")
        self._views = []
        for k in range(self.nslots):
            ptr = L.am_uploader_host(self._h, k)
            self._views.append(np.ctypeslib.as_array((C.c_float * (2 * self.capacity)).from_address(ptr)))

    def buffer(self, slot):
        return self._views[slot]

    def start(self, slot, n_complex):
        rc = self.lib.L.am_uploader_start(self._h, int(slot), int(n_complex))
        if rc != AM_OK:
            raise AirModesError(rc, "am_uploader_start failed")

    def wait(self, slot):
        ptr = self.lib.L.am_uploader_wait(self._h, int(slot))
        if not ptr:
            raise AirModesError(AM_EHIP, "am_uploader_wait failed")
        return int(ptr)

    def close(self):
        if getattr(self, "_h", None):
            self._views = []
            self.lib.L.am_uploader_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_entries(lib, tables, starts=None, cur_in=0, with_exits=False):
    """am_shard_entry2: the position at which the scan enters every chunk (and, with_exits, leaves it), from the chunks' exit
    tables and the position at which it left the last chunk of the step before (cur_in; 0 at the start of a stream)."""
    n = len(tables)
    tabs = [np.ascontiguousarray(t, EXIT_DTYPE) for t in tables]
    ptrs = (C.c_void_p * n)(*[t.ctypes.data if t.size else None for t in tabs])
    counts = np.array([t.size for t in tabs], np.uint64)
    entry = np.zeros(n, np.uint64)
    leave = np.zeros(n, np.uint64)
    rc = lib.L.am_shard_entry2(ptrs, counts.ctypes.data, n, int(cur_in), entry.ctypes.data, leave.ctypes.data)
    if rc != AM_OK:
        raise AirModesError(rc, "am_shard_entry2")
    return (entry, leave) if with_exits else entry
