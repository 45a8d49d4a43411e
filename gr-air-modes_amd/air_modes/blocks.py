"""The two native blocks of the reference, backed by the HIP library.

gr::air_modes::preamble  (include/gr_air_modes/preamble.h:36-46, lib/preamble_impl.cc)
gr::air_modes::slicer    (include/gr_air_modes/slicer.h:37-42,  lib/slicer_impl.cc)

GNU Radio's scheduler calls general_work()/work() on circular buffers; here the owner of
the block calls work() on whole arrays (the GPU path batches a stream chunk at a time).
"""
import numpy as np

from . import _capi
from .msg_queue import message


class preamble(object):
    """air_modes.preamble(channel_rate, threshold_db): Mode-S preamble detector.

    Inputs: stream 0 = received (pulse-matched) power, stream 1 = its moving average
    (lib/preamble_impl.cc:43); output: 240 soft chips per detected preamble plus a
    "preamble_found" tag carrying the timestamp (:219-232)."""

    def __init__(self, channel_rate, threshold_db, device=-1, lib=None):
        self._ctx = _capi.Context(float(channel_rate), float(threshold_db), use_pmf=True, device=device, lib=lib)

    def set_rate(self, channel_rate):
        self._ctx.set_rate(float(channel_rate))

    def set_threshold(self, threshold_db):
        self._ctx.set_threshold(float(threshold_db))

    def get_rate(self):
        return self._ctx.get_rate()

    def get_threshold(self):
        return self._ctx.get_threshold()

    def work(self, in0, in1, rx_time=()):
        """Run the detector over whole streams; returns (bursts[n,240] float32, tags).  rx_time: the
        (offset, secs, frac) "rx_time" tags on input 0 (lib/preamble_impl.cc:165-170), ascending."""
        self._ctx.reset()                      # every call is a stream of its own: item 0, no tags
        for tag in rx_time:
            self._ctx.set_rx_time(*tag)
        return self._ctx.preamble_work(in0, in1)

    def general_work(self, in0, in1, rx_time=(), flush=False):
        """The block under a scheduler (lib/preamble_impl.cc:139: general_work, called again and again on the next
        items): the next items of both input streams in, this call's hits out; the calls together give what one
        work() over the concatenation gives.  Decisions wait for 244 chips of look-ahead (:150,212); the undecided
        tail is carried inside the block.  rx_time offsets are stream-absolute; flush=True: these are the stream's
        last items, the next call starts a new stream.  reset() drops the carried state."""
        for tag in rx_time:
            self._ctx.set_rx_time(*tag)
        return self._ctx.preamble_stream(in0, in1, flush=flush)

    def reset(self):
        self._ctx.reset()


class slicer(object):
    """air_modes.slicer(queue): PPM bit slicer + framer + CRC; posts one text message per
    accepted reply to `queue` (lib/slicer_impl.cc:186-194)."""

    def __init__(self, queue, device=-1, lib=None, _ctx=None):
        self._queue = queue
        self._ctx = _ctx or _capi.Context(4e6, 7.0, use_pmf=True, device=device, lib=lib)
        self._first = True          # the member ostringstream's precision quirk, slicer_impl.h:43

    def post(self, packets):
        """Format accepted packets exactly like the reference and hand them to the queue."""
        if len(packets) == 0:
            return 0
        # one call into the library per batch (am_format_messages), not one per packet
        for text in self._ctx.lib.format_messages(packets, self._first):
            self._queue.handle(message.make_from_string(text))
        self._first = False
        return len(packets)

    def work(self, bursts, tags):
        """Slice tagged bursts (as produced by preamble.work); returns the accepted packets."""
        pk = self._ctx.slicer_work(bursts, tags)
        self.post(pk)
        return pk
