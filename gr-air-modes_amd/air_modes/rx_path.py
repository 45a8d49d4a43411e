"""air_modes.rx_path -- the receive hier block (python/rx_path.py:25-88), MI355X edition.

The reference wires five CPU blocks (complex_to_mag_squared, two moving_average_ff,
preamble, slicer) inside a gr.hier_block2 and lets the GNU Radio scheduler stream through
them.  Here the whole chain is ONE batched GPU sink: work(iq) pushes a chunk of the complex
stream through am_process_iq and posts the resulting messages to the queue.  Constructor
signature, setter/getter names and the message format are the reference's.
"""
import numpy as np

from . import _capi
from .blocks import slicer as _slicer


class rx_path(object):
    def __init__(self, rate, threshold, queue, use_pmf=False, use_dcblock=False, device=-1, lib=None):
        self._rate = int(rate)
        self._threshold = threshold
        self._queue = queue
        self._spc = int(rate / 2e6)
        self._use_pmf = bool(use_pmf)
        self._ctx = _capi.Context(float(self._rate), float(threshold), use_pmf=use_pmf,
                                  use_dcblock=use_dcblock, device=device, lib=lib)
        self._slicer = _slicer(queue, _ctx=self._ctx)
        self.packets = 0
        self.samples = 0

    # --- reference surface: python/rx_path.py:67-87 ---
    def set_rate(self, rate):
        self._ctx.set_rate(float(int(rate)))
        self._rate = int(rate)
        self._spc = int(rate / 2e6)

    def set_threshold(self, threshold):
        self._ctx.set_threshold(float(threshold))
        self._threshold = threshold

    def set_pmf(self, pmf):
        # the reference's setter is a no-op too ("must be done when top block is stopped")
        pass

    def get_pmf(self, pmf=None):
        return self._ctx.get_pmf()

    def get_threshold(self):
        return self._ctx.get_threshold()

    # --- what the scheduler does for the reference: push samples through ---
    def set_rx_time(self, offset, secs, frac):
        """What an "rx_time" stream tag does in the reference (lib/preamble_impl.cc:165-170): item
        `offset` of the stream was received at (secs, frac); later packets are stamped from it."""
        self._ctx.set_rx_time(offset, secs, frac)

    def work(self, iq, flush=False, rx_time=()):
        """Consume a chunk of the gr_complex stream (complex64 array or interleaved float32);
        flush=True marks the end of the stream.  rx_time: the (offset, secs, frac) "rx_time" tags that
        fall into this chunk, offsets counted over the whole stream.  Returns the accepted packets
        (structured array) after posting their messages to the queue."""
        for tag in rx_time:
            self._ctx.set_rx_time(*tag)
        pk = self._ctx.process_iq(iq, flush=flush)
        return self._account(pk, (np.asarray(iq).size // (1 if np.iscomplexobj(iq) else 2)))

    def work_device(self, dev_ptr, n_complex, flush=False):
        """Same with the samples already resident in this GPU's memory (interleaved f32)."""
        pk = self._ctx.process_iq_device(dev_ptr, n_complex, flush=flush)
        return self._account(pk, n_complex)

    def _account(self, pk, n):
        self._slicer.post(pk)
        self.packets += len(pk)
        self.samples += int(n)
        return pk

    def context(self):
        return self._ctx


class rx_path_bank(object):
    """K receivers of one kind on one GPU: what K rx_path instances (python/rx_path.py:25-88) give on K finite captures, from
    ONE scan per call (am_process_multi: the captures lie behind one another in one buffer, zeros between them).  queues: one
    gr.msg_queue per receiver -- receiver j's messages go to queues[j], formatted as its own slicer would (the first message of
    every receiver carries the six significant digits of a fresh ostringstream, lib/slicer_impl.cc:186-192).  Every call is a set
    of WHOLE streams (item counts and time stamps start at 0): the batch form of rx_path.work(capture, flush=True)."""

    def __init__(self, rate, threshold, queues, use_pmf=False, device=-1, lib=None):
        self._ctx = _capi.Context(float(int(rate)), float(threshold), use_pmf=use_pmf, device=device, lib=lib)
        self._slicers = [_slicer(q, _ctx=self._ctx) for q in queues]
        self.packets = [0] * len(queues)

    def work(self, captures):
        """captures: K complex64 (or interleaved float32) arrays, one per receiver; returns the K packet arrays."""
        assert len(captures) == len(self._slicers)
        buf, n = self._ctx.multi_pack(captures)
        return self._post(self._ctx.process_multi(buf, n))

    def work_device(self, dev_ptr, lengths):
        """The same with the packed buffer (layout: context().multi_layout(lengths), zeros between the streams) already on the GPU."""
        return self._post(self._ctx.process_multi(None, lengths, device_ptr=dev_ptr))

    def _post(self, per_stream):
        for j, pk in enumerate(per_stream):
            self._slicers[j].post(pk)
            self.packets[j] += len(pk)
        return per_stream

    def context(self):
        return self._ctx
