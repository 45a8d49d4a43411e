"""File-source subset of the reference's command-line receiver (apps/modes_rx:32-110,
python/radio.py:90-118,221-234):

    python -m air_modes.modes_rx -s capture.cf32 -r 2e6 [-T 7.0] [--no-pmf] [-l lat,lon] [-n] [--raw]

Reads a gr_complex file (interleaved little-endian float32 I,Q -- what
blocks.file_source(gr.sizeof_gr_complex, path) reads), pushes it through air_modes.rx_path on
the GPU chunk by chunk, and prints one line per decoded report in the reference's format
(python/msprint.py), or the slicer's raw messages with --raw.

Input slower than 4 Msps is resampled to 4 Msps first, as modes_radio does (python/radio.py:49-53) -- with this
package's own polyphase interpolator (air_modes/resample.py: GNU Radio's taps are not reproducible here, so that
stage is compared by packet recall, not bit for bit); `--no-resample` processes at the rate given, which is what
"through rx_path directly" means for the 2 Msps capture of BASELINE.json configs[0..1].

A reader thread keeps one chunk ahead of the GPU (file read and resampling of chunk k+1 run under the GPU call of
chunk k).  Differences from modes_radio, all of them outside the demodulator: no live sources (UHD / osmocom /
UDP) and no ZeroMQ relay (the parser is called directly).
"""
import argparse
import sys

import numpy as np

from . import resample


def build_parser():
    ap = argparse.ArgumentParser(prog="modes_rx", description=__doc__.split("\n\n")[0])
    ap.add_argument("-s", "--source", required=True, help="gr_complex (cf32) file")        # radio.py:94
    ap.add_argument("-r", "--rate", type=float, default=4e6, help="sample rate [default=%(default)s]")   # :112
    ap.add_argument("-T", "--threshold", type=float, default=7.0,
                    help="pulse detection threshold above noise in dB [default=%(default)s]")            # :114
    ap.add_argument("-p", "--pmf", action="store_true", default=True, help="use pulse matched filtering")  # :116
    ap.add_argument("--no-pmf", dest="pmf", action="store_false")
    ap.add_argument("-d", "--dcblock", action="store_true", default=False,
                    help="use a DC blocking filter (best for HackRF Jawbreaker)")                         # :118
    ap.add_argument("-l", "--location", default=None, help="receiver position as lat,lon (enables range/bearing "
                    "and surface positions)")                                                            # modes_rx:40
    ap.add_argument("-n", "--no-print", action="store_true", help="do not print decoded reports")       # modes_rx:45
    ap.add_argument("--raw", action="store_true", help="print the slicer's raw messages instead of parsed reports")
    ap.add_argument("--chunk", type=int, default=1 << 22, help="complex samples per GPU call")
    ap.add_argument("--no-resample", action="store_true", help="process input below 4 Msps at its own rate")
    return ap


def main(argv=None, out=None):
    args = build_parser().parse_args(argv)
    out = out or sys.stdout
    from . import cpr_decoder, make_parser, msg_queue, output_print, pubsub, rx_path

    queue = msg_queue()
    rx_rate, resampler = args.rate, None
    if args.rate < 4e6 and not args.no_resample:                      # radio.py:49-53
        # on the GPU, bit-identical to resample.arb_resampler (its definition); the output stays on the device
        rx_rate, resampler = 4e6, resample.gpu_resampler(4e6 / args.rate)
    rx = rx_path(rx_rate, args.threshold, queue, use_pmf=args.pmf, use_dcblock=args.dcblock)
    publisher = pubsub()
    feed = make_parser(publisher)
    my_position = [float(n) for n in args.location.split(",")] if args.location else None
    if not args.no_print and not args.raw:
        output_print(cpr_decoder(my_position), publisher, callback=lambda line: print(line, file=out))
    print("Using file source %s" % args.source, file=sys.stderr)
    print("Rate is %i" % int(args.rate), file=sys.stderr)

    # File -> pinned host buffer -> device, two buffers in flight (python/radio.py:221-234's file source): the reader thread
    # fills one pinned buffer from the file and starts its copy to the device while the GPU still works on the chunk before
    # it; the receive path (and the interpolator in front of it) take device pointers, so a sample crosses PCIe once and
    # never comes back.
    import queue as _queue
    import threading
    from . import _capi
    nslots = 2
    drain = resample.TAPS_PER_PHASE if resampler is not None else 0
    up = _capi.Uploader(args.chunk + drain, nslots=nslots)
    ready = _queue.Queue()                    # (slot, samples, last) in stream order, or an exception
    free = _queue.Queue()
    for k in range(nslots):
        free.put(k)

    def reader():
        try:
            with open(args.source, "rb") as f:
                while True:
                    slot = free.get()
                    buf = up.buffer(slot)
                    want = 8 * args.chunk
                    got = f.readinto(memoryview(buf).cast("B")[:want])
                    n = got // 8                                      # whole complex samples
                    last = got < want
                    if last and drain:
                        # drain: the interpolator holds its last outputs back until it has seen what follows them
                        buf[2 * n:2 * (n + drain)] = 0.0
                        n += drain
                    up.start(slot, n)
                    ready.put((slot, n, last))
                    if last:
                        return
        except Exception as err:                                      # surfaces in the consumer
            ready.put((err, 0, True))

    th = threading.Thread(target=reader, daemon=True)
    th.start()
    while True:
        slot, n, last = ready.get()
        if isinstance(slot, Exception):
            raise slot
        ptr = up.wait(slot)
        if resampler is not None:
            ptr, n = resampler.work_device_in(ptr, n)                 # (python/radio.py:49-53) 2 -> 4 Msps on the GPU
        rx.work_device(ptr, n, flush=last)
        free.put(slot)                                                # (the call returned: the device is done with the slot)
        while not queue.empty_p():
            text = queue.delete_head().to_string()
            if args.raw:
                if not args.no_print:
                    print(text, file=out)
            else:
                try:
                    feed(text)
                except (IndexError, KeyError, ValueError) as err:
                    # table lookups the reference does unguarded (e.g. emitter category 7 in category
                    # set B, python/parse.py:274-280): there the exception ends the subscriber thread,
                    # here the report is skipped and the receiver keeps going
                    print("skipped %s: %r" % (text.split()[0], err), file=sys.stderr)
        if last:
            break
    th.join()
    up.close()
    print("%d samples, %d packets" % (rx.samples, rx.packets), file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
