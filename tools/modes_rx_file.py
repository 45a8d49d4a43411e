#!/usr/bin/env python3
"""File-source subset of apps/modes_rx (apps/modes_rx:32-95, python/radio.py:221-234):

    python tools/modes_rx_file.py -s capture.cf32 -r 2e6 [-T 7.0] [--no-pmf] [--chunk 4000000]

Reads a gr_complex file (interleaved little-endian float32 I,Q -- what
blocks.file_source(gr.sizeof_gr_complex, path) reads), pushes it through air_modes.rx_path on
the GPU chunk by chunk and prints the slicer's raw messages ("packet list to stdout").
Unlike modes_radio it does not resample sub-4-Msps input to 4 Msps (radio.py:49-53); the
rate given is the processing rate.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-air-modes_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-s", "--source", required=True, help="gr_complex (cf32) file")
    ap.add_argument("-r", "--rate", type=float, default=4e6)              # radio.py:112
    ap.add_argument("-T", "--threshold", type=float, default=7.0)         # radio.py:114
    ap.add_argument("-p", "--pmf", action="store_true", default=True)     # radio.py:116
    ap.add_argument("--no-pmf", dest="pmf", action="store_false")
    ap.add_argument("--chunk", type=int, default=1 << 22, help="complex samples per GPU call")
    args = ap.parse_args()

    import air_modes
    q = air_modes.msg_queue()
    rx = air_modes.rx_path(args.rate, args.threshold, q, use_pmf=args.pmf)
    print("Using file source %s" % args.source, file=sys.stderr)
    print("Rate is %i" % int(args.rate), file=sys.stderr)
    with open(args.source, "rb") as f:
        while True:
            raw = np.fromfile(f, dtype=np.float32, count=2 * args.chunk)
            last = raw.size < 2 * args.chunk
            rx.work(raw[: raw.size // 2 * 2], flush=last)
            while not q.empty_p():
                print(q.delete_head().to_string())
            if last:
                break
    print("%d samples, %d packets" % (rx.samples, rx.packets), file=sys.stderr)


if __name__ == "__main__":
    main()
