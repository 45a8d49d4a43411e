"""Generate tests/golden/*.npz with the REFERENCE's own code (oracle/_ref, compiled by path
from /root/reference) -- run in the development container only; the fixtures travel.

Per fixture: a seeded synthetic capture (IQ), the canonical front-end streams bb/avg (our
documented summation order; the GNU Radio front end is not in /root/reference so this stage
is "parity unpinned"), and what the reference's preamble_impl.cc + slicer_impl.cc +
modes_crc.cc produce from those two streams: bursts, tag timestamps, message texts.
Also crc_kat.json: modes_check_crc() known answers.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle  # noqa: E402
import synth   # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

# (name, rate, n, lambda/s, seed, thr_db, use_pmf)
CASES = [
    ("g_2msps", 2e6, 40000, 4000.0, 101, 7.0, True),
    ("g_4msps", 4e6, 48000, 6000.0, 102, 7.0, True),
    ("g_4msps_nopmf_thr5", 4e6, 48000, 6000.0, 103, 5.0, False),
    ("g_20msps", 20e6, 120000, 5000.0, 104, 7.0, True),
    ("g_64msps", 64e6, 256000, 5000.0, 105, 7.0, True),
]


# rates that are not multiples of 2 MHz (round 4): 2.5 and 3.125 samples per chip -- the reference's float geometry
FRAC_CASES = [
    ("g_5msps", 5e6, 60000, 5000.0, 106, 7.0, True),
    ("g_6p25msps", 6.25e6, 70000, 5000.0, 107, 7.0, True),
]


def main(cases=None, crc=True):
    assert oracle.have_ref() or os.path.isdir("/root/reference/lib"), "needs /root/reference"
    oracle.build()
    os.makedirs(OUT, exist_ok=True)
    for name, rate, n, lam, seed, thr, pmf in (cases or CASES):
        spc = int(rate / 2e6)
        iq, truth = synth.synth_capture(rate, n, lam, seed, snr_db=(12.0, 35.0))
        bb, avg = oracle.frontend(iq, spc, pmf)
        rb, rt, rmsgs, keep = oracle.ref_preamble_slicer(bb, avg, spc, thr, rate)
        # the reference (run on a zero-padded buffer) also reports hits that start too close to
        # the end of the stream; apply the canonical end-of-stream rule (they are always last)
        if not keep.all():
            nk = int(keep.sum())
            assert keep[:nk].all()
            limit = int(rt["sample"][nk - 1]) if nk else -1
            rb, rt = rb[:nk], rt[:nk]

            def msg_sample(m):
                f = m.split()
                return int(f[3]) * int(rate) + int(round(float(f[4]) * rate))
            rmsgs = [m for m in rmsgs if msg_sample(m) <= limit]
        np.savez_compressed(
            os.path.join(OUT, name + ".npz"), iq=iq, rate=np.float64(rate), thr_db=np.float32(thr),
            use_pmf=np.int32(pmf), bb_sha256=np.array(hashlib.sha256(bb.tobytes()).hexdigest()),
            avg_sha256=np.array(hashlib.sha256(avg.tobytes()).hexdigest()), ref_bursts=rb, ref_tag_sample=rt["sample"],
            ref_tag_secs=rt["secs"], ref_tag_frac=rt["frac"], ref_msgs=np.array(rmsgs),
            truth_frames=np.array([t["frame"] for t in truth]))
        print("%s: %d samples, %d truth bursts, %d reference tags, %d reference messages"
              % (name, n, len(truth), len(rt), len(rmsgs)))
    if not crc:
        return
    # CRC known answers straight from the reference's modes_check_crc
    rng = np.random.default_rng(7)
    kat = []
    frames = ["8D4840D6202CC371C32CE0576098", "8D40621D58C382D690C8AC2863A7", "02E197B0A9A3B1",
              "5D4840D6000000", "00000000000000", "FFFFFFFFFFFFFFFFFFFFFFFFFFFF"]
    frames += [rng.integers(0, 256, 14, dtype=np.uint8).tobytes().hex() for _ in range(20)]
    frames += [rng.integers(0, 256, 7, dtype=np.uint8).tobytes().hex() for _ in range(20)]
    for h in frames:
        b = np.frombuffer(bytes.fromhex(h), np.uint8).copy()
        for nb in sorted({len(b) - 3, len(b)}):
            kat.append({"hex": h.lower(), "nbytes": int(nb), "crc": int(oracle.ref().ref_crc24(b, nb))})
    with open(os.path.join(OUT, "crc_kat.json"), "w") as f:
        json.dump(kat, f, indent=0)
    print("crc_kat.json: %d vectors" % len(kat))


# "rx_time" stream tags: (name, rate, n, lambda/s, seed); the capture is silenced in front of every tag
# so that the reference's scheduler-dependent early latch cannot matter (oracle/ref_driver.cc)
RX_TIME_CASES = [("rxtime_4msps", 4e6, 90000, 6000.0, 111), ("rxtime_20msps", 20e6, 140000, 14000.0, 112)]


def rx_time_tags(n, spc):
    return [(0, 1600000000, 0.125), (n // 3 + 5, 1600000007, 0.9999995), (2 * n // 3, 3, 0.5),
            (2 * n // 3, 1700000000, 0.75)]


def gen_rx_time():
    for name, rate, n, lam, seed in RX_TIME_CASES:
        spc = int(rate / 2e6)
        iq, _ = synth.synth_capture(rate, n, lam, seed, snr_db=(12.0, 35.0))
        rx = rx_time_tags(n, spc)
        for off, _, _ in rx[1:]:
            iq[off - 900 * spc:off + 32] = 0
        bb, avg = oracle.frontend(iq, spc, True)
        rb, rt, rmsgs, keep = oracle.ref_preamble_slicer(bb, avg, spc, 7.0, rate, rx_time=rx)
        nk = int(keep.sum())
        assert keep[:nk].all()
        # messages of hits past the end-of-stream rule come last, if any; count the kept ones
        _, _, plain_msgs, _ = oracle.ref_preamble_slicer(bb, avg, spc, 7.0, rate)
        opk = oracle.demod(iq, rate, 7.0, True, rx_time=rx)
        rmsgs = rmsgs[:len(opk)]
        assert oracle.format_messages(opk) == rmsgs, "oracle and reference disagree"
        assert len(set(m.split()[3] for m in rmsgs)) >= 3, "tags must matter"
        np.savez_compressed(
            os.path.join(OUT, name + ".npz"), iq=iq, rate=np.float64(rate), thr_db=np.float32(7.0),
            use_pmf=np.int32(1), rx_offset=np.array([t[0] for t in rx], np.uint64),
            rx_secs=np.array([t[1] for t in rx], np.uint64), rx_frac=np.array([t[2] for t in rx], np.float64),
            ref_tag_sample=rt["sample"][:nk], ref_tag_secs=rt["secs"][:nk], ref_tag_frac=rt["frac"][:nk],
            ref_msgs=np.array(rmsgs))
        print("%s: %d samples, %d reference tags, %d reference messages" % (name, n, nk, len(rmsgs)))


if __name__ == "__main__":
    if sys.argv[1:] == ["rx_time"]:
        oracle.build()
        gen_rx_time()          # only the rx_time fixtures (the others stay byte-identical in git)
    elif sys.argv[1:] == ["frac"]:
        main(FRAC_CASES, crc=False)   # only the fractional-rate fixtures
    else:
        main()
        gen_rx_time()
