#!/usr/bin/env python3
"""How far is the canonical summation order of the two moving averages from what GNU Radio 3.8's
moving_average_ff computes?  GNU Radio's block keeps a running sum (add the newest, subtract the oldest) that it
re-seeds at the start of every work() call and every <= 4096 outputs inside one; where those points fall depends on
the scheduler.  This script runs the oracle's front end both ways on seeded captures

    rates 2 / 4 / 20 / 64 Msps  x  10 seeds  x  re-seed grids {4096 @ 0, 4096 @ 1000, 4096 @ 3000, 2048 @ 0}

and reports, per rate: the largest distance of bb and of the reference level in units in the last place and relative
to the largest sample, and how many decoded packets (payload, syndrome, timestamp) differ from the canonical run.
It also counts the packets a real GNU Radio run never delivers at the end of a file (python/rx_path.py under the
scheduler: lib/preamble_impl.cc:61 asks for 1 + 480 spc items per call and needs 240 spc of room (:212), the slicer
holds the last 480 items = two bursts back (lib/slicer_impl.cc:62,107,197)): the canonical semantics ("one infinite
work() call") deliver them.

    python tools/frontend_sensitivity.py [--quick] [--json out.json]      (CPU only; about two minutes)
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

GRIDS = ((4096, 0), (4096, 1000), (4096, 3000), (2048, 0))
CASES = (  # rate, samples, bursts per second
    (2e6, 1_000_000, 1500.0), (4e6, 1_000_000, 1500.0), (20e6, 2_000_000, 5000.0), (64e6, 3_200_000, 20000.0))


def ulp_distance(a, b):
    """Largest distance in representable floats over the samples that are at least a thousandth of the largest one
    (near zero the running sum leaves a residue of either sign where the canonical order gives the small true value:
    a relative figure says more there, and is reported next to this one)."""
    keep = (a >= 1e-3 * a.max()) & (b > 0)
    ia = a[keep].view(np.int32).astype(np.int64)
    ib = b[keep].view(np.int32).astype(np.int64)
    return int(np.abs(ia - ib).max()) if ia.size else 0


def keyed(oracle, bb, avg, spc, rate, thr=7.0):
    bursts, tags = oracle.preamble_scan(bb, avg, spc, thr, rate)
    pk = oracle.slice_bursts(bursts, tags)
    return [(int(p["sample"]), bytes(p["data"]), int(p["crc"])) for p in pk]


def study(rate, n, lam, seeds, grids=GRIDS):
    import oracle
    import synth
    spc = int(rate / 2e6)
    out = {"rate": rate, "samples": n, "seeds": len(seeds), "packets": 0, "bb_ulp": 0, "avg_ulp": 0, "bb_rel": 0.0,
           "avg_rel": 0.0, "missing": 0, "extra": 0, "runs": 0, "eof_tail_packets": 0}
    tail = (1 + 480 * spc) + 240 * spc                   # items the preamble block never examines at the end of a file
    for seed in seeds:
        iq, _ = synth.synth_capture(rate, n, lam, 1000 + seed)
        bb, avg = oracle.frontend(iq, spc, True)
        ref = keyed(oracle, bb, avg, spc, rate)
        out["packets"] += len(ref)
        # end of file under the GNU Radio scheduler: nothing that starts in the unexamined tail, and the slicer
        # never sees the last two bursts the preamble block produced
        in_tail = [p for p in ref if p[0] >= n - tail]
        out["eof_tail_packets"] += len(in_tail) + min(2, len(ref) - len(in_tail))
        for chunk, first in grids:
            b2, a2 = oracle.frontend(iq, spc, True, running_chunk=chunk, running_first=first)
            out["bb_ulp"] = max(out["bb_ulp"], ulp_distance(bb, b2))
            out["avg_ulp"] = max(out["avg_ulp"], ulp_distance(avg, a2))
            out["bb_rel"] = max(out["bb_rel"], float(np.abs(bb - b2).max() / bb.max()))
            out["avg_rel"] = max(out["avg_rel"], float(np.abs(avg - a2).max() / avg.max()))
            got = keyed(oracle, b2, a2, spc, rate)
            sr, sg = set(ref), set(got)
            out["missing"] += len(sr - sg)
            out["extra"] += len(sg - sr)
            out["runs"] += 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="2 seeds, shorter captures (the CI test)")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    import oracle
    oracle.build()
    rows = []
    for rate, n, lam in CASES:
        rows.append(study(rate, n // (4 if args.quick else 1), lam, range(2 if args.quick else 10)))
    print("| rate | captures x grids | packets (canonical) | max bb distance | max reference-level distance | "
          "packets missing / extra vs canonical | packets GNU Radio drops at the end of a file |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %g Msps | %d x %d | %d | %d ulp (%.1e of the largest sample) | %d ulp (%.1e) | %d / %d (of %d) | %d of %d |"
              % (r["rate"] / 1e6, r["seeds"], len(GRIDS), r["packets"], r["bb_ulp"], r["bb_rel"], r["avg_ulp"], r["avg_rel"],
                 r["missing"], r["extra"], r["packets"] * len(GRIDS), r["eof_tail_packets"], r["packets"]))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
