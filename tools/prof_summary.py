"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a small text summary that
is committed under profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"


def first(pattern):
    f = sorted(glob.glob(os.path.join(root, pattern), recursive=True))
    return f[0] if f else None


f = first("prof_stats/**/*kernel_stats.csv")
if f:
    print("== kernel stats (%s)" % f)
    rows = list(csv.DictReader(open(f)))
    for r in rows[:20]:
        print("%-60s calls %6s total_ns %14s avg_ns %12s pct %6s" % (
            r.get("Name", "")[:60], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))
for tag, counter in (("prof_fetch", "FETCH_SIZE"), ("prof_write", "WRITE_SIZE")):
    f = first("%s/**/*counter_collection.csv" % tag)
    if not f:
        continue
    print("== %s per dispatch (%s)" % (counter, f))
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == counter:
            acc[r.get("Kernel_Name", "")[:60]].append(float(r.get("Counter_Value", 0)))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:12]:
        print("%-60s dispatches %5d mean %.6g max %.6g (raw counter units: KiB)" % (k, len(v), sum(v) / len(v), max(v)))
