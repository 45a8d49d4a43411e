"""profiles/current_traffic.json from a gpu_prof.sh summary (rocprofv3 FETCH_SIZE / WRITE_SIZE passes).

    python tools/update_traffic.py profiles/<round>/summary.txt 64msps
FETCH_SIZE is doubled (gfx950 reports half the bytes of a wide coalesced read: MI355X_MICROARCH.md,
HBM section); WRITE_SIZE is taken as reported."""
import hashlib
import json
import os
import re
import sys

text = open(sys.argv[1]).read()
workload = sys.argv[2] if len(sys.argv) > 2 else "64msps"


def mean_of(section):
    part = text.split("== %s per dispatch" % section)[1]
    m = re.search(r"am_k_fe3\(.*?mean ([0-9.e+]+)", part)
    if m:
        return "am_k_fe3", float(m.group(1))
    m = re.search(r"am_k_fe4<(\d+), (\d+), (\d+)>.*?mean ([0-9.e+]+)", part)
    if m:
        return "am_k_fe4<%s,G>" % m.group(1), float(m.group(4))
    m = re.search(r"am_k_fe2<(\d+), (\d+)>.*?mean ([0-9.e+]+)", part)
    return "am_k_fe2<%s,%s>" % (m.group(1), m.group(2)), float(m.group(3))


kernel, fetch = mean_of("FETCH_SIZE")
_, write = mean_of("WRITE_SIZE")
ksrc = "am_fe3.hip" if kernel.startswith("am_k_fe3") else ("am_fe4.hip" if kernel.startswith("am_k_fe4") else "am_fe2.hip")
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
with open(os.path.join(root, "gr-air-modes_amd", "csrc", ksrc), "rb") as kf:
    ksha = hashlib.sha256(kf.read()).hexdigest()[:16]
doc = {"workload": workload, "kernel": kernel, "kernel_source": ksrc, "kernel_source_sha16": ksha, "fetch_size_kib_raw": fetch, "write_size_kib_raw": write,
       "fetch_bytes_corrected": int(fetch * 1024 * 2), "write_bytes": int(write * 1024),
       "traffic_bytes": int(fetch * 1024 * 2 + write * 1024), "source": sys.argv[1],
       "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `python bench.py --steps 2 "
               "--warmup 1 --no-cpu-baseline --no-extra [--workload ...]`; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 "
               "reports half of a coalesced stream); WRITE_SIZE uncalibrated (streaming kernels: candidate bitmap + sparse bb / reference level "
               "runs; tile kernel: dense bb + sparse reference level + candidate lists)"}
# one entry per workload
path = os.path.join(root, "profiles", "current_traffic.json")
try:
    allw = json.load(open(path))
    if "workload" in allw:                        # (the single-entry form of rounds 1-2)
        allw = {allw["workload"]: allw}
except Exception:
    allw = {}
allw[workload] = doc
json.dump(allw, open(path, "w"), indent=1)
print(doc)
