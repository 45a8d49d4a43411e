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


def per_kernel():
    """every am_k_* kernel of a step: rocprofv3's average duration (kernel trace), FETCH_SIZE x 2 and WRITE_SIZE per dispatch"""
    out = {}
    for line in text.split("== kernel stats")[1].split("==")[0].splitlines():
        m = re.match(r"\s*(?:void )?(am_k_[A-Za-z0-9_]+(?:<[^>]*>)?).*?calls\s+(\d+).*?avg_ns\s+([0-9.]+)", line)
        if m:
            out.setdefault(m.group(1), {})["avg_us_rocprof"] = float(m.group(3)) / 1e3
            out[m.group(1)]["calls"] = int(m.group(2))
    for section, key, mul in (("FETCH_SIZE", "fetch_bytes_corrected", 2048.0), ("WRITE_SIZE", "write_bytes", 1024.0)):
        if "== %s per dispatch" % section not in text:
            continue
        for line in text.split("== %s per dispatch" % section)[1].split("==")[0].splitlines():
            m = re.match(r"\s*(?:void )?(am_k_[A-Za-z0-9_]+(?:<[^>]*>)?).*?mean ([0-9.e+]+)", line)
            if m:
                out.setdefault(m.group(1), {})[key] = int(float(m.group(2)) * mul)
    return out


def tree_sha():
    """one hash over the product library's sources: the per-step figures belong to ONE version of the kernels"""
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gr-air-modes_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".inc")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


kernel, fetch = mean_of("FETCH_SIZE")
_, write = mean_of("WRITE_SIZE")
kernels = per_kernel()
ksrc = "am_fe3.hip" if kernel.startswith("am_k_fe3") else ("am_fe4.hip" if kernel.startswith("am_k_fe4") else "am_fe2.hip")
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
with open(os.path.join(root, "gr-air-modes_amd", "csrc", ksrc), "rb") as kf:
    ksha = hashlib.sha256(kf.read()).hexdigest()[:16]
doc = {"workload": workload, "kernel": kernel, "kernel_source": ksrc, "kernel_source_sha16": ksha, "fetch_size_kib_raw": fetch, "write_size_kib_raw": write,
       "fetch_bytes_corrected": int(fetch * 1024 * 2), "write_bytes": int(write * 1024),
       "traffic_bytes": int(fetch * 1024 * 2 + write * 1024), "source": sys.argv[1],
       # every kernel of a step, and their sum: what the whole path moves per step
       "kernels": kernels,
       "kernel_ms_rocprof": next((v.get("avg_us_rocprof", 0.0) / 1e3 for k, v in kernels.items() if k.startswith(kernel.split("<")[0])), None),
       "path_traffic_bytes": int(sum(v.get("fetch_bytes_corrected", 0) + v.get("write_bytes", 0) for k, v in kernels.items() if k != "am_k_scan_u32")),
       "path_kernel_us_rocprof": sum(v.get("avg_us_rocprof", 0.0) for k, v in kernels.items() if v.get("calls", 0) > 1),
       "path_launches_per_step": sum(1 for k, v in kernels.items() if v.get("calls", 0) > 1),
       "path_source_sha16": tree_sha(),
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
