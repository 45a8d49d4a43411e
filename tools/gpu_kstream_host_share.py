"""Where does a K-streams-per-scan step spend its time?   gpurun -- 'python tools/gpu_kstream_host_share.py [rate] [K]'
(the knobs build's host trace: us per call inside am_process_multi, by phase; the packets' way into per-stream lists is one
linear pass over the accepted packets on the host -- sort_into_streams, am_capi.hip -- timed here on its own too)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gr-air-modes_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
os.environ["AIRMODES_HOST_TRACE"] = "1"
import numpy as np
import torch
import synth
from air_modes import _capi

rate = float(sys.argv[1]) if len(sys.argv) > 1 else 20e6
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(float(sys.argv[3])) if len(sys.argv) > 3 else int(rate)
lib = _capi.Library(os.path.join(ROOT, "tests", "gpu_variants", "libairmodes_hip_knobs.so"))
ctx = _capi.Context(rate, 7.0, True, device=0, lib=lib)
lam = float(sys.argv[4]) if len(sys.argv) > 4 else 20000.0
streams = [synth.synth_capture(rate, n, lam, 100 + j)[0] for j in range(K)]
buf, lens = ctx.multi_pack(streams)
d = torch.from_numpy(buf.view(np.float32)).cuda()
for _ in range(3):
    got = ctx.process_multi(None, lens, device_ptr=d.data_ptr())
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
gpu_ms = []
for _ in range(N):
    got = ctx.process_multi(None, lens, device_ptr=d.data_ptr())
    gpu_ms.append(ctx.last_timing())
dt = (time.perf_counter() - t0) / N
a = np.array(gpu_ms)
print("K=%d x %d samples at %g Msps: %.1f us per step wall (%.1f GS/s); device: whole call %.1f us, dominant kernel %.1f us; packets per stream %s"
      % (K, n, rate / 1e6, dt * 1e6, K * n / dt / 1e9, a[:, 0].mean() * 1e3, a[:, 1].mean() * 1e3, [len(g) for g in got]))
# the hand-over alone: what a host does with the accepted packets of one scan (copy out of the pinned array + the pass into streams)
allp = np.concatenate(got)
off = np.cumsum([0] + [len(g) for g in got])
t1 = time.perf_counter()
for _ in range(200):
    parts = [allp[off[j]:off[j + 1]].copy() for j in range(K)]
print("splitting %d packets into %d arrays on the host (numpy slices): %.1f us" % (len(allp), K, (time.perf_counter() - t1) / 200 * 1e6))
ctx.close()
# scans in flight behind one handle: where does the host's time go?
for depth in (2, 3, 4):
    pipe = _capi.Pipe(rate, 7.0, True, device=0, depth=depth, lib=lib)
    ts, tc = [], []
    def fly(count):
        for k in range(count):
            if pipe.in_flight() == pipe.depth():
                a0 = time.perf_counter(); pipe.collect_multi(); tc.append(time.perf_counter() - a0)
            a0 = time.perf_counter(); pipe.submit_multi_device(d.data_ptr(), lens); ts.append(time.perf_counter() - a0)
        while pipe.in_flight():
            a0 = time.perf_counter(); pipe.collect_multi(); tc.append(time.perf_counter() - a0)
    fly(6)
    ts.clear(); tc.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fly(24)
    dtp = (time.perf_counter() - t0) / 24
    print("pipe depth %d: %.1f us per scan (%.1f GS/s); host: submit %.1f us (max %.1f), collect %.1f us (max %.1f)"
          % (depth, dtp * 1e6, K * n / dtp / 1e9, np.mean(ts) * 1e6, np.max(ts) * 1e6, np.mean(tc) * 1e6, np.max(tc) * 1e6))
    pipe.close()
