"""Throughput of am_pipe against its depth (batches in flight from one host thread), 64 Msps stress workload."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", sys.argv[1] if len(sys.argv) > 1 else "8")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("gr-air-modes_amd", "tools"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import synth
from air_modes import _capi

n = 64000000
batches = [torch.from_numpy(np.asarray(synth.synth_capture(64e6, n, 20000.0, seed=6400 + 100 * k)[0], dtype=np.complex64).view(np.float32)).cuda()
           for k in range(3)]
torch.cuda.synchronize()
for depth in (1, 2, 3, 4, 5, 6, 8):
    pipe = _capi.Pipe(64e6, 7.0, True, device=0, depth=depth)

    def run(count):
        got = 0
        for k in range(count):
            if pipe.in_flight() == pipe.depth():
                got += len(pipe.collect())
            pipe.submit_device(batches[k % 3].data_ptr(), n)
        while pipe.in_flight():
            got += len(pipe.collect())
        return got
    run(3 * depth + 3)
    torch.cuda.synchronize()
    best = 0.0
    for rep in range(3):
        t0 = time.perf_counter()
        run(24)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = max(best, 24 * n / dt)
    print("depth %d: %.1f GS/s (%.3f ms per batch)" % (depth, best / 1e9, n / best * 1e3), flush=True)
    pipe.close()
