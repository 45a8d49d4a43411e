"""Would ONE synchronous step be faster as several chunks of the same stream in flight (am_spipe inside the call)?
gpurun -- 'python tools/gpu_split_step.py'   -- 64 M samples (one bench step) as 1 / 2 / 3 / 4 contiguous chunks through a stream pipe,
timed from the first submit to the last collect, against am_process_iq on the whole batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gr-air-modes_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np
import torch
import synth
from air_modes import _capi

rate, n = 64e6, 64000000
bufs = []
for b in range(3):
    iq = synth.synth_capture(rate, n, 20000.0, 6400 + b)[0]
    bufs.append(torch.from_numpy(np.asarray(iq, np.complex64).view(np.float32)).cuda())
ctx = _capi.Context(rate, 7.0, True, device=0)
for k in range(6):
    want = ctx.process_iq_device(bufs[k % 3].data_ptr(), n, flush=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(30):
    want = ctx.process_iq_device(bufs[k % 3].data_ptr(), n, flush=True)
torch.cuda.synchronize()
print("am_process_iq, whole batch: %.1f us per step, %d packets" % ((time.perf_counter() - t0) / 30 * 1e6, len(want)))
for parts in (1, 2, 3, 4):
    pipe = _capi.StreamPipe(rate, 7.0, True, device=0, depth=max(parts, 2))
    edges = [int(round(n * i / parts / 3072.0)) * 3072 for i in range(parts)] + [n]

    def step(buf):
        out = []
        for i in range(parts):
            pipe.submit_device(buf.data_ptr() + 8 * edges[i], edges[i + 1] - edges[i], flush=(i == parts - 1))
        while pipe.in_flight():
            out.append(pipe.collect())
        return np.concatenate(out)
    for k in range(6):
        got = step(bufs[k % 3])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(30):
        got = step(bufs[k % 3])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    same = got.tobytes() == want.tobytes()
    print("stream pipe, %d chunk(s) in flight: %.1f us per 64 M samples, same packets as the whole batch: %s, redone %d"
          % (parts, dt * 1e6, same, pipe.redone()))
    pipe.close()
