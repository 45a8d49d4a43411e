"""Where does a bench step's wall-clock go?  call time vs GPU span vs Python loop (tuning aid)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("gr-air-modes_amd", "tools"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import synth
from air_modes import _capi

rate, n, lam, seed = synth.CONFIGS["64msps"][:4] if isinstance(synth.CONFIGS["64msps"], (tuple, list)) else (64e6, 64000000, 20000.0, 6400)
iq, _ = synth.synth_capture(64e6, 64000000, 20000.0, seed=6400)
d = torch.from_numpy(np.asarray(iq, dtype=np.complex64).view(np.float32)).cuda()
ctx = _capi.Context(64e6, 7.0, True)
for _ in range(3):
    ctx.process_iq_device(d.data_ptr(), 64000000, flush=True)
torch.cuda.synchronize()
calls, spans = [], []
t_loop0 = time.perf_counter()
for _ in range(20):
    t0 = time.perf_counter()
    pk = ctx.process_iq_device(d.data_ptr(), 64000000, flush=True)
    t1 = time.perf_counter()
    calls.append((t1 - t0) * 1e3)
    spans.append(ctx.last_timing()[0])
t_loop = (time.perf_counter() - t_loop0) * 1e3 / 20
print("per step: loop %.3f ms, call %.3f ms, GPU span (first event .. last event) %.3f ms, candidates %d, packets %d"
      % (t_loop, np.mean(calls), np.mean(spans), ctx.last_num_candidates(), len(pk)))
ctx.close()

# the same through am_pipe (three batches in flight, one host thread): what submit and collect cost the host
pipe = _capi.Pipe(64e6, 7.0, True, device=0, depth=3)
sub, col = [], []
for k in range(30):
    if pipe.in_flight() == pipe.depth():
        t0 = time.perf_counter(); pipe.collect(); col.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter(); pipe.submit_device(d.data_ptr(), 64000000); sub.append((time.perf_counter() - t0) * 1e3)
while pipe.in_flight():
    pipe.collect()
print("am_pipe depth 3: submit %.3f ms (min %.3f), collect %.3f ms (min %.3f; includes waiting for the GPU)"
      % (np.mean(sub[6:]), np.min(sub[6:]), np.mean(col[6:]), np.min(col[6:])))
# submit alone, nothing to wait for: enqueue one batch on an idle pipe, time the call, drain
idle = []
for k in range(8):
    t0 = time.perf_counter(); pipe.submit_device(d.data_ptr(), 64000000); idle.append((time.perf_counter() - t0) * 1e3)
    pipe.collect()
print("submit on an idle pipe: %.3f ms (min %.3f)" % (np.mean(idle[2:]), np.min(idle[2:])))
pipe.close()
