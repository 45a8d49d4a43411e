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
