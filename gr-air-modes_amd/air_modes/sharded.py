"""Time-sharded operation over several GPUs of one node (BASELINE.json configs[3]).

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  A stream of world*n samples is cut into `world` contiguous time
chunks.  Per step:

  1. halo exchange   every rank publishes [its last `left` samples | its first `right` samples]
                     in ONE all_gather of fixed-size slabs (KB-scale: latency bound, far below
                     the 153 GB/s per xGMI link) and keeps its two neighbours' slabs;
  2. local scan      am_shard_scan: front end + detection + refinement of the chunk's own
                     positions -> candidate records (16 bytes each);
  3. record exchange all_gather of the record counts, then of the padded record arrays;
  4. resolve         am_shard_resolve on the concatenated list: the greedy chain of the
                     reference's sequential scan (lib/preamble_impl.cc:172,209,237) is resolved
                     identically on every rank; each rank extracts + slices the hits whose
                     first-stage position lies in its chunk.

Packets of all ranks, concatenated in rank order, equal the single-GPU (and the reference's)
packet list for the whole stream.
"""
import ctypes as C

import numpy as np

from . import _capi


class ShardedReceiver(object):
    def __init__(self, ctx, rank, world, n_per_rank, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.ctx, self.rank, self.world, self.n = ctx, int(rank), int(world), int(n_per_rank)
        self.group = group
        self.total = self.world * self.n
        self.left, self.right = ctx.shard_halo()
        if self.world > 1 and self.n < max(self.left, self.right):
            raise ValueError("chunk shorter than the halo (%d samples)" % max(self.left, self.right))
        self.a0, self.a1 = self.rank * self.n, (self.rank + 1) * self.n
        self._buf = None

    def _alloc(self, like):
        t = self.torch
        hl, hr, n = self.left, self.right, self.n
        dev = like.device
        self._buf = t.zeros((hl + n + hr) * 2, dtype=t.float32, device=dev)
        self._slab = t.empty((hl + hr) * 2, dtype=t.float32, device=dev)
        self._slabs = [t.empty_like(self._slab) for _ in range(self.world)]
        self._cap = max(4096, n // 8)
        self._recs = t.zeros(self._cap * 2, dtype=t.int64, device=dev)        # am_cand = 16 bytes
        self._cnt = t.zeros(1, dtype=t.int64, device=dev)
        self._cnts = [t.zeros(1, dtype=t.int64, device=dev) for _ in range(self.world)]

    def step(self, own):
        """own: this rank's chunk, interleaved float32 I,Q (2*n values), torch tensor on the
        GPU (or on the CPU for the gloo tests).  Returns this rank's accepted packets."""
        t, dist = self.torch, self.dist
        hl, hr, n, world, rank = self.left, self.right, self.n, self.world, self.rank
        assert own.numel() == 2 * n and own.dtype == t.float32
        if self._buf is None:
            self._alloc(own)
        buf = self._buf
        on_gpu = own.is_cuda
        buf[hl * 2:(hl + n) * 2] = own
        if world > 1:
            self._slab[:hl * 2] = own[(n - hl) * 2:]
            self._slab[hl * 2:] = own[:hr * 2]
            dist.all_gather(self._slabs, self._slab, group=self.group)
            if rank > 0:
                buf[:hl * 2] = self._slabs[rank - 1][:hl * 2]
            if rank < world - 1:
                buf[(hl + n) * 2:] = self._slabs[rank + 1][hl * 2:]
        if on_gpu:
            t.cuda.synchronize()
        lo = max(0, self.a0 - hl)
        off = (hl - (self.a0 - lo)) * 2                      # floats to skip at the stream start
        flags_in = _capi.AM_F_DEVICE_IN if on_gpu else 0
        L = self.ctx.lib.L
        while True:
            got = C.c_uint64(0)
            rc = L.am_shard_scan(self.ctx._h, buf.data_ptr() + off * 4, self.a0, self.a1, self.total,
                                 flags_in | (_capi.AM_F_DEVICE_OUT if on_gpu else 0),
                                 self._recs.data_ptr(), self._cap, C.byref(got))
            if rc == _capi.AM_ECAPACITY:
                self._cap = int(got.value) + 1024
                self._recs = t.zeros(self._cap * 2, dtype=t.int64, device=own.device)
                continue
            self.ctx._chk(rc)
            break
        m = int(got.value)
        if world > 1:
            self._cnt[0] = m
            dist.all_gather(self._cnts, self._cnt, group=self.group)
            ms = [int(c.item()) for c in self._cnts]
            mmax = max(max(ms), 1)
            if mmax > self._cap:
                grown = t.zeros(mmax * 2, dtype=t.int64, device=own.device)
                grown[:self._cap * 2] = self._recs
                self._recs, self._cap = grown, mmax
            parts = [t.empty(mmax * 2, dtype=t.int64, device=own.device) for _ in range(world)]
            dist.all_gather(parts, self._recs[:mmax * 2].contiguous(), group=self.group)
            allr = t.cat([parts[r][:ms[r] * 2] for r in range(world)]).contiguous()
            mall = sum(ms)
        else:
            allr, mall = self._recs, m
        if on_gpu:
            t.cuda.synchronize()
        cap = max(64, n // 2000 + 64)
        while True:
            out = np.zeros(cap, _capi.PACKET_DTYPE)
            g2 = C.c_uint64(0)
            rc = L.am_shard_resolve(self.ctx._h, allr.data_ptr() if mall else None, mall, flags_in,
                                    out.ctypes.data, cap, C.byref(g2))
            if rc == _capi.AM_ECAPACITY:
                return self.ctx._fetch(int(g2.value))
            self.ctx._chk(rc)
            return out[:g2.value]
