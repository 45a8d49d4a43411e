"""Time-sharded operation over several GPUs of one node (BASELINE.json configs[3]).

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  A stream of world*n samples is cut into `world` contiguous time
chunks.  The reference's preamble scan is sequential, but the only state that crosses a chunk
boundary is the position at which the scan resumes, and that can reach at most 241*spc
samples into the next chunk.  Per step:

  1. halo exchange   every rank sends its first `right` samples to the rank before it and its last `left`
                     samples to the rank after it (two point-to-point pairs over xGMI, KB-scale: latency
                     bound, far below the 153 GB/s of a link), received straight into the halo regions of
                     the chunk buffer; the context's own stream then waits ON THE DEVICE for the stream the
                     receives are ordered on (an event: am_wait_for_stream), so no host synchronisation
                     separates the exchange from the scan, whatever PyTorch's current stream is at step();
  2. local scan      am_shard_scan_async: front end, detection, refinement, the successor array and block exits of
                     the chunk's own greedy chain, plus an EXIT TABLE: for every candidate the scan could enter the
                     chunk at (those in its first 241*spc samples), where the scan would leave the chunk.  All of it
                     is only enqueued; the table lands in a device message;
  3. table exchange  one all_gather of the fixed-size messages (count + 512 entries of 16 bytes), device to device,
                     ordered behind the scan and in front of the next step by events (am_signal_stream /
                     am_wait_for_stream): no host copy;
  4. resolve         am_shard_resolve_async: the entry position of the own chunk is composed from all tables by a
                     kernel, the chain is marked from there, hits are extracted and sliced.  ONE completion wait
                     per step (round 2: two waits and a host round trip for the tables between them).
  A step whose table does not fit the message, or whose scan met more candidates than the capacity it was launched for
  (both are flagged in the message header, so every rank takes the same decision without another collective), is repeated
  on the synchronous path of round 2 (am_shard_scan ->
  host tables -> am_shard_entry -> am_shard_resolve; `sync_steps` counts them; the first step of a receiver, which has
  no candidate-density estimate yet, reads one count back).

Packets of all ranks, concatenated in rank order, equal the single-GPU (and the reference's)
packet list for the whole stream; the per-rank work does not grow with the number of ranks.
"rx_time" tags (ctx.set_rx_time) carry stream-absolute offsets: give every rank's context the same tags.
"""
import ctypes as C

import numpy as np

from . import _capi


class ShardedReceiver(object):
    """`chunk` is this rank's 2*n float32 I,Q samples (a view into the halo'd device buffer:
    write the samples there once, no per-step copy); step() runs one pass."""

    def __init__(self, ctx, rank, world, n_per_rank, group=None, device=None, small_table=512, host_free=True):
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
        spc = max(int(ctx.get_rate() / 2e6), 1)
        self.tab_cap = 241 * spc + 4                      # a lead-in cannot hold more candidates than positions
        # the tables are exchanged in a short fixed-size message; only when some rank's table does not fit
        # (every rank sees every count) the full-size message follows
        self.small_cap = max(1, min(int(small_table), self.tab_cap))
        self.full_exchanges = 0                           # steps that needed the full-size message
        self.host_free = bool(host_free)                  # the device-side table exchange (False: round 2's synchronous step)
        self.sync_steps = 0                               # steps the host-free path had to repeat synchronously
        self._alloc(device if device is not None else "cpu")
        self.chunk = self._buf[self.left * 2:(self.left + self.n) * 2]

    def _alloc(self, dev):
        t = self.torch
        hl, hr, n = self.left, self.right, self.n
        self._buf = t.zeros((hl + n + hr) * 2, dtype=t.float32, device=dev)
        # exit table message: [count, pos0, exit0, pos1, exit1, ...] as int64, in two sizes
        self._msg = t.zeros(1 + 2 * self.tab_cap, dtype=t.int64, device=dev)
        self._msgs = [t.empty_like(self._msg) for _ in range(self.world)]
        self._msg_s = t.zeros(1 + 2 * self.small_cap, dtype=t.int64, device=dev)
        self._msgs_s = [t.empty_like(self._msg_s) for _ in range(self.world)]
        self._host_tab = np.zeros(self.tab_cap, _capi.EXIT_DTYPE)
        # host-free step: this rank's message {count, -} + small_cap entries of (pos, exit), and everybody's
        self._amsg = t.zeros(2 * (1 + self.small_cap), dtype=t.int64, device=dev)
        self._agath = t.zeros(self.world * 2 * (1 + self.small_cap), dtype=t.int64, device=dev)
        self._agath_list = list(self._agath.chunk(self.world))

    def _exchange(self, m, cap, msg_dev, msgs_dev):
        """all_gather of [count | first min(count, cap) table entries]; returns the gathered rows (host)."""
        k = min(m, cap)
        msg = np.zeros(1 + 2 * cap, np.int64)
        msg[0] = m
        msg[1:1 + 2 * k] = self._host_tab[:k].view(np.int64)
        msg_dev.copy_(self.torch.from_numpy(msg))
        self.dist.all_gather(msgs_dev, msg_dev, group=self.group)
        return self.torch.stack(msgs_dev).cpu().numpy()

    def step(self):
        """One pass over the samples currently in `chunk`.  Returns this rank's accepted packets."""
        t, dist = self.torch, self.dist
        hl, hr, n, world, rank = self.left, self.right, self.n, self.world, self.rank
        buf = self._buf
        own = self.chunk
        on_gpu = buf.is_cuda
        if world > 1:
            ops = []
            if rank > 0:
                ops.append(dist.P2POp(dist.isend, own[:hr * 2], rank - 1, self.group))
                ops.append(dist.P2POp(dist.irecv, buf[:hl * 2], rank - 1, self.group))
            if rank < world - 1:
                ops.append(dist.P2POp(dist.isend, own[(n - hl) * 2:], rank + 1, self.group))
                ops.append(dist.P2POp(dist.irecv, buf[(hl + n) * 2:], rank + 1, self.group))
            for req in dist.batch_isend_irecv(ops):
                req.wait()              # (RCCL: orders the current stream behind the transfer, the host does not block)
            if on_gpu:
                # ... and the scan behind the current stream: the context keeps its own stream, whatever stream is
                # current when step() is called (a stream captured once at construction raced)
                self.ctx.wait_for_stream(t.cuda.current_stream(buf.device).cuda_stream)
        lo = max(0, self.a0 - hl)
        off = (hl - (self.a0 - lo)) * 2                      # floats to skip at the stream start
        ptr = buf.data_ptr() + off * 4
        cap_pk = max(64, n // 2000 + 64)
        if self.host_free:
            cur = t.cuda.current_stream(buf.device).cuda_stream if on_gpu else 0
            self.ctx.shard_scan_async(ptr, self.a0, self.a1, self.total, self._amsg.data_ptr(), self.small_cap,
                                      device_in=on_gpu)
            if world > 1:
                if on_gpu:
                    self.ctx.signal_stream(cur)              # the collective waits (on the device) for the table
                dist.all_gather(self._agath_list, self._amsg, group=self.group)
                if on_gpu:
                    self.ctx.wait_for_stream(cur)            # ... and the resolve step for the collective
                msgs = self._agath
            else:
                msgs = self._amsg
            # `redo` is the same on every rank without another collective: a table that does not fit its message and a scan
            # that met more candidates than the capacity it was launched for are both flagged in the message header, and
            # every rank reads every header
            pk, redo = self.ctx.shard_resolve_async(msgs.data_ptr(), world, rank, self.small_cap, capacity=cap_pk)
            if not redo:
                return pk
            self.sync_steps += 1
        L = self.ctx.lib.L
        got = C.c_uint64(0)
        rc = L.am_shard_scan(self.ctx._h, ptr, self.a0, self.a1, self.total,
                             _capi.AM_F_DEVICE_IN if on_gpu else 0, self._host_tab.ctypes.data, self.tab_cap,
                             C.byref(got))
        self.ctx._chk(rc)
        m = int(got.value)
        if world > 1:
            allm = self._exchange(m, self.small_cap, self._msg_s, self._msgs_s)
            if int(allm[:, 0].max()) > self.small_cap:       # the same decision on every rank
                self.full_exchanges += 1
                allm = self._exchange(m, self.tab_cap, self._msg, self._msgs)
            tables = [allm[r, 1:1 + 2 * int(allm[r, 0])].copy().view(_capi.EXIT_DTYPE) for r in range(world)]
            entry = _capi.shard_entries(self.ctx.lib, tables, [r * n for r in range(world)])
            cur_in = int(entry[rank])
        else:
            cur_in = 0
        return self.ctx.shard_resolve(cur_in, capacity=cap_pk)
