"""Time-sharded operation over several GPUs of one node (BASELINE.json configs[3]) -- a RECEIVER, not a batch.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
The stream arrives in steps of world*n samples; step k hands rank r the samples [k*W*n + r*n, k*W*n + (r+1)*n).
The reference's preamble scan is sequential (lib/preamble_impl.cc:213,237,244: it resumes where the last general_work
left off, for ever), but the only state that crosses a chunk boundary -- between ranks and between steps alike -- is
the position at which the scan resumes, and that can reach at most 241*spc samples into the next chunk.

Who decides what.  A decision about position p needs the samples up to p + H, H = 244*spc (late shift + the 240-chip
burst).  So rank r DECIDES the positions [k*W*n + r*n - H, k*W*n + (r+1)*n - H): everything its own samples let it
decide.  It needs `left + H` samples in front of its own (reference-level history + the positions it takes over from
its predecessor) -- the tail of the rank before it, or, for rank 0, the tail the LAST rank kept from the step before --
and nothing from the rank after it.  The positions the last rank cannot decide yet are rank 0's first in the next
step; `flush=True` ends the stream: the last rank then decides up to the end under the reference's end-of-buffer rule
(lib/preamble_impl.cc:150,212) and the receiver starts over at sample 0.

Per step:
  1. tail exchange   every rank sends its last left + H samples to the rank after it (the last rank: what it kept from
                     the step before, to rank 0) -- point-to-point over xGMI, KB-scale: latency bound, far below the
                     153 GB/s of a link --, received straight into the halo in front of the chunk; the context's own
                     stream then waits ON THE DEVICE for the stream the receive is ordered on (am_wait_for_stream);
  2. local scan      am_shard_scan_async: front end, detection, refinement, the successor array and block exits of
                     the chunk's own greedy chain, plus an EXIT TABLE: for every candidate the scan could enter the
                     chunk at (those in its first 241*spc positions), where the scan would leave the chunk.  All of it
                     is only enqueued; the table lands in a device message whose header also carries where the scan
                     left this rank's chunk in the step BEFORE;
  3. table exchange  one all_gather of the fixed-size messages (2 header entries + 512 of 16 bytes), device to device,
                     ordered behind the scan and in front of the next step by events (am_signal_stream /
                     am_wait_for_stream): no host copy;
  4. resolve         am_shard_resolve_async: starting from where the scan left the LAST rank's chunk a step ago (that
                     rank's header), the entry position of the own chunk is composed from all tables by a kernel, the
                     chain is marked from there, hits are extracted and sliced, and where the scan leaves the own
                     chunk stays in a device word for the next step's header.  ONE completion wait per step.
  A step whose table does not fit the message, or whose scan met more candidates than the capacity it was launched
  for (both are flagged in the message header, so every rank takes the same decision without another collective), is
  repeated on the synchronous path (am_shard_scan -> host tables -> am_shard_entry2 -> am_shard_resolve; `sync_steps`
  counts them).

Packets of all ranks and steps, concatenated in (step, rank) order, equal the single-GPU (and the reference's) packet
list for the whole stream, with item counts and time stamps that keep counting; the per-rank work does not grow with
the number of ranks.  "rx_time" tags (ctx.set_rx_time) carry stream-absolute offsets: give every rank's context the
same tags.
"""
import ctypes as C
import time

import numpy as np

from . import _capi


class ShardedReceiver(object):
    """`chunk` is this rank's 2*n float32 I,Q samples of the current step (a view into the halo'd device buffer:
    write each step's samples there, no copy inside step()); step() runs one pass and returns this rank's packets."""

    def __init__(self, ctx, rank, world, n_per_rank, group=None, device=None, small_table=512, host_free=True,
                 force_collectives=False, share_stream=True):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.ctx, self.rank, self.world, self.n = ctx, int(rank), int(world), int(n_per_rank)
        self.group = group
        # force_collectives: a ONE-rank receiver also goes through the process group -- the tail travels by a send / receive to
        # itself (or, where the backend refuses that, an all_gather of the tail), the exit table by all_gather_into_tensor, ordered
        # by the same events as at world 8.  What it is for: RCCL executing this code on a box with one GPU.
        self.force = bool(force_collectives) and int(world) == 1 and dist.is_available() and dist.is_initialized()
        self.tail_by_gather = False                       # the backend refused a send to oneself: the tail goes by all_gather
        self.left, self.hold = ctx.shard_halo()           # history in front of a position / look-ahead behind it
        self.halo = self.left + self.hold                 # samples wanted in front of the own ones
        self.right = self.hold                            # (name kept: the look-ahead a chunk needs behind its last position)
        if self.n < self.halo:
            raise ValueError("chunk shorter than the halo (%d samples)" % self.halo)
        spc_hi = max(int(-(-ctx.get_rate() // 2e6)), 1)  # samples per chip, rounded up (a rate need not be a multiple of 2 MHz)
        self.tab_cap = 241 * spc_hi + 4                   # a lead-in cannot hold more candidates than positions
        # the tables are exchanged in a short fixed-size message; only when some rank's table does not fit
        # (every rank sees every count) the full-size message follows
        self.small_cap = max(1, min(int(small_table), self.tab_cap))
        self.full_exchanges = 0                           # steps that needed the full-size message
        self.sync_steps = 0                               # steps the host-free path had to repeat synchronously
        self.k = 0                                        # steps of the current stream so far
        # host time spent inside the torch.distributed calls of step(), summed over steps (microseconds; bench.py reports
        # them per step beside the no-collective floor of a one-rank receiver)
        self.host_us = {"tail_exchange": 0.0, "all_gather": 0.0, "steps": 0}
        self.host_us_steps = {"tail_exchange": [], "all_gather": []}
        self._wrap_posted = False                         # the ring-closing transfer of the next step is already under way     # ... and per step (the first call of a backend sets its communicator up)
        self._alloc(device if device is not None else "cpu")
        # the device-side exchange hands device pointers to kernels: only where the buffers live on the GPU (or where
        # "device memory" is host memory: the CPU emulation the tests run on)
        emulated = bool(getattr(ctx.lib, "emulated", False))   # (am_is_emulated(): asked of the library, not guessed from its path)
        self.host_free = bool(host_free) and (self._buf.is_cuda or emulated)
        self.chunk = self._buf[self.halo * 2:(self.halo + self.n) * 2]
        # the rank whose tail the next STEP needs (the last one) keeps it inside the resolve call: behind the slicing, which
        # still reads the samples, in front of the completion -- done when step() returns, before `chunk` is overwritten
        self._keeps = self.rank == self.world - 1
        if self._keeps:
            ctx.shard_keep_tail(self._tail.data_ptr(), self._own_tail.data_ptr(), self.halo * 8)
        else:
            ctx.shard_keep_tail(0, 0, 0)
        ctx._keep_owner = id(self)                        # (a context serves one receiver at a time: the newest)
        # share_stream (round 6): with collectives in the step, the context and torch.distributed work on ONE stream of the receiver's
        # own -- the collectives are issued with it current -- so that the only stream hops left are the backend's own (its stream
        # waits for ours, ours for its).  Measured at world 1 under RCCL (profiles/r6_rccl): every hop between the context's stream
        # and the one the collective was issued from cost 20-60 us, four of them per step.
        self._tstream = None
        if bool(share_stream) and self._buf.is_cuda and (self.world > 1 or self.force):
            self._tstream = torch.cuda.Stream(device=self._buf.device)
            ctx.set_stream(self._tstream.cuda_stream)

    def close(self):
        """The context forgets this receiver's buffers (before the receiver goes away while the context lives on)."""
        ctx = getattr(self, "ctx", None)
        if ctx is not None and getattr(ctx, "_h", None) and getattr(ctx, "_keep_owner", None) == id(self):
            ctx.shard_keep_tail(0, 0, 0)
            ctx._keep_owner = None
            if getattr(self, "_tstream", None) is not None:
                ctx.set_stream(None)                      # (the context's own stream again)
                self._tstream = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _alloc(self, dev):
        t = self.torch
        self._buf = t.zeros((self.halo + self.n) * 2, dtype=t.float32, device=dev)
        self._tail = t.zeros(self.halo * 2, dtype=t.float32, device=dev)     # the last rank's tail of the step before
        # views used every step (slicing a tensor costs microseconds of host time each)
        self._halo_view = self._buf[:self.halo * 2]
        self._own_tail = self._buf[self.n * 2:]                                # the own samples' last `halo`
        # synchronous path: [count, exit of the step before, pos0, exit0, pos1, exit1, ...] as int64, in two sizes
        self._msg = t.zeros(2 + 2 * self.tab_cap, dtype=t.int64, device=dev)
        self._msgs = [t.empty_like(self._msg) for _ in range(self.world)]
        self._msg_s = t.zeros(2 + 2 * self.small_cap, dtype=t.int64, device=dev)
        self._msgs_s = [t.empty_like(self._msg_s) for _ in range(self.world)]
        self._host_tab = np.zeros(self.tab_cap, _capi.EXIT_DTYPE)
        # host-free step: this rank's message (header + small_cap entries of (pos, exit)), and everybody's
        words = 2 * (_capi.SHARD_MSG_HEADER + self.small_cap)
        self._amsg = t.zeros(words, dtype=t.int64, device=dev)
        self._agath = t.zeros(self.world * words, dtype=t.int64, device=dev)   # (all_gather_into_tensor: no list of views, no copies)

    def reset(self):
        """Start a new stream at sample 0 (what step(flush=True) does at its end)."""
        self.ctx.reset()
        self.k = 0
        self._wrap_posted = False

    def _exchange_p2p(self, ops):
        t0 = time.perf_counter()
        try:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()              # (RCCL: orders the current stream behind the transfer, the host does not block)
        except Exception:
            if not self.force:
                raise
            self.tail_by_gather = True                   # (one rank, a backend without send-to-self: the all_gather fallback)
        dt = (time.perf_counter() - t0) * 1e6
        self.host_us["tail_exchange"] += dt
        self.host_us_steps["tail_exchange"].append(dt)

    def _post_wrap(self):
        """The transfer that closes the ring, for the NEXT step: the last rank's kept tail to rank 0's halo.  Posted right behind this
        step's resolve (the kept tail is complete: the resolve step's completion was waited for), so that it runs while the host
        hands the packets out and the caller fills the next chunk -- at world 1 that is the whole tail exchange."""
        if not (self.world > 1 or (self.force and not self.tail_by_gather)):
            return
        ops = []
        if self.rank == self.world - 1:
            ops.append(self.dist.P2POp(self.dist.isend, self._tail, 0, self.group))
        if self.rank == 0:
            ops.append(self.dist.P2POp(self.dist.irecv, self._halo_view, self.world - 1, self.group))
        if ops:
            self._exchange_p2p(ops)
            self._wrap_posted = not self.tail_by_gather

    def _exchange(self, m, last_exit, cap, msg_dev, msgs_dev):
        """all_gather of [count | exit of the step before | first min(count, cap) table entries]; returns the rows (host)."""
        k = min(m, cap)
        msg = np.zeros(2 + 2 * cap, np.int64)
        msg[0] = m
        msg[1] = np.uint64(last_exit).astype(np.int64)
        msg[2:2 + 2 * k] = self._host_tab[:k].view(np.int64)
        msg_dev.copy_(self.torch.from_numpy(msg))
        self.dist.all_gather(msgs_dev, msg_dev, group=self.group)
        return self.torch.stack(msgs_dev).cpu().numpy()

    def step(self, flush=False):
        """One pass over the samples currently in `chunk`: the next world*n samples of the stream.  flush: they are the
        stream's last (every rank must say so).  Returns this rank's accepted packets."""
        if self._tstream is not None:
            # the receiver's stream: behind whoever filled `chunk` on the caller's current stream, then current for the whole step
            self._tstream.wait_stream(self.torch.cuda.current_stream(self._buf.device))
            with self.torch.cuda.stream(self._tstream):
                return self._step(flush)
        return self._step(flush)

    def _step(self, flush):
        t, dist = self.torch, self.dist
        n, world, rank, halo, H = self.n, self.world, self.rank, self.halo, self.hold
        buf, own = self._buf, self.chunk
        on_gpu = buf.is_cuda
        S0 = self.k * world * n                              # absolute index of the step's first sample
        last = rank == world - 1
        # 1. the samples in front of the own ones.  The transfer that closes the ring -- the last rank's tail of the step BEFORE, to
        # rank 0 -- was posted at the end of that step (round 6: _post_wrap), off this step's critical path; what is left here are the
        # transfers inside the step, from every rank to the next
        if world > 1 or (self.force and not self.tail_by_gather):
            ops = []
            if not last:
                ops.append(dist.P2POp(dist.isend, self._own_tail, rank + 1, self.group))
            elif self.k > 0 and not self._wrap_posted:
                ops.append(dist.P2POp(dist.isend, self._tail, 0, self.group))
            if rank > 0:
                ops.append(dist.P2POp(dist.irecv, self._halo_view, rank - 1, self.group))
            elif self.k > 0 and not self._wrap_posted:
                ops.append(dist.P2POp(dist.irecv, self._halo_view, world - 1, self.group))
            self._wrap_posted = False
            if ops:
                self._exchange_p2p(ops)
        if self.force and self.tail_by_gather and self.k > 0:
            tc = time.perf_counter()
            dist.all_gather_into_tensor(self._halo_view, self._tail, group=self.group)     # world 1: the gathered tensor IS the tail
            self.host_us["tail_exchange"] += (time.perf_counter() - tc) * 1e6
        elif world == 1 and not self.force and self.k > 0:
            # one rank: it is its own predecessor (the kept tail goes in front of the chunk on the context's own stream)
            self.ctx.stream_copy(self._halo_view.data_ptr(), self._tail.data_ptr(), halo * 8)
        cur = t.cuda.current_stream(buf.device).cuda_stream if on_gpu else 0
        shared = self._tstream is not None               # (the context enqueues on the current stream itself: nothing to order)
        if on_gpu and not shared and (cur != 0 or world > 1 or self.force):
            # the scan behind the current stream (whoever filled `chunk`, the receive above): the context keeps its own
            # stream, whatever stream is current when step() is called.  (The legacy default stream needs no event: the
            # context's stream is a blocking one, which the runtime orders behind it -- and an event across two idle
            # hardware queues costs tens of microseconds per step.)
            self.ctx.wait_for_stream(cur)
        # 2. the positions this rank decides, and the samples it has for that
        first_chunk = self.k == 0 and rank == 0
        a0 = 0 if first_chunk else S0 + rank * n - H
        total = S0 + world * n                               # samples so far (flush: the length of the stream)
        a1 = total if (flush and last) else S0 + (rank + 1) * n - H
        b0 = S0 + rank * n - halo                            # absolute index of buf[0]
        lo = max(0, a0 - self.left)                          # first sample the library wants
        ptr = buf.data_ptr() + (lo - b0) * 8
        more = not flush
        cap_pk = max(64, n // 2000 + 64)
        pk = None
        if self.host_free:
            self.ctx.shard_scan_async(ptr, a0, a1, total, self._amsg.data_ptr(), self.small_cap, device_in=on_gpu, more=more)
            if world > 1 or self.force:
                if on_gpu and not shared:
                    self.ctx.signal_stream(cur)              # the collective waits (on the device) for the table
                tc = time.perf_counter()
                dist.all_gather_into_tensor(self._agath, self._amsg, group=self.group)
                self.host_us["all_gather"] += (time.perf_counter() - tc) * 1e6
                self.host_us_steps["all_gather"].append((time.perf_counter() - tc) * 1e6)
                if on_gpu and not shared:
                    self.ctx.wait_for_stream(cur)            # ... and the resolve step for the collective
                msgs = self._agath
            else:
                msgs = self._amsg
            # `redo` is the same on every rank without another collective: a table that does not fit its message and a scan
            # that met more candidates than the capacity it was launched for are both flagged in the message header, and
            # every rank reads every header
            pk, redo = self.ctx.shard_resolve_async(msgs.data_ptr(), world, rank, self.small_cap, capacity=cap_pk)
            if redo:
                self.sync_steps += 1
                pk = None
        if pk is None:
            pk = self._step_sync(ptr, a0, a1, total, more, cap_pk, on_gpu)
        self.host_us["steps"] += 1
        # 3. what the next step needs from this one
        if flush:
            self.reset()
        else:
            self.k += 1                                      # (the last rank's tail was kept inside the resolve call)
            self._post_wrap()
        return pk

    def _step_sync(self, ptr, a0, a1, total, more, cap_pk, on_gpu):
        """The step with the tables on the host: am_shard_scan -> all_gather of the tables -> am_shard_entry2 ->
        am_shard_resolve.  The scan position of the step before comes from the context (the host-free path keeps it on
        the device) and goes back there."""
        world, rank = self.world, self.rank
        L = self.ctx.lib.L
        got = C.c_uint64(0)
        flags = (_capi.AM_F_DEVICE_IN if on_gpu else 0) | (_capi.AM_F_MORE if more else 0)
        rc = L.am_shard_scan(self.ctx._h, ptr, a0, a1, total, flags, self._host_tab.ctypes.data, self.tab_cap, C.byref(got))
        self.ctx._chk(rc)
        m = int(got.value)
        last_exit = self.ctx.shard_get_exit()
        if world > 1:
            allm = self._exchange(m, last_exit, self.small_cap, self._msg_s, self._msgs_s)
            if int(allm[:, 0].max()) > self.small_cap:       # the same decision on every rank
                self.full_exchanges += 1
                allm = self._exchange(m, last_exit, self.tab_cap, self._msg, self._msgs)
            tables = [allm[r, 2:2 + 2 * int(allm[r, 0])].copy().view(_capi.EXIT_DTYPE) for r in range(world)]
            cur_in = int(np.int64(allm[world - 1, 1]).astype(np.uint64))
        else:
            tables = [self._host_tab[:m].copy()]
            cur_in = last_exit
        entry, leave = _capi.shard_entries(self.ctx.lib, tables, cur_in=cur_in, with_exits=True)
        pk = self.ctx.shard_resolve(int(entry[rank]), capacity=cap_pk)
        self.ctx.shard_set_exit(int(leave[rank]))
        return pk
