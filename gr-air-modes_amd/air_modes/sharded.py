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
  lookahead=True (round 6): steps 1 and 3 are ONE collective -- every rank appends to its message the tail the NEXT step needs
  (of its next chunk; the last rank: of its current one), and the next step starts with a device copy out of the gathered buffer.
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
                 force_collectives=False, share_stream=True, buffers=1, lookahead=False):
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
        # buffers > 1: that many halo'd chunk buffers, used in turn (the step after this one finds its samples in the next one:
        # a source that writes ahead of the decode; bench.py rotates three distinct captures so that the Infinity Cache cannot
        # serve a step's samples from the step before)
        self._nbuf = max(1, int(buffers))
        # lookahead (round 6): ONE collective per step.  The all-gather that carries the exit tables also carries the samples the NEXT
        # step needs in front of its chunks -- every rank appends the last `halo` samples of its next chunk (the last rank: of its
        # current one, which closes the ring) -- so the next step starts with a device copy instead of a send / receive.  Needs the
        # next step's samples in the next buffer when step(ahead=True) is called (buffers >= 2; a one-rank receiver needs nothing).
        self.lookahead = bool(lookahead)
        self._halo_ready = False
        self._halo_copied = False
        self._alloc(device if device is not None else "cpu")
        # the device-side exchange hands device pointers to kernels: only where the buffers live on the GPU (or where
        # "device memory" is host memory: the CPU emulation the tests run on)
        emulated = bool(getattr(ctx.lib, "emulated", False))   # (am_is_emulated(): asked of the library, not guessed from its path)
        self.host_free = bool(host_free) and (self._buf.is_cuda or emulated)
        # the rank whose tail the next STEP needs (the last one) keeps it inside the resolve call: behind the slicing, which
        # still reads the samples, in front of the completion -- done when step() returns, before `chunk` is overwritten
        self._keeps = self.rank == self.world - 1
        ctx._keep_owner = id(self)                        # (a context serves one receiver at a time: the newest)
        self._select(0)
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
        self._bufs = [t.zeros((self.halo + self.n) * 2, dtype=t.float32, device=dev) for _ in range(self._nbuf)]
        self._buf = self._bufs[0]
        self._tail = t.zeros(self.halo * 2, dtype=t.float32, device=dev)     # the last rank's tail of the step before
        # views used every step (slicing a tensor costs microseconds of host time each)
        self._views = [(b, b[:self.halo * 2], b[self.n * 2:], b[self.halo * 2:(self.halo + self.n) * 2]) for b in self._bufs]
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
        if self.lookahead:
            ext = words + self.halo                      # (halo complex samples = halo int64 words)
            self._words = words
            self._amsg_x = t.zeros(ext, dtype=t.int64, device=dev)
            self._amsg_x_tail = self._amsg_x[words:].view(t.float32)
            self._agath_x = t.zeros(self.world * ext, dtype=t.int64, device=dev)
            rows = self._agath_x.view(self.world, ext)
            self._gathered_tables = rows[:, :words]
            self._gathered_tails = [rows[r, words:].view(t.float32) for r in range(self.world)]
            self._tables_dense = self._agath.view(self.world, words)

    def _select(self, i):
        """The buffer the NEXT step reads: `chunk` (the caller's samples), the halo in front of it, the own samples' last `halo`."""
        self._bi = i % self._nbuf
        self._buf, self._halo_view, self._own_tail, self.chunk = self._views[self._bi]
        if self._keeps:
            self.ctx.shard_keep_tail(self._tail.data_ptr(), self._own_tail.data_ptr(), self.halo * 8)
        elif self._nbuf == 1 or i == 0:
            self.ctx.shard_keep_tail(0, 0, 0)

    @property
    def chunk_ahead(self):
        """buffers >= 2: where the samples of the step AFTER the next one to run go (written before step(ahead=True))."""
        return self._views[(self._bi + 1) % self._nbuf][3]

    def reset(self):
        """Start a new stream at sample 0 (what step(flush=True) does at its end)."""
        self.ctx.reset()
        self.k = 0
        self._wrap_posted = False
        self._halo_ready = False

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

    def step(self, flush=False, ahead=False):
        """One pass over the samples currently in `chunk`: the next world*n samples of the stream.  flush: they are the
        stream's last (every rank must say so).  ahead (with lookahead=True, every rank alike): the NEXT step's samples are
        already in the next buffer.  Returns this rank's accepted packets."""
        if self._tstream is not None:
            # the receiver's stream: behind whoever filled `chunk` on the caller's current stream, then current for the whole step
            self._tstream.wait_stream(self.torch.cuda.current_stream(self._buf.device))
            with self.torch.cuda.stream(self._tstream):
                return self._step(flush, ahead)
        return self._step(flush, ahead)

    def _step(self, flush, ahead=False):
        t, dist = self.torch, self.dist
        n, world, rank, halo, H = self.n, self.world, self.rank, self.halo, self.hold
        buf, own = self._buf, self.chunk
        on_gpu = buf.is_cuda
        S0 = self.k * world * n                              # absolute index of the step's first sample
        last = rank == world - 1
        # 1. the samples in front of the own ones.  The transfer that closes the ring -- the last rank's tail of the step BEFORE, to
        # rank 0 -- was posted at the end of that step (round 6: _post_wrap), off this step's critical path; what is left here are the
        # transfers inside the step, from every rank to the next
        look = self.lookahead and self.host_free and (world > 1 or self.force)
        prefetched = look and self._halo_ready
        self._halo_ready = False
        if prefetched:
            # the samples in front of the own ones travelled with the last step's exit tables (with more than one buffer they were
            # copied into place right behind that all-gather, in the shadow of the resolve kernels)
            if not self._halo_copied:
                self._halo_view.copy_(self._gathered_tails[(rank - 1) % world])
        elif world > 1 or (self.force and not self.tail_by_gather):
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
        if prefetched:
            pass
        elif self.force and self.tail_by_gather and self.k > 0:
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
            carry_tails = look and ((ahead and not flush and self._nbuf > 1) or world == 1) and not flush
            amsg = self._amsg_x if look else self._amsg
            self.ctx.shard_scan_async(ptr, a0, a1, total, amsg.data_ptr(), self.small_cap, device_in=on_gpu, more=more)
            if world > 1 or self.force:
                if carry_tails:
                    # what the next step needs in front of its chunks: the last rank's tail of THIS step (the ring closes), everybody
                    # else's tail of the NEXT one
                    self._amsg_x_tail.copy_(self._own_tail if last else self._views[(self._bi + 1) % self._nbuf][2])
                if on_gpu and not shared:
                    self.ctx.signal_stream(cur)              # the collective waits (on the device) for the table
                tc = time.perf_counter()
                if look:
                    dist.all_gather_into_tensor(self._agath_x, self._amsg_x, group=self.group)
                    self._tables_dense.copy_(self._gathered_tables)      # (the resolve step reads the tables at their own stride)
                    self._halo_ready = carry_tails
                    self._halo_copied = carry_tails and self._nbuf > 1
                    if self._halo_copied:
                        self._views[(self._bi + 1) % self._nbuf][1].copy_(self._gathered_tails[(rank - 1) % world])
                else:
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
        if self._nbuf > 1:
            self._select(self._bi + 1)                       # (before the ring-closing transfer: it lands in the next step's halo)
        if flush:
            self.reset()
        else:
            self.k += 1                                      # (the last rank's tail was kept inside the resolve call)
            if not self._halo_ready:
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


class PipelinedShardedReceiver(object):
    """The time-sharded receiver with STEPS IN FLIGHT (round 6).  ShardedReceiver.step() is one step at a time: the host waits for a
    step's completion ticket before it enqueues the next one, and the device idles meanwhile (~70 us per step through the Python
    side).  Here step k + 1 is scanned BEFORE step k is collected:

        rx.chunk.copy_(samples of step 0); rx.submit()
        rx.chunk.copy_(samples of step 1); rx.submit()        # all-gather + resolve of step 0, then the scan of step 1: all enqueued
        pk0 = rx.collect()                                    # step 0's packets (the scan of step 1 is queued behind them)
        rx.chunk.copy_(samples of step 2); rx.submit(); pk1 = rx.collect(); ...   # collect(j) comes before submit(j + 2)

    What makes it possible: a step's scan never depended on where the greedy scan enters the chunk; the position it starts from --
    where the scan left the LAST chunk of the step before -- no longer travels in the next step's message (written at scan time,
    i.e. too early) but stays in a device word every rank keeps for itself: the resolve step composes the entry through ALL ranks'
    tables (am_shard_resolve_submit: cur_in / carry_out) and leaves the step's final position there.  Two contexts and two halo'd
    chunk buffers per rank alternate.  ONE collective per step, issued synchronously on the receiver's own stream (stream order is
    all the ordering there is; nothing hops between streams): the all-gather of step k's exit tables is issued when step k + 1 is
    submitted, so that it also carries what step k + 1 needs in front of its chunks -- every rank's tail of the chunk it has just
    been handed, the last rank's tail of step k (the ring closes).  A stream's first step, and a step submitted after the caller
    drained the pipeline, exchange their tails by an all-gather of their own (in-stream as well: no send / receive anywhere).  A step whose table did not fit its message or whose scan outgrew
    its capacity is flagged in the headers every rank reads: collect() repeats it on the synchronous path (host tables) on every
    rank alike; the step scanned behind it stays valid (its resolve is only enqueued once its predecessor is known to be good).
    Measured at world 1, 64 M samples per step: 0.310 ms per step through RCCL at 12 steps, 0.295 over 200 (one step at a time:
    0.349-0.360); 0.289 without a group (0.305-0.313) -- profiles/r6_rccl/steps_in_flight.txt.
    Packets of all (step, rank) pairs in order == the single-stream packet list.  Give BOTH contexts the same rx_time tags."""

    def __init__(self, ctxs, rank, world, n_per_rank, group=None, device=None, small_table=512, force_collectives=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        if len(ctxs) != 2:
            raise ValueError("two contexts (same rate, threshold, filters): steps alternate between them")
        self.ctxs, self.rank, self.world, self.n, self.group = list(ctxs), int(rank), int(world), int(n_per_rank), group
        ctx = self.ctxs[0]
        self.force = bool(force_collectives) and self.world == 1 and dist.is_available() and dist.is_initialized()
        self.tail_by_gather = False
        self.left, self.hold = ctx.shard_halo()
        self.halo = self.left + self.hold
        if self.n < self.halo:
            raise ValueError("chunk shorter than the halo (%d samples)" % self.halo)
        spc_hi = max(int(-(-ctx.get_rate() // 2e6)), 1)
        self.tab_cap = 241 * spc_hi + 4
        self.small_cap = max(1, min(int(small_table), self.tab_cap))
        dev = device if device is not None else "cpu"
        t = torch
        self._bufs = [t.zeros((self.halo + self.n) * 2, dtype=t.float32, device=dev) for _ in range(2)]
        emulated = bool(getattr(ctx.lib, "emulated", False))
        if not (self._bufs[0].is_cuda or emulated):
            raise ValueError("steps in flight need device buffers (or the CPU emulation): the tables stay on the device")
        words = 2 * (_capi.SHARD_MSG_HEADER + self.small_cap)
        ext = words + self.halo                           # a message: header + table, then `halo` complex samples (int64 words)
        self._amsg = [t.zeros(ext, dtype=t.int64, device=dev) for _ in range(2)]
        self._amsg_tail = [m[words:].view(t.float32) for m in self._amsg]
        self._agath_x = [t.zeros(self.world * ext, dtype=t.int64, device=dev) for _ in range(2)]
        self._gathered_tables = [g.view(self.world, ext)[:, :words] for g in self._agath_x]
        self._gathered_tails = [[g.view(self.world, ext)[r, words:].view(t.float32) for r in range(self.world)] for g in self._agath_x]
        self._tail_msg = t.zeros(self.halo * 2, dtype=t.float32, device=dev)     # tails alone (a stream's first step, after a drain)
        self._tails_x = t.zeros(self.world * self.halo * 2, dtype=t.float32, device=dev)
        self._tails_rows = [self._tails_x[r * self.halo * 2:(r + 1) * self.halo * 2] for r in range(self.world)]
        self._agath = t.zeros(self.world * words, dtype=t.int64, device=dev)     # the tables at the stride the resolve step reads
        self._tables_dense = self._agath.view(self.world, words)
        self._carry = t.zeros(2, dtype=t.int64, device=dev)          # [0]: where the scan left the last chunk of the step resolved last
        self._host_tab = np.zeros(self.tab_cap, _capi.EXIT_DTYPE)
        self._msg = t.zeros(2 + 2 * self.tab_cap, dtype=t.int64, device=dev)
        self._msgs = [t.empty_like(self._msg) for _ in range(self.world)]
        self._tstream = None
        if self._bufs[0].is_cuda:
            self._tstream = t.cuda.Stream(device=self._bufs[0].device)
            self._caller = None
            for c in self.ctxs:
                c.set_stream(self._tstream.cuda_stream)
        for c in self.ctxs:
            c.shard_keep_tail(0, 0, 0)
        self.k = 0                       # steps of the current stream submitted so far
        self._scanned = None             # (k, slot, flush): scanned; its all-gather and resolve step not enqueued yet
        self._resolved = None            # (k, slot, flush): resolve enqueued, not collected yet
        self._ended = False
        self.sync_steps = 0
        self.host_us = {"tail_exchange": 0.0, "all_gather": 0.0, "steps": 0}
        self.host_us_steps = {"tail_exchange": [], "all_gather": []}

    # ---- what ShardedReceiver offers too -----------------------------------------------------------------------------------------
    @property
    def chunk(self):
        """This rank's 2*n float32 samples of the NEXT step to submit (write them here, then submit())."""
        b = self._bufs[self.k % 2]
        return b[self.halo * 2:]

    def set_rx_time(self, offset, secs, frac):
        for c in self.ctxs:
            c.set_rx_time(offset, secs, frac)

    def close(self):
        for c in getattr(self, "ctxs", []):
            if getattr(c, "_h", None) and self._tstream is not None:
                c.set_stream(None)
        self._tstream = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        if self._scanned is not None or self._resolved is not None:
            raise RuntimeError("collect the steps in flight first")
        for c in self.ctxs:
            c.reset()
        self._carry.zero_()
        self.k = 0
        self._ended = False

    def in_flight(self):
        return (self._scanned is not None) + (self._resolved is not None)

    # ---- one step's geometry --------------------------------------------------------------------------------------------------------
    def _geometry(self, k, flush):
        n, world, rank, H = self.n, self.world, self.rank, self.hold
        S0 = k * world * n
        last = rank == world - 1
        a0 = 0 if (k == 0 and rank == 0) else S0 + rank * n - H
        total = S0 + world * n
        a1 = total if (flush and last) else S0 + (rank + 1) * n - H
        b0 = S0 + rank * n - self.halo                      # absolute index of the buffer's first sample
        lo = max(0, a0 - self.left)
        return a0, a1, total, (lo - b0) * 8

    def _with_stream(self, fn, *a):
        if self._tstream is None:
            return fn(*a)
        with self.torch.cuda.stream(self._tstream):
            return fn(*a)

    # ---- submit / collect --------------------------------------------------------------------------------------------------------------
    def submit(self, flush=False):
        """Enqueue the step whose samples are in `chunk`: its tail exchange, scan and all-gather -- and the resolve step of the step
        before it.  flush: the stream's last step (every rank must say so)."""
        if self._ended:
            raise RuntimeError("the stream was flushed: collect its steps before the next stream starts")
        if self._resolved is not None and self._scanned is not None:
            raise RuntimeError("collect the oldest step first (collect(j) comes before submit(j + 2))")
        if self._tstream is not None:
            self._caller = self.torch.cuda.current_stream(self._bufs[0].device)
            self._tstream.wait_stream(self._caller)
        self._with_stream(self._submit, bool(flush))

    def _submit(self, flush):
        t, dist = self.torch, self.dist
        k, world, rank, halo, n = self.k, self.world, self.rank, self.halo, self.n
        s = k % 2
        buf, prev = self._bufs[s], self._bufs[1 - s]
        halo_view, own_tail, prev_tail = buf[:halo * 2], buf[n * 2:], prev[n * 2:]
        last = rank == world - 1
        on_gpu = buf.is_cuda
        collectives = world > 1 or self.force
        # 1. the step before: ITS all-gather only now, so that it can carry what THIS step needs in front of its chunks (every rank's
        #    tail of the chunk it has just been handed; the last rank's tail of the step before, which closes the ring) -- one
        #    collective per step, issued synchronously on the receiver's stream: stream order is the only ordering, nothing hops --
        #    then its resolve step, queued in front of this step's scan (the host runs ahead of the device by it)
        prefetched = self._gather_and_resolve(s)
        # 2. no step before this one in flight (a stream's first step, or the caller drained the pipeline): the tails travel alone --
        #    by an all-gather of their own, in-stream like the other one (a send / receive would run on the backend's stream: two
        #    hops; measured ~0.2 ms once per drain through RCCL)
        if not prefetched and (k > 0 or world > 1):
            tc = time.perf_counter()
            if collectives:
                self._tail_msg.copy_(prev_tail if (last and k > 0) else own_tail)
                dist.all_gather_into_tensor(self._tails_x, self._tail_msg, group=self.group)
                if rank > 0 or k > 0:
                    halo_view.copy_(self._tails_rows[(rank - 1) % world])
            elif k > 0:
                self.ctxs[s].stream_copy(halo_view.data_ptr(), prev_tail.data_ptr(), halo * 8)   # one rank, no group
            dtt = (time.perf_counter() - tc) * 1e6
            self.host_us["tail_exchange"] += dtt
            self.host_us_steps["tail_exchange"].append(dtt)
        # 3. the scan
        a0, a1, total, off = self._geometry(k, flush)
        tc = time.perf_counter()
        self.ctxs[s].shard_scan_async(buf.data_ptr() + off, a0, a1, total, self._amsg[s].data_ptr(), self.small_cap,
                                      device_in=on_gpu, more=not flush)
        self.host_us_steps.setdefault("scan_enqueue", []).append((time.perf_counter() - tc) * 1e6)
        self._scanned = (k, s, flush)
        self.host_us["steps"] += 1
        self.k += 1
        if flush:
            self._ended = True

    def _gather_and_resolve(self, next_slot):
        """The scanned step's all-gather and resolve step, enqueued.  next_slot: the buffer of the step submitted right now (None: no
        successor yet -- collect() drains) -- its halo is filled from what the gather brings.  Returns whether it was."""
        if self._scanned is None:
            return False
        if self._resolved is not None:
            raise RuntimeError("collect the oldest step first (collect(j) comes before submit(j + 2))")
        t, dist = self.torch, self.dist
        k, s, flush = self._scanned
        world, rank, halo, n = self.world, self.rank, self.halo, self.n
        last = rank == world - 1
        carry = next_slot is not None and not flush
        msgs = self._amsg[s]
        if world > 1 or self.force:
            tc = time.perf_counter()
            if carry:
                self._amsg_tail[s].copy_((self._bufs[s] if last else self._bufs[next_slot])[n * 2:])
            dist.all_gather_into_tensor(self._agath_x[s], self._amsg[s], group=self.group)
            self._tables_dense.copy_(self._gathered_tables[s])
            if carry:
                self._bufs[next_slot][:halo * 2].copy_(self._gathered_tails[s][(rank - 1) % world])
            dta = (time.perf_counter() - tc) * 1e6
            self.host_us["all_gather"] += dta
            self.host_us_steps["all_gather"].append(dta)
            msgs = self._agath
        elif carry:
            # one rank, no group: it is its own predecessor
            self.ctxs[s].stream_copy(self._bufs[next_slot].data_ptr(), self._bufs[s].data_ptr() + n * 8, halo * 8)
        tc = time.perf_counter()
        self.ctxs[s].shard_resolve_submit(msgs.data_ptr(), world, rank, self.small_cap,
                                          cur_in_ptr=self._carry.data_ptr(), carry_out_ptr=self._carry.data_ptr())
        self.host_us_steps.setdefault("resolve_enqueue", []).append((time.perf_counter() - tc) * 1e6)
        self._resolved, self._scanned = (k, s, flush), None
        return carry

    def collect(self):
        """The packets of the oldest step in flight."""
        if self._resolved is None:
            if self._scanned is None:
                raise RuntimeError("no step in flight")
            self._with_stream(self._gather_and_resolve, None)
        k, s, flush = self._resolved
        cap_pk = max(64, self.n // 2000 + 64)
        tc = time.perf_counter()
        pk, redo = self.ctxs[s].shard_resolve_collect(capacity=cap_pk)
        self.host_us_steps.setdefault("collect_wait_and_fetch", []).append((time.perf_counter() - tc) * 1e6)
        if redo:
            self.sync_steps += 1
            pk = self._with_stream(self._redo, k, s, flush, cap_pk)
        self._resolved = None
        if flush:
            self._ended = False
            if self._scanned is None:
                self.reset()
        return pk

    def step(self, flush=False):
        """submit + collect (no overlap between steps; ShardedReceiver's interface)."""
        self.submit(flush)
        return self.collect()

    def _redo(self, k, s, flush, cap_pk):
        """Step k once more with the tables on the host (am_shard_scan -> all_gather -> am_shard_entry2 -> am_shard_resolve), on every
        rank alike; the carry of the step before it is still in the device word (a flagged step does not write it)."""
        t, dist = self.torch, self.dist
        world, rank = self.world, self.rank
        c, buf = self.ctxs[s], self._bufs[s]
        on_gpu = buf.is_cuda
        a0, a1, total, off = self._geometry(k, flush)
        L = c.lib.L
        got = C.c_uint64(0)
        flags = (_capi.AM_F_DEVICE_IN if on_gpu else 0) | (0 if flush else _capi.AM_F_MORE)
        rc = L.am_shard_scan(c._h, buf.data_ptr() + off, a0, a1, total, flags, self._host_tab.ctypes.data, self.tab_cap, C.byref(got))
        c._chk(rc)
        m = int(got.value)
        cur_in = int(np.int64(self._carry[0].item()).astype(np.uint64))
        if world > 1:
            msg = np.zeros(2 + 2 * self.tab_cap, np.int64)
            msg[0] = m
            msg[2:2 + 2 * m] = self._host_tab[:m].view(np.int64)
            self._msg.copy_(t.from_numpy(msg))
            dist.all_gather(self._msgs, self._msg, group=self.group)
            allm = t.stack(self._msgs).cpu().numpy()
            tables = [allm[r, 2:2 + 2 * int(allm[r, 0])].copy().view(_capi.EXIT_DTYPE) for r in range(world)]
        else:
            tables = [self._host_tab[:m].copy()]
        entry, leave = _capi.shard_entries(c.lib, tables, cur_in=cur_in, with_exits=True)
        pk = c.shard_resolve(int(entry[rank]), capacity=cap_pk)
        self._carry[0] = int(np.uint64(leave[world - 1]).astype(np.int64))
        return pk
