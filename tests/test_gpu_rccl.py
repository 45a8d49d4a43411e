"""RCCL executes the time-sharded step on ONE GPU: a world-1 "nccl" process group, the receiver with
force_collectives -- the tail exchange (a send / receive to itself inside batch_isend_irecv, or the all_gather fallback where the
backend refuses that) and the exit table's all_gather_into_tensor really run through the backend, ordered against the context's
own stream by am_signal_stream / am_wait_for_stream exactly as at world 8.  No scaling claim follows from it: one rank."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rccl_runs_the_sharded_step_at_world_one(hip_lib, oracle_mod):
    import torch
    import torch.distributed as dist
    import synth
    from air_modes import _capi
    from air_modes.sharded import ShardedReceiver
    rate, n, steps = 64e6, 3_000_000, 3
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29731", rank=0, world_size=1)
    try:
        assert dist.get_backend() == "nccl"
        dev = torch.device("cuda", 0)
        iq, _ = synth.synth_capture(rate, steps * n, 12000.0, seed=6161)
        ctx = _capi.Context(rate, 7.0, True, device=0, lib=hip_lib)
        ctx.set_rx_time(0, 1000, 0.25)
        ctx.set_rx_time(n + 12345, 2000, 0.5)
        rx = ShardedReceiver(ctx, 0, 1, n, device=dev, force_collectives=True)
        assert rx.force and rx.host_free
        out = []
        for k in range(steps):
            rx.chunk.copy_(torch.from_numpy(iq[k * n:(k + 1) * n].copy().view(np.float32)).to(dev))
            out.append(rx.step(flush=(k == steps - 1)))
        got = np.concatenate(out)
        want = oracle_mod.demod(iq, rate, rx_time=[(0, 1000, 0.25), (n + 12345, 2000, 0.5)])
        assert len(want) > 100 and np.array_equal(got, want)
        assert rx.sync_steps == 0                                  # every step host-free: nothing fell back to the host tables
        assert rx.host_us["all_gather"] > 0.0 and rx.host_us["tail_exchange"] > 0.0
        print("rccl world 1: tail by %s, host us per step: tail exchange %.1f, all_gather %.1f, rccl %s"
              % ("all_gather (send to self refused)" if rx.tail_by_gather else "send / receive to itself",
                 rx.host_us["tail_exchange"] / steps, rx.host_us["all_gather"] / steps, ".".join(map(str, torch.cuda.nccl.version()))))
        rx.close()
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("through_rccl,small_table", [(True, 512), (False, 512), (True, 1)])
def test_steps_in_flight_at_world_one(hip_lib, oracle_mod, through_rccl, small_table):
    """PipelinedShardedReceiver on the device: step k + 1's tail exchange, scan and all-gather enqueued before step k is resolved --
    through a world-1 "nccl" group (asynchronous all_gather_into_tensor / batch_isend_irecv on the receiver's own stream) and
    without a group (the floor).  Six steps, twice == the oracle over the whole stream; a density jump in the second stream makes
    a flagged step (repeated on the synchronous path while its successor is scanned)."""
    import os
    import torch
    import torch.distributed as dist
    import synth
    from air_modes import _capi
    from air_modes.sharded import PipelinedShardedReceiver
    rate, n, steps = 64e6, 3_000_000, 6
    if through_rccl:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29733", rank=0, world_size=1)
    try:
        dev = torch.device("cuda", 0)
        ctxs = [_capi.Context(rate, 7.0, True, device=0, lib=hip_lib) for _ in range(2)]
        # small_table = 1: a message holds ONE table entry -- nearly every step is flagged in its header and repeated on the
        # synchronous path (host tables) while its successor is already scanned: the redo path on the device
        rx = PipelinedShardedReceiver(ctxs, 0, 1, n, device=dev, force_collectives=through_rccl, small_table=small_table)
        assert rx.force == through_rccl
        for lams in ([12000.0] * steps, [300.0, 300.0, 300.0, 40000.0, 300.0, 12000.0]):
            iq = np.concatenate([synth.synth_capture(rate, n, lam, seed=6200 + k)[0] for k, lam in enumerate(lams)])
            tags = [(0, 1000, 0.25), (n + 12345, 2000, 0.5)]
            for tg in tags:
                rx.set_rx_time(*tg)
            out = []
            for k in range(steps):
                rx.chunk.copy_(torch.from_numpy(iq[k * n:(k + 1) * n].copy().view(np.float32)).to(dev))
                rx.submit(flush=(k == steps - 1))
                if k > 0:
                    out.append(rx.collect())
            out.append(rx.collect())
            want = oracle_mod.demod(iq, rate, rx_time=tags)
            got = np.concatenate(out)
            assert len(want) > 100 and got.tobytes() == want.tobytes(), (len(got), len(want), lams)
            print("steps in flight (%s): %d packets, %d steps on the synchronous path so far"
                  % ("rccl world 1" if through_rccl else "no group", len(got), rx.sync_steps))
        assert (rx.sync_steps >= steps) if small_table == 1 else (rx.sync_steps <= 2)
        rx.close()
        for c in ctxs:
            c.close()
    finally:
        if through_rccl:
            dist.destroy_process_group()


def test_one_collective_per_step_through_rccl(hip_lib, oracle_mod):
    """ShardedReceiver(lookahead=True) at world 1 through RCCL: the all_gather_into_tensor of the exit table carries the tail that
    closes the ring; no send / receive at all.  Four steps over three rotating buffers == the oracle over the whole stream."""
    import torch
    import torch.distributed as dist
    import synth
    from air_modes import _capi
    from air_modes.sharded import ShardedReceiver
    rate, n, steps = 64e6, 3_000_000, 4
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29735", rank=0, world_size=1)
    try:
        dev = torch.device("cuda", 0)
        iq, _ = synth.synth_capture(rate, steps * n, 12000.0, seed=6363)
        ctx = _capi.Context(rate, 7.0, True, device=0, lib=hip_lib)
        ctx.set_rx_time(0, 1000, 0.25)
        rx = ShardedReceiver(ctx, 0, 1, n, device=dev, force_collectives=True, buffers=3, lookahead=True)
        rx.chunk.copy_(torch.from_numpy(iq[:n].copy().view(np.float32)).to(dev))
        out = []
        for k in range(steps):
            if k + 1 < steps:
                rx.chunk_ahead.copy_(torch.from_numpy(iq[(k + 1) * n:(k + 2) * n].copy().view(np.float32)).to(dev))
            out.append(rx.step(flush=(k == steps - 1), ahead=(k + 1 < steps)))
        want = oracle_mod.demod(iq, rate, rx_time=[(0, 1000, 0.25)])
        got = np.concatenate(out)
        assert len(want) > 100 and got.tobytes() == want.tobytes()
        assert rx.sync_steps == 0 and len(rx.host_us_steps["tail_exchange"]) == 0 and len(rx.host_us_steps["all_gather"]) == steps
        rx.close()
        ctx.close()
    finally:
        dist.destroy_process_group()
