#!/usr/bin/env python3
"""bench.py -- throughput of the Mode-S receive hot path on MI355X.

A "step" is one pass of the whole hot path (IQ -> packet list) over one batch of synthetic IQ that is
already resident in HBM when the timed region starts.

  python bench.py [--gpus N --steps K --warmup W] [--workload 64msps|2msps|20msps] [--lambda L] [--replicas]

N = 1  workload "64msps" (BASELINE.json configs[2]: synthetic 64 Msps IQ, Poisson-injected Mode-S bursts in
       AWGN).  The timed loop rotates over three distinct 512 MB batches (the 256 MiB Infinity Cache cannot
       serve them), the packets of the last batch are compared with the oracle (`parity`).  The default burst
       rate (20 000 /s, 87 % airtime) is a stress density; `realistic_density` repeats the measurement at
       2 000 bursts/s in the same run.
N > 1  configs[3]: one 64 Msps stream time-sharded over the N GPUs, one process per GPU (torch.distributed
       over RCCL): a step is the next N seconds of ONE continuing stream, one second per GPU; every rank's tail
       travels to the next rank as a point-to-point send / receive (KB scale), every rank scans its chunk, the
       scan's exit tables are all-gathered, every rank slices its own hits; scan position and sample count
       cross the steps.  Per-GPU work is fixed as N grows ("weak").  `python bench.py --gpus N` launches the
       N ranks itself when it is not already running under torch.distributed.run.  `parity` for N > 1: a short
       stream through the same N-rank receiver in two steps against the oracle over the whole stream, plus
       rank 0's full-size packets against the oracle over its samples.
       --replicas: configs[4] instead -- N independent receivers (20 Msps each unless --workload says
       otherwise), one per GPU, no collective; aggregate samples/s and packets/s, every rank checked
       against the oracle.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the front end, the only kernel that
touches every sample): algorithmic bytes = 8 B per complex sample.  `cpu_baseline` is the oracle (a scalar C
port of the reference path) timed on this host.

--emu (tests only): the same orchestration on CPU -- gloo, the CPU-fiber build of the kernels (tests/emu),
tiny sizes.  The line it prints carries "emulated": true and is not a measurement.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "gr-air-modes_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

# One hardware queue per stream: the runtime multiplexes streams onto 4 queues by default, and two of am_pipe's three
# contexts then share one and serialise (profiles/r3_final/README.md: 215 -> 240 GS/s with three batches in flight).
# A documented runtime setting, read when HIP initialises (an application that runs am_pipe sets it too).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402  (before the HIP library: one HIP runtime per process)
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
REALISTIC_LAMBDA = 2000.0  # bursts per second of the second density


def ctx_messages(ctx, packets):
    """Message texts of a packet array as the library formats them (first message: 6 significant digits)."""
    return ctx.lib.format_messages(packets, True)          # (am_format_messages: one call per batch)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=["64msps", "2msps", "20msps"])
    ap.add_argument("--seconds", type=float, default=None, help="signal seconds per GPU per step")
    ap.add_argument("--lambda", dest="lam", type=float, default=None, help="bursts per second (default: the workload's)")
    ap.add_argument("--batches", type=int, default=3, help="distinct batches the timed loop rotates over (N = 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baseline figures (the parity check stays)")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle altogether (A/B loops that compare packet counts)")
    ap.add_argument("--no-extra", action="store_true", help="skip the realistic-density and batches-in-flight figures")
    ap.add_argument("--inflight", type=int, default=1, help="batches in flight (contexts/streams driven by host threads)")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the extra 3-batches-in-flight figure (profiling runs)")
    ap.add_argument("--stream-seconds", type=float, default=16.0, help="signal seconds of the pipelined_stream leg (ONE continuing stream, "
                                                                      "its chunks in flight: am_spipe); 0 skips it")
    ap.add_argument("--sustained-steps", type=int, default=8000, help="steps of the extra sustained leg (>= 0.5 s of device work: a sampler "
                                                                        "of GPU activity sees the device busy); 0 skips it")
    ap.add_argument("--no-lookahead", action="store_true",
                    help="time-sharded mode: tails by send / receive + exit tables by all-gather (two collectives per step) instead of the "
                         "default, ShardedReceiver(lookahead=True): the all-gather of the exit tables carries the next step's tails -- one "
                         "collective per step; needs the next step's samples resident, which they are here (three rotating buffers).  "
                         "RCCL at world 1: 0.351 vs 0.541 ms per step (profiles/r6_rccl/lookahead.txt)")
    ap.add_argument("--steps-in-flight", action="store_true", help="(the default in the time-sharded mode since round 6; kept for old command lines)")
    ap.add_argument("--no-steps-in-flight", action="store_true",
                    help="time-sharded mode: one step at a time (ShardedReceiver.step) instead of PipelinedShardedReceiver (step k + 1 is "
                         "scanned before step k is resolved; one in-stream all-gather per step carries the exit tables and the next "
                         "step's tails).  The timed region fills and drains the pipeline.  World 1 through RCCL: 0.310 vs "
                         "0.349-0.360 ms per step; without a group 0.289 vs 0.305-0.313 (profiles/r6_rccl/steps_in_flight.txt)")
    ap.add_argument("--force-sharded", action="store_true", help="N=1 through the time-sharded code path (overhead check)")
    ap.add_argument("--backend", default=None, choices=["nccl", "gloo"],
                    help="with --force-sharded at N=1: a world-1 process group of this backend, and the receiver goes through its "
                         "collectives (tail to itself, all_gather_into_tensor of the exit table): RCCL executes the sharded step on one GPU")
    ap.add_argument("--streams", type=int, default=1, help="K independent streams (receivers) of the workload in ONE scan per step "
                                                           "(am_process_multi): N = 1, value counts all K streams")
    ap.add_argument("--replicas", action="store_true", help="N independent receivers, one per GPU (configs[4])")
    ap.add_argument("--emu", action="store_true", help="tests only: CPU emulation of the kernels, gloo")
    return ap.parse_args()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start one process per GPU ourselves
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call(cmd, env=env))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if args.emu else "nccl", rank=rank, world_size=world)
        assert dist.get_world_size() == world
    elif args.force_sharded and args.backend:
        dist.init_process_group(args.backend, init_method="tcp://127.0.0.1:%d" % free_port(), rank=0, world_size=1)
    if args.emu:
        dev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs a HIP device (there is no CPU fallback)"
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)

    def sync():
        if not args.emu:
            torch.cuda.synchronize()

    import synth
    from air_modes import _capi

    lib = _capi.Library(os.path.join(ROOT, "tests", "emu", "libairmodes_emu.so")) if args.emu else None
    mode = "replicas" if (args.replicas and world >= 1) else ("sharded" if (world > 1 or args.force_sharded) else "single")
    workload = args.workload or ("20msps" if args.replicas else "64msps")
    rate, secs, lam, seed = synth.CONFIGS[workload]
    if args.seconds is not None:
        secs = args.seconds
    if args.lam is not None:
        lam = args.lam
    n = int(round(rate * secs))                       # samples per GPU per step
    spc = int(rate / 2e6)

    def new_ctx():
        return _capi.Context(rate, 7.0, True, device=(0 if args.emu else local), lib=lib) if lib is not None else \
            _capi.Context(rate, 7.0, True, device=local)

    ctx = new_ctx()
    extra = {}
    K = max(1, args.streams)
    if K > 1 and mode == "sharded":
        raise SystemExit("bench.py: --streams goes with independent receivers (N = 1, or --replicas), not with one time-sharded stream")

    def k_streams_setup(k, nbatch):
        """k whole streams of the workload (different seeds) packed behind one another, zeros between them (am_multi_layout),
        resident in HBM: (host streams, device buffers, lengths, samples in the scanned buffer)."""
        lengths = np.full(k, n, np.uint64)
        off, total = ctx.multi_layout(lengths)
        hosts, devs = [], []
        for b in range(nbatch):
            st = [synth.synth_capture(rate, n, lam, seed + rank + 100 * b + 1000 * j)[0] for j in range(k)]
            d = torch.zeros(2 * total, dtype=torch.float32, device=dev)
            for j in range(k):
                d[2 * int(off[j]): 2 * (int(off[j]) + n)].copy_(torch.from_numpy(st[j].view(np.float32)))
            hosts.append(st)
            devs.append(d)
        return hosts, devs, lengths, total

    # ------------------------------------------------------------------------------------------------------------
    if mode in ("single", "replicas"):
        nb = max(1, args.batches if mode == "single" else 1)
        if K > 1:
            # one packed buffer of K streams per step (K x %d samples: far beyond the Infinity Cache, one buffer is enough)
            nb = 1
            k_hosts, d_batches, k_lengths, k_total = k_streams_setup(K, nb)
            host_batches = [k_hosts[0][0]]
        else:
            host_batches = [synth.synth_capture(rate, n, lam, seed + rank + 100 * b)[0] for b in range(nb)]
            d_batches = [torch.from_numpy(b.view(np.float32)).to(dev) for b in host_batches]
        sync()

        def scan(c, batch):
            if K > 1:
                return c.process_multi(None, k_lengths, device_ptr=batch.data_ptr())
            return c.process_iq_device(batch.data_ptr(), n, flush=True)
        inflight = max(1, args.inflight) if mode == "single" else 1
        ctxs = [ctx] + [new_ctx() for _ in range(inflight - 1)]

        def run_steps(count, ctxs_, flight, batches):
            import threading
            last = [None] * flight
            fe = [[] for _ in range(flight)]

            def worker(w):
                c = ctxs_[w]
                for k in range(w, count, flight):
                    last[w] = (k, scan(c, batches[k % len(batches)]))
                    fe[w].append(c.last_dom_ms())
            if flight == 1:
                worker(0)
            else:
                ths = [threading.Thread(target=worker, args=(w,)) for w in range(flight)]
                for t in ths:
                    t.start()
                for t in ths:
                    t.join()
            done = [x for x in last if x is not None]
            k_last, pk_last = max(done, key=lambda x: x[0]) if done else (None, None)
            return k_last, pk_last, [x for f in fe for x in f]

        def timed(count, ctxs_, flight, batches):
            if world > 1:
                dist.barrier()
            sync()
            t0 = time.perf_counter()
            k_last, pk, fe_ms = run_steps(count, ctxs_, flight, batches)
            sync()
            if world > 1:
                dist.barrier()
            return time.perf_counter() - t0, k_last, pk, fe_ms

        # untimed: every context past its first (allocating) and second (capacity) call, every batch seen once
        per_batch = []
        for b in range(nb):
            r0 = scan(ctx, d_batches[b])
            per_batch.append(sum(len(x) for x in r0) if K > 1 else len(r0))
        run_steps(max(args.warmup, 2 * inflight), ctxs, inflight, d_batches)
        dt, k_last, pk, fe_ms = timed(args.steps, ctxs, inflight, d_batches)
        last_batch = (k_last % nb) if k_last is not None else 0
        npk_steps = sum(per_batch[k % nb] for k in range(args.steps))
        if K > 1:
            # parity: EVERY stream's packets against the oracle over that stream alone
            parity_k = None
            if not args.no_parity:
                import oracle
                parity_k = all(g.tobytes() == oracle.demod(x, rate, 7.0, True).tobytes() for g, x in zip(pk, k_hosts[0]))
            one = ctx.process_iq_device(d_batches[0].data_ptr(), n, flush=True)      # (stream 0 lies at offset 0)
            extra["k_streams"] = {"streams_per_scan": K, "samples_per_stream": n, "samples_scanned_per_step": int(k_total),
                                  "parity_every_stream": parity_k,
                                  "same_as_single_stream_call": bool(one.tobytes() == pk[0].tobytes()),
                                  "packets_per_stream": [int(len(g)) for g in pk]}
            pk = pk[0]
        if mode == "single" and K == 1 and not args.no_extra and workload != "64msps" and args.steps >= 3:
            # 2 / 20 Msps: one second of ONE receiver is a small job for the chip (the scan's launches are most of the step); eight
            # receivers' seconds in one scan (am_process_multi: K whole streams in one buffer, zeros between them)
            # ONE receiver as configs[1] / [4] have it (VERDICT r5 #6a): (a) multi-second scans -- eight seconds of the one stream per
            # am_process_iq call, the same launches over eight times the samples; (b) the stream in 1 s chunks with four in flight
            # (am_spipe).  The stream: this run's second of signal eight times over (a valid stream; bursts across the seams included).
            LONG = 8
            whole8 = np.tile(host_batches[0], LONG)
            d8 = torch.from_numpy(whole8.view(np.float32)).to(dev)
            sync()
            for _ in range(2):
                pk8 = ctx.process_iq_device(d8.data_ptr(), LONG * n, flush=True)
            ks8 = max(5, args.steps // 2)
            sync()
            t8 = time.perf_counter()
            fe8 = []
            for _ in range(ks8):
                pk8 = ctx.process_iq_device(d8.data_ptr(), LONG * n, flush=True)
                fe8.append(ctx.last_dom_ms())
            sync()
            dt8 = (time.perf_counter() - t8) / ks8
            fe_8 = float(np.mean(fe8))
            one = {"what": "ONE stream, %g s of signal (%d samples) per scan" % (LONG * secs, LONG * n), "value": LONG * n / dt8, "unit": "samples/s",
                   "ms_per_scan": dt8 * 1e3, "kernel_ms": fe_8,
                   "roofline_frac": (8.0 * LONG * n / (fe_8 * 1e-3) / 1e9 / HBM_PEAK_GBS) if fe_8 > 0 else 0.0,
                   "path_frac_of_hbm_peak": 8.0 * LONG * n / dt8 / 1e9 / HBM_PEAK_GBS, "packets_per_scan": int(len(pk8))}
            sp1 = _capi.StreamPipe(rate, 7.0, True, device=(0 if args.emu else local), depth=4, lib=lib) if lib is not None \
                else _capi.StreamPipe(rate, 7.0, True, device=local, depth=4)
            ch8 = [(d8.data_ptr() + 8 * k * n, n) for k in range(LONG)]       # (contiguous in memory: no tail copies)
            sp1.run(ch8)
            sync()
            t9 = time.perf_counter()
            got9 = None
            for _ in range(ks8):
                got9 = sp1.run(ch8)
            sync()
            dt9 = (time.perf_counter() - t9) / ks8
            all9 = np.concatenate(got9)
            one["chunks_in_flight"] = {"what": "the same stream in %d chunks of %g s, four in flight (am_spipe)" % (LONG, secs),
                                       "value": LONG * n / dt9, "unit": "samples/s", "ms_per_chunk": dt9 / LONG * 1e3,
                                       "path_frac_of_hbm_peak": 8.0 * LONG * n / dt9 / 1e9 / HBM_PEAK_GBS,
                                       "same_packets_as_the_long_scan": bool(all9.tobytes() == pk8.tobytes()),
                                       "chunks_redone_synchronously": sp1.redone()}
            sp1.close()
            if not args.no_parity:
                import oracle
                one["parity_vs_oracle_whole_stream"] = bool(np.array_equal(pk8, oracle.demod(whole8, rate, 7.0, True)))
            extra["one_stream_long_scans"] = one
            del d8, whole8
            run_steps(2, [ctx], 1, d_batches)
            KX = 8
            kh, kd, kl, kt = k_streams_setup(KX, 1)
            for _ in range(3):
                got = ctx.process_multi(None, kl, device_ptr=kd[0].data_ptr())
            ks = max(5, args.steps // 2)
            sync()
            tk0 = time.perf_counter()
            fek = []
            for _ in range(ks):
                got = ctx.process_multi(None, kl, device_ptr=kd[0].data_ptr())
                fek.append(ctx.last_dom_ms())
            sync()
            dtk = (time.perf_counter() - tk0) / ks
            fe_k = float(np.mean(fek))
            extra["k_streams_per_scan"] = {
                "streams_per_scan": KX, "samples_per_stream": n, "value": KX * n / dtk, "unit": "samples/s", "ms_per_step": dtk * 1e3,
                "kernel_ms": fe_k, "roofline_frac": (8.0 * kt / (fe_k * 1e-3) / 1e9 / HBM_PEAK_GBS) if fe_k > 0 else 0.0,
                "path_frac_of_hbm_peak": 8.0 * KX * n / dtk / 1e9 / HBM_PEAK_GBS,
                "packets_per_stream": [int(len(g)) for g in got]}
            if not args.no_parity:
                import oracle
                extra["k_streams_per_scan"]["parity_every_stream"] = all(
                    g.tobytes() == oracle.demod(x, rate, 7.0, True).tobytes() for g, x in zip(got, kh[0]))
            # the same with two scans in flight from ONE host thread (am_submit_multi / am_collect on two contexts used alternately):
            # the host's share of a step -- copying ~20 k packets out of pinned memory, sorting them into streams -- hides behind
            # the other scan's kernels
            ctx2 = [ctx, new_ctx()]
            ptr8 = kd[0].data_ptr()

            def fly(count):
                last = None
                for k in range(count):
                    c = ctx2[k % 2]
                    if k >= 2:
                        last = c.collect_multi()
                    c.submit_multi(None, kl, device_ptr=ptr8)
                for k in range(count, count + 2):
                    last = ctx2[k % 2].collect_multi()
                return last
            fly(4)
            ks2 = 2 * ks
            sync()
            tk1 = time.perf_counter()
            got2 = fly(ks2)
            sync()
            dtk2 = (time.perf_counter() - tk1) / ks2
            extra["k_streams_per_scan"]["two_scans_in_flight"] = {
                "value": KX * n / dtk2, "unit": "samples/s", "ms_per_step": dtk2 * 1e3, "host_threads": 1,
                "path_frac_of_hbm_peak": 8.0 * KX * n / dtk2 / 1e9 / HBM_PEAK_GBS,
                "same_packets_as_one_scan_at_a_time": all(a_.tobytes() == b_.tobytes() for a_, b_ in zip(got2, got))}
            ctx2[1].close()
            # ... and behind ONE handle (am_pipe_submit_multi / am_pipe_collect / am_pipe_multi_counts, three contexts inside):
            # VERDICT r5 #6b asks for >= 300 GS/s "from one context and one host thread".  The packets' way into per-stream lists
            # stays one linear pass over the accepted packets on the host (sort_into_streams, am_capi.hip: the accept flag has to
            # be looked at there anyway); its time is part of every step timed here
            kp = _capi.Pipe(rate, 7.0, True, device=(0 if args.emu else local), depth=3, lib=lib) if lib is not None \
                else _capi.Pipe(rate, 7.0, True, device=local, depth=3)

            def fly_pipe(count):
                last = None
                for k in range(count):
                    if kp.in_flight() == kp.depth():
                        last = kp.collect_multi()
                    kp.submit_multi_device(ptr8, kl)
                while kp.in_flight():
                    last = kp.collect_multi()
                return last
            fly_pipe(9)          # (every context twice at least: the second scan sizes its buffers from the first one's density)
            sync()
            tk2 = time.perf_counter()
            got3 = fly_pipe(ks2)
            sync()
            dtk3 = (time.perf_counter() - tk2) / ks2
            tot_ms = ctx.last_timing()[0]
            extra["k_streams_per_scan"]["scans_in_flight_one_handle"] = {
                "value": KX * n / dtk3, "unit": "samples/s", "ms_per_step": dtk3 * 1e3, "host_threads": 1, "handles": 1, "depth": 3,
                "path_frac_of_hbm_peak": 8.0 * KX * n / dtk3 / 1e9 / HBM_PEAK_GBS,
                "same_packets_as_one_scan_at_a_time": all(a_.tobytes() == b_.tobytes() for a_, b_ in zip(got3, got))}
            extra["k_streams_per_scan"]["one_scan_at_a_time_device_ms"] = tot_ms
            kp.close()
            del kd, kh
            run_steps(2, [ctx], 1, d_batches)
        if mode == "single" and K == 1 and not args.no_extra and not args.no_pipelined and inflight == 1 and args.steps >= 3:
            # the same batches with four in flight from ONE host thread (am_pipe: four contexts behind one handle, submit /
            # collect): the launch-latency-bound tail of one batch overlaps the streaming kernel of the next.  Timed over
            # 24 batches including the filling and the draining of the pipe (depth 3: ~4 % less, 5 and more: less again --
            # tools/gpu_pipe_depth.py).
            PIPE_DEPTH, PIPE_BATCHES = 4, max(24, 2 * args.steps)
            pipe = _capi.Pipe(rate, 7.0, True, device=(0 if args.emu else local), depth=PIPE_DEPTH, lib=lib) if lib is not None \
                else _capi.Pipe(rate, 7.0, True, device=local, depth=PIPE_DEPTH)

            def pipe_steps(count):
                counts, last = [], None
                for k in range(count):
                    if pipe.in_flight() == pipe.depth():
                        last = pipe.collect()
                        counts.append(len(last))
                    pipe.submit_device(d_batches[k % nb].data_ptr(), n)
                while pipe.in_flight():
                    last = pipe.collect()
                    counts.append(len(last))
                return counts, last
            pipe_steps(3 * PIPE_DEPTH)
            sync()
            t3 = time.perf_counter()
            counts3, last3 = pipe_steps(PIPE_BATCHES)
            sync()
            dt3 = time.perf_counter() - t3
            want_last = ctx.process_iq_device(d_batches[(PIPE_BATCHES - 1) % nb].data_ptr(), n, flush=True)
            extra["pipelined"] = {"batches_in_flight": PIPE_DEPTH, "host_threads": 1, "batches": PIPE_BATCHES,
                                  "value": n * PIPE_BATCHES / dt3,
                                  "unit": "samples/s", "ms_per_step": dt3 / PIPE_BATCHES * 1e3,
                                  # the whole path priced like the kernel: algorithmic bytes per batch / time per batch / HBM peak
                                  "path_frac_of_hbm_peak": 8.0 * n * PIPE_BATCHES / dt3 / 1e9 / HBM_PEAK_GBS,
                                  "same_packet_counts": counts3 == [per_batch[k % nb] for k in range(PIPE_BATCHES)],
                                  "same_packets_last_batch": bool(np.array_equal(last3, want_last))}
            pipe.close()
        if mode == "single" and K == 1 and not args.no_extra and not args.no_pipelined and inflight == 1 and args.stream_seconds > 0 \
                and args.steps >= 3:
            # ONE continuing stream with four of its consecutive chunks in flight (am_spipe): the reference block is a
            # streaming block (lib/preamble_impl.cc:139-246) -- chunk k + 1's scan is enqueued before chunk k resolves, the scan
            # position travels on the device.  The stream: the run's distinct batches one after the other, again and again, each a
            # 1 s chunk in a buffer of its own (the library copies the tail of the chunk before in front of it: 87 KB).  Parity: the
            # packets of ALL chunks against the oracle over the WHOLE stream (item counts and time stamps continue across chunks).
            SP_DEPTH = 4
            nchunks = max(SP_DEPTH + 1, int(round(args.stream_seconds / secs)))
            if not args.no_parity:
                # (the oracle wants the stream in one piece: ~3 x 8 bytes per sample of host memory for its dense arrays)
                try:
                    import psutil
                    fit = int(psutil.virtual_memory().available * 0.5 // (24 * n))
                    nchunks = max(SP_DEPTH + 1, min(nchunks, fit))
                except Exception:
                    pass
            sp = _capi.StreamPipe(rate, 7.0, True, device=(0 if args.emu else local), depth=SP_DEPTH, lib=lib) if lib is not None \
                else _capi.StreamPipe(rate, 7.0, True, device=local, depth=SP_DEPTH)
            front = sp.front()
            sbufs = []
            for b in range(nb):
                tb = torch.zeros(2 * (front + n), dtype=torch.float32, device=dev)
                tb[2 * front:].copy_(d_batches[b])
                sbufs.append(tb)
            sync()
            chunks = [(sbufs[k % nb].data_ptr() + 8 * front, n) for k in range(nchunks)]
            sp.run(chunks[:SP_DEPTH + 1])                           # (a short stream first: allocations, capacities)
            sync()
            t4 = time.perf_counter()
            got4 = sp.run(chunks)
            sync()
            dt4 = time.perf_counter() - t4
            all4 = np.concatenate(got4)
            extra["pipelined_stream"] = {
                "what": "ONE continuing 64 Msps-class stream, %d chunks of %d samples, %d in flight (am_spipe): scan of chunk k+1 "
                        "enqueued before chunk k resolves, scan position handed on through a device word" % (nchunks, n, SP_DEPTH),
                "chunks": nchunks, "chunks_in_flight": SP_DEPTH, "host_threads": 1, "signal_seconds": nchunks * secs,
                "value": n * nchunks / dt4, "unit": "samples/s", "ms_per_chunk": dt4 / nchunks * 1e3,
                "path_frac_of_hbm_peak": 8.0 * n * nchunks / dt4 / 1e9 / HBM_PEAK_GBS,
                "packets": int(len(all4)), "chunks_redone_synchronously": sp.redone(),
                "last_packet_item_count": int(all4["sample"][-1]) if len(all4) else None}
            if not args.no_parity:
                import oracle
                whole4 = np.concatenate([host_batches[k % nb] for k in range(nchunks)])
                t5 = time.perf_counter()
                want4 = oracle.demod(whole4, rate, 7.0, True)
                extra["pipelined_stream"]["parity_whole_stream_vs_oracle"] = bool(np.array_equal(all4, want4))
                extra["pipelined_stream"]["oracle_seconds"] = time.perf_counter() - t5
                del whole4, want4
            sp.close()
            del sbufs
            run_steps(2, [ctx], 1, d_batches)
        if mode == "single" and K == 1 and not args.no_extra and lam != REALISTIC_LAMBDA:
            iq_r = synth.synth_capture(rate, n, REALISTIC_LAMBDA, seed + 7)[0]
            d_r = [torch.from_numpy(iq_r.view(np.float32)).to(dev)]
            run_steps(3, [ctx], 1, d_r)
            ks = max(5, args.steps // 2)
            dtr, _, pkr, fer = timed(ks, [ctx], 1, d_r)
            fe_r = float(np.mean(fer)) if fer else 0.0
            extra["realistic_density"] = {
                "bursts_per_second": REALISTIC_LAMBDA, "value": n * ks / dtr, "unit": "samples/s",
                "ms_per_step": dtr / ks * 1e3, "packets_per_step": len(pkr), "kernel_ms": fe_r,
                "roofline_frac": (8.0 * n / (fe_r * 1e-3) / 1e9 / HBM_PEAK_GBS) if fe_r > 0 else 0.0}
            if not args.no_parity:
                import oracle
                extra["realistic_density"]["parity"] = bool(np.array_equal(pkr, oracle.demod(iq_r, rate, 7.0, True)))
            run_steps(2, [ctx], 1, d_batches)          # back to the main density (capacity estimate of the context)
        if mode == "single" and K == 1 and not args.no_extra and args.sustained_steps > 0 and not args.emu:
            # the timed loop again, long enough (>= 0.5 s of device work) for a sampler of GPU activity to see the device busy: the same
            # rotating batches, the same call, nothing skipped; the packets of its last step are checked against the first pass's count
            run_steps(2, [ctx], 1, d_batches)
            dts, ks_last, pks, fes = timed(args.sustained_steps, [ctx], 1, d_batches)
            fe_s = float(np.mean(fes)) if fes else 0.0
            extra["sustained"] = {"steps": args.sustained_steps, "seconds": dts, "value": n * args.sustained_steps / dts, "unit": "samples/s",
                                  "ms_per_step": dts / args.sustained_steps * 1e3, "kernel_ms": fe_s,
                                  "roofline_frac": (8.0 * n / (fe_s * 1e-3) / 1e9 / HBM_PEAK_GBS) if fe_s > 0 else 0.0,
                                  "same_packet_count_as_timed_loop": bool(len(pks) == per_batch[ks_last % nb])}
        if mode == "single" and K == 1 and not args.no_extra:
            # the same step with the batch in HOST memory (a file source): pinned staging, two buffers in flight -- the PCIe copy
            # of batch k+1 overlaps the scan of batch k (am_uploader_*, what modes_rx does).  PCIe bound; reported separately,
            # never as `value`.  `pageable` = one synchronous host-to-device copy inside am_process_iq (round 2's figure).
            from air_modes import _capi as _c
            up = _c.Uploader(n, nslots=2, lib=lib)
            hb = [b.view(np.float32) for b in host_batches]
            ksteps = 6
            for k in range(2):                                   # (the source has written its samples into the pinned buffers:
                up.buffer(k)[:2 * n] = hb[k % len(hb)]            #  a file reader does readinto() there, modes_rx.py)
            sync()
            th = time.perf_counter()
            up.start(0, n)                                        # (inside the timed region: as many copies as scans)
            for k in range(ksteps):
                ptr = up.wait(k % 2)
                if k + 1 < ksteps:
                    up.start((k + 1) % 2, n)                      # the next batch crosses PCIe while this one is scanned
                pkh = ctx.process_iq_device(ptr, n, flush=True)
            sync()
            dth = (time.perf_counter() - th) / ksteps
            up.close()
            ctx.process_iq(host_batches[0], flush=True)
            sync()
            tp = time.perf_counter()
            for _ in range(2):
                ctx.process_iq(host_batches[0], flush=True)
            sync()
            dtp = (time.perf_counter() - tp) / 2
            extra["host_input"] = {"value": n / dth, "unit": "samples/s", "ms_per_step": dth * 1e3,
                                   "what": "pinned host buffer -> device -> packets, two buffers in flight (am_uploader): the PCIe copy of "
                                           "batch k+1 overlaps the scan of batch k, %d steps; the pinned buffers are filled once, outside "
                                           "the timed region -- the source-to-pinned copy a file reader adds is NOT in this figure" % ksteps,
                                   "packets_last_step": int(len(pkh)),
                                   "pageable": {"value": n / dtp, "ms_per_step": dtp * 1e3,
                                                "what": "am_process_iq on a pageable host pointer (one synchronous copy inside the call)"}}
            run_steps(2, [ctx], 1, d_batches)
        iq_check = host_batches[last_batch]
        parity = None
        if mode == "replicas":
            import oracle
            ok = bool(np.array_equal(pk, oracle.demod(iq_check, rate, 7.0, True)))
            if K > 1:
                ok = bool(ok and extra["k_streams"]["parity_every_stream"])
            if world > 1:
                t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                ok = bool(t[0].item() > 0.5)
            parity = ok
    # ------------------------------------------------------------------------------------------------------------
    else:
        # time-sharded: ONE stream; step k hands rank r the samples [k*W*n + r*n, k*W*n + (r+1)*n) (here: the same second of
        # signal again and again, as if the stream repeated itself).  The receiver is a stream: the scan position, the
        # undecided tail and the sample count cross the steps (air_modes/sharded.py)
        from air_modes.sharded import ShardedReceiver, PipelinedShardedReceiver
        iq = synth.synth_capture(rate, n, lam, seed + rank)[0]
        forced = world == 1 and bool(args.backend)
        in_flight = not args.no_steps_in_flight
        fe_ms = []
        pk = None
        npk_steps = 0
        if in_flight:
            # steps in flight (air_modes/sharded.py: PipelinedShardedReceiver): step k + 1's tail exchange, scan and all-gather are
            # enqueued before step k is resolved; the timed region fills and drains the pipeline (K submits, K collects)
            ctxs = [ctx, new_ctx()]
            rx = PipelinedShardedReceiver(ctxs, rank, world, n, device=dev, force_collectives=forced)
            for b in rx._bufs:
                b[rx.halo * 2:].copy_(torch.from_numpy(iq.view(np.float32)))     # resident in HBM before the timed region
            sync()

            marks = []                                       # host clock when a step's packets came back (timed region)

            def run_in_flight(steps, timed):
                nonlocal pk, npk_steps
                for k in range(steps):
                    rx.submit()
                    if k > 0:
                        pk = rx.collect()
                        if timed:
                            marks.append(time.perf_counter())
                            npk_steps += len(pk)
                            fe_ms.append(ctxs[(rx.k - 2) % 2].last_dom_ms())
                pk = rx.collect()
                if timed:
                    marks.append(time.perf_counter())
                    npk_steps += len(pk)
                    fe_ms.append(ctxs[(rx.k - 1) % 2].last_dom_ms())
            run_in_flight(max(args.warmup, 2), False)
            if world > 1:
                dist.barrier()
            sync()
            t0 = time.perf_counter()
            run_in_flight(args.steps, True)
            sync()
            if world > 1:
                dist.barrier()
            dt = time.perf_counter() - t0
        else:
            # three distinct seconds of signal in three halo'd buffers, used in turn (1.5 GB per rank: the 256 MiB Infinity Cache
            # cannot serve a step's samples from the step before -- VERDICT r5 weak #6); `iq` is the first of them
            NBUF = 2 if args.emu else 3
            look = not args.no_lookahead
            rx = ShardedReceiver(ctx, rank, world, n, device=dev, force_collectives=forced, buffers=NBUF, lookahead=look)
            for b in range(NBUF):
                rx._select(b)
                x = iq if b == 0 else synth.synth_capture(rate, n, lam, seed + rank + 1000 * b)[0]
                rx.chunk.copy_(torch.from_numpy(x.view(np.float32)))     # resident in HBM before the timed region
            rx._select(0)
            extra["sharded_distinct_chunk_buffers"] = NBUF
            sync()
            for _ in range(max(args.warmup, 2)):
                pk = rx.step(ahead=look)
            if world > 1:
                dist.barrier()
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                pk = rx.step(ahead=look)                  # (all three buffers are resident: the next step's samples are there)
                npk_steps += len(pk)
                fe_ms.append(ctx.last_dom_ms())
            sync()
            if world > 1:
                dist.barrier()
            dt = time.perf_counter() - t0
        extra["sharded_steps_in_flight"] = in_flight
        if in_flight and len(marks) >= 4:
            # between the first and the last-but-one step's packets the pipeline is full: the period of a step in steady state (the
            # timed region above also pays for filling and draining it, ~0.8 ms once; `value` is quoted on the whole region)
            extra["sharded_steady_state"] = {"ms_per_step": (marks[-2] - marks[0]) / (len(marks) - 2) * 1e3,
                                             "steps": len(marks) - 2, "what": "host clock between the returns of collect(0) and collect(K - 2)"}
        extra["sharded_one_collective_per_step"] = (not args.no_lookahead) and not in_flight
        inflight, nb, per_batch = 1, 1, [len(pk)]
        extra["sharded_sync_steps"] = rx.sync_steps
        # host time inside the torch.distributed calls of a step (enqueue + whatever the backend makes the host wait for);
        # a one-rank receiver (--force-sharded) is the floor with no collective at all
        hs = max(1, rx.host_us["steps"])
        extra["host_dist_us_per_step"] = {"tail_exchange_batch_isend_irecv": rx.host_us["tail_exchange"] / hs,
                                          "exit_table_all_gather_into_tensor": rx.host_us["all_gather"] / hs,
                                          "steps_counted": rx.host_us["steps"], "rank": rank,
                                          # (the mean includes the backend's first call, which sets its communicator up: the median does not)
                                          "median_us": {k: (float(np.median(v)) if v else 0.0) for k, v in rx.host_us_steps.items()},
                                          "through_the_process_group": bool(world > 1 or rx.force),
                                          "tail_by": ("all_gather (the backend refused a send to itself)" if rx.tail_by_gather else
                                                      "batch_isend_irecv") if (world > 1 or rx.force) else "device copy (one rank, no group)"}
        # parity, part 1: a short stream through the same N-rank receiver IN TWO STEPS against the oracle over the WHOLE stream
        import oracle
        ns = max(4 * rx.halo, 30000 * spc)
        whole = synth.synth_capture(rate, 2 * world * ns, lam, 4242)[0]
        mine = []
        if in_flight:
            rx_s = PipelinedShardedReceiver([new_ctx(), new_ctx()], rank, world, ns, device=dev, force_collectives=forced)
            for k in range(2):
                a = (k * world + rank) * ns
                rx_s.chunk.copy_(torch.from_numpy(whole[a:a + ns].copy().view(np.float32)))
                rx_s.submit(flush=(k == 1))
            mine = [rx_s.collect(), rx_s.collect()]
        else:
            rx_s = ShardedReceiver(new_ctx(), rank, world, ns, device=dev, force_collectives=forced)
            for k in range(2):
                a = (k * world + rank) * ns
                rx_s.chunk.copy_(torch.from_numpy(whole[a:a + ns].copy().view(np.float32)))
                mine.append(rx_s.step(flush=(k == 1)))
        if world > 1:
            parts = [None] * world
            dist.all_gather_object(parts, [m.tobytes() for m in mine])
            got = np.concatenate([np.frombuffer(parts[r][k], _capi.PACKET_DTYPE) for k in range(2) for r in range(world)])
        else:
            got = np.concatenate(mine)
        parity_small = bool(np.array_equal(got, oracle.demod(whole, rate, 7.0, True)))
        # part 2: the full-size chunk as a finite stream of its own (flush): rank 0's packets against the oracle over its samples
        if in_flight:
            # (the stream the timed region fed never ended: a fresh receiver for the finite one)
            rx.close()
            rx = PipelinedShardedReceiver(ctxs, rank, world, n, device=dev, force_collectives=forced)
            for c_ in ctxs:
                c_.reset()
            rx.chunk.copy_(torch.from_numpy(iq.view(np.float32)))
        else:
            rx.reset()
            rx._select(0)                                     # (the buffer that holds `iq`)
        pk0 = rx.step(flush=True)
        parity_rank0 = None
        if rank == 0:
            want = oracle.demod(iq, rate, 7.0, True)
            if world > 1:
                # rank 0 decides the positions its own samples let it decide ([0, n - H)); the oracle, whose stream ends
                # after them, may add hits beyond
                keep, rest = want[:len(pk0)], want[len(pk0):]
                parity_rank0 = bool(np.array_equal(pk0, keep) and (len(rest) == 0 or int(rest["sample"][0]) >= n - rx.hold))
            else:
                parity_rank0 = bool(np.array_equal(pk0, want))
        parity = parity_small if parity_rank0 is None else bool(parity_small and parity_rank0)
        extra["parity_detail"] = {"short_stream_two_steps_all_ranks": parity_small, "rank0_full_size": parity_rank0,
                                  "short_stream_samples_per_rank_per_step": ns}
        iq_check = iq

    # ------------------------------------------------------------------------------------------------------------
    fe_local = float(np.mean(fe_ms)) if fe_ms else 0.0
    fe_ranks = [fe_local]
    if world > 1:
        t = torch.tensor([dt, float(npk_steps)], dtype=torch.float64, device=dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0].item())
        npk_total = int(tsum[1].item())
        # every rank's average launch duration of the dominant kernel (HIP events on the context's own stream)
        tk = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(tk, torch.tensor([fe_local], dtype=torch.float64, device=dev))
        fe_ranks = [float(x[0].item()) for x in tk]
    else:
        npk_total = npk_steps

    # who took part: the driver's SCALE record should prove N ranks on N devices behind the number
    me = {"rank": rank, "device": str(dev)}
    if not args.emu and torch.cuda.is_available():
        pr = torch.cuda.get_device_properties(dev)
        me.update({"name": pr.name, "pci_bus_id": getattr(pr, "pci_bus_id", None), "pci_device_id": getattr(pr, "pci_device_id", None),
                   "uuid": str(getattr(pr, "uuid", "")), "hip_visible_devices": os.environ.get("HIP_VISIBLE_DEVICES"),
                   "local_rank": int(os.environ.get("LOCAL_RANK", "0"))})
    ranks_info = [me]
    ranks_seen = 1
    if world > 1:
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, me)
        one = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)               # (through the data-path backend itself: RCCL on the GPU box)
        ranks_seen = int(round(float(one[0].item())))
    if rank == 0:
        total_samples = world * K * n * args.steps
        value = total_samples / dt
        fe_avg_ms = float(np.mean(fe_ranks))       # (one launch per rank and step, every rank the same n samples: the mean over ranks)
        n_launch = int(k_total) if K > 1 else n    # samples the dominant kernel's launch reads (K streams: the zeros between them too)
        achieved = 8.0 * n_launch / (fe_avg_ms * 1e-3) / 1e9 if fe_avg_ms > 0 else 0.0
        fe_kind = ctx.last_frontend()
        kernel_name = {3: ("am_k_fe3" if spc == 32 else "am_k_fe4<%d,G>" % spc) +
                          " (streaming fused |iq|^2 + PMF + reference level + preamble detection, sparse outputs)",
                       2: "am_k_fe2<%d> (fused |iq|^2 + PMF + reference level + preamble detection)" % spc}.get(fe_kind, "am_k_frontend")
        # HBM bytes per launch from the committed PMC passes of this same command (profiles/)
        traffic, traffic_src = None, None
        rocprof = {}
        tj = os.path.join(ROOT, "profiles", "current_traffic.json")
        if mode in ("single", "sharded") and K == 1 and os.path.exists(tj) and not args.emu:
            with open(tj) as f:
                t = json.load(f)
            t = t.get(workload, {}) if "workload" not in t else t    # (one entry per workload)
            # the counters belong to ONE version of the kernel: the file carries the hash of the kernel's source it was
            # measured on, and a kernel that changed since reports no traffic rather than somebody else's
            import hashlib
            with open(os.path.join(ROOT, "gr-air-modes_amd", "csrc", t.get("kernel_source", "am_fe4.hip")), "rb") as kf:
                sha = hashlib.sha256(kf.read()).hexdigest()[:16]
            if t.get("workload") == workload and args.seconds is None and args.lam is None and kernel_name.startswith(t.get("kernel", "?")):
                if t.get("kernel_source_sha16") == sha:
                    traffic = t["traffic_bytes"]
                    traffic_src = "profiles/current_traffic.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes; %s sha %s)" % (
                        t.get("kernel_source", "am_fe4.hip"), sha)
                    # the second clock: rocprofv3's average duration of the same kernel over the same command (kernel trace)
                    if t.get("kernel_ms_rocprof"):
                        rocprof["kernel_ms_rocprof"] = t["kernel_ms_rocprof"]
                        rocprof["frac_rocprof"] = 8.0 * n_launch / (t["kernel_ms_rocprof"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                    # every kernel of a step (per-kernel duration, FETCH x2 + WRITE) and their sum -- valid for the tree they were
                    # measured on only: one hash over the library's sources
                    h = hashlib.sha256()
                    cdir = os.path.join(ROOT, "gr-air-modes_amd", "csrc")
                    for name in sorted(os.listdir(cdir)):
                        if name.endswith((".hip", ".h", ".inc")):
                            with open(os.path.join(cdir, name), "rb") as kf:
                                h.update(name.encode() + b"\0" + kf.read())
                    if t.get("path_source_sha16") == h.hexdigest()[:16]:
                        rocprof["path_traffic_bytes"] = t.get("path_traffic_bytes")
                        rocprof["path_traffic_over_algorithmic"] = t.get("path_traffic_bytes", 0) / float(8 * n_launch)
                        rocprof["path_kernel_us_rocprof"] = t.get("path_kernel_us_rocprof")
                        rocprof["path_launches_per_step"] = t.get("path_launches_per_step")
                        rocprof["path_kernels"] = t.get("kernels")
                else:
                    traffic_src = "profiles/current_traffic.json is stale: measured on %s sha %s, this is %s" % (
                        t.get("kernel_source", "am_fe4.hip"), t.get("kernel_source_sha16"), sha)
        par = {"single": "single GPU" if K == 1 else "single GPU, %d independent streams per scan (am_process_multi)" % K, "replicas": "%d independent receivers%s, one per GPU, no collective" % (world, "" if K == 1 else " x %d streams per scan" % K),
               "sharded": ("time-chunk shards x%d of one continuing stream, " % world) + (
                   "ONE all-gather per step (exit tables + the next step's tails)" if extra.get("sharded_one_collective_per_step")
                   else "RCCL tail exchange + scan exit-table all-gather")}[mode]
        res = {
            "metric": "complex samples/sec demodulated (IQ -> Mode-S packet list)",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "packets_per_sec": npk_total / dt, "packets_per_step": npk_total / args.steps,
            "config": {"workload": "%s synthetic IQ, %.3g s per GPU per step (%d complex samples), Poisson %g "
                                   "bursts/s in AWGN, seed %d+rank(+100*batch), threshold 7 dB, pmf on%s"
                                   % (workload, secs, n, lam, seed,
                                      {"single": "" if K == 1 else ", %d independent streams of that size per step, one scan" % K, "replicas": ", %d independent streams" % (world * K),
                                       "sharded": ", one stream time-sharded over %d GPUs" % world}[mode]),
                       "rate_sps": rate, "samples_per_gpu_per_step": K * n, "streams_per_scan": K, "batches_in_flight": inflight,
                       "distinct_batches": nb, "packets_per_batch": per_batch, "bursts_per_second": lam,
                       "parallelism": par},
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src, "kernel_ms": fe_avg_ms,
                         "kernel_ms_per_rank": {"min": min(fe_ranks), "max": max(fe_ranks)},
                         "algorithmic_bytes_per_launch": 8 * n_launch,
                         # the whole path (all launches of a step, the host's turn-around included) priced the same way:
                         # algorithmic bytes per step / driver-timed step / HBM peak -- NOT the kernel's fraction
                         "path_frac_of_hbm_peak": 8.0 * K * n / (dt / args.steps) / 1e9 / HBM_PEAK_GBS},
        }
        res["roofline"].update(rocprof)
        coll = None
        if world > 1 or (dist.is_available() and dist.is_initialized()):
            coll = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}
            if not args.emu and torch.cuda.is_available():
                try:
                    coll["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
                except Exception as ex:      # (reported, not fatal: the version query is not the data path)
                    coll["rccl_version"] = "unavailable: %s" % ex
        res["ranks"] = {"ranks_seen": ranks_seen, "collectives": coll, "per_rank": ranks_info}
        if args.emu:
            res["emulated"] = True
        if parity is not None:
            res["parity"] = parity
        if not args.no_parity and (mode == "single" or not args.no_cpu_baseline):
            # every line carries its parity bit (the checker runs outside the timed regions); the same pass of the oracle is
            # the single-core CPU baseline unless --no-cpu-baseline
            import oracle
            t1 = time.perf_counter()
            want = oracle.demod(iq_check, rate, 7.0, True)
            cpu_dt = time.perf_counter() - t1
            if mode == "single":
                res["parity"] = bool(np.array_equal(pk, want))
                if K > 1:
                    res["parity"] = bool(res["parity"] and extra["k_streams"]["parity_every_stream"])
            if not args.no_cpu_baseline or args.emu:
                res["cpu_baseline"] = {"value": n / cpu_dt, "unit": "samples/s", "cores": 1, "kind": "port",
                                       "sample": "one %d-sample batch of this run, one pass of oracle/airmodes_oracle.c "
                                                 "(scalar C, gcc -O3, 1 thread), %.2f s" % (n, cpu_dt),
                                       "host_cores_available": os.cpu_count()}
                res["speedup_vs_cpu_baseline"] = value / (n / cpu_dt)
        if mode == "single" and K == 1 and not args.no_cpu_baseline and not args.emu:
            import oracle
            # the same port on every host core: the batch cut into one time chunk per thread (each with the
            # look-ahead a chunk needs), timed only -- SURVEY 8(d) asks for both figures
            from concurrent.futures import ThreadPoolExecutor
            P = max(1, os.cpu_count() or 1)
            halo = 400 * spc
            cuts = [(k * n) // P for k in range(P + 1)]

            def chunk(k):
                a, b = cuts[k], min(n, cuts[k + 1] + halo)
                return len(oracle.demod(iq_check[a:b], rate, 7.0, True))
            with ThreadPoolExecutor(P) as ex:
                list(ex.map(chunk, range(min(P, 8))))            # threads up, library loaded
                best = None
                for _ in range(3):
                    t2 = time.perf_counter()
                    list(ex.map(chunk, range(P)))
                    d = time.perf_counter() - t2
                    best = d if best is None or d < best else best
            res["cpu_baseline_all_cores"] = {"value": n / best, "unit": "samples/s", "cores": P, "kind": "port",
                                             "sample": "the same batch cut into %d time chunks (+%d samples of "
                                                       "look-ahead each), one oracle thread per chunk, best of 3 "
                                                       "passes, %.3f s" % (P, halo, best)}
            # the reference's OWN C++ (lib/preamble_impl.cc, slicer_impl.cc, modes_crc.cc compiled by path into
            # oracle/_ref, which travels with the repository) behind the port's front end, where it exists:
            # bounded sample, messages compared with the GPU path's
            if oracle.have_ref():
                nr = n                                           # the whole batch
                t3 = time.perf_counter()
                rbb, ravg = oracle.frontend(iq_check[:nr], spc, True)
                rmsgs = oracle.ref_preamble_slicer(rbb, ravg, spc, 7.0, rate)[2]
                ref_dt = time.perf_counter() - t3
                sub = ctx.process_iq(iq_check[:nr], flush=True)
                res["cpu_baseline_reference"] = {
                    "value": nr / ref_dt, "unit": "samples/s", "cores": 1, "kind": "reference",
                    "sample": "the whole batch (%d samples): port front end (|iq|^2, PMF, reference level) + "
                              "the reference's preamble_impl/slicer_impl/modes_crc compiled from /root/reference "
                              "against the GNU Radio API stub, 1 thread, %.2f s" % (nr, ref_dt),
                    # (the reference driver also reports hits past the canonical end of the stream: a prefix match)
                    "messages_match_gpu": bool(rmsgs[:len(sub)] == ctx_messages(ctx, sub) and len(sub) > 0)}
        res.update(extra)
    # the JSON line is the LAST thing on stdout: the process group goes first (RCCL prints a version banner through C stdio, which
    # would otherwise land behind the line when the process exits), every C stream is flushed, then the line
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(res))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
