"""bench.py's own orchestration on CPU: `--gpus N` launches the ranks itself, the N > 1 legs carry a
parity verdict, `--replicas` runs independent receivers.  Kernels = the CPU-fiber build (tests/emu),
collectives = gloo; the line printed is marked "emulated" and is not a measurement."""
import json
import os
import subprocess
import sys

import pytest

import conftest


def run_bench(*extra):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(conftest.ROOT, "bench.py"), "--emu", "--steps", "2", "--warmup", "1",
                          "--no-extra"] + list(extra), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout.decode()[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line(emu_lib):
    d = run_bench("--workload", "20msps", "--seconds", "0.01", "--batches", "2")
    assert d["emulated"] and d["n_gpus"] == 1 and d["parity"] is True
    assert d["config"]["distinct_batches"] == 2 and len(d["config"]["packets_per_batch"]) == 2
    assert d["value"] > 0 and d["roofline"]["bound"] == "hbm" and "cpu_baseline" in d


def test_gpus_2_launches_two_ranks_time_sharded(emu_lib):
    d = run_bench("--gpus", "2", "--workload", "20msps", "--seconds", "0.01", "--no-cpu-baseline")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["parity"] is True and d["parity_detail"]["short_stream_two_steps_all_ranks"] is True
    assert d["parity_detail"]["rank0_full_size"] is True
    assert "time-chunk shards x2" in d["config"]["parallelism"]
    # the N > 1 line carries the roofline of the dominant kernel (it printed 0): every rank's event pair
    r = d["roofline"]
    assert r["frac"] > 0 and r["kernel_ms"] > 0 and 0 < r["kernel_ms_per_rank"]["min"] <= r["kernel_ms_per_rank"]["max"]
    assert d["sharded_sync_steps"] == 0


def test_replicas_two_receivers(emu_lib):
    d = run_bench("--gpus", "2", "--replicas", "--seconds", "0.008", "--no-cpu-baseline")
    assert d["n_gpus"] == 2 and d["parity"] is True
    assert "independent receivers" in d["config"]["parallelism"]
    assert d["config"]["rate_sps"] == 20e6


def test_lambda_knob(emu_lib):
    d = run_bench("--workload", "20msps", "--seconds", "0.01", "--lambda", "500", "--batches", "1")
    assert d["config"]["bursts_per_second"] == 500 and d["parity"] is True


def test_force_sharded_one_rank_line(emu_lib):
    """--force-sharded: one rank through the streaming time-shard receiver; the line carries the dominant kernel's time."""
    d = run_bench("--force-sharded", "--workload", "20msps", "--seconds", "0.01", "--no-cpu-baseline")
    assert d["n_gpus"] == 1 and d["parity"] is True and d["roofline"]["frac"] > 0


def test_replicas_eight_receivers(emu_lib):
    """BASELINE.json configs[4]: eight independent 20 Msps receivers, no collective on the data path; the line proves eight ranks."""
    d = run_bench("--gpus", "8", "--replicas", "--seconds", "0.004", "--no-cpu-baseline")
    assert d["n_gpus"] == 8 and d["parity"] is True and "8 independent receivers" in d["config"]["parallelism"]
    assert d["ranks"]["ranks_seen"] == 8 and sorted(r["rank"] for r in d["ranks"]["per_rank"]) == list(range(8))
    assert d["ranks"]["collectives"]["backend"] == "gloo" and d["ranks"]["collectives"]["world_size"] == 8


def test_gpus_2_line_names_its_ranks_and_times_the_collectives(emu_lib):
    d = run_bench("--gpus", "2", "--workload", "20msps", "--seconds", "0.01", "--no-cpu-baseline", "--no-parity")
    assert d["ranks"]["ranks_seen"] == 2 and len(d["ranks"]["per_rank"]) == 2
    h = d["host_dist_us_per_step"]
    assert h["steps_counted"] >= 2 and h["exit_table_all_gather_into_tensor"] > 0 and h["tail_exchange_batch_isend_irecv"] > 0


def test_k_streams_per_scan_line(emu_lib):
    """--streams K: K receivers' seconds in one scan per step (am_process_multi); value counts all of them, parity is every
    stream's against the oracle."""
    d = run_bench("--workload", "20msps", "--seconds", "0.01", "--streams", "3", "--no-cpu-baseline")
    assert d["parity"] is True and d["config"]["streams_per_scan"] == 3
    k = d["k_streams"]
    assert k["parity_every_stream"] is True and k["same_as_single_stream_call"] is True and len(k["packets_per_stream"]) == 3
    assert d["config"]["samples_per_gpu_per_step"] == 3 * k["samples_per_stream"]
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 8 * k["samples_scanned_per_step"] > 8 * 3 * k["samples_per_stream"]


def test_replicas_with_k_streams_each(emu_lib):
    d = run_bench("--gpus", "2", "--replicas", "--streams", "2", "--seconds", "0.008", "--no-cpu-baseline")
    assert d["n_gpus"] == 2 and d["parity"] is True and d["config"]["streams_per_scan"] == 2
    assert "x 2 streams per scan" in d["config"]["parallelism"] and "4 independent streams" in d["config"]["workload"]


@pytest.mark.parametrize("flags,in_flight,one", [((), True, False), (("--no-steps-in-flight",), False, True),
                                                 (("--no-steps-in-flight", "--no-lookahead"), False, False)])
def test_gpus_2_forms_of_the_sharded_step(emu_lib, flags, in_flight, one):
    """The time-sharded loop's three forms: steps in flight (PipelinedShardedReceiver, the default: two contexts per rank, the scan
    of step k + 1 queued before step k is collected), one step at a time with one collective per step, and with two; same line,
    same parity verdicts."""
    d = run_bench("--gpus", "2", "--workload", "20msps", "--seconds", "0.01", "--no-cpu-baseline", "--steps", "4", *flags)
    assert d["n_gpus"] == 2 and d["sharded_steps_in_flight"] is in_flight and d["sharded_one_collective_per_step"] is one
    assert d["parity"] is True and d["parity_detail"]["short_stream_two_steps_all_ranks"] is True and d["parity_detail"]["rank0_full_size"] is True
    assert d["sharded_sync_steps"] == 0 and d["roofline"]["kernel_ms"] > 0
