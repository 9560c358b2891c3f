"""bench.py end to end at a small batch: one JSON line on stdout carrying the contract's keys."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_roofline():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "8", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["value"] > 0 and abs(j["value"] - 8 / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-2
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert key in r, key
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3


@pytest.mark.gpu
def test_bench_stdout_is_the_json_line_alone_when_a_process_group_starts():
    """RCCL writes a version banner to stdout when a communicator is created (and gloo its connection notes): the result
    line must still be the only thing on bench.py's stdout.  A one-rank RCCL group stands for the multi-GPU launch."""
    env = dict(os.environ, MMAE_DP_SINGLE_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "8", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-roofline"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[:2000]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1
    # the exchange describes itself: backend, the communicator's own rank count, buckets and bytes, the un-hidden wait
    c = j["comm"]
    assert c["backend"] == "torch-nccl" and c["ranks_seen"] == 1 and c["requested"] == "torch" and c["fallback_reason"] is None
    assert c["buckets"] == len(c["bucket_bytes"]) > 3 and c["bytes_per_step"] == sum(c["bucket_bytes"]) > 3.5e8
    assert c["eager_optimizer"] is True and c["exposed_ms_per_step"] >= 0.0
    assert "gradient exchange: torch-nccl over 1 rank(s)" in out.stderr


@pytest.mark.gpu
def test_bench_comm_flag_selects_the_native_communicator_and_times_its_collectives():
    env = dict(os.environ, MMAE_DP_SINGLE_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29543")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "8", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--comm", "native"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[:2000]
    c = json.loads(lines[0])["comm"]
    assert c["backend"] == "native" and c["requested"] == "native" and c["ranks_seen"] == 1 and c["eager_optimizer"] is True
    assert len(c["allreduce_ms_per_bucket"]) == c["buckets"] and c["allreduce_ms_per_step"] > 0.0


def test_pmc_traffic_lookup_reads_the_committed_table():
    sys.path.insert(0, ROOT)
    import bench
    nbytes, src = bench.pmc_traffic("gemm_tn_pp_kernel")
    assert src == "profiles/pmc_traffic.json" and 1e8 < nbytes < 1e9
    assert bench.pmc_traffic("no_such_kernel") == (None, None)
