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
    assert json.loads(lines[0])["n_gpus"] == 1


def test_pmc_traffic_lookup_reads_the_committed_table():
    sys.path.insert(0, ROOT)
    import bench
    nbytes, src = bench.pmc_traffic("gemm_tn_pp_kernel")
    assert src == "profiles/pmc_traffic.json" and 1e8 < nbytes < 1e9
    assert bench.pmc_traffic("no_such_kernel") == (None, None)
