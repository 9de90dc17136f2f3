"""bench.py's output contract, single rank and the N > 1 launch path (torch.distributed.run, 127.0.0.1)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _last_json(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_single_rank_line(gpu_required):
    r = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--spectra", "3000", "--proteins", "400",
                        "--cpu-sample", "512"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    for k in REQUIRED:
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["unit"] == "spectra/s" and j["value"] > 0
    assert j["scaling"] == "weak" and j["vs_baseline"] is None and j["higher_is_better"] is True and j["data"] == "synthetic"
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "identical" in cb["parity"]
    assert "workload" in j["config"]


@pytest.mark.gpu
def test_bench_two_ranks_through_torchrun(gpu_required):
    """The driver's N > 1 launch line with two ranks.  A 1-GPU box cannot give each rank its own device or run RCCL between
    them, so the rehearsal backend (gloo; ranks share the device) stands in: everything else — env parsing, per-rank
    workloads, barrier, max-over-ranks, whole-job aggregate, rank-0-only output — is the code the real run executes."""
    env = dict(os.environ, SAGE_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--spectra", "2000",
           "--proteins", "300"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["cpu_baseline"] is None
    # weak scaling: both ranks scored their own batch, the aggregate counts both
    per_rank = j["config"]["spectra_per_gpu"]
    assert 1500 < per_rank <= 2000
    assert abs(j["value"] - 2 * per_rank * 3 / (j["ms_per_step"] * 3 / 1000.0)) / j["value"] < 0.05
