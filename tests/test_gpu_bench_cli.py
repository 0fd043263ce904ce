"""GPU: bench.py as the driver launches it.  The N > 1 code path (process group, both partitions, RCCL collectives,
max-over-ranks timing, rank-0 JSON line) is executed under ``python -m torch.distributed.run`` on a ONE-rank RCCL group
(ALLSET_FORCE_COLLECTIVES=1: the very same RCCL calls, the only way to run them on a 1-GPU box), plus the plain N = 1 line
with its roofline / cpu_baseline objects at a small size."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _json_line(stdout: str) -> dict:
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("shard,model", [("rows", "deepsets"), ("columns", "deepsets"), ("auto", "pma")])
def test_bench_under_torchrun_one_rank_rccl_group(shard, model):
    env = dict(os.environ, ALLSET_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--n-per-gpu", "20000", "--shard", shard, "--model", model, "--pipeline-chunks", "2"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    line = _json_line(res.stdout)
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1
    parts = line["partitions"]
    assert set(parts) >= {"rows", "columns"} and sum(bool(parts[k]["is_value"]) for k in ("rows", "columns")) == 1
    primary = "rows" if shard == "rows" else "columns" if shard == "columns" else [k for k in ("rows", "columns") if parts[k]["is_value"]][0]
    assert parts[primary]["is_value"] and line["value"] == parts[primary]["value"]
    nnz = line["config"]["nnz"]
    assert nnz == 20000 * 16
    for k in ("rows", "columns"):
        assert parts[k]["ms_per_step"] > 0
        assert abs(parts[k]["value"] - nnz * 128 / (parts[k]["ms_per_step"] * 1e-3)) <= 1e-6 * parts[k]["value"]
    assert line["roofline"]["kernel"].startswith("allset_") and line["roofline"]["launches"] > 0


def test_bench_default_line_small(tmp_path):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--n-per-gpu", "30000",
                          "--cpu-threads", "8"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    line = _json_line(res.stdout)
    assert line["config"]["parallelism"] == "single GPU" and "partitions" not in line
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["traffic"] is None and rf["traffic_source"] is None          # not the profiled shape
    assert abs(rf["frac_of_copy_ceiling"] - rf["achieved"] / 6300.0) < 1e-12
    assert "segreduce_fwd" in rf["per_kernel"] and any(k.startswith("fused_linear") for k in rf["per_kernel"])
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 8 and cb["value"] > 0 and "the GPU workload itself" in cb["sample"]
    assert abs(line["value"] - line["config"]["nnz"] * 128 / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    # both arithmetics of the dense tail in the same run: the default's numbers are the line's, the strict ones beside them
    dt, strict = line["dense_tail"], line["strict"]
    assert dt["arithmetic"] == "auto" and strict["arithmetic"] == "bf16x6"
    assert dt["strict_ms_per_step"] == strict["dense_tail_ms_per_step"] > 0 and strict["ms_per_step"] > 0
    assert any(k.startswith("fused_linear") for k in strict["kernels"])
    assert line["config"]["workload"].startswith("variant of BASELINE configs[2] (30000 rows per GPU)")
