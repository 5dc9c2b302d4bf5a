"""bench.py's launcher logic, no GPU needed: `--gpus N` with no launcher around it and fewer than N GPUs on the node must
print ONE {"error": ...} line and exit non-zero -- never a metric for fewer GPUs than asked for."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    return {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MZHIP_BENCH_SHARE_GPU")}


def test_gpus_n_without_n_gpus_is_an_error_line():
    import torch

    n = torch.cuda.device_count() + 8
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert "error" in line and "value" not in line and str(n) in line["error"]


def test_world_size_must_equal_gpus():
    """a launcher with another world size than --gpus is refused before anything is measured"""
    env = dict(_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert "error" in line and "WORLD_SIZE" in line["error"]
