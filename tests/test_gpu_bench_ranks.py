"""GPU test of bench.py's N > 1 path on a box with one GPU: two and three ranks share device 0
(MZHIP_BENCH_SHARE_GPU=1: gloo on host copies instead of RCCL, which refuses two ranks on one device), so the
sharding of ONE entry table, the agreement on the unique streams, the {crc, status} gather and the max / sum
reductions run on the real kernels.  The line must account for every entry exactly once."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(n, extra):
    env = dict(os.environ, MZHIP_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                 # rank 0 alone prints
    return json.loads(lines[0])


# (8, ...): the rank count of the driver's SCALE run, as a dry run on one device
@pytest.mark.parametrize("n,cfg,entries,unique", [(2, 2, 3001, 256), (4, 2, 5003, 256), (8, 2, 10007, 256), (3, 3, 20000, 512), (4, 3, 30001, 512), (8, 3, 40009, 512),
                                                  (2, 4, 48, 6), (2, 5, 2000, 256)])
def test_strong_scaling_line_accounts_for_every_entry(n, cfg, entries, unique):
    line = _run(n, ["--config", str(cfg), "--entries", str(entries), "--unique", str(unique)])
    assert line["n_gpus"] == n and line["scaling"] == "strong"
    assert line["config"]["entries_total"] == entries and 0 < line["config"]["entries_rank0"] < entries
    assert line["crc32_match_rate"] == 1.0 and line["bytes_spot_check"] is True
    assert line["value"] > 0 and line["roofline"]["frac"] > 0 and "cpu_baseline" not in line
    # every rank reports its own launch, and the ranks built ONE table (LOCAL_RANK 0 compresses, the others load its file)
    assert len(line["roofline"]["kernel_ms_per_rank"]) == n and all(x > 0 for x in line["roofline"]["kernel_ms_per_rank"])
    assert sum(line["config"]["launch"]["entries_per_rank"]) == entries and line["config"]["launch"]["gather_ms"] > 0


def test_weak_scaling_line():
    line = _run(2, ["--config", "2", "--entries", "1500", "--unique", "256", "--scaling", "weak"])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["config"]["entries_total"] == 3000 and line["config"]["entries_rank0"] == 1500
    assert line["crc32_match_rate"] == 1.0


def test_plain_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher around it (WORLD_SIZE unset, the way the driver starts --gpus 1) must
    start two ranks itself, not measure one GPU and call it two."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MZHIP_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--entries", "3001", "--unique", "256"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["entries_total"] == 3001
    assert len(line["roofline"]["kernel_ms_per_rank"]) == 2 and all(x > 0 for x in line["roofline"]["kernel_ms_per_rank"])
    assert sum(line["config"]["launch"]["entries_per_rank"]) == 3001
    assert line["crc32_match_rate"] == 1.0


def test_more_gpus_than_the_box_has_is_an_error_not_a_smaller_run():
    import torch

    n = torch.cuda.device_count() + 7
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MZHIP_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and "error" in json.loads(lines[0]) and "value" not in json.loads(lines[0])
