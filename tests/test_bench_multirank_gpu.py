"""The multi-rank path of bench.py on a single-GPU box: two ranks share cuda:0 and exchange through gloo
(CCC_BENCH_BACKEND=gloo; on the 8-GPU node the driver uses the default RCCL backend, one rank per GPU).  Guards the
launch protocol, the double-buffered all-gather and the JSON contract for the headline and one secondary workload."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(port, extra, backend="gloo"):
    env = dict(os.environ, CCC_BENCH_BACKEND=backend)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_headline_two_ranks():
    d = _run(29611, ["--steps", "6", "--warmup", "2", "--batch", "4096"])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["config"]["collective"] == "all_gather(zmp)" and d["unsolved"] == 0
    assert d["value"] > 0 and abs(d["value"] - 2 * 4096 * 6 / (d["ms_per_step"] * 6e-3)) < 1e-6 * d["value"]
    assert set(["bound", "achieved", "peak", "unit", "frac", "traffic", "valu"]) <= set(d["roofline"])
    assert d["roofline"]["bound"] == "valu" and 0 < d["roofline"]["valu"]["frac"] < 1
    # p50 = pinned host -> planned ZMPs gathered on every rank; the kernel-only median beside it
    assert d["p50_ms"] > d["p50_kernel_ms"] > 0
    # BASELINE's literal configuration (the batch in TOTAL, shards of batch / N) is timed in the same run
    ss = d["strong_scaling"]
    assert ss["total_batch"] == 4096 and ss["batch_per_gpu"] == 2048 and ss["value"] > 0


def test_headline_two_ranks_strong_scaling_flag():
    d = _run(29614, ["--steps", "4", "--warmup", "1", "--batch", "4096", "--scaling", "strong"])
    assert d["scaling"] == "strong" and d["config"]["batch_per_gpu"] == 2048 and d["unsolved"] == 0
    assert abs(d["value"] - 4096 * 4 / (d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]


def test_secondary_workload_two_ranks():
    d = _run(29612, ["--workload", "z", "--steps", "4", "--warmup", "1", "--batch", "2048"])
    assert d["n_gpus"] == 2 and d["unsolved"] == 0 and d["value"] > 0
    assert d["config"]["collective"].startswith("all_gather")


def test_ddpzmp_workload_two_ranks():
    d = _run(29613, ["--workload", "ddpzmp", "--steps", "3", "--warmup", "1", "--batch", "1024"])
    assert d["n_gpus"] == 2 and d["unsolved"] == 0 and d["value"] > 0 and d["mean_iterations"] == 3.0
    assert d["config"]["collective"].startswith("all_gather")


def test_secondary_workload_strong_scaling_flag():
    d = _run(29615, ["--workload", "srb", "--steps", "1", "--warmup", "1", "--batch", "512", "--scaling", "strong"])
    assert d["scaling"] == "strong" and d["config"]["batch_per_gpu"] == 256 and d["config"]["total_batch"] == 512
    assert d["distributed"]["world_size"] == 2 and d["distributed"]["backend"] == "gloo"
    assert d["distributed"]["distinct_gpus"] == 1  # (two ranks share the one GPU of this box)


def test_rccl_backend_two_gpus():
    """The driver's own launch (backend nccl = RCCL, one rank per GPU) at world size 2: needs two GPUs, skips on the
    single-GPU box.  Checks the line's `distributed` object: two ranks on two DISTINCT devices."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU: the RCCL path needs two")
    d = _run(29616, ["--steps", "4", "--warmup", "1", "--batch", "8192"], backend="nccl")
    assert d["n_gpus"] == 2 and d["unsolved"] == 0
    assert d["distributed"] == dict(d["distributed"], backend="nccl", world_size=2, distinct_gpus=2)
