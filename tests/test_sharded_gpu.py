"""The C-ABI's one-process multi-GPU path (csrc/sharded.hip) on the devices this box has: host arrays sharded over the
device list, and device-resident shards with the RCCL all-gather.  On a one-GPU box the device list is [0] -- the code
path (one handle, stream and communicator per listed device, in-place ncclAllGather) is the one eight GPUs take."""
import numpy as np
import pytest

from centroidalcontrolcollection_amd import LinearMpcZmp
from centroidalcontrolcollection_amd import fixtures as fx
from centroidalcontrolcollection_amd.sharding import ShardedLinearMpcZmp

pytestmark = pytest.mark.gpu


def _devices():
    import torch

    return list(range(torch.cuda.device_count()))


def test_host_arrays_sharded_equal_single_device():
    n, N, dt = 3001, 32, 0.0625
    b = fx.make_zmp_batch(n, N, dt, seed=5)
    ref = LinearMpcZmp(1.0, 2.0, dt).planOnceBatch(b["x0"], b["zlim"], 0.005)
    sh = ShardedLinearMpcZmp(1.0, 2.0, dt, _devices())
    r = sh.planOnceBatch(b["x0"], b["zlim"], 0.005)
    assert np.array_equal(r["zmp"], ref["zmp"]) and np.array_equal(r["pivots"], ref["pivots"])
    assert np.all(r["status"] == 0)


def test_device_shards_with_rccl_all_gather():
    import torch

    devs = _devices()
    D, m, N, dt = len(devs), 2048, 32, 0.0625
    b = fx.make_zmp_batch(D * m, N, dt, seed=6)
    ref = LinearMpcZmp(1.0, 2.0, dt).planOnceBatch(b["x0"], b["zlim"], 0.005)["zmp"]
    sh = ShardedLinearMpcZmp(1.0, 2.0, dt, devs)
    x0 = [torch.from_numpy(b["x0"][r * m:(r + 1) * m]).to("cuda:%d" % d) for r, d in enumerate(devs)]
    zl = [torch.from_numpy(b["zlim"][r * m:(r + 1) * m]).to("cuda:%d" % d) for r, d in enumerate(devs)]
    out = [torch.full((D * m, 2), float("nan"), dtype=torch.float64, device="cuda:%d" % d) for d in devs]
    sh.plan_batch_device(x0, zl, 0.005, out)
    for t in out:  # every device holds the planned ZMPs of every shard
        assert np.array_equal(t.cpu().numpy(), ref)
    assert torch.cuda.current_device() == 0  # the entry points restore the caller's device


def test_duplicate_device_is_refused():
    from centroidalcontrolcollection_amd import _lib

    with pytest.raises(_lib.CccError):
        ShardedLinearMpcZmp(1.0, 2.0, 0.0625, [0, 0])


def test_cpp_host_reaches_every_device_through_the_c_abi():
    """examples/sharded_linear_mpc_zmp.cpp: plain C++ over include/ccc_amd.h, all visible devices, 20000 instances in
    host memory; bit-identical to the one-device entry point."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "sharded_linear_mpc_zmp")
    if not os.path.exists(exe):
        import __graft_entry__

        __graft_entry__.build()
    out = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "differ=0 unsolved=0" in out.stdout and "devices=%d" % len(_devices()) in out.stdout
