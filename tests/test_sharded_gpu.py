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


def test_device_entry_orders_itself_after_the_callers_stream():
    """The sharded device entry runs on streams of its own; it has to wait for what the caller enqueued before it.
    Inputs produced by a long asynchronous chain of kernels on torch's current stream (they are NOT complete when the
    call is made) and an output buffer whose NaN fill is still queued: the plan must see the finished inputs and its
    results must survive."""
    import torch

    devs = _devices()
    D, m, N, dt = len(devs), 4096, 32, 0.0625
    b = fx.make_zmp_batch(D * m, N, dt, seed=9)
    ref = LinearMpcZmp(1.0, 2.0, dt).planOnceBatch(b["x0"], b["zlim"], 0.005)["zmp"]
    sh = ShardedLinearMpcZmp(1.0, 2.0, dt, devs)
    x0, zl, out = [], [], []
    for r, d in enumerate(devs):
        with torch.cuda.device(d):
            hx = torch.from_numpy(b["x0"][r * m:(r + 1) * m]).to("cuda:%d" % d)
            hz = torch.from_numpy(b["zlim"][r * m:(r + 1) * m]).to("cuda:%d" % d)
            x, z = torch.zeros_like(hx), torch.zeros_like(hz)
            big = torch.zeros((64, 1 << 20), dtype=torch.float64, device="cuda:%d" % d)
            for _ in range(20):  # ~ms of queued work in front of the producers
                big.add_(1.0)
            x.copy_(hx * 1.0)
            z.copy_(hz + 0.0)
            o = torch.empty((D * m, 2), dtype=torch.float64, device="cuda:%d" % d)
            o.fill_(float("nan"))
            x0.append(x), zl.append(z), out.append(o)
    sh.plan_batch_device(x0, zl, 0.005, out)  # (no synchronize in between)
    for t in out:
        assert np.array_equal(t.cpu().numpy(), ref)


def test_xy_and_ddp_shards_with_rccl_all_gather():
    """ccc_xy_sharded_* / ccc_ddp_sharded_*: the classes BASELINE's configs 4 and 5 put on eight GPUs, through the same
    shard group (per-device handle + stream, grouped in-place all-gather of the planned first-step force scales)."""
    import torch

    from centroidalcontrolcollection_amd import DdpSingleRigidBody, LinearMpcXY, fixtures_ddp as fd
    from centroidalcontrolcollection_amd.sharding import ShardedDdp, ShardedLinearMpcXY

    devs = _devices()
    D = len(devs)
    # LinearMpcXY
    m, N = 1024, 20
    prob, x0 = fd.make_xy_batch(D * m, N, 0.1, seed=3)
    mpc = LinearMpcXY(100.0, 0.1, N)
    ref = mpc.planOnceBatch(prob, x0)["u0"]
    sh = ShardedLinearMpcXY(mpc, devs)
    probs = [{k: torch.from_numpy(np.ascontiguousarray(v[r * m:(r + 1) * m])).to("cuda:%d" % d) for k, v in prob.items()}
             for r, d in enumerate(devs)]
    xs = [torch.from_numpy(x0[r * m:(r + 1) * m]).to("cuda:%d" % d) for r, d in enumerate(devs)]
    outs = [torch.full((D * m, 16), float("nan"), dtype=torch.float64, device="cuda:%d" % d) for d in devs]
    st = [torch.zeros(m, dtype=torch.int32, device="cuda:%d" % d) for d in devs]
    sh.plan_batch_device(probs, xs, outs, st)
    for t in outs:
        assert np.array_equal(t.cpu().numpy(), ref)
    # DdpSingleRigidBody
    m, N, dt = 256, 50, 0.03
    prob, x0 = fd.make_centroidal_batch(D * m, N, dt, seed=4, srb=True)
    w = DdpSingleRigidBody.WeightParam(running_pos=(1.0, 1.0, 10.0), running_ori=(0.5,) * 3,
                                       terminal_pos=(1.0, 1.0, 10.0), terminal_ori=(0.5,) * 3)
    d = DdpSingleRigidBody(100.0, dt, N, w)
    d.ddp_solver_.config().max_iter = 8
    ref = d.planOnceBatch(prob, x0)
    sh = ShardedDdp(d, devs)
    probs = [{k: torch.from_numpy(np.ascontiguousarray(v[r * m:(r + 1) * m])).to("cuda:%d" % dv) for k, v in prob.items()}
             for r, dv in enumerate(devs)]
    xs = [torch.from_numpy(x0[r * m:(r + 1) * m]).to("cuda:%d" % dv) for r, dv in enumerate(devs)]
    us = [torch.zeros((m, N, 16), dtype=torch.float64, device="cuda:%d" % dv) for dv in devs]
    u0 = [torch.full((D * m, 16), float("nan"), dtype=torch.float64, device="cuda:%d" % dv) for dv in devs]
    it = [torch.zeros(m, dtype=torch.int32, device="cuda:%d" % dv) for dv in devs]
    sh.plan_batch_device(probs, xs, us, u0, iters=it)
    for t in u0:
        assert np.array_equal(t.cpu().numpy(), ref["u"][:, 0, :])
    assert np.array_equal(torch.cat([t.cpu() for t in us]).numpy(), ref["u"])
    assert np.array_equal(torch.cat([t.cpu() for t in it]).numpy(), ref["iters"])


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


def test_cpp_host_with_device_resident_ddp_shards():
    """examples/sharded_ddp_device.cpp: plain C++ over include/ccc_amd.h + the HIP runtime for its own buffers;
    DdpSingleRigidBody shards resident on every visible device, ccc_ddp_sharded_plan_batch_device, the all-gathered
    first-step force scales on every device bit-identical to the one-device host entry."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "sharded_ddp_device")
    if not os.path.exists(exe):
        import __graft_entry__

        __graft_entry__.build()
    out = subprocess.run([exe, "300"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "differ=0" in out.stdout and "devices=%d" % len(_devices()) in out.stdout
