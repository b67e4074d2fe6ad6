"""hipGraph capture of the *_plan_batch_device calls: everything a call enqueues (counter resets, the kernels, the fork
to / join from the second stream of LinearMpcXY) is capturable once the handle's workspace exists (one eager call first:
the workspace allocation is synchronous).  The replayed graph must give, bit for bit, what the eager call gives -- also
after the inputs were overwritten in place, which is how a control loop uses it."""
import numpy as np
import pytest

from centroidalcontrolcollection_amd import IntrinsicallyStableMpc, LinearMpcXY, LinearMpcZ, LinearMpcZmp
from centroidalcontrolcollection_amd import fixtures as fx
from centroidalcontrolcollection_amd import fixtures_ddp as fd

pytestmark = pytest.mark.gpu


def _capture(launch):
    import torch

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        launch(s)  # eager: workspaces grow here
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        launch(torch.cuda.current_stream())
    return g


@pytest.mark.parametrize("n", [512, 40000])  # static pairing / work queue (a counter reset per launch)
def test_zmp_graph_replay(n):
    import torch

    dev = torch.device("cuda:0")
    mpc = LinearMpcZmp(1.0, 2.0, 0.0625)
    a, b = fx.make_zmp_batch(n, 32, 0.0625, seed=3), fx.make_zmp_batch(n, 32, 0.0625, seed=4)
    x0, zlim = torch.from_numpy(a["x0"]).to(dev), torch.from_numpy(a["zlim"]).to(dev)
    zmp = torch.zeros((n, 2), dtype=torch.float64, device=dev)
    st = torch.zeros((n, 2), dtype=torch.int32, device=dev)
    g = _capture(lambda s: mpc.plan_batch_device(x0, zlim, 0.005, zmp, None, st, s))
    for batch in (b, a, b):
        x0.copy_(torch.from_numpy(batch["x0"]))
        zlim.copy_(torch.from_numpy(batch["zlim"]))
        zmp.zero_()
        g.replay()
        torch.cuda.synchronize()
        eager = mpc.planOnceBatch(batch["x0"], batch["zlim"], 0.005)
        assert np.array_equal(zmp.cpu().numpy(), eager["zmp"]) and np.all((st.cpu().numpy() & 0xff) == 0)


def test_zmp_stage_kernel_graph_replay():
    """KS + the exact kernel behind it (csrc/zmp_stage.inc, N = 100): the counter reset, KS and the list kernel are three
    capturable operations; the hand-over list's ORDER depends on the race for its slots, the answers do not (each QP's
    solve is its own).  A later larger eager call retires KS's workspace and list instead of freeing them."""
    import torch

    dev = torch.device("cuda:0")
    n = 6000  # (12000 QPs: above the batch size KS takes N = 100 from)
    mpc = LinearMpcZmp(1.0, 2.0, 0.02)
    a, b = fx.make_zmp_batch(n, 100, 0.02, seed=3), fx.make_zmp_batch(n, 100, 0.02, seed=4)
    x0, zlim = torch.from_numpy(a["x0"]).to(dev), torch.from_numpy(a["zlim"]).to(dev)
    zmp = torch.zeros((n, 2), dtype=torch.float64, device=dev)
    st = torch.zeros((n, 2), dtype=torch.int32, device=dev)
    g = _capture(lambda s: mpc.plan_batch_device(x0, zlim, 0.005, zmp, None, st, s))
    assert mpc.last_kernel() == "zmp_plan_stage_kernel"
    big = fx.make_zmp_batch(3 * n, 100, 0.02, seed=5)
    mpc.planOnceBatch(big["x0"], big["zlim"], 0.005)  # (grows the workspace and the list: the graph keeps the old ones)
    for batch in (b, a, b):
        x0.copy_(torch.from_numpy(batch["x0"]))
        zlim.copy_(torch.from_numpy(batch["zlim"]))
        zmp.zero_()
        g.replay()
        torch.cuda.synchronize()
        eager = mpc.planOnceBatch(batch["x0"], batch["zlim"], 0.005)
        assert np.array_equal(zmp.cpu().numpy(), eager["zmp"]) and np.all((st.cpu().numpy() & 0xff) == 0)
        assert (st.cpu().numpy() >> 8).max() > 20  # (some QPs went through the list: the exact kernel's pivot counts)


def test_a_zmp_graph_survives_a_later_larger_eager_call():
    """the scheduling buffers of a LinearMpcZmp handle (last call's counts, predicted counts, the order
    made from them) grow with the batch; a hipGraph captured at a smaller size has their addresses baked into its launches.
    The N <= 32 path has no other growable buffer, and the outgrown ones are retired, not freed: the graph still replays to
    the eager answers after larger eager calls on the same handle.  (The classes with a per-batch WORKSPACE -- LinearMpcXY,
    the DDP planners, DdpZmp -- free and re-allocate it when a larger batch arrives: there a graph is valid until the
    handle's next larger call, as include/ccc_amd.h says.)"""
    import torch

    dev = torch.device("cuda:0")
    n, big = 9000, 30000
    mpc = LinearMpcZmp(1.0, 2.0, 0.0625)
    a, c = fx.make_zmp_batch(n, 32, 0.0625, seed=3), fx.make_zmp_batch(big, 32, 0.0625, seed=5)
    x0, zlim = torch.from_numpy(a["x0"]).to(dev), torch.from_numpy(a["zlim"]).to(dev)
    out = torch.zeros((n, 2), dtype=torch.float64, device=dev)
    mpc.plan_batch_device(x0, zlim, 0.005, out, None, None, torch.cuda.current_stream())  # (a call before: a history exists)
    g = _capture(lambda s: mpc.plan_batch_device(x0, zlim, 0.005, out, None, None, s))
    want = mpc.planOnceBatch(a["x0"], a["zlim"], 0.005)["zmp"]
    bx, bz = torch.from_numpy(c["x0"]).to(dev), torch.from_numpy(c["zlim"]).to(dev)
    bo = torch.zeros((big, 2), dtype=torch.float64, device=dev)
    for _ in range(2):
        mpc.plan_batch_device(bx, bz, 0.005, bo, None, None, torch.cuda.current_stream())
    torch.cuda.synchronize()
    for _ in range(3):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)


def test_xy_graph_replay_with_rounds_and_second_stream(monkeypatch):
    import torch

    monkeypatch.setenv("CCC_XY_STREAM", "1")  # the stage-recursion kernel in rounds + the dual kernel beside them
    dev = torch.device("cuda:0")
    n, N = 1500, 20
    mpc = LinearMpcXY(100.0, 0.1, N)
    pa, xa = fd.make_xy_batch(n, N, 0.1, seed=5)
    pb, xb = fd.make_xy_batch(n, N, 0.1, seed=6)
    tp = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in pa.items()}
    tx = torch.from_numpy(xa).to(dev)
    u0 = torch.zeros((n, 16), dtype=torch.float64, device=dev)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    g = _capture(lambda s: mpc.plan_batch_device(tp, tx, u0, status=st, stream=s))
    for prob, x in ((pb, xb), (pa, xa)):
        for k, v in prob.items():
            tp[k].copy_(torch.from_numpy(np.ascontiguousarray(v)))
        tx.copy_(torch.from_numpy(x))
        u0.fill_(-1.0)
        g.replay()
        torch.cuda.synchronize()
        eager = mpc.planOnceBatch(prob, x)
        assert np.array_equal(u0.cpu().numpy(), eager["u0"])
        assert np.all((st.cpu().numpy() & 0xff) == 0) and (st.cpu().numpy() >> 8).max() > 16  # some went to the dual kernel


@pytest.mark.parametrize("path", ["CCC_Z_STREAM", "CCC_Z_TABLEAU"])
def test_z_graph_replay(monkeypatch, path):
    import torch

    monkeypatch.setenv(path, "1")
    dev = torch.device("cuda:0")
    n = 3000
    z = LinearMpcZ(100.0, 0.05, 40)
    a, b = fx.make_z_batch(n, 40, 0.05, seed=1), fx.make_z_batch(n, 40, 0.05, seed=2)
    t = {k: torch.from_numpy(v).to(dev) for k, v in a.items()}
    force = torch.zeros(n, dtype=torch.float64, device=dev)
    g = _capture(lambda s: z.plan_batch_device(t["contact"], t["ref_pos"], t["x0"], force, stream=s))
    for batch in (b, a):
        for k in t:
            t[k].copy_(torch.from_numpy(batch[k]))
        force.fill_(-1.0)
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(force.cpu().numpy(), z.planOnceBatch(batch["contact"], batch["ref_pos"], batch["x0"])["force"])


def test_ism_graph_replay():
    import torch

    dev = torch.device("cuda:0")
    n = 3000
    ism = IntrinsicallyStableMpc(1.0, 2.0, 0.02)
    a, b = fx.make_ism_batch(n, 100, 0.02, seed=1), fx.make_ism_batch(n, 100, 0.02, seed=2)
    t = {k: torch.from_numpy(v).to(dev) for k, v in a.items()}
    zmp = torch.zeros((n, 2), dtype=torch.float64, device=dev)
    g = _capture(lambda s: ism.plan_batch_device(t["init"], t["ref"], 0.005, zmp, stream=s))
    for batch in (b, a):
        for k in t:
            t[k].copy_(torch.from_numpy(batch[k]))
        zmp.fill_(-1.0)
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(zmp.cpu().numpy(), ism.planOnceBatch(batch["init"], batch["ref"], 0.005)["zmp"])


def test_ddp_classes_graph_replay():
    """DdpZmp (HBM-streaming kernel) and DdpCentroidal (one instance per workgroup, a workspace cleared per launch)."""
    import torch

    from centroidalcontrolcollection_amd import DdpCentroidal, DdpZmp

    dev = torch.device("cuda:0")
    n, N = 256, 40
    d = DdpZmp(100.0, 0.02, N)
    d.ddp_solver_.config().max_iter = 5
    a, b = fx.make_ddpzmp_batch(n, N, 0.02, seed=1), fx.make_ddpzmp_batch(n, N, 0.02, seed=2)
    t = {k: torch.from_numpy(v).to(dev) for k, v in a.items()}
    u = torch.zeros((n, N, 3), dtype=torch.float64, device=dev)
    g = _capture(lambda s: d.plan_batch_device(t["ref"], t["x0"], t["u_init"], u, stream=s))
    for batch in (b, a):
        for k in t:
            t[k].copy_(torch.from_numpy(batch[k]))
        u.fill_(-1.0)
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(u.cpu().numpy(), d.planOnceBatch(batch["ref"], batch["x0"], batch["u_init"])["u"])

    n, N = 48, 30
    c = DdpCentroidal(100.0, 0.03, N, DdpCentroidal.WeightParam(running_pos=(1, 1, 10), terminal_pos=(1, 1, 10)))
    c.ddp_solver_.config().max_iter = 4
    pa, xa = fd.make_centroidal_batch(n, N, 0.03, seed=1)
    pb, xb = fd.make_centroidal_batch(n, N, 0.03, seed=2)
    tp = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in pa.items()}
    tx = torch.from_numpy(xa).to(dev)
    out = torch.zeros((n, N, 16), dtype=torch.float64, device=dev)
    g = _capture(lambda s: c.plan_batch_device(tp, tx, out, stream=s))
    for prob, x in ((pb, xb), (pa, xa)):
        for k, v in prob.items():
            tp[k].copy_(torch.from_numpy(np.ascontiguousarray(v)))
        tx.copy_(torch.from_numpy(x))
        out.fill_(-1.0)
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), c.planOnceBatch(prob, x)["u"])


def test_workspace_growth_inside_a_capture_is_refused():
    """A batch larger than any the handle has seen needs hipMalloc / hipFree, which would invalidate the capture: the
    entry point returns CCC_ERR_INVALID_ARGUMENT instead (and the capture itself survives)."""
    import torch

    from centroidalcontrolcollection_amd import _lib

    dev = torch.device("cuda:0")
    n, N = 300, 40
    mpc = LinearMpcZ(100.0, 0.05, N)
    b = fx.make_z_batch(n, N, 0.05, seed=9)
    contact = torch.from_numpy(b["contact"]).to(dev)
    ref = torch.from_numpy(b["ref_pos"]).to(dev)
    x0 = torch.from_numpy(b["x0"]).to(dev)
    force = torch.zeros(n, dtype=torch.float64, device=dev)
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with pytest.raises(_lib.CccError) as err:
        with torch.cuda.graph(g, stream=s):
            mpc.plan_batch_device(contact, ref, x0, force, stream=torch.cuda.current_stream())
    assert err.value.code == _lib.CCC_ERR_INVALID_ARGUMENT and "capture" in str(err.value)
