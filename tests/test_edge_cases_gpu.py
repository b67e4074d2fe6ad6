"""GPU edge cases across the kernels: smallest and largest supported horizons, empty problems, degenerate inputs --
each against the CPU oracle."""
import numpy as np
import pytest

from centroidalcontrolcollection_amd import IntrinsicallyStableMpc, LinearMpcXY, LinearMpcZ, LinearMpcZmp
from centroidalcontrolcollection_amd import _lib
from centroidalcontrolcollection_amd._lib import CccError
from centroidalcontrolcollection_amd import fixtures as fx
from centroidalcontrolcollection_amd import fixtures_ddp as fd

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import oracle

    return oracle


def test_xy_single_step_and_no_contact_at_all():
    mass = 100.0
    for N in (1, 2):
        prob, x0 = fd.make_xy_batch(16, N, 0.1, mass, seed=N)
        o = _oracle().LinearMpcXY(mass, 0.1, N).plan_batch(prob, x0)
        r = LinearMpcXY(mass, 0.1, N).planOnceBatch(prob, x0)
        assert np.all(r["status"] == 0) and np.all(o["status"] == 0)
        assert np.abs(r["u0"] - o["u0"]).max() <= 1e-7 * (1.0 + np.abs(o["u0"]).max())
    prob, x0 = fd.make_xy_batch(4, 6, 0.1, mass, seed=3)
    prob["dim"][:] = 0
    r = LinearMpcXY(mass, 0.1, 6).planOnceBatch(prob, x0, want_all=True)
    assert np.all(r["status"] == 0) and np.all(r["u0"] == 0.0) and np.all(r["lam"] == 0.0) and np.all(r["pivots"] == 0)


def test_xy_rejects_unsupported_sizes():
    LinearMpcXY(100.0, 0.1, 21)  # (beyond the dual kernel's 20 steps: the stage-recursion kernel alone, tests/test_xy_gpu.py)
    with pytest.raises(CccError):
        LinearMpcXY(100.0, 0.1, 257)
    with pytest.raises(CccError):
        LinearMpcXY(100.0, 0.1, 20, max_ridges=48)
    with pytest.raises(CccError):
        LinearMpcXY(-1.0, 0.1, 10)


def test_ism_largest_and_smallest_horizon():
    for T, dt in ((2.54, 0.02), (0.02, 0.02), (0.06, 0.02), (2.56, 0.02)):
        o = _oracle().IntrinsicallyStableMpc(1.0, T, dt)
        N = o.horizon_steps
        assert N in (127, 1, 3, 128)  # (127: the largest the tridiagonal kernel takes; 128: the first of the tableau alone)
        b = fx.make_ism_batch(24, N, dt, seed=N)
        ro = o.plan_batch(b["init"], b["ref"], 0.005, want_vel=False, nthreads=8)
        r = IntrinsicallyStableMpc(1.0, T, dt).planOnceBatch(b["init"], b["ref"], 0.005)
        ok = ro["status"] == 0
        assert ok.sum() >= 12 and np.all(r["status"][ok] == 0)
        assert np.abs(r["zmp"][ok] - ro["zmp"][ok]).max() <= 1e-9
        assert np.all(r["status"][~ok].max(axis=1) != 0) if (~ok).any() else True
    with pytest.raises(CccError):
        IntrinsicallyStableMpc(1.0, 3.84, 0.02)  # 192 steps + the stability row exceed the LDS-resident tableau


def test_z_single_step_and_all_flight():
    mass = 100.0
    b = fx.make_z_batch(32, 1, 0.05, seed=1)
    o = _oracle().LinearMpcZ(mass, 0.05, 1).plan_batch(b["contact"], b["ref_pos"], b["x0"])
    r = LinearMpcZ(mass, 0.05, 1).planOnceBatch(b["contact"], b["ref_pos"], b["x0"])
    assert np.all(r["status"] == 0)
    assert (np.abs(r["force"] - o["force"]) / (np.abs(o["force"]) + 1.0)).max() <= 1e-8
    b = fx.make_z_batch(8, 40, 0.05, seed=2)
    b["contact"][:] = 0
    r = LinearMpcZ(mass, 0.05, 40).planOnceBatch(b["contact"], b["ref_pos"], b["x0"], want_all=True)
    assert np.all(r["force"] == 0.0) and np.all(r["force_all"] == 0.0) and np.all(r["status"] == 0)
    LinearMpcZ(mass, 0.05, 65)  # (beyond the tableau kernel's 64 steps: the streaming kernel alone, tests/test_z_gpu.py)
    with pytest.raises(CccError):
        LinearMpcZ(mass, 0.05, 257)


def test_zmp_timeline_without_footsteps_and_single_instance_loop():
    """K = 0 footsteps: constant double-support limits; the closed loop holds the CoM inside them."""
    import torch

    mpc = LinearMpcZmp(1.0, 2.0, 0.0625)
    tl = dict(foot0=torch.tensor([[[0.0, 0.1], [0.0, -0.1]]], dtype=torch.float64, device="cuda:0"),
              foot_pos=torch.zeros((1, 0, 2), dtype=torch.float64, device="cuda:0"),
              foot_id=torch.zeros((1, 0), dtype=torch.int32, device="cuda:0"),
              swing_start=torch.zeros((1, 0), dtype=torch.float64, device="cuda:0"),
              swing_end=torch.zeros((1, 0), dtype=torch.float64, device="cuda:0"))
    zl = torch.zeros((1, 2, 2, 32), dtype=torch.float64, device="cuda:0")
    mpc.sample_limits_device(tl, zl, t_common=1.0)
    torch.cuda.synchronize()
    z = zl.cpu().numpy()[0]
    assert np.all(z[0, 0] == -0.05) and np.all(z[0, 1] == 0.05) and np.all(z[1, 0] == -0.125) and np.all(z[1, 1] == 0.125)
    com = torch.tensor([[[0.02, 0.05], [-0.05, 0.0]]], dtype=torch.float64, device="cuda:0")  # capture point inside
    zmp = com[:, :, 0].clone().contiguous()
    viol = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    mpc.closed_loop_device(tl, com, zmp, 0.0, 0.01, 300, violations=viol)
    torch.cuda.synchronize()
    assert viol.item() == 0
    c = com.cpu().numpy()[0]
    assert abs(c[0, 0]) < 0.05 and abs(c[1, 0]) < 0.125 and np.abs(c[:, 1]).max() < 0.05


@pytest.mark.parametrize("N", [40, 100, 128, 200])
def test_zmp_packed_tableau_edge_instances(N):
    """The packed-tableau kernel (32 < N <= 200) on the instances the N = 32 tests cover: nothing active (zero pivots,
    zero jerk), every row active (N pivots), an infeasible axis (flagged, the other instances untouched), ragged batch
    sizes, determinism."""
    dt = 2.0 / N
    mpc = LinearMpcZmp(1.0, 2.0, dt)
    o = _oracle().LinearMpcZmp(1.0, 2.0, dt)
    x0 = np.zeros((2, 2, 3))
    zlim = np.empty((2, 2, 2, N))
    zlim[0, :, 0], zlim[0, :, 1] = -10.0, 10.0
    zlim[1, :, 0], zlim[1, :, 1] = 0.02, 0.02 + 1e-9
    r = mpc.planOnceBatch(x0, zlim, 0.005, want_jerk=True)
    ref = o.plan_batch(x0, zlim, 0.005)
    assert np.all(r["status"] == 0) and np.all(r["jerk"][0] == 0.0) and np.all(r["pivots"][0] == 0)
    assert np.all(r["pivots"][1] >= N)
    assert np.abs(r["zmp"] - ref["zmp"]).max() <= 1e-9
    b = fx.make_zmp_batch(67, N, dt, seed=N)
    full = mpc.planOnceBatch(b["x0"], b["zlim"], 0.005)
    for n in (1, 2, 63, 67):
        part = mpc.planOnceBatch(b["x0"][:n], b["zlim"][:n], 0.005)
        assert np.array_equal(part["zmp"], full["zmp"][:n])
    bad = b["zlim"].copy()
    bad[5, 0, 0, N // 2] = bad[5, 0, 1, N // 2] + 0.3
    rb = mpc.planOnceBatch(b["x0"], bad, 0.005)
    assert rb["status"][5, 0] == _lib.CCC_STATUS_INFEASIBLE and rb["status"].sum() == _lib.CCC_STATUS_INFEASIBLE
    keep = np.ones((67, 2), bool)
    keep[5, 0] = False
    assert np.array_equal(rb["zmp"][keep], full["zmp"][keep])


def test_ddp_out_of_range_phase_indices_are_clamped_not_followed():
    """step_phase / phase_dim come from the caller: entries outside the tables are clamped (phase to [0, P), ridge count
    to [0, M]) instead of indexing past them -- the answers equal those of the clamped inputs, on both builds."""
    from centroidalcontrolcollection_amd import DdpCentroidal
    from centroidalcontrolcollection_amd import fixtures_ddp as fd

    for kw, gen in ((dict(), lambda: fd.make_centroidal_batch(8, 30, 0.05, seed=4)),
                    (dict(max_ridges=32), lambda: fd.make_walking_batch(8, 30, 0.05, seed=4))):
        prob, x0 = gen()
        P = prob["phase_dim"].shape[1]
        d = DdpCentroidal(100.0, 0.05, 30, DdpCentroidal.WeightParam(running_pos=(1, 1, 10), terminal_pos=(1, 1, 10)),
                          max_phases=P, **kw)
        d.ddp_solver_.config().max_iter = 5
        good = d.planOnceBatch(prob, x0)
        bad = {k: v.copy() for k, v in prob.items()}
        last = prob["step_phase"] == P - 1
        first = prob["step_phase"] == 0
        bad["step_phase"][last] = P + 1000
        bad["step_phase"][first] = -7
        M = prob["phase_vertex"].shape[2]
        full = prob["phase_dim"] == M
        bad["phase_dim"][full] = M + 99
        r = d.planOnceBatch(bad, x0)
        assert np.array_equal(r["u"], good["u"]) and np.array_equal(r["iters"], good["iters"])
