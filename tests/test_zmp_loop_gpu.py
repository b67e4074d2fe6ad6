"""GPU tests of the device-side steps either side of LinearMpcZmp::planOnce (csrc/zmp_loop.hip): reference sampling
(footstep timelines -> ZMP-limit sequences) and the batched closed loop of TestLinearMpcZmp.cpp:55-102."""
import numpy as np
import pytest

from centroidalcontrolcollection_amd import LinearMpcZmp, InitialParam
from centroidalcontrolcollection_amd import fixtures as fx

pytestmark = pytest.mark.gpu


def _to_dev(tl):
    import torch

    return {k: torch.from_numpy(np.ascontiguousarray(v)).to("cuda:0") for k, v in tl.items()}


def test_sampled_limits_are_bit_identical_to_the_host_fixture():
    import torch

    n, N, dt = 3000, 32, 0.0625
    tl = fx.make_zmp_timelines(n, seed=3)
    rng = np.random.default_rng(1)
    t_eval = rng.uniform(0.0, 8.0, size=n)
    times = t_eval[:, None] + dt * np.arange(N)[None, :] + 2e-6
    zmin, zmax = fx.zmp_limits_timeline(tl["foot0"], tl["foot_pos"], tl["foot_id"], tl["swing_start"], tl["swing_end"], times)
    want = np.empty((n, 2, 2, N))
    want[:, :, 0, :] = np.transpose(zmin, (0, 2, 1))
    want[:, :, 1, :] = np.transpose(zmax, (0, 2, 1))
    mpc = LinearMpcZmp(1.0, 2.0, dt)
    zl = torch.zeros((n, 2, 2, N), dtype=torch.float64, device="cuda:0")
    mpc.sample_limits_device(_to_dev(tl), zl, t_eval=torch.from_numpy(t_eval).to("cuda:0"))
    torch.cuda.synchronize()
    assert np.array_equal(zl.cpu().numpy(), want)
    # common time for every instance
    times = 2.5 + dt * np.arange(N)[None, :].repeat(n, axis=0) + 2e-6
    zmin, zmax = fx.zmp_limits_timeline(tl["foot0"], tl["foot_pos"], tl["foot_id"], tl["swing_start"], tl["swing_end"], times)
    mpc.sample_limits_device(_to_dev(tl), zl, t_common=2.5)
    torch.cuda.synchronize()
    assert np.array_equal(zl.cpu().numpy()[:, :, 0, :], np.transpose(zmin, (0, 2, 1)))
    assert np.array_equal(zl.cpu().numpy()[:, :, 1, :], np.transpose(zmax, (0, 2, 1)))


def test_reference_closed_loop_entirely_on_the_device():
    """The scenario of TestLinearMpcZmp.cpp (2 s horizon @ 20 ms, 10 s at 5 ms, two disturbances) as ONE call: planned ZMP
    inside the limits in every cycle, final CoM inside, and the same trajectory as the host-driven loop through
    planOnce (same kernel; the host loop samples with FootstepManager and simulates in numpy)."""
    import torch

    mpc = LinearMpcZmp(1.0, 2.0, 0.02)
    tl = _to_dev(fx.reference_scenario_timeline())
    com = torch.zeros((1, 2, 2), dtype=torch.float64, device="cuda:0")
    zmp = torch.zeros((1, 2), dtype=torch.float64, device="cuda:0")
    viol = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    cycles = 2001
    tc = torch.zeros((cycles, 1, 2), dtype=torch.float64, device="cuda:0")
    tz = torch.zeros_like(tc)
    t_end = mpc.closed_loop_device(tl, com, zmp, 0.0, 0.005, cycles, disturb_times=(4.5, 8.5), disturb_impulse=0.05,
                                   violations=viol, traj_com=tc, traj_zmp=tz)
    torch.cuda.synchronize()
    assert viol.item() == 0
    log, fin = fx.run_closed_loop(lambda f, ip, t, sdt: mpc.planOnce(f, ip, t, sdt))
    assert len(log) in (2000, 2001)
    m = len(log)
    hz = np.array([r["zmp"] for r in log])
    hc = np.array([r["com"] for r in log])
    assert np.abs(tz.cpu().numpy()[:m, 0] - hz).max() <= 1e-9
    assert np.abs(tc.cpu().numpy()[:m, 0] - hc).max() <= 1e-9
    c = com.cpu().numpy()[0, :, 0]
    if m == cycles:
        assert abs(t_end - fin["t"]) < 1e-12
        assert np.all(c - fin["zmin"] >= 0) and np.all(fin["zmax"] - c >= 0)  # TestLinearMpcZmp.cpp:108-109


def test_batched_closed_loop_matches_per_instance_host_loops():
    """Random timelines and states: the device loop against a host loop that samples with the numpy fixture, plans with
    the CPU oracle and simulates in numpy (planned ZMP within 1e-9 in every cycle)."""
    import torch
    from oracle import oracle

    n, N, dt, sim_dt, cycles = 6, 32, 0.0625, 0.01, 150
    tl = fx.make_zmp_timelines(n, seed=9)
    rng = np.random.default_rng(4)
    com0 = np.zeros((n, 2, 2))
    com0[:, :, 0] = rng.uniform(-0.02, 0.02, size=(n, 2))
    com0[:, :, 1] = rng.uniform(-0.05, 0.05, size=(n, 2))
    zmp0 = com0[:, :, 0].copy()
    mpc = LinearMpcZmp(1.0, 2.0, dt)
    com = torch.from_numpy(com0.copy()).to("cuda:0")
    zmp = torch.from_numpy(zmp0.copy()).to("cuda:0")
    tz = torch.zeros((cycles, n, 2), dtype=torch.float64, device="cuda:0")
    viol = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    mpc.closed_loop_device(_to_dev(tl), com, zmp, 0.3, sim_dt, cycles, disturb_times=(0.9,), disturb_impulse=0.03,
                           violations=viol, traj_zmp=tz)
    torch.cuda.synchronize()
    o = oracle.LinearMpcZmp(1.0, 2.0, dt)
    sims = []
    for k in range(n):
        s = fx.ComZmpSim2d(1.0, sim_dt)
        s.x, s.y = com0[k, 0].copy(), com0[k, 1].copy()
        sims.append(s)
    planned = zmp0.copy()
    t = 0.3
    worst = 0.0
    for c in range(cycles):
        times = (t + dt * np.arange(N))[None, :].repeat(n, axis=0) + 2e-6
        zmin, zmax = fx.zmp_limits_timeline(tl["foot0"], tl["foot_pos"], tl["foot_id"], tl["swing_start"], tl["swing_end"], times)
        zlim = np.empty((n, 2, 2, N))
        zlim[:, :, 0, :] = np.transpose(zmin, (0, 2, 1))
        zlim[:, :, 1, :] = np.transpose(zmax, (0, 2, 1))
        pos = np.array([s.pos() for s in sims])
        vel = np.array([s.vel() for s in sims])
        x0 = np.stack([pos, vel, fx.G / 1.0 * (pos - planned)], axis=2)
        planned = o.plan_batch(x0, zlim, sim_dt, want_jerk=False)["zmp"]
        worst = max(worst, np.abs(tz[c].cpu().numpy() - planned).max())
        t += sim_dt
        for k, s in enumerate(sims):
            s.update(planned[k])
            if 0.9 <= t < 0.9 + sim_dt:
                s.addDisturb((0.03, 0.03))
    assert worst <= 1e-9
    assert np.abs(com.cpu().numpy()[:, :, 0] - np.array([s.pos() for s in sims])).max() <= 1e-9


def test_closed_loop_at_the_reference_horizon_through_the_state_space_kernel(monkeypatch):
    """TestLinearMpcZmp's own horizon (2 s @ 20 ms = 100 steps) on 6000 random timelines, 40 control cycles of 5 ms with a
    kick: a batch that size takes the state-space kernel (csrc/zmp_stage.inc) with the exact kernel on its hand-overs; the
    trajectories are those of a handle held to the exact kernels alone (planned ZMP of every cycle within 1e-9, no limit
    violated that the other does not violate)."""
    import torch

    n, dt, sim_dt, cycles = 6000, 0.02, 0.005, 40
    tl = fx.make_zmp_timelines(n, seed=21)
    rng = np.random.default_rng(8)
    com0 = np.zeros((n, 2, 2))
    com0[:, :, 0] = rng.uniform(-0.02, 0.02, size=(n, 2))
    com0[:, :, 1] = rng.uniform(-0.05, 0.05, size=(n, 2))
    out = {}
    for stage in ("1", "0"):
        monkeypatch.setenv("CCC_ZMP_STAGE", stage)
        mpc = LinearMpcZmp(1.0, 2.0, dt)
        monkeypatch.delenv("CCC_ZMP_STAGE")
        com = torch.from_numpy(com0.copy()).to("cuda:0")
        zmp = torch.from_numpy(com0[:, :, 0].copy()).to("cuda:0")
        tz = torch.zeros((cycles, n, 2), dtype=torch.float64, device="cuda:0")
        viol = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        mpc.closed_loop_device(_to_dev(tl), com, zmp, 0.3, sim_dt, cycles, disturb_times=(0.4,), disturb_impulse=0.03,
                               violations=viol, traj_zmp=tz)
        torch.cuda.synchronize()
        out[stage] = (tz.cpu().numpy(), com.cpu().numpy(), viol.cpu().numpy(), mpc.last_kernel())
    assert out["1"][3] == "zmp_plan_stage_kernel" and out["0"][3] == "zmp_plan_reg_kernel"
    assert np.abs(out["1"][0] - out["0"][0]).max() <= 1e-9
    assert np.abs(out["1"][1] - out["0"][1]).max() <= 1e-8
    assert np.array_equal(out["1"][2] > 0, out["0"][2] > 0)
