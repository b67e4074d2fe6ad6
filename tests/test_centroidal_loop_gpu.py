"""The reference's closed-loop tests of the force-scale planners as ONE device call each (csrc/centroidal_loop.hip):
TestDdpCentroidal.cpp:15-174, TestDdpSingleRigidBody.cpp:15-195, TestLinearMpcXY.cpp:15-152.  The property assertions
of the tests are evaluated from the per-instance statistics the loop keeps; the trajectories are compared with the same
loop driven from the host (GPU planner through the Python mirror, numpy CentroidalSim)."""
import numpy as np
import pytest

from centroidalcontrolcollection_amd import DdpCentroidal, DdpSingleRigidBody, LinearMpcXY
from centroidalcontrolcollection_amd import centroidal_loop as cl
from centroidalcontrolcollection_amd import fixtures_ddp as fd

pytestmark = pytest.mark.gpu

INERTIA = (40.0, 20.0, 10.0)


def _ddp_timeline(n, srb, device=0):
    """The schedule of TestDdpCentroidal.cpp:35-80 / TestDdpSingleRigidBody.cpp:36-87 (epsilon_t = 1e-6)."""
    hy = 0.5 if srb else 0.1
    V0, R0 = fd.contact_from_rect((-0.1, -hy), (0.1, hy))
    V2, R2 = fd.contact_from_rect((0.4, -hy), (0.6, hy))
    K = 5
    seg_end = np.tile(np.array([1.4, 1.6, 2.2, 2.4, 1e30]), (n, 1))
    seg_contact = np.tile(np.array([0, 1, 2, 2, 2], dtype=np.int32), (n, 1))
    seg_ref = np.zeros((n, K, 6))
    seg_ref[:, 0, :3] = [0.0, 0.0, 1.0]
    seg_ref[:, 1, :3] = [0.25, 0.0, 1.2]
    seg_ref[:, 2:, :3] = [0.5, 0.0, 1.0]
    if srb:
        seg_ref[:, 3, 3:] = [0.0, 0.0, 0.3]  # roll reference between 2.2 s and 2.4 s (ZYX order)
    dim = np.tile(np.array([16, 0, 16, 0], dtype=np.int32), (n, 1))
    vert = np.zeros((n, 4, 16, 3))
    ridge = np.zeros((n, 4, 16, 3))
    vert[:, 0], ridge[:, 0], vert[:, 2], ridge[:, 2] = V0, R0, V2, R2
    return cl.ContactTimeline(seg_end, seg_contact, seg_ref, dim, vert, ridge, 1e-6, device)


def _sim_state(n, pos, perturb=0.0, seed=0):
    s = np.zeros((n, 18))
    s[:, :3] = pos
    if perturb:
        s[1:, :3] += perturb * np.random.default_rng(seed).uniform(-1, 1, size=(n - 1, 3))
    return s


@pytest.mark.parametrize("srb", [False, True])
def test_ddp_reference_closed_loop_on_the_device(srb):
    """Both reference loops AS WRITTEN (one iteration per control cycle, TestDdpCentroidal.cpp:116 /
    TestDdpSingleRigidBody.cpp:125) as one device call for the reference instance and five instances that start up to 1 cm off: every one of them
    meets the per-cycle and final assertions (round 4: the warm-start guard, tests/test_oracle_ddp.py)."""
    warm_iter = 1
    import torch

    dev = torch.device("cuda:0")
    n, N, dt = 6, 100, 0.03
    if srb:
        w = DdpSingleRigidBody.WeightParam(running_pos=(1.0, 1.0, 10.0), running_ori=(0.5,) * 3,
                                           terminal_pos=(1.0, 1.0, 10.0), terminal_ori=(0.5,) * 3)
        d = DdpSingleRigidBody(100.0, dt, N, w)
    else:
        d = DdpCentroidal(100.0, dt, N, DdpCentroidal.WeightParam(running_pos=(1.0, 1.0, 10.0), terminal_pos=(1.0, 1.0, 10.0)))
    tl = _ddp_timeline(n, srb)
    # instance 0 = the reference test; the others start up to 1 cm off
    state0 = _sim_state(n, (0.0, 0.0, 1.0), perturb=0.01, seed=3)
    sim = torch.from_numpy(state0).to(dev)
    inertia = torch.from_numpy(np.tile(np.array(INERTIA), (n, 1))).to(dev)
    cycles = 601  # while(t < 3.0) with t += 0.005 in floating point: 601 passes (fixtures_ddp.run_closed_loop_ddp)
    stats = torch.zeros((n, 8), dtype=torch.float64, device=dev)
    log = torch.zeros((cycles, n, 9), dtype=torch.float64, device=dev)
    t_end = cl.ddp_closed_loop(d, tl, inertia, sim, 0.0, 0.005, cycles, 500, warm_iter, disturb_times=(1.0,),
                               disturb_lin=(0.05, 0.05, 0.0), stats=stats, log=log)
    st, fin, lg = stats.cpu().numpy(), sim.cpu().numpy(), log.cpu().numpy()
    assert abs(t_end - 3.005) < 1e-9
    # per-cycle assertions (TestDdpCentroidal.cpp:133-135 / TestDdpSingleRigidBody.cpp:150-153) on the reference instance
    # and on the perturbed ones
    assert st[:, 0].max() < 2.0 and st[:, 2].max() < 2.0
    ref_end = np.array([0.5, 0.0, 1.0])
    # stats[5]: the cycles whose warm start the guard replaced (the deviation from the reference's
    # warm-start semantics is counted) -- SRB 1 to 7 of the 601 cycles on the reference instance, a handful on the others;
    # the centroidal loop: at the kick (it passes either way)
    assert np.all(st[:, 5] == np.round(st[:, 5])) and np.all(st[:, 6:] == 0.0)
    if srb:
        assert 1 <= st[0, 5] <= 7 and st[:, 5].max() <= 12, st[:, 5]
    else:
        assert 1 <= st[0, 5] <= 4 and st[:, 5].max() <= 10, st[:, 5]  # (measured: 2 on the reference instance, 1-6 on the others)
    if srb:
        assert st[:, 1].max() < 1.0 and st[:, 3].max() < 2.0
        assert np.linalg.norm(fin[:, :3] - ref_end, axis=1).max() < 0.1 and np.linalg.norm(fin[:, 3:6], axis=1).max() < 0.1
        assert np.linalg.norm(fin[:, 6:9], axis=1).max() < 0.1 and np.linalg.norm(fin[:, 9:12], axis=1).max() < 0.1
    else:
        assert st[:, 4].max() < 1.0
        assert np.linalg.norm(fin[:, :3] - ref_end, axis=1).max() < 0.1
        assert np.linalg.norm(fin[:, 6:9], axis=1).max() < 0.1 and np.linalg.norm(fin[:, 15:18], axis=1).max() < 0.01
    # the same loop driven from the host for the reference instance: planner through planOnceBatch, numpy simulator
    d.ddp_solver_.config().max_iter = 500
    pos_host = []

    def plan(prob, x0, u_init, max_iter):
        d.ddp_solver_.config().max_iter = max_iter
        return d.planOnceBatch(prob, x0, u_init)["u"]

    hlog, hfin = fd.run_closed_loop_ddp(plan, srb=srb, warm_max_iter=warm_iter)
    hp = np.array([r["pos"] for r in hlog])
    k = 120 if srb else len(hlog)  # the SRB loop amplifies last-bit differences of the wrench sums after the take-off
    assert np.abs(lg[:k, 0, :3] - hp[:k]).max() < 1e-7
    if not srb:
        assert np.abs(fin[0, :3] - hfin["pos"]).max() < 1e-6


def test_xy_reference_closed_loop_on_the_device():
    import torch

    dev = torch.device("cuda:0")
    n, N, dt, mass = 5, 15, 0.1, 100.0
    mpc = LinearMpcXY(mass, dt, N)
    # TestLinearMpcXY.cpp:29-80 (no epsilon in this test)
    rects = [((0.9, -0.15), (1.1, 0.15)), ((0.9, 0.05), (1.1, 0.15)), ((1.15, -0.15), (1.35, -0.05)),
             ((1.4, 0.05), (1.6, 0.15)), ((1.4, -0.15), (1.6, 0.15))]
    refs = [(1.0, 0.0), (1.0, 0.1), (1.25, -0.1), (1.5, 0.1), (1.5, 0.0)]
    K = C = 5
    seg_end = np.tile(np.array([3.0, 4.0, 5.0, 6.0, 1e30]), (n, 1))
    seg_contact = np.tile(np.arange(5, dtype=np.int32), (n, 1))
    seg_ref = np.zeros((n, K, 6))
    dim = np.full((n, C), 16, dtype=np.int32)
    vert, ridge = np.zeros((n, C, 16, 3)), np.zeros((n, C, 16, 3))
    for c in range(5):
        vert[:, c], ridge[:, c] = fd.contact_from_rect(*rects[c])
        seg_ref[:, c, :3] = [refs[c][0], refs[c][1], 1.0]
    tl = cl.ContactTimeline(seg_end, seg_contact, seg_ref, dim, vert, ridge, 0.0)
    state0 = _sim_state(n, (1.0, 0.0, 1.0), perturb=0.01, seed=4)
    state0[:, 2] = 1.0
    sim = torch.from_numpy(state0).to(dev)
    inertia = torch.from_numpy(np.tile(np.array(INERTIA), (n, 1))).to(dev)
    cycles = 160  # while(t < 8.0) with t += 0.05
    stats = torch.zeros((n, 8), dtype=torch.float64, device=dev)
    t_end = cl.xy_closed_loop(mpc, tl, 1.0, inertia, sim, 0.0, 0.05, cycles, stats=stats)
    st, fin = stats.cpu().numpy(), sim.cpu().numpy()
    assert abs(t_end - 8.0) < 1e-6
    # TestLinearMpcXY.cpp:126-128 per cycle, :140-142 at the end
    assert st[:, 0].max() < 2.0 and st[:, 2].max() < 2.0 and st[:, 4].max() < 5.0
    ref_end = np.array([1.5, 0.0, 1.0])
    assert np.linalg.norm(fin[:, :3] - ref_end, axis=1).max() < 0.1
    assert np.linalg.norm(fin[:, 6:9], axis=1).max() < 0.1 and np.linalg.norm(fin[:, 15:18], axis=1).max() < 0.1
    # host-driven loop of the reference instance (planOnce through the mirror, numpy simulator)
    sim_h = fd.CentroidalSim(mass, np.array(INERTIA), 0.05)
    sim_h.pos = np.array([1.0, 0.0, 1.0])
    t = 0.0
    for _ in range(cycles):
        prob = fd.xy_problem(t, N, dt, mass)
        x0 = np.array([[mass * sim_h.pos[0], mass * sim_h.vel[0], mass * sim_h.pos[1], mass * sim_h.vel[1],
                        sim_h.ang_mom[0], sim_h.ang_mom[1]]])
        u0 = mpc.planOnceBatch(prob, x0)["u0"][0]
        moment, force = fd.total_wrench(prob["vertex"][0, 0], prob["ridge"][0, 0], u0[:prob["dim"][0, 0]], sim_h.pos)
        t += 0.05
        sim_h.update(force, moment)
    assert np.abs(fin[0, :3] - sim_h.pos).max() < 1e-6 and np.abs(fin[0, 6:9] - sim_h.vel).max() < 1e-6


# ---------------------------------------------------------------------------------------------------------------------
# Walking with two separate foot contacts in double support: 32-ridge contact entries, the wide DDP kernel / the
# 32-slot XY kernel inside the loop
def _walking_timeline(n, com_z=1.0):
    """DS (L0, R0) -> left stance -> DS (L0, R1) -> right stance -> DS (L1, R1), feet 0.2 x 0.1 at y = +-0.1."""
    def foot(x, y):
        return fd.contact_from_rect((x - 0.1, y - 0.05), (x + 0.1, y + 0.05))

    L0, R0, R1, L1 = foot(0.0, 0.1), foot(0.0, -0.1), foot(0.2, -0.1), foot(0.4, 0.1)
    entries = [[L0, R0], [L0], [L0, R1], [R1], [L1, R1]]
    K = C = 5
    seg_end = np.tile(np.array([0.4, 0.9, 1.3, 1.8, 1e30]), (n, 1))
    seg_contact = np.tile(np.arange(5, dtype=np.int32), (n, 1))
    seg_ref = np.zeros((n, K, 6))
    for c, xy in enumerate([(0.0, 0.0), (0.0, 0.06), (0.1, 0.0), (0.2, -0.06), (0.3, 0.0)]):
        seg_ref[:, c, :3] = [xy[0], xy[1], com_z]
    dim = np.zeros((n, C), dtype=np.int32)
    vert, ridge = np.zeros((n, C, 32, 3)), np.zeros((n, C, 32, 3))
    for c, feet in enumerate(entries):
        r = 0
        for V, R in feet:
            vert[:, c, r:r + 16], ridge[:, c, r:r + 16] = V, R
            r += 16
        dim[:, c] = r
    return seg_end, seg_contact, seg_ref, dim, vert, ridge


def _segment(seg_end, t):
    return min(int(np.searchsorted(seg_end, t, side="right")), len(seg_end) - 1)


def test_ddp_walking_closed_loop_with_double_support_on_the_device():
    import torch

    dev = torch.device("cuda:0")
    n, N, dt, mass, sim_dt, cycles = 4, 30, 0.05, 100.0, 0.005, 500
    seg_end, seg_contact, seg_ref, dim, vert, ridge = _walking_timeline(n)
    tl = cl.ContactTimeline(seg_end, seg_contact, seg_ref, dim, vert, ridge, 1e-6)
    d = DdpCentroidal(mass, dt, N, DdpCentroidal.WeightParam(running_pos=(1.0, 1.0, 10.0), terminal_pos=(1.0, 1.0, 10.0)),
                      max_phases=5, max_ridges=32)
    state0 = _sim_state(n, (0.0, 0.0, 1.0), perturb=0.005, seed=7)
    sim = torch.from_numpy(state0).to(dev)
    inertia = torch.from_numpy(np.tile(np.array(INERTIA), (n, 1))).to(dev)
    stats = torch.zeros((n, 8), dtype=torch.float64, device=dev)
    log = torch.zeros((cycles, n, 9), dtype=torch.float64, device=dev)
    t_end = cl.ddp_closed_loop(d, tl, inertia, sim, 0.0, sim_dt, cycles, 50, 1, stats=stats, log=log)
    st, fin, lg = stats.cpu().numpy(), sim.cpu().numpy(), log.cpu().numpy()
    assert abs(t_end - cycles * sim_dt) < 1e-9
    # the properties the reference asserts on its own scenario (TestDdpCentroidal.cpp:133-135,:152-155), on the walk
    assert st[:, 0].max() < 2.0 and st[:, 2].max() < 2.0 and st[:, 4].max() < 1.0
    assert np.linalg.norm(fin[:, :3] - np.array([0.3, 0.0, 1.0]), axis=1).max() < 0.1
    assert np.linalg.norm(fin[:, 6:9], axis=1).max() < 0.2
    # the planned contact forces carry the robot throughout (vertical force of the logged wrench, transients included)
    assert lg[10:, :, 5].min() > 0.2 * mass * fd.G and lg[10:, :, 5].max() < 2.0 * mass * fd.G
    # the same loop driven from the host for instance 0: planner through planOnceBatch, numpy simulator
    sim_h = fd.CentroidalSim(mass, np.array(INERTIA), sim_dt)
    sim_h.pos = state0[0, :3].copy()
    prob = dict(phase_dim=dim[:1], phase_vertex=vert[:1], phase_ridge=ridge[:1], step_phase=np.zeros((1, N), np.int32),
                ref_pos=np.zeros((1, N + 1, 3)))
    u_prev = dims_prev = None
    t, pos_h = 0.0, []
    for c in range(120):
        segs = [_segment(seg_end[0], t + i * dt + 1e-6) for i in range(N + 1)]
        prob["step_phase"][0] = seg_contact[0][segs[:N]]
        prob["ref_pos"][0] = seg_ref[0][segs, :3]
        dims = dim[0][prob["step_phase"][0]]
        u_init = None
        if u_prev is not None:
            u_init = u_prev.copy()
            u_init[0, dims != dims_prev, :] = 0.0
        d.ddp_solver_.config().max_iter = 50 if c == 0 else 1
        x0 = np.concatenate([sim_h.pos, mass * sim_h.vel, sim_h.ang_mom])[None]
        u = d.planOnceBatch(prob, x0, u_init)["u"]
        u_prev, dims_prev = u, dims
        ph = prob["step_phase"][0, 0]
        moment, force = fd.total_wrench(vert[0, ph], ridge[0, ph], u[0, 0, :dim[0, ph]], sim_h.pos)
        pos_h.append(sim_h.pos.copy())
        t += sim_dt
        sim_h.update(force, moment)
    assert np.abs(lg[:120, 0, :3] - np.array(pos_h)).max() < 1e-7


def test_xy_walking_closed_loop_with_double_support_on_the_device():
    import torch

    dev = torch.device("cuda:0")
    n, N, dt, mass, sim_dt, cycles = 4, 24, 0.1, 100.0, 0.05, 60
    seg_end, seg_contact, seg_ref, dim, vert, ridge = _walking_timeline(n)
    tl = cl.ContactTimeline(seg_end, seg_contact, seg_ref, dim, vert, ridge, 0.0)
    mpc = LinearMpcXY(mass, dt, N, max_ridges=32)
    state0 = _sim_state(n, (0.0, 0.0, 1.0), perturb=0.005, seed=8)
    state0[:, 2] = 1.0
    sim = torch.from_numpy(state0).to(dev)
    inertia = torch.from_numpy(np.tile(np.array(INERTIA), (n, 1))).to(dev)
    stats = torch.zeros((n, 8), dtype=torch.float64, device=dev)
    log = torch.zeros((cycles, n, 9), dtype=torch.float64, device=dev)
    t_end = cl.xy_closed_loop(mpc, tl, 1.0, inertia, sim, 0.0, sim_dt, cycles, stats=stats, log=log)
    st, fin, lg = stats.cpu().numpy(), sim.cpu().numpy(), log.cpu().numpy()
    assert abs(t_end - cycles * sim_dt) < 1e-9
    # TestLinearMpcXY.cpp:126-128 per cycle, :140-142 at the end, on the walk
    assert st[:, 0].max() < 2.0 and st[:, 2].max() < 2.0 and st[:, 4].max() < 5.0
    assert np.linalg.norm(fin[:, :3] - np.array([0.3, 0.0, 1.0]), axis=1).max() < 0.1
    # (Lx, Ly: what this planner controls; the yaw momentum of alternating feet is outside its model)
    assert np.linalg.norm(fin[:, 6:9], axis=1).max() < 0.1 and np.linalg.norm(fin[:, 15:17], axis=1).max() < 0.1
    # host-driven loop of instance 0 (planOnceBatch on the same 32-slot handle, numpy simulator)
    sim_h = fd.CentroidalSim(mass, np.array(INERTIA), sim_dt)
    sim_h.pos = state0[0, :3].copy()
    t, pos_h = 0.0, []
    for _ in range(cycles):
        segs = [_segment(seg_end[0], t + i * dt) for i in range(N)]
        cs = seg_contact[0][segs]
        prob = dict(dim=dim[0][cs][None], vertex=vert[0][cs][None], ridge=ridge[0][cs][None], com_z=np.full((1, N), 1.0),
                    total_force_z=np.full((1, N), mass * fd.G), ref_out=np.zeros((1, N, 6)))
        prob["ref_out"][0, :, 0] = mass * seg_ref[0][segs, 0]
        prob["ref_out"][0, :, 2] = mass * seg_ref[0][segs, 1]
        x0 = np.array([[mass * sim_h.pos[0], mass * sim_h.vel[0], mass * sim_h.pos[1], mass * sim_h.vel[1],
                        sim_h.ang_mom[0], sim_h.ang_mom[1]]])
        u0 = mpc.planOnceBatch(prob, x0)["u0"][0]
        m0 = prob["dim"][0, 0]
        moment, force = fd.total_wrench(prob["vertex"][0, 0], prob["ridge"][0, 0], u0[:m0], sim_h.pos)
        pos_h.append(sim_h.pos.copy())
        t += sim_dt
        sim_h.update(force, moment)
    assert np.abs(lg[:, 0, :3] - np.array(pos_h)).max() < 1e-6
