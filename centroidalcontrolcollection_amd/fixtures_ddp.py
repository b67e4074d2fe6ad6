"""Input generators for the DDP paths (DdpCentroidal / DdpSingleRigidBody): restated reference fixtures + synthetic batches.

Restates (numpy, host side -- these produce INPUTS, they are not on the timed path):
  /root/reference/tests/src/ContactManager.h:10-21          makeContactFromRect (4 vertices, mu = 0.5)
  ForceColl::SurfaceContact / FrictionPyramid (EXTERNAL, not in the reference tree; convention fixed here, see
      `friction_pyramid`: 4 ridges per vertex, normalize([mu cos th, mu sin th, 1]), th = 2 pi i / 4 -- SURVEY.md B.2)
  ForceColl::calcTotalWrench (EXTERNAL; [moment; force] about an origin, tests/src/TestDdpCentroidal.cpp:119-121)
  /root/reference/tests/src/SimModels.h:233-340             CentroidalSim (exact ZOH of the 18-state linear model)
  /root/reference/tests/src/TestDdpCentroidal.cpp:35-80     the contact / reference schedule of the closed-loop test
and the synthetic workloads of SURVEY.md section 8(d) (`make_centroidal_batch`, `make_srb_batch`).

Flattened problem layout (shared by oracle and C-ABI), per instance:
  phase_dim [P] i32, phase_vertex [P,M,3], phase_ridge [P,M,3], step_phase [N] i32, ref_pos [N+1,3]
  (+ ref_ori [N+1,3], inertia [3,3] for the single-rigid-body model).
"""
import math

import numpy as np

G = 9.80665


def friction_pyramid(mu=0.5, ridge_num=4):
    """Ridge directions of one vertex in the contact frame (z = normal)."""
    out = []
    for i in range(ridge_num):
        th = 2.0 * math.pi * i / ridge_num
        v = np.array([mu * math.cos(th), mu * math.sin(th), 1.0])
        out.append(v / np.linalg.norm(v))
    return np.array(out)


def contact_from_rect(rect_min, rect_max, mu=0.5):
    """ContactManager.h:10-21: vertices (min,min),(min.x,max.y),(max,max),(max.x,min.y) at z=0, identity pose.
    Returns (vertex [16,3], ridge [16,3]) in contact -> vertex -> ridge order (DdpCentroidal.cpp:49-60)."""
    verts = [(rect_min[0], rect_min[1], 0.0), (rect_min[0], rect_max[1], 0.0), (rect_max[0], rect_max[1], 0.0),
             (rect_max[0], rect_min[1], 0.0)]
    pyr = friction_pyramid(mu)
    V, R = [], []
    for v in verts:
        for r in pyr:
            V.append(v)
            R.append(r)
    return np.array(V, dtype=np.float64), np.array(R, dtype=np.float64)


def total_wrench(vertex, ridge, scales, origin):
    """ForceColl::calcTotalWrench: returns (moment [3], force [3]) of sum_r scales_r * ridge_r at vertex_r about origin."""
    m = len(scales)
    f = (np.asarray(scales)[:, None] * ridge[:m]).sum(axis=0) if m else np.zeros(3)
    n = (np.asarray(scales)[:, None] * np.cross(vertex[:m] - origin, ridge[:m])).sum(axis=0) if m else np.zeros(3)
    return n, f


class CentroidalSim:
    """SimModels.h:233-340: state (pos, ori, vel, ang_vel, lin_momentum, ang_momentum); input wrench (force, moment about
    the CoM).  The continuous model is a chain of integrators, so the ZOH of StateSpaceModel.h:205-214 is exact:
    x += v dt + a dt^2/2, v += a dt, momentum += wrench dt."""

    def __init__(self, mass, moment_of_inertia, sim_dt):
        self.mass, self.inertia, self.dt = float(mass), np.asarray(moment_of_inertia, dtype=np.float64), float(sim_dt)
        self.pos, self.ori = np.zeros(3), np.zeros(3)
        self.vel, self.ang_vel = np.zeros(3), np.zeros(3)
        self.lin_mom, self.ang_mom = np.zeros(3), np.zeros(3)

    def update(self, force, moment):
        dt = self.dt
        a = np.asarray(force) / self.mass + np.array([0.0, 0.0, -G])
        al = np.asarray(moment) / self.inertia
        self.pos = self.pos + self.vel * dt + 0.5 * a * dt * dt
        self.ori = self.ori + self.ang_vel * dt + 0.5 * al * dt * dt
        self.vel = self.vel + a * dt
        self.ang_vel = self.ang_vel + al * dt
        self.lin_mom = self.lin_mom + (np.asarray(force) + np.array([0.0, 0.0, -self.mass * G])) * dt
        self.ang_mom = self.ang_mom + np.asarray(moment) * dt

    def addDisturb(self, lin_impulse_per_mass, ang_impulse_per_mass):
        # SimModels.h:326-330: velocities only (the momentum states are separate integrators).  The reference tests
        # build the impulse as sva::ForceVecd(Zero, (0.05, 0.05, 0)) (TestDdpCentroidal.cpp:24-25): sva's constructor
        # order is (couple, force), so the kick is LINEAR, 0.05 m/s in x and y, and the angular part is zero.
        self.vel = self.vel + np.asarray(lin_impulse_per_mass)
        self.ang_vel = self.ang_vel + np.asarray(ang_impulse_per_mass)


def empty_problem(n, N, P, M, srb=False):
    prob = dict(phase_dim=np.zeros((n, P), dtype=np.int32), phase_vertex=np.zeros((n, P, M, 3)),
                phase_ridge=np.zeros((n, P, M, 3)), step_phase=np.zeros((n, N), dtype=np.int32),
                ref_pos=np.zeros((n, N + 1, 3)))
    if srb:
        prob["ref_ori"] = np.zeros((n, N + 1, 3))
        prob["inertia"] = np.tile(np.eye(3), (n, 1, 1))
    return prob


def reference_schedule(t, second_rect_x=(0.4, 0.6)):
    """TestDdpCentroidal.cpp:35-80 at time t (the +1e-6 of :39,:60 included): (phase id, ref pos)."""
    t = t + 1e-6
    if t < 1.4:
        return 0, np.array([0.0, 0.0, 1.0])
    if t < 1.6:
        return 1, np.array([0.25, 0.0, 1.2])
    return 2, np.array([0.5, 0.0, 1.0])


def reference_problem(current_time, N, dt, P=4, M=16, rect_half=(0.1, 0.1), srb=False, inertia=None,
                      ori_ref_func=None):
    """One flattened instance of the reference closed-loop scenario sampled at current_time + i*dt, i = 0..N."""
    prob = empty_problem(1, N, P, M, srb)
    hx, hy = rect_half
    V0, R0 = contact_from_rect((-hx, -hy), (hx, hy))
    V2, R2 = contact_from_rect((0.5 - hx, -hy), (0.5 + hx, hy))
    prob["phase_dim"][0, :3] = [16, 0, 16]
    prob["phase_vertex"][0, 0], prob["phase_ridge"][0, 0] = V0, R0
    prob["phase_vertex"][0, 2], prob["phase_ridge"][0, 2] = V2, R2
    for i in range(N + 1):
        ph, ref = reference_schedule(current_time + i * dt)
        if i < N:
            prob["step_phase"][0, i] = ph
        prob["ref_pos"][0, i] = ref
        if srb and ori_ref_func is not None:
            prob["ref_ori"][0, i] = ori_ref_func(current_time + i * dt)
    if srb and inertia is not None:
        prob["inertia"][0] = inertia
    return prob


def centroidal_weights(running_pos=(1.0, 1.0, 10.0), terminal_pos=(1.0, 1.0, 10.0)):
    """WeightParam of TestDdpCentroidal.cpp:28-31 on top of the defaults of DdpCentroidal.h:66-80."""
    return dict(run=list(running_pos) + [0.0] * 3 + [1.0] * 3, term=list(terminal_pos) + [0.0] * 3 + [1.0] * 3,
                force=1e-6)


def srb_weights(running_pos=(1.0, 1.0, 10.0), terminal_pos=(1.0, 1.0, 10.0), running_ori=0.5, terminal_ori=0.5):
    """WeightParam of TestDdpSingleRigidBody.cpp:28-33 on top of the defaults of DdpSingleRigidBody.h:89-97."""
    return dict(run=list(running_pos) + [running_ori] * 3 + [0.01] * 6,
                term=list(terminal_pos) + [terminal_ori] * 3 + [0.01] * 6, force=1e-6)


def make_varying_inertia_batch(n, N, dt, seed):
    """A single-rigid-body batch whose MotionParam::inertia_mat changes over the horizon (the reference reads
    motion_param_func_(t).inertia_mat at every step, src/DdpSingleRigidBody.cpp:56-57,120-123): P = 7 phases -- the three
    contact phases of make_centroidal_batch, each with its own full (non-diagonal) SPD matrix, plus copies of the first
    stance phase's contacts with OTHER matrices for some of its steps (same contact list, another inertia: a phase boundary
    inside a stance).  Returns (prob with inertia [n,3,3], prob with inertia [n,P,3,3], x0)."""
    prob, x0 = make_centroidal_batch(n, N, dt, seed=seed, srb=True, P=7)
    rng = np.random.default_rng(seed)
    A = rng.normal(size=(n, 7, 3, 3)) * 2.0
    I4 = A @ np.swapaxes(A, -1, -2) + np.diag([40.0, 20.0, 10.0])
    for p in (4, 5):
        prob["phase_dim"][:, p] = prob["phase_dim"][:, 0]
        prob["phase_vertex"][:, p], prob["phase_ridge"][:, p] = prob["phase_vertex"][:, 0], prob["phase_ridge"][:, 0]
    first = prob["step_phase"] == 0
    idx = np.arange(N)[None, :]
    prob["step_phase"][first & (idx % 7 >= 3)] = 4
    prob["step_phase"][first & (idx % 7 >= 5)] = 5
    return prob, dict(prob, inertia=np.ascontiguousarray(I4)), x0


def make_centroidal_batch(n, N=100, dt=0.03, mass=100.0, P=4, M=16, seed=20250928, srb=False):
    """Synthetic DDP workload of SURVEY.md section 8(d): rect 0.2x0.2 contact at the origin, a 0.2 s flight window
    starting at U(0.9, 1.9) s, then the rect shifted by U(0.2, 0.6) in x; reference CoM height 1.0 (1.2 in flight);
    x0: c = ref + U(-0.05, 0.05)^3, v ~ U(-0.1, 0.1)^3, L = 0.  PRNG numpy default_rng(seed) (PCG64).
    Returns (prob, x0 [n,S])."""
    rng = np.random.default_rng(seed)
    prob = empty_problem(n, N, P, M, srb)
    hx, hy = (0.1, 0.5) if srb else (0.1, 0.1)
    V0, R0 = contact_from_rect((-hx, -hy), (hx, hy))
    t_flight = rng.uniform(0.9, 1.9, size=n)
    shift = rng.uniform(0.2, 0.6, size=n)
    prob["phase_dim"][:, 0] = 16
    prob["phase_dim"][:, 2] = 16
    prob["phase_vertex"][:, 0] = V0
    prob["phase_ridge"][:, 0] = R0
    prob["phase_vertex"][:, 2] = V0[None] + np.concatenate([shift[:, None, None], np.zeros((n, 1, 2))], axis=2)
    prob["phase_ridge"][:, 2] = R0
    t = dt * np.arange(N + 1)[None, :] + 1e-6
    ph = np.where(t < t_flight[:, None], 0, np.where(t < t_flight[:, None] + 0.2, 1, 2))
    prob["step_phase"][:] = ph[:, :N]
    ref = np.zeros((n, N + 1, 3))
    ref[:, :, 0] = np.where(ph == 0, 0.0, np.where(ph == 1, 0.5 * shift[:, None], shift[:, None]))
    ref[:, :, 2] = np.where(ph == 1, 1.2, 1.0)
    prob["ref_pos"][:] = ref
    c0 = ref[:, 0, :] + rng.uniform(-0.05, 0.05, size=(n, 3))
    v0 = rng.uniform(-0.1, 0.1, size=(n, 3))
    if srb:
        prob["inertia"][:] = np.diag([40.0, 20.0, 10.0])
        x0 = np.concatenate([c0, rng.uniform(-0.05, 0.05, size=(n, 3)), v0, rng.uniform(-0.1, 0.1, size=(n, 3))],
                            axis=1)
    else:
        x0 = np.concatenate([c0, mass * v0, np.zeros((n, 3))], axis=1)
    return prob, np.ascontiguousarray(x0)


def make_walking_batch(n, N=40, dt=0.05, mass=100.0, M=32, seed=20250928, srb=False, step_duration=(0.3, 0.5),
                       double_support=(0.1, 0.2), flight_prob=0.25):
    """Walking sequences that need the contact lists of the reference in full (src/DdpCentroidal.cpp:49-60 iterates
    an arbitrary contact_list): DOUBLE support = two surface contacts = 32 ridges, single support = 16, an
    occasional flight phase = 0, and as many distinct contact phases as the horizon holds (P = number of phases of the
    longest instance, typically 6-10, i.e. more than the four of the reference test scenario).

    Per instance: feet 0.2 x 0.1 rects at y = +-0.1, the swing foot lands 0.1-0.25 m ahead of the stance foot; phases
    alternate double support / single support (left, right, ...), a single support being replaced by a short flight
    with probability flight_prob; durations U(step_duration), U(double_support).  Reference CoM: mid-point of the
    supporting feet at height 1.0.  x0: c = ref + U(-0.03, 0.03)^3, v ~ U(-0.1, 0.1)^3.
    PRNG numpy default_rng(seed) (PCG64).  Returns (prob, x0 [n,S]) with P = prob["phase_dim"].shape[1]."""
    rng = np.random.default_rng(seed)
    hx, hy = 0.1, 0.05
    plans = []
    for k in range(n):
        feet = {0: np.array([0.0, 0.1]), 1: np.array([rng.uniform(-0.05, 0.05), -0.1])}  # left, right
        swing = int(rng.integers(0, 2))
        t, phases = 0.0, []  # (end time, [foot centres], ref xy)
        horizon = N * dt + 1e-3
        while t < horizon:
            dur = rng.uniform(*double_support)
            both = [feet[0].copy(), feet[1].copy()]
            t += dur
            phases.append((t, both, 0.5 * (both[0] + both[1])))
            if t >= horizon:
                break
            stance = 1 - swing
            dur = rng.uniform(*step_duration)
            t += dur
            if rng.uniform() < flight_prob:
                phases.append((t - 0.5 * dur, [feet[stance].copy()], feet[stance].copy()))
                phases.append((t, [], feet[stance] + np.array([0.05, 0.0])))
            else:
                phases.append((t, [feet[stance].copy()], feet[stance].copy()))
            feet[swing] = np.array([feet[stance][0] + rng.uniform(0.1, 0.25), feet[swing][1]])
            swing = stance
        plans.append(phases)
    P = max(len(ph) for ph in plans)
    prob = empty_problem(n, N, P, M, srb)
    for k, phases in enumerate(plans):
        ends = np.array([e for e, _, _ in phases])
        for p, (_, centres, _) in enumerate(phases):
            r = 0
            for c in centres:
                V, R = contact_from_rect((c[0] - hx, c[1] - hy), (c[0] + hx, c[1] + hy))
                prob["phase_vertex"][k, p, r:r + 16], prob["phase_ridge"][k, p, r:r + 16] = V, R
                r += 16
            prob["phase_dim"][k, p] = r
        for i in range(N + 1):
            p = min(int(np.searchsorted(ends, i * dt + 1e-6, side="right")), len(phases) - 1)
            if i < N:
                prob["step_phase"][k, i] = p
            prob["ref_pos"][k, i] = [phases[p][2][0], phases[p][2][1], 1.0]
    ref0 = prob["ref_pos"][:, 0, :]
    c0 = ref0 + rng.uniform(-0.03, 0.03, size=(n, 3))
    v0 = rng.uniform(-0.1, 0.1, size=(n, 3))
    if srb:
        prob["inertia"][:] = np.diag([40.0, 20.0, 10.0])
        x0 = np.concatenate([c0, rng.uniform(-0.05, 0.05, size=(n, 3)), v0, rng.uniform(-0.1, 0.1, size=(n, 3))],
                            axis=1)
    else:
        x0 = np.concatenate([c0, mass * v0, np.zeros((n, 3))], axis=1)
    return prob, np.ascontiguousarray(x0)


def contact_from_pose(centre, normal, tangent, half=(0.05, 0.05), mu=0.5):
    """A 4-vertex surface contact in an arbitrary pose (a hand on a wall, a foot on a slope): ForceColl::SurfaceContact
    with pose (R = [tangent, normal x tangent, normal], p = centre) -- vertices and friction-pyramid ridges of
    `contact_from_rect` mapped by the pose.  Returns (vertex [16,3], ridge [16,3])."""
    nz = np.asarray(normal, dtype=np.float64)
    nz = nz / np.linalg.norm(nz)
    tx = np.asarray(tangent, dtype=np.float64)
    tx = tx - nz * (tx @ nz)
    tx = tx / np.linalg.norm(tx)
    Rm = np.stack([tx, np.cross(nz, tx), nz], axis=1)
    V, R = contact_from_rect((-half[0], -half[1]), (half[0], half[1]), mu)
    return V @ Rm.T + np.asarray(centre, dtype=np.float64), R @ Rm.T


def make_multicontact_batch(n, N=30, dt=0.05, mass=100.0, M=64, seed=20250928, srb=False, P=6):
    """Multi-contact motions -- what CCC::DdpCentroidal exists for (src/DdpCentroidal.cpp:49-60 iterates an arbitrary
    contact_list): both feet on the ground plus one or two HAND contacts on vertical walls at x = 0.45 (normal -x) and
    y = +-0.4 (normal -+y): 32, 48 or 64 ridges per step.  Per instance P contact phases drawn from
    {feet, feet + right hand, feet + left hand, feet + both hands, one foot + both hands, left foot only},
    each held for 3-9 steps; the reference CoM drifts forward at height 0.9.  x0 as in `make_walking_batch`.
    PRNG numpy default_rng(seed) (PCG64).  Returns (prob, x0 [n,S])."""
    assert M >= 64
    rng = np.random.default_rng(seed)
    prob = empty_problem(n, N, P, M, srb)
    for k in range(n):
        lf = contact_from_rect((-0.1, 0.05), (0.1, 0.15))
        rf = contact_from_rect((-0.1 + rng.uniform(-0.03, 0.03), -0.15), (0.1, -0.05))
        rh = contact_from_pose((0.45, -0.2 + rng.uniform(-0.05, 0.05), 1.0 + rng.uniform(-0.1, 0.1)), (-1, 0, 0), (0, 1, 0))
        lh = contact_from_pose((0.1 + rng.uniform(-0.05, 0.05), 0.4, 1.1 + rng.uniform(-0.1, 0.1)), (0, -1, 0), (1, 0, 0))
        menu = [[lf, rf], [lf, rf, rh], [lf, rf, lh], [lf, rf, rh, lh], [lf, rh, lh], [lf]]
        order = rng.permutation(len(menu))[:P]
        if 3 not in order:
            order[rng.integers(0, P)] = 3  # every instance has a four-contact phase
        for p, q in enumerate(order):
            r = 0
            for V, R in menu[q]:
                prob["phase_vertex"][k, p, r:r + 16], prob["phase_ridge"][k, p, r:r + 16] = V, R
                r += 16
            prob["phase_dim"][k, p] = r
        i, p = 0, 0
        while i < N:
            d = int(rng.integers(3, 10))
            prob["step_phase"][k, i:i + d] = p % P
            i, p = i + d, p + 1
        prob["ref_pos"][k, :, 0] = 0.05 * np.arange(N + 1) * dt
        prob["ref_pos"][k, :, 2] = 0.9
    c0 = prob["ref_pos"][:, 0, :] + rng.uniform(-0.03, 0.03, size=(n, 3))
    v0 = rng.uniform(-0.1, 0.1, size=(n, 3))
    if srb:
        prob["inertia"][:] = np.diag([40.0, 20.0, 10.0])
        x0 = np.concatenate([c0, rng.uniform(-0.05, 0.05, size=(n, 3)), v0, rng.uniform(-0.1, 0.1, size=(n, 3))],
                            axis=1)
    else:
        x0 = np.concatenate([c0, mass * v0, np.zeros((n, 3))], axis=1)
    return prob, np.ascontiguousarray(x0)


def srb_ori_ref(t):
    """TestDdpSingleRigidBody.cpp:78-85 (the +1e-6 included): roll reference bump between 2.2 s and 2.4 s."""
    t = t + 1e-6
    return np.array([0.0, 0.0, 0.3]) if 2.2 < t < 2.4 else np.zeros(3)


def run_closed_loop_ddp(plan, srb=False, mass=100.0, N=100, dt=0.03, sim_dt=0.005, end_time=3.0, P=4, M=16,
                        disturb_time=1.0, lin_disturb=(0.05, 0.05, 0.0), warm_max_iter=1):
    """The control loops of TestDdpCentroidal.cpp:96-150 / TestDdpSingleRigidBody.cpp:103-170 around any
    `plan(prob, x0 [1,S], u_init [1,N,M] | None, max_iter) -> u [1,N,M]`: first cycle cold start with the full
    iteration budget, afterwards warm start (unshifted u_list, zeroed where the input dimension changed) with
    max_iter = warm_max_iter (1 in the reference).  Returns per-cycle records for the reference's property assertions."""
    inertia = np.array([40.0, 20.0, 10.0])
    sim = CentroidalSim(mass, inertia, sim_dt)
    sim.pos = reference_schedule(0.0)[1].copy()
    rect_half = (0.1, 0.5) if srb else (0.1, 0.1)
    u_prev, dims_prev = None, None
    log, t, cycle = [], 0.0, 0
    while t < end_time:
        prob = reference_problem(t, N, dt, P, M, rect_half, srb, np.diag(inertia), srb_ori_ref if srb else None)
        dims = prob["phase_dim"][0][prob["step_phase"][0]]
        if srb:
            x0 = np.concatenate([sim.pos, sim.ori[::-1], sim.vel, sim.ang_vel])[None]
        else:
            x0 = np.concatenate([sim.pos, mass * sim.vel, sim.ang_mom])[None]
        u_init = None
        if u_prev is not None:
            u_init = u_prev.copy()
            changed = dims != dims_prev
            u_init[0, changed, :] = 0.0
            for i in range(N):
                u_init[0, i, dims[i]:] = 0.0
        u = plan(prob, x0, u_init, 500 if cycle == 0 else warm_max_iter)
        u_prev, dims_prev = u, dims
        ph = prob["step_phase"][0, 0]
        m0 = prob["phase_dim"][0, ph]
        moment, force = total_wrench(prob["phase_vertex"][0, ph], prob["phase_ridge"][0, ph], u[0, 0, :m0], sim.pos)
        ref = prob["ref_pos"][0, 0]
        log.append(dict(t=t, pos=sim.pos.copy(), ref=ref.copy(), vel=sim.vel.copy(), ang_mom=sim.ang_mom.copy(),
                        ori=sim.ori.copy(), ang_vel=sim.ang_vel.copy(),
                        ori_ref=(prob["ref_ori"][0, 0][::-1].copy() if srb else np.zeros(3)),
                        ori_ref_zyx=(prob["ref_ori"][0, 0].copy() if srb else np.zeros(3)), force=force))
        t += sim_dt
        sim.update(force, moment)
        if disturb_time <= t < disturb_time + sim_dt:
            sim.addDisturb(lin_disturb, np.zeros(3))
        cycle += 1
    ref_end = reference_schedule(t)[1]
    ori_end = srb_ori_ref(t)[::-1] if srb else np.zeros(3)
    return log, dict(t=t, pos=sim.pos.copy(), ref=ref_end, vel=sim.vel.copy(), ang_mom=sim.ang_mom.copy(),
                     ori=sim.ori.copy(), ori_ref=ori_end, ori_ref_zyx=ori_end[::-1].copy(),
                     ang_vel=sim.ang_vel.copy())


# ===================================================================== LinearMpcXY fixtures
def xy_reference_schedule(t):
    """TestLinearMpcXY.cpp:29-80: (rect_min, rect_max, ref pos) at time t (no epsilon in this test)."""
    if t < 3.0:
        return (0.9, -0.15), (1.1, 0.15), (1.0, 0.0)
    if t < 4.0:
        return (0.9, 0.05), (1.1, 0.15), (1.0, 0.1)
    if t < 5.0:
        return (1.15, -0.15), (1.35, -0.05), (1.25, -0.1)
    if t < 6.0:
        return (1.4, 0.05), (1.6, 0.15), (1.5, 0.1)
    return (1.4, -0.15), (1.6, 0.15), (1.5, 0.0)


def xy_problem(current_time, N, dt, mass=100.0, M=16, schedule=xy_reference_schedule, com_z=1.0):
    """One flattened LinearMpcXY instance sampled at current_time + i*dt (src/LinearMpcXY.cpp:102-110):
    dict(dim [1,N], vertex [1,N,M,3], ridge [1,N,M,3], com_z [1,N], total_force_z [1,N], ref_out [1,N,6])."""
    prob = dict(dim=np.zeros((1, N), dtype=np.int32), vertex=np.zeros((1, N, M, 3)), ridge=np.zeros((1, N, M, 3)),
                com_z=np.full((1, N), com_z), total_force_z=np.full((1, N), mass * G), ref_out=np.zeros((1, N, 6)))
    for i in range(N):
        rmin, rmax, ref = schedule(current_time + i * dt)
        V, R = contact_from_rect(rmin, rmax)
        prob["dim"][0, i] = len(V)
        prob["vertex"][0, i, :len(V)] = V
        prob["ridge"][0, i, :len(V)] = R
        # RefData::toOutput(mass): [m px, m vx, m py, m vy, Lx, Ly] with vel = L = 0
        prob["ref_out"][0, i] = [mass * ref[0], 0.0, mass * ref[1], 0.0, 0.0, 0.0]
    return prob


def make_xy_batch(n, N=20, dt=0.1, mass=100.0, M=16, seed=20250928):
    """Synthetic LinearMpcXY workload: the contact pattern of TestLinearMpcXY.cpp:29-57 with a
    random phase (evaluation time U(0, 7) s) and a random lateral scale, x0 near the reference.
    Returns (prob, x0 [n,6]) with x0 = InitialParam::toState(mass) = [m px, m vx, m py, m vy, Lx, Ly]."""
    rng = np.random.default_rng(seed)
    t0 = rng.uniform(0.0, 7.0, size=n)
    probs = [xy_problem(t0[k], N, dt, mass, M) for k in range(n)]
    prob = {k: np.concatenate([p[k] for p in probs]) for k in probs[0]}
    ref0 = prob["ref_out"][:, 0, :]
    pos = np.stack([ref0[:, 0] / mass, ref0[:, 2] / mass], axis=1) + rng.uniform(-0.03, 0.03, size=(n, 2))
    vel = rng.uniform(-0.1, 0.1, size=(n, 2))
    am = rng.uniform(-0.5, 0.5, size=(n, 2))
    x0 = np.stack([mass * pos[:, 0], mass * vel[:, 0], mass * pos[:, 1], mass * vel[:, 1], am[:, 0], am[:, 1]], axis=1)
    return prob, np.ascontiguousarray(x0)


def make_xy_walking_batch(n, N=30, dt=0.1, mass=100.0, M=32, seed=20250928, step_duration=(0.5, 0.8),
                          double_support=(0.2, 0.4), flight_prob=0.1):
    """LinearMpcXY walking sequences that use the reference's contact lists in full (src/LinearMpcXY.cpp:69-82 and
    :126-133 iterate an arbitrary contact_list per step): DOUBLE support = two separate foot contacts = 32 ridges, single
    support = 16, an occasional flight step = 0 ridges (no variables, no equality row), over horizons longer than the 20
    steps of the reference test.  Feet 0.2 x 0.1 rects at y = +-0.1, the swing foot lands 0.15-0.3 m ahead of the stance
    foot; reference = mid-point of the supporting feet; x0 near the reference as in make_xy_batch.
    PRNG numpy default_rng(seed) (PCG64).  Returns (prob, x0 [n,6])."""
    rng = np.random.default_rng(seed)
    hx, hy = 0.1, 0.05
    prob = dict(dim=np.zeros((n, N), dtype=np.int32), vertex=np.zeros((n, N, M, 3)), ridge=np.zeros((n, N, M, 3)),
                com_z=np.full((n, N), 1.0), total_force_z=np.full((n, N), mass * G), ref_out=np.zeros((n, N, 6)))
    for k in range(n):
        feet = {0: np.array([1.0, 0.1]), 1: np.array([1.0 + rng.uniform(-0.05, 0.05), -0.1])}
        swing = int(rng.integers(0, 2))
        t, phases = -rng.uniform(0.0, 0.3), []  # (end time, [foot centres], ref xy)
        horizon = N * dt
        while t < horizon:
            t += rng.uniform(*double_support)
            both = [feet[0].copy(), feet[1].copy()]
            phases.append((t, both, 0.5 * (both[0] + both[1])))
            if t >= horizon:
                break
            stance = 1 - swing
            dur = rng.uniform(*step_duration)
            t += dur
            if rng.uniform() < flight_prob:
                phases.append((t - 0.2, [feet[stance].copy()], feet[stance].copy()))
                phases.append((t, [], feet[stance] + np.array([0.05, 0.0])))
            else:
                phases.append((t, [feet[stance].copy()], feet[stance].copy()))
            feet[swing] = np.array([feet[stance][0] + rng.uniform(0.15, 0.3), feet[swing][1]])
            swing = stance
        ends = np.array([e for e, _, _ in phases])
        for i in range(N):
            p = min(int(np.searchsorted(ends, i * dt, side="right")), len(phases) - 1)
            r = 0
            for c in phases[p][1]:
                V, R = contact_from_rect((c[0] - hx, c[1] - hy), (c[0] + hx, c[1] + hy))
                prob["vertex"][k, i, r:r + 16], prob["ridge"][k, i, r:r + 16] = V, R
                r += 16
            prob["dim"][k, i] = r
            if r == 0:
                prob["total_force_z"][k, i] = 0.0  # no contact, no force (src/LinearMpcXY.cpp:126-133 skips the row)
            prob["ref_out"][k, i] = [mass * phases[p][2][0], 0.0, mass * phases[p][2][1], 0.0, 0.0, 0.0]
    ref0 = prob["ref_out"][:, 0, :]
    pos = np.stack([ref0[:, 0] / mass, ref0[:, 2] / mass], axis=1) + rng.uniform(-0.03, 0.03, size=(n, 2))
    vel = rng.uniform(-0.1, 0.1, size=(n, 2))
    am = rng.uniform(-0.5, 0.5, size=(n, 2))
    x0 = np.stack([mass * pos[:, 0], mass * vel[:, 0], mass * pos[:, 1], mass * vel[:, 1], am[:, 0], am[:, 1]], axis=1)
    return prob, np.ascontiguousarray(x0)


def make_xy_multicontact_batch(n, N=20, dt=0.1, mass=100.0, M=64, seed=20250928):
    """LinearMpcXY with more than two contacts per step (src/LinearMpcXY.cpp:69-82, :126-133 take any contact_list): the
    contact phases of `make_multicontact_batch` (feet + hands on walls: 16 .. 64 ridges) laid out per horizon step.
    com_z 0.9, total_force_z = m g, reference = the drifting CoM of that fixture; x0 near the reference.
    Returns (prob, x0 [n,6])."""
    d, _ = make_multicontact_batch(n, N, dt, mass, M, seed)
    rng = np.random.default_rng(seed + 1)
    prob = dict(dim=np.zeros((n, N), dtype=np.int32), vertex=np.zeros((n, N, M, 3)), ridge=np.zeros((n, N, M, 3)),
                com_z=np.full((n, N), 0.9), total_force_z=np.full((n, N), mass * G), ref_out=np.zeros((n, N, 6)))
    idx = np.arange(n)[:, None]
    ph = d["step_phase"]
    prob["dim"][:] = d["phase_dim"][idx, ph]
    prob["vertex"][:] = d["phase_vertex"][idx, ph]
    prob["ridge"][:] = d["phase_ridge"][idx, ph]
    prob["ref_out"][:, :, 0] = mass * d["ref_pos"][:, :N, 0]
    prob["ref_out"][:, :, 2] = mass * d["ref_pos"][:, :N, 1]
    ref0 = prob["ref_out"][:, 0, :]
    pos = np.stack([ref0[:, 0] / mass, ref0[:, 2] / mass], axis=1) + rng.uniform(-0.03, 0.03, size=(n, 2))
    vel = rng.uniform(-0.1, 0.1, size=(n, 2))
    am = rng.uniform(-0.5, 0.5, size=(n, 2))
    x0 = np.stack([mass * pos[:, 0], mass * vel[:, 0], mass * pos[:, 1], mass * vel[:, 1], am[:, 0], am[:, 1]], axis=1)
    return prob, np.ascontiguousarray(x0)
