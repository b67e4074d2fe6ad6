"""Input generators for the LinearMpcZmp path: restated reference test fixtures + the synthetic batch.

Restates (numpy, host-side; these produce INPUTS, they are not on the timed path):
  /root/reference/tests/src/FootstepManager.h:35-76     Footstep
  /root/reference/tests/src/FootstepManager.h:81-127    Footstance.midPos / supportRegion
  /root/reference/tests/src/FootstepManager.h:130-254   FootstepManager.update / appendFootstep / zmpLimits
  /root/reference/tests/src/FootstepManager.h:356-365   makeLinearMpcZmpRefData (the second +1e-6)
  /root/reference/tests/src/SimModels.h:11-41,76-137    ComZmpSimModel1d / ComZmpSim2d (exact ZOH of the LIPM)
and the synthetic workload of SURVEY.md section 8(d) / BASELINE.md section 3 (`make_zmp_batch`).
"""
import bisect
import math

import numpy as np

from .linear_mpc_zmp import InitialParam, RefData

G = 9.80665  # include/CCC/Constants.h:10
LEFT, RIGHT = 0, 1


def opposite(foot):
    return RIGHT if foot == LEFT else LEFT


class Footstep:
    """FootstepManager.h:35-76."""

    def __init__(self, foot, pos, transit_start_time, transit_duration, swing_duration):
        self.foot = foot
        self.pos = np.asarray(pos, dtype=np.float64)
        self.transit_start_time = transit_start_time
        self.swing_start_time = transit_start_time + 0.5 * transit_duration
        self.swing_end_time = transit_start_time + 0.5 * transit_duration + swing_duration
        self.transit_end_time = transit_start_time + transit_duration + swing_duration


def _mid_pos(stance):
    if len(stance) == 1:
        return next(iter(stance.values())).copy()
    return 0.5 * (stance[LEFT] + stance[RIGHT])


def _support_region(stance):
    if len(stance) == 1:
        v = next(iter(stance.values()))
        return [v.copy(), v.copy()]
    return [np.minimum(stance[LEFT], stance[RIGHT]), np.maximum(stance[LEFT], stance[RIGHT])]


class FootstepManager:
    """FootstepManager.h:130-254 (only what the LinearMpcZmp path uses)."""

    def __init__(self, initial_footstance=None):
        self.footstance_ = initial_footstance or {LEFT: np.array([0.0, 0.1]), RIGHT: np.array([0.0, -0.1])}
        self.footstep_list_ = []
        self.horizon_duration_ = 10.0
        self.foot_size_ = np.array([0.1, 0.05])
        self._stance_times = []
        self._stances = []

    def appendFootstep(self, footstep):
        if self.footstep_list_ and footstep.transit_start_time < self.footstep_list_[-1].transit_end_time:
            raise RuntimeError("[FootstepManager::appendFootstep] transit_start_time of specified footstep must be "
                               "after transit_end_time of last footstep.")
        self.footstep_list_.append(footstep)

    def update(self, current_time):
        # :147-160
        if self.footstep_list_ and self.footstep_list_[0].swing_end_time <= current_time:
            self.footstance_[self.footstep_list_[0].foot] = self.footstep_list_[0].pos
        while self.footstep_list_ and self.footstep_list_[0].transit_end_time < current_time:
            self.footstep_list_.pop(0)
        # :162-206  (std::map::emplace keeps the FIRST value of a key)
        entries = {}
        zmps = {}  # ref_zmp_list_

        def emplace(t, stance):
            if t not in entries:
                entries[t] = {k: v.copy() for k, v in stance.items()}

        def emplace_zmp(t, zmp):
            if t not in zmps:
                zmps[t] = np.array(zmp, dtype=np.float64)

        if not self.footstep_list_:
            emplace_zmp(current_time, _mid_pos(self.footstance_))
            emplace_zmp(current_time + self.horizon_duration_, _mid_pos(self.footstance_))
            emplace(current_time, self.footstance_)
            emplace(current_time + self.horizon_duration_, self.footstance_)
        else:
            if current_time < self.footstep_list_[0].transit_start_time:
                emplace_zmp(current_time, _mid_pos(self.footstance_))
                emplace(current_time, self.footstance_)
            tmp = {k: v.copy() for k, v in self.footstance_.items()}
            for fs in self.footstep_list_:
                if not fs.transit_start_time <= current_time + self.horizon_duration_:
                    break
                emplace_zmp(fs.transit_start_time, _mid_pos(tmp))
                emplace(fs.transit_start_time, tmp)
                tmp.pop(fs.foot, None)
                emplace_zmp(fs.swing_start_time, tmp[opposite(fs.foot)])
                emplace(fs.swing_start_time, tmp)
                tmp.setdefault(fs.foot, fs.pos.copy())
                emplace_zmp(fs.swing_end_time, tmp[opposite(fs.foot)])
                emplace(fs.swing_end_time, tmp)
                emplace_zmp(fs.transit_end_time, _mid_pos(tmp))
            # the reference compares against the last key of ref_zmp_list_ (:201)
            if max(zmps) < current_time + self.horizon_duration_:
                emplace_zmp(current_time + self.horizon_duration_, _mid_pos(tmp))
                emplace(current_time + self.horizon_duration_, tmp)
        self._stance_times = sorted(entries)
        self._stances = [entries[t] for t in self._stance_times]
        self._zmp_times = sorted(zmps)
        self._zmps = [zmps[t] for t in self._zmp_times]

    def zmpLimits(self, t):
        # :242-254
        t = t + 1e-6
        k = bisect.bisect_right(self._stance_times, t) - 1
        region = _support_region(self._stances[k])
        return [region[0] - 0.5 * self.foot_size_, region[1] + 0.5 * self.foot_size_]

    def refZmp(self, t):
        # :228-237 (linear interpolation in ref_zmp_list_)
        t = t + 1e-6
        k = bisect.bisect_right(self._zmp_times, t)
        t0, t1 = self._zmp_times[k - 1], self._zmp_times[k]
        ratio = (t - t0) / (t1 - t0)
        return (1 - ratio) * self._zmps[k - 1] + ratio * self._zmps[k]

    def makeIntrinsicallyStableMpcRefData(self, t):
        # :370-380 (its own +1e-6 on top of those of refZmp / zmpLimits): (ref zmp, zmin, zmax)
        t = t + 1e-6
        lim = self.zmpLimits(t)
        return self.refZmp(t), lim[0], lim[1]

    def makeLinearMpcZmpRefData(self, t):
        # :356-365
        t = t + 1e-6
        lim = self.zmpLimits(t)
        return RefData(lim[0], lim[1])


class ComZmpSim2d:
    """SimModels.h:76-137 with the 1-D model of :11-41, ZOH-discretised exactly:
    x'' = w^2 (x - zmp)  ->  [x, v]+ = [[ch, sh/w],[w sh, ch]] [x, v] + [1 - ch, -w sh] zmp."""

    def __init__(self, com_height, sim_dt):
        w = math.sqrt(G / com_height)
        ch, sh = math.cosh(w * sim_dt), math.sinh(w * sim_dt)
        self.Ad = np.array([[ch, sh / w], [w * sh, ch]])
        self.Bd = np.array([1 - ch, -w * sh])
        self.x = np.zeros(2)
        self.y = np.zeros(2)

    def pos(self):
        return np.array([self.x[0], self.y[0]])

    def vel(self):
        return np.array([self.x[1], self.y[1]])

    def update(self, zmp):
        self.x = self.Ad @ self.x + self.Bd * zmp[0]
        self.y = self.Ad @ self.y + self.Bd * zmp[1]

    def addDisturb(self, impulse_per_mass):
        # SimModels.h:125-129 adds impulse.x() to BOTH axes (reference quirk, kept)
        self.x[1] += impulse_per_mass[0]
        self.y[1] += impulse_per_mass[0]


def zmp_limits_timeline(foot0, foot_pos, foot_id, swing_start, swing_end, times, foot_size=(0.1, 0.05)):
    """Vectorised FootstepManager::zmpLimits for many independent footstep timelines.

    foot0 [n,2(L/R),2], foot_pos [n,K,2], foot_id [n,K] (0 = left, 1 = right), swing_start/swing_end [n,K],
    times [n,T] (already including the two +1e-6 of FootstepManager.h:245,360).
    Returns (zmin, zmax) each [n,T,2].  A foot is off the ground during [swing_start, swing_end) and
    sits at its new position from swing_end on (FootstepManager.h:179-196)."""
    n, K = foot_id.shape
    T = times.shape[1]
    pos = np.repeat(foot0[:, None, :, :], T, axis=1).copy()  # [n,T,2 feet,2]
    on_ground = np.ones((n, T, 2), dtype=bool)
    for k in range(K):
        f = foot_id[:, k]
        landed = times >= swing_end[:, k:k + 1]
        swinging = (times >= swing_start[:, k:k + 1]) & ~landed
        for foot in (0, 1):
            sel = (f == foot)[:, None]
            upd = landed & sel
            pos[:, :, foot, :] = np.where(upd[..., None], foot_pos[:, k][:, None, :], pos[:, :, foot, :])
            on_ground[:, :, foot] &= ~(swinging & sel)
    big = 1e30
    lo = np.where(on_ground[..., None], pos, big).min(axis=2)
    hi = np.where(on_ground[..., None], pos, -big).max(axis=2)
    half = 0.5 * np.asarray(foot_size)
    return lo - half, hi + half


def make_zmp_batch(n, horizon_steps=32, horizon_dt=0.0625, com_height=1.0, seed=20250928, num_footsteps=6):
    """Synthetic LinearMpcZmp workload of SURVEY.md section 8(d): random 6-step footstep sequences of the
    shape of TestLinearMpcZmp.cpp:30-43, evaluated at a random time, with a random initial CoM state.
    PRNG: numpy.random.default_rng(seed) (PCG64).  Returns dict(x0 [n,2,3], zlim [n,2,2,N])."""
    rng = np.random.default_rng(seed)
    K, N = num_footsteps, horizon_steps
    foot0 = np.empty((n, 2, 2))
    foot0[:, LEFT] = [0.0, 0.1]
    foot0[:, RIGHT] = [0.0, -0.1]
    first_foot = rng.integers(0, 2, size=n)
    foot_id = (first_foot[:, None] + np.arange(K)[None, :]) % 2
    step_x = rng.uniform(-0.1, 0.3, size=(n, K))
    lateral = 0.1 + rng.uniform(-0.02, 0.05, size=(n, K))
    foot_pos = np.empty((n, K, 2))
    foot_pos[:, :, 0] = np.cumsum(step_x, axis=1)
    foot_pos[:, :, 1] = np.where(foot_id == LEFT, lateral, -lateral)
    transit_duration, swing_duration, period = 0.2, 0.8, 1.0
    t_first = rng.uniform(0.3, 2.0, size=n)
    transit_start = t_first[:, None] + period * np.arange(K)[None, :]
    swing_start = transit_start + 0.5 * transit_duration
    swing_end = swing_start + swing_duration
    t_eval = rng.uniform(0.0, 8.0, size=n)
    times = t_eval[:, None] + horizon_dt * np.arange(N)[None, :] + 2e-6
    zmin, zmax = zmp_limits_timeline(foot0, foot_pos, foot_id, swing_start, swing_end, times)
    zlim = np.empty((n, 2, 2, N))
    zlim[:, :, 0, :] = np.transpose(zmin, (0, 2, 1))
    zlim[:, :, 1, :] = np.transpose(zmax, (0, 2, 1))
    # initial state: CoM near the middle of the current support region
    mid = 0.5 * (zmin[:, 0, :] + zmax[:, 0, :])
    pos = mid + rng.uniform(-0.03, 0.03, size=(n, 2))
    vel = rng.uniform(-0.2, 0.2, size=(n, 2))
    zmp_prev = pos + rng.uniform(-0.02, 0.02, size=(n, 2))
    acc = G / com_height * (pos - zmp_prev)  # TestLinearMpcZmp.cpp:69
    x0 = np.stack([pos, vel, acc], axis=2)  # [n,2 axes,3]
    return dict(x0=np.ascontiguousarray(x0), zlim=np.ascontiguousarray(zlim), t_eval=t_eval)


def reference_scenario_footsteps():
    """The six footsteps of TestLinearMpcZmp.cpp:30-43."""
    td, sd = 0.2, 0.8
    return [
        Footstep(LEFT, (0.2, 0.1), 2.0, td, sd),
        Footstep(RIGHT, (0.4, -0.1), 3.0, td, sd),
        Footstep(LEFT, (0.6, 0.1), 4.0, td, sd),
        Footstep(RIGHT, (0.8, -0.1), 5.0, td, sd),
        Footstep(LEFT, (0.6, 0.1), 6.0, td, sd),
        Footstep(RIGHT, (0.6, -0.1), 7.0, td, sd),
    ]


def run_closed_loop(plan_once, com_height=1.0, sim_dt=0.005, end_time=10.0, disturb_times=(4.5, 8.5),
                    disturb=(0.05, 0.05)):
    """The control loop of TestLinearMpcZmp.cpp:55-102 around any `plan_once(ref_func, InitialParam, t, sim_dt)`.
    Returns a list of per-cycle records and the final state, for the property assertions of :86-87,:106-109."""
    fm = FootstepManager()
    for fs in reference_scenario_footsteps():
        fm.appendFootstep(fs)
    sim = ComZmpSim2d(com_height, sim_dt)
    planned = sim.pos()
    t = 0.0
    log = []
    while t < end_time:
        fm.update(t)
        ip = InitialParam(sim.pos(), sim.vel(), G / com_height * (sim.pos() - planned))
        planned = np.asarray(plan_once(fm.makeLinearMpcZmpRefData, ip, t, sim_dt))
        lim = fm.zmpLimits(t)
        log.append(dict(t=t, com=sim.pos(), zmp=planned.copy(), zmin=lim[0], zmax=lim[1]))
        t += sim_dt
        sim.update(planned)
        for dtm in disturb_times:
            if dtm <= t < dtm + sim_dt:
                sim.addDisturb(disturb)
                break
    lim = fm.zmpLimits(t)
    return log, dict(t=t, com=sim.pos(), zmp=planned, zmin=lim[0], zmax=lim[1])


def run_closed_loop_ism(plan_once, com_height=1.0, sim_dt=0.005, end_time=10.0, disturb_times=(4.5, 8.5),
                        disturb=(0.05, 0.05)):
    """The control loop of TestIntrinsicallyStableMpc.cpp:55-100 around any
    `plan_once(ref_func, capture_point [2], planned_zmp [2], t, sim_dt) -> zmp [2]` with
    `ref_func(t) -> (ref zmp [2], zmin [2], zmax [2])`.  Returns per-cycle records and the final state for the property
    assertions of :84-85,:103-106."""
    fm = FootstepManager()
    for fs in reference_scenario_footsteps():
        fm.appendFootstep(fs)
    sim = ComZmpSim2d(com_height, sim_dt)
    planned = sim.pos()
    t = 0.0
    log = []
    while t < end_time:
        fm.update(t)
        cp = sim.pos() + math.sqrt(com_height / G) * sim.vel()  # :67
        planned = np.asarray(plan_once(fm.makeIntrinsicallyStableMpcRefData, cp, planned, t, sim_dt))
        lim = fm.zmpLimits(t)
        log.append(dict(t=t, com=sim.pos(), zmp=planned.copy(), zmin=lim[0], zmax=lim[1], cp=cp))
        t += sim_dt
        sim.update(planned)
        for dtm in disturb_times:
            if dtm <= t < dtm + sim_dt:
                sim.addDisturb(disturb)
                break
    lim = fm.zmpLimits(t)
    return log, dict(t=t, com=sim.pos(), zmp=planned, zmin=lim[0], zmax=lim[1])


def sample_ism_refs(ref_func, t, N, dt):
    """IntrinsicallyStableMpc::planOnce sampling (src/IntrinsicallyStableMpc.cpp:112-124): [2 axes][3][N] rows
    (ref zmp, zmin, zmax)."""
    out = np.zeros((2, 3, N))
    for i in range(N):
        z, lo, hi = ref_func(t + i * dt)
        out[:, 0, i], out[:, 1, i], out[:, 2, i] = z, lo, hi
    return out


def make_ism_batch(n, horizon_steps=100, horizon_dt=0.02, com_height=1.0, seed=20250928):
    """Synthetic IntrinsicallyStableMpc workload: random evaluation times / states along the reference scenario of
    TestIntrinsicallyStableMpc.cpp:30-43 (capture point = stance reference + U(-0.04, 0.04), planned ZMP inside the
    current limits).  Returns dict(init [n,2,2], ref [n,2,3,N])."""
    rng = np.random.default_rng(seed)
    init = np.zeros((n, 2, 2))
    ref = np.zeros((n, 2, 3, horizon_steps))
    cache = {}
    for k in range(n):
        t = round(float(rng.uniform(0.0, 9.0)) / 0.005) * 0.005
        if t not in cache:
            fm = FootstepManager()
            for fs in reference_scenario_footsteps():
                fm.appendFootstep(fs)
            # replay update() up to t like the control loop does (it mutates the footstance)
            tt = 0.0
            while tt < t - 1e-12:
                fm.update(tt)
                tt += 0.05
            fm.update(t)
            cache[t] = sample_ism_refs(fm.makeIntrinsicallyStableMpcRefData, t, horizon_steps, horizon_dt)
        ref[k] = cache[t]
        lo, hi, z = ref[k, :, 1, 0], ref[k, :, 2, 0], ref[k, :, 0, 0]
        init[k, :, 1] = np.clip(z + rng.uniform(-0.03, 0.03, size=2), lo, hi)
        # capture point = the one a feasible ZMP trajectory implies through the stability constraint (eq. (14)):
        # z_{i+1} = zmin_i + rho (zmax_i - zmin_i), u_i = (z_{i+1} - z_i)/dt, cp = z0 + a'u.  rho near 0 or 1 makes
        # the limit rows of eq. (8) bind; an arbitrary capture point is usually infeasible.
        om = math.sqrt(G / com_height)
        lam = math.exp(-om * horizon_dt)
        a = (1 - lam) / (om * (1 - lam ** horizon_steps)) * lam ** np.arange(horizon_steps)
        for ax in range(2):
            rho = rng.choice([rng.uniform(0.0, 1.0), rng.uniform(0.0, 0.03), rng.uniform(0.97, 1.0)])
            traj = ref[k, ax, 1] + rho * (ref[k, ax, 2] - ref[k, ax, 1])
            u = np.diff(np.concatenate([[init[k, ax, 1]], traj])) / horizon_dt
            init[k, ax, 0] = init[k, ax, 1] + a @ u
    return dict(init=np.ascontiguousarray(init), ref=np.ascontiguousarray(ref))


# ------------------------------------------------------------------------------------------------ LinearMpcZ
def z_reference_contact(t):
    """TestLinearMpcZ.cpp:26: two flight windows."""
    return not ((5.0 < t < 5.25) or (6.0 < t < 6.5))


def z_reference_height(t):
    """TestLinearMpcZ.cpp:27."""
    return 1.0 if t < 8.5 else 0.8


class VerticalSim:
    """SimModels.h:44-73: state [z, zdot], input force; exact ZOH of the double integrator with gravity."""

    def __init__(self, mass, sim_dt):
        self.Ad = np.array([[1.0, sim_dt], [0.0, 1.0]])
        self.Bd = np.array([0.5 * sim_dt * sim_dt, sim_dt]) / mass
        self.Ed = -G * np.array([0.5 * sim_dt * sim_dt, sim_dt])
        self.state = np.zeros(2)

    def update(self, force):
        self.state = self.Ad @ self.state + self.Bd * force + self.Ed


def run_closed_loop_z(plan_once, mass=100.0, sim_dt=0.04, end_time=10.0):
    """The control loop of TestLinearMpcZ.cpp:41-72 around any `plan_once(contact_func, ref_pos_func, state [2], t)`.
    Returns per-cycle records and the final (t, state)."""
    sim = VerticalSim(mass, sim_dt)
    sim.state = np.array([z_reference_height(0.0), 0.0])
    t, log = 0.0, []
    while t < end_time:
        f = float(plan_once(z_reference_contact, z_reference_height, sim.state.copy(), t))
        log.append(dict(t=t, state=sim.state.copy(), ref=z_reference_height(t), force=f, contact=z_reference_contact(t)))
        t += sim_dt
        sim.update(f)
    return log, (t, sim.state.copy())


def make_z_batch(n, horizon_steps=40, horizon_dt=0.05, seed=20250928):
    """Synthetic LinearMpcZ workload: up to two random flight windows inside the horizon (never at step 0 for 7 of 8
    instances), a reference height with one random step change, and initial heights / velocities far enough from the
    reference that the force bounds (10 N, 10 m g) bind.  Returns dict(contact [n,N] i32, ref_pos [n,N], x0 [n,2])."""
    rng = np.random.default_rng(seed)
    N = horizon_steps
    contact = np.ones((n, N), dtype=np.int32)
    ref = np.ones((n, N))
    for k in range(n):
        for _ in range(rng.integers(0, 3) if N > 1 else 0):
            a = 0 if rng.random() < 0.125 else rng.integers(1, N)
            contact[k, a:a + rng.integers(1, 9)] = 0
        ref[k, rng.integers(0, N):] = rng.uniform(0.7, 1.2)
    x0 = np.stack([1.0 + rng.uniform(-0.5, 0.5, size=n), rng.uniform(-3.0, 3.0, size=n)], axis=1)
    return dict(contact=contact, ref_pos=ref, x0=np.ascontiguousarray(x0))


def make_zmp_timelines(n, seed=20250928, num_footsteps=6):
    """Random footstep timelines of the shape of TestLinearMpcZmp.cpp:30-43 (the ones make_zmp_batch draws), in the
    device format of include/ccc_amd.h: dict(foot0 [n,2,2], foot_pos [n,K,2], foot_id [n,K] i32, swing_start [n,K],
    swing_end [n,K])."""
    rng = np.random.default_rng(seed)
    K = num_footsteps
    foot0 = np.empty((n, 2, 2))
    foot0[:, LEFT] = [0.0, 0.1]
    foot0[:, RIGHT] = [0.0, -0.1]
    first_foot = rng.integers(0, 2, size=n)
    foot_id = ((first_foot[:, None] + np.arange(K)[None, :]) % 2).astype(np.int32)
    step_x = rng.uniform(-0.1, 0.3, size=(n, K))
    lateral = 0.1 + rng.uniform(-0.02, 0.05, size=(n, K))
    foot_pos = np.empty((n, K, 2))
    foot_pos[:, :, 0] = np.cumsum(step_x, axis=1)
    foot_pos[:, :, 1] = np.where(foot_id == LEFT, lateral, -lateral)
    t_first = rng.uniform(0.3, 2.0, size=n)
    transit_start = t_first[:, None] + 1.0 * np.arange(K)[None, :]
    swing_start = transit_start + 0.1
    swing_end = swing_start + 0.8
    return dict(foot0=foot0, foot_pos=np.ascontiguousarray(foot_pos), foot_id=np.ascontiguousarray(foot_id),
                swing_start=np.ascontiguousarray(swing_start), swing_end=np.ascontiguousarray(swing_end))


def reference_scenario_timeline():
    """The footsteps of TestLinearMpcZmp.cpp:30-43 as one timeline (n = 1)."""
    steps = reference_scenario_footsteps()
    return dict(foot0=np.array([[[0.0, 0.1], [0.0, -0.1]]]), foot_pos=np.array([[fs.pos for fs in steps]]),
                foot_id=np.array([[fs.foot for fs in steps]], dtype=np.int32),
                swing_start=np.array([[fs.swing_start_time for fs in steps]]),
                swing_end=np.array([[fs.swing_end_time for fs in steps]]))


# ------------------------------------------------------------------------------------------------ DdpZmp
class ComZmpSim3d:
    """SimModels.h:140-222: the 1-D CoM-ZMP model per horizontal axis, rebuilt every cycle with the current CoM height
    (:196), and the vertical double integrator under gravity (VerticalSimModel)."""

    def __init__(self, mass, sim_dt):
        self.sim_dt = sim_dt
        self.x = np.zeros(2)
        self.y = np.zeros(2)
        self.zsim = VerticalSim(mass, sim_dt)

    @property
    def z(self):
        return self.zsim.state

    def pos(self):
        return np.array([self.x[0], self.y[0], self.z[0]])

    def vel(self):
        return np.array([self.x[1], self.y[1], self.z[1]])

    def update(self, zmp, force_z):
        w = math.sqrt(G / self.z[0])
        ch, sh = math.cosh(w * self.sim_dt), math.sinh(w * self.sim_dt)
        Ad = np.array([[ch, sh / w], [w * sh, ch]])
        Bd = np.array([1 - ch, -w * sh])
        self.x = Ad @ self.x + Bd * zmp[0]
        self.y = Ad @ self.y + Bd * zmp[1]
        self.zsim.update(force_z)

    def addDisturb(self, impulse_per_mass):
        # SimModels.h:206-210 adds impulse.x() to BOTH axes (reference quirk, kept)
        self.x[1] += impulse_per_mass[0]
        self.y[1] += impulse_per_mass[0]


def sample_ddpzmp_refs(fm, t, N, dt, com_height=1.0):
    """RefData of TestDdpZmp.cpp:45-51 at t + i dt, i = 0..N: [N+1, 4] = (ref zmp x, y, 0, ref_com_height)."""
    ref = np.zeros((N + 1, 4))
    for i in range(N + 1):
        ref[i, :2] = fm.refZmp(t + i * dt)
        ref[i, 3] = com_height
    return ref


def run_closed_loop_ddpzmp(plan_once, mass=100.0, com_height=1.0, horizon_steps=100, horizon_dt=0.02, sim_dt=0.005,
                           end_time=10.0, disturb_times=(4.5, 8.5), disturb=(0.05, 0.05)):
    """The control loop of TestDdpZmp.cpp:70-125 around any
    `plan_once(ref [N+1,4], x0 [6], u_init [N,3]) -> u [N,3]` (the planned input sequence, warm start of the next
    cycle, :88-91).  Returns per-cycle records and the final state for the assertions of :108-109,:131-134."""
    fm = FootstepManager()
    for fs in reference_scenario_footsteps():
        fm.appendFootstep(fs)
    sim = ComZmpSim3d(mass, sim_dt)
    sim.z[0] = com_height
    t, log, u_list = 0.0, [], None
    while t < end_time:
        fm.update(t)
        p, v = sim.pos(), sim.vel()
        if u_list is None:
            u_list = np.tile(np.array([p[0], p[1], mass * G]), (horizon_steps, 1))
        x0 = np.array([p[0], v[0], p[1], v[1], p[2], v[2]])
        u_list = np.asarray(plan_once(sample_ddpzmp_refs(fm, t, horizon_steps, horizon_dt, com_height), x0, u_list))
        zmp, fz = u_list[0, :2].copy(), float(u_list[0, 2])
        log.append(dict(t=t, com=p, zmp=zmp, force_z=fz, ref_zmp=fm.refZmp(t)))
        t += sim_dt
        sim.update(zmp, fz)
        for dtm in disturb_times:
            if dtm <= t < dtm + sim_dt:
                sim.addDisturb(disturb)
                break
    return log, dict(t=t, com=sim.pos(), vel=sim.vel(), zmp=zmp, ref_zmp=fm.refZmp(t))


def make_ddpzmp_batch(n, horizon_steps=100, horizon_dt=0.02, mass=100.0, com_height=1.0, seed=20250928):
    """Synthetic DdpZmp workload: random evaluation times along the reference scenario of TestDdpZmp.cpp:32-44, CoM
    near the reference ZMP (+- 5 cm, +- 0.2 m/s, height +- 2 cm), warm start = (CoM xy, m g) as in :84-86.
    Returns dict(ref [n,N+1,4], x0 [n,6], u_init [n,N,3])."""
    rng = np.random.default_rng(seed)
    N = horizon_steps
    ref = np.zeros((n, N + 1, 4))
    x0 = np.zeros((n, 6))
    u_init = np.zeros((n, N, 3))
    cache = {}
    for k in range(n):
        t = round(float(rng.uniform(0.0, 9.0)) / 0.005) * 0.005
        if t not in cache:
            fm = FootstepManager()
            for fs in reference_scenario_footsteps():
                fm.appendFootstep(fs)
            tt = 0.0
            while tt < t - 1e-12:
                fm.update(tt)
                tt += 0.05
            fm.update(t)
            cache[t] = sample_ddpzmp_refs(fm, t, N, horizon_dt, com_height)
        ref[k] = cache[t]
        x0[k, 0] = ref[k, 0, 0] + rng.uniform(-0.05, 0.05)
        x0[k, 2] = ref[k, 0, 1] + rng.uniform(-0.05, 0.05)
        x0[k, 4] = com_height + rng.uniform(-0.02, 0.02)
        x0[k, 1], x0[k, 3], x0[k, 5] = rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(-0.05, 0.05)
        u_init[k, :, 0], u_init[k, :, 1], u_init[k, :, 2] = x0[k, 0], x0[k, 2], mass * G
    return dict(ref=ref, x0=x0, u_init=u_init)
