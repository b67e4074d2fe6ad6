"""GPU parity tests of the LinearMpcXY HIP path (csrc/xy.hip) through the C-ABI.

Tolerance.  The QP Hessian H = B'WB + 1e-5 I has condition number 1e6..1e7 (thin force regularisation against a rank-4N
tracking term), so two exact solvers agree only to about cond * eps: even an exact KKT solve on the oracle's own
active set moves its force scales by 5e-10 relative.  The planned force scales are therefore compared to 1e-7 relative
to the largest force scale of the instance, the total wrench of the first step (what the reference's control loop
consumes) to 5e-9 relative to the robot's weight, and the vertical force -- an equality constraint -- to 1e-10.

Measured (profiles/r06_xy_parity_measured.json, the largest value over the 84 cases of this module; written by running
it with CCC_TEST_REPORT=<file>): force scales 7.5e-8 (head) / 7.0e-8 (whole horizon) against the ORACLE -- the same
kernels sit 2.2e-9 from the long-double golden vectors, so most of that distance is the oracle's own -- total force
1.1e-9, total moment 1.7e-9, 40-step horizons 2.2e-6 (bound 5e-6: the oracle's accuracy there, see the test)."""
import numpy as np
import pytest

from centroidalcontrolcollection_amd import LinearMpcXY
from centroidalcontrolcollection_amd import fixtures_ddp as fd

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["stream", "dual", "safeguard"])
def _xy_path(request, monkeypatch):
    """Every test of this module runs three ways: the stage-recursion (primal-dual active set) kernel that large batches
    take by default with the dual active-set kernel working off its hand-overs, the dual active-set kernel alone (small
    batches), and the stage-recursion kernel with its own single-change rounds as the safeguard (what problems beyond
    20 steps x 16 ridges take, forced here on the sizes every kernel solves)."""
    monkeypatch.setenv("CCC_XY_DUAL" if request.param == "dual" else "CCC_XY_STREAM", "1")
    if request.param == "safeguard":
        monkeypatch.setenv("CCC_XY_SAFEGUARD", "1")
    yield

LAM_RTOL = 1e-7
_MEASURED = {}


def _within(value, tol, what):
    """assert value <= tol, and remember the largest value seen per tolerance (written at module teardown when
    CCC_TEST_REPORT names a file: how the tolerances of this header were sized)."""
    value = float(value)
    _MEASURED[what] = max(_MEASURED.get(what, 0.0), value)
    assert value <= tol, (what, value, tol)


@pytest.fixture(scope="module", autouse=True)
def _report_measured():
    yield
    import json
    import os

    path = os.environ.get("CCC_TEST_REPORT")
    if path:
        with open(path, "w") as f:
            json.dump(_MEASURED, f, indent=1, sort_keys=True)


WIDE_GOLDEN_RTOL = 1e-7  # measured: 2.2e-9 (30 steps x 32 ridges), 2.7e-8 (40 steps); the oracle reaches 1.2e-6 there
WRENCH_RTOL = 5e-9


def _oracle():
    from oracle import oracle

    return oracle


def _compare(prob, r, o, N):
    assert np.all(r["status"] == 0) and np.all(o["status"] == 0)
    scale = np.abs(o["u0"]).max(axis=1, keepdims=True) + 1.0
    _within((np.abs(r["u0"] - o["u0"]) / scale).max(), LAM_RTOL, "u0")
    # total force / moment (about the origin) of the planned first-step forces
    for k in range(len(scale)):
        m0 = prob["dim"][k, 0]
        mg, fg = fd.total_wrench(prob["vertex"][k, 0], prob["ridge"][k, 0], r["u0"][k, :m0], np.zeros(3))
        mo, fo = fd.total_wrench(prob["vertex"][k, 0], prob["ridge"][k, 0], o["u0"][k, :m0], np.zeros(3))
        _within(np.abs(fg - fo).max() / (1.0 + np.abs(fo).max()), WRENCH_RTOL, "force")
        _within(np.abs(mg - mo).max() / (1.0 + np.abs(fo).max()), WRENCH_RTOL, "moment")
        assert abs(fg[2] - prob["total_force_z"][k, 0]) <= 1e-10 * prob["total_force_z"][k, 0]


def test_against_golden_vectors():
    """tests/golden/xy_golden.npz (make_golden_qp.py: independent model construction, primal active set, long-double KKT
    polish with a certificate): every force scale of the whole horizon within 5e-9 of the largest one of the instance.
    Measured 1.4e-9 on both kernels (which agree with each other to 1e-13): the golden data come from scipy's expm, the
    kernels' from the closed-form ZOH, and the last-bit differences of Ad / Bd are amplified by cond(H) ~ 1e7 -- that,
    cond(H) * eps, is the resolution of ANY double-precision statement of this QP.  The oracle itself reaches 5e-8
    against these vectors (it factorises the condensed Hessian), which is why they and not the oracle are the yardstick
    of this class."""
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "xy_golden.npz"))
    for tag, N in (("n20", 20), ("n15", 15)):
        prob = {k: g["%s_%s" % (tag, k)] for k in ("dim", "vertex", "ridge", "com_z", "total_force_z", "ref_out")}
        x0, lam = g[tag + "_x0"], g[tag + "_lambda"]
        r = LinearMpcXY(100.0, 0.1, N).planOnceBatch(prob, x0, want_all=True)
        assert np.all(r["status"] == 0)
        scale = np.abs(lam).reshape(len(x0), -1).max(axis=1)
        err = np.abs(r["lam"] - lam).reshape(len(x0), -1).max(axis=1)
        assert (err / scale).max() <= 5e-9, (tag, (err / scale).max())
        assert np.array_equal(r["u0"], r["lam"][:, 0, :])


@pytest.mark.parametrize("N,dt", [(15, 0.1), (20, 0.1)])
def test_parity_with_oracle(N, dt):
    """N = 15: the reference test's horizon (TestLinearMpcXY.cpp:17-19); N = 20: BASELINE.json configs[3]."""
    mass = 100.0
    prob, x0 = fd.make_xy_batch(96, N, dt, mass, seed=20250928)
    o = _oracle().LinearMpcXY(mass, dt, N).plan_batch(prob, x0, nthreads=8, want_all=True)
    r = LinearMpcXY(mass, dt, N).planOnceBatch(prob, x0, want_all=True)
    _compare(prob, r, o, N)
    # every QP variable, not only the returned head
    dims = prob["dim"]
    for k in range(96):
        lam_o = o["lam"][k, :dims[k].sum()]
        lam_g = np.concatenate([r["lam"][k, i, :dims[k, i]] for i in range(N)])
        _within(np.abs(lam_g - lam_o).max() / (1.0 + np.abs(lam_o).max()), LAM_RTOL, "lam")


def test_active_bounds_and_friction_limits():
    """A large velocity error drives many force scales to the 3 N lower bound (100+ pivots per instance)."""
    N, dt, mass = 12, 0.1, 100.0
    prob, x0 = fd.make_xy_batch(48, N, dt, mass, seed=3)
    x0[:, 1] += 40.0
    o = _oracle().LinearMpcXY(mass, dt, N).plan_batch(prob, x0, nthreads=8)
    r = LinearMpcXY(mass, dt, N).planOnceBatch(prob, x0, want_all=True)
    _compare(prob, r, o, N)
    assert r["lam"][prob["dim"][:, :, None] > np.arange(16)[None, None, :]].min() >= 3.0 - 1e-9
    assert (np.abs(r["lam"] - 3.0) < 1e-9).sum() > 100  # bounds really are active
    # equality rows: sum_r rho_z lambda = total_force_z on every step
    fz = (r["lam"] * prob["ridge"][..., 2]).sum(axis=2)
    assert np.abs(fz - prob["total_force_z"]).max() <= 1e-8


def test_steps_without_contact_are_skipped():
    N, dt, mass = 8, 0.1, 100.0
    prob = fd.xy_problem(0.0, N, dt)
    prob["dim"][0, 3:5] = 0
    x0 = np.array([[mass * 1.0, 5.0, 0.0, 0.0, 0.0, 0.0]])
    o = _oracle().LinearMpcXY(mass, dt, N).plan_batch(prob, x0)
    r = LinearMpcXY(mass, dt, N).planOnceBatch(prob, x0, want_all=True)
    _compare(prob, r, o, N)
    assert np.all(r["lam"][0, 3:5] == 0.0)


def _random_geometry_batch(n, N, dt, mass, seed):
    """Steps with 0 / 4 / 8 / 12 / 16 ridges (vertices dropped from the rect contact, i.e. point / line / partial
    contacts and the two-contact case of two 2-vertex feet), random vertical force and CoM height per step."""
    rng = np.random.default_rng(seed)
    prob, x0 = fd.make_xy_batch(n, N, dt, mass, seed=seed)
    for k in range(n):
        for i in range(N):
            nv = rng.choice([0, 1, 2, 3, 4, 4, 4])
            prob["dim"][k, i] = 4 * nv
            prob["vertex"][k, i, 4 * nv:] = 0.0
            prob["ridge"][k, i, 4 * nv:] = 0.0
        if prob["dim"][k, 0] == 0:  # keep an output to compare
            prob["dim"][k, 0] = 16
            V, R = fd.contact_from_rect((0.9, -0.15), (1.1, 0.15))
            prob["vertex"][k, 0], prob["ridge"][k, 0] = V, R
    prob["total_force_z"] *= rng.uniform(0.8, 1.2, size=prob["total_force_z"].shape)
    prob["com_z"] *= rng.uniform(0.9, 1.1, size=prob["com_z"].shape)
    return prob, x0


def test_random_contact_dimensions_and_weights():
    """Ragged inputs: per-step ridge counts in {0,4,8,12,16}, varying f_z and c_z, non-default weights (momentum
    weights switched on: all 6N output rows carry weight)."""
    N, dt, mass, n = 14, 0.08, 80.0, 64
    prob, x0 = _random_geometry_batch(n, N, dt, mass, seed=11)
    w = dict(w_lmi=(2.0, 0.5), w_lm=(0.1, 0.3), w_am=(0.7, 1.5), w_force=3e-5)
    o = _oracle().LinearMpcXY(mass, dt, N, **w).plan_batch(prob, x0, nthreads=8, want_all=True)
    wp = LinearMpcXY.WeightParam(w["w_lmi"], w["w_lm"], w["w_am"], w["w_force"])
    r = LinearMpcXY(mass, dt, N, wp).planOnceBatch(prob, x0, want_all=True)
    ok = o["status"] == 0
    assert ok.sum() >= n // 2 and np.all(r["status"][ok] == 0)
    # instances the oracle finds infeasible (f_z outside what the remaining ridges can carry) must not report success
    assert np.all(r["status"][~ok] != 0)
    sel = np.where(ok)[0]
    sub = {k2: v[sel] for k2, v in prob.items()}
    _compare(sub, {k2: (v[sel] if v is not None else None) for k2, v in r.items()},
             {k2: (v[sel] if v is not None else None) for k2, v in o.items()}, N)
    import os

    if os.environ.get("CCC_XY_DUAL"):  # the dual active-set kernel follows the oracle pivot by pivot
        assert np.all(r["pivots"][sel] == o["iters"][sel])
    for k in sel:
        lam_o = o["lam"][k, :prob["dim"][k].sum()]
        lam_g = np.concatenate([r["lam"][k, i, :prob["dim"][k, i]] for i in range(N)])
        _within(np.abs(lam_g - lam_o).max() / (1.0 + np.abs(lam_o).max()), LAM_RTOL, "lam")
        assert np.all(r["lam"][k][prob["dim"][k][:, None] <= np.arange(16)[None, :]] == 0.0)


def test_infeasible_instance_is_flagged():
    """f_z below what the 3 N lower bounds already produce: no feasible point; status must be non-zero (never a
    silently wrong plan), and the neighbouring instances of the batch are unaffected."""
    N, dt, mass = 10, 0.1, 100.0
    prob, x0 = fd.make_xy_batch(3, N, dt, mass, seed=2)
    prob["total_force_z"][1, 4] = 1.0
    o = _oracle().LinearMpcXY(mass, dt, N).plan_batch(prob, x0)
    r = LinearMpcXY(mass, dt, N).planOnceBatch(prob, x0)
    assert o["status"][1] != 0 and r["status"][1] != 0
    assert r["status"][0] == 0 and r["status"][2] == 0
    scale = np.abs(o["u0"][[0, 2]]).max() + 1.0
    assert np.abs(r["u0"][[0, 2]] - o["u0"][[0, 2]]).max() <= LAM_RTOL * scale


def test_reference_closed_loop_through_planonce():
    """TestLinearMpcXY.cpp:15-160 through planOnce(motion_param_func, ref_data_func, initial_param, t): per-cycle and
    final property assertions (:126-128, :140-142)."""
    N, dt, mass = 15, 0.1, 100.0
    mpc = LinearMpcXY(mass, dt, N)

    def motion(t):
        rmin, rmax, _ = fd.xy_reference_schedule(t)
        return LinearMpcXY.MotionParam(1.0, mass * fd.G, [fd.contact_from_rect(rmin, rmax)])

    def ref(t):
        return LinearMpcXY.RefData(fd.xy_reference_schedule(t)[2])

    sim = fd.CentroidalSim(mass, (40.0, 20.0, 10.0), 0.05)
    sim.pos = np.array([1.0, 0.0, 1.0])
    t = 0.0
    while t < 8.0:
        ip = LinearMpcXY.InitialParam(sim.pos[:2], sim.vel[:2], sim.ang_mom[:2])
        scales = mpc.planOnce(motion, ref, ip, t)
        V, R = motion(t).contact_list[0]
        moment, force = fd.total_wrench(V, R, scales, sim.pos)
        refp = np.array([*ref(t).pos, 1.0])
        assert np.linalg.norm(sim.pos - refp) < 2.0 and np.linalg.norm(sim.vel) < 2.0
        assert np.linalg.norm(sim.ang_mom) < 5.0
        t += 0.05
        sim.update(force, moment)
    refp = np.array([*ref(t).pos, 1.0])
    assert np.linalg.norm(sim.pos - refp) < 0.1 and np.linalg.norm(sim.vel) < 0.1 and np.linalg.norm(sim.ang_mom) < 0.1


def test_device_entry_and_determinism():
    import torch

    N, dt, mass, n = 20, 0.1, 100.0, 600
    prob, x0 = fd.make_xy_batch(n, N, dt, mass, seed=5)
    mpc = LinearMpcXY(mass, dt, N)
    dev = torch.device("cuda:0")
    tp = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in prob.items()}
    tx0 = torch.from_numpy(x0).to(dev)
    u1 = torch.zeros((n, 16), dtype=torch.float64, device=dev)
    u2 = torch.zeros_like(u1)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    mpc.plan_batch_device(tp, tx0, u1, status=st)
    mpc.plan_batch_device(tp, tx0, u2)
    torch.cuda.synchronize()
    assert torch.equal(u1, u2)
    assert np.all((st.cpu().numpy() & 0xff) == 0)
    host = mpc.planOnceBatch(prob, x0)
    assert np.array_equal(host["u0"], u1.cpu().numpy())


def test_cpp_header_shim_matches_python_mirror():
    """Host C++ against include/CCC/LinearMpcXY.h (examples/plan_once_linear_mpc_xy.cpp): same kernel, same sampled
    inputs as the Python mirror -> identical force scales, for planOnce and planOnceBatch."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "plan_once_linear_mpc_xy")
    if not os.path.exists(exe):
        import __graft_entry__

        __graft_entry__.build()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    N, dt, mass = 15, 0.1, 100.0
    mpc = LinearMpcXY(mass, dt, N)

    def motion(t):
        rmin, rmax, _ = fd.xy_reference_schedule(t)
        return LinearMpcXY.MotionParam(1.0, mass * fd.G, [fd.contact_from_rect(rmin, rmax)])

    def ref(t):
        return LinearMpcXY.RefData(fd.xy_reference_schedule(t)[2])

    ip = LinearMpcXY.InitialParam((1.01, -0.02), (0.05, 0.0))
    for ln, t in zip(lines[:3], (0.0, 2.45, 4.3)):
        assert ln.startswith("t=%.2f dim=16" % t)
        cpp = np.array([float(v) for v in ln.split("u0=")[1].split()])
        u = mpc.planOnce(motion, ref, ip, t)
        assert np.array_equal(cpp, u)
        bl = [b for b in lines[3:] if b.startswith("batch[%d]" % (0.0, 2.45, 4.3).index(t))][0]
        assert float(bl.split("u0[0]=")[1]) == u[0]
    # walking with two foot contacts in double support over 30 steps: both front ends route to the 32-slot handle
    def foot(x, y):
        return fd.contact_from_rect((x - 0.1, y - 0.05), (x + 0.1, y + 0.05))

    def wmotion(t):
        ph = int((t + 1e-9) / 0.5)
        xl, xr = 1.0 + 0.4 * ((ph + 1) // 4), 1.2 + 0.4 * max((ph + 3) // 4 - 1, 0)
        cl = [foot(xl, 0.1), foot(xr, -0.1)] if ph % 2 == 0 else ([foot(xl, 0.1)] if ph % 4 == 1 else [foot(xr, -0.1)])
        return LinearMpcXY.MotionParam(1.0, mass * fd.G, cl)

    wl = [ln for ln in lines if ln.startswith("walking")][0]
    uw = LinearMpcXY(mass, dt, 30).planOnce(wmotion, lambda t: LinearMpcXY.RefData((1.1 + 0.2 * t, 0.0)),
                                            LinearMpcXY.InitialParam((1.1, 0.01), (0.0, 0.0)), 0.0)
    assert "dim=32" in wl and "status=0" in wl and len(uw) == 32
    assert np.array_equal(np.array([float(v) for v in wl.split("u0=")[1].split()]), uw)
    # multi-contact: feet + the right hand on a wall for the first second (48 ridges): both front ends route to the 64-slot handle
    lfoot, rfoot = fd.contact_from_rect((0.9, 0.05), (1.1, 0.15)), fd.contact_from_rect((0.9, -0.15), (1.1, -0.05))
    Vl, Rl = fd.contact_from_rect((-0.05, -0.05), (0.05, 0.05))
    hand = (np.stack([1.45 - Vl[:, 2], -0.2 + Vl[:, 0], 1.0 - Vl[:, 1]], axis=1), np.stack([-Rl[:, 2], Rl[:, 0], -Rl[:, 1]], axis=1))
    ml = [ln for ln in lines if ln.startswith("multicontact")][0]
    um = LinearMpcXY(mass, dt, 20).planOnce(
        lambda t: LinearMpcXY.MotionParam(0.9, mass * fd.G, [lfoot, rfoot, hand] if t + 1e-9 < 1.0 else [lfoot, rfoot]),
        lambda t: LinearMpcXY.RefData((1.0 + 0.02 * t, 0.0)), LinearMpcXY.InitialParam((1.01, -0.01), (0.0, 0.0)), 0.0)
    assert "dim=48" in ml and "status=0" in ml and len(um) == 48
    assert np.array_equal(np.array([float(v) for v in ml.split("u0=")[1].split()]), um)


def test_rounds_of_the_stage_recursion_kernel_are_bit_identical(monkeypatch):
    """The stage-recursion kernel runs in rounds: the instances still changing their clamped set when a round ends are
    repacked into whole wavefronts and go on from the set they had.  The iteration is a function of the set alone, so any
    split into rounds gives the same answers, bit for bit, and the same iteration counts.  (With the early hand-over of
    wandering instances off: which kernel finishes such an instance depends on where the second round ends, and the two
    kernels agree to 1e-9, not bit for bit -- checked below.)"""
    monkeypatch.delenv("CCC_XY_DUAL", raising=False)
    monkeypatch.setenv("CCC_XY_STREAM", "1")
    monkeypatch.setenv("CCC_XY_WANDER", "0")
    prob, x0 = fd.make_xy_batch(700, 20, 0.1, seed=21)
    res = []
    for rounds in ("99", "6,10", "2,4,6", "1", "3,4,5"):
        monkeypatch.setenv("CCC_XY_ROUNDS", rounds)  # (development switches are read when the handle is created)
        res.append(LinearMpcXY(100.0, 0.1, 20).planOnceBatch(prob, x0, want_all=True))
    assert np.all(res[0]["status"] == 0)
    for r in res[1:]:
        assert np.array_equal(r["u0"], res[0]["u0"]) and np.array_equal(r["lam"], res[0]["lam"])
        assert np.array_equal(r["pivots"], res[0]["pivots"])
    # the default: instances that still change a tenth of their variables after ten iterations go to the dual kernel
    # beside the last round -- the same minimisers
    monkeypatch.setenv("CCC_XY_ROUNDS", "6,10")
    monkeypatch.delenv("CCC_XY_WANDER")
    early = LinearMpcXY(100.0, 0.1, 20).planOnceBatch(prob, x0, want_all=True)
    assert np.all(early["status"] == 0)
    scale = 1.0 + np.abs(res[0]["lam"]).max()
    assert np.abs(early["lam"] - res[0]["lam"]).max() <= 1e-7 * scale


def test_history_order_of_the_first_round_is_a_schedule_only(monkeypatch):
    """Round 5: a handle keeps the sweeps every instance of its last call took and runs the first round of a call of the
    same size in that order, longest first, so that the lanes of a wavefront stop together (csrc/xy.hip
    ccc_xy_plan_batch_device, common.hip order_by_count).  An instance's arithmetic does not depend on its lane: the
    repeated batch, another batch of the same size (a history that predicts nothing), another size and the way back all
    give what a handle without a history gives, bit for bit -- inputs, multipliers and iteration counts."""
    import torch

    monkeypatch.delenv("CCC_XY_DUAL", raising=False)
    N, dt, mass, n = 20, 0.1, 100.0, 5000
    pA, xA = fd.make_xy_batch(n, N, dt, mass, seed=31)
    xB = xA * 1.05 + 0.01
    dev = torch.device("cuda:0")
    tp = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in pA.items()}

    def run(m, x, k=n):
        tk = {a: v[:k].contiguous() for a, v in tp.items()}
        tx = torch.from_numpy(np.ascontiguousarray(x[:k])).to(dev)
        u = torch.zeros((k, 16), dtype=torch.float64, device=dev)
        lam = torch.zeros((k, N, 16), dtype=torch.float64, device=dev)
        st = torch.zeros(k, dtype=torch.int32, device=dev)
        m.plan_batch_device(tk, tx, u, lambda_all=lam, status=st)
        torch.cuda.synchronize()
        return u.cpu().numpy(), lam.cpu().numpy(), st.cpu().numpy()

    monkeypatch.setenv("CCC_XY_HISTORY", "0")
    ref = LinearMpcXY(mass, dt, N)
    monkeypatch.delenv("CCC_XY_HISTORY")
    rA, rB, rC = run(ref, xA), run(ref, xB), run(ref, xA, n - 129)
    assert np.all((rA[2] & 0xff) == 0)
    sweeps = rA[2] >> 8
    assert sweeps.min() <= 2 and np.percentile(sweeps, 90) >= 6  # (there is something to order)
    m = LinearMpcXY(mass, dt, N)
    for got, want in ((run(m, xA), rA), (run(m, xA), rA), (run(m, xB), rB), (run(m, xA, n - 129), rC), (run(m, xA), rA),
                      (run(m, xA), rA)):
        for a, b in zip(got, want):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("env", [{"CCC_XY_DUAL": "1"}, {"CCC_XY_PDAS_ITERS": "1"}, {"CCC_XY_PDAS_ITERS": "3"},
                                 {"CCC_XY_STREAM": "1", "CCC_XY_ROUNDS": "2,5,9"}])
def test_dual_kernel_and_fallback_list(env):
    """The stage-recursion (primal-dual active set) kernel is the default; the dual active-set kernel stays as its
    fallback.  In a subprocess with the development switches: the dual kernel alone, and the stage-recursion kernel starved
    of iterations so that most instances go through the work list -- same answers as the oracle either way."""
    import os
    import subprocess
    import sys

    code = (
        "import numpy as np\n"
        "from centroidalcontrolcollection_amd import LinearMpcXY, fixtures_ddp as fd\n"
        "from oracle import oracle\n"
        "prob, x0 = fd.make_xy_batch(200, 20, 0.1, seed=9)\n"
        "o = oracle.LinearMpcXY(100.0, 0.1, 20).plan_batch(prob, x0, nthreads=8)\n"
        "r = LinearMpcXY(100.0, 0.1, 20).planOnceBatch(prob, x0)\n"
        "assert np.all(r['status'] == 0) and np.all(o['status'] == 0)\n"
        "print(np.abs(r['u0'] - o['u0']).max() / (1.0 + np.abs(o['u0']).max()))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root,
                         env=dict({k: v for k, v in os.environ.items() if not k.startswith("CCC_XY_")}, PYTHONPATH=root, **env))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert float(out.stdout.strip().splitlines()[-1]) <= 1e-7


# ---------------------------------------------------------------------------------------------------------------------
# Contact lists beyond one surface contact / horizons beyond 20 steps (src/LinearMpcXY.cpp:69-82, :126-133, :149-176)
@pytest.mark.parametrize("N,M,n", [(30, 32, 40), (20, 32, 64), (40, 16, 40)])
def test_double_support_and_long_horizons_against_the_oracle(N, M, n):
    """Walking sequences with two separate foot contacts in double support (32 ridges per step), single support (16),
    flight steps (no variables, no equality row) and horizons of 30 / 40 steps: the stage-recursion kernel with its
    single-change safeguard rounds, against the oracle's dense dual active set on the same QP."""
    if M == 32:
        prob, x0 = fd.make_xy_walking_batch(n, N, 0.1, M=32, seed=5)
        assert prob["dim"].max() == 32 and (prob["dim"] == 0).any()
    else:
        prob, x0 = fd.make_xy_batch(n, N, 0.1, seed=5)
    mpc = LinearMpcXY(100.0, 0.1, N, max_ridges=M)
    r = mpc.planOnceBatch(prob, x0, want_all=True)
    o = _oracle().LinearMpcXY(100.0, 0.1, N, M=M).plan_batch(prob, x0, nthreads=16, want_all=True)
    _compare(prob, r, o, N)
    # ... and every force scale of the horizon, inside the bounds.  Tolerance: the ORACLE's accuracy at these sizes -- its
    # condensed Hessian is worse conditioned the longer the horizon (1.2e-6 of the largest force scale against the
    # certified golden vectors at 40 steps, tests/test_golden_qp.py); the kernel itself is held to those vectors below
    for k in range(n):
        lam = np.concatenate([r["lam"][k, i, :prob["dim"][k, i]] for i in range(N)])
        ref = o["lam"][k][:len(lam)]
        _within(np.abs(lam - ref).max() / (1.0 + np.abs(ref).max()), 5e-6, "lam_long_horizon")
        assert lam.min() >= 3.0 - 1e-9 and lam.max() <= 3.0 * 100.0 * 9.80665 + 1e-6


def test_against_wide_golden_vectors():
    """tests/golden/xy_wide_golden.npz: 32-ridge double-support walking over 30 steps, the reference scenario over 40
    (640-670 variables, certified in long double): every force scale of the horizon."""
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "xy_wide_golden.npz"))
    for tag, N in (("w30", 30), ("l40", 40)):
        prob = {k: g["%s_%s" % (tag, k)] for k in ("dim", "vertex", "ridge", "com_z", "total_force_z", "ref_out")}
        x0, lam = g[tag + "_x0"], g[tag + "_lambda"]
        r = LinearMpcXY(100.0, 0.1, N, max_ridges=lam.shape[2]).planOnceBatch(prob, x0, want_all=True)
        assert np.all(r["status"] == 0)
        scale = np.abs(lam).reshape(len(x0), -1).max(axis=1)
        err = np.abs(r["lam"] - lam).reshape(len(x0), -1).max(axis=1)
        print(tag, "max rel err vs golden", (err / scale).max())
        assert (err / scale).max() <= WIDE_GOLDEN_RTOL, (tag, (err / scale).max())


def test_safeguard_rounds_take_the_instances_that_cycle(monkeypatch):
    """Starved of block iterations (CCC_XY_PDAS_ITERS=2) nearly every instance goes through the single-change rounds from
    the clamped set it was handed over with: same answers as the oracle, statuses solved, and more set changes counted
    than the block iteration needs."""
    monkeypatch.delenv("CCC_XY_DUAL", raising=False)
    monkeypatch.setenv("CCC_XY_STREAM", "1")
    monkeypatch.setenv("CCC_XY_SAFEGUARD", "1")
    prob, x0 = fd.make_xy_batch(300, 20, 0.1, seed=13)
    mpc = LinearMpcXY(100.0, 0.1, 20)
    full = mpc.planOnceBatch(prob, x0, want_all=True)
    monkeypatch.setenv("CCC_XY_PDAS_ITERS", "2")  # (development switches are read when the handle is created)
    starved = LinearMpcXY(100.0, 0.1, 20).planOnceBatch(prob, x0, want_all=True)
    o = _oracle().LinearMpcXY(100.0, 0.1, 20).plan_batch(prob, x0, nthreads=16)
    _compare(prob, starved, o, 20)
    assert np.all(starved["status"] == 0) and starved["pivots"].mean() > full["pivots"].mean()
    scale = 1.0 + np.abs(full["lam"]).max(axis=(1, 2), keepdims=True)
    _within((np.abs(starved["lam"] - full["lam"]) / scale).max(), LAM_RTOL, "lam_starved_vs_full")


def test_more_than_two_contacts_per_step_64_ridge_slots():
    """Feet + hands on walls (48 and 64 ridges per step; src/LinearMpcXY.cpp:69-82 takes any contact_list; VERDICT round 2,
    item 7): max_ridges = 64, the stage-recursion kernel with a 128-bit clamped set per stage.  Checked against the
    oracle's dense dual active set and, solver-independently, by the KKT residuals of the restated QP."""
    N, n = 12, 48
    prob, x0 = fd.make_xy_multicontact_batch(n, N, 0.1, seed=4)
    assert set(np.unique(prob["dim"])) >= {48, 64}
    mpc = LinearMpcXY(100.0, 0.1, N, max_ridges=64)
    r = mpc.planOnceBatch(prob, x0, want_all=True)
    assert np.all(r["status"] == 0)
    eq, viol, stat, ok = _xy_kkt_residuals(prob, x0, r["lam"])
    assert ok.all() and eq.max() <= 1e-10 and viol.max() <= 1e-9
    assert np.median(stat) <= 1e-7 and stat.max() <= 2e-5
    o = _oracle().LinearMpcXY(100.0, 0.1, N, M=64).plan_batch(prob, x0, nthreads=16, want_all=True)
    _compare(prob, r, o, N)
    # the hands carry force somewhere in the batch
    four = prob["dim"] == 64
    assert four.any() and (r["lam"][four][:, 32:] > 3.0 + 1e-6).any()
    # a 16-slot problem through the 64-slot handle gives the 16-slot handle's plan
    p16, x16 = fd.make_xy_batch(40, 20, 0.1, seed=9)
    wide = dict(p16)
    for key in ("vertex", "ridge"):
        a = np.zeros(p16[key].shape[:2] + (64, 3))
        a[:, :, :16] = p16[key]
        wide[key] = a
    r16 = LinearMpcXY(100.0, 0.1, 20).planOnceBatch(p16, x16)
    r64 = LinearMpcXY(100.0, 0.1, 20, max_ridges=64).planOnceBatch(wide, x16)
    assert np.all(r64["status"] == 0)
    assert np.abs(r64["u0"][:, :16] - r16["u0"]).max() <= 1e-7 * (1.0 + np.abs(r16["u0"]).max())


def test_xy_limits_are_reported():
    from centroidalcontrolcollection_amd import _lib

    with pytest.raises(_lib.CccError) as e:
        LinearMpcXY(100.0, 0.1, 20, max_ridges=24)
    assert e.value.code == _lib.CCC_ERR_UNSUPPORTED
    with pytest.raises(_lib.CccError) as e:
        LinearMpcXY(100.0, 0.1, 257)
    assert e.value.code == _lib.CCC_ERR_UNSUPPORTED


@pytest.mark.parametrize("n", [1, 63, 65])
def test_wide_configuration_ragged_batches(n):
    """32 ridge slots: batches that do not fill a wavefront, and one that spills into a second (the stage-recursion
    kernel is the only path here, whatever the batch size)."""
    prob, x0 = fd.make_xy_walking_batch(65, 22, 0.1, M=32, seed=11)
    mpc = LinearMpcXY(100.0, 0.1, 22, max_ridges=32)
    full = mpc.planOnceBatch(prob, x0, want_all=True)
    part = mpc.planOnceBatch({k: v[:n] for k, v in prob.items()}, x0[:n], want_all=True)
    assert np.all(part["status"] == 0)
    assert np.array_equal(part["u0"], full["u0"][:n]) and np.array_equal(part["lam"], full["lam"][:n])


def _xy_kkt_residuals(prob, x0, lam, mass=100.0, dt=0.1, w_force=1e-5, Wdiag=(1.0, 0.0, 1.0, 0.0, 1.0, 1.0)):
    """Solver-independent KKT residuals of the LinearMpcXY QP for a whole batch, numpy only, nothing condensed: the
    states are SIMULATED with the closed-form ZOH of src/LinearMpcXY.cpp:59-83 (A^3 = 0: Ad = I + A dt + A^2 dt^2/2,
    Bd = B dt + A B dt^2/2 + A^2 B dt^3/6), the gradient of 1/2 sum |x_{i+1} - ref_i|^2_W + w/2 |lam|^2
    (:141-147, extend_for_output = false) comes from the adjoint recursion, the multiplier of each step's equality
    sum rho_z lam = total_force_z (:149-176) is fitted on the step's free variables.  Returns per instance
    (equality residual / f_z, bound violation, stationarity residual relative to the gradient scale)."""
    n, N, M = lam.shape
    G = 9.80665
    lo, hi = 3.0, 3.0 * mass * G  # force_range_, src/LinearMpcXY.cpp:93
    W = np.asarray(Wdiag)
    fz, cz = prob["total_force_z"], prob["com_z"]
    V, R, dim = prob["vertex"], prob["ridge"], prob["dim"]
    mask = np.arange(M)[None, None, :] < dim[:, :, None]
    lam = lam * mask
    A = np.zeros((n, N, 6, 6))
    A[:, :, 0, 1] = 1.0
    A[:, :, 2, 3] = 1.0
    A[:, :, 4, 2] = -fz / mass
    A[:, :, 5, 0] = fz / mass
    B = np.zeros((n, N, 6, M))
    B[:, :, 1] = R[..., 0]
    B[:, :, 3] = R[..., 1]
    B[:, :, 4] = -(V[..., 2] - cz[..., None]) * R[..., 1] + V[..., 1] * R[..., 2]
    B[:, :, 5] = (V[..., 2] - cz[..., None]) * R[..., 0] - V[..., 0] * R[..., 2]
    B = B * mask[:, :, None, :]
    A2 = A @ A
    Ad = np.eye(6) + A * dt + A2 * (dt * dt / 2)
    Bd = B * dt + (A @ B) * (dt * dt / 2) + (A2 @ B) * (dt ** 3 / 6)
    x = np.zeros((n, N + 1, 6))
    x[:, 0] = x0
    for i in range(N):
        x[:, i + 1] = np.einsum("nab,nb->na", Ad[:, i], x[:, i]) + np.einsum("nar,nr->na", Bd[:, i], lam[:, i])
    grad = np.zeros_like(lam)
    p = np.zeros((n, 6))
    for i in range(N - 1, -1, -1):
        p = W * (x[:, i + 1] - prob["ref_out"][:, i]) + (np.einsum("nab,na->nb", Ad[:, i + 1], p) if i + 1 < N else 0.0)
        grad[:, i] = w_force * lam[:, i] + np.einsum("nar,na->nr", Bd[:, i], p)
    rz = R[..., 2] * mask
    eq = np.abs((rz * lam).sum(axis=2) - fz * (dim > 0)).max(axis=1) / fz.max(axis=1)
    viol = np.maximum(np.where(mask, lo - lam, 0.0), np.where(mask, lam - hi, 0.0)).max(axis=(1, 2))
    tol = 1e-9 * hi
    at_lo, at_hi = mask & (lam <= lo + tol), mask & (lam >= hi - tol)
    free = mask & ~at_lo & ~at_hi
    # nu_i: least squares of grad + nu rho_z = 0 over the free variables of the step (a step keeps at least one)
    num = -(grad * rz * free).sum(axis=2)
    den = (rz * rz * free).sum(axis=2)
    nu = np.where(den > 0, num / np.where(den > 0, den, 1.0), 0.0)
    res = grad + nu[:, :, None] * rz
    stat = np.where(free, np.abs(res), np.where(at_lo, np.maximum(-res, 0.0), np.where(at_hi, np.maximum(res, 0.0), 0.0)))
    gscale = np.abs(grad).max(axis=(1, 2)) + 1e-300
    return eq, viol, stat.max(axis=(1, 2)) / gscale, (den > 0) | (dim == 0)


def test_config4_full_size_kkt_properties():
    """BASELINE.json configs[3] AT FULL SIZE: LinearMpcXY, batch 65536, N = 20.  From the planned force scales of the whole
    horizon alone: every instance solved, bounds respected (1e-9 N), every contact step's vertical-force equality met to 1e-10,
    and the KKT stationarity residual (adjoint gradient + fitted equality multipliers, wrong-signed part at the bounds)
    below 1e-7 of the gradient's scale -- a strictly convex QP has one KKT point, so this pins the answer itself."""
    n, N, base = 65536, 20, 2048
    pb, xb = fd.make_xy_batch(base, N, 0.1, seed=20250928)
    prob = {k: np.concatenate([v] * (n // base)) for k, v in pb.items()}
    rng = np.random.default_rng(4)
    x0 = np.concatenate([xb] * (n // base)) + rng.uniform(-1.0, 1.0, size=(n, 6)) * np.array([2.0, 5.0, 2.0, 5.0, 0.3, 0.3])
    r = LinearMpcXY(100.0, 0.1, N).planOnceBatch(prob, x0, want_all=True)
    assert np.all(r["status"] == 0)
    assert np.array_equal(r["u0"], r["lam"][:, 0, :])
    eq, viol, stat, has_free = _xy_kkt_residuals(prob, x0, r["lam"])
    print("config 4: eq %.2e  bound violation %.2e  stationarity max %.2e p99.9 %.2e p99 %.2e median %.2e" % (
        eq.max(), viol.max(), stat.max(), np.percentile(stat, 99.9), np.percentile(stat, 99), np.median(stat)))
    assert np.all(has_free)
    assert eq.max() <= 1e-10 and viol.max() <= 1e-9
    # (relative to the gradient's scale; the Hessian's condition number is 1e6 .. 1e7, the CPU oracle's own answers sit at
    #  6e-8 on this measure.  Measured on MI355X: median 1e-8, max 3e-6 over 65536 instances)
    assert np.median(stat) <= 1e-7 and np.percentile(stat, 99.9) <= 2e-6 and stat.max() <= 2e-5


def test_walking_and_multi_contact_bench_size_kkt_properties():
    """The KKT check of the test above on the workloads beyond BASELINE's configs, at bench size: double-support walking
    (32 ridge slots, 30 steps, batch 8192) and feet + hands multi-contact steps (64 ridge slots, 20 steps, batch 2048)."""
    for name, (prob, x0), N, M in (("walking", fd.make_xy_walking_batch(8192, 30, 0.1, M=32, seed=20250928), 30, 32),
                                   ("multi-contact", fd.make_xy_multicontact_batch(2048, 20, 0.1, seed=20250928), 20, 64)):
        r = LinearMpcXY(100.0, 0.1, N, max_ridges=M).planOnceBatch(prob, x0, want_all=True)
        assert np.all(r["status"] == 0), name
        eq, viol, stat, has_free = _xy_kkt_residuals(prob, x0, r["lam"])
        print("%s: eq %.2e  bound violation %.2e  stationarity max %.2e p99 %.2e median %.2e" % (
            name, eq.max(), viol.max(), stat.max(), np.percentile(stat, 99), np.median(stat)))
        assert np.all(has_free), name
        assert eq.max() <= 1e-10 and viol.max() <= 1e-9, name
        # (30 steps: the Hessian's condition number grows with the horizon -- measured median 1.4e-7, max 3.6e-6 for walking)
        assert np.median(stat) <= 5e-7 and stat.max() <= 5e-5, name
