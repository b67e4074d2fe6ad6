"""CPU tests of the oracle (oracle/*.c) for the LinearMpcZmp path: pins it against the scipy-generated golden
vectors, certifies its QP answers by their KKT residuals, and replays the reference's own tests
(/root/reference/tests/src/TestInvariantSequentialExtension.cpp, TestLinearMpcZmp.cpp) on it."""
import numpy as np
import pytest

from centroidalcontrolcollection_amd import fixtures as fx
from oracle import oracle

G = 9.80665


def test_expm_matches_closed_form_for_nilpotent_model():
    # StateSpaceModel.h:195-203 on the jerk-input model: Ad, Bd have the closed form of SURVEY.md A.1
    dt = 0.0625
    A = np.zeros((3, 3))
    A[0, 1] = A[1, 2] = 1
    B = np.array([[0.0], [0.0], [1.0]])
    Ad, Bd, Ed = oracle.calc_disc_matrix(A, B, dt)
    assert np.allclose(Ad, [[1, dt, dt * dt / 2], [0, 1, dt], [0, 0, 1]], rtol=0, atol=1e-16)
    assert np.allclose(Bd[:, 0], [dt**3 / 6, dt**2 / 2, dt], rtol=0, atol=1e-17)
    assert np.all(Ed == 0)


def test_expm_general_matrix():
    from scipy.linalg import expm

    rng = np.random.default_rng(3)
    for n in (2, 5, 9):
        M = rng.normal(size=(n, n))
        ref = expm(M)
        assert np.abs(oracle.expm(M) - ref).max() <= 1e-12 * np.abs(ref).max()


def test_disc_matrix_with_offset_vector():
    # StateSpaceModel.h:205-214 (E != 0 branch) against the analytic ZOH of a double integrator with gravity
    dt, m = 0.01, 60.0
    A = np.array([[0.0, 1.0], [0.0, 0.0]])
    B = np.array([[0.0], [1.0 / m]])
    E = np.array([0.0, -G])
    Ad, Bd, Ed = oracle.calc_disc_matrix(A, B, dt, E)
    assert np.allclose(Ad, [[1, dt], [0, 1]], atol=1e-16)
    assert np.allclose(Bd[:, 0], [dt * dt / (2 * m), dt / m], atol=1e-18)
    assert np.allclose(Ed, [-G * dt * dt / 2, -G * dt], atol=1e-16)


def test_invariant_sequential_extension_identity():
    # TestInvariantSequentialExtension.cpp:64-97 (N=5, dt=0.01, x0=(1,2,3), u=(5,2.5,0,-1,-2)), output form:
    # row i of A_seq x0 + B_seq u must equal C x_{i+1} of the step-by-step rollout, tol 1e-10
    h, dt, N = 1.0, 0.01, 5
    o = oracle.LinearMpcZmp(h, N * dt, dt)
    assert o.horizon_steps == N
    A_seq, B_seq = o.seq()
    x = np.array([1.0, 2.0, 3.0])
    u = np.array([5.0, 2.5, 0.0, -1.0, -2.0])
    Ad = np.array([[1, dt, dt * dt / 2], [0, 1, dt], [0, 0, 1]])
    Bd = np.array([dt**3 / 6, dt**2 / 2, dt])
    C = np.array([1.0, 0.0, -h / G])
    y = A_seq @ x + B_seq @ u
    for i in range(N):
        x = Ad @ x + Bd * u[i]
        assert abs(C @ x - y[i]) < 1e-10


@pytest.mark.parametrize("N,dt", [(32, 0.0625), (100, 0.02)])
def test_seq_matrices_match_golden_model(golden_zmp, N, dt):
    o = oracle.LinearMpcZmp(1.0, N * dt, dt)
    assert o.horizon_steps == N
    A_seq, B_seq = o.seq()
    n = np.arange(N)
    b = dt**3 * (1 + 3 * n + 3 * n * n) / 6 - (1.0 / G) * dt
    assert np.allclose(B_seq[:, 0], b, rtol=0, atol=1e-15)
    assert np.allclose(A_seq[:, 2], ((n + 1) * dt)**2 / 2 - 1.0 / G, rtol=0, atol=1e-14)
    if N == 32:
        assert np.abs(A_seq - golden_zmp["n32_A_seq"]).max() < 1e-14
        assert np.abs(B_seq - golden_zmp["n32_B_seq"]).max() < 1e-15


def test_horizon_steps_is_ceil():
    # src/LinearMpcZmp.cpp:13
    assert oracle.LinearMpcZmp(1.0, 2.0, 0.0625).horizon_steps == 32
    assert oracle.LinearMpcZmp(1.0, 2.0, 0.02).horizon_steps == 100
    assert oracle.LinearMpcZmp(1.0, 1.0, 0.3).horizon_steps == 4


@pytest.mark.parametrize("key,N,dt", [("n32", 32, 0.0625), ("n100", 100, 0.02)])
def test_oracle_matches_golden(golden_zmp, key, N, dt):
    """The oracle against the independent scipy/BVLS known answers: ZMP to 1e-9 (north_star tolerance)."""
    o = oracle.LinearMpcZmp(1.0, N * dt, dt)
    r = o.plan_batch(golden_zmp[key + "_x0"], golden_zmp[key + "_zlim"], 0.005)
    assert np.all(r["status"] == 0)
    assert np.abs(r["zmp"] - golden_zmp[key + "_zmp"]).max() <= 1e-9
    jg = golden_zmp[key + "_jerk"]
    scale = np.maximum(1.0, np.abs(jg).max(axis=2, keepdims=True))
    assert (np.abs(r["jerk"] - jg) / scale).max() <= 1e-8


def _kkt_residuals(B_seq, A_seq, x0, zlim, jerk):
    """Certify u as THE minimiser of 1/2|u|^2 s.t. lo <= B u <= hi: returns (primal violation, dual residual).
    mu = B^-T u must be >= 0 exactly where B u sits on lo, <= 0 on hi, and 0 on rows strictly inside."""
    fr = A_seq @ x0
    lo, hi = zlim[0] - fr, zlim[1] - fr
    z = B_seq @ jerk
    primal = max((lo - z).max(), (z - hi).max(), 0.0)
    mu = np.linalg.solve(B_seq.T, jerk)
    on_lo = np.abs(z - lo) <= 1e-9
    on_hi = np.abs(z - hi) <= 1e-9
    scale = max(1.0, np.abs(mu).max())
    dual = 0.0
    dual = max(dual, np.abs(mu[~on_lo & ~on_hi]).max(initial=0.0) / scale)
    dual = max(dual, (-mu[on_lo & ~on_hi]).max(initial=0.0) / scale)
    dual = max(dual, (mu[on_hi & ~on_lo]).max(initial=0.0) / scale)
    return primal, dual


def test_oracle_kkt_self_certification():
    N, dt = 32, 0.0625
    o = oracle.LinearMpcZmp(1.0, N * dt, dt)
    A_seq, B_seq = o.seq()
    b = fx.make_zmp_batch(512, N, dt, seed=101)
    r = o.plan_batch(b["x0"], b["zlim"], 0.005)
    assert np.all(r["status"] == 0)
    worst_p = worst_d = 0.0
    for i in range(512):
        for ax in range(2):
            p, d = _kkt_residuals(B_seq, A_seq, b["x0"][i, ax], b["zlim"][i, ax], r["jerk"][i, ax])
            worst_p, worst_d = max(worst_p, p), max(worst_d, d)
    assert worst_p <= 1e-11
    assert worst_d <= 1e-7  # mu = B^-T u amplifies rounding by cond(B) ~ 2.5e3


def test_oracle_reference_closed_loop_properties():
    """TestLinearMpcZmp.cpp:15-126 replayed on the oracle (N=100, dt=0.02, sim_dt=0.005, 10 s, two kicks):
    planned ZMP inside the current limits every cycle (:86-87), ZMP and CoM inside at the end (:106-109)."""
    dt = 0.02
    o = oracle.LinearMpcZmp(1.0, 2.0, dt)
    N = o.horizon_steps

    def plan(ref, ip, t, cdt):
        zl = np.empty((2, 2, N))
        for i in range(N):
            r = ref(t + i * dt)
            for j in range(2):
                zl[0, j, i] = r.zmp_limits[j][0]
                zl[1, j, i] = r.zmp_limits[j][1]
        x0 = np.array([[[ip.pos[0], ip.vel[0], ip.acc[0]], [ip.pos[1], ip.vel[1], ip.acc[1]]]])
        out = o.plan_batch(x0, zl[None], cdt, want_jerk=False)
        assert out["status"][0] == 0
        return out["zmp"][0]

    log, fin = fx.run_closed_loop(plan)
    assert len(log) == 2000
    for rec in log:
        assert np.all(rec["zmp"] - rec["zmin"] >= 0) and np.all(rec["zmax"] - rec["zmp"] >= 0)
    assert np.all(fin["zmp"] - fin["zmin"] >= 0) and np.all(fin["zmax"] - fin["zmp"] >= 0)
    assert np.all(fin["com"] - fin["zmin"] >= 0) and np.all(fin["zmax"] - fin["com"] >= 0)


def test_qp_solver_with_equalities_and_bounds_against_scipy():
    """oracle_qp_solve in its general form (the QpCoeff convention of src/LinearMpcXY.cpp:134-181)."""
    from scipy.optimize import minimize

    rng = np.random.default_rng(5)
    for trial in range(12):
        n, me, mi = 12, 2, 9
        M = rng.normal(size=(n + 4, n))
        H = M.T @ M + 1e-3 * np.eye(n)
        g = rng.normal(size=n)
        Aeq = rng.normal(size=(me, n))
        xf = rng.uniform(0.2, 0.8, size=n)
        beq = Aeq @ xf
        Cin = rng.normal(size=(mi, n))
        din = Cin @ xf + rng.uniform(0.0, 0.5, size=mi)
        xl, xu = np.zeros(n), np.ones(n)
        x, rc, it, lam = oracle.qp_solve(H, g, Aeq, beq, Cin, din, xl, xu)
        assert rc == 0
        assert np.abs(Aeq @ x - beq).max() < 1e-10
        assert (Cin @ x - din).max() < 1e-10 and (xl - x).max() < 1e-10 and (x - xu).max() < 1e-10
        res = minimize(lambda v: 0.5 * v @ H @ v + g @ v, xf, jac=lambda v: H @ v + g, method="SLSQP",
                       bounds=list(zip(xl, xu)),
                       constraints=[dict(type="eq", fun=lambda v: Aeq @ v - beq, jac=lambda v: Aeq),
                                    dict(type="ineq", fun=lambda v: din - Cin @ v, jac=lambda v: -Cin)],
                       options=dict(ftol=1e-14, maxiter=500))
        f = lambda v: 0.5 * v @ H @ v + g @ v  # noqa: E731
        assert f(x) <= f(res.x) + 1e-8
        assert np.abs(x - res.x).max() < 1e-4


def test_qp_solver_reports_infeasible():
    H = np.eye(2)
    x, rc, it, lam = oracle.qp_solve(H, np.zeros(2), None, None, np.array([[1.0, 0.0], [-1.0, 0.0]]),
                                     np.array([-1.0, -1.0]))  # x0 <= -1 and x0 >= 1
    assert rc == 1


def test_timeline_generator_matches_stateful_footstep_manager():
    fm = fx.FootstepManager()
    steps = fx.reference_scenario_footsteps()
    for fs in steps:
        fm.appendFootstep(fs)
    foot0 = np.array([[[0, 0.1], [0, -0.1]]], float)
    fp = np.array([[s.pos for s in steps]])
    fid = np.array([[s.foot for s in steps]])
    ss = np.array([[s.swing_start_time for s in steps]])
    se = np.array([[s.swing_end_time for s in steps]])
    for t in np.arange(0, 10, 0.05):
        fm.update(t)
        ts = t + 0.02 * np.arange(100)
        ref = np.array([np.concatenate(fm.makeLinearMpcZmpRefData(x).zmp_limits) for x in ts])
        zmin, zmax = fx.zmp_limits_timeline(foot0, fp, fid, ss, se, (ts + 2e-6)[None])
        assert np.array_equal(ref[:, :2], zmin[0]) and np.array_equal(ref[:, 2:], zmax[0])


def test_append_footstep_rejects_overlap():
    # FootstepManager.h:216-220
    fm = fx.FootstepManager()
    fm.appendFootstep(fx.Footstep(fx.LEFT, (0.2, 0.1), 2.0, 0.2, 0.8))
    with pytest.raises(RuntimeError):
        fm.appendFootstep(fx.Footstep(fx.RIGHT, (0.4, -0.1), 2.5, 0.2, 0.8))
