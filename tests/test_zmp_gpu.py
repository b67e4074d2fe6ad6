"""GPU parity tests of the LinearMpcZmp HIP path (csrc/zmp.hip) -- all calls go through the C-ABI.

Bar (BASELINE.json north_star): per-instance ZMP within 1e-9 of the CPU oracle, fp64."""
import numpy as np
import pytest

from centroidalcontrolcollection_amd import LinearMpcZmp, _lib
from centroidalcontrolcollection_amd import fixtures as fx

pytestmark = pytest.mark.gpu

ZMP_TOL = 1e-9  # north_star: "per-instance CoM/ZMP within 1e-9 of the CPU reference"
JERK_RTOL = 1e-7  # relative to max(1, |jerk|_inf) of the instance: the QP solution itself


def _oracle():
    from oracle import oracle

    return oracle


def _jerk_err(a, b):
    scale = np.maximum(1.0, np.abs(b).max(axis=-1, keepdims=True))
    return (np.abs(a - b) / scale).max()


@pytest.fixture(scope="module")
def mpc32():
    return LinearMpcZmp(1.0, 2.0, 0.0625)


def test_model_matrices_match_oracle(mpc32):
    o = _oracle().LinearMpcZmp(1.0, 2.0, 0.0625)
    assert mpc32.horizon_steps_ == o.horizon_steps == 32
    A, B = mpc32.seq()
    Ao, Bo = o.seq()
    assert np.abs(A - Ao).max() <= 1e-15 and np.abs(B - Bo).max() <= 1e-16


def test_golden_vectors_n32(mpc32, golden_zmp):
    r = mpc32.planOnceBatch(golden_zmp["n32_x0"], golden_zmp["n32_zlim"], 0.005, want_jerk=True)
    assert np.all(r["status"] == _lib.CCC_STATUS_SOLVED)
    assert np.abs(r["zmp"] - golden_zmp["n32_zmp"]).max() <= ZMP_TOL
    assert _jerk_err(r["jerk"], golden_zmp["n32_jerk"]) <= JERK_RTOL


def test_parity_with_oracle_config2_batch4096(mpc32):
    """BASELINE.json configs[1]: LinearMpcZmp batch=4096 random footstep sequences, N=32, fp64."""
    b = fx.make_zmp_batch(4096, 32, 0.0625, seed=20250928)
    ref = _oracle().LinearMpcZmp(1.0, 2.0, 0.0625).plan_batch(b["x0"], b["zlim"], 0.005, nthreads=8)
    r = mpc32.planOnceBatch(b["x0"], b["zlim"], 0.005, want_jerk=True)
    assert np.all(ref["status"] == 0) and np.all(r["status"] == 0)
    assert np.abs(r["zmp"] - ref["zmp"]).max() <= ZMP_TOL
    assert _jerk_err(r["jerk"], ref["jerk"]) <= JERK_RTOL


def test_default_control_dt_is_horizon_dt(mpc32):
    # src/LinearMpcZmp.cpp:72-75
    b = fx.make_zmp_batch(64, 32, 0.0625, seed=3)
    a = mpc32.planOnceBatch(b["x0"], b["zlim"], -1.0)["zmp"]
    c = mpc32.planOnceBatch(b["x0"], b["zlim"], 0.0625)["zmp"]
    assert np.array_equal(a, c)
    ref = _oracle().LinearMpcZmp(1.0, 2.0, 0.0625).plan_batch(b["x0"], b["zlim"], -1.0)
    assert np.abs(a - ref["zmp"]).max() <= ZMP_TOL


def test_ragged_batches_and_empty(mpc32):
    b = fx.make_zmp_batch(131, 32, 0.0625, seed=9)
    full = mpc32.planOnceBatch(b["x0"], b["zlim"], 0.005)["zmp"]
    for n in (1, 2, 3, 63, 64, 65, 131):
        part = mpc32.planOnceBatch(b["x0"][:n], b["zlim"][:n], 0.005)["zmp"]
        assert np.array_equal(part, full[:n])
    empty = mpc32.planOnceBatch(np.zeros((0, 2, 3)), np.zeros((0, 2, 2, 32)), 0.005)
    assert empty["zmp"].shape == (0, 2)


def test_unconstrained_and_fully_clamped_instances(mpc32):
    N = 32
    x0 = np.zeros((2, 2, 3))
    zlim = np.empty((2, 2, 2, N))
    zlim[0, :, 0], zlim[0, :, 1] = -10.0, 10.0  # nothing active: jerk = 0
    zlim[1, :, 0], zlim[1, :, 1] = 0.02, 0.02 + 1e-9  # every row active (32 active constraints)
    r = mpc32.planOnceBatch(x0, zlim, 0.005, want_jerk=True)
    ref = _oracle().LinearMpcZmp(1.0, 2.0, 0.0625).plan_batch(x0, zlim, 0.005)
    assert np.all(r["status"] == 0)
    assert np.all(r["jerk"][0] == 0.0) and np.all(r["pivots"][0] == 0)
    assert np.abs(r["zmp"] - ref["zmp"]).max() <= ZMP_TOL
    assert _jerk_err(r["jerk"], ref["jerk"]) <= 1e-6


def test_infeasible_instance_is_flagged(mpc32):
    b = fx.make_zmp_batch(8, 32, 0.0625, seed=5)
    zlim = b["zlim"].copy()
    zlim[3, 1, 0, 7] = zlim[3, 1, 1, 7] + 0.5  # zmin > zmax on the y axis of instance 3
    r = mpc32.planOnceBatch(b["x0"], zlim, 0.005)
    assert r["status"][3, 1] == _lib.CCC_STATUS_INFEASIBLE
    assert r["status"].sum() == _lib.CCC_STATUS_INFEASIBLE
    ok = mpc32.planOnceBatch(b["x0"], b["zlim"], 0.005)
    keep = np.ones((8, 2), bool)
    keep[3, 1] = False
    assert np.array_equal(r["zmp"][keep], ok["zmp"][keep])


def test_device_entry_full_size_properties(mpc32):
    """BASELINE metric size (batch 65536) through the device-pointer entry: size-independent properties
    (primal feasibility, clamp, translation equivariance, determinism) + oracle parity on a sample."""
    import torch

    n, N = 65536, 32
    b = fx.make_zmp_batch(n, N, 0.0625, seed=20250928)
    dev = torch.device("cuda:0")
    x0 = torch.from_numpy(b["x0"]).to(dev)
    zlim = torch.from_numpy(b["zlim"]).to(dev)
    zmp = torch.empty((n, 2), dtype=torch.float64, device=dev)
    jerk = torch.empty((n, 2, N), dtype=torch.float64, device=dev)
    status = torch.empty((n, 2), dtype=torch.int32, device=dev)
    mpc32.plan_batch_device(x0, zlim, 0.005, zmp, jerk, status)
    torch.cuda.synchronize()
    st = status.cpu().numpy()
    assert np.all((st & 0xff) == 0)
    zmp_h, jerk_h = zmp.cpu().numpy(), jerk.cpu().numpy()
    # primal feasibility of the whole planned sequence: lo <= A_seq x0 + B_seq u <= hi
    A_seq, B_seq = mpc32.seq()
    pred = np.einsum("ik,nak->nai", A_seq, b["x0"]) + np.einsum("ij,naj->nai", B_seq, jerk_h)
    assert (b["zlim"][:, :, 0, :] - pred).max() <= 1e-9 and (pred - b["zlim"][:, :, 1, :]).max() <= 1e-9
    # planned ZMP inside the step-0 limits (src/LinearMpcZmp.cpp:78)
    assert np.all(zmp_h >= b["zlim"][:, :, 0, 0]) and np.all(zmp_h <= b["zlim"][:, :, 1, 0])
    # determinism: a second launch gives the same bits
    zmp2 = torch.empty_like(zmp)
    mpc32.plan_batch_device(x0, zlim, 0.005, zmp2)
    torch.cuda.synchronize()
    assert torch.equal(zmp, zmp2)
    # translation equivariance: shifting CoM position and all limits by c shifts the planned ZMP by c
    shift = 0.37
    x0s = x0.clone()
    x0s[:, :, 0] += shift
    zmp3 = torch.empty_like(zmp)
    mpc32.plan_batch_device(x0s, zlim + shift, 0.005, zmp3)
    torch.cuda.synchronize()
    assert (zmp3 - shift - zmp).abs().max().item() <= 1e-9
    # oracle parity on a strided sample of 2048 instances
    sel = np.arange(0, n, 32)
    ref = _oracle().LinearMpcZmp(1.0, 2.0, 0.0625).plan_batch(b["x0"][sel], b["zlim"][sel], 0.005, nthreads=8)
    assert np.abs(zmp_h[sel] - ref["zmp"]).max() <= ZMP_TOL
    assert _jerk_err(jerk_h[sel], ref["jerk"]) <= JERK_RTOL


def test_n50_one_axis_per_wavefront_path():
    """32 < N <= 64: one QP per wavefront, the rows in registers (zmp_plan_kernel_w since round 5; before: the packed LDS
    tableau with a one-wavefront workgroup)."""
    dt = 0.04
    mpc = LinearMpcZmp(1.0, 2.0, dt)
    assert mpc.horizon_steps_ == 50
    b = fx.make_zmp_batch(512, 50, dt, seed=17)
    ref = _oracle().LinearMpcZmp(1.0, 2.0, dt).plan_batch(b["x0"], b["zlim"], 0.005, nthreads=8)
    r = mpc.planOnceBatch(b["x0"], b["zlim"], 0.005, want_jerk=True)
    assert np.all(r["status"] == 0)
    assert np.abs(r["zmp"] - ref["zmp"]).max() <= ZMP_TOL
    assert _jerk_err(r["jerk"], ref["jerk"]) <= JERK_RTOL


def test_short_horizon_n5():
    mpc = LinearMpcZmp(1.0, 0.05, 0.01)
    assert mpc.horizon_steps_ == 5
    b = fx.make_zmp_batch(256, 5, 0.01, seed=23)
    ref = _oracle().LinearMpcZmp(1.0, 0.05, 0.01).plan_batch(b["x0"], b["zlim"], 0.005)
    r = mpc.planOnceBatch(b["x0"], b["zlim"], 0.005, want_jerk=True)
    assert np.abs(r["zmp"] - ref["zmp"]).max() <= ZMP_TOL


def test_golden_vectors_n100_block_kernel(golden_zmp):
    """The reference test's horizon (2 s @ 20 ms = 100 steps): packed-tableau kernel, 104 rows, three workgroups per CU."""
    mpc = LinearMpcZmp(1.0, 2.0, 0.02)
    assert mpc.horizon_steps_ == 100
    r = mpc.planOnceBatch(golden_zmp["n100_x0"], golden_zmp["n100_zlim"], 0.005, want_jerk=True)
    assert np.all(r["status"] == _lib.CCC_STATUS_SOLVED)
    assert np.abs(r["zmp"] - golden_zmp["n100_zmp"]).max() <= ZMP_TOL
    assert _jerk_err(r["jerk"], golden_zmp["n100_jerk"]) <= JERK_RTOL
    b = fx.make_zmp_batch(300, 100, 0.02, seed=29)
    ref = _oracle().LinearMpcZmp(1.0, 2.0, 0.02).plan_batch(b["x0"], b["zlim"], 0.005, nthreads=8)
    r = mpc.planOnceBatch(b["x0"], b["zlim"], 0.005, want_jerk=True)
    assert np.all(r["status"] == 0)
    assert np.abs(r["zmp"] - ref["zmp"]).max() <= ZMP_TOL
    assert _jerk_err(r["jerk"], ref["jerk"]) <= JERK_RTOL


@pytest.mark.parametrize("N", [40, 56, 64, 80, 100, 112, 128])
def test_register_tile_kernel_against_the_oracle_and_the_lds_tableau(N, monkeypatch):
    """csrc/zmp_k2r.inc (round 5): the packed symmetric tableau of K2 kept in registers across pivots, the default for
    48 < N <= 128.  Every size it is built for, with two and with three tiles per thread: solved, planned ZMP and jerk
    sequence within the parity tolerances of the oracle, no more pivots than the LDS tableau (the same iteration; since
    round 6 K2r enters rows by the dual gain and skips the ratio test on the pivots that cannot drop a row) with ZMPs within 1e-12."""
    dt = 2.0 / N
    b = fx.make_zmp_batch(384, N, dt, seed=41 + N)
    ref = _oracle().LinearMpcZmp(1.0, 2.0, dt).plan_batch(b["x0"], b["zlim"], 0.005, nthreads=8)
    monkeypatch.setenv("CCC_ZMP_K2", "0")  # (development switches are read when a handle is created)
    monkeypatch.setenv("CCC_ZMP_KW", "0")  # (32 < N <= 64 would otherwise run the one-QP-per-wavefront register kernel)
    lds = LinearMpcZmp(1.0, 2.0, dt)
    assert lds.horizon_steps_ == N
    r0 = lds.planOnceBatch(b["x0"], b["zlim"], 0.005, want_jerk=True)
    assert lds.last_kernel() == "zmp_plan_sym_kernel"
    for k2 in ("12", "13") if N > 48 else ("1",):
        monkeypatch.setenv("CCC_ZMP_K2", k2)
        mpc = LinearMpcZmp(1.0, 2.0, dt)
        r = mpc.planOnceBatch(b["x0"], b["zlim"], 0.005, want_jerk=True)
        assert mpc.last_kernel() == "zmp_plan_reg_kernel"
        assert np.all(r["status"] == 0)
        assert np.abs(r["zmp"] - ref["zmp"]).max() <= ZMP_TOL
        assert _jerk_err(r["jerk"], ref["jerk"]) <= JERK_RTOL
        # (round 6: K2r enters rows by the dual objective's gain, K1's rule, instead of by the largest violation: fewer
        #  pivots than the LDS tableau, the same minimiser)
        assert r["pivots"].sum() <= r0["pivots"].sum() and np.abs(r["zmp"] - r0["zmp"]).max() <= 1e-12
    monkeypatch.delenv("CCC_ZMP_K2")
    monkeypatch.delenv("CCC_ZMP_KW")
    d = LinearMpcZmp(1.0, 2.0, dt)
    d.planOnceBatch(b["x0"][:8], b["zlim"][:8], 0.005)
    assert d.last_kernel() == ("zmp_plan_reg_kernel" if N > 64 else "zmp_plan_kernel_w")


@pytest.mark.parametrize("N", [33, 40, 48, 50, 56, 64])
def test_one_qp_per_wavefront_register_kernel(N):
    """zmp_plan_kernel_w (round 5): K1's iteration -- the sections of csrc/zmp_k1.inc -- with ONE QP per wavefront and the
    rows in three or four sixteen-double register tuples, the default for 32 < N <= 64 (48 columns up to N = 48, 64 beyond).
    Solved everywhere, planned ZMP and jerk sequence within the parity tolerances of the oracle, deterministic."""
    dt = 2.0 / N
    b = fx.make_zmp_batch(1024, N, dt, seed=7 + N)
    ref = _oracle().LinearMpcZmp(1.0, 2.0, dt).plan_batch(b["x0"], b["zlim"], 0.005, nthreads=8)
    mpc = LinearMpcZmp(1.0, 2.0, dt)
    assert mpc.horizon_steps_ == N
    r = mpc.planOnceBatch(b["x0"], b["zlim"], 0.005, want_jerk=True)
    assert mpc.last_kernel() == "zmp_plan_kernel_w"
    assert np.all(r["status"] == 0)
    assert np.abs(r["zmp"] - ref["zmp"]).max() <= ZMP_TOL
    assert _jerk_err(r["jerk"], ref["jerk"]) <= JERK_RTOL
    r2 = mpc.planOnceBatch(b["x0"], b["zlim"], 0.005, want_jerk=True)
    assert np.array_equal(r2["zmp"], r["zmp"]) and np.array_equal(r2["jerk"], r["jerk"])


def test_n200_packed_lds_tableau():
    """BASELINE.json configs[0] as worded (2 s horizon @ dt = 10 ms = 200 steps): the largest horizon whose packed
    symmetric tableau (2 x 2 tiles) fits the 160 KB of LDS.  Parity with the oracle, and a short stretch of the reference
    closed loop."""
    mpc = LinearMpcZmp(1.0, 2.0, 0.01)
    assert mpc.horizon_steps_ == 200
    b = fx.make_zmp_batch(40, 200, 0.01, seed=31)
    ref = _oracle().LinearMpcZmp(1.0, 2.0, 0.01).plan_batch(b["x0"], b["zlim"], 0.005, nthreads=8)
    r = mpc.planOnceBatch(b["x0"], b["zlim"], 0.005, want_jerk=True)
    assert np.all(r["status"] == 0)
    assert np.abs(r["zmp"] - ref["zmp"]).max() <= ZMP_TOL
    assert _jerk_err(r["jerk"], ref["jerk"]) <= JERK_RTOL
    log, fin = fx.run_closed_loop(mpc.planOnce, end_time=3.0)
    for rec in log:
        assert np.all(rec["zmp"] - rec["zmin"] >= 0) and np.all(rec["zmax"] - rec["zmp"] >= 0)


def test_n400_config1_at_5_ms_and_the_horizon_limit():
    """BASELINE.json configs[0] as worded with the reference's control period as horizon step: 2 s @ 5 ms = 400 steps (the
    HBM-resident tableau at 512 rows, round 4 -- VERDICT r3 item 7).  Parity with the oracle; beyond 512 steps the
    constructor refuses (CCC_ERR_UNSUPPORTED), never a silent truncation."""
    mpc = LinearMpcZmp(1.0, 2.0, 0.005)
    assert mpc.horizon_steps_ == 400
    b = fx.make_zmp_batch(12, 400, 0.005, seed=41)
    ref = _oracle().LinearMpcZmp(1.0, 2.0, 0.005).plan_batch(b["x0"], b["zlim"], 0.005, nthreads=8)
    r = mpc.planOnceBatch(b["x0"], b["zlim"], 0.005, want_jerk=True)
    assert np.all(r["status"] == 0)
    assert np.abs(r["zmp"] - ref["zmp"]).max() <= ZMP_TOL
    assert _jerk_err(r["jerk"], ref["jerk"]) <= JERK_RTOL
    with pytest.raises(_lib.CccError):
        LinearMpcZmp(1.0, 2.0, 0.0025)  # 800 steps: not built


@pytest.mark.parametrize("N", [36, 47, 64, 72, 90, 112, 128, 150, 180, 230])
def test_horizon_sizes_against_oracle(N):
    """One parity case per kernel instantiation: packed LDS tableau with 4 x 4 tiles (40..192 rows), and the HBM-resident
    tableau beyond 200 steps."""
    dt = 2.0 / N
    mpc = LinearMpcZmp(1.0, 2.0, dt)
    assert mpc.horizon_steps_ == N
    b = fx.make_zmp_batch(24, N, dt, seed=100 + N)
    ref = _oracle().LinearMpcZmp(1.0, 2.0, dt).plan_batch(b["x0"], b["zlim"], 0.005, nthreads=8)
    r = mpc.planOnceBatch(b["x0"], b["zlim"], 0.005, want_jerk=True)
    assert np.all(r["status"] == 0)
    assert np.abs(r["zmp"] - ref["zmp"]).max() <= ZMP_TOL
    assert _jerk_err(r["jerk"], ref["jerk"]) <= JERK_RTOL


def _stage_handle(monkeypatch, T, dt, **env):
    """A handle whose N > 32 calls take KS (csrc/zmp_stage.inc) whatever the batch (development switches are read at create)."""
    monkeypatch.setenv("CCC_ZMP_STAGE", "1")
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    mpc = LinearMpcZmp(1.0, T, dt)
    monkeypatch.delenv("CCC_ZMP_STAGE")
    for k in env:
        monkeypatch.delenv(k)
    return mpc


def test_stage_kernel_golden_vectors_and_oracle_n100(golden_zmp, monkeypatch):
    """KS (round 6, csrc/zmp_stage.inc): the QP in its state-space form -- Riccati recursion per guessed set of clamped
    stages, primal-dual active set on the guess, one QP per lane -- at the reference test's horizon (2 s @ 20 ms,
    tests/src/TestLinearMpcZmp.cpp:17-19).  Golden vectors and the oracle on 1200 QPs, the jerks of the whole horizon."""
    mpc = _stage_handle(monkeypatch, 2.0, 0.02)
    r = mpc.planOnceBatch(golden_zmp["n100_x0"], golden_zmp["n100_zlim"], 0.005, want_jerk=True)
    assert mpc.last_kernel() == "zmp_plan_stage_kernel"
    assert np.all(r["status"] == _lib.CCC_STATUS_SOLVED)
    assert np.abs(r["zmp"] - golden_zmp["n100_zmp"]).max() <= ZMP_TOL
    assert _jerk_err(r["jerk"], golden_zmp["n100_jerk"]) <= JERK_RTOL
    b = fx.make_zmp_batch(600, 100, 0.02, seed=31)
    ref = _oracle().LinearMpcZmp(1.0, 2.0, 0.02).plan_batch(b["x0"], b["zlim"], 0.005, nthreads=8)
    r = mpc.planOnceBatch(b["x0"], b["zlim"], 0.005, want_jerk=True)
    assert np.all(r["status"] == 0)
    assert np.abs(r["zmp"] - ref["zmp"]).max() <= ZMP_TOL
    assert _jerk_err(r["jerk"], ref["jerk"]) <= JERK_RTOL
    # the exact kernel alone gives the same plan (its own route to the same minimiser)
    monkeypatch.setenv("CCC_ZMP_STAGE", "0")
    ex = LinearMpcZmp(1.0, 2.0, 0.02)
    monkeypatch.delenv("CCC_ZMP_STAGE")
    r0 = ex.planOnceBatch(b["x0"], b["zlim"], 0.005, want_jerk=True)
    assert ex.last_kernel() == "zmp_plan_reg_kernel"
    assert np.abs(r["zmp"] - r0["zmp"]).max() <= 1e-11 and _jerk_err(r["jerk"], r0["jerk"]) <= 1e-9


@pytest.mark.parametrize("iters", [1, 3, 12])
def test_stage_kernel_hands_over_what_it_does_not_finish(iters, monkeypatch):
    """Starved of iterations KS hands most QPs to the exact kernel in the same call (the list and its length stay on the
    device): every QP comes back solved with the same plan, whoever solved it."""
    b = fx.make_zmp_batch(500, 100, 0.02, seed=7)
    ref = _oracle().LinearMpcZmp(1.0, 2.0, 0.02).plan_batch(b["x0"], b["zlim"], 0.005, nthreads=8)
    mpc = _stage_handle(monkeypatch, 2.0, 0.02, CCC_ZMP_STAGE_ITERS=iters)
    for _ in range(2):  # (the second call reuses the list)
        r = mpc.planOnceBatch(b["x0"], b["zlim"], 0.005, want_jerk=True)
        assert mpc.last_kernel() == "zmp_plan_stage_kernel"
        assert np.all(r["status"] == 0)
        assert np.abs(r["zmp"] - ref["zmp"]).max() <= ZMP_TOL
        assert _jerk_err(r["jerk"], ref["jerk"]) <= JERK_RTOL


def test_stage_kernel_certificate_on_a_long_horizon(monkeypatch):
    """5 s of horizon (100 steps of 50 ms): the cost-to-go of a run of clamped stages grows like exp(2 w T) = 4e13, the
    recursion's gains lose their digits, and the QP itself is that badly conditioned (jerks to 5e6 in this batch: holding
    the ZMP for 5 s of the inverted pendulum's exp(w t)).  KS returns only what passes its certificate (limits, multiplier
    signs, stationarity residual, all measured on the point itself): those answers are held to the usual 1e-9 of the
    oracle.  What it hands over is the exact kernel's, and there the exact kernel and the oracle -- two dual active sets
    on the same 1e13-conditioned tableau -- agree to 2.5e-6 (measured on this batch with KS off; not KS's doing)."""
    b = fx.make_zmp_batch(400, 100, 0.05, seed=5)
    ref = _oracle().LinearMpcZmp(1.0, 5.0, 0.05).plan_batch(b["x0"], b["zlim"], 0.005, nthreads=8)
    mpc = _stage_handle(monkeypatch, 5.0, 0.05)
    r = mpc.planOnceBatch(b["x0"], b["zlim"], 0.005, want_jerk=True)
    ok = (ref["status"] == 0)[:, None] & np.ones((1, 2), bool)
    assert np.all(r["status"][ok] == 0)
    certified = ok & (r["pivots"] <= 20)  # (KS's iteration limit; the exact kernel needs ~100 pivots on these)
    print("certified by KS: %d of %d QPs" % (certified.sum(), ok.sum()))
    assert certified.sum() >= 0.25 * ok.sum()
    err = np.abs(r["zmp"] - ref["zmp"])
    assert err[certified].max() <= ZMP_TOL
    scale = np.maximum(1.0, np.abs(ref["jerk"]).max(axis=-1))
    assert (np.abs(r["jerk"] - ref["jerk"]).max(axis=-1) / scale)[certified].max() <= JERK_RTOL
    assert err[ok].max() <= 1e-5


@pytest.mark.parametrize("N", [40, 72, 128, 160, 256, 400])
def test_stage_kernel_other_horizons_and_their_exact_kernels(N, monkeypatch):
    """KS takes any horizon; what it hands over goes to the exact kernel of that size (register tiles to 128 rows, the LDS
    tableau to 200, the HBM tableau beyond).  CCC_ZMP_STAGE_ITERS=6 makes sure each of them gets work."""
    dt = 2.0 / N
    n = (48 if N > 256 else 96) if N > 128 else 256
    b = fx.make_zmp_batch(n, N, dt, seed=300 + N)
    ref = _oracle().LinearMpcZmp(1.0, 2.0, dt).plan_batch(b["x0"], b["zlim"], 0.005, nthreads=8)
    for env in ({}, {"CCC_ZMP_STAGE_ITERS": 6}):
        mpc = _stage_handle(monkeypatch, 2.0, dt, **env)
        assert mpc.horizon_steps_ == N
        r = mpc.planOnceBatch(b["x0"], b["zlim"], 0.005, want_jerk=True)
        assert mpc.last_kernel() == "zmp_plan_stage_kernel"
        assert np.all(r["status"] == 0)
        assert np.abs(r["zmp"] - ref["zmp"]).max() <= ZMP_TOL
        assert _jerk_err(r["jerk"], ref["jerk"]) <= JERK_RTOL


def test_stage_kernel_flags_an_infeasible_qp_and_ragged_batches(monkeypatch):
    """lo > hi at one stage: INFEASIBLE for that axis alone; batches that do not fill a wavefront."""
    mpc = _stage_handle(monkeypatch, 2.0, 0.02)
    for n in (1, 31, 33, 97):
        b = fx.make_zmp_batch(n, 100, 0.02, seed=n)
        ref = _oracle().LinearMpcZmp(1.0, 2.0, 0.02).plan_batch(b["x0"], b["zlim"], 0.005)
        r = mpc.planOnceBatch(b["x0"], b["zlim"], 0.005)
        assert np.all(r["status"] == 0) and np.abs(r["zmp"] - ref["zmp"]).max() <= ZMP_TOL
    b = fx.make_zmp_batch(5, 100, 0.02, seed=2)
    b["zlim"][2, 1, 0, 40] = b["zlim"][2, 1, 1, 40] + 0.1
    ref = _oracle().LinearMpcZmp(1.0, 2.0, 0.02).plan_batch(b["x0"], b["zlim"], 0.005)
    r = mpc.planOnceBatch(b["x0"], b["zlim"], 0.005)
    assert r["status"][2, 1] != 0 and r["status"][2, 0] == 0 and ref["status"][2] != 0
    keep = [0, 1, 3, 4]
    assert np.all(r["status"][keep] == 0) and np.abs(r["zmp"][keep] - ref["zmp"][keep]).max() <= ZMP_TOL


def test_stage_kernel_random_horizons_heights_and_control_periods(monkeypatch):
    """Twenty random (horizon, CoM height, control period, seed) combinations through KS: ZMP and jerk parity with the oracle
    (the model constants, the penalty 30 (g / h)^3, the certificate's bound and the flag words all depend on them), ragged
    batches that leave lanes of the last wavefront without a QP."""
    rng = np.random.default_rng(777)
    for _ in range(20):
        N = int(rng.integers(33, 257))
        h = float(rng.uniform(0.5, 1.4))
        dt = 2.0 / N
        cdt = float(rng.choice([-1.0, 0.002, 0.005, dt]))
        n = int(rng.integers(40, 200))
        monkeypatch.setenv("CCC_ZMP_STAGE", "1")
        mpc = LinearMpcZmp(h, 2.0, dt)
        monkeypatch.delenv("CCC_ZMP_STAGE")
        N = mpc.horizon_steps_
        b = fx.make_zmp_batch(n, N, dt, com_height=h, seed=int(rng.integers(1, 10**6)))
        ref = _oracle().LinearMpcZmp(h, 2.0, dt).plan_batch(b["x0"], b["zlim"], cdt, nthreads=8)
        r = mpc.planOnceBatch(b["x0"], b["zlim"], cdt, want_jerk=True)
        assert mpc.last_kernel() == "zmp_plan_stage_kernel", (N, h)
        assert np.all(r["status"] == 0), (N, h, cdt)
        assert np.abs(r["zmp"] - ref["zmp"]).max() <= ZMP_TOL, (N, h, cdt)
        assert _jerk_err(r["jerk"], ref["jerk"]) <= JERK_RTOL, (N, h, cdt)


def test_stage_kernel_is_the_default_for_large_batches_of_long_horizons():
    mpc = LinearMpcZmp(1.0, 2.0, 0.02)
    b = fx.make_zmp_batch(8192, 100, 0.02, seed=77)
    r = mpc.planOnceBatch(b["x0"], b["zlim"], 0.005)
    assert mpc.last_kernel() == "zmp_plan_stage_kernel" and np.all(r["status"] == 0)
    s = fx.make_zmp_batch(64, 100, 0.02, seed=78)
    mpc.planOnceBatch(s["x0"], s["zlim"], 0.005)
    assert mpc.last_kernel() == "zmp_plan_reg_kernel"
    ref = _oracle().LinearMpcZmp(1.0, 2.0, 0.02).plan_batch(b["x0"][:512], b["zlim"][:512], 0.005, nthreads=8)
    assert np.abs(r["zmp"][:512] - ref["zmp"]).max() <= ZMP_TOL


def test_reference_scenario_n100_closed_loop():
    """BASELINE.json configs[0]: TestLinearMpcZmp.cpp:15-126 exactly (2 s horizon @ 20 ms, sim_dt 5 ms, 10 s, two
    kicks) through planOnce(callback, ...): the reference's per-cycle and final property assertions, and the
    final CoM agrees with the oracle-driven replay."""
    mpc = LinearMpcZmp(1.0, 2.0, 0.02)
    log, fin = fx.run_closed_loop(mpc.planOnce, end_time=10.0)
    assert len(log) == 2000
    for rec in log:
        assert np.all(rec["zmp"] - rec["zmin"] >= 0) and np.all(rec["zmax"] - rec["zmp"] >= 0)
    assert np.all(fin["zmp"] - fin["zmin"] >= 0) and np.all(fin["zmax"] - fin["zmp"] >= 0)
    assert np.all(fin["com"] - fin["zmin"] >= 0) and np.all(fin["zmax"] - fin["com"] >= 0)
    assert np.abs(fin["com"] - np.array([0.64215196, -0.05244137])).max() < 1e-6  # oracle replay, SURVEY.md App. C


def test_plan_once_callback_surface_closed_loop():
    """The reference's own test scenario (TestLinearMpcZmp.cpp:15-126) through planOnce(callback, ...), with a
    50-step horizon (2 s @ 40 ms): property assertions of :86-87 and :106-109."""
    mpc = LinearMpcZmp(1.0, 2.0, 0.04)
    log, fin = fx.run_closed_loop(mpc.planOnce, end_time=10.0)
    for rec in log:
        assert np.all(rec["zmp"] - rec["zmin"] >= 0) and np.all(rec["zmax"] - rec["zmp"] >= 0)
    assert np.all(fin["zmp"] - fin["zmin"] >= 0) and np.all(fin["zmax"] - fin["zmp"] >= 0)
    assert np.all(fin["com"] - fin["zmin"] >= 0) and np.all(fin["zmax"] - fin["com"] >= 0)


def test_bad_shapes_raise(mpc32):
    with pytest.raises(ValueError):
        mpc32.planOnceBatch(np.zeros((4, 2, 3)), np.zeros((4, 2, 2, 31)))


def test_cpp_header_shim_reference_scenario():
    """Host C++ against include/CCC/LinearMpcZmp.h (examples/closed_loop_linear_mpc_zmp.cpp): the reference's
    TestLinearMpcZmp scenario at its own horizon (N=100) and at the headline horizon (N=32)."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "closed_loop_linear_mpc_zmp")
    if not os.path.exists(exe):
        import __graft_entry__

        __graft_entry__.build()
    for dt, steps in (("0.02", 100), ("0.0625", 32)):
        out = subprocess.run([exe, dt], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "horizon_steps=%d" % steps in out.stdout and "violations=0" in out.stdout and "com_inside=1" in out.stdout
        assert "1d_matches_2d=1" in out.stdout  # CCC::LinearMpcZmp1d = one axis of CCC::LinearMpcZmp, bit for bit
    # same closed loop through the Python mirror: final CoM agrees (same kernels, same inputs)
    fin_cpp = [float(v) for v in subprocess.run([exe, "0.02"], capture_output=True, text=True).stdout.split("final_com=")[1].split()[:2]]
    assert np.abs(np.array(fin_cpp) - np.array([0.64215196, -0.05244137])).max() < 1e-6


def test_work_queue_kernel_equals_static_pairing(mpc32, monkeypatch):
    """Large first calls run zmp_plan_kernel_dyn (a work queue per 32-lane group), small ones the static pairing: the
    arithmetic of a QP is the same, so an odd-sized batch on the queue kernel must reproduce, bit for bit, what chunks of
    4096 on the static kernel give (ZMP, jerk and pivot counts), and the queue must hand out every QP exactly once."""
    n = 20001  # 40002 QPs: not a multiple of anything (CCC_ZMP_QUEUE_MIN=0: the queue kernel whatever the size)
    b = fx.make_zmp_batch(n, 32, 0.0625, seed=77)
    monkeypatch.setenv("CCC_ZMP_QUEUE_MIN", "0")
    monkeypatch.setenv("CCC_ZMP_PREDICT", "0")  # (round 6: by default a batch this large is ordered by the predicted trips)
    monkeypatch.setenv("CCC_ZMP_HOST_CHUNK", "1000000")  # (one launch for the whole batch)
    mq = LinearMpcZmp(1.0, 2.0, 0.0625)
    monkeypatch.delenv("CCC_ZMP_QUEUE_MIN")
    monkeypatch.delenv("CCC_ZMP_PREDICT")
    monkeypatch.delenv("CCC_ZMP_HOST_CHUNK")
    full = mq.planOnceBatch(b["x0"], b["zlim"], 0.005, want_jerk=True)
    assert mq.last_kernel() == "zmp_plan_kernel_dyn<32,2>"
    assert np.all(full["status"] == 0)
    for a in range(0, n, 4096):
        part = mpc32.planOnceBatch(b["x0"][a:a + 4096], b["zlim"][a:a + 4096], 0.005, want_jerk=True)
        assert np.array_equal(part["zmp"], full["zmp"][a:a + 4096])
        assert np.array_equal(part["jerk"], full["jerk"][a:a + 4096])
        assert np.array_equal(part["pivots"], full["pivots"][a:a + 4096])


def test_predicted_schedule_gives_the_same_answers(monkeypatch):
    """Round 6: a call that has no usable history is ordered by a PREDICTION of every QP's pivot trips
    (csrc/zmp.hip zmp_predict_kernel: the rows the unconstrained optimum violates, weighted towards the start of the
    horizon) and runs on the static kernel, like predictions paired in a wavefront.  A schedule only: unrelated batches in
    turn, on a default handle and on one that keeps no history, give bit for bit what a handle without any ordering
    gives; and the prediction does predict -- the lock-step trips of the pairs it makes are well below those of the
    (x, y) pairs of an instance."""
    import torch

    n = 20000
    dev = torch.device("cuda:0")
    batches = [fx.make_zmp_batch(n, 32, 0.0625, seed=60 + k) for k in range(3)]

    def run(m, b):
        x0, zl = torch.from_numpy(b["x0"]).to(dev), torch.from_numpy(b["zlim"]).to(dev)
        z = torch.empty((n, 2), dtype=torch.float64, device=dev)
        j = torch.empty((n, 2, 32), dtype=torch.float64, device=dev)
        st = torch.empty((n, 2), dtype=torch.int32, device=dev)
        m.plan_batch_device(x0, zl, 0.005, z, j, st)
        torch.cuda.synchronize()
        return z.cpu().numpy(), j.cpu().numpy(), st.cpu().numpy()

    monkeypatch.setenv("CCC_ZMP_PREDICT", "0")
    monkeypatch.setenv("CCC_ZMP_HISTORY", "0")
    plain = LinearMpcZmp(1.0, 2.0, 0.0625)
    monkeypatch.delenv("CCC_ZMP_PREDICT")
    nohist = LinearMpcZmp(1.0, 2.0, 0.0625)
    monkeypatch.delenv("CCC_ZMP_HISTORY")
    default = LinearMpcZmp(1.0, 2.0, 0.0625)
    seen = set()
    for rep in range(3):
        for b in batches:
            ref = run(plain, b)
            assert plain.last_schedule() == "none"
            for m in (nohist, default):
                got = run(m, b)
                seen.add(m.last_schedule())
                for a, r in zip(got, ref):
                    assert np.array_equal(a, r)
            assert nohist.last_schedule() == "predicted pivot counts" and nohist.last_kernel() == "zmp_plan_kernel<32,2>"
    assert "predicted pivot counts" in seen
    # the prediction's quality, from the kernel's own counts: the same sort key, computed here
    b = batches[0]
    piv = (run(plain, b)[2] >> 8).reshape(-1).astype(np.int64)
    i = np.arange(32)
    A = np.stack([np.ones(32), (i + 1) * 0.0625, ((i + 1) * 0.0625) ** 2 / 2 - 1.0 / 9.80665], axis=1)
    fr = np.einsum("ik,nak->nai", A, b["x0"])
    viol = ((b["zlim"][:, :, 0, :] - fr > 0) | (b["zlim"][:, :, 1, :] - fr < 0)).reshape(-1, 32)
    key = np.rint(16.0 * (viol * np.exp(-i * 0.0625 / (2.35 * np.sqrt(1.0 / 9.80665)))).sum(1))
    paired = piv[np.argsort(-key, kind="stable")].reshape(-1, 2).max(1).sum() * 2.0 / piv.sum()
    as_is = piv.reshape(-1, 2).max(1).sum() * 2.0 / piv.sum()
    assert paired < 1.2 < 1.3 < as_is, (paired, as_is)


@pytest.mark.parametrize("N,n", [(24, 9001), (13, 16385), (32, 8192)])
def test_predicted_schedule_at_shorter_horizons_and_odd_sizes(N, n, monkeypatch):
    """The prediction pass and the ordered static kernel on horizons below 32 rows (idle lanes in every group), on an odd
    number of instances and exactly at the threshold (16384 QPs): planned ZMP, jerk and pivot counts bit for bit those of a
    handle without any ordering, and within the parity tolerance of the oracle."""
    dt = 2.0 / N
    b = fx.make_zmp_batch(n, N, dt, seed=90 + N)
    monkeypatch.setenv("CCC_ZMP_PREDICT", "0")
    monkeypatch.setenv("CCC_ZMP_HISTORY", "0")
    plain = LinearMpcZmp(1.0, 2.0, dt)
    monkeypatch.delenv("CCC_ZMP_PREDICT")
    monkeypatch.delenv("CCC_ZMP_HISTORY")
    monkeypatch.setenv("CCC_ZMP_HOST_CHUNK", "1000000")  # (one launch for the whole batch through the host entry)
    monkeypatch.setenv("CCC_ZMP_PREDICT_MIN", "1")
    pred = LinearMpcZmp(1.0, 2.0, dt)
    assert pred.horizon_steps_ == N
    import torch

    dev = torch.device("cuda:0")
    x0, zl = torch.from_numpy(b["x0"]).to(dev), torch.from_numpy(b["zlim"]).to(dev)

    def run(m):
        z = torch.empty((n, 2), dtype=torch.float64, device=dev)
        j = torch.empty((n, 2, N), dtype=torch.float64, device=dev)
        st = torch.empty((n, 2), dtype=torch.int32, device=dev)
        m.plan_batch_device(x0, zl, 0.005, z, j, st)
        torch.cuda.synchronize()
        return z.cpu().numpy(), j.cpu().numpy(), st.cpu().numpy()

    ref, got = run(plain), run(pred)
    assert plain.last_schedule() == "none" and pred.last_schedule() == "predicted pivot counts"
    for a, r in zip(got, ref):
        assert np.array_equal(a, r)
    assert np.all((got[2] & 0xff) == 0)
    o = _oracle().LinearMpcZmp(1.0, 2.0, dt).plan_batch(b["x0"][:2048], b["zlim"][:2048], 0.005, nthreads=8)
    assert np.abs(got[0][:2048] - o["zmp"]).max() <= ZMP_TOL


def test_history_schedule_gives_the_same_answers_whatever_the_caller_repeats(monkeypatch):
    """Round 5: a handle keeps the pivot counts of its last call and runs a call of the same size longest-first, the two QPs
    of a wavefront neighbours in that order (csrc/zmp.hip launch, zmp_order_kernel).  The order is a schedule, never an
    input of the arithmetic: repeating a batch, handing over ANOTHER batch of the same size (a history that predicts
    nothing), changing the size and coming back must all give, bit for bit, what a handle without a history gives -- from
    the static kernel and from the work-queue kernel alike (CCC_ZMP_QUEUE_MIN=0 puts the first, unordered call on it)."""
    import torch

    n = 6000
    bA = fx.make_zmp_batch(n, 32, 0.0625, seed=5)
    bB = fx.make_zmp_batch(n, 32, 0.0625, seed=6)
    dev = torch.device("cuda:0")

    def run(m, b, k=None):
        k = n if k is None else k
        x0 = torch.from_numpy(b["x0"][:k]).to(dev)
        zl = torch.from_numpy(b["zlim"][:k]).to(dev)
        z = torch.empty((k, 2), dtype=torch.float64, device=dev)
        j = torch.empty((k, 2, 32), dtype=torch.float64, device=dev)
        st = torch.empty((k, 2), dtype=torch.int32, device=dev)
        m.plan_batch_device(x0, zl, 0.005, z, j, st)
        torch.cuda.synchronize()
        return z.cpu().numpy(), j.cpu().numpy(), st.cpu().numpy(), m.last_kernel()

    monkeypatch.setenv("CCC_ZMP_HISTORY", "0")
    ref = LinearMpcZmp(1.0, 2.0, 0.0625)
    monkeypatch.delenv("CCC_ZMP_HISTORY")
    rA, rB, rC = run(ref, bA), run(ref, bB), run(ref, bB, n - 7)
    assert np.all((rA[2] & 0xff) == 0) and rA[2].max() >> 8 > 20  # (solved; pivot counts spread: there is something to order)
    for qmin in (None, "0"):
        if qmin is not None:
            monkeypatch.setenv("CCC_ZMP_QUEUE_MIN", qmin)
        m = LinearMpcZmp(1.0, 2.0, 0.0625)
        if qmin is not None:
            monkeypatch.delenv("CCC_ZMP_QUEUE_MIN")
        first = run(m, bA)
        assert first[3] == ("zmp_plan_kernel_dyn<32,2>" if qmin is not None else "zmp_plan_kernel<32,2>")
        calls = [(first, rA), (run(m, bA), rA), (run(m, bB), rB), (run(m, bB, n - 7), rC), (run(m, bA), rA), (run(m, bA), rA)]
        for got, want in calls:
            for a, b in zip(got[:3], want[:3]):
                assert np.array_equal(a, b)
        assert calls[1][0][3] == calls[2][0][3] == calls[5][0][3] == "zmp_plan_kernel<32,2>"  # (ordered calls: the static kernel)


def test_a_history_that_does_not_predict_is_dropped_and_taken_up_again(monkeypatch):
    """The order is only worth following when the last call's pivot counts say something about this call's: with unrelated
    batches of one size, call after call, two QPs picked by a wrong guess are worse company in a wavefront than the two axes
    of an instance.  The kernels keep |count - last count| per QP, the next call's sort adds them up, and the handle stops
    following (and goes on watching) when they are a fifth of the counts themselves -- read from page-locked memory a call
    or two late.  With CCC_ZMP_QUEUE_MIN=0 the kernel's name tells which way a call went; the answers never change."""
    import torch

    n = 5000
    dev = torch.device("cuda:0")
    bs = [fx.make_zmp_batch(n, 32, 0.0625, seed=40 + k) for k in range(4)]
    monkeypatch.setenv("CCC_ZMP_HISTORY", "0")
    ref = LinearMpcZmp(1.0, 2.0, 0.0625)
    monkeypatch.delenv("CCC_ZMP_HISTORY")
    monkeypatch.setenv("CCC_ZMP_QUEUE_MIN", "0")
    m = LinearMpcZmp(1.0, 2.0, 0.0625)
    monkeypatch.delenv("CCC_ZMP_QUEUE_MIN")

    def run(h, b):
        x0 = torch.from_numpy(b["x0"]).to(dev)
        zl = torch.from_numpy(b["zlim"]).to(dev)
        z = torch.empty((n, 2), dtype=torch.float64, device=dev)
        st = torch.empty((n, 2), dtype=torch.int32, device=dev)
        h.plan_batch_device(x0, zl, 0.005, z, None, st)
        torch.cuda.synchronize()
        return z.cpu().numpy(), st.cpu().numpy(), h.last_kernel()

    want = [run(ref, b)[:2] for b in bs]
    kernels = []
    for k in [0] * 5 + [1, 2, 3] * 5 + [0] * 10:
        z, st, name = run(m, bs[k])
        assert np.array_equal(z, want[k][0]) and np.array_equal(st, want[k][1])
        kernels.append(name)
    static, queue = "zmp_plan_kernel<32,2>", "zmp_plan_kernel_dyn<32,2>"
    assert kernels[0] == queue and kernels[1:5] == [static] * 4  # (first call: no counts yet; then the repeated batch)
    assert kernels[14:20] == [queue] * 6                         # (unrelated batches: dropped within a few calls)
    assert kernels[-3:] == [static] * 3                          # (the repeated batch again: taken up again)


def test_random_horizons_sweep():
    """Thirty random (horizon, com_height, seed) combinations across the packed-tableau range: ZMP and jerk parity
    with the oracle (guards the tile bookkeeping at row counts that are not multiples of the tile size)."""
    rng = np.random.default_rng(12345)
    for _ in range(30):
        N = int(rng.integers(33, 201))
        h = float(rng.uniform(0.6, 1.2))
        dt = 2.0 / N
        mpc = LinearMpcZmp(h, 2.0, dt)
        if mpc.horizon_steps_ != N:  # ceil(2 / (2 / N)) can round up
            N = mpc.horizon_steps_
        b = fx.make_zmp_batch(6, N, dt, com_height=h, seed=int(rng.integers(1, 10**6)))
        ref = _oracle().LinearMpcZmp(h, 2.0, dt).plan_batch(b["x0"], b["zlim"], 0.005, nthreads=4)
        r = mpc.planOnceBatch(b["x0"], b["zlim"], 0.005, want_jerk=True)
        assert np.all(r["status"] == 0), N
        assert np.abs(r["zmp"] - ref["zmp"]).max() <= ZMP_TOL, N
        assert _jerk_err(r["jerk"], ref["jerk"]) <= JERK_RTOL, N


@pytest.mark.parametrize("n", [1, 700, 20001])
def test_host_entry_in_place_on_pinned_buffers(mpc32, n):
    """ccc_zmp_plan_batch on page-locked caller buffers (N <= 32: the kernel works on them in place, no copy) gives,
    bit for bit, what the pageable route (pinned staging, chunked) gives: ZMP, jerk and status; and a call without
    jerk / status arrays leaves nothing behind."""
    import torch

    b = fx.make_zmp_batch(n, 32, 0.0625, seed=41)
    ref = mpc32.planOnceBatch(b["x0"], b["zlim"], 0.005, want_jerk=True)  # pageable numpy arrays
    x0 = torch.from_numpy(b["x0"]).pin_memory()
    zl = torch.from_numpy(b["zlim"]).pin_memory()
    zmp = torch.full((n, 2), np.nan, dtype=torch.float64).pin_memory()
    jerk = torch.full((n, 2, 32), np.nan, dtype=torch.float64).pin_memory()
    st = torch.full((n, 2), -1, dtype=torch.int32).pin_memory()
    mpc32.plan_batch_pinned(x0, zl, 0.005, zmp, status=st, jerk=jerk)
    assert np.array_equal(zmp.numpy(), ref["zmp"])
    assert np.array_equal(jerk.numpy(), ref["jerk"])
    assert np.array_equal(st.numpy() & 0xFF, ref["status"]) and np.array_equal(st.numpy() >> 8, ref["pivots"])
    zmp2 = torch.full((n, 2), np.nan, dtype=torch.float64).pin_memory()
    mpc32.plan_batch_pinned(x0, zl, 0.005, zmp2)
    assert np.array_equal(zmp2.numpy(), ref["zmp"])


def test_host_entry_long_horizon_pinned_buffers():
    """N > 32 keeps the copy route (those kernels stream their operands more than once); pinned and pageable agree."""
    import torch

    mpc = LinearMpcZmp(1.0, 100 * 0.05, 0.05)
    b = fx.make_zmp_batch(37, 100, 0.05, seed=5)
    ref = mpc.planOnceBatch(b["x0"], b["zlim"], 0.005)
    x0 = torch.from_numpy(b["x0"]).pin_memory()
    zl = torch.from_numpy(b["zlim"]).pin_memory()
    zmp = torch.empty((37, 2), dtype=torch.float64).pin_memory()
    mpc.plan_batch_pinned(x0, zl, 0.005, zmp)
    assert np.array_equal(zmp.numpy(), ref["zmp"])
