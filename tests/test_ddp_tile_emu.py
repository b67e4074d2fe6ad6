"""CPU check of the DEFAULT DDP kernel (csrc/ddp_tile.h, written against csrc/w64.h): tests/emu/ddp_tile_emu.cpp compiles
the product's kernel source for the host, 64 lanes in lock step, and this test compares it BIT FOR BIT with the
independently written specification of its arithmetic, oracle/ddp_tile.c (oracle_ddp_config_t::arith = 1).  Also pinned
here: the tile arithmetic and the left-to-right arithmetic of oracle/ddp.c are the same algorithm up to rounding.
The emulation is a test aid -- the product never runs it."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from centroidalcontrolcollection_amd import fixtures_ddp as fd
from oracle import oracle


class Params(ctypes.Structure):
    """csrc/ddp_batch.h ddp_common::Params"""
    _fields_ = [("model", ctypes.c_int), ("N", ctypes.c_int), ("P", ctypes.c_int), ("mass", ctypes.c_double),
                ("dt", ctypes.c_double), ("w_run", ctypes.c_double * 12), ("w_term", ctypes.c_double * 12),
                ("w_force", ctypes.c_double), ("flo", ctypes.c_double), ("fhi", ctypes.c_double),
                ("max_iter", ctypes.c_int), ("lambda0", ctypes.c_double), ("dlambda0", ctypes.c_double),
                ("lambda_factor", ctypes.c_double), ("lambda_min", ctypes.c_double), ("lambda_max", ctypes.c_double),
                ("k_rel_norm_thre", ctypes.c_double), ("lambda_thre", ctypes.c_double), ("ratio_thre", ctypes.c_double),
                ("cost_thre", ctypes.c_double), ("alpha", ctypes.c_double * 11), ("reg_type", ctypes.c_int),
                ("warm_guard", ctypes.c_int), ("inertia_per_phase", ctypes.c_int), ("update_kmax", ctypes.c_int)]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu():
    so = os.path.join(ROOT, "tests", "emu", "libddp_tile_emu.so")
    src = os.path.join(ROOT, "tests", "emu", "ddp_tile_emu.cpp")
    hdrs = [os.path.join(ROOT, "centroidalcontrolcollection_amd", "csrc", h) for h in ("ddp_tile.h", "w64.h", "ddp_batch.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in [src] + hdrs):
        from centroidalcontrolcollection_amd.build import host_fma_flags

        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off"] + host_fma_flags()
                              + ["-o", so, src])
    L = ctypes.CDLL(so)
    L.ccc_ddp_tile_emu_lds_bytes.restype = ctypes.c_int

    def run(model, N, dt, w, prob, x0, max_iter, u_init=None, reg_type=1, slice=0, guard=1):
        P = Params()
        P.model, P.N, P.P, P.mass, P.dt = model, N, prob["phase_dim"].shape[1], 100.0, dt
        S = 9 if model == 0 else 12
        for a in range(S):
            P.w_run[a], P.w_term[a] = w["run"][a], w["term"][a]
        P.w_force, P.flo, P.fhi, P.max_iter, P.reg_type = w["force"], 0.0, 1e6, max_iter, reg_type
        P.warm_guard = guard  # 1 = ccc_ddp_default_config
        P.inertia_per_phase = 1 if (model == 1 and np.ndim(prob["inertia"]) == 4) else 0  # [n,P,3,3]: one matrix per phase
        P.update_kmax = 4  # S_UPDATE_KMAX of the specification
        P.lambda0, P.dlambda0, P.lambda_factor, P.lambda_min, P.lambda_max = 1e-6, 1.0, 1.6, 1e-8, 1e10
        P.k_rel_norm_thre, P.lambda_thre, P.ratio_thre, P.cost_thre = 1e-4, 1e-7, 0.0, 1e-7
        for i in range(11):
            P.alpha[i] = 10 ** (-3.0 * i / 10)
        n = x0.shape[0]
        c = lambda a, t=np.float64: np.ascontiguousarray(a, dtype=t)  # noqa: E731
        arr = [c(prob["phase_dim"], np.int32), c(prob["phase_vertex"]), c(prob["phase_ridge"]),
               c(prob["step_phase"], np.int32), c(prob["ref_pos"]), c(prob["ref_ori"]) if model == 1 else None,
               c(prob["inertia"]) if model == 1 else None, c(x0), None if u_init is None else c(u_init)]
        M = prob["phase_vertex"].shape[2]
        u, x = np.zeros((n, N, M)), np.zeros((n, N + 1, S))
        it, st, cost = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n)
        p = lambda a: None if a is None else ctypes.c_void_p(a.ctypes.data)  # noqa: E731
        rc = L.ccc_ddp_tile_emu_plan_batch(ctypes.byref(P), ctypes.c_int(M), ctypes.c_long(n), *[p(a) for a in arr], p(u), p(x), p(it),
                                           p(st), p(cost), ctypes.c_int(slice))
        assert rc == 0
        return dict(u=u, x=x, iters=it, status=st, cost=cost)

    run.lds_bytes = L.ccc_ddp_tile_emu_lds_bytes
    return run


def _ora(model, N, dt, w, max_iter, arith, P=4, M=16):
    return oracle.Ddp(model, 100.0, dt, N, w, max_iter=max_iter, arith=arith, P=P, M=M)


def _same(a, b):
    for k in ("u", "x", "cost", "iters", "status"):
        assert np.array_equal(a[k], b[k]), (k, np.abs(a[k].astype(float) - b[k].astype(float)).max())


def test_lds_footprint_allows_sixteen_wavefronts_per_cu(emu):
    """160 KB of LDS per CU / 16 wavefronts = 10240 B: what four wavefronts per SIMD would need at 16 ridges (the kernel asks
    for two -- csrc/ddp_tile.hip -- so twice that is there; the 12-state build went 300 B over it with the step table of
    round 5)."""
    assert emu.lds_bytes(9, 16) <= 10240 and emu.lds_bytes(12, 16) <= 10752
    # 32 ridges: two wavefronts per SIMD (eight per CU); 64 ridges: three wavefronts per CU
    assert emu.lds_bytes(9, 32) <= 20480 and emu.lds_bytes(12, 32) <= 20480
    assert emu.lds_bytes(9, 64) <= 54613 and emu.lds_bytes(12, 64) <= 54613


@pytest.mark.parametrize("model,N,max_iter", [(0, 100, 3), (0, 60, 500), (1, 50, 3), (1, 50, 500)])
def test_kernel_source_reproduces_the_tile_specification_bit_for_bit(emu, model, N, max_iter):
    w = fd.srb_weights() if model else fd.centroidal_weights()
    prob, x0 = fd.make_centroidal_batch(10, N, 0.03, seed=5 + N, srb=bool(model))
    _same(emu(model, N, 0.03, w, prob, x0, max_iter), _ora(model, N, 0.03, w, max_iter, 1).plan_batch(prob, x0, nthreads=8))


@pytest.mark.parametrize("model,N,max_iter,slice", [(0, 24, 7, 2), (1, 20, 9, 1), (1, 20, 9, 4)])
def test_a_solve_suspended_and_resumed_every_few_iterations_is_the_same_solve_bit_for_bit(emu, model, N, max_iter, slice):
    """csrc/ddp_tile.hip schedules batches larger than a resident set in slices of iterations (suspend() / resume());
    here: every `slice` iterations, with everything a wavefront owns overwritten in between."""
    prob, x0 = fd.make_centroidal_batch(12, N, 0.03, seed=21 + model, srb=model == 1)
    w = fd.srb_weights() if model else fd.centroidal_weights()
    whole = emu(model, N, 0.03, w, prob, x0, max_iter)
    _same(emu(model, N, 0.03, w, prob, x0, max_iter, slice=slice), whole)
    assert whole["iters"].max() > slice  # (the slices were needed)
    warm = emu(model, N, 0.03, w, prob, x0 + 0.01, 3, u_init=whole["u"])
    _same(emu(model, N, 0.03, w, prob, x0 + 0.01, 3, u_init=whole["u"], slice=1), warm)


@pytest.mark.parametrize("model", [0, 1])
def test_warm_start_partial_contacts_and_many_phases(emu, model):
    """Ridge counts other than 0 and 16 (a triangle contact: 12 ridges, a line contact: 8), seven contact phases, a warm
    start: the masks of the kernel (lanes beyond a step's dimension) against the zero-padded sums of the specification."""
    N, dt = 40, 0.03
    w = fd.srb_weights() if model else fd.centroidal_weights()
    prob, x0 = fd.make_centroidal_batch(6, N, dt, seed=3, srb=bool(model), P=7)
    rng = np.random.default_rng(1)
    prob["phase_dim"][:, 3], prob["phase_dim"][:, 4] = 12, 8
    for p, m in ((3, 12), (4, 8)):
        prob["phase_vertex"][:, p, :m] = prob["phase_vertex"][:, 0, :m]
        prob["phase_ridge"][:, p, :m] = prob["phase_ridge"][:, 0, :m]
        prob["phase_vertex"][:, p, m:] = 7.0  # garbage beyond the dimension must not matter
        prob["phase_ridge"][:, p, m:] = -3.0
    prob["step_phase"][:, 5:12] = 3
    prob["step_phase"][:, 30:36] = 4
    prob["step_phase"][:, 36:] = rng.integers(0, 7, size=(6, N - 36))
    o1 = _ora(model, N, dt, w, 6, 1, P=7)
    cold = o1.plan_batch(prob, x0)
    _same(emu(model, N, dt, w, prob, x0, 6), cold)
    _same(emu(model, N, dt, w, prob, x0 + 0.01, 2, u_init=cold["u"]),
          _ora(model, N, dt, w, 2, 1, P=7).plan_batch(prob, x0 + 0.01, u_init=cold["u"]))


@pytest.mark.parametrize("max_iter,slice", [(4, 0), (60, 0), (9, 2)])
def test_inertia_that_varies_over_the_horizon_bit_for_bit(emu, max_iter, slice):
    """ccc_ddp_params_t::inertia_per_phase (ABI 5, VERDICT r5 missing #1): the kernel caches the phase's inertia matrix, its
    Cholesky factor and the factor's reciprocals with the phase's contact vectors; the specification indexes the step's
    matrix.  Kernel source against specification bit for bit (whole and suspended / resumed), the dense left-to-right
    oracle to rounding, and the layout with the SAME matrix in every phase against the one-matrix-per-instance layout."""
    N, dt = 40, 0.03
    w = fd.srb_weights()
    prob, prob4, x0 = fd.make_varying_inertia_batch(6, N, dt, seed=31)
    o1 = _ora(1, N, dt, w, max_iter, 1, P=7)
    r4 = o1.plan_batch(prob4, x0)
    _same(emu(1, N, dt, w, prob4, x0, max_iter, slice=slice), r4)
    # it matters: the plan differs from the plan with phase 0's matrix throughout
    const = dict(prob, inertia=np.ascontiguousarray(prob4["inertia"][:, 0]))
    rc = o1.plan_batch(const, x0)
    assert not np.array_equal(rc["u"], r4["u"])
    # one matrix repeated in every phase IS the per-instance layout, bit for bit (kernel source and specification)
    rep = dict(prob, inertia=np.ascontiguousarray(np.repeat(prob4["inertia"][:, :1], 7, axis=1)))
    _same(o1.plan_batch(rep, x0), rc)
    _same(emu(1, N, dt, w, rep, x0, max_iter), emu(1, N, dt, w, const, x0, max_iter))
    # the independent dense restatement (oracle/ddp.c + ddp_models.c, arith 0) reads the step's matrix too
    if max_iter >= 60:
        r0 = _ora(1, N, dt, w, max_iter, 0, P=7).plan_batch(prob4, x0)
        same_path = (r0["iters"] == r4["iters"]) & (r0["status"] == r4["status"])
        assert same_path.mean() >= 0.8
        assert (np.abs(r0["cost"] - r4["cost"]) / np.abs(r0["cost"]))[same_path].max() <= 1e-9


@pytest.mark.parametrize("model,M", [(0, 16), (1, 16), (0, 32)])
def test_reg_type_2_bit_for_bit(emu, model, M):
    """ccc_ddp_config_t::reg_type = 2 (lambda on Vxx: V6r = V6 + lambda I, Wr = W + lambda Fx[rows6]) in the tile
    arithmetic (round 4: every ridge stride): kernel source against specification, and against the dense left-to-right
    oracle to rounding."""
    N, dt = 40, 0.03 if M == 16 else 0.05
    w = fd.srb_weights() if model else fd.centroidal_weights()
    if M == 16:
        prob, x0 = fd.make_centroidal_batch(6, N, dt, seed=4, srb=bool(model))
    else:
        prob, x0 = fd.make_walking_batch(4, N, dt, seed=4, srb=bool(model))
    P = prob["phase_dim"].shape[1]
    o1 = _ora(model, N, dt, w, 8, 1, P=P, M=M)
    o1.cfg.reg_type = 2
    r1 = o1.plan_batch(prob, x0)
    _same(emu(model, N, dt, w, prob, x0, 8, reg_type=2), r1)
    o0 = _ora(model, N, dt, w, 8, 0, P=P, M=M)
    o0.cfg.reg_type = 2
    r0 = o0.plan_batch(prob, x0)
    assert np.abs(r0["cost"] - r1["cost"]).max() <= 1e-8 * np.abs(r0["cost"]).max()
    # (and it is a different regularisation from reg_type 1: the iterates differ)
    assert not np.array_equal(r1["u"], _ora(model, N, dt, w, 8, 1, P=P, M=M).plan_batch(prob, x0)["u"])


@pytest.mark.parametrize("model", [0, 1])
def test_warm_start_guard_bit_for_bit(emu, model):
    """ccc_ddp_config_t::warm_start_guard: warm starts that roll out worse than zero inputs (here: the converged plan of
    ANOTHER initial state scaled by 3, and a plan with a NaN in it) are dropped, good ones are kept -- kernel source and
    specification take the same decision and then the same bits."""
    N, dt = 40, 0.03
    w = fd.srb_weights() if model else fd.centroidal_weights()
    prob, x0 = fd.make_centroidal_batch(6, N, dt, seed=9, srb=bool(model))
    o = _ora(model, N, dt, w, 2, 1)
    good = _ora(model, N, dt, w, 30, 1).plan_batch(prob, x0)["u"]
    bad = 3.0 * good
    bad[1, 5, 3] = np.nan
    bad[2] = good[2]
    r_bad = o.plan_batch(prob, x0, u_init=bad)
    _same(emu(model, N, dt, w, prob, x0, 2, u_init=bad), r_bad)
    cold = o.plan_batch(prob, x0)
    warm = o.plan_batch(prob, x0, u_init=good)
    for k in (0, 1, 3, 4, 5):  # dropped: the solve is the cold solve
        assert np.array_equal(r_bad["u"][k], cold["u"][k])
    assert np.array_equal(r_bad["u"][2], warm["u"][2]) and not np.array_equal(warm["u"][2], cold["u"][2])
    assert np.all(np.isfinite(r_bad["u"]))
    # the replacement is reported: the status word carries CCC_DDP_STATUS_WARM_REPLACED_BIT on exactly
    # the dropped instances -- through a suspended / resumed solve too -- and is the plain exit code everywhere else
    fired = np.array([True, True, False, True, True, True])
    for r in (r_bad, emu(model, N, dt, w, prob, x0, 2, u_init=bad, slice=1)):
        assert np.array_equal(oracle.ddp_warm_start_replaced(r["status"]), fired)
        assert np.array_equal(r["status"][fired], 0x100 | (oracle.ddp_exit_code(r["status"])[fired] & 0xff))
        assert np.array_equal(oracle.ddp_exit_code(r["status"]), cold["status"] * fired + warm["status"] * ~fired)
    assert not oracle.ddp_warm_start_replaced(cold["status"]).any() and not oracle.ddp_warm_start_replaced(warm["status"]).any()
    # guard off = the recalled nmpc_ddp behaviour (src/DdpSingleRigidBody.cpp:299-303: u_list goes to solve() as is): same
    # bits from kernel source and specification on the very instances where the guard WOULD fire, and no flag
    off = oracle.Ddp(model, 100.0, dt, N, w, max_iter=2, arith=1, warm_start_guard=False)
    bad_finite = np.where(np.isfinite(bad), bad, 0.0)
    r_off = off.plan_batch(prob, x0, u_init=bad_finite)
    _same(emu(model, N, dt, w, prob, x0, 2, u_init=bad_finite, guard=0), r_off)
    assert not oracle.ddp_warm_start_replaced(r_off["status"]).any()
    assert not any(np.array_equal(r_off["u"][k], cold["u"][k]) for k in (0, 1, 3, 4, 5))


@pytest.mark.parametrize("model,N", [(0, 100), (1, 50)])
def test_the_two_arithmetics_are_the_same_algorithm_up_to_rounding(model, N):
    """oracle/ddp.c (dense matrices, left-to-right sums, Cholesky) and oracle/ddp_tile.c (trees, fma chains, the
    structured backward step: Woodbury form of the box-QP, 6 x 6 Gauss-Jordan, the D'E form of the value update) on the
    same problems, run to convergence: same iteration counts and exit codes on >= 90 % of the instances, costs to 1e-12
    relative on most of them and to 1e-6 on all (both stop within nmpc_ddp's cost_update_thre = 1e-7 of the same
    minimiser; an iterate that differs in the 10th digit may stop one line-search candidate apart), force scales to
    5e-2 (flat directions: held by the 1e-6 force weight only) -- the re-specification changed roundings, not the
    algorithm."""
    w = fd.srb_weights() if model else fd.centroidal_weights()
    prob, x0 = fd.make_centroidal_batch(48, N, 0.03, seed=11, srb=bool(model))
    r0 = _ora(model, N, 0.03, w, 500, 0).plan_batch(prob, x0, nthreads=8)
    r1 = _ora(model, N, 0.03, w, 500, 1).plan_batch(prob, x0, nthreads=8)
    same_path = (r0["iters"] == r1["iters"]) & (r0["status"] == r1["status"])
    assert same_path.mean() >= 0.9, same_path.mean()  # (a discrete decision may flip on a rounding: rare)
    rel = np.abs(r0["cost"] - r1["cost"]) / np.abs(r0["cost"])
    assert np.mean(rel[same_path] <= 1e-12) >= 0.85 and rel[same_path].max() <= 1e-6, np.sort(rel[same_path])[-5:]
    assert np.abs(r0["u"] - r1["u"])[same_path].max() <= 5e-2
    # (round 5, the single-rigid-body solves multiply by reciprocals: one of the 48 cold solves of the 12-state model now
    #  wanders off to a tumbling rollout and spends its 500 iterations there -- the chaotic cold solve of DESIGN.md 7.1, on
    #  another instance than before; the dense arithmetic has its own such instances on other seeds)
    #  -- pinned: exactly instance 4 of the 12-state batch, which spends its 500 iterations at a cost of ~19.5
    #  against a median of ~2; its plan stays finite and inside the limits.  Every centroidal instance converges.
    bad = np.nonzero(r1["status"] < 1)[0]
    assert list(bad) == ([4] if model else []), bad
    if model:
        assert r1["iters"][4] == 500 and np.isfinite(r1["cost"][4]) and r1["cost"][4] < 25.0
        assert np.isfinite(r1["u"][4]).all() and r1["u"][4].min() >= 0.0 and r1["u"][4].max() <= 1e6


@pytest.mark.parametrize("model,max_iter", [(0, 4), (1, 4), (0, 200)])
def test_double_support_walking_32_ridges_bit_for_bit(emu, model, max_iter):
    """M = 32 (two surface contacts per step): the kernel's two blocks of 16 ridges against the specification's treeM /
    rows4 / block-wise substitutions, on walking sequences with 0-, 16- and 32-ridge steps and 6-10 contact phases."""
    N, dt = 40, 0.05
    w = fd.srb_weights() if model else fd.centroidal_weights()
    prob, x0 = fd.make_walking_batch(5, N, dt, seed=77, srb=bool(model))
    P = prob["phase_dim"].shape[1]
    assert prob["phase_dim"].max() == 32
    _same(emu(model, N, dt, w, prob, x0, max_iter), _ora(model, N, dt, w, max_iter, 1, P=P, M=32).plan_batch(prob, x0, nthreads=8))


@pytest.mark.parametrize("model,max_iter", [(0, 3), (1, 3), (0, 100)])
def test_multi_contact_64_ridges_bit_for_bit(emu, model, max_iter):
    """M = 64: feet + hands (src/DdpCentroidal.cpp:49-60 takes any contact_list) -- steps with 16, 32, 48 and 64 ridges."""
    N, dt = 30, 0.05
    w = fd.srb_weights() if model else fd.centroidal_weights()
    prob, x0 = fd.make_multicontact_batch(4, N, dt, seed=13, srb=bool(model))
    assert set(np.unique(prob["phase_dim"])) >= {32, 48, 64}
    r = emu(model, N, dt, w, prob, x0, max_iter)
    _same(r, _ora(model, N, dt, w, max_iter, 1, P=6, M=64).plan_batch(prob, x0, nthreads=8))
    assert np.all(r["status"] >= 0)


@pytest.mark.parametrize("model", [0, 1])
def test_ridge_stride_does_not_change_the_answer(model):
    """A 16-ridge problem embedded in 32- and 64-ridge tables (what the shims do when another step of the horizon has more
    contacts): the extra blocks are exact zeros in every sum, so the specification returns the same bits."""
    N, dt = 40, 0.03
    w = fd.srb_weights() if model else fd.centroidal_weights()
    prob, x0 = fd.make_centroidal_batch(6, N, dt, seed=21, srb=bool(model))
    r16 = _ora(model, N, dt, w, 50, 1).plan_batch(prob, x0)
    for M in (32, 64):
        wide = dict(prob)
        for key in ("phase_vertex", "phase_ridge"):
            a = np.zeros(prob[key].shape[:2] + (M, 3))
            a[:, :, :16] = prob[key]
            wide[key] = a
        rM = _ora(model, N, dt, w, 50, 1, M=M).plan_batch(wide, x0)
        assert np.array_equal(rM["u"][:, :, :16], r16["u"]) and np.all(rM["u"][:, :, 16:] == 0.0)
        assert np.array_equal(rM["cost"], r16["cost"]) and np.array_equal(rM["iters"], r16["iters"])


def test_tile_arithmetic_matches_left_to_right_on_wide_problems():
    """arith 0 against arith 1 at 32 and 64 ridges, run to convergence: the same algorithm up to rounding."""
    w = fd.centroidal_weights()
    for M, (prob, x0), N, dt in ((32, fd.make_walking_batch(12, 40, 0.05, seed=5), 40, 0.05),
                                 (64, fd.make_multicontact_batch(8, 30, 0.05, seed=5), 30, 0.05)):
        P = prob["phase_dim"].shape[1]
        r0 = _ora(0, N, dt, w, 500, 0, P=P, M=M).plan_batch(prob, x0, nthreads=8)
        r1 = _ora(0, N, dt, w, 500, 1, P=P, M=M).plan_batch(prob, x0, nthreads=8)
        ok = (r0["status"] >= 1) & (r1["status"] >= 1)
        assert ok.mean() >= 0.9
        rel = np.abs(r0["cost"] - r1["cost"]) / np.abs(r0["cost"])
        assert rel[ok].max() <= 1e-8, rel
