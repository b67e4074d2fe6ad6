"""CPU check of the DDP kernel's phase logic: tests/emu/ddp_emu.cpp compiles the PRODUCT's wavefront code
(csrc/ddp_core.h) for the host (the 64 lanes of a phase run one after the other) and this test compares it with the
independent oracle (oracle/ddp.c).  Both are built by gcc without FMA contraction and sum in the same order, so the
comparison is exact; the GPU build differs only by FMA contraction and libm (tests/test_ddp_gpu.py).
The emulation is a test aid -- the product never runs it."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from centroidalcontrolcollection_amd import fixtures_ddp as fd
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Params(ctypes.Structure):
    _fields_ = [("model", ctypes.c_int), ("N", ctypes.c_int), ("P", ctypes.c_int), ("mass", ctypes.c_double),
                ("dt", ctypes.c_double), ("w_run", ctypes.c_double * 12), ("w_term", ctypes.c_double * 12),
                ("w_force", ctypes.c_double), ("flo", ctypes.c_double), ("fhi", ctypes.c_double),
                ("max_iter", ctypes.c_int), ("lambda0", ctypes.c_double), ("dlambda0", ctypes.c_double),
                ("lambda_factor", ctypes.c_double), ("lambda_min", ctypes.c_double), ("lambda_max", ctypes.c_double),
                ("k_rel_norm_thre", ctypes.c_double), ("lambda_thre", ctypes.c_double), ("ratio_thre", ctypes.c_double),
                ("cost_thre", ctypes.c_double), ("alpha", ctypes.c_double * 11), ("reg_type", ctypes.c_int),
                ("warm_guard", ctypes.c_int)]


def _build():
    so = os.path.join(ROOT, "tests", "emu", "libddp_emu.so")
    src = os.path.join(ROOT, "tests", "emu", "ddp_emu.cpp")
    hdr = os.path.join(ROOT, "centroidalcontrolcollection_amd", "csrc", "ddp_core.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, src])
    return ctypes.CDLL(so)


@pytest.fixture(scope="module")
def emu():
    """The phase versions of csrc/ddp_core.h (the row-per-lane solver: <= 16 ridges, <= 4 phases) compiled for the host."""
    L = _build()

    def run(model, N, dt, w, prob, x0, max_iter, u_init=None):
        M = prob["phase_vertex"].shape[2]
        assert M == 16 and prob["phase_dim"].shape[1] <= 4
        P = Params()
        P.model, P.N, P.P, P.mass, P.dt = model, N, prob["phase_dim"].shape[1], 100.0, dt
        S = 9 if model == 0 else 12
        for a in range(S):
            P.w_run[a], P.w_term[a] = w["run"][a], w["term"][a]
        P.w_force, P.flo, P.fhi, P.max_iter, P.reg_type = w["force"], 0.0, 1e6, max_iter, 1
        P.warm_guard = 1  # ccc_ddp_default_config
        P.lambda0, P.dlambda0, P.lambda_factor, P.lambda_min, P.lambda_max = 1e-6, 1.0, 1.6, 1e-8, 1e10
        P.k_rel_norm_thre, P.lambda_thre, P.ratio_thre, P.cost_thre = 1e-4, 1e-7, 0.0, 1e-7
        for i in range(11):
            P.alpha[i] = 10 ** (-3.0 * i / 10)
        n = x0.shape[0]
        u, x = np.zeros((n, N, M)), np.zeros((n, N + 1, S))
        it, st, c = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n)
        arr = {k: np.ascontiguousarray(v) for k, v in prob.items()}

        def p(a):
            return None if a is None else a.ctypes.data_as(ctypes.c_void_p)

        rc = L.ccc_ddp_emu_plan_batch(ctypes.byref(P), ctypes.c_long(n), M, p(arr["phase_dim"]),
                                      p(arr["phase_vertex"]), p(arr["phase_ridge"]), p(arr["step_phase"]),
                                      p(arr["ref_pos"]), p(arr.get("ref_ori")), p(arr.get("inertia")),
                                      p(np.ascontiguousarray(x0)), p(u_init), p(u), p(x), p(it), p(st), p(c))
        assert rc == 0
        return dict(u=u, x=x, iters=it, status=st, cost=c)

    return run


@pytest.mark.parametrize("max_iter", [1, 5, 20, 500])
def test_centroidal_kernel_logic_matches_oracle_exactly(emu, max_iter):
    N, dt = 100, 0.03
    prob, x0 = fd.make_centroidal_batch(6, N, dt, seed=5)
    e = emu(0, N, dt, fd.centroidal_weights(), prob, x0, max_iter)
    o = oracle.Ddp(0, 100.0, dt, N, fd.centroidal_weights(), max_iter=max_iter).plan_batch(prob, x0)
    assert np.array_equal(e["iters"], o["iters"]) and np.array_equal(e["status"], o["status"])
    assert np.array_equal(e["u"], o["u"]) and np.array_equal(e["x"], o["x"]) and np.array_equal(e["cost"], o["cost"])


@pytest.mark.parametrize("max_iter", [1, 20])
def test_srb_kernel_logic_matches_oracle_exactly(emu, max_iter):
    N, dt = 50, 0.03
    prob, x0 = fd.make_centroidal_batch(4, N, dt, seed=6, srb=True)
    e = emu(1, N, dt, fd.srb_weights(), prob, x0, max_iter)
    o = oracle.Ddp(1, 100.0, dt, N, fd.srb_weights(), max_iter=max_iter).plan_batch(prob, x0)
    assert np.array_equal(e["iters"], o["iters"]) and np.array_equal(e["status"], o["status"])
    assert np.array_equal(e["u"], o["u"]) and np.array_equal(e["cost"], o["cost"])


def test_warm_start_path_matches_oracle(emu):
    N, dt = 100, 0.03
    prob, x0 = fd.make_centroidal_batch(3, N, dt, seed=9)
    cold = oracle.Ddp(0, 100.0, dt, N, fd.centroidal_weights(), max_iter=4).plan_batch(prob, x0)
    x1 = x0 + 0.01
    e = emu(0, N, dt, fd.centroidal_weights(), prob, x1, 1, u_init=np.ascontiguousarray(cold["u"]))
    o = oracle.Ddp(0, 100.0, dt, N, fd.centroidal_weights(), max_iter=1).plan_batch(prob, x1, u_init=cold["u"])
    assert np.array_equal(e["u"], o["u"])


