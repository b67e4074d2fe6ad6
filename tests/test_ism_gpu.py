"""GPU parity tests of the IntrinsicallyStableMpc HIP path (csrc/ism.hip) through the C-ABI.
Tolerance: planned ZMP within 1e-9 of the CPU oracle (the north star's fp64 bound for ZMP outputs); the QP is strictly
convex (H = w_vel I + w_zmp P'P), so the minimiser is unique and both exact solvers must agree to rounding."""
import numpy as np
import pytest

from centroidalcontrolcollection_amd import IntrinsicallyStableMpc
from centroidalcontrolcollection_amd import fixtures as fx

pytestmark = pytest.mark.gpu

TOL = 1e-9


def _oracle():
    from oracle import oracle

    return oracle


@pytest.mark.parametrize("T,dt,n", [(2.0, 0.02, 192), (1.0, 0.05, 128), (2.5, 0.02, 64), (3.0, 0.02, 48), (3.82, 0.02, 32)])
def test_parity_with_oracle(T, dt, n):
    """N = 100 is the reference test's horizon (TestIntrinsicallyStableMpc.cpp:17-18); N = 20 and N = 125 cover a
    short horizon and (almost) the largest one the tridiagonal kernel takes; N = 150 and N = 191 (the largest one built)
    run the packed LDS tableau alone (160 / 192 rows)."""
    o = _oracle().IntrinsicallyStableMpc(1.0, T, dt)
    N = o.horizon_steps
    b = fx.make_ism_batch(n, N, dt, seed=5)
    ro = o.plan_batch(b["init"], b["ref"], 0.005, nthreads=8)
    mpc = IntrinsicallyStableMpc(1.0, T, dt)
    assert mpc.horizon_steps_ == N
    r = mpc.planOnceBatch(b["init"], b["ref"], 0.005, want_vel=True)
    assert np.all(ro["status"] == 0) and np.all(r["status"] == 0)
    assert np.abs(r["zmp"] - ro["zmp"]).max() <= TOL
    assert np.abs(r["vel"] - ro["vel"]).max() <= 1e-7 * (1.0 + np.abs(ro["vel"]).max())
    assert ro["iters"].max() > 10 and r["pivots"].max() >= 3  # limits bind (GPU: tridiagonal solves, not pivots)
    # limits and the stability row hold at the returned sequence
    P = dt * np.tril(np.ones((N, N)))
    z = b["init"][:, :, 1, None] + np.einsum("ij,kaj->kai", P, r["vel"])
    assert (b["ref"][:, :, 1] - z).max() <= 1e-9 and (z - b["ref"][:, :, 2]).max() <= 1e-9


def test_horizon_limit_is_reported():
    from centroidalcontrolcollection_amd._lib import CccError

    assert IntrinsicallyStableMpc(1.0, 3.82, 0.02).horizon_steps_ == 191
    with pytest.raises(CccError):
        IntrinsicallyStableMpc(1.0, 3.84, 0.02)  # 192 steps


def test_non_default_weights_and_control_dt():
    o = _oracle().IntrinsicallyStableMpc(0.8, 1.6, 0.04, w_zmp=3.0, w_zmp_vel=5e-3)
    N = o.horizon_steps
    b = fx.make_ism_batch(96, N, 0.04, com_height=0.8, seed=9)
    mpc = IntrinsicallyStableMpc(0.8, 1.6, 0.04, IntrinsicallyStableMpc.WeightParam(3.0, 5e-3))
    for cdt in (-1.0, 0.002):
        ro = o.plan_batch(b["init"], b["ref"], cdt, nthreads=8, want_vel=False)
        r = mpc.planOnceBatch(b["init"], b["ref"], cdt)
        assert np.all(r["status"] == 0)
        assert np.abs(r["zmp"] - ro["zmp"]).max() <= TOL


def test_infeasible_capture_point_is_flagged():
    """A capture point the bounded ZMP cannot catch (eq. (14) against eq. (8)): status must not be SOLVED."""
    o = _oracle().IntrinsicallyStableMpc(1.0, 2.0, 0.02)
    b = fx.make_ism_batch(4, 100, 0.02, seed=1)
    b["init"][1, 0, 0] += 0.5
    ro = o.plan_batch(b["init"], b["ref"], 0.005)
    r = IntrinsicallyStableMpc(1.0, 2.0, 0.02).planOnceBatch(b["init"], b["ref"], 0.005)
    assert ro["status"][1] != 0 and r["status"][1, 0] != 0
    ok = [0, 2, 3]
    assert np.all(r["status"][ok] == 0) and np.abs(r["zmp"][ok] - ro["zmp"][ok]).max() <= TOL


def test_reference_closed_loop_through_planonce():
    """TestIntrinsicallyStableMpc.cpp:15-106 through planOnce(ref_data_func, initial_param, t, sim_dt) on the GPU."""
    mpc = IntrinsicallyStableMpc(1.0, 2.0, 0.02)

    def plan(ref_func, cp, planned, t, sim_dt):
        def rd(tt):
            z, lo, hi = ref_func(tt)
            return IntrinsicallyStableMpc.RefData(z, lo, hi)

        return mpc.planOnce(rd, IntrinsicallyStableMpc.InitialParam(cp, planned), t, sim_dt)

    log, fin = fx.run_closed_loop_ism(plan, end_time=10.0)
    for rec in log:
        assert np.all(rec["zmp"] - rec["zmin"] >= 0) and np.all(rec["zmax"] - rec["zmp"] >= 0)
    assert np.all(fin["com"] - fin["zmin"] >= 0) and np.all(fin["zmax"] - fin["com"] >= 0)


def test_device_entry_and_determinism():
    import torch

    mpc = IntrinsicallyStableMpc(1.0, 2.0, 0.02)
    b = fx.make_ism_batch(500, 100, 0.02, seed=2)
    dev = torch.device("cuda:0")
    init, ref = torch.from_numpy(b["init"]).to(dev), torch.from_numpy(b["ref"]).to(dev)
    z1 = torch.zeros((500, 2), dtype=torch.float64, device=dev)
    z2 = torch.zeros_like(z1)
    st = torch.zeros((500, 2), dtype=torch.int32, device=dev)
    mpc.plan_batch_device(init, ref, 0.005, z1, status=st)
    mpc.plan_batch_device(init, ref, 0.005, z2)
    torch.cuda.synchronize()
    assert torch.equal(z1, z2) and np.all((st.cpu().numpy() & 0xff) == 0)
    assert np.array_equal(mpc.planOnceBatch(b["init"], b["ref"], 0.005)["zmp"], z1.cpu().numpy())


def test_against_golden_vectors():
    """tests/golden/ism_golden.npz (make_golden_qp.py: the QP of src/IntrinsicallyStableMpc.cpp:8-104 built with numpy,
    solved in the ZMP-position variables by a primal active set, polished in long double with a KKT certificate):
    planned ZMP within 1e-12, the whole ZMP-velocity sequence within 1e-8 relative."""
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ism_golden.npz"))
    for tag, N, hd in (("n100", 100, 2.0), ("n20", 20, 0.4)):
        ok = g[tag + "_ok"].astype(bool)
        r = IntrinsicallyStableMpc(1.0, hd, hd / N).planOnceBatch(g[tag + "_init"], g[tag + "_ref"], 0.005, want_vel=True)
        assert np.abs(r["zmp"] - g[tag + "_zmp"])[ok].max() <= 1e-12
        assert np.abs(r["vel"] - g[tag + "_vel"])[ok].max() <= 1e-8 * max(1.0, np.abs(g[tag + "_vel"]).max())


def test_cpp_header_shim_matches_python_mirror():
    """Host C++ against include/CCC/IntrinsicallyStableMpc.h (examples/plan_once_intrinsically_stable_mpc.cpp): same
    kernel, same sampled inputs as the Python mirror -> identical planned ZMPs, for planOnce and planOnceBatch."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "plan_once_intrinsically_stable_mpc")
    if not os.path.exists(exe):
        import __graft_entry__

        __graft_entry__.build()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[-2] == "horizon_steps=100"
    # CCC::IntrinsicallyStableMpc1d (QpSolverType::QLD passed as reference code does): the x axis of the first plan
    assert lines[-1].startswith("1d zmp=") and lines[-1].endswith("status=0")
    assert float(lines[-1].split()[2]) == float(lines[0].split("zmp=")[1].split()[0])
    mpc = IntrinsicallyStableMpc(1.0, 2.0, 0.02)

    def ref(t):
        x = 0.0 if t < 1.0 else 0.2
        return IntrinsicallyStableMpc.RefData((x, 0.0), (x - 0.05, -0.125), (x + 0.05, 0.125))

    ip = IntrinsicallyStableMpc.InitialParam((0.03, -0.06), (0.01, 0.0))
    for k, t in enumerate((0.0, 0.5, 0.9)):
        z = mpc.planOnce(ref, ip, t, 0.005)
        cpp = np.array([float(v) for v in lines[k].split("zmp=")[1].split()])
        cppb = np.array([float(v) for v in lines[3 + k].split("zmp=")[1].split()])
        assert np.array_equal(cpp, z) and np.array_equal(cppb, z)


@pytest.mark.parametrize("env", [{"CCC_ISM_TABLEAU": "1"}, {"CCC_ISM_PCR_OUTER": "1"}, {"CCC_ISM_PCR_OUTER": "3"}])
def test_tableau_kernel_and_fallback_list(env):
    """The tridiagonal (PCR) kernel is the default; the packed-tableau kernel stays as its fallback.  In a subprocess with
    the development switches: the tableau kernel alone, and the PCR kernel starved of outer iterations so that most QPs
    go through the work list -- same answers as the oracle either way."""
    import os
    import subprocess
    import sys

    code = (
        "import numpy as np\n"
        "from centroidalcontrolcollection_amd import IntrinsicallyStableMpc, fixtures as fx\n"
        "from oracle import oracle\n"
        "b = fx.make_ism_batch(300, 100, 0.02, seed=11)\n"
        "o = oracle.IntrinsicallyStableMpc(1.0, 2.0, 0.02).plan_batch(b['init'], b['ref'], 0.005, nthreads=8)\n"
        "r = IntrinsicallyStableMpc(1.0, 2.0, 0.02).planOnceBatch(b['init'], b['ref'], 0.005, want_vel=True)\n"
        "assert np.all(r['status'] == 0) and np.all(o['status'] == 0)\n"
        "print(np.abs(r['zmp'] - o['zmp']).max(), np.abs(r['vel'] - o['vel']).max() / (1 + np.abs(o['vel']).max()))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root,
                         env=dict(os.environ, PYTHONPATH=root, **env))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    dz, dv = (float(v) for v in out.stdout.strip().splitlines()[-1].split())
    assert dz <= TOL and dv <= 1e-7
