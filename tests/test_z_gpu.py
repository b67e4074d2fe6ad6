"""GPU parity tests of the LinearMpcZ HIP path (csrc/z.hip) through the C-ABI.
Tolerance: H = w_pos B'B + 1e-7 I has condition number ~1e4 and forces are O(1e3) N, so two exact solvers agree to
~1e-9 relative; the planned forces are compared to 1e-8 relative to the largest force of the instance."""
import numpy as np
import pytest

from centroidalcontrolcollection_amd import LinearMpcZ
from centroidalcontrolcollection_amd import fixtures as fx

pytestmark = pytest.mark.gpu

RTOL = 1e-8


@pytest.fixture(autouse=True, params=["stream", "tableau"])
def _z_path(request, monkeypatch):
    """Every test of this module runs on both kernels: the streaming (projected Newton, Riccati sweeps) kernel that large
    batches take by default, and the LDS-tableau kernel that small batches (and the fallback list) take."""
    monkeypatch.setenv("CCC_Z_STREAM" if request.param == "stream" else "CCC_Z_TABLEAU", "1")
    yield


def _oracle():
    from oracle import oracle

    return oracle


def _expand(fall_compact, contact):
    out = np.zeros(contact.shape)
    for k in range(contact.shape[0]):
        if contact[k, 0]:
            steps = np.where(contact[k])[0]
            out[k, steps] = fall_compact[k, :len(steps)]
    return out


@pytest.mark.parametrize("N,dt,n", [(40, 0.05, 256), (64, 0.03, 96), (12, 0.1, 64), (100, 0.02, 96), (200, 0.01, 40)])
def test_parity_with_oracle(N, dt, n):
    """N = 40 is the reference test's horizon (TestLinearMpcZ.cpp:15-17); 64 is the largest the tableau kernel holds;
    beyond, the streaming projected-Newton kernel alone (the reference allocates for any horizon)."""
    mass = 100.0
    b = fx.make_z_batch(n, N, dt, seed=8)
    o = _oracle().LinearMpcZ(mass, dt, N).plan_batch(b["contact"], b["ref_pos"], b["x0"], nthreads=8, want_all=True)
    r = LinearMpcZ(mass, dt, N).planOnceBatch(b["contact"], b["ref_pos"], b["x0"], want_all=True)
    assert np.all(o["status"] == 0) and np.all(r["status"] == 0)
    fo = _expand(o["force_all"], b["contact"])
    scale = np.abs(fo).max(axis=1) + 1.0
    assert (np.abs(r["force"] - o["force"]) / scale).max() <= RTOL
    assert (np.abs(r["force_all"] - fo).max(axis=1) / scale).max() <= RTOL
    assert o["iters"].max() >= 3 and r["pivots"].max() >= 3  # bounds bind
    # no contact at the first step: planned force exactly 0 (src/LinearMpcZ.cpp:54-57), nothing planned
    nc = b["contact"][:, 0] == 0
    assert nc.any() and np.all(r["force"][nc] == 0.0) and np.all(r["force_all"][nc] == 0.0)
    assert np.all(r["force_all"][b["contact"] == 0] == 0.0)
    act = b["contact"] != 0
    act[nc] = False
    assert r["force_all"][act].min() >= 10.0 - 1e-9 and r["force_all"][act].max() <= 10.0 * mass * fx.G + 1e-6


def test_against_golden_vectors():
    """tests/golden/z_golden.npz (make_golden_qp.py: both phase models through scipy.linalg.expm, the condensed QP by
    simulation, primal active set + long-double polish with a KKT certificate): the planned force and the whole force
    sequence within 1e-9 relative (the oracle agrees with these vectors to 6e-13)."""
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "z_golden.npz"))
    for tag, N, dt in (("n40", 40, 0.05), ("n12", 12, 0.1)):
        r = LinearMpcZ(100.0, dt, N).planOnceBatch(g[tag + "_contact"], g[tag + "_ref_pos"], g[tag + "_x0"], want_all=True)
        assert np.all(r["status"] == 0)
        scale = np.maximum(1.0, np.abs(g[tag + "_force_all"]).max(axis=1))
        assert (np.abs(r["force"] - g[tag + "_force"]) / scale).max() <= 1e-9
        started = g[tag + "_contact"][:, 0] != 0  # no contact now -> the reference returns 0 without solving
        assert (np.abs(r["force_all"] - g[tag + "_force_all"])[started].max(axis=1) / scale[started]).max() <= 1e-9


def test_non_default_weights():
    mass, dt, N = 60.0, 0.04, 32
    b = fx.make_z_batch(96, N, dt, seed=10)
    o = _oracle().LinearMpcZ(mass, dt, N, w_pos=3.0, w_force=5e-7).plan_batch(b["contact"], b["ref_pos"], b["x0"], nthreads=8)
    r = LinearMpcZ(mass, dt, N, LinearMpcZ.WeightParam(3.0, 5e-7)).planOnceBatch(b["contact"], b["ref_pos"], b["x0"])
    assert np.all(r["status"] == 0)
    assert (np.abs(r["force"] - o["force"]) / (np.abs(o["force"]) + 1.0)).max() <= RTOL


def test_reference_closed_loop_through_planonce():
    """TestLinearMpcZ.cpp:15-78 through planOnce(contact_func, ref_pos_func, initial_param, t) on the GPU."""
    mpc = LinearMpcZ(100.0, 0.05, 40)
    log, (t, state) = fx.run_closed_loop_z(mpc.planOnce)
    for rec in log:
        assert abs(rec["state"][0] - rec["ref"]) < 2.0 and abs(rec["state"][1]) < 5.0
        if not rec["contact"]:
            assert abs(rec["force"]) < 1e-8
    assert abs(state[0] - fx.z_reference_height(t)) < 1e-2 and abs(state[1]) < 1e-2


def test_device_entry_and_determinism():
    import torch

    mass, dt, N, n = 100.0, 0.05, 40, 700
    mpc = LinearMpcZ(mass, dt, N)
    b = fx.make_z_batch(n, N, dt, seed=12)
    dev = torch.device("cuda:0")
    tc = torch.from_numpy(b["contact"]).to(dev)
    tr, tx = torch.from_numpy(b["ref_pos"]).to(dev), torch.from_numpy(b["x0"]).to(dev)
    f1 = torch.zeros(n, dtype=torch.float64, device=dev)
    f2 = torch.zeros_like(f1)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    mpc.plan_batch_device(tc, tr, tx, f1, status=st)
    mpc.plan_batch_device(tc, tr, tx, f2)
    torch.cuda.synchronize()
    assert torch.equal(f1, f2) and np.all((st.cpu().numpy() & 0xff) == 0)
    assert np.array_equal(mpc.planOnceBatch(b["contact"], b["ref_pos"], b["x0"])["force"], f1.cpu().numpy())


def test_cpp_header_shim_matches_python_mirror():
    """Host C++ against include/CCC/LinearMpcZ.h (examples/plan_once_linear_mpc_z.cpp): same kernel, same sampled
    inputs as the Python mirror -> identical planned forces (incl. the zero force in the flight window at t = 5.1)."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "plan_once_linear_mpc_z")
    if not os.path.exists(exe):
        import __graft_entry__

        __graft_entry__.build()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    mpc = LinearMpcZ(100.0, 0.05, 40)
    for k, t in enumerate((0.0, 4.4, 5.1, 7.9)):
        f = mpc.planOnce(fx.z_reference_contact, fx.z_reference_height, (1.05, -0.4), t)
        assert float(lines[k].split("force=")[1]) == f and float(lines[4 + k].split("force=")[1]) == f
    assert float(lines[2].split("force=")[1]) == 0.0


@pytest.mark.parametrize("env", [{"CCC_Z_TABLEAU": "1"}, {"CCC_Z_SWEEPS": "2"}, {"CCC_Z_SWEEPS": "4"}, {"CCC_Z_SWEEPS": "40"}, {}])
def test_tableau_kernel_and_fallback_list(env):
    """The streaming (Riccati / projected Newton) kernel is the default; the LDS-tableau kernel stays as its fallback.
    In a subprocess with the development switches: the tableau kernel alone, and the streaming kernel starved of
    sweeps so that (nearly) every instance goes through the fallback list, or given a budget nobody exceeds -- same answers as the oracle."""
    import os
    import subprocess
    import sys

    code = (
        "import numpy as np\n"
        "from centroidalcontrolcollection_amd import LinearMpcZ, fixtures as fx\n"
        "from oracle import oracle\n"
        "b = fx.make_z_batch(700, 40, 0.05, seed=21)\n"
        "o = oracle.LinearMpcZ(100.0, 0.05, 40).plan_batch(b['contact'], b['ref_pos'], b['x0'], nthreads=8)\n"
        "r = LinearMpcZ(100.0, 0.05, 40).planOnceBatch(b['contact'], b['ref_pos'], b['x0'])\n"
        "assert np.all(r['status'] == 0)\n"
        "print((np.abs(r['force'] - o['force']) / (np.abs(o['force']) + 1.0)).max())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root,
                         env=dict({k: v for k, v in os.environ.items() if not k.startswith("CCC_Z_")}, PYTHONPATH=root, **env))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert float(out.stdout.strip().splitlines()[-1]) <= RTOL
