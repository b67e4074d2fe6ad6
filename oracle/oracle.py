"""ctypes loader for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by the product package.  See oracle/ccc_oracle.h for the parity statement
("parity unpinned": the reference holds no golden vectors for this path and cannot be built here).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    """Compile oracle/*.c into oracle/liboracle.so with gcc (recipe: oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _ptr(a, ctype=ctypes.c_double):
    if a is None:
        return None
    return a.ctypes.data_as(ctypes.POINTER(ctype))


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = ctypes.CDLL(_LIB_PATH)
    L.oracle_expm.argtypes = [ctypes.c_int, _dp, _dp]
    L.oracle_expm.restype = None
    L.oracle_calc_disc_matrix.argtypes = [ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, ctypes.c_double, _dp, _dp, _dp]
    L.oracle_calc_disc_matrix.restype = None
    L.oracle_qp_solve.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int] + [_dp] * 8 + [_dp, _ip, _dp]
    L.oracle_qp_solve.restype = ctypes.c_int
    L.oracle_zmp_create.argtypes = [ctypes.c_double] * 3
    L.oracle_zmp_create.restype = ctypes.c_void_p
    L.oracle_zmp_destroy.argtypes = [ctypes.c_void_p]
    L.oracle_zmp_destroy.restype = None
    L.oracle_zmp_horizon_steps.argtypes = [ctypes.c_void_p]
    L.oracle_zmp_horizon_steps.restype = ctypes.c_int
    L.oracle_zmp_get_seq.argtypes = [ctypes.c_void_p, _dp, _dp]
    L.oracle_zmp_get_seq.restype = None
    L.oracle_zmp_proc_once.argtypes = [ctypes.c_void_p, _dp, _dp, _dp, ctypes.c_double, _dp, _dp, _ip]
    L.oracle_zmp_proc_once.restype = ctypes.c_int
    L.oracle_zmp_plan_batch.argtypes = [ctypes.c_void_p, ctypes.c_long, _dp, _dp, ctypes.c_double, _dp, _dp, _ip,
                                        _ip, ctypes.c_int]
    L.oracle_zmp_plan_batch.restype = ctypes.c_int
    _lib = L
    return L


def expm(M):
    M = np.ascontiguousarray(M, dtype=np.float64)
    out = np.empty_like(M)
    lib().oracle_expm(M.shape[0], _ptr(M), _ptr(out))
    return out


def calc_disc_matrix(A, B, dt, E=None):
    A = np.ascontiguousarray(A, dtype=np.float64)
    B = np.ascontiguousarray(B, dtype=np.float64).reshape(A.shape[0], -1)
    ns, ni = B.shape
    Ad = np.empty((ns, ns))
    Bd = np.empty((ns, ni))
    Ed = np.empty(ns)
    Ec = None if E is None else np.ascontiguousarray(E, dtype=np.float64)
    lib().oracle_calc_disc_matrix(ns, ni, _ptr(A), _ptr(B), _ptr(Ec), float(dt), _ptr(Ad), _ptr(Bd), _ptr(Ed))
    return Ad, Bd, Ed


def qp_solve(H, g, Aeq=None, beq=None, Cin=None, din=None, xl=None, xu=None):
    """min 1/2 x'Hx + g'x s.t. Aeq x = beq, Cin x <= din, xl <= x <= xu. Returns (x, status, iters, lam_in)."""
    H = np.ascontiguousarray(H, dtype=np.float64)
    n = H.shape[0]
    g = np.ascontiguousarray(g, dtype=np.float64)
    me = 0 if Aeq is None else np.atleast_2d(Aeq).shape[0]
    mi = 0 if Cin is None else np.atleast_2d(Cin).shape[0]
    Aeq = None if me == 0 else np.ascontiguousarray(np.atleast_2d(Aeq), dtype=np.float64)
    beq = None if me == 0 else np.ascontiguousarray(beq, dtype=np.float64)
    Cin = None if mi == 0 else np.ascontiguousarray(np.atleast_2d(Cin), dtype=np.float64)
    din = None if mi == 0 else np.ascontiguousarray(din, dtype=np.float64)
    xl = np.full(n, -np.inf) if xl is None else np.ascontiguousarray(xl, dtype=np.float64)
    xu = np.full(n, np.inf) if xu is None else np.ascontiguousarray(xu, dtype=np.float64)
    x = np.zeros(n)
    it = ctypes.c_int(0)
    lam = np.zeros(max(mi, 1))
    rc = lib().oracle_qp_solve(n, me, mi, _ptr(H), _ptr(g), _ptr(Aeq), _ptr(beq), _ptr(Cin), _ptr(din), _ptr(xl),
                               _ptr(xu), _ptr(x), ctypes.byref(it), _ptr(lam))
    return x, rc, it.value, lam[:mi]


class LinearMpcZmp:
    """CPU restatement of CCC::LinearMpcZmp on pre-sampled limit sequences (oracle/linear_mpc_zmp.c)."""

    def __init__(self, com_height, horizon_duration, horizon_dt):
        self._h = lib().oracle_zmp_create(float(com_height), float(horizon_duration), float(horizon_dt))
        self.horizon_steps = lib().oracle_zmp_horizon_steps(self._h)
        self.horizon_dt = float(horizon_dt)
        self.com_height = float(com_height)

    def __del__(self):
        try:
            if self._h:
                lib().oracle_zmp_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def seq(self):
        N = self.horizon_steps
        A = np.empty((N, 3))
        B = np.empty((N, N))
        lib().oracle_zmp_get_seq(self._h, _ptr(A), _ptr(B))
        return A, B

    def proc_once(self, zmin, zmax, x0, control_dt=-1.0):
        N = self.horizon_steps
        zmin = np.ascontiguousarray(zmin, dtype=np.float64)
        zmax = np.ascontiguousarray(zmax, dtype=np.float64)
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        zmp = ctypes.c_double(0)
        jerk = np.empty(N)
        it = ctypes.c_int(0)
        rc = lib().oracle_zmp_proc_once(self._h, _ptr(zmin), _ptr(zmax), _ptr(x0), float(control_dt),
                                        ctypes.byref(zmp), _ptr(jerk), ctypes.byref(it))
        return zmp.value, jerk, rc, it.value

    def plan_batch(self, x0, zlim, control_dt=-1.0, want_jerk=True, nthreads=1):
        """x0 [n,2,3], zlim [n,2,2,N] -> dict(zmp [n,2], jerk [n,2,N], status [n], iters [n,2])."""
        N = self.horizon_steps
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        zlim = np.ascontiguousarray(zlim, dtype=np.float64)
        n = x0.shape[0]
        assert x0.shape == (n, 2, 3) and zlim.shape == (n, 2, 2, N)
        zmp = np.empty((n, 2))
        jerk = np.empty((n, 2, N)) if want_jerk else None
        status = np.empty(n, dtype=np.int32)
        iters = np.empty((n, 2), dtype=np.int32)
        lib().oracle_zmp_plan_batch(self._h, n, _ptr(x0), _ptr(zlim), float(control_dt), _ptr(zmp), _ptr(jerk),
                                    _ptr(status, ctypes.c_int), _ptr(iters, ctypes.c_int), int(nthreads))
        return dict(zmp=zmp, jerk=jerk, status=status, iters=iters)
