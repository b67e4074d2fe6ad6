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


# ===================================================================== DDP (ddp.c, ddp_models.c)
class _DdpConfig(ctypes.Structure):
    _fields_ = [("with_input_constraint", ctypes.c_int), ("max_iter", ctypes.c_int),
                ("initial_lambda", ctypes.c_double), ("initial_dlambda", ctypes.c_double),
                ("lambda_factor", ctypes.c_double), ("lambda_min", ctypes.c_double), ("lambda_max", ctypes.c_double),
                ("k_rel_norm_thre", ctypes.c_double), ("lambda_thre", ctypes.c_double),
                ("cost_update_ratio_thre", ctypes.c_double), ("cost_update_thre", ctypes.c_double),
                ("alpha_list", ctypes.c_double * 11), ("reg_type", ctypes.c_int), ("arith", ctypes.c_int),
                ("warm_start_guard", ctypes.c_int)]


class _DdpModel(ctypes.Structure):
    _fields_ = [("model", ctypes.c_int), ("N", ctypes.c_int), ("P", ctypes.c_int), ("M", ctypes.c_int),
                ("mass", ctypes.c_double), ("dt", ctypes.c_double),
                ("w_run", ctypes.c_double * 12), ("w_term", ctypes.c_double * 12), ("w_force", ctypes.c_double),
                ("force_lo", ctypes.c_double), ("force_hi", ctypes.c_double),
                ("phase_dim", ctypes.c_void_p), ("phase_vertex", ctypes.c_void_p), ("phase_ridge", ctypes.c_void_p),
                ("step_phase", ctypes.c_void_p), ("ref_pos", ctypes.c_void_p), ("ref_ori", ctypes.c_void_p),
                ("inertia", ctypes.c_void_p), ("inertia_per_phase", ctypes.c_int)]


class _DdpProblem(ctypes.Structure):
    _fields_ = [("S", ctypes.c_int), ("N", ctypes.c_int), ("M", ctypes.c_int), ("user", ctypes.c_void_p)] + \
               [(name, ctypes.c_void_p) for name in ("input_dim", "state_eq", "running_cost", "terminal_cost",
                                                      "state_eq_deriv", "running_cost_deriv", "terminal_cost_deriv",
                                                      "input_limits")]


_ddp_bound = False


class _DdpZmpParams(ctypes.Structure):
    _fields_ = [("N", ctypes.c_int), ("mass", ctypes.c_double), ("dt", ctypes.c_double),
                ("w_run_com_z", ctypes.c_double), ("w_run_zmp", ctypes.c_double), ("w_run_force_z", ctypes.c_double),
                ("w_term_com_xy", ctypes.c_double), ("w_term_com_z", ctypes.c_double),
                ("w_term_com_vel", ctypes.c_double)]


def _bind_ddp():
    global _ddp_bound
    L = lib()
    if _ddp_bound:
        return L
    L.oracle_ddp_default_config.argtypes = [ctypes.POINTER(_DdpConfig)]
    L.oracle_ddp_default_config.restype = None
    L.oracle_box_qp.argtypes = [ctypes.c_int, _dp, _dp, _dp, _dp, _dp, _ip, _dp, _dp, _ip, _ip]
    L.oracle_box_qp.restype = ctypes.c_int
    L.oracle_ddp_model_problem.argtypes = [ctypes.POINTER(_DdpModel), ctypes.POINTER(_DdpProblem)]
    L.oracle_ddp_model_problem.restype = None
    L.oracle_ddp_plan_batch.argtypes = [ctypes.POINTER(_DdpModel), ctypes.POINTER(_DdpConfig), ctypes.c_long,
                                        _ip, _dp, _dp, _ip, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip, _ip, _dp,
                                        ctypes.c_int]
    L.oracle_ddp_plan_batch.restype = ctypes.c_int
    L.oracle_ddp_model_eval.argtypes = [ctypes.POINTER(_DdpModel), ctypes.c_int] + [_dp] * 10
    L.oracle_ddp_model_eval.restype = None
    L.oracle_det_sincos.argtypes = [ctypes.c_double, _dp, _dp]
    L.oracle_det_sincos.restype = None
    L.oracle_ddpzmp_default_config.argtypes = [ctypes.POINTER(_DdpConfig)]
    L.oracle_ddpzmp_default_config.restype = None
    L.oracle_ddpzmp_plan_batch.argtypes = [ctypes.POINTER(_DdpZmpParams), ctypes.POINTER(_DdpConfig), ctypes.c_long,
                                           _dp, _dp, _dp, _dp, _dp, _ip, _ip, _dp, ctypes.c_int]
    L.oracle_ddpzmp_plan_batch.restype = ctypes.c_int
    L.oracle_ddpzmp_eval.argtypes = [ctypes.POINTER(_DdpZmpParams), _dp, ctypes.c_int] + [_dp] * 10
    L.oracle_ddpzmp_eval.restype = None
    _ddp_bound = True
    return L


def det_sincos(x):
    L = _bind_ddp()
    s, c = ctypes.c_double(0), ctypes.c_double(0)
    L.oracle_det_sincos(float(x), ctypes.cast(ctypes.byref(s), _dp), ctypes.cast(ctypes.byref(c), _dp))
    return s.value, c.value


def box_qp(H, g, lo, hi, x0=None):
    """Tassa's boxQP restated (oracle_box_qp). Returns (x, result, is_free, iters)."""
    L = _bind_ddp()
    H = np.ascontiguousarray(H, dtype=np.float64)
    n = H.shape[0]
    g = np.ascontiguousarray(g, dtype=np.float64)
    lo = np.ascontiguousarray(lo, dtype=np.float64)
    hi = np.ascontiguousarray(hi, dtype=np.float64)
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    is_free = np.zeros(n, dtype=np.int32)
    Lf = np.zeros((n, n))
    rd = np.zeros(n)
    nf = ctypes.c_int(0)
    it = ctypes.c_int(0)
    rc = L.oracle_box_qp(n, _ptr(H), _ptr(g), _ptr(lo), _ptr(hi), _ptr(x), _ptr(is_free, ctypes.c_int), _ptr(Lf),
                         _ptr(rd), ctypes.byref(nf), ctypes.byref(it))
    return x, rc, is_free.astype(bool), it.value


def ddp_exit_code(status):
    """ORACLE_DDP_STATUS_EXIT: 0 max_iter reached, 1 gradient small, 2 cost change small, -1 lambda > lambda_max."""
    return (np.asarray(status).astype(np.int64) & 0xff).astype(np.uint8).view(np.int8).astype(np.int32)


def ddp_warm_start_replaced(status):
    """ORACLE_DDP_STATUS_WARM_REPLACED: the warm-start guard replaced u_init by zero inputs."""
    s = np.asarray(status).astype(np.int64)
    return (s >= 0) & ((s & 0x100) != 0)


class Ddp:
    """CPU restatement of CCC::DdpCentroidal (model=0, S=9) / CCC::DdpSingleRigidBody (model=1, S=12) on
    pre-sampled, flattened per-instance problem data (oracle/ddp.c + oracle/ddp_models.c).

    weights: dict(run=[S], term=[S], force=float). Config overrides of the reference constructors
    (src/DdpCentroidal.cpp:197-201): initial_lambda=1e-6, lambda_min=1e-8, lambda_thre=1e-7."""

    def __init__(self, model, mass, horizon_dt, horizon_steps, weights, max_iter=500, P=4, M=16,
                 force_limits=(0.0, 1e6), arith=0, warm_start_guard=True):
        L = _bind_ddp()
        self.model, self.S = int(model), (9 if model == 0 else 12)
        self.N, self.P, self.M = int(horizon_steps), int(P), int(M)
        self.cfg = _DdpConfig()
        L.oracle_ddp_default_config(ctypes.byref(self.cfg))
        self.cfg.initial_lambda = 1e-6
        self.cfg.lambda_min = 1e-8
        self.cfg.lambda_thre = 1e-7
        self.cfg.max_iter = int(max_iter)
        # order of the long sums: 0 = left to right (ddp.c), 1 = the tile arithmetic (ddp_tile.c; M = 16, reg_type 1)
        self.cfg.arith = int(arith)
        # the default of both oracle_ddp_default_config and the product's ccc_ddp_default_config: a warm start that rolls
        # out worse than zero inputs is dropped (and reported in the status word); False = the recalled nmpc_ddp behaviour
        self.cfg.warm_start_guard = 1 if warm_start_guard else 0
        self.mdl = _DdpModel()
        self.mdl.model, self.mdl.N, self.mdl.P, self.mdl.M = self.model, self.N, self.P, self.M
        self.mdl.mass, self.mdl.dt = float(mass), float(horizon_dt)
        for a in range(self.S):
            self.mdl.w_run[a] = float(weights["run"][a])
            self.mdl.w_term[a] = float(weights["term"][a])
        self.mdl.w_force = float(weights["force"])
        self.mdl.force_lo, self.mdl.force_hi = float(force_limits[0]), float(force_limits[1])

    def plan_batch(self, prob, x0, u_init=None, nthreads=1):
        """prob: dict(phase_dim [n,P] i32, phase_vertex [n,P,M,3], phase_ridge [n,P,M,3], step_phase [n,N] i32,
        ref_pos [n,N+1,3], ref_ori [n,N+1,3] (SRB), inertia [n,3,3] or [n,P,3,3] (SRB: per instance / per contact phase));
        x0 [n,S]; u_init [n,N,M] | None.
        Returns dict(u [n,N,M], x [n,N+1,S], iters [n], status [n], cost [n])."""
        L = _bind_ddp()
        N, P, M, S = self.N, self.P, self.M, self.S
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        n = x0.shape[0]
        pd = np.ascontiguousarray(prob["phase_dim"], dtype=np.int32)
        pv = np.ascontiguousarray(prob["phase_vertex"], dtype=np.float64)
        pr = np.ascontiguousarray(prob["phase_ridge"], dtype=np.float64)
        sp = np.ascontiguousarray(prob["step_phase"], dtype=np.int32)
        rp = np.ascontiguousarray(prob["ref_pos"], dtype=np.float64)
        assert pd.shape == (n, P) and pv.shape == (n, P, M, 3) and pr.shape == (n, P, M, 3)
        assert sp.shape == (n, N) and rp.shape == (n, N + 1, 3) and x0.shape == (n, S)
        ro = ine = None
        if self.model == 1:
            ro = np.ascontiguousarray(prob["ref_ori"], dtype=np.float64)
            ine = np.ascontiguousarray(prob["inertia"], dtype=np.float64)
            # [n,3,3]: one matrix per instance; [n,P,3,3]: one per contact phase (oracle_ddp_model_t::inertia_per_phase)
            assert ro.shape == (n, N + 1, 3) and ine.shape in ((n, 3, 3), (n, P, 3, 3))
            self.mdl.inertia_per_phase = 1 if ine.ndim == 4 else 0
        ui = None if u_init is None else np.ascontiguousarray(u_init, dtype=np.float64)
        u = np.zeros((n, N, M))
        x = np.zeros((n, N + 1, S))
        iters = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        cost = np.zeros(n)
        L.oracle_ddp_plan_batch(ctypes.byref(self.mdl), ctypes.byref(self.cfg), n, _ptr(pd, ctypes.c_int), _ptr(pv),
                                _ptr(pr), _ptr(sp, ctypes.c_int), _ptr(rp), _ptr(ro), _ptr(ine), _ptr(x0), _ptr(ui),
                                _ptr(u), _ptr(x), _ptr(iters, ctypes.c_int), _ptr(status, ctypes.c_int), _ptr(cost),
                                int(nthreads))
        # status = the product's word (ccc_amd.h CCC_DDP_STATUS_*): the exit code, | 0x100 where the guard replaced u_init
        return dict(u=u, x=x, iters=iters, status=status, cost=cost, exit_code=ddp_exit_code(status),
                    warm_replaced=ddp_warm_start_replaced(status))

    def eval(self, prob, k, step, x, u):
        """Problem callbacks of instance k at (step, x, u): dict(x_next, Fx [S,S], Fu [S,M], run_cost, term_cost,
        Lx, Lu, Vx)."""
        L = _bind_ddp()
        S, M = self.S, self.M
        arrs = {}
        mdl = _DdpModel.from_buffer_copy(self.mdl)
        for name, dtype in (("phase_dim", np.int32), ("phase_vertex", np.float64), ("phase_ridge", np.float64),
                            ("step_phase", np.int32), ("ref_pos", np.float64), ("ref_ori", np.float64),
                            ("inertia", np.float64)):
            if name in prob and prob[name] is not None:
                arrs[name] = np.ascontiguousarray(prob[name][k], dtype=dtype)
                setattr(mdl, name, arrs[name].ctypes.data)
        if "inertia" in arrs:
            mdl.inertia_per_phase = 1 if arrs["inertia"].ndim == 3 else 0  # [P,3,3] of this instance: one per phase
        x = np.ascontiguousarray(x, dtype=np.float64)
        u = np.ascontiguousarray(np.pad(np.asarray(u, dtype=np.float64), (0, M - len(u))))
        xn, Fx, Fu = np.zeros(S), np.zeros((S, S)), np.zeros((S, M))
        Lx, Lu, Vx = np.zeros(S), np.zeros(M), np.zeros(S)
        rc, tc = ctypes.c_double(0), ctypes.c_double(0)
        L.oracle_ddp_model_eval(ctypes.byref(mdl), int(step), _ptr(x), _ptr(u), _ptr(xn), _ptr(Fx), _ptr(Fu),
                                ctypes.cast(ctypes.byref(rc), _dp), ctypes.cast(ctypes.byref(tc), _dp), _ptr(Lx),
                                _ptr(Lu), _ptr(Vx))
        return dict(x_next=xn, Fx=Fx, Fu=Fu, run_cost=rc.value, term_cost=tc.value, Lx=Lx, Lu=Lu, Vx=Vx)


# ===================================================================== LinearMpcXY (linear_mpc_xy.c)
class _XyParams(ctypes.Structure):
    _fields_ = [("mass", ctypes.c_double), ("horizon_dt", ctypes.c_double), ("horizon_steps", ctypes.c_int),
                ("M", ctypes.c_int), ("w_lmi", ctypes.c_double * 2), ("w_lm", ctypes.c_double * 2),
                ("w_am", ctypes.c_double * 2), ("w_force", ctypes.c_double)]


class IntrinsicallyStableMpc:
    """CPU restatement of CCC::IntrinsicallyStableMpc on pre-sampled reference sequences
    (oracle/intrinsically_stable_mpc.c)."""

    def __init__(self, com_height, horizon_duration, horizon_dt, w_zmp=1.0, w_zmp_vel=1e-3):
        L = lib()
        L.oracle_ism_create.restype = ctypes.c_void_p
        L.oracle_ism_create.argtypes = [ctypes.c_double] * 5
        L.oracle_ism_destroy.argtypes = [ctypes.c_void_p]
        L.oracle_ism_horizon_steps.argtypes = [ctypes.c_void_p]
        L.oracle_ism_plan_batch.restype = ctypes.c_int
        L.oracle_ism_plan_batch.argtypes = [ctypes.c_void_p, ctypes.c_long, _dp, _dp, ctypes.c_double, _dp, _dp, _ip, _ip,
                                            ctypes.c_int]
        self._h = L.oracle_ism_create(float(com_height), float(horizon_duration), float(horizon_dt), float(w_zmp),
                                      float(w_zmp_vel))
        self.horizon_steps = L.oracle_ism_horizon_steps(self._h)
        self.horizon_dt = float(horizon_dt)

    def __del__(self):
        try:
            if self._h:
                lib().oracle_ism_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def plan_batch(self, init, ref, control_dt=-1.0, want_vel=True, nthreads=1):
        """init [n,2,2] (capture_point, planned_zmp per axis), ref [n,2,3,N] (ref zmp, zmin, zmax rows per axis)
        -> dict(zmp [n,2], vel [n,2,N], status [n], iters [n,2])."""
        N = self.horizon_steps
        init = np.ascontiguousarray(init, dtype=np.float64)
        ref = np.ascontiguousarray(ref, dtype=np.float64)
        n = init.shape[0]
        assert init.shape == (n, 2, 2) and ref.shape == (n, 2, 3, N)
        zmp = np.empty((n, 2))
        vel = np.empty((n, 2, N)) if want_vel else None
        status = np.empty(n, dtype=np.int32)
        iters = np.empty((n, 2), dtype=np.int32)
        lib().oracle_ism_plan_batch(self._h, n, _ptr(init), _ptr(ref), float(control_dt), _ptr(zmp), _ptr(vel),
                                    _ptr(status, ctypes.c_int), _ptr(iters, ctypes.c_int), int(nthreads))
        return dict(zmp=zmp, vel=vel, status=status, iters=iters)


class LinearMpcZ:
    """CPU restatement of CCC::LinearMpcZ on pre-sampled contact flags / reference heights (oracle/linear_mpc_z.c)."""

    def __init__(self, mass, horizon_dt, horizon_steps, w_pos=1.0, w_force=1e-7):
        L = lib()
        L.oracle_z_create.restype = ctypes.c_void_p
        L.oracle_z_create.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_double]
        L.oracle_z_destroy.argtypes = [ctypes.c_void_p]
        L.oracle_z_plan_batch.restype = ctypes.c_int
        L.oracle_z_plan_batch.argtypes = [ctypes.c_void_p, ctypes.c_long, _ip, _dp, _dp, _dp, _dp, _ip, _ip, ctypes.c_int]
        self._h = L.oracle_z_create(float(mass), float(horizon_dt), int(horizon_steps), float(w_pos), float(w_force))
        self.N, self.mass = int(horizon_steps), float(mass)

    def __del__(self):
        try:
            if self._h:
                lib().oracle_z_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def plan_batch(self, contact, ref_pos, x0, nthreads=1, want_all=False):
        """contact [n,N] (0/1), ref_pos [n,N], x0 [n,2] (pos, vel) -> dict(force [n], force_all [n,N] (compact) | None,
        status [n], iters [n])."""
        N = self.N
        contact = np.ascontiguousarray(contact, dtype=np.int32)
        ref_pos = np.ascontiguousarray(ref_pos, dtype=np.float64)
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        n = x0.shape[0]
        assert contact.shape == (n, N) and ref_pos.shape == (n, N) and x0.shape == (n, 2)
        force = np.zeros(n)
        fall = np.zeros((n, N)) if want_all else None
        status = np.zeros(n, dtype=np.int32)
        iters = np.zeros(n, dtype=np.int32)
        lib().oracle_z_plan_batch(self._h, n, _ptr(contact, ctypes.c_int), _ptr(ref_pos), _ptr(x0), _ptr(force),
                                  _ptr(fall), _ptr(status, ctypes.c_int), _ptr(iters, ctypes.c_int), int(nthreads))
        return dict(force=force, force_all=fall, status=status, iters=iters)


class LinearMpcXY:
    """CPU restatement of CCC::LinearMpcXY on pre-sampled, flattened per-step data (oracle/linear_mpc_xy.c)."""

    def __init__(self, mass, horizon_dt, horizon_steps, w_lmi=(1.0, 1.0), w_lm=(0.0, 0.0), w_am=(1.0, 1.0),
                 w_force=1e-5, M=16):
        L = lib()
        L.oracle_xy_plan_batch.argtypes = [ctypes.POINTER(_XyParams), ctypes.c_long, _ip, _dp, _dp, _dp, _dp, _dp, _dp,
                                           _dp, _dp, _ip, _ip, ctypes.c_int]
        L.oracle_xy_plan_batch.restype = ctypes.c_int
        self.p = _XyParams()
        self.p.mass, self.p.horizon_dt, self.p.horizon_steps, self.p.M = float(mass), float(horizon_dt), int(horizon_steps), int(M)
        for a in range(2):
            self.p.w_lmi[a], self.p.w_lm[a], self.p.w_am[a] = float(w_lmi[a]), float(w_lm[a]), float(w_am[a])
        self.p.w_force = float(w_force)
        self.N, self.M, self.mass = int(horizon_steps), int(M), float(mass)

    def plan_batch(self, prob, x0, nthreads=1, want_all=False):
        """prob: dict(dim [n,N] i32, vertex [n,N,M,3], ridge [n,N,M,3], com_z [n,N], total_force_z [n,N],
        ref_out [n,N,6]); x0 [n,6].  Returns dict(u0 [n,M], lam [n,N*M] (compact) | None, iters, status)."""
        N, M = self.N, self.M
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        n = x0.shape[0]
        dim = np.ascontiguousarray(prob["dim"], dtype=np.int32)
        V = np.ascontiguousarray(prob["vertex"], dtype=np.float64)
        R = np.ascontiguousarray(prob["ridge"], dtype=np.float64)
        cz = np.ascontiguousarray(prob["com_z"], dtype=np.float64)
        fz = np.ascontiguousarray(prob["total_force_z"], dtype=np.float64)
        ref = np.ascontiguousarray(prob["ref_out"], dtype=np.float64)
        assert dim.shape == (n, N) and V.shape == (n, N, M, 3) and R.shape == (n, N, M, 3)
        assert cz.shape == (n, N) and fz.shape == (n, N) and ref.shape == (n, N, 6) and x0.shape == (n, 6)
        u0 = np.zeros((n, M))
        lam = np.zeros((n, N * M)) if want_all else None
        iters = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        lib().oracle_xy_plan_batch(ctypes.byref(self.p), n, _ptr(dim, ctypes.c_int), _ptr(V), _ptr(R), _ptr(cz),
                                   _ptr(fz), _ptr(ref), _ptr(x0), _ptr(u0), _ptr(lam), _ptr(iters, ctypes.c_int),
                                   _ptr(status, ctypes.c_int), int(nthreads))
        return dict(u0=u0, lam=lam, iters=iters, status=status)


class DdpZmp:
    """CPU restatement of CCC::DdpZmp (oracle/ddp_zmp.c + oracle/ddp.c) on pre-sampled RefData.

    weights = (running_com_pos_z, running_zmp, running_force_z, terminal_com_pos_xy, terminal_com_pos_z,
    terminal_com_vel), defaults of include/CCC/DdpZmp.h:72-77."""

    def __init__(self, mass, horizon_dt, horizon_steps, weights=(1e2, 1e-1, 1e-4, 1.0, 1e2, 1.0), max_iter=500):
        L = _bind_ddp()
        self.N = int(horizon_steps)
        self.cfg = _DdpConfig()
        L.oracle_ddpzmp_default_config(ctypes.byref(self.cfg))
        self.cfg.max_iter = int(max_iter)
        self.prm = _DdpZmpParams(self.N, float(mass), float(horizon_dt), *[float(w) for w in weights])

    def plan_batch(self, ref, x0, u_init=None, nthreads=1):
        """ref [n,N+1,4] (zmp x, y, z, com_z at t + i dt), x0 [n,6] ([cx,vx,cy,vy,cz,vz]), u_init [n,N,3] | None.
        Returns dict(u [n,N,3], x [n,N+1,6], iters, status, cost); planned zmp = u[:,0,:2], force_z = u[:,0,2]."""
        L = _bind_ddp()
        N = self.N
        ref = np.ascontiguousarray(ref, dtype=np.float64)
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        n = x0.shape[0]
        assert ref.shape == (n, N + 1, 4) and x0.shape == (n, 6)
        ui = None if u_init is None else np.ascontiguousarray(u_init, dtype=np.float64)
        assert ui is None or ui.shape == (n, N, 3)
        u = np.zeros((n, N, 3))
        x = np.zeros((n, N + 1, 6))
        iters = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        cost = np.zeros(n)
        L.oracle_ddpzmp_plan_batch(ctypes.byref(self.prm), ctypes.byref(self.cfg), n, _ptr(ref), _ptr(x0), _ptr(ui),
                                   _ptr(u), _ptr(x), _ptr(iters, ctypes.c_int), _ptr(status, ctypes.c_int), _ptr(cost),
                                   int(nthreads))
        return dict(u=u, x=x, iters=iters, status=status, cost=cost)

    def eval(self, ref, step, x, u):
        """Problem callbacks at (step, x, u) for one instance: ref [N+1,4]."""
        L = _bind_ddp()
        ref = np.ascontiguousarray(ref, dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        xn, Fx, Fu = np.zeros(6), np.zeros((6, 6)), np.zeros((6, 3))
        rc, tc = np.zeros(1), np.zeros(1)
        Lx, Lu, Vx = np.zeros(6), np.zeros(3), np.zeros(6)
        L.oracle_ddpzmp_eval(ctypes.byref(self.prm), _ptr(ref), int(step), _ptr(x), _ptr(u), _ptr(xn), _ptr(Fx), _ptr(Fu),
                             _ptr(rc), _ptr(tc), _ptr(Lx), _ptr(Lu), _ptr(Vx))
        return dict(x_next=xn, Fx=Fx, Fu=Fu, run_cost=rc[0], term_cost=tc[0], Lx=Lx, Lu=Lu, Vx=Vx)
