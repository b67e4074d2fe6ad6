"""Host-side mirrors of CCC::DdpCentroidal and CCC::DdpSingleRigidBody over the C-ABI (include/ccc_amd.h, csrc/ddp.hip).

Same names and argument meaning as the reference (/root/reference/include/CCC/DdpCentroidal.h:13-366,
include/CCC/DdpSingleRigidBody.h): ``DdpCentroidal(mass, horizon_dt, horizon_steps, weight_param)``,
``planOnce(motion_param_func, ref_data_func, initial_param, current_time)`` returning the force scales of the first
horizon step, ``ddp_solver_.config().max_iter``, ``ddp_solver_.controlData().u_list`` -- plus the batched entry points.
A contact (ForceColl::Contact, external to the reference) is represented by its flattened ridge list:
``(vertex [m,3], ridge [m,3])`` in contact -> vertex -> ridge order (src/DdpCentroidal.cpp:49-60).
"""
import ctypes

import numpy as np

from . import _lib

MAX_RIDGES = 16  # default ridge stride (one surface contact per step)
MAX_RIDGES_WIDE = 32  # max_ridges=32: two surface contacts per step (double support)
MAX_RIDGES_MULTI = 64  # max_ridges=64: up to four surface contacts per step (feet + hands)
STATUS_WARM_REPLACED_BIT = 0x100  # CCC_DDP_STATUS_WARM_REPLACED_BIT (include/ccc_amd.h)


def exit_code(status):
    """CCC_DDP_STATUS_EXIT: 0 max_iter reached, 1 gradient small, 2 cost change small, -1 lambda > lambda_max (array or
    scalar status words of ccc_ddp_plan_batch*; torch tensors: pass ``.cpu().numpy()``)."""
    return (np.asarray(status).astype(np.int64) & 0xff).astype(np.uint8).view(np.int8).astype(np.int32)


def warm_start_replaced(status):
    """CCC_DDP_STATUS_WARM_REPLACED: the warm-start guard (Config.warm_start_guard, on by default, not a nmpc_ddp option)
    replaced this instance's u_init by zero inputs -- on exactly these instances the result is not what the reference
    computes from the same u_list (src/DdpSingleRigidBody.cpp:299-303)."""
    s = np.asarray(status).astype(np.int64)
    return (s >= 0) & ((s & STATUS_WARM_REPLACED_BIT) != 0)


class _Params(ctypes.Structure):
    _fields_ = [("model", ctypes.c_int), ("mass", ctypes.c_double), ("horizon_dt", ctypes.c_double),
                ("horizon_steps", ctypes.c_int), ("w_run", ctypes.c_double * 12), ("w_term", ctypes.c_double * 12),
                ("w_force", ctypes.c_double), ("force_scale_limits", ctypes.c_double * 2), ("max_phases", ctypes.c_int), ("max_ridges", ctypes.c_int),
                ("inertia_per_phase", ctypes.c_int)]


class Config(ctypes.Structure):
    """ddp_solver_->config(): nmpc_ddp::DDPSolver::Configuration fields the path uses."""
    _fields_ = [("max_iter", ctypes.c_int), ("initial_lambda", ctypes.c_double), ("initial_dlambda", ctypes.c_double),
                ("lambda_factor", ctypes.c_double), ("lambda_min", ctypes.c_double), ("lambda_max", ctypes.c_double),
                ("k_rel_norm_thre", ctypes.c_double), ("lambda_thre", ctypes.c_double),
                ("cost_update_ratio_thre", ctypes.c_double), ("cost_update_thre", ctypes.c_double),
                ("alpha_list", ctypes.c_double * 11), ("reg_type", ctypes.c_int), ("precision", ctypes.c_int),
                ("warm_start_guard", ctypes.c_int)]


def _bind(L):
    if getattr(L, "_ddp_bound", False):
        return
    vp = ctypes.c_void_p
    L.ccc_ddp_default_config.restype = None
    L.ccc_ddp_default_config.argtypes = [ctypes.POINTER(Config)]
    L.ccc_ddp_create.restype = ctypes.c_int
    L.ccc_ddp_create.argtypes = [ctypes.POINTER(_Params), ctypes.c_int, ctypes.POINTER(vp)]
    L.ccc_ddp_destroy.restype = None
    L.ccc_ddp_destroy.argtypes = [vp]
    L.ccc_ddp_set_config.restype = ctypes.c_int
    L.ccc_ddp_set_config.argtypes = [vp, ctypes.POINTER(Config)]
    L.ccc_ddp_set_limits.restype = ctypes.c_int
    L.ccc_ddp_set_limits.argtypes = [vp, ctypes.c_double, ctypes.c_double]
    L.ccc_ddp_set_inertia_per_phase.restype = ctypes.c_int
    L.ccc_ddp_set_inertia_per_phase.argtypes = [vp, ctypes.c_int]
    L.ccc_ddp_state_dim.restype = ctypes.c_int
    L.ccc_ddp_state_dim.argtypes = [vp]
    L.ccc_ddp_arithmetic.restype = ctypes.c_int
    L.ccc_ddp_arithmetic.argtypes = [vp]
    L.ccc_ddp_effective_precision.restype = ctypes.c_int
    L.ccc_ddp_effective_precision.argtypes = [vp]
    L.ccc_ddp_plan_batch_device.restype = ctypes.c_int
    L.ccc_ddp_plan_batch_device.argtypes = [vp, ctypes.c_int64] + [vp] * 15
    L.ccc_ddp_plan_batch.restype = ctypes.c_int
    L.ccc_ddp_plan_batch.argtypes = [vp, ctypes.c_int64] + [vp] * 14
    L._ddp_bound = True


class _ControlData:
    def __init__(self):
        self.u_list = []
        self.x_list = []


class _Solver:
    """Stand-in for the ddp_solver_ member: config(), controlData(), traceDataList()[-1].iter."""

    def __init__(self, cfg):
        self._cfg = cfg
        self._control = _ControlData()
        self.last_iter = 0
        self.last_status = 0  # exit code of the last planOnce
        self.last_warm_start_replaced = False  # the warm-start guard dropped the last planOnce's u_list

    def config(self):
        return self._cfg

    def controlData(self):
        return self._control


class _DdpBase:
    MODEL = 0
    S = 9

    def __init__(self, mass, horizon_dt, horizon_steps, w_run, w_term, w_force, device=0, max_phases=4,
                 max_ridges=MAX_RIDGES):
        L = _lib.load()
        _bind(L)
        self._L = L
        p = _Params()
        p.model, p.mass, p.horizon_dt, p.horizon_steps = self.MODEL, float(mass), float(horizon_dt), int(horizon_steps)
        for a in range(self.S):
            p.w_run[a], p.w_term[a] = float(w_run[a]), float(w_term[a])
        p.w_force = float(w_force)
        # force_scale_limits_ (DdpCentroidal.h:364): a public member the reference reads at EVERY solve
        # (src/DdpCentroidal.cpp:202-210) -- assign to it any time; every plan call pushes its current value (_push_state)
        self.force_scale_limits_ = [0.0, 1e6]
        p.force_scale_limits[0], p.force_scale_limits[1] = self.force_scale_limits_
        p.max_phases = int(max_phases)
        max_ridges = int(max_ridges) or MAX_RIDGES  # (the C-ABI reads 0 as the default stride, 16)
        p.max_ridges = max_ridges
        self.max_ridges_ = max_ridges
        self._w_run, self._w_term, self._w_force = list(w_run), list(w_term), float(w_force)
        h = ctypes.c_void_p()
        _lib.check(L.ccc_ddp_create(ctypes.byref(p), int(device), ctypes.byref(h)))
        self._h = h
        self.device = int(device)
        self.mass_, self.dt_, self.horizon_steps_, self.max_phases_ = float(mass), float(horizon_dt), int(horizon_steps), int(max_phases)
        cfg = Config()
        L.ccc_ddp_default_config(ctypes.byref(cfg))
        self.ddp_solver_ = _Solver(cfg)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.ccc_ddp_destroy(h)
            self._h = None

    # ------------------------------------------------------------------ batched entry points
    def planOnceBatch(self, prob, x0, u_init=None, want_x=False):
        """Host arrays in / out (ccc_ddp_plan_batch).  prob: dict(phase_dim [n,P] i32, phase_vertex [n,P,M,3],
        phase_ridge [n,P,M,3], step_phase [n,N] i32, ref_pos [n,N+1,3] (+ ref_ori [n,N+1,3], inertia [n,3,3]));
        x0 [n,S]; u_init [n,N,M] | None, M = max_ridges.  Returns dict(u [n,N,M], x | None, iters, status, cost,
        exit_code, warm_replaced): status = the C-ABI's status word, exit_code / warm_replaced its two parts."""
        N, P, S, M = self.horizon_steps_, self.max_phases_, self.S, self.max_ridges_
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        n = x0.shape[0]
        arr = dict(phase_dim=np.ascontiguousarray(prob["phase_dim"], dtype=np.int32),
                   phase_vertex=np.ascontiguousarray(prob["phase_vertex"], dtype=np.float64),
                   phase_ridge=np.ascontiguousarray(prob["phase_ridge"], dtype=np.float64),
                   step_phase=np.ascontiguousarray(prob["step_phase"], dtype=np.int32),
                   ref_pos=np.ascontiguousarray(prob["ref_pos"], dtype=np.float64))
        shapes = dict(phase_dim=(n, P), phase_vertex=(n, P, M, 3), phase_ridge=(n, P, M, 3), step_phase=(n, N),
                      ref_pos=(n, N + 1, 3))
        if self.MODEL == 1:
            arr["ref_ori"] = np.ascontiguousarray(prob["ref_ori"], dtype=np.float64)
            arr["inertia"] = np.ascontiguousarray(prob["inertia"], dtype=np.float64)
            # [n,3,3]: one matrix per instance; [n,P,3,3]: one per contact phase (MotionParam::inertia_mat of its steps)
            shapes.update(ref_ori=(n, N + 1, 3), inertia=(n, P, 3, 3) if arr["inertia"].ndim == 4 else (n, 3, 3))
        for k, shp in shapes.items():
            if arr[k].shape != shp:
                raise ValueError("%s must have shape %s, got %s" % (k, shp, arr[k].shape))
        if x0.shape != (n, S):
            raise ValueError("x0 must be [n,%d]" % S)
        ui = None
        if u_init is not None:
            ui = np.ascontiguousarray(u_init, dtype=np.float64)
            if ui.shape != (n, N, M):
                raise ValueError("u_init must be [n,%d,%d]" % (N, M))
        u = np.zeros((n, N, M))
        x = np.zeros((n, N + 1, S)) if want_x else None
        iters = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        cost = np.zeros(n)
        self._push_state(arr.get("inertia"))

        def p(a):
            return None if a is None else ctypes.c_void_p(a.ctypes.data)

        _lib.check(self._L.ccc_ddp_plan_batch(self._h, n, p(arr["phase_dim"]), p(arr["phase_vertex"]),
                                              p(arr["phase_ridge"]), p(arr["step_phase"]), p(arr["ref_pos"]),
                                              p(arr.get("ref_ori")), p(arr.get("inertia")), p(x0), p(ui), p(u), p(x),
                                              p(iters), p(status), p(cost)))
        return dict(u=u, x=x, iters=iters, status=status, cost=cost, exit_code=exit_code(status),
                    warm_replaced=warm_start_replaced(status))

    def _push_state(self, inertia=None):
        """What the reference's solve() reads from the object at every call, pushed to the handle: ddp_solver_->config(),
        force_scale_limits_ (src/DdpCentroidal.cpp:202-210) and the layout of `inertia` (4 dimensions = per phase)."""
        _lib.check(self._L.ccc_ddp_set_config(self._h, ctypes.byref(self.ddp_solver_.config())))
        _lib.check(self._L.ccc_ddp_set_limits(self._h, float(self.force_scale_limits_[0]), float(self.force_scale_limits_[1])))
        if self.MODEL == 1 and inertia is not None:
            ndim = inertia.dim() if hasattr(inertia, "dim") else np.ndim(inertia)
            _lib.check(self._L.ccc_ddp_set_inertia_per_phase(self._h, 1 if ndim == 4 else 0))

    def effective_precision(self):
        """ccc_ddp_effective_precision: 64, whatever Config.precision asked for."""
        return int(self._L.ccc_ddp_effective_precision(self._h))

    def arithmetic(self):
        """ccc_ddp_arithmetic for the object's current solver configuration: 1 = the tile arithmetic
        (oracle/ddp_tile.c), 0 = left-to-right sums (oracle/ddp.c)."""
        _lib.check(self._L.ccc_ddp_set_config(self._h, ctypes.byref(self.ddp_solver_.config())))
        return int(self._L.ccc_ddp_arithmetic(self._h))

    def plan_batch_device(self, prob, x0, u_out, u_init=None, x_out=None, iters=None, status=None, cost=None,
                          stream=None):
        """Device-resident torch tensors (same names/shapes as planOnceBatch), asynchronous on `stream`."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        n = x0.shape[0]
        self._push_state(prob.get("inertia"))

        def p(t):
            return None if t is None else ctypes.c_void_p(t.data_ptr())

        for name in ("phase_dim", "phase_vertex", "phase_ridge", "step_phase", "ref_pos"):
            if not prob[name].is_cuda or not prob[name].is_contiguous():
                raise ValueError("%s must be a contiguous CUDA tensor" % name)
        _lib.check(self._L.ccc_ddp_plan_batch_device(
            self._h, n, p(prob["phase_dim"]), p(prob["phase_vertex"]), p(prob["phase_ridge"]), p(prob["step_phase"]),
            p(prob["ref_pos"]), p(prob.get("ref_ori")), p(prob.get("inertia")), p(x0), p(u_init), p(u_out), p(x_out),
            p(iters), p(status), p(cost), ctypes.c_void_p(stream.cuda_stream)))

    # ------------------------------------------------------------------ reference surface
    def _sample(self, motion_param_func, ref_data_func, current_time):
        """src/DdpCentroidal.cpp:218-229: sample the callbacks at current_time + i*dt and flatten the contact lists
        into contact phases (consecutive steps with the same contact list share a phase).  Returns (planner, prob): this
        object when its tables hold the problem, else a twin with the smallest ridge stride that does (16, 32 or 64; one
        phase per step if need be) created on first need -- the reference takes any contact_list
        (src/DdpCentroidal.cpp:49-60)."""
        N = self.horizon_steps_
        ref_pos, ref_ori = np.zeros((1, N + 1, 3)), np.zeros((1, N + 1, 3))
        step_phase = np.zeros((1, N), dtype=np.int32)
        phases = []
        for i in range(N + 1):
            t = current_time + i * self.dt_
            ref = ref_data_func(t)
            ref_pos[0, i] = ref.pos
            if self.MODEL == 1:
                ref_ori[0, i] = ref.ori
            if i == N:
                break
            mp = motion_param_func(t)
            if mp.contact_list:
                V = np.concatenate([np.asarray(c[0], dtype=np.float64).reshape(-1, 3) for c in mp.contact_list])
                R = np.concatenate([np.asarray(c[1], dtype=np.float64).reshape(-1, 3) for c in mp.contact_list])
            else:
                V, R = np.zeros((0, 3)), np.zeros((0, 3))
            if len(V) > MAX_RIDGES_MULTI:
                raise _lib.CccError(_lib.CCC_ERR_UNSUPPORTED, "%d ridges in one contact list, the kernels are built "
                                    "for %d (four 4-vertex surface contacts)" % (len(V), MAX_RIDGES_MULTI))
            # a phase is a distinct MotionParam: contact list and, for the single-rigid-body model, inertia matrix -- the
            # reference reads motion_param_func_(t).inertia_mat at every step (src/DdpSingleRigidBody.cpp:56-57,120-123)
            In = np.array(mp.inertia_mat, dtype=np.float64).reshape(3, 3) if self.MODEL == 1 else None
            for k, (Vk, Rk, Ik) in enumerate(phases):
                if Vk.shape == V.shape and np.array_equal(Vk, V) and np.array_equal(Rk, R) and (In is None or np.array_equal(Ik, In)):
                    step_phase[0, i] = k
                    break
            else:
                phases.append((V, R, In))
                step_phase[0, i] = len(phases) - 1
        planner = self
        widest = max([len(ph[0]) for ph in phases] + [0])
        if len(phases) > self.max_phases_ or widest > self.max_ridges_:
            need = MAX_RIDGES if widest <= MAX_RIDGES else (MAX_RIDGES_WIDE if widest <= MAX_RIDGES_WIDE else MAX_RIDGES_MULTI)
            planner = self._twin(need)
        P, M = planner.max_phases_, planner.max_ridges_
        prob = dict(phase_dim=np.zeros((1, P), dtype=np.int32), phase_vertex=np.zeros((1, P, M, 3)),
                    phase_ridge=np.zeros((1, P, M, 3)), step_phase=step_phase, ref_pos=ref_pos)
        if self.MODEL == 1:
            prob["ref_ori"], prob["inertia"] = ref_ori, np.tile(np.eye(3), (1, P, 1, 1))
        for k, (V, R, In) in enumerate(phases):
            prob["phase_dim"][0, k] = len(V)
            prob["phase_vertex"][0, k, :len(V)] = V
            prob["phase_ridge"][0, k, :len(V)] = R
            if In is not None:
                prob["inertia"][0, k] = In
        return planner, prob

    def _twin(self, max_ridges):
        twins = self.__dict__.setdefault("_twins", {})
        if max_ridges not in twins:
            t = _DdpBase.__new__(type(self))
            _DdpBase.__init__(t, self.mass_, self.dt_, self.horizon_steps_, self._w_run, self._w_term,
                              self._w_force, self.device, max_phases=self.horizon_steps_, max_ridges=max_ridges)
            twins[max_ridges] = t
        twins[max_ridges].ddp_solver_ = self.ddp_solver_  # one configuration / control data, as the caller sees one solver
        twins[max_ridges].force_scale_limits_ = self.force_scale_limits_  # ... and one force_scale_limits_
        return twins[max_ridges]

    def _plan_once(self, motion_param_func, ref_data_func, x0, u_list, current_time):
        N = self.horizon_steps_
        planner, prob = self._sample(motion_param_func, ref_data_func, current_time)
        M = planner.max_ridges_
        dims = prob["phase_dim"][0][prob["step_phase"][0]]
        u_init = None
        if u_list:
            if len(u_list) != N:
                raise ValueError("u_list must have horizon_steps entries")
            u_init = np.zeros((1, N, M))
            for i, ui in enumerate(u_list):
                ui = np.asarray(ui, dtype=np.float64)
                if len(ui) != dims[i]:
                    raise ValueError("u_list[%d] has %d entries, inputDim is %d" % (i, len(ui), dims[i]))
                u_init[0, i, :len(ui)] = ui
        r = planner.planOnceBatch(prob, x0[None], u_init, want_x=True)
        cd = self.ddp_solver_.controlData()
        cd.u_list = [r["u"][0, i, :dims[i]].copy() for i in range(N)]
        cd.x_list = [r["x"][0, i].copy() for i in range(N + 1)]
        self.ddp_solver_.last_iter = int(r["iters"][0])
        self.ddp_solver_.last_status = int(r["exit_code"][0])
        self.ddp_solver_.last_warm_start_replaced = bool(r["warm_replaced"][0])
        return cd.u_list[0]


class DdpCentroidal(_DdpBase):
    """CCC::DdpCentroidal (include/CCC/DdpCentroidal.h:13-366)."""
    MODEL, S = 0, 9

    class MotionParam:
        def __init__(self, contact_list=None):
            self.contact_list = contact_list or []

    class RefData:
        def __init__(self, pos=(0.0, 0.0, 0.0)):
            self.pos = np.asarray(pos, dtype=np.float64)

    class WeightParam:
        """DdpCentroidal.h:37-81 (same defaults)."""

        def __init__(self, running_pos=(1.0, 1.0, 1.0), running_linear_momentum=(0.0, 0.0, 0.0),
                     running_angular_momentum=(1.0, 1.0, 1.0), running_force=1e-6, terminal_pos=(1.0, 1.0, 1.0),
                     terminal_linear_momentum=(0.0, 0.0, 0.0), terminal_angular_momentum=(1.0, 1.0, 1.0)):
            self.running_pos, self.running_linear_momentum = np.array(running_pos, float), np.array(running_linear_momentum, float)
            self.running_angular_momentum, self.running_force = np.array(running_angular_momentum, float), float(running_force)
            self.terminal_pos, self.terminal_linear_momentum = np.array(terminal_pos, float), np.array(terminal_linear_momentum, float)
            self.terminal_angular_momentum = np.array(terminal_angular_momentum, float)

    class InitialParam:
        """DdpCentroidal.h:295-330."""

        def __init__(self, pos=(0, 0, 0), vel=(0, 0, 0), angular_momentum=(0, 0, 0), u_list=None):
            self.pos, self.vel = np.asarray(pos, float), np.asarray(vel, float)
            self.angular_momentum = np.asarray(angular_momentum, float)
            self.u_list = u_list or []

        def toState(self, mass):
            # src/DdpCentroidal.cpp:186-191
            return np.concatenate([self.pos, mass * self.vel, self.angular_momentum])

    def __init__(self, mass, horizon_dt, horizon_steps, weight_param=None, device=0, max_phases=4,
                 max_ridges=MAX_RIDGES):
        w = weight_param or DdpCentroidal.WeightParam()
        super().__init__(mass, horizon_dt, horizon_steps,
                         np.concatenate([w.running_pos, w.running_linear_momentum, w.running_angular_momentum]),
                         np.concatenate([w.terminal_pos, w.terminal_linear_momentum, w.terminal_angular_momentum]),
                         w.running_force, device, max_phases, max_ridges)

    def planOnce(self, motion_param_func, ref_data_func, initial_param, current_time):
        """src/DdpCentroidal.cpp:213-237: returns controlData().u_list[0] (planned force scales)."""
        return self._plan_once(motion_param_func, ref_data_func, initial_param.toState(self.mass_),
                               initial_param.u_list, current_time)


class DdpSingleRigidBody(_DdpBase):
    """CCC::DdpSingleRigidBody (include/CCC/DdpSingleRigidBody.h)."""
    MODEL, S = 1, 12

    class MotionParam:
        def __init__(self, contact_list=None, inertia_mat=None):
            self.contact_list = contact_list or []
            self.inertia_mat = np.eye(3) if inertia_mat is None else np.asarray(inertia_mat, float)

    class RefData:
        def __init__(self, pos=(0.0, 0.0, 0.0), ori=(0.0, 0.0, 0.0)):
            self.pos, self.ori = np.asarray(pos, float), np.asarray(ori, float)

    class WeightParam:
        """DdpSingleRigidBody.h:52-110 (same defaults)."""

        def __init__(self, running_pos=(1.0,) * 3, running_ori=(1.0,) * 3, running_linear_vel=(0.01,) * 3,
                     running_angular_vel=(0.01,) * 3, running_force=1e-6, terminal_pos=(1.0,) * 3,
                     terminal_ori=(1.0,) * 3, terminal_linear_vel=(0.01,) * 3, terminal_angular_vel=(0.01,) * 3):
            self.running = np.concatenate([np.array(v, float) for v in (running_pos, running_ori, running_linear_vel,
                                                                         running_angular_vel)])
            self.terminal = np.concatenate([np.array(v, float) for v in (terminal_pos, terminal_ori,
                                                                          terminal_linear_vel, terminal_angular_vel)])
            self.running_force = float(running_force)

    class InitialParam:
        def __init__(self, pos=(0, 0, 0), ori=(0, 0, 0), linear_vel=(0, 0, 0), angular_vel=(0, 0, 0), u_list=None):
            self.pos, self.ori = np.asarray(pos, float), np.asarray(ori, float)
            self.linear_vel, self.angular_vel = np.asarray(linear_vel, float), np.asarray(angular_vel, float)
            self.u_list = u_list or []

        def toState(self):
            # src/DdpSingleRigidBody.cpp:253-258
            return np.concatenate([self.pos, self.ori, self.linear_vel, self.angular_vel])

    def __init__(self, mass, horizon_dt, horizon_steps, weight_param=None, device=0, max_phases=4,
                 max_ridges=MAX_RIDGES):
        w = weight_param or DdpSingleRigidBody.WeightParam()
        super().__init__(mass, horizon_dt, horizon_steps, w.running, w.terminal, w.running_force, device, max_phases,
                         max_ridges)

    def planOnce(self, motion_param_func, ref_data_func, initial_param, current_time):
        """src/DdpSingleRigidBody.cpp:283-307."""
        return self._plan_once(motion_param_func, ref_data_func, initial_param.toState(), initial_param.u_list,
                               current_time)
