"""Host-side mirror of CCC::LinearMpcXY over the C-ABI (include/ccc_amd.h, csrc/xy.hip).

Same names and argument meaning as the reference (/root/reference/include/CCC/LinearMpcXY.h:27-257):
``LinearMpcXY(mass, horizon_dt, horizon_steps, weight_param)``, ``MotionParam{com_z, total_force_z, contact_list}``,
``RefData{pos, vel, angular_momentum}``, ``InitialParam{pos, vel, angular_momentum}``,
``planOnce(motion_param_func, ref_data_func, initial_param, current_time)`` returning the force scales of the first
horizon step -- plus the batched entry points.  A contact is its flattened ridge list ``(vertex [m,3], ridge [m,3])``.
"""
import ctypes

import numpy as np

from . import _lib

MAX_RIDGES = 16  # default ridge slots per step (one surface contact)
MAX_RIDGES_WIDE = 32  # max_ridges=32: two surface contacts per step (double support)
MAX_RIDGES_MULTI = 64  # max_ridges=64: up to four surface contacts per step (feet + hands)
MAX_STEPS = 20  # horizon steps both kernels take; beyond (<= 256) the stage-recursion kernel alone


class _Params(ctypes.Structure):
    _fields_ = [("mass", ctypes.c_double), ("horizon_dt", ctypes.c_double), ("horizon_steps", ctypes.c_int),
                ("w_lmi", ctypes.c_double * 2), ("w_lm", ctypes.c_double * 2), ("w_am", ctypes.c_double * 2),
                ("w_force", ctypes.c_double), ("max_ridges", ctypes.c_int)]


def _bind(L):
    if getattr(L, "_xy_bound", False):
        return
    vp = ctypes.c_void_p
    L.ccc_xy_create.restype = ctypes.c_int
    L.ccc_xy_create.argtypes = [ctypes.POINTER(_Params), ctypes.c_int, ctypes.POINTER(vp)]
    L.ccc_xy_destroy.restype = None
    L.ccc_xy_destroy.argtypes = [vp]
    L.ccc_xy_plan_batch_device.restype = ctypes.c_int
    L.ccc_xy_plan_batch_device.argtypes = [vp, ctypes.c_int64] + [vp] * 11
    L.ccc_xy_plan_batch.restype = ctypes.c_int
    L.ccc_xy_plan_batch.argtypes = [vp, ctypes.c_int64] + [vp] * 10
    L._xy_bound = True


class LinearMpcXY:
    class MotionParam:
        """LinearMpcXY.h:38-52."""

        def __init__(self, com_z=1.0, total_force_z=0.0, contact_list=None):
            self.com_z, self.total_force_z = float(com_z), float(total_force_z)
            self.contact_list = contact_list or []

    class RefData:
        """LinearMpcXY.h:76-101."""

        def __init__(self, pos=(0.0, 0.0), vel=(0.0, 0.0), angular_momentum=(0.0, 0.0)):
            self.pos, self.vel = np.asarray(pos, float), np.asarray(vel, float)
            self.angular_momentum = np.asarray(angular_momentum, float)

        def toOutput(self, mass):
            # src/LinearMpcXY.cpp:33-38
            return np.array([mass * self.pos[0], mass * self.vel[0], mass * self.pos[1], mass * self.vel[1],
                             self.angular_momentum[0], self.angular_momentum[1]])

    class InitialParam(RefData):
        """LinearMpcXY.h:55-73 (same fields as RefData)."""

        def toState(self, mass):
            # src/LinearMpcXY.cpp:26-31
            return self.toOutput(mass)

    class WeightParam:
        """LinearMpcXY.h:104-142 (same defaults)."""

        def __init__(self, linear_momentum_integral=(1.0, 1.0), linear_momentum=(0.0, 0.0),
                     angular_momentum=(1.0, 1.0), force=1e-5):
            self.linear_momentum_integral = np.asarray(linear_momentum_integral, float)
            self.linear_momentum = np.asarray(linear_momentum, float)
            self.angular_momentum = np.asarray(angular_momentum, float)
            self.force = float(force)

    def __init__(self, mass, horizon_dt, horizon_steps, weight_param=None, device=0, max_ridges=MAX_RIDGES):
        L = _lib.load()
        _bind(L)
        self._L = L
        w = weight_param or LinearMpcXY.WeightParam()
        p = _Params()
        p.mass, p.horizon_dt, p.horizon_steps = float(mass), float(horizon_dt), int(horizon_steps)
        for a in range(2):
            p.w_lmi[a], p.w_lm[a], p.w_am[a] = w.linear_momentum_integral[a], w.linear_momentum[a], w.angular_momentum[a]
        p.w_force = w.force
        p.max_ridges = int(max_ridges)
        self.max_ridges_, self._weight_param = int(max_ridges), w
        h = ctypes.c_void_p()
        _lib.check(L.ccc_xy_create(ctypes.byref(p), int(device), ctypes.byref(h)))
        self._h = h
        self.device = int(device)
        self.mass_, self.horizon_dt_, self.horizon_steps_ = float(mass), float(horizon_dt), int(horizon_steps)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.ccc_xy_destroy(h)
            self._h = None

    def planOnceBatch(self, prob, x0, want_all=False):
        """Host arrays (ccc_xy_plan_batch).  prob: dict(dim [n,N] i32, vertex/ridge [n,N,M,3], com_z [n,N],
        total_force_z [n,N], ref_out [n,N,6]); x0 [n,6]; M = max_ridges.  Returns dict(u0 [n,M], lam [n,N,M] | None,
        status, pivots)."""
        N, M = self.horizon_steps_, self.max_ridges_
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        n = x0.shape[0]
        arr = dict(dim=np.ascontiguousarray(prob["dim"], dtype=np.int32),
                   vertex=np.ascontiguousarray(prob["vertex"], dtype=np.float64),
                   ridge=np.ascontiguousarray(prob["ridge"], dtype=np.float64),
                   com_z=np.ascontiguousarray(prob["com_z"], dtype=np.float64),
                   total_force_z=np.ascontiguousarray(prob["total_force_z"], dtype=np.float64),
                   ref_out=np.ascontiguousarray(prob["ref_out"], dtype=np.float64))
        shapes = dict(dim=(n, N), vertex=(n, N, M, 3), ridge=(n, N, M, 3), com_z=(n, N), total_force_z=(n, N),
                      ref_out=(n, N, 6))
        for k, shp in shapes.items():
            if arr[k].shape != shp:
                raise ValueError("%s must have shape %s, got %s" % (k, shp, arr[k].shape))
        if x0.shape != (n, 6):
            raise ValueError("x0 must be [n,6]")
        u0 = np.zeros((n, M))
        lam = np.zeros((n, N, M)) if want_all else None
        status = np.zeros(n, dtype=np.int32)

        def p(a):
            return None if a is None else ctypes.c_void_p(a.ctypes.data)

        _lib.check(self._L.ccc_xy_plan_batch(self._h, n, p(arr["dim"]), p(arr["vertex"]), p(arr["ridge"]),
                                             p(arr["com_z"]), p(arr["total_force_z"]), p(arr["ref_out"]), p(x0), p(u0),
                                             p(lam), p(status)))
        return dict(u0=u0, lam=lam, status=status & 0xff, pivots=status >> 8)

    def plan_batch_device(self, prob, x0, u0, lambda_all=None, status=None, stream=None):
        """Device-resident torch tensors (same names/shapes), asynchronous on `stream`."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(self.device)

        def p(t):
            return None if t is None else ctypes.c_void_p(t.data_ptr())

        _lib.check(self._L.ccc_xy_plan_batch_device(
            self._h, x0.shape[0], p(prob["dim"]), p(prob["vertex"]), p(prob["ridge"]), p(prob["com_z"]),
            p(prob["total_force_z"]), p(prob["ref_out"]), p(x0), p(u0), p(lambda_all), p(status),
            ctypes.c_void_p(stream.cuda_stream)))

    def planOnce(self, motion_param_func, ref_data_func, initial_param, current_time):
        """CCC::LinearMpcXY::planOnce (LinearMpcXY.h:224-227, src/LinearMpcXY.cpp:96-114)."""
        N, M = self.horizon_steps_, MAX_RIDGES_MULTI
        prob = dict(dim=np.zeros((1, N), dtype=np.int32), vertex=np.zeros((1, N, M, 3)), ridge=np.zeros((1, N, M, 3)),
                    com_z=np.zeros((1, N)), total_force_z=np.zeros((1, N)), ref_out=np.zeros((1, N, 6)))
        for i in range(N):
            t = current_time + i * self.horizon_dt_
            mp = motion_param_func(t)
            if mp.contact_list:
                V = np.concatenate([np.asarray(c[0], float).reshape(-1, 3) for c in mp.contact_list])
                R = np.concatenate([np.asarray(c[1], float).reshape(-1, 3) for c in mp.contact_list])
            else:
                V, R = np.zeros((0, 3)), np.zeros((0, 3))
            if len(V) > M:
                raise _lib.CccError(_lib.CCC_ERR_UNSUPPORTED, "%d ridges in one contact list, the kernels are built for "
                                    "%d (four 4-vertex surface contacts)" % (len(V), M))
            prob["dim"][0, i] = len(V)
            prob["vertex"][0, i, :len(V)], prob["ridge"][0, i, :len(V)] = V, R
            prob["com_z"][0, i], prob["total_force_z"][0, i] = mp.com_z, mp.total_force_z
            prob["ref_out"][0, i] = ref_data_func(t).toOutput(self.mass_)
        # the reference takes any contact_list (src/LinearMpcXY.cpp:69-82): what this object's ridge slots do not hold goes
        # to a twin with the smallest ridge stride that does (32 or 64), created on first need
        planner = self
        if prob["dim"].max() > self.max_ridges_:
            need = MAX_RIDGES_WIDE if prob["dim"].max() <= MAX_RIDGES_WIDE else MAX_RIDGES_MULTI
            twins = self.__dict__.setdefault("_twins", {})
            if need not in twins:
                twins[need] = LinearMpcXY(self.mass_, self.horizon_dt_, N, self._weight_param, self.device, need)
            planner = twins[need]
        Mp = planner.max_ridges_
        prob["vertex"], prob["ridge"] = prob["vertex"][:, :, :Mp], prob["ridge"][:, :, :Mp]
        r = planner.planOnceBatch(prob, initial_param.toState(self.mass_)[None])
        return r["u0"][0, :prob["dim"][0, 0]].copy()
