"""Host-side mirror of CCC::IntrinsicallyStableMpc over the C-ABI (include/ccc_amd.h, csrc/ism.hip).

Same names and argument meaning as the reference (/root/reference/include/CCC/IntrinsicallyStableMpc.h:127-193):
``IntrinsicallyStableMpc(com_height, horizon_duration, horizon_dt, weight_param)``, ``RefData{zmp, zmp_limits}``,
``InitialParam{capture_point, planned_zmp}``, ``WeightParam{zmp, zmp_vel}``,
``planOnce(ref_data_func, initial_param, current_time, control_dt)`` -- plus the batched entry points.
"""
import ctypes

import numpy as np

from . import _lib


def _bind(L):
    if getattr(L, "_ism_bound", False):
        return
    vp, d = ctypes.c_void_p, ctypes.c_double
    L.ccc_ism_create.restype = ctypes.c_int
    L.ccc_ism_create.argtypes = [d, d, d, d, d, ctypes.c_int, ctypes.POINTER(vp)]
    L.ccc_ism_destroy.restype = None
    L.ccc_ism_destroy.argtypes = [vp]
    L.ccc_ism_horizon_steps.restype = ctypes.c_int
    L.ccc_ism_horizon_steps.argtypes = [vp]
    L.ccc_ism_plan_batch_device.restype = ctypes.c_int
    L.ccc_ism_plan_batch_device.argtypes = [vp, ctypes.c_int64, vp, vp, d, vp, vp, vp, vp]
    L.ccc_ism_plan_batch.restype = ctypes.c_int
    L.ccc_ism_plan_batch.argtypes = [vp, ctypes.c_int64, vp, vp, d, vp, vp, vp]
    L._ism_bound = True


class IntrinsicallyStableMpc:
    class RefData:
        """IntrinsicallyStableMpc.h:133-141."""

        def __init__(self, zmp=(0.0, 0.0), zmp_min=(0.0, 0.0), zmp_max=(0.0, 0.0)):
            self.zmp = np.asarray(zmp, float)
            self.zmp_limits = [np.asarray(zmp_min, float), np.asarray(zmp_max, float)]

    class InitialParam:
        """IntrinsicallyStableMpc.h:144-153."""

        def __init__(self, capture_point=(0.0, 0.0), planned_zmp=(0.0, 0.0)):
            self.capture_point = np.asarray(capture_point, float)
            self.planned_zmp = np.asarray(planned_zmp, float)

    class WeightParam:
        """IntrinsicallyStableMpc.h:42-55 (same defaults)."""

        def __init__(self, zmp=1.0, zmp_vel=1e-3):
            self.zmp, self.zmp_vel = float(zmp), float(zmp_vel)

    def __init__(self, com_height, horizon_duration, horizon_dt, weight_param=None, device=0):
        L = _lib.load()
        _bind(L)
        self._L = L
        w = weight_param or IntrinsicallyStableMpc.WeightParam()
        h = ctypes.c_void_p()
        _lib.check(L.ccc_ism_create(float(com_height), float(horizon_duration), float(horizon_dt), w.zmp, w.zmp_vel,
                                    int(device), ctypes.byref(h)))
        self._h = h
        self.device = int(device)
        self.horizon_dt_ = float(horizon_dt)
        self.horizon_steps_ = L.ccc_ism_horizon_steps(h)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.ccc_ism_destroy(h)
            self._h = None

    def planOnceBatch(self, init, ref, control_dt=-1.0, want_vel=False):
        """Host arrays (ccc_ism_plan_batch): init [n,2,2] (capture_point, planned_zmp per axis), ref [n,2,3,N] (rows ref
        zmp, zmin, zmax per axis).  Returns dict(zmp [n,2], vel [n,2,N] | None, status [n,2], pivots [n,2])."""
        N = self.horizon_steps_
        init = np.ascontiguousarray(init, dtype=np.float64)
        ref = np.ascontiguousarray(ref, dtype=np.float64)
        n = init.shape[0]
        if init.shape != (n, 2, 2) or ref.shape != (n, 2, 3, N):
            raise ValueError("init must be [n,2,2] and ref [n,2,3,%d]" % N)
        zmp = np.zeros((n, 2))
        vel = np.zeros((n, 2, N)) if want_vel else None
        status = np.zeros((n, 2), dtype=np.int32)

        def p(a):
            return None if a is None else ctypes.c_void_p(a.ctypes.data)

        _lib.check(self._L.ccc_ism_plan_batch(self._h, n, p(init), p(ref), float(control_dt), p(zmp), p(vel), p(status)))
        return dict(zmp=zmp, vel=vel, status=status & 0xff, pivots=status >> 8)

    def plan_batch_device(self, init, ref, control_dt, zmp, vel=None, status=None, stream=None):
        """Device-resident torch tensors (same shapes), asynchronous on `stream`."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(self.device)

        def p(t):
            return None if t is None else ctypes.c_void_p(t.data_ptr())

        _lib.check(self._L.ccc_ism_plan_batch_device(self._h, init.shape[0], p(init), p(ref), float(control_dt), p(zmp),
                                                     p(vel), p(status), ctypes.c_void_p(stream.cuda_stream)))

    def planOnce(self, ref_data_func, initial_param, current_time, control_dt=-1.0):
        """CCC::IntrinsicallyStableMpc::planOnce (IntrinsicallyStableMpc.h:178-181, src/IntrinsicallyStableMpc.cpp:106-139)."""
        N = self.horizon_steps_
        ref = np.zeros((1, 2, 3, N))
        for i in range(N):
            rd = ref_data_func(current_time + i * self.horizon_dt_)
            ref[0, :, 0, i], ref[0, :, 1, i], ref[0, :, 2, i] = rd.zmp, rd.zmp_limits[0], rd.zmp_limits[1]
        init = np.stack([initial_param.capture_point, initial_param.planned_zmp], axis=1)[None]
        return self.planOnceBatch(init, ref, control_dt)["zmp"][0].copy()
