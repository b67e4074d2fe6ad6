"""Host-side mirror of CCC::LinearMpcZmp over the C-ABI (include/ccc_amd.h).

Keeps the reference's names and argument meaning (/root/reference/include/CCC/LinearMpcZmp.h:97-165):
``LinearMpcZmp(com_height, horizon_duration, horizon_dt)``, ``RefData.zmp_limits``, ``InitialParam.{pos,vel,acc}``,
``planOnce(ref_data_func, initial_param, current_time, control_dt=-1)``; and adds the batched entry points
the MI355X path exists for.  All arithmetic happens in the HIP kernels of csrc/zmp.hip.
"""
import ctypes

import numpy as np

from . import _lib


class RefData:
    """LinearMpcZmp::RefData (LinearMpcZmp.h:105-111): min/max limits of ZMP, each a 2-vector (x, y)."""

    def __init__(self, zmp_min=(0.0, 0.0), zmp_max=(0.0, 0.0)):
        self.zmp_limits = [np.asarray(zmp_min, dtype=np.float64), np.asarray(zmp_max, dtype=np.float64)]


class InitialParam:
    """LinearMpcZmp::InitialParam (LinearMpcZmp.h:113-125)."""

    def __init__(self, pos=(0.0, 0.0), vel=(0.0, 0.0), acc=(0.0, 0.0)):
        self.pos = np.asarray(pos, dtype=np.float64)
        self.vel = np.asarray(vel, dtype=np.float64)
        self.acc = np.asarray(acc, dtype=np.float64)


class LinearMpcZmp:
    RefData = RefData
    InitialParam = InitialParam

    def __init__(self, com_height, horizon_duration, horizon_dt, device=0):
        L = _lib.load()
        h = ctypes.c_void_p()
        _lib.check(L.ccc_zmp_create(float(com_height), float(horizon_duration), float(horizon_dt), int(device),
                                    ctypes.byref(h)))
        self._h = h
        self._L = L
        self.device = int(device)
        self.com_height = float(com_height)
        self.horizon_dt_ = float(horizon_dt)
        self.horizon_steps_ = L.ccc_zmp_horizon_steps(h)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.ccc_zmp_destroy(h)
            self._h = None

    # ------------------------------------------------------------------ model access
    def seq(self):
        """(A_seq [N,3], B_seq [N,N]) -- InvariantSequentialExtension<3,1,1>::A_seq_/B_seq_."""
        N = self.horizon_steps_
        A = np.empty((N, 3))
        B = np.empty((N, N))
        _lib.check(self._L.ccc_zmp_get_seq(self._h, A.ctypes.data_as(_lib.c_double_p),
                                           B.ctypes.data_as(_lib.c_double_p)))
        return A, B

    # ------------------------------------------------------------------ reference surface
    def sample(self, ref_data_func, current_time):
        """src/LinearMpcZmp.cpp:86-98: sample the callback at current_time + i*horizon_dt into [2,2,N]."""
        N = self.horizon_steps_
        zlim = np.empty((2, 2, N))
        for i in range(N):
            ref = ref_data_func(current_time + i * self.horizon_dt_)
            for j in range(2):
                zlim[0, j, i] = ref.zmp_limits[j][0]
                zlim[1, j, i] = ref.zmp_limits[j][1]
        return zlim

    def planOnce(self, ref_data_func, initial_param, current_time, control_dt=-1.0):
        """CCC::LinearMpcZmp::planOnce (LinearMpcZmp.h:151-154): returns the planned ZMP (2-vector)."""
        zlim = self.sample(ref_data_func, current_time)[None]
        x0 = np.array([[[initial_param.pos[0], initial_param.vel[0], initial_param.acc[0]],
                        [initial_param.pos[1], initial_param.vel[1], initial_param.acc[1]]]])
        return self.planOnceBatch(x0, zlim, control_dt)["zmp"][0]

    # ------------------------------------------------------------------ batched entry points
    def planOnceBatch(self, x0, zlim, control_dt=-1.0, want_jerk=False):
        """Host arrays in, host arrays out (ccc_zmp_plan_batch).

        x0 [n,2,3], zlim [n,2,2,N] -> dict(zmp [n,2], jerk [n,2,N] | None, status [n,2], pivots [n,2])."""
        N = self.horizon_steps_
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        zlim = np.ascontiguousarray(zlim, dtype=np.float64)
        n = x0.shape[0]
        if x0.shape != (n, 2, 3) or zlim.shape != (n, 2, 2, N):
            raise ValueError("x0 must be [n,2,3] and zlim [n,2,2,%d]; got %s, %s" % (N, x0.shape, zlim.shape))
        zmp = np.empty((n, 2))
        jerk = np.empty((n, 2, N)) if want_jerk else None
        status = np.empty((n, 2), dtype=np.int32)
        _lib.check(self._L.ccc_zmp_plan_batch(
            self._h, n, x0.ctypes.data_as(_lib.c_double_p), zlim.ctypes.data_as(_lib.c_double_p), float(control_dt),
            zmp.ctypes.data_as(_lib.c_double_p),
            jerk.ctypes.data_as(_lib.c_double_p) if want_jerk else None, status.ctypes.data_as(_lib.c_int32_p)))
        return dict(zmp=zmp, jerk=jerk, status=status & 0xff, pivots=status >> 8)

    def plan_batch_device(self, x0, zlim, control_dt, zmp, jerk=None, status=None, stream=None):
        """Device-resident torch tensors in/out, asynchronous on `stream` (ccc_zmp_plan_batch_device).

        x0 [n,2,3] f64, zlim [n,2,2,N] f64, zmp [n,2] f64, jerk [n,2,N] f64 | None, status [n,2] i32 | None."""
        import torch

        N = self.horizon_steps_
        n = x0.shape[0]
        for name, t, shape, dt in (("x0", x0, (n, 2, 3), torch.float64), ("zlim", zlim, (n, 2, 2, N), torch.float64),
                                   ("zmp", zmp, (n, 2), torch.float64), ("jerk", jerk, (n, 2, N), torch.float64),
                                   ("status", status, (n, 2), torch.int32)):
            if t is None:
                continue
            if not t.is_cuda or t.device.index != self.device or t.dtype != dt or tuple(t.shape) != shape \
                    or not t.is_contiguous():
                raise ValueError("%s must be a contiguous %s tensor of shape %s on cuda:%d" %
                                 (name, dt, shape, self.device))
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        sp = ctypes.c_void_p(stream.cuda_stream)
        _lib.check(self._L.ccc_zmp_plan_batch_device(
            self._h, n, ctypes.c_void_p(x0.data_ptr()), ctypes.c_void_p(zlim.data_ptr()), float(control_dt),
            ctypes.c_void_p(zmp.data_ptr()), ctypes.c_void_p(jerk.data_ptr()) if jerk is not None else None,
            ctypes.c_void_p(status.data_ptr()) if status is not None else None, sp))
