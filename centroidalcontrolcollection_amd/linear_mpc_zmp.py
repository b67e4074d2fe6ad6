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

    def last_schedule(self):
        """What the last plan call's schedule came from (ccc_zmp_last_schedule): "last call's pivot counts", "predicted
        pivot counts" or "none"."""
        return self._L.ccc_zmp_last_schedule(self._h).decode()

    def last_kernel(self):
        """Name of the kernel the last plan call launched (ccc_zmp_last_kernel): what a profile of that call lists."""
        return self._L.ccc_zmp_last_kernel(self._h).decode()

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

    def plan_batch_pinned(self, x0, zlim, control_dt, zmp, status=None, jerk=None):
        """ccc_zmp_plan_batch on PINNED host tensors (torch ... pin_memory=True): for N <= 32 the kernel reads the
        inputs and writes the results in the caller's page-locked memory, no copy (the p50 path).
        Synchronous."""
        import torch

        N = self.horizon_steps_
        n = x0.shape[0]
        for name, t, shape, dt in (("x0", x0, (n, 2, 3), torch.float64), ("zlim", zlim, (n, 2, 2, N), torch.float64),
                                   ("zmp", zmp, (n, 2), torch.float64), ("status", status, (n, 2), torch.int32),
                                   ("jerk", jerk, (n, 2, N), torch.float64)):
            if t is None:
                continue
            if t.is_cuda or not t.is_pinned() or t.dtype != dt or tuple(t.shape) != shape or not t.is_contiguous():
                raise ValueError("%s must be a contiguous pinned host %s tensor of shape %s" % (name, dt, shape))
        _lib.check(self._L.ccc_zmp_plan_batch(
            self._h, n, ctypes.cast(x0.data_ptr(), _lib.c_double_p), ctypes.cast(zlim.data_ptr(), _lib.c_double_p),
            float(control_dt), ctypes.cast(zmp.data_ptr(), _lib.c_double_p),
            ctypes.cast(jerk.data_ptr(), _lib.c_double_p) if jerk is not None else None,
            ctypes.cast(status.data_ptr(), _lib.c_int32_p) if status is not None else None))

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

    # ------------------------------------------------------------------ either side of the path, on the device
    def _timeline_args(self, tl):
        """tl: dict of contiguous CUDA tensors foot0 [n,2,2] f64, foot_pos [n,K,2] f64, foot_id [n,K] i32,
        swing_start / swing_end [n,K] f64 (the footstep timeline format of include/ccc_amd.h)."""
        import torch

        n, K = tl["foot_id"].shape
        for name, dt in (("foot0", torch.float64), ("foot_pos", torch.float64), ("foot_id", torch.int32),
                         ("swing_start", torch.float64), ("swing_end", torch.float64)):
            t = tl[name]
            if t.dtype != dt or not t.is_cuda or not t.is_contiguous():
                raise ValueError("%s must be a contiguous %s CUDA tensor" % (name, dt))
        p = [ctypes.c_void_p(tl[k].data_ptr()) for k in ("foot0", "foot_pos", "foot_id", "swing_start", "swing_end")]
        return n, K, p

    def sample_limits_device(self, timeline, zlim, t_eval=None, t_common=0.0, foot_size=None, stream=None):
        """ccc_zmp_sample_limits_device: zlim [n,2,2,N] <- the limits FootstepManager::makeLinearMpcZmpRefData returns at
        t + i*horizon_dt, t = t_eval[k] (CUDA tensor) or t_common."""
        import torch

        L = self._L
        L.ccc_zmp_sample_limits_device.restype = ctypes.c_int
        L.ccc_zmp_sample_limits_device.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int] + [ctypes.c_void_p] * 7 + [
            ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
        n, K, p = self._timeline_args(timeline)
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        fs = (ctypes.c_double * 2)(*foot_size) if foot_size is not None else None
        _lib.check(L.ccc_zmp_sample_limits_device(self._h, n, K, *p, fs,
                                                  ctypes.c_void_p(t_eval.data_ptr()) if t_eval is not None else None,
                                                  float(t_common), ctypes.c_void_p(zlim.data_ptr()),
                                                  ctypes.c_void_p(stream.cuda_stream)))

    def closed_loop_device(self, timeline, com_state, planned_zmp, t0, sim_dt, cycles, disturb_times=(),
                           disturb_impulse=0.0, violations=None, traj_com=None, traj_zmp=None, foot_size=None,
                           stream=None):
        """ccc_zmp_closed_loop_device: `cycles` control cycles of TestLinearMpcZmp.cpp:55-102 for every instance, on the
        device.  com_state [n,2,2] and planned_zmp [n,2] are updated in place; returns the time after the last cycle."""
        import torch

        L = self._L
        L.ccc_zmp_closed_loop_device.restype = ctypes.c_int
        L.ccc_zmp_closed_loop_device.argtypes = ([ctypes.c_void_p, ctypes.c_int64, ctypes.c_int] + [ctypes.c_void_p] * 8
                                                 + [ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                                    ctypes.c_void_p, ctypes.c_double] + [ctypes.c_void_p] * 7)
        n, K, p = self._timeline_args(timeline)
        N = self.horizon_steps_
        dev = com_state.device
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        wx0 = torch.empty((n, 2, 3), dtype=torch.float64, device=dev)
        wz = torch.empty((n, 2, 2, N), dtype=torch.float64, device=dev)
        fs = (ctypes.c_double * 2)(*foot_size) if foot_size is not None else None
        dts = (ctypes.c_double * max(1, len(disturb_times)))(*disturb_times)
        t_end = ctypes.c_double(0.0)

        def q(t):
            return None if t is None else ctypes.c_void_p(t.data_ptr())

        _lib.check(L.ccc_zmp_closed_loop_device(self._h, n, K, *p, fs, q(com_state), q(planned_zmp), float(t0),
                                                float(sim_dt), int(cycles), len(disturb_times), dts,
                                                float(disturb_impulse), q(wx0), q(wz), q(violations), q(traj_com),
                                                q(traj_zmp), ctypes.byref(t_end), ctypes.c_void_p(stream.cuda_stream)))
        self._keep = (wx0, wz)  # the launches are asynchronous: keep the workspaces alive
        return t_end.value
