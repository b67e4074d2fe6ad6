"""Host-side mirror of CCC::DdpZmp over the C-ABI (include/ccc_amd.h, csrc/ddpzmp.hip).

Same names and argument meaning as the reference (/root/reference/include/CCC/DdpZmp.h):
``DdpZmp(mass, horizon_dt, horizon_steps, weight_param)``, ``RefData{zmp [3], com_z}``, ``InitialParam{pos, vel, u_list}``,
``PlannedData{zmp [2], force_z}``, ``planOnce(ref_data_func, initial_param, current_time)``, ``ddp_solver_->config()`` /
``controlData().u_list`` -- plus the batched entry points.
"""
import ctypes

import numpy as np

from . import _lib
from .ddp import Config, _Solver


def _bind(L):
    if getattr(L, "_ddpzmp_bound", False):
        return
    vp, d = ctypes.c_void_p, ctypes.c_double
    L.ccc_ddpzmp_default_config.restype = None
    L.ccc_ddpzmp_default_config.argtypes = [ctypes.POINTER(Config)]
    L.ccc_ddpzmp_create.restype = ctypes.c_int
    L.ccc_ddpzmp_create.argtypes = [d, d, ctypes.c_int, vp, ctypes.c_int, ctypes.POINTER(vp)]
    L.ccc_ddpzmp_destroy.restype = None
    L.ccc_ddpzmp_destroy.argtypes = [vp]
    L.ccc_ddpzmp_set_config.restype = ctypes.c_int
    L.ccc_ddpzmp_set_config.argtypes = [vp, ctypes.POINTER(Config)]
    L.ccc_ddpzmp_workspace_bytes.restype = ctypes.c_int64
    L.ccc_ddpzmp_workspace_bytes.argtypes = [vp, ctypes.c_int64]
    L.ccc_ddpzmp_plan_batch_device.restype = ctypes.c_int
    L.ccc_ddpzmp_plan_batch_device.argtypes = [vp, ctypes.c_int64] + [vp] * 9
    L.ccc_ddpzmp_plan_batch.restype = ctypes.c_int
    L.ccc_ddpzmp_plan_batch.argtypes = [vp, ctypes.c_int64] + [vp] * 8
    L.ccc_ddpzmp_closed_loop_device.restype = ctypes.c_int
    L.ccc_ddpzmp_closed_loop_device.argtypes = [vp, ctypes.c_int64, ctypes.c_int, vp, vp, d, vp, d, d, ctypes.c_int,
                                                ctypes.c_int, vp, d, vp, vp, vp]
    L._ddpzmp_bound = True


class DdpZmp:
    class RefData:
        """DdpZmp.h:19-28."""

        def __init__(self, zmp=(0.0, 0.0, 0.0), com_z=0.0):
            self.zmp = np.asarray(zmp, dtype=np.float64)
            self.com_z = float(com_z)

    class PlannedData:
        """DdpZmp.h:31-40."""

        def __init__(self, zmp, force_z):
            self.zmp = np.asarray(zmp, dtype=np.float64)
            self.force_z = float(force_z)

    class WeightParam:
        """DdpZmp.h:43-88 (same defaults)."""

        def __init__(self, running_com_pos_z=1e2, running_zmp=1e-1, running_force_z=1e-4, terminal_com_pos_xy=1.0,
                     terminal_com_pos_z=1e2, terminal_com_vel=1.0):
            self.running_com_pos_z, self.running_zmp = float(running_com_pos_z), float(running_zmp)
            self.running_force_z, self.terminal_com_pos_xy = float(running_force_z), float(terminal_com_pos_xy)
            self.terminal_com_pos_z, self.terminal_com_vel = float(terminal_com_pos_z), float(terminal_com_vel)

        def as_array(self):
            return np.array([self.running_com_pos_z, self.running_zmp, self.running_force_z, self.terminal_com_pos_xy,
                             self.terminal_com_pos_z, self.terminal_com_vel])

    class InitialParam:
        """DdpZmp.h:247-266: toState() = [pos_x, vel_x, pos_y, vel_y, pos_z, vel_z] (src/DdpZmp.cpp:149-154)."""

        def __init__(self, pos=(0.0, 0.0, 0.0), vel=(0.0, 0.0, 0.0), u_list=None):
            self.pos = np.asarray(pos, dtype=np.float64)
            self.vel = np.asarray(vel, dtype=np.float64)
            self.u_list = u_list

        def toState(self):
            return np.array([self.pos[0], self.vel[0], self.pos[1], self.vel[1], self.pos[2], self.vel[2]])

    def __init__(self, mass, horizon_dt, horizon_steps, weight_param=None, device=0):
        L = _lib.load()
        _bind(L)
        self._L = L
        w = (weight_param or DdpZmp.WeightParam()).as_array()
        h = ctypes.c_void_p()
        _lib.check(L.ccc_ddpzmp_create(float(mass), float(horizon_dt), int(horizon_steps),
                                       ctypes.c_void_p(w.ctypes.data), int(device), ctypes.byref(h)))
        self._h = h
        self.device = int(device)
        self.mass_, self.dt_, self.horizon_steps_ = float(mass), float(horizon_dt), int(horizon_steps)
        cfg = Config()
        L.ccc_ddpzmp_default_config(ctypes.byref(cfg))
        self.ddp_solver_ = _Solver(cfg)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.ccc_ddpzmp_destroy(h)
            self._h = None

    def workspace_bytes(self, n):
        return int(self._L.ccc_ddpzmp_workspace_bytes(self._h, int(n)))

    def _push_config(self):
        _lib.check(self._L.ccc_ddpzmp_set_config(self._h, ctypes.byref(self.ddp_solver_.config())))

    # ------------------------------------------------------------------ batched entry points
    def planOnceBatch(self, ref, x0, u_init=None, want_x=False):
        """Host arrays (ccc_ddpzmp_plan_batch): ref [n,N+1,4] (zmp x, y, z, com_z at t + i dt), x0 [n,6], u_init [n,N,3] |
        None.  Returns dict(u [n,N,3], x | None, iters, status, cost, zmp [n,2], force_z [n])."""
        N = self.horizon_steps_
        ref = np.ascontiguousarray(ref, dtype=np.float64)
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        n = x0.shape[0]
        if ref.shape != (n, N + 1, 4) or x0.shape != (n, 6):
            raise ValueError("ref must be [n,%d,4] and x0 [n,6]" % (N + 1))
        ui = None
        if u_init is not None:
            ui = np.ascontiguousarray(u_init, dtype=np.float64)
            if ui.shape != (n, N, 3):
                raise ValueError("u_init must be [n,%d,3]" % N)
        u = np.zeros((n, N, 3))
        x = np.zeros((n, N + 1, 6)) if want_x else None
        iters = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        cost = np.zeros(n)

        def p(a):
            return None if a is None else ctypes.c_void_p(a.ctypes.data)

        self._push_config()
        _lib.check(self._L.ccc_ddpzmp_plan_batch(self._h, n, p(ref), p(x0), p(ui), p(u), p(x), p(iters), p(status),
                                                 p(cost)))
        return dict(u=u, x=x, iters=iters, status=status, cost=cost, zmp=u[:, 0, :2].copy(), force_z=u[:, 0, 2].copy())

    def plan_batch_device(self, ref, x0, u_init, u_out, x_out=None, iters=None, status=None, cost=None, stream=None):
        """Device-resident torch tensors, asynchronous on `stream` (ccc_ddpzmp_plan_batch_device)."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(self.device)

        def p(t):
            return None if t is None else ctypes.c_void_p(t.data_ptr())

        self._push_config()
        _lib.check(self._L.ccc_ddpzmp_plan_batch_device(self._h, x0.shape[0], p(ref), p(x0), p(u_init), p(u_out), p(x_out),
                                                        p(iters), p(status), p(cost),
                                                        ctypes.c_void_p(stream.cuda_stream)))

    def closed_loop_device(self, knot_t, knot_zmp, com_height, state, t0, sim_dt, cycles, disturb_times=(), disturb_impulse=0.0,
                           stats=None, log=None, stream=None):
        """The control loop of TestDdpZmp.cpp:70-125 for n instances in one launch (ccc_ddpzmp_closed_loop_device).
        Device tensors, instance-fastest: knot_t [K,n], knot_zmp [K,2,n], state [6,n] (in / out), stats [4,n] | None,
        log [cycles,3,n] | None; disturb_times: host sequence (<= 8)."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(self.device)

        def p(t):
            return None if t is None else ctypes.c_void_p(t.data_ptr())

        dt_ = np.ascontiguousarray(disturb_times, dtype=np.float64)
        self._push_config()
        _lib.check(self._L.ccc_ddpzmp_closed_loop_device(
            self._h, state.shape[1], knot_t.shape[0], p(knot_t), p(knot_zmp), float(com_height), p(state), float(t0),
            float(sim_dt), int(cycles), int(dt_.size), ctypes.c_void_p(dt_.ctypes.data) if dt_.size else None,
            float(disturb_impulse), p(stats), p(log), ctypes.c_void_p(stream.cuda_stream)))

    # ------------------------------------------------------------------ the reference's call
    def planOnce(self, ref_data_func, initial_param, current_time):
        """CCC::DdpZmp::planOnce (DdpZmp.h:290-292, src/DdpZmp.cpp:156-174)."""
        N = self.horizon_steps_
        ref = np.zeros((1, N + 1, 4))
        for i in range(N + 1):
            rd = ref_data_func(current_time + i * self.dt_)
            ref[0, i, :3] = rd.zmp
            ref[0, i, 3] = rd.com_z
        ui = None
        if initial_param.u_list is not None and len(initial_param.u_list) > 0:
            ui = np.asarray(initial_param.u_list, dtype=np.float64).reshape(1, N, 3)
        r = self.planOnceBatch(ref, initial_param.toState()[None], ui)
        self.ddp_solver_.controlData().u_list = [r["u"][0, i].copy() for i in range(N)]
        self.ddp_solver_.last_iter = int(r["iters"][0])
        return DdpZmp.PlannedData(r["zmp"][0], r["force_z"][0])
