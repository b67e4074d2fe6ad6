"""Host-side mirror of CCC::LinearMpcZ over the C-ABI (include/ccc_amd.h, csrc/z.hip).

Same names and argument meaning as the reference (/root/reference/include/CCC/LinearMpcZ.h:14-179):
``LinearMpcZ(mass, horizon_dt, horizon_steps, weight_param)``, ``InitialParam = (pos, vel)``, ``WeightParam{pos, force}``,
``planOnce(contact_func, ref_pos_func, initial_param, current_time)`` returning the planned vertical force -- plus the
batched entry points.
"""
import ctypes

import numpy as np

from . import _lib


def _bind(L):
    if getattr(L, "_z_bound", False):
        return
    vp, d = ctypes.c_void_p, ctypes.c_double
    L.ccc_z_create.restype = ctypes.c_int
    L.ccc_z_create.argtypes = [d, d, ctypes.c_int, d, d, ctypes.c_int, ctypes.POINTER(vp)]
    L.ccc_z_destroy.restype = None
    L.ccc_z_destroy.argtypes = [vp]
    L.ccc_z_plan_batch_device.restype = ctypes.c_int
    L.ccc_z_plan_batch_device.argtypes = [vp, ctypes.c_int64] + [vp] * 7
    L.ccc_z_plan_batch.restype = ctypes.c_int
    L.ccc_z_plan_batch.argtypes = [vp, ctypes.c_int64] + [vp] * 6
    L._z_bound = True


class LinearMpcZ:
    class WeightParam:
        """LinearMpcZ.h:34-47 (same defaults)."""

        def __init__(self, pos=1.0, force=1e-7):
            self.pos, self.force = float(pos), float(force)

    def __init__(self, mass, horizon_dt, horizon_steps, weight_param=None, device=0):
        L = _lib.load()
        _bind(L)
        self._L = L
        w = weight_param or LinearMpcZ.WeightParam()
        h = ctypes.c_void_p()
        _lib.check(L.ccc_z_create(float(mass), float(horizon_dt), int(horizon_steps), w.pos, w.force, int(device),
                                  ctypes.byref(h)))
        self._h = h
        self.device = int(device)
        self.mass_, self.horizon_dt_, self.horizon_steps_ = float(mass), float(horizon_dt), int(horizon_steps)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.ccc_z_destroy(h)
            self._h = None

    def planOnceBatch(self, contact, ref_pos, x0, want_all=False):
        """Host arrays (ccc_z_plan_batch): contact [n,N] (bool/int), ref_pos [n,N], x0 [n,2] (height, velocity).
        Returns dict(force [n], force_all [n,N] | None, status [n], pivots [n])."""
        N = self.horizon_steps_
        contact = np.ascontiguousarray(np.asarray(contact) != 0, dtype=np.int32)
        ref_pos = np.ascontiguousarray(ref_pos, dtype=np.float64)
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        n = x0.shape[0]
        if contact.shape != (n, N) or ref_pos.shape != (n, N) or x0.shape != (n, 2):
            raise ValueError("contact/ref_pos must be [n,%d] and x0 [n,2]" % N)
        force = np.zeros(n)
        fall = np.zeros((n, N)) if want_all else None
        status = np.zeros(n, dtype=np.int32)

        def p(a):
            return None if a is None else ctypes.c_void_p(a.ctypes.data)

        _lib.check(self._L.ccc_z_plan_batch(self._h, n, p(contact), p(ref_pos), p(x0), p(force), p(fall), p(status)))
        return dict(force=force, force_all=fall, status=status & 0xff, pivots=status >> 8)

    def plan_batch_device(self, contact, ref_pos, x0, force, force_all=None, status=None, stream=None):
        """Device-resident torch tensors (contact int32), asynchronous on `stream`."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(self.device)

        def p(t):
            return None if t is None else ctypes.c_void_p(t.data_ptr())

        _lib.check(self._L.ccc_z_plan_batch_device(self._h, x0.shape[0], p(contact), p(ref_pos), p(x0), p(force),
                                                   p(force_all), p(status), ctypes.c_void_p(stream.cuda_stream)))

    def planOnce(self, contact_func, ref_pos_func, initial_param, current_time):
        """CCC::LinearMpcZ::planOnce (LinearMpcZ.h:140-143, src/LinearMpcZ.cpp:48-71)."""
        N = self.horizon_steps_
        ts = [current_time + i * self.horizon_dt_ for i in range(N)]
        contact = np.array([[bool(contact_func(t)) for t in ts]])
        ref = np.array([[float(ref_pos_func(t)) for t in ts]])
        return float(self.planOnceBatch(contact, ref, np.asarray(initial_param, float)[None])["force"][0])
