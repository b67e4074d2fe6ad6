"""Python mirror of the device-side closed loops of the force-scale planners (csrc/centroidal_loop.hip,
include/ccc_amd.h "The closed-loop tests of the force-scale planners on the device"): the control loops of
tests/src/TestDdpCentroidal.cpp:96-150, TestDdpSingleRigidBody.cpp:103-170 and TestLinearMpcXY.cpp:98-132 around
CentroidalSim (tests/src/SimModels.h:233-340) for n instances in one call, everything resident in HBM."""
import ctypes

import numpy as np

from . import _lib


class _Timeline(ctypes.Structure):
    _fields_ = [("K", ctypes.c_int), ("C", ctypes.c_int), ("seg_end", ctypes.c_void_p), ("seg_contact", ctypes.c_void_p),
                ("seg_ref", ctypes.c_void_p), ("contact_dim", ctypes.c_void_p), ("contact_vertex", ctypes.c_void_p),
                ("contact_ridge", ctypes.c_void_p), ("time_eps", ctypes.c_double)]


class ContactTimeline:
    """Per-instance piecewise-constant contact / reference schedule on the device.

    seg_end [n,K], seg_contact [n,K] i32, seg_ref [n,K,6], contact_dim [n,C] i32, contact_vertex / contact_ridge
    [n,C,M,3] with M = the planner's max_ridges (16, or 32: two surface contacts per entry) (numpy, host) -> CUDA tensors
    on `device`."""

    def __init__(self, seg_end, seg_contact, seg_ref, contact_dim, contact_vertex, contact_ridge, time_eps, device=0):
        import torch

        dev = torch.device("cuda", device)
        self.n, self.K = seg_end.shape
        self.C = contact_dim.shape[1]
        assert seg_contact.shape == (self.n, self.K) and seg_ref.shape == (self.n, self.K, 6)
        self.M = contact_vertex.shape[2]
        assert contact_vertex.shape == (self.n, self.C, self.M, 3) and contact_ridge.shape == (self.n, self.C, self.M, 3)
        f = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)  # noqa: E731
        self.t = dict(seg_end=f(seg_end, np.float64), seg_contact=f(seg_contact, np.int32), seg_ref=f(seg_ref, np.float64),
                      contact_dim=f(contact_dim, np.int32), contact_vertex=f(contact_vertex, np.float64),
                      contact_ridge=f(contact_ridge, np.float64))
        self.time_eps = float(time_eps)

    def c_struct(self):
        p = {k: v.data_ptr() for k, v in self.t.items()}
        return _Timeline(self.K, self.C, p["seg_end"], p["seg_contact"], p["seg_ref"], p["contact_dim"], p["contact_vertex"],
                         p["contact_ridge"], self.time_eps)


def _bind(L):
    if getattr(L, "_loop_bound", False):
        return
    vp = ctypes.c_void_p
    L.ccc_ddp_closed_loop_device.restype = ctypes.c_int
    L.ccc_ddp_closed_loop_device.argtypes = [vp, ctypes.c_int64, ctypes.POINTER(_Timeline), vp, vp, ctypes.c_double,
                                             ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), vp, vp,
                                             ctypes.POINTER(ctypes.c_double), vp]
    L.ccc_xy_closed_loop_device.restype = ctypes.c_int
    L.ccc_xy_closed_loop_device.argtypes = [vp, ctypes.c_int64, ctypes.POINTER(_Timeline), ctypes.c_double, vp, vp,
                                            ctypes.c_double, ctypes.c_double, ctypes.c_int, vp, vp,
                                            ctypes.POINTER(ctypes.c_double), vp]
    L._loop_bound = True


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def ddp_closed_loop(planner, timeline, inertia_diag, sim_state, t0, sim_dt, cycles, first_max_iter=500, warm_max_iter=1,
                    disturb_times=(), disturb_lin=(0.0, 0.0, 0.0), stats=None, log=None, stream=None):
    """ccc_ddp_closed_loop_device for a DdpCentroidal / DdpSingleRigidBody mirror object.  inertia_diag [n,3] and
    sim_state [n,18] are CUDA tensors (sim_state is advanced in place); returns the time after the last cycle."""
    import torch

    L = _lib.load()
    _bind(L)
    if stream is None:
        stream = torch.cuda.current_stream(planner.device)
    if timeline.M != planner.max_ridges_:
        raise ValueError("the timeline carries %d ridges per contact entry, the planner %d" % (timeline.M, planner.max_ridges_))
    tl = timeline.c_struct()
    # the loop plans with the object's solver configuration (first_max_iter / warm_max_iter override max_iter per cycle)
    _lib.check(L.ccc_ddp_set_config(planner._h, ctypes.byref(planner.ddp_solver_.config())))
    dts = (ctypes.c_double * max(1, len(disturb_times)))(*disturb_times)
    dl = (ctypes.c_double * 3)(*disturb_lin)
    t_end = ctypes.c_double(0.0)
    _lib.check(L.ccc_ddp_closed_loop_device(planner._h, timeline.n, ctypes.byref(tl), _ptr(inertia_diag), _ptr(sim_state),
                                            float(t0), float(sim_dt), int(cycles), int(first_max_iter), int(warm_max_iter),
                                            len(disturb_times), dts, dl, _ptr(stats), _ptr(log), ctypes.byref(t_end),
                                            ctypes.c_void_p(stream.cuda_stream)))
    return t_end.value


def xy_closed_loop(planner, timeline, com_z, inertia_diag, sim_state, t0, sim_dt, cycles, stats=None, log=None, stream=None):
    """ccc_xy_closed_loop_device for a LinearMpcXY mirror object; returns the time after the last cycle."""
    import torch

    L = _lib.load()
    _bind(L)
    if stream is None:
        stream = torch.cuda.current_stream(planner.device)
    if timeline.M != planner.max_ridges_:
        raise ValueError("the timeline carries %d ridges per contact entry, the planner %d" % (timeline.M, planner.max_ridges_))
    tl = timeline.c_struct()
    t_end = ctypes.c_double(0.0)
    _lib.check(L.ccc_xy_closed_loop_device(planner._h, timeline.n, ctypes.byref(tl), float(com_z), _ptr(inertia_diag),
                                           _ptr(sim_state), float(t0), float(sim_dt), int(cycles), _ptr(stats), _ptr(log),
                                           ctypes.byref(t_end), ctypes.c_void_p(stream.cuda_stream)))
    return t_end.value
