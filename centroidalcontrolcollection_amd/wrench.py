"""ForceColl::calcTotalWrench on the device (include/ccc_amd.h, csrc/wrench.hip): the step after planOnce of the
force-scale planners (LinearMpcXY, DdpCentroidal, DdpSingleRigidBody) -- /root/reference/tests/src/TestLinearMpcXY.cpp:119-120."""
import ctypes

from . import _lib


def total_wrench_device(dim, vertex, ridge, scales, origin, wrench=None, stream=None):
    """Device-resident torch tensors: dim [n] i32, vertex / ridge [n,M,3], scales [n,K] (first dim entries of a row used;
    a strided view of a larger tensor is fine as long as the last dimension is contiguous), origin [n,3].
    Returns wrench [n,6] = [moment; force] (asynchronous on `stream`)."""
    import torch

    L = _lib.load()
    if not getattr(L, "_wrench_bound", False):
        vp = ctypes.c_void_p
        L.ccc_total_wrench_device.restype = ctypes.c_int
        L.ccc_total_wrench_device.argtypes = [ctypes.c_int64, ctypes.c_int, vp, vp, vp, vp, ctypes.c_int, vp, vp, vp]
        L._wrench_bound = True
    n, M = vertex.shape[0], vertex.shape[1]
    if scales.stride(-1) != 1 or vertex.shape != ridge.shape or not vertex.is_contiguous() or not ridge.is_contiguous():
        raise ValueError("vertex / ridge must be contiguous [n,M,3] and the rows of scales contiguous")
    if wrench is None:
        wrench = torch.empty((n, 6), dtype=torch.float64, device=vertex.device)
    if stream is None:
        stream = torch.cuda.current_stream(vertex.device)
    _lib.check(L.ccc_total_wrench_device(n, M, ctypes.c_void_p(dim.data_ptr()), ctypes.c_void_p(vertex.data_ptr()),
                                         ctypes.c_void_p(ridge.data_ptr()), ctypes.c_void_p(scales.data_ptr()),
                                         int(scales.stride(0)) if n > 1 else int(scales.shape[-1]),
                                         ctypes.c_void_p(origin.data_ptr()), ctypes.c_void_p(wrench.data_ptr()),
                                         ctypes.c_void_p(stream.cuda_stream)))
    return wrench
