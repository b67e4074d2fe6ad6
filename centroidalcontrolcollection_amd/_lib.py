"""ctypes binding of the C-ABI in include/ccc_amd.h (libccc_amd.so).

There is NO CPU fallback: if the HIP library is missing or no gfx950 device is visible, every entry
point raises.  PyTorch is used by callers only as plumbing (device buffers, streams, torch.distributed).
"""
import ctypes
import os

from . import build as _build

_lib = None

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int32_p = ctypes.POINTER(ctypes.c_int32)

# return codes / status (include/ccc_amd.h)
CCC_OK = 0
CCC_ERR_INVALID_ARGUMENT = 1
CCC_ERR_UNSUPPORTED = 2
CCC_ERR_HIP = 3
CCC_ERR_NO_DEVICE = 4
CCC_STATUS_SOLVED = 0
CCC_STATUS_INFEASIBLE = 1
CCC_STATUS_MAX_ITER = 2

ABI_VERSION = 5  # CCC_ABI_VERSION of include/ccc_amd.h

# every symbol include/ccc_amd.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "ccc_shard_bounds",
    "ccc_ddp_get_params",
    "ccc_ddp_get_config",
    "ccc_ddp_get_device",
    "ccc_ddp_arithmetic",
    "ccc_ddp_set_limits",
    "ccc_ddp_last_call_aborted",
    "ccc_ddp_set_inertia_per_phase",
    "ccc_ddp_effective_precision",
    "ccc_xy_get_params",
    "ccc_ddp_closed_loop_device",
    "ccc_xy_closed_loop_device",
    "ccc_device_count",
    "ccc_zmp_sharded_create",
    "ccc_zmp_sharded_destroy",
    "ccc_zmp_sharded_num_devices",
    "ccc_zmp_sharded_plan_batch",
    "ccc_zmp_sharded_plan_batch_device",
    "ccc_zmp_sharded_plan_batch_device_ordered",
    "ccc_xy_sharded_create",
    "ccc_xy_sharded_destroy",
    "ccc_xy_sharded_num_devices",
    "ccc_xy_sharded_plan_batch_device",
    "ccc_ddp_sharded_create",
    "ccc_ddp_sharded_destroy",
    "ccc_ddp_sharded_num_devices",
    "ccc_ddp_sharded_set_config",
    "ccc_ddp_sharded_set_limits",
    "ccc_ddp_sharded_plan_batch_device",
    "ccc_last_error_string",
    "ccc_abi_version",
    "ccc_zmp_create",
    "ccc_zmp_destroy",
    "ccc_zmp_horizon_steps",
    "ccc_zmp_get_seq",
    "ccc_zmp_last_kernel",
    "ccc_zmp_last_schedule",
    "ccc_zmp_plan_batch_device",
    "ccc_zmp_plan_batch",
    "ccc_zmp_get_model",
    "ccc_zmp_sample_limits_device",
    "ccc_zmp_closed_loop_device",
    "ccc_ddp_default_config",
    "ccc_ddp_create",
    "ccc_ddp_destroy",
    "ccc_ddp_set_config",
    "ccc_ddp_state_dim",
    "ccc_ddp_plan_batch_device",
    "ccc_ddp_plan_batch",
    "ccc_xy_create",
    "ccc_xy_destroy",
    "ccc_xy_plan_batch_device",
    "ccc_xy_plan_batch",
    "ccc_ism_create",
    "ccc_ism_destroy",
    "ccc_ism_horizon_steps",
    "ccc_ism_plan_batch_device",
    "ccc_ism_plan_batch",
    "ccc_z_create",
    "ccc_z_destroy",
    "ccc_z_plan_batch_device",
    "ccc_z_plan_batch",
    "ccc_ddpzmp_default_config",
    "ccc_ddpzmp_create",
    "ccc_ddpzmp_destroy",
    "ccc_ddpzmp_set_config",
    "ccc_ddpzmp_workspace_bytes",
    "ccc_ddpzmp_plan_batch_device",
    "ccc_ddpzmp_plan_batch",
    "ccc_ddpzmp_closed_loop_device",
    "ccc_total_wrench_device",
]


class CccError(RuntimeError):
    """Raised when a C-ABI call returns a non-zero code (the reference throws std::runtime_error)."""

    def __init__(self, code, message):
        super().__init__("libccc_amd error %d: %s" % (code, message))
        self.code = code


def lib_path():
    # CCC_AMD_LIB selects another build of the SAME library (e.g. an instrumented one while profiling)
    return os.environ.get("CCC_AMD_LIB", _build.LIB_PATH)


def load():
    """Load libccc_amd.so (built in-tree by build.build_lib); raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            "libccc_amd.so not found at %s: build it with `python -m centroidalcontrolcollection_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    if "CCC_AMD_LIB" not in os.environ and _build.is_stale():
        # a library built from other sources than the tree's would be tested silently: rebuild it (hipcc cross-compiles
        # anywhere this image runs) or fail loudly
        try:
            _build.build_lib(force=True)
        except Exception as e:
            raise ImportError("libccc_amd.so at %s was not built from the sources in this tree and rebuilding it "
                              "failed (%s); run `python -m centroidalcontrolcollection_amd.build`" % (path, e))
    try:
        # torch bundles its own libamdhip64.so.7; importing it first makes both share one HIP runtime
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing, the C-ABI does not need it
        pass
    L = ctypes.CDLL(path)
    L.ccc_last_error_string.restype = ctypes.c_char_p
    L.ccc_last_error_string.argtypes = []
    L.ccc_abi_version.restype = ctypes.c_int
    L.ccc_abi_version.argtypes = []
    if L.ccc_abi_version() != ABI_VERSION:
        # (the ctypes structures below mirror include/ccc_amd.h at this version: another library would read or write past
        #  their ends)
        raise ImportError("libccc_amd.so at %s answers ABI version %d, this package binds version %d"
                          % (path, L.ccc_abi_version(), ABI_VERSION))
    L.ccc_zmp_create.restype = ctypes.c_int
    L.ccc_zmp_create.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int,
                                 ctypes.POINTER(ctypes.c_void_p)]
    L.ccc_zmp_destroy.restype = None
    L.ccc_zmp_destroy.argtypes = [ctypes.c_void_p]
    L.ccc_zmp_horizon_steps.restype = ctypes.c_int
    L.ccc_zmp_horizon_steps.argtypes = [ctypes.c_void_p]
    L.ccc_zmp_last_kernel.restype = ctypes.c_char_p
    L.ccc_zmp_last_kernel.argtypes = [ctypes.c_void_p]
    L.ccc_zmp_last_schedule.restype = ctypes.c_char_p
    L.ccc_zmp_last_schedule.argtypes = [ctypes.c_void_p]
    L.ccc_zmp_get_seq.restype = ctypes.c_int
    L.ccc_zmp_get_seq.argtypes = [ctypes.c_void_p, c_double_p, c_double_p]
    L.ccc_zmp_plan_batch_device.restype = ctypes.c_int
    L.ccc_zmp_plan_batch_device.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p]
    L.ccc_zmp_plan_batch.restype = ctypes.c_int
    L.ccc_zmp_plan_batch.argtypes = [ctypes.c_void_p, ctypes.c_int64, c_double_p, c_double_p, ctypes.c_double,
                                     c_double_p, c_double_p, c_int32_p]
    _lib = L
    return L


def check(code):
    if code != CCC_OK:
        msg = load().ccc_last_error_string()
        raise CccError(code, msg.decode() if msg else "")
