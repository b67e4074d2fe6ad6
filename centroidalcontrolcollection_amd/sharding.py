"""Batch sharding of the planOnce() path over the GPUs of one node.

The instances are independent, so the only parallel axis is the batch: rank r solves the contiguous range
[start_r, end_r) on its own GPU (one process per GPU), with the batch-constant model replicated.  The one
exchange step the north star asks for is an all-gather of the planned outputs (RCCL over xGMI through
torch.distributed's "nccl" backend on GPUs; "gloo" on CPU in the tests).  No other collective exists on the path.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world_size):
    """Contiguous, balanced ranges: the first (n % world_size) ranks get one extra instance."""
    base, extra = divmod(int(n), int(world_size))
    bounds, start = [], 0
    for r in range(world_size):
        end = start + base + (1 if r < extra else 0)
        bounds.append((start, end))
        start = end
    return bounds


def all_gather_outputs(local, n_total, group=None):
    """Gather per-rank output rows ([n_local, ...]) into the full [n_total, ...] tensor on every rank.

    Equal shards use one all_gather_into_tensor (a single RCCL all-gather); ragged shards are padded to the
    largest shard first.  Returns `local` unchanged when torch.distributed is not initialised."""
    if not dist.is_available() or not dist.is_initialized():
        return local
    world = dist.get_world_size(group)
    bounds = shard_bounds(n_total, world)
    sizes = [e - s for s, e in bounds]
    if local.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError("rank %d holds %d rows, expected %d" % (dist.get_rank(group), local.shape[0],
                                                                  sizes[dist.get_rank(group)]))
    tail = tuple(local.shape[1:])
    if min(sizes) == max(sizes):
        out = torch.empty((n_total,) + tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    m = max(sizes)
    padded = torch.zeros((m,) + tail, dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    buf = torch.empty((world * m,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    return torch.cat([buf[r * m:r * m + sizes[r]] for r in range(world)], dim=0)


def plan_sharded(solve_local, x0, zlim, group=None):
    """Solve a full batch held on every rank: each rank solves its shard with `solve_local(x0_shard, zlim_shard)
    -> zmp_shard` (a torch tensor) and the shards are all-gathered.  Returns zmp [n, 2] on every rank."""
    n = x0.shape[0]
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    s, e = shard_bounds(n, world)[rank]
    local = solve_local(x0[s:e], zlim[s:e])
    return all_gather_outputs(local, n, group)


def plan_sharded_problem(solve_local, prob, x0, group=None):
    """The same for the classes whose problem is a dict of per-instance arrays (LinearMpcXY: dim / vertex / ridge / ...; the
    DDP planners: phase tables, references, ...): every tensor of `prob` and `x0` is cut along its first dimension into
    this rank's contiguous shard, `solve_local(prob_shard, x0_shard) -> out_shard [n_local, ...]` plans it, and the planned
    outputs (the first-step force scales) are all-gathered: [n, ...] on every rank (configs 4 and 5)."""
    n = x0.shape[0]
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    s, e = shard_bounds(n, world)[rank]
    local = solve_local({k: v[s:e] for k, v in prob.items()}, x0[s:e])
    return all_gather_outputs(local, n, group)


class ShardedLinearMpcZmp:
    """The C-ABI's own multi-GPU path (include/ccc_amd.h "One node, several GPUs", csrc/sharded.hip) for a host that is
    ONE process -- what a C++ controller linking libccc_amd.so gets: contiguous shards over a device list, one
    ccc_zmp_t per device, host arrays in / out (`planOnceBatch`) or device-resident shards with an RCCL all-gather of the
    planned ZMPs onto every device (`plan_batch_device`).  (torch.distributed, above, is the one-process-per-GPU path the
    benchmark driver uses.)"""

    def __init__(self, com_height, horizon_duration, horizon_dt, devices):
        import ctypes

        from . import _lib

        self._ct, self._lib = ctypes, _lib
        L = _lib.load()
        vp, dp = ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)
        L.ccc_zmp_sharded_create.restype = ctypes.c_int
        L.ccc_zmp_sharded_create.argtypes = [ctypes.c_double] * 3 + [ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                                                                     ctypes.POINTER(vp)]
        L.ccc_zmp_sharded_destroy.restype = None
        L.ccc_zmp_sharded_destroy.argtypes = [vp]
        L.ccc_zmp_sharded_plan_batch.restype = ctypes.c_int
        L.ccc_zmp_sharded_plan_batch.argtypes = [vp, ctypes.c_int64, dp, dp, ctypes.c_double, dp,
                                                 ctypes.POINTER(ctypes.c_int32)]
        L.ccc_zmp_sharded_plan_batch_device.restype = ctypes.c_int
        L.ccc_zmp_sharded_plan_batch_device.argtypes = [vp, ctypes.c_int64, ctypes.POINTER(vp), ctypes.POINTER(vp),
                                                        ctypes.c_double, ctypes.POINTER(vp), ctypes.POINTER(vp)]
        self._L = L
        self.devices = [int(d) for d in devices]
        arr = (ctypes.c_int * len(self.devices))(*self.devices)
        h = vp()
        _lib.check(L.ccc_zmp_sharded_create(float(com_height), float(horizon_duration), float(horizon_dt), arr,
                                            len(self.devices), ctypes.byref(h)))
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.ccc_zmp_sharded_destroy(h)
            self._h = None

    def planOnceBatch(self, x0, zlim, control_dt=-1.0):
        """Host arrays x0 [n,2,3], zlim [n,2,2,N] -> dict(zmp [n,2], status [n,2], pivots [n,2])."""
        import numpy as np

        ct = self._ct
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        zlim = np.ascontiguousarray(zlim, dtype=np.float64)
        n = x0.shape[0]
        zmp = np.empty((n, 2))
        status = np.empty((n, 2), dtype=np.int32)
        dp = ct.POINTER(ct.c_double)
        self._lib.check(self._L.ccc_zmp_sharded_plan_batch(self._h, n, x0.ctypes.data_as(dp), zlim.ctypes.data_as(dp),
                                                            float(control_dt), zmp.ctypes.data_as(dp),
                                                            status.ctypes.data_as(ct.POINTER(ct.c_int32))))
        return dict(zmp=zmp, status=status & 0xff, pivots=status >> 8)

    def plan_batch_device(self, x0, zlim, control_dt, zmp_all, status=None):
        """Lists (one CUDA tensor per device of the handle): x0[r] [m,2,3], zlim[r] [m,2,2,N] on devices[r]; zmp_all[r]
        [D*m,2] on devices[r] receives the planned ZMPs of ALL shards (RCCL all-gather).  Synchronous."""
        ct = self._ct
        D = len(self.devices)
        m = x0[0].shape[0]

        def arr(ts):
            return (ct.c_void_p * D)(*[ct.c_void_p(t.data_ptr()) for t in ts])

        # the handle's streams wait for what torch has enqueued on each device's current stream (inputs produced by
        # asynchronous kernels, output buffers still being filled)
        self._L.ccc_zmp_sharded_plan_batch_device_ordered.restype = ct.c_int
        self._L.ccc_zmp_sharded_plan_batch_device_ordered.argtypes = [
            ct.c_void_p, ct.c_int64, ct.POINTER(ct.c_void_p), ct.POINTER(ct.c_void_p), ct.c_double,
            ct.POINTER(ct.c_void_p), ct.POINTER(ct.c_void_p), ct.POINTER(ct.c_void_p)]
        self._lib.check(self._L.ccc_zmp_sharded_plan_batch_device_ordered(
            self._h, m, arr(x0), arr(zlim), float(control_dt), arr(zmp_all), arr(status) if status else None,
            _current_streams(self.devices)))


def _current_streams(devices):
    """torch's current stream of every listed device, as the void * list the sharded entry points order themselves after."""
    import ctypes

    import torch

    return (ctypes.c_void_p * len(devices))(*[ctypes.c_void_p(torch.cuda.current_stream(d).cuda_stream) for d in devices])


def _ptr_list(ts, D):
    import ctypes

    if ts is None:
        return None
    return (ctypes.c_void_p * D)(*[None if t is None else ctypes.c_void_p(t.data_ptr()) for t in ts])


class ShardedLinearMpcXY:
    """ccc_xy_sharded_* (csrc/sharded.hip): LinearMpcXY over a device list in ONE process -- BASELINE config 4's "sharded
    8xMI355X over xGMI" for a C++ host.  Built from a LinearMpcXY mirror object (its parameters are replicated on every
    listed device)."""

    def __init__(self, planner, devices):
        import ctypes

        from . import _lib
        from .linear_mpc_xy import _Params

        self._lib, L = _lib, _lib.load()
        vp = ctypes.c_void_p
        L.ccc_xy_get_params.restype = ctypes.c_int
        L.ccc_xy_get_params.argtypes = [vp, ctypes.POINTER(_Params), ctypes.POINTER(ctypes.c_int)]
        L.ccc_xy_sharded_create.restype = ctypes.c_int
        L.ccc_xy_sharded_create.argtypes = [ctypes.POINTER(_Params), ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                                            ctypes.POINTER(vp)]
        L.ccc_xy_sharded_destroy.restype = None
        L.ccc_xy_sharded_destroy.argtypes = [vp]
        L.ccc_xy_sharded_plan_batch_device.restype = ctypes.c_int
        L.ccc_xy_sharded_plan_batch_device.argtypes = [vp, ctypes.c_int64] + [ctypes.POINTER(vp)] * 10
        self._L = L
        prm = _Params()
        _lib.check(L.ccc_xy_get_params(planner._h, ctypes.byref(prm), None))
        self.max_ridges_ = prm.max_ridges or 16
        self.devices = [int(d) for d in devices]
        arr = (ctypes.c_int * len(self.devices))(*self.devices)
        h = vp()
        _lib.check(L.ccc_xy_sharded_create(ctypes.byref(prm), arr, len(self.devices), ctypes.byref(h)))
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.ccc_xy_sharded_destroy(h)
            self._h = None

    def plan_batch_device(self, probs, x0, u0_all, status=None):
        """probs: one dict of CUDA tensors per device (keys of LinearMpcXY.plan_batch_device), x0[r] [m,6], u0_all[r]
        [D*m, M] on devices[r]: receives the planned first-step force scales of ALL shards.  Synchronous."""
        D = len(self.devices)
        m = x0[0].shape[0]
        cols = [[p[k] for p in probs] for k in ("dim", "vertex", "ridge", "com_z", "total_force_z", "ref_out")]
        self._lib.check(self._L.ccc_xy_sharded_plan_batch_device(
            self._h, m, *[_ptr_list(c, D) for c in cols], _ptr_list(x0, D), _ptr_list(u0_all, D), _ptr_list(status, D),
            _current_streams(self.devices)))


class ShardedDdp:
    """ccc_ddp_sharded_* (csrc/sharded.hip): DdpCentroidal / DdpSingleRigidBody over a device list in ONE process --
    BASELINE config 5's "8xMI355X" for a C++ host.  Built from a mirror object (parameters and the current solver
    configuration are replicated on every listed device)."""

    def __init__(self, planner, devices):
        import ctypes

        from . import _lib
        from .ddp import Config, _Params as Params

        self._lib, L = _lib, _lib.load()
        vp = ctypes.c_void_p
        L.ccc_ddp_get_params.restype = ctypes.c_int
        L.ccc_ddp_get_params.argtypes = [vp, ctypes.POINTER(Params)]
        L.ccc_ddp_sharded_create.restype = ctypes.c_int
        L.ccc_ddp_sharded_create.argtypes = [ctypes.POINTER(Params), ctypes.POINTER(Config), ctypes.POINTER(ctypes.c_int),
                                             ctypes.c_int, ctypes.POINTER(vp)]
        L.ccc_ddp_sharded_destroy.restype = None
        L.ccc_ddp_sharded_destroy.argtypes = [vp]
        L.ccc_ddp_sharded_plan_batch_device.restype = ctypes.c_int
        L.ccc_ddp_sharded_plan_batch_device.argtypes = [vp, ctypes.c_int64] + [ctypes.POINTER(vp)] * 15
        self._L = L
        prm = Params()
        _lib.check(L.ccc_ddp_get_params(planner._h, ctypes.byref(prm)))
        self.S, self.N, self.M = planner.S, prm.horizon_steps, prm.max_ridges
        self.devices = [int(d) for d in devices]
        arr = (ctypes.c_int * len(self.devices))(*self.devices)
        h = vp()
        _lib.check(L.ccc_ddp_sharded_create(ctypes.byref(prm), ctypes.byref(planner.ddp_solver_.config()), arr,
                                            len(self.devices), ctypes.byref(h)))
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.ccc_ddp_sharded_destroy(h)
            self._h = None

    def plan_batch_device(self, probs, x0, u_out, u0_all, u_init=None, iters=None, status=None, cost=None):
        """probs: one dict of CUDA tensors per device (keys of the planners' plan_batch_device); x0[r] [m,S]; u_out[r]
        [m,N,M] (the shard's whole planned sequence, stays on its device); u0_all[r] [D*m, M]: the planned first-step
        force scales of ALL shards.  Synchronous."""
        D = len(self.devices)
        m = x0[0].shape[0]

        def col(k):
            return None if k not in probs[0] or probs[0][k] is None else _ptr_list([p[k] for p in probs], D)

        self._lib.check(self._L.ccc_ddp_sharded_plan_batch_device(
            self._h, m, col("phase_dim"), col("phase_vertex"), col("phase_ridge"), col("step_phase"), col("ref_pos"),
            col("ref_ori"), col("inertia"), _ptr_list(x0, D), _ptr_list(u_init, D), _ptr_list(u_out, D),
            _ptr_list(u0_all, D), _ptr_list(iters, D), _ptr_list(status, D), _ptr_list(cost, D),
            _current_streams(self.devices)))
