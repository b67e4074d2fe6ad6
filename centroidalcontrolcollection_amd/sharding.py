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
