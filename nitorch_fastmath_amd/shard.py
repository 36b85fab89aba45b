"""Multi-GPU use of the hot path: the batch dimension shards embarrassingly.

Every matrix / solve / reduction chunk is independent, so the ONLY strategy is a
contiguous split of the flattened batch over the ranks of one node (one process per
GPU, `torch.distributed`; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for
tests).  No collective sits on the data path.  Two optional epilogues exist:
  * `gather_outputs`  -- all-gather of the per-rank outputs (off by default: on xGMI a
                         gather of C5's 2.4 GB/GPU costs ~16 ms against ~3 ms of compute);
  * `combine_scalar`  -- one 8-byte all-reduce for sharded full reductions
                         (nansum -> SUM, nanmax -> MAX, nanmin -> MIN).
"""
import torch
import torch.distributed as dist

__all__ = ['shard_bounds', 'shard_of', 'gather_outputs', 'combine_scalar', 'max_over_ranks']


def shard_bounds(n, rank, world):
    """[lo, hi) of rank's contiguous chunk of n items; sizes differ by at most one and
    the chunks tile [0, n) in rank order."""
    if not (0 <= rank < world):
        raise ValueError(f'rank {rank} outside world of size {world}')
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_of(t, rank, world, dim=0):
    """This rank's contiguous slice of `t` along the (flattened-batch) dimension `dim`."""
    lo, hi = shard_bounds(t.shape[dim], rank, world)
    return t.narrow(dim, lo, hi - lo)


def gather_outputs(local, n_total, group=None):
    """All-gather per-rank outputs (possibly of unequal length along dim 0) into the
    full `(n_total, ...)` tensor on every rank."""
    world = dist.get_world_size(group)
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    maxlen = max(hi - lo for lo, hi in sizes)
    pad = local.new_zeros((maxlen,) + tuple(local.shape[1:]))
    pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:hi - lo] for b, (lo, hi) in zip(bufs, sizes)], 0)


def combine_scalar(value, op, group=None):
    """Combine per-shard results of a full reduction: op in {'nansum','sum','nanmax','max','nanmin','min'}."""
    red = {'nansum': dist.ReduceOp.SUM, 'sum': dist.ReduceOp.SUM,
           'nanmax': dist.ReduceOp.MAX, 'max': dist.ReduceOp.MAX,
           'nanmin': dist.ReduceOp.MIN, 'min': dist.ReduceOp.MIN}[op]
    v = value.detach().clone().reshape(1)
    dist.all_reduce(v, op=red, group=group)
    return v.reshape(())


def max_over_ranks(seconds, device=None, group=None):
    """Wall time of the slowest rank (what a weak-scaling throughput must be quoted on)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    v = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(v, op=dist.ReduceOp.MAX, group=group)
    return float(v.item())
