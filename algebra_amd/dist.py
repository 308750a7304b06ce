"""Multi-GPU MSM: one process per GPU, base-range sharding, RCCL all-gather of the per-rank partials.

The reference itself splits an MSM by base range and sums the chunk results
(ec/src/scalar_mul/variable_base/mod.rs:521-557); across GPUs the same split needs exactly one exchange:
every rank contributes one Projective point (3 field elements: 144 B for BLS12-381 G1).  The reduction
operator is elliptic-curve addition, which RCCL does not offer as a reduce op, so the collective is an
all-gather (latency-bound, a few microseconds over xGMI) followed by world_size-1 point additions on
the host.  No other data-path collective exists: bases and scalars never leave their GPU.
"""
import numpy as np

from . import curves as cv
from .msm import msm_bigint, msm_unchecked, sum_projective


def shard_bounds(n, rank, world):
    """[lo, hi) of rank's contiguous base range; the first n % world ranks take one extra element."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def combine_partials(curve, partial, group=None):
    """All-gather every rank's partial Projective and return their sum (identical on all ranks)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return np.ascontiguousarray(partial, dtype=np.uint64)
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    mine = torch.from_numpy(np.ascontiguousarray(partial, dtype=np.uint64).view(np.int64)).to(dev)
    bufs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(bufs, mine, group=group)
    allp = torch.stack(bufs).cpu().numpy().view(np.uint64)
    return sum_projective(curve, allp)


def msm_bigint_sharded(curve, bases_shard, bigints_shard, group=None, local_msm=msm_bigint):
    """MSM over the union of all ranks' shards; each rank passes only its own (base, scalar) range."""
    return combine_partials(curve, local_msm(curve, bases_shard, bigints_shard), group)


def msm_sharded(curve, bases_shard, scalars_shard, group=None):
    """Same for Fr (Montgomery) scalars: the sharded form of VariableBaseMSM::msm_unchecked."""
    return combine_partials(curve, msm_unchecked(curve, bases_shard, scalars_shard), group)
