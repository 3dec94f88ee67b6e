"""rroi_align.sharded -- RoIRotate over the GPUs of one node.

New work (the reference is single-process, single-GPU).
ROIs are independent, so the rows of ``rois`` (and of the output) are split
across ranks with NO collective on the data path: every rank holds the feature
map it samples from (replicated, or simply its own images in data-parallel
training) and produces the crops of its own rows, which the recognition head
consumes rank-locally.  Two optional exchange steps exist for callers that
need them, both plain RCCL collectives over xGMI (``backend="nccl"`` on ROCm):

* ``gather_crops``   -- all_gather of the pooled crops (rows back in global order);
* ``allreduce_feature_grad`` -- sum of the partial feature gradients when the
  SAME feature map was replicated on all ranks.

One process per GPU; ``torch.distributed`` must be initialised by the caller.
"""
import torch
import torch.distributed as dist
from torch.nn.modules.module import Module


def shard_bounds(num_rois, world_size, rank):
    """Contiguous, balanced row range [lo, hi) of ``rank``: the first
    ``num_rois % world_size`` ranks get one extra row."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError("bad rank/world_size")
    base, rem = divmod(int(num_rois), world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def gather_crops(local_crops, num_rois, group=None):
    """all_gather of row shards produced under ``shard_bounds`` -> (num_rois, C, PH, PW)
    on every rank.  Shards are padded to the largest shard for the collective."""
    world, rank = _world(group)
    if world == 1:
        return local_crops
    sizes = [shard_bounds(num_rois, world, r) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = local_crops.new_zeros((mx,) + tuple(local_crops.shape[1:]))
    pad[: local_crops.shape[0]] = local_crops
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], 0)


def allreduce_feature_grad(grad, group=None):
    """Sum the per-shard feature gradients (only when the feature map is replicated)."""
    world, _ = _world(group)
    if world > 1:
        dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=group)
    return grad


class ShardedRRoiAlign(Module):
    """``_RRoiAlign`` on this rank's rows of a globally known ``rois`` tensor.

    forward(features, rois_global) -> crops of rows ``shard_bounds(R, world, rank)``
    (or all R rows, in order, when ``gather=True``).  ``op`` is the per-rank
    operator; it defaults to the HIP ``_RRoiAlign``.
    """

    def __init__(self, pooled_height, pooled_width, spatial_scale, group=None, gather=False,
                 op=None, channels_last_out=False):
        super(ShardedRRoiAlign, self).__init__()
        if op is None:
            from .modules.rroi_align import _RRoiAlign
            op = _RRoiAlign(pooled_height, pooled_width, spatial_scale, channels_last_out)
        self.op = op
        self.group = group
        self.gather = gather

    def forward(self, features, rois):
        world, rank = _world(self.group)
        lo, hi = shard_bounds(rois.shape[0], world, rank)
        local = self.op(features, rois[lo:hi].contiguous())
        if self.gather:
            return gather_crops(local, rois.shape[0], self.group)
        return local
