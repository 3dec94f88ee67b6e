"""rroi_align.sharded -- RoIRotate over the GPUs of one node.

New work (the reference is single-process, single-GPU).
ROIs are independent, so the rows of ``rois`` (and of the output) are split
across ranks with NO collective on the data path: every rank holds the feature
map it samples from (replicated, or simply its own images in data-parallel
training) and produces the crops of its own rows, which the recognition head
consumes rank-locally.  Two optional exchange steps exist for callers that
need them, both plain RCCL collectives over xGMI (``backend="nccl"`` on ROCm):

* ``gather_crops``   -- all_gather of the pooled crops (rows back in global order),
  differentiable: the backward is the adjoint of the gather;
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


def _all_gather_rows(local, num_rois, group, out=None):
    """Row shards under ``shard_bounds`` -> (num_rois, ...) on every rank.

    Equal shards (BASELINE configs[3]: 512 per GPU): ONE ``all_gather_into_tensor`` straight into
    the final buffer -- on the fully connected xGMI mesh that is a single direct exchange step, each
    shard crossing each link once, and nothing is copied afterwards.  Ragged shards are padded to
    the largest one, gathered the same way, and the rows compacted with one copy.
    """
    world, rank = _world(group)
    bounds = [shard_bounds(num_rois, world, r) for r in range(world)]
    sizes = [hi - lo for lo, hi in bounds]
    if local.shape[0] != sizes[rank]:
        raise ValueError("local shard has %d rows, shard_bounds gives %d" % (local.shape[0], sizes[rank]))
    tail = tuple(local.shape[1:])
    local = local.contiguous()
    if min(sizes) == max(sizes):
        if out is None:
            out = local.new_empty((num_rois,) + tail)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    mx = max(sizes)
    padded = local.new_zeros((mx,) + tail)
    padded[: sizes[rank]] = local
    buf = local.new_empty((world * mx,) + tail)
    dist.all_gather_into_tensor(buf, padded, group=group)
    if out is None:
        out = local.new_empty((num_rois,) + tail)
    for r, (lo, hi) in enumerate(bounds):
        out[lo:hi] = buf[r * mx: r * mx + sizes[r]]
    return out


class _GatherCrops(torch.autograd.Function):
    """all_gather whose backward is its adjoint.  ``reduce``: every rank holds a gradient for ALL
    rows; the gradient of this rank's shard is the SUM over ranks of their slice [lo, hi)
    (reduce-scatter) when the ranks computed different losses on the gathered crops, or just this
    rank's own slice when every rank computed the same, replicated loss (``reduce=False``)."""

    @staticmethod
    def forward(ctx, local, num_rois, group, reduce):
        ctx.num_rois, ctx.group, ctx.reduce = num_rois, group, reduce
        return _all_gather_rows(local, num_rois, group)

    @staticmethod
    def backward(ctx, grad_all):
        world, rank = _world(ctx.group)
        lo, hi = shard_bounds(ctx.num_rois, world, rank)
        if ctx.reduce and world > 1:
            sizes = [b - a for a, b in (shard_bounds(ctx.num_rois, world, r) for r in range(world))]
            backend = dist.get_backend(ctx.group)
            if min(sizes) == max(sizes) and backend != "gloo":  # gloo has no reduce_scatter
                mine = grad_all.new_empty((hi - lo,) + tuple(grad_all.shape[1:]))
                dist.reduce_scatter_tensor(mine, grad_all.contiguous(), op=dist.ReduceOp.SUM, group=ctx.group)
                return mine, None, None, None
            grad_all = grad_all.contiguous().clone()
            dist.all_reduce(grad_all, op=dist.ReduceOp.SUM, group=ctx.group)
        return grad_all[lo:hi].contiguous(), None, None, None


def gather_crops(local_crops, num_rois, group=None, reduce_grad=True, out=None):
    """all_gather of row shards produced under ``shard_bounds`` -> (num_rois, C, PH, PW) on every
    rank, rows in global order.  Differentiable (see ``_GatherCrops``); ``out`` (inference only) is a
    preallocated destination."""
    world, _ = _world(group)
    if world == 1:
        return local_crops
    if out is not None:
        if torch.is_grad_enabled() and local_crops.requires_grad:
            raise ValueError("gather_crops(out=...) is for inference: the result would be detached")
        return _all_gather_rows(local_crops, num_rois, group, out)
    return _GatherCrops.apply(local_crops, int(num_rois), group, bool(reduce_grad))


def allreduce_feature_grad(grad, group=None):
    """Sum the per-shard feature gradients (only when the feature map is replicated)."""
    world, _ = _world(group)
    if world > 1:
        dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=group)
    return grad


class ShardedRRoiAlign(Module):
    """``_RRoiAlign`` on this rank's rows of a globally known ``rois`` tensor.

    forward(features, rois_global) -> crops of rows ``shard_bounds(R, world, rank)``
    (or all R rows, in order, when ``gather=True``; gradients flow back through the gather).
    ``op`` is the per-rank operator; it defaults to the HIP ``_RRoiAlign``.
    """

    def __init__(self, pooled_height, pooled_width, spatial_scale, group=None, gather=False,
                 op=None, channels_last_out=False, reduce_grad=True):
        super(ShardedRRoiAlign, self).__init__()
        if op is None:
            from .modules.rroi_align import _RRoiAlign
            op = _RRoiAlign(pooled_height, pooled_width, spatial_scale, channels_last_out)
        self.op = op
        self.group = group
        self.gather = gather
        self.reduce_grad = reduce_grad

    def forward(self, features, rois):
        world, rank = _world(self.group)
        lo, hi = shard_bounds(rois.shape[0], world, rank)
        local = self.op(features, rois[lo:hi].contiguous())
        if self.gather:
            return gather_crops(local, rois.shape[0], self.group, self.reduce_grad)
        return local
