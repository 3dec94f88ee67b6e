"""CPU, world_size 2 over gloo: the ROI sharding and the optional exchange steps of
rroi_align.sharded.  The per-rank operator is injected (the HIP op needs a GPU;
here the oracle stands in, as the checker of the host logic only)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import workloads as Wk


def test_shard_bounds_partition():
    from rroi_align.sharded import shard_bounds
    for R in (0, 1, 7, 512, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(R, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == R
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(4096, 8, 3) == (1536, 2048)  # BASELINE cfg4: 512 per GPU
    with pytest.raises(ValueError):
        shard_bounds(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, R, gather, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "fots.pytorch_amd"), os.path.join(root, "tests")]
    from oracle import rroi_align_oracle as O
    from rroi_align.sharded import ShardedRRoiAlign, allreduce_feature_grad, shard_bounds
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        f, r = Wk.bench_inputs(R=R, C=3, H=48, W=48, img=192, seed=9)

        def op(features, rois):
            return torch.from_numpy(O.forward_c(features.numpy(), rois.numpy(), 8, 16, 0.25))

        m = ShardedRRoiAlign(8, 16, 0.25, gather=gather, op=op)
        out = m(torch.from_numpy(f), torch.from_numpy(r))
        lo, hi = shard_bounds(R, world, rank)
        if gather:  # the inference form: one collective straight into a preallocated buffer
            from rroi_align.sharded import gather_crops
            buf = torch.full_like(out, 7.0)
            with torch.no_grad():
                res = gather_crops(op(torch.from_numpy(f), torch.from_numpy(r[lo:hi])), R, out=buf)
            assert res is buf and torch.equal(buf, out)
        gout = torch.from_numpy(O.forward_c(f, r[lo:hi], 8, 16, 0.25)) * 2
        g = torch.from_numpy(O.backward_c(gout.numpy(), r[lo:hi], f.shape, 0.25))
        g = allreduce_feature_grad(g)
        q.put((rank, out.numpy(), g.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("R,gather", [(10, True), (7, True), (8, False)])
def test_world2_gloo_matches_single_process(oracle, R, gather):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, R, gather, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(world):
        rank, out, g = q.get(timeout=120)
        got[rank] = (out, g)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from rroi_align.sharded import shard_bounds
    f, r = Wk.bench_inputs(R=R, C=3, H=48, W=48, img=192, seed=9)
    full = oracle.forward_c(f, r, 8, 16, 0.25)
    gfull = oracle.backward_c((2 * full).astype(np.float32), r, f.shape, 0.25)
    for rank in range(world):
        out, g = got[rank]
        if gather:  # every rank holds all rows, bit-identical to the single-process result
            assert np.array_equal(out, full)
        else:
            lo, hi = shard_bounds(R, world, rank)
            assert np.array_equal(out, full[lo:hi])
        assert np.allclose(g, gfull, rtol=1e-5, atol=1e-5)


def _grad_worker(rank, world, port, R, reduce_grad, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "fots.pytorch_amd"), os.path.join(root, "tests")]
    from rroi_align.sharded import ShardedRRoiAlign
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        f = torch.arange(24, dtype=torch.float32).view(1, 2, 3, 4).requires_grad_(True)
        rois = torch.arange(R * 6, dtype=torch.float32).view(R, 6)

        def op(features, r):  # any differentiable per-row operator
            return (r[:, 1].view(-1, 1, 1, 1) * features.mean()).expand(-1, 3, 2, 2).contiguous()

        m = ShardedRRoiAlign(2, 2, 1.0, gather=True, op=op, reduce_grad=reduce_grad)
        crops = m(f, rois)
        assert crops.requires_grad and crops.shape == (R, 3, 2, 2)
        ((rank + 1.0) * crops).sum().backward()   # every rank a different loss on ALL crops
        q.put((rank, crops.detach().numpy(), f.grad.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("R,reduce_grad", [(6, True), (5, True), (6, False)])
def test_gather_is_differentiable(R, reduce_grad):
    """the gathered crops used to be detached.  The backward of the gather is its
    adjoint: rank q's shard receives the sum over ranks of their gradient slice (reduce_grad) or
    its own slice (replicated loss)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, R, reduce_grad, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(world):
        rank, crops, g = q.get(timeout=120)
        got[rank] = (crops, g)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from rroi_align.sharded import shard_bounds
    rois = np.arange(R * 6, dtype=np.float32).reshape(R, 6)
    mean = np.arange(24, dtype=np.float32).mean()
    full = np.broadcast_to((rois[:, 1] * mean).reshape(-1, 1, 1, 1), (R, 3, 2, 2))
    for rank in range(world):
        crops, g = got[rank]
        assert np.array_equal(crops, full)
        lo, hi = shard_bounds(R, world, rank)
        weight = 3.0 if reduce_grad else rank + 1.0          # sum over ranks of (r + 1), or the own one
        want = weight * rois[lo:hi, 1].sum() * 12 / 24.0    # 12 crop elements per row, mean over 24
        assert np.allclose(g, want, rtol=1e-6), (rank, g.ravel()[0], want)
