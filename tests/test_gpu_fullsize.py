"""GPU (MI355X), BASELINE.json configs[1..3] at full size: 1x256x160x160 map, 512 rotated
ROIs, pooled 8x64.  Compared with the oracle element for element (the OpenMP oracle takes a
second or two) and through size-independent properties."""
import numpy as np
import pytest
import torch

import workloads as Wk

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full(oracle):
    f, r = Wk.bench_inputs()
    want = oracle.forward_c(f, r, 8, 64, 0.25, threads=oracle.max_threads())
    return f, r, want


@pytest.fixture(scope="module")
def ext():
    from rroi_align._ext import rroi_align as e
    return e


def test_cfg2_forward_bit_exact(ext, full):
    f, r, want = full
    F, R = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
    for p in (ext.PATH_TILED, ext.PATH_DIRECT, ext.PATH_AUTO):
        got = ext.forward(F, R, 8, 64, 0.25, path=p).cpu().numpy()
        assert np.array_equal(got, want), f"path {p}: {(got != want).sum()} of {got.size} differ"


def test_cfg2_properties(ext, full):
    f, r, want = full
    F, R = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
    out = ext.forward(F, R, 8, 64, 0.25)
    # ROI independence / order equivariance: permuting the ROIs permutes the crops, bit for bit
    perm = torch.randperm(512, device="cuda", generator=torch.Generator("cuda").manual_seed(0))
    assert torch.equal(ext.forward(F, R[perm].contiguous(), 8, 64, 0.25), out[perm])
    # shard equivalence (BASELINE configs[3]): 8 shards of 64 == one call
    shards = [ext.forward(F, R[i * 64:(i + 1) * 64].contiguous(), 8, 64, 0.25) for i in range(8)]
    assert torch.equal(torch.cat(shards), out)
    # channel independence: a channel slice of the map gives the channel slice of the crops
    assert torch.equal(ext.forward(F[:, 64:128].contiguous(), R, 8, 64, 0.25), out[:, 64:128])
    # exact linearity under power-of-two scaling
    assert torch.equal(ext.forward(F * 4, R, 8, 64, 0.25), out * 4)
    # right-hand zero padding beyond roi_pooled_width = 8*w/h (kernel.cu:107)
    rpw = (8 * R[:, 4] / R[:, 3]).floor().long()
    for n in (0, 100, 511):
        assert not out[n, :, :, int(rpw[n]) + 1:].any()
    # idempotent / deterministic
    assert torch.equal(ext.forward(F, R, 8, 64, 0.25), out)
    # channels_last crops: same values in (R, PH, PW, C) storage
    ocl = ext.forward(F, R, 8, 64, 0.25, channels_last_out=True)
    assert ocl.is_contiguous(memory_format=torch.channels_last) and torch.equal(ocl, out)


def test_cfg3_forward_backward(ext, oracle, full):
    from rroi_align.modules.rroi_align import _RRoiAlign
    f, r, want = full
    feats = torch.from_numpy(f).cuda().requires_grad_(True)
    rois = torch.from_numpy(r).cuda()
    pooled = _RRoiAlign(8, 64, 0.25)(feats, rois)
    pooled.pow(2).sum().backward()
    assert np.array_equal(pooled.detach().cpu().numpy(), want)
    gwant = oracle.backward_c((2 * want).astype(np.float32), r, f.shape, 0.25)
    got = feats.grad.cpu().numpy()
    scale = float(np.abs(gwant).max())
    # BASELINE configs[2]: "parity <= 1e-4" -- max-abs, not relative
    Wk.check_backward(got, gwant, "cfg3 through autograd", require_abs=True)
    # backward is linear in grad_output
    g1 = ext.backward(pooled.detach() * 2, rois, f.shape, 0.25)
    g2 = ext.backward(pooled.detach() * 4, rois, f.shape, 0.25)
    assert torch.allclose(g2, 2 * g1, rtol=1e-4, atol=1e-4 * scale)
    # direct, both gathers (lists in HBM / lists built in the kernel) and the atomic scatter agree
    for p in (ext.PATH_DIRECT, ext.PATH_TILED_LISTS, ext.PATH_TILED_BUCKETS, ext.PATH_TILED_INKERNEL, ext.PATH_TILED_ATOMIC):
        gd = ext.backward(pooled.detach() * 2, rois, f.shape, 0.25, path=p)
        assert (gd - g1).abs().max().item() <= 1e-4 * scale


def test_cfg4_shard_of_4096(ext, oracle):
    """Rank r of 8 takes rows [512r, 512(r+1)) of a 4096-ROI set: spot-check two shards."""
    from rroi_align.sharded import shard_bounds
    f, r = Wk.bench_inputs(R=4096)
    F = torch.from_numpy(f).cuda()
    for rank in (0, 5):
        lo, hi = shard_bounds(4096, 8, rank)
        assert hi - lo == 512
        got = ext.forward(F, torch.from_numpy(r[lo:hi]).cuda(), 8, 64, 0.25).cpu().numpy()
        assert np.array_equal(got, oracle.forward_c(f, r[lo:hi], 8, 64, 0.25, threads=oracle.max_threads()))


def test_cfg4_single_call_4096_rois(ext):
    """All 4096 ROIs of configs[3] in ONE call (2 GiB of crops, R*C*PH*PW = 2^29): equals the eight
    512-ROI shards bit for bit; the backward (2.4 GB workspace, 3.5 M pairs) equals the sum of the
    shards' backward within the atomic-order tolerance and is linear."""
    from rroi_align.sharded import shard_bounds
    f, r = Wk.bench_inputs(R=4096)
    F, R = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
    out = ext.forward(F, R, 8, 64, 0.25)
    assert out.shape == (4096, 256, 8, 64)
    for rank in range(8):
        lo, hi = shard_bounds(4096, 8, rank)
        assert torch.equal(out[lo:hi], ext.forward(F, R[lo:hi].contiguous(), 8, 64, 0.25)), f"shard {rank}"
    g = torch.empty_like(out).normal_(generator=torch.Generator("cuda").manual_seed(1))
    gin = ext.backward(g, R, f.shape, 0.25)
    acc = torch.zeros_like(gin)
    for rank in range(8):
        lo, hi = shard_bounds(4096, 8, rank)
        acc += ext.backward(g[lo:hi].contiguous(), R[lo:hi].contiguous(), f.shape, 0.25)
    scale = float(acc.abs().max())
    assert float((gin - acc).abs().max()) <= 1e-4 * scale
    del out, acc
    gat = ext.backward(g, R, f.shape, 0.25, path=ext.PATH_TILED_ATOMIC)
    assert float((gin - gat).abs().max()) <= 1e-4 * scale
    del gat
    # 2.4 GB of relaid-out top_diff fit the in-kernel gather's 32-bit source offsets
    gik = ext.backward(g, R, f.shape, 0.25, path=ext.PATH_TILED_INKERNEL)
    assert float((gin - gik).abs().max()) <= 1e-4 * scale
