"""GPU (MI355X): the PRODUCT on the vectors the reference's own kernels produced.

tests/golden/refhip_*.npz hold inputs and outputs of rroi_align_kernel.cu:28-312 itself (built by
oracle/Makefile: ref with -ffp-contract=off, run on an MI355X by tests/golden/make_ref_golden.py).
tests/test_oracle_refhip.py pins the CPU oracle to them; this file runs the HIP library on the same
inputs -- every forward path bit for bit (the rounding-tie set `refhip_ties` included: the most
sensitive fixture in the repository), the reference-ABI entry point's con_idx_x / con_idx_y bit
for bit, every backward path to 1e-4 of the gradient scale.  No oracle, no oracle/_ref binary:
frozen data only, so the pin survives on a checkout that has neither."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["refhip_cfg1", "refhip_mid", "refhip_edge", "refhip_ph11", "refhip_ties"]


def _load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    ph, pw = (int(v) for v in z["pooled"])
    return z, ph, pw, float(z["scale"])


@pytest.mark.parametrize("name", CASES)
def test_forward_paths_reproduce_the_reference_vectors(name):
    from rroi_align._ext import rroi_align as ext
    z, ph, pw, s = _load(name)
    F, R = torch.from_numpy(z["features"]).cuda(), torch.from_numpy(z["rois"]).cuda()
    want = z["out"]
    for path in ext.FORWARD_PATHS:
        got = ext.forward(F, R, ph, pw, s, path=path).cpu().numpy()
        nd = int((got.view(np.uint32) != want.view(np.uint32)).sum())
        # NaN payloads aside (none in these sets), bit for bit
        assert np.array_equal(got, want, equal_nan=True) and nd == 0, f"{name} path {path}: {nd} elements differ"
    # channels-last features consumed in place, channels-last crops: same values element for element
    if z["features"].shape[1] % 4 == 0:
        got = ext.forward(F.contiguous(memory_format=torch.channels_last), R, ph, pw, s,
                          channels_last_out=True).cpu().numpy()
        assert np.array_equal(got, want, equal_nan=True)


@pytest.mark.parametrize("name", CASES)
def test_reference_abi_fills_con_idx_like_the_reference(name):
    from rroi_align._ext import rroi_align as ext
    z, ph, pw, s = _load(name)
    F, R = torch.from_numpy(z["features"]).cuda(), torch.from_numpy(z["rois"]).cuda()
    n, C = R.shape[0], F.shape[1]
    out, ix, iy = (torch.full((n, C, ph, pw), float("nan"), device="cuda") for _ in range(3))
    assert ext.rroi_align_forward_cuda(ph, pw, s, F, R, out, ix, iy) == 1
    assert np.array_equal(out.cpu().numpy(), z["out"], equal_nan=True)
    # kernel.cu:144-145 stores the same centre for every channel
    for c in range(C):
        assert np.array_equal(ix[:, c].cpu().numpy(), z["idx_x"]) and np.array_equal(iy[:, c].cpu().numpy(), z["idx_y"])
    # the test hook that exposes the bin centres directly
    geom = ext.bin_centres(R, ph, pw, s, F.shape[2], F.shape[3]).cpu().numpy()
    assert np.array_equal(geom[..., 0], z["idx_x"]) and np.array_equal(geom[..., 1], z["idx_y"])


@pytest.mark.parametrize("name", CASES)
def test_backward_paths_reproduce_the_reference_gradient(name):
    from rroi_align._ext import rroi_align as ext
    z, ph, pw, s = _load(name)
    R = torch.from_numpy(z["rois"]).cuda()
    # make_ref_golden.py: grad_output = 2 * output (kernel.cu:193-278 on it)
    gout = torch.from_numpy((2 * np.nan_to_num(z["out"])).astype(np.float32)).cuda()
    want = z["grad_in"]
    scale = max(1.0, float(np.abs(want).max()))
    for path in ext.BACKWARD_PATHS:
        got = ext.backward(gout, R, z["features"].shape, s, path=path).cpu().numpy()
        assert np.abs(got - want).max() <= 1e-4 * scale, (name, path)
        assert np.array_equal(got == 0, want == 0), f"{name} path {path}: support of the gradient differs"
    # the reference-ABI launcher, driven as functions/rroi_align.py:32-40 drives it
    n, C = R.shape[0], z["features"].shape[1]
    ix = torch.from_numpy(np.repeat(z["idx_x"][:, None], C, 1).copy()).cuda()
    iy = torch.from_numpy(np.repeat(z["idx_y"][:, None], C, 1).copy()).cuda()
    gin = torch.zeros(z["features"].shape, device="cuda")
    assert ext.rroi_align_backward_cuda(ph, pw, s, gout, R, gin, ix, iy) == 1
    assert np.abs(gin.cpu().numpy() - want).max() <= 1e-4 * scale
