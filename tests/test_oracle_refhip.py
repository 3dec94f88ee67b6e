"""CPU: pin the oracle to outputs of the REFERENCE'S OWN KERNELS.

tests/golden/refhip_*.npz were produced on an MI355X by rroi_align_kernel.cu:28-312 itself --
run through ROCm's hipify-perl and hipcc (oracle/Makefile: ref), driven exactly as
functions/rroi_align.py:13-40 drives it (tests/golden/make_ref_golden.py).  The oracle must
reproduce: the pooled output and both con_idx tensors BIT FOR BIT, the feature gradient to fp32
summation-order noise."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["refhip_cfg1", "refhip_mid", "refhip_edge", "refhip_ph11"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_reference_kernels(oracle, name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    ph, pw = (int(v) for v in z["pooled"])
    s = float(z["scale"])
    out, geom = oracle.forward_c(z["features"], z["rois"], ph, pw, s, return_geom=True)
    assert np.array_equal(out, z["out"], equal_nan=True), f"{(out != z['out']).sum()} outputs differ"
    # kernel.cu:144-145 stores the same centre for every channel
    assert bool(z["idx_same_over_channels"])
    assert np.array_equal(geom[..., 0], z["idx_x"]) and np.array_equal(geom[..., 1], z["idx_y"])
    # the literal per-element restatement and the numpy restatement agree as well
    lit, ix, iy = oracle.forward_literal_c(z["features"], z["rois"], ph, pw, s)
    assert np.array_equal(lit, z["out"], equal_nan=True) and np.array_equal(ix[:, 0], z["idx_x"])
    npo, _, _ = oracle.forward_np(z["features"], z["rois"], ph, pw, s)
    assert np.array_equal(npo, z["out"], equal_nan=True)
    # backward: kernel.cu:193-278 on grad_output = 2 * output
    gin = oracle.backward_c((2 * np.nan_to_num(out)).astype(np.float32), z["rois"], z["features"].shape, s)
    scale = max(1.0, float(np.abs(z["grad_in"]).max()))
    assert np.abs(gin - z["grad_in"]).max() <= 1e-5 * scale
    assert np.array_equal(gin == 0, z["grad_in"] == 0), "support of the gradient differs"
