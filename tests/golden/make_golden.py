"""Regenerate the oracle regression vectors (run from the repo root):

    python tests/golden/make_golden.py

Inputs are seeded; outputs come from oracle/rroi_align_oracle.c, which is pinned
against the reference's JPEG artefacts by tests/test_oracle_kat.py.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import workloads as Wk  # noqa: E402
from oracle import rroi_align_oracle as O  # noqa: E402


def pack(name, feats, rois, ph, pw, scale):
    out, geom = O.forward_c(feats, rois, ph, pw, scale, return_geom=True)
    gout = (2.0 * np.nan_to_num(out)).astype(np.float32)  # d/dx of sum(x^2), test2.py:74
    gin = O.backward_c(gout, rois, feats.shape, scale)
    np.savez_compressed(os.path.join(HERE, name), features=feats, rois=rois,
                        pooled=np.asarray([ph, pw], np.int32), scale=np.float32(scale), out=out,
                        geom=geom, grad_in=gin)
    print(name, out.shape, "nonzero", float((out != 0).mean()))


if __name__ == "__main__":
    f, r = Wk.cfg1_inputs()
    pack("oracle_cfg1.npz", f, r, 8, 32, 1.0)
    f, r = Wk.bench_inputs(R=48, C=4, seed=3)
    pack("oracle_mid.npz", f, r, 8, 64, 0.25)
    rng = np.random.default_rng(7)
    f = rng.standard_normal((1, 2, 160, 160), dtype=np.float32)
    pack("oracle_edge.npz", f, Wk.edge_rois(), 8, 64, 0.25)
