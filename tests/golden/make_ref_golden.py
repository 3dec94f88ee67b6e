"""Run the REFERENCE's own kernels (oracle/_ref/librroi_ref_hip_nofma.so = rroi_align_kernel.cu through
ROCm's hipify-perl + hipcc with -ffp-contract=off, i.e. the SOURCE semantics the oracle and the
product implement; see oracle/Makefile) on an MI355X and freeze their outputs.  The same sources
built with the compiler's default FMA contraction (librroi_ref_hip.so -- what nvcc's default
-fmad=true did to the shipped binary) run next to it: `out_fma_diff` records the output elements at
which the two builds differ (rounding ties flipped by contraction), so that drift stays visible:

    gpurun -- 'python tests/golden/make_ref_golden.py'      # writes gpurun_out/refhip_*.npz
    cp gpurun_out/refhip_*.npz tests/golden/

The host side reproduces functions/rroi_align.py:13-40: zero-filled output / idx_x / idx_y, then
RROIAlignForwardLaucher; zero-filled grad_input, then RROIAlignBackwardLaucher on
grad_output = 2 * output (d/dx of pooled.pow(2).sum(), rroi_align/test2.py:74).
"""
import ctypes
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import workloads as Wk  # noqa: E402

vp, fl, it = ctypes.c_void_p, ctypes.c_float, ctypes.c_int


def _load(name):
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", name))
    lib.RROIAlignForwardLaucher.argtypes = [vp, fl, it, it, it, it, it, it, vp, vp, vp, vp, vp]
    lib.RROIAlignBackwardLaucher.argtypes = [vp, fl, it, it, it, it, it, it, it, vp, vp, vp, vp, vp]
    return lib


ref = _load("librroi_ref_hip_nofma.so")
ref_fma = _load("librroi_ref_hip.so")


def run(name, feats, rois, ph, pw, scale):
    F, R = torch.from_numpy(feats).cuda(), torch.from_numpy(rois).cuda()
    B, C, H, W = feats.shape
    n = rois.shape[0]
    out, ix, iy = (torch.zeros((n, C, ph, pw), device="cuda") for _ in range(3))
    st = torch.cuda.current_stream().cuda_stream
    ref.RROIAlignForwardLaucher(F.data_ptr(), scale, n, H, W, C, ph, pw, R.data_ptr(), out.data_ptr(),
                                ix.data_ptr(), iy.data_ptr(), st)
    gout = (2 * torch.nan_to_num(out)).contiguous()
    gin = torch.zeros_like(F)
    ref.RROIAlignBackwardLaucher(gout.data_ptr(), scale, B, n, H, W, C, ph, pw, R.data_ptr(),
                                 gin.data_ptr(), ix.data_ptr(), iy.data_ptr(), st)
    out_f, ix_f, iy_f = (torch.zeros((n, C, ph, pw), device="cuda") for _ in range(3))
    ref_fma.RROIAlignForwardLaucher(F.data_ptr(), scale, n, H, W, C, ph, pw, R.data_ptr(), out_f.data_ptr(),
                                    ix_f.data_ptr(), iy_f.data_ptr(), st)
    torch.cuda.synchronize()
    # flat indices (into the (n, ph, pw) bin grid) of the bins whose centre the contracted build moves
    moved = ((ix_f[:, 0] != ix[:, 0]) | (iy_f[:, 0] != iy[:, 0])).flatten().nonzero().flatten().cpu().numpy()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", name), features=feats, rois=rois,
                        pooled=np.asarray([ph, pw], np.int32), scale=np.float32(scale),
                        out=out.cpu().numpy(), idx_x=ix[:, 0].cpu().numpy(), idx_y=iy[:, 0].cpu().numpy(),
                        idx_same_over_channels=bool((ix == ix[:, :1]).all() and (iy == iy[:, :1]).all()),
                        grad_in=gin.cpu().numpy(), fma_moved_bins=moved.astype(np.int64),
                        fma_idx_x=ix_f[:, 0].flatten()[moved].cpu().numpy(),
                        fma_idx_y=iy_f[:, 0].flatten()[moved].cpu().numpy(),
                        fma_out_diff=int((torch.nan_to_num(out_f) != torch.nan_to_num(out)).sum()))
    print(name, tuple(out.shape), "nonzero", float((out != 0).float().mean()), "bins moved by contraction:",
          len(moved), "of", n * ph * pw)


if __name__ == "__main__":
    f, r = Wk.cfg1_inputs()
    run("refhip_cfg1.npz", f, r, 8, 32, 1.0)
    f, r = Wk.bench_inputs(R=48, C=4, seed=3)
    run("refhip_mid.npz", f, r, 8, 64, 0.25)
    rng = np.random.default_rng(7)
    f = rng.standard_normal((1, 2, 160, 160), dtype=np.float32)
    run("refhip_edge.npz", f, Wk.edge_rois(), 8, 64, 0.25)
    f, r = Wk.bench_inputs(R=16, C=3, H=90, W=120, img=480, seed=5)
    run("refhip_ph11.npz", f, r, 11, 77, 0.25)
    f = rng.standard_normal((1, 2, 160, 160), dtype=np.float32)
    run("refhip_ties.npz", f, Wk.tie_rois(), 8, 64, 0.25)
