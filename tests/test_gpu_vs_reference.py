"""GPU (MI355X): our op next to the REFERENCE'S OWN KERNELS on the same device.

oracle/_ref/librroi_ref_hip.so is rroi_align_kernel.cu (forward :28-162, backward :193-278, with
its launchers) run through ROCm's hipify-perl and compiled by hipcc (oracle/Makefile: ref).  It is
built where /root/reference exists and travels to the GPU box as a binary; these tests skip when
it is absent.  The host side reproduces functions/rroi_align.py:13-40 (zero-filled buffers)."""
import ctypes
import os

import numpy as np
import pytest
import torch

import workloads as Wk

pytestmark = pytest.mark.gpu

REFDIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")


def _load(name):
    path = os.path.join(REFDIR, name)
    if not os.path.exists(path):
        if os.path.isdir("/root/reference"):
            pytest.fail("oracle/_ref is not built although /root/reference is here: run `make -C oracle ref` "
                        "(__graft_entry__.build() does)")
        pytest.skip("oracle/_ref not built (it is built where /root/reference exists and travels as a binary)")
    lib = ctypes.CDLL(path)
    vp, fl, it = ctypes.c_void_p, ctypes.c_float, ctypes.c_int
    lib.RROIAlignForwardLaucher.argtypes = [vp, fl, it, it, it, it, it, it, vp, vp, vp, vp, vp]
    lib.RROIAlignBackwardLaucher.argtypes = [vp, fl, it, it, it, it, it, it, it, vp, vp, vp, vp, vp]
    return lib


@pytest.fixture(scope="module")
def ref():
    """The reference kernels with the source's arithmetic (-ffp-contract=off): every * and +
    rounded separately -- the semantics the oracle and the product implement."""
    return _load("librroi_ref_hip_nofma.so")


@pytest.fixture(scope="module")
def ref_fma():
    """The same sources with the compiler's default contraction (like nvcc's -fmad=true)."""
    return _load("librroi_ref_hip.so")


def ref_forward(lib, F, R, ph, pw, scale):
    n, (B, C, H, W) = R.shape[0], F.shape
    out, ix, iy = (torch.zeros((n, C, ph, pw), device="cuda") for _ in range(3))
    lib.RROIAlignForwardLaucher(F.data_ptr(), scale, n, H, W, C, ph, pw, R.data_ptr(), out.data_ptr(),
                                ix.data_ptr(), iy.data_ptr(), torch.cuda.current_stream().cuda_stream)
    return out, ix, iy


@pytest.mark.parametrize("case", ["cfg2_full", "train_like", "train_many", "image_like"])
def test_forward_identical_to_reference_kernel(ref, case):
    from rroi_align._ext import rroi_align as ext
    if case == "cfg2_full":      # BASELINE configs[1]
        f, r = Wk.bench_inputs()
        ph, pw, s = 8, 64, 0.25
    elif case == "train_like":   # src/ocr_process.py:259-267 regime
        f, r = Wk.bench_inputs(R=32, C=64, H=120, W=160, img=640, seed=5, batch=2)
        ph, pw, s = 11, 83, 0.25
    elif case == "train_many":   # the same pooled shape with enough ROIs for the SHIFT kernels
        f, r = Wk.bench_inputs(R=400, C=64, H=120, W=160, img=640, seed=8, batch=2)
        ph, pw, s = 11, 83, 0.25
    else:                        # rroi_align/test2.py regime: image as the map, 44 x 349
        f, r = Wk.bench_inputs(R=3, C=3, H=276, W=500, img=500, seed=6)
        ph, pw, s = 44, 349, 1.0
    F, R = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
    want, ix, iy = ref_forward(ref, F, R, ph, pw, s)
    for path in (ext.PATH_TILED, ext.PATH_DIRECT):
        got = ext.forward(F, R, ph, pw, s, path=path)
        ndiff = int((got != want).sum())
        assert ndiff == 0, f"{case} path {path}: {ndiff} of {got.numel()} elements differ from the reference kernel"
    # the reference-ABI entry point of this library fills con_idx_x / con_idx_y identically
    out2, ix2, iy2 = (torch.empty_like(want) for _ in range(3))
    assert ext.rroi_align_forward_cuda(ph, pw, s, F, R, out2, ix2, iy2) == 1
    assert torch.equal(out2, want) and torch.equal(ix2, ix) and torch.equal(iy2, iy)


def test_backward_matches_reference_kernel(ref):
    from rroi_align._ext import rroi_align as ext
    f, r = Wk.bench_inputs(R=128, C=64, seed=21)
    F, R = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
    out, ix, iy = ref_forward(ref, F, R, 8, 64, 0.25)
    gout = torch.randn_like(out)
    want = torch.zeros_like(F)
    ref.RROIAlignBackwardLaucher(gout.data_ptr(), 0.25, 1, 128, 160, 160, 64, 8, 64, R.data_ptr(),
                                 want.data_ptr(), ix.data_ptr(), iy.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
    scale = float(want.abs().max())
    for path in (ext.PATH_TILED_LISTS, ext.PATH_TILED_BUCKETS, ext.PATH_TILED_INKERNEL, ext.PATH_TILED_ATOMIC, ext.PATH_DIRECT):
        got = ext.backward(gout, R, f.shape, 0.25, path=path)
        assert float((got - want).abs().max()) <= 1e-4 * scale


def test_contracted_build_differs_only_at_rounding_ties(ref_fma):
    """with FMA contraction the same source flips round() at a few ties
    (measured there: 6 of 2,097,152 bins).  The product follows the un-contracted source; against
    the contracted build it may differ in a handful of bins, each by a half-pixel shift of the
    sample point -- never more than a few per million."""
    from rroi_align._ext import rroi_align as ext
    total_bins = diff_bins = 0
    for seed in (6, 7, 8):
        f, r = Wk.bench_inputs(R=8, C=3, H=276, W=500, img=500, seed=seed)
        F, R = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
        want, _, _ = ref_forward(ref_fma, F, R, 44, 349, 1.0)
        got = ext.forward(F, R, 44, 349, 1.0)
        diff_bins += int((got != want).any(1).sum())
        total_bins += got.shape[0] * got.shape[2] * got.shape[3]
    assert diff_bins <= max(8, 2e-5 * total_bins), (diff_bins, total_bins)


def test_cos_sin_recipe_drift_budget_against_the_reference_build(ref):
    """the one library-dependent step.  The hipified reference evaluates ocml's
    cos(float) / sin(float) (rroi_align_kernel.cu:73-74); oracle and product evaluate
    (float)cos((double)angle) (csrc/rroi_device_common.h).  ocml's float cosine is not correctly
    rounded, so over enough ROIs some affines differ in the last place, and where that meets a
    rounding tie a bin's sample point moves by half a pixel.  Measured on MI355X / ROCm 7.2 with
    tools/fuzz_ref.py (profiles/r03_fuzz_ref.json): 565 of 33,554,432 bins (16.8 per million; 163 of
    65,536 ROIs, a quarter of them built to sit on ties), every one a half-pixel move, and NO output
    element differs where the sample points agree.  This test re-measures a 4 M-bin slice of that sweep
    and holds the budget."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "fuzz_ref", os.path.join(os.path.dirname(REFDIR), "..", "tools", "fuzz_ref.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    from rroi_align._ext import rroi_align as ext
    rng = np.random.default_rng(11)
    ph, pw, s, H, W = 8, 64, 0.25, 160, 160
    F = torch.from_numpy(rng.standard_normal((1, 1, H, W), dtype=np.float32)).cuda()
    bins = moved = wrong_elsewhere = 0
    shift = 0.0
    for _ in range(2):
        r = fz.random_rois(rng, 4096)
        R = torch.from_numpy(r).cuda()
        want, ix, iy = ref_forward(ref, F, R, ph, pw, s)
        geom = ext.bin_centres(R, ph, pw, s, H, W)
        dxy = (geom[..., 0] != ix[:, 0]) | (geom[..., 1] != iy[:, 0])
        got = ext.forward(F, R, ph, pw, s)
        d = ~((got == want) | (got.isnan() & want.isnan()))
        bins += dxy.numel()
        moved += int(dxy.sum())
        wrong_elsewhere += int((d[:, 0] & ~dxy).sum())
        shift = max(shift, float(torch.maximum((geom[..., 0] - ix[:, 0]).abs(), (geom[..., 1] - iy[:, 0]).abs()).max()))
    assert bins >= 4_000_000
    assert wrong_elsewhere == 0, "outputs differ although the sample points agree"
    assert moved <= 50e-6 * bins, (moved, bins)      # measured: 16.8 per million on this mix
    assert shift <= 1.0


def test_ocml_trig_recipe_moves_no_bin(ref):
    """the opt-in fp32 recipe (`trig=ext.TRIG_FP32`, PER CALL since round 5: cosf / sinf of
    the device library, as the reference's sources call them, kernel.cu:73-74) against the reference's own kernels
    built for this GPU: >= 33 M bins over all angles, a quarter of the ROIs on rounding ties (the mix on which the
    default recipe moves 16.8 bins per million) -- NO bin's sample point moves, no output element differs, on either
    forward path.  Nothing is switched device-wide: calls with the default recipe in between stay bit-exact against
    the oracle (the last assertion)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "fuzz_ref", os.path.join(os.path.dirname(REFDIR), "..", "tools", "fuzz_ref.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    from rroi_align._ext import rroi_align as ext
    rng = np.random.default_rng(11)          # the seed of profiles/r03_fuzz_ref.json
    ph, pw, s, H, W = 8, 64, 0.25, 160, 160
    F = torch.from_numpy(rng.standard_normal((1, 1, H, W), dtype=np.float32)).cuda()
    T = ext.TRIG_FP32
    bins = moved = differ = moved_default = 0
    for rnd in range(16):
        r = fz.random_rois(rng, 4096)
        R = torch.from_numpy(r).cuda()
        want, ix, iy = ref_forward(ref, F, R, ph, pw, s)
        geom = ext.bin_centres(R, ph, pw, s, H, W, trig=T)
        moved += int(((geom[..., 0] != ix[:, 0]) | (geom[..., 1] != iy[:, 0])).sum())
        g0 = ext.bin_centres(R, ph, pw, s, H, W)
        moved_default += int(((g0[..., 0] != ix[:, 0]) | (g0[..., 1] != iy[:, 0])).sum())
        for path in (ext.PATH_TILED, ext.PATH_DIRECT):
            got = ext.forward(F, R, ph, pw, s, path=path, trig=T)
            differ += int((~((got == want) | (got.isnan() & want.isnan()))).sum())
        bins += geom[..., 0].numel()
    assert bins >= 33_000_000
    assert moved == 0 and differ == 0, (moved, differ, bins)
    assert 0 < moved_default <= 50e-6 * bins      # the default recipe, interleaved with the fp32 calls: its own result
    # training-like pooled shape (11 x 83: the SHIFT kernels), several images, through the backward as well
    f, r = Wk.bench_inputs(R=400, C=64, H=120, W=160, img=640, seed=8, batch=2)
    F2, R2 = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
    want, ix, iy = ref_forward(ref, F2, R2, 11, 83, 0.25)
    assert torch.equal(ext.forward(F2, R2, 11, 83, 0.25, trig=T), want)
    gout = torch.randn_like(want)
    gwant = torch.zeros_like(F2)
    ref.RROIAlignBackwardLaucher(gout.data_ptr(), 0.25, 2, 400, 120, 160, 64, 11, 83, R2.data_ptr(), gwant.data_ptr(),
                                 ix.data_ptr(), iy.data_ptr(), torch.cuda.current_stream().cuda_stream)
    got = ext.backward(gout, R2, f.shape, 0.25, trig=T)
    assert float((got - gwant).abs().max()) <= 1e-4 * float(gwant.abs().max())


def test_trig_recipe_is_per_call(ref):
    """the recipe is carried by the call.  Two streams run the two recipes CONCURRENTLY on the
    same ROIs (a tie-heavy draw on which the recipes give different sample points) and each gets its own bit-exact
    result -- TRIG_DOUBLE the oracle's, TRIG_FP32 the reference build's; forward + backward are capturable with
    either recipe and a replay reproduces it; the autograd surface hands the forward's recipe to its backward."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "fuzz_ref", os.path.join(os.path.dirname(REFDIR), "..", "tools", "fuzz_ref.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    from oracle import rroi_align_oracle as O
    from rroi_align._ext import rroi_align as ext
    from rroi_align.modules.rroi_align import _RRoiAlign
    rng = np.random.default_rng(11)
    ph, pw, s, H, W, C = 8, 64, 0.25, 160, 160, 8
    f = rng.standard_normal((1, C, H, W), dtype=np.float32)
    r = fz.random_rois(rng, 4096)
    F, R = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
    g_d, g_f = ext.bin_centres(R, ph, pw, s, H, W), ext.bin_centres(R, ph, pw, s, H, W, trig=ext.TRIG_FP32)
    assert int((g_d != g_f).any(-1).sum()) > 0, "the draw does not separate the recipes"
    want_f, _, _ = ref_forward(ref, F, R, ph, pw, s)
    want_d = torch.from_numpy(O.forward_c(f, r, ph, pw, s)).cuda()
    assert not torch.equal(want_f, want_d)
    sd, sf = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = {}
    for rep in range(6):      # interleaved launches on two streams: the kernels of the two recipes overlap
        for name, st_, trig in (("d", sd, ext.TRIG_DOUBLE), ("f", sf, ext.TRIG_FP32)):
            with torch.cuda.stream(st_):
                for path in (ext.PATH_TILED, ext.PATH_DIRECT):
                    outs[name, path, rep] = ext.forward(F, R, ph, pw, s, path=path, trig=trig)
    torch.cuda.synchronize()
    for (name, path, rep), got in outs.items():
        want = want_d if name == "d" else want_f
        assert bool(((got == want) | (got.isnan() & want.isnan())).all()), (name, path, rep)
    del outs
    # graph capture with either recipe (native entry points, caller's workspace)
    n = 256
    Rn, B = R[:n].contiguous(), 1
    out = torch.empty((n, C, ph, pw), device="cuda")
    gin = torch.empty(f.shape, device="cuda")
    nf = ext._lib.rroi_align_forward_workspace_bytes(B, C, H, W, n, 0)
    nb = ext._lib.rroi_align_backward_workspace_bytes(B, C, H, W, n, ph, pw)
    wf, wb = torch.empty(nf, dtype=torch.uint8, device="cuda"), torch.empty(nb, dtype=torch.uint8, device="cuda")
    for trig, want in ((ext.TRIG_DOUBLE, want_d), (ext.TRIG_FP32, want_f)):
        word = ext.PATH_TILED | (ext.PATH_TRIG_FP32 if trig else 0)

        def calls():
            st_ = torch.cuda.current_stream().cuda_stream
            assert ext._lib.rroi_align_forward_hip(F.data_ptr(), 0, s, B, n, H, W, C, ph, pw, Rn.data_ptr(),
                                                   out.data_ptr(), wf.data_ptr(), nf, word, st_) == 1
            assert ext._lib.rroi_align_backward_hip(out.data_ptr(), s, B, n, H, W, C, ph, pw, Rn.data_ptr(),
                                                    gin.data_ptr(), wb.data_ptr(), nb, word, st_) == 1
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            calls()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gin_eager = gin.clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            calls()
        out.zero_()
        gin.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert bool(((out == want[:n]) | (out.isnan() & want[:n].isnan())).all()), trig
        assert float((gin - gin_eager).abs().max()) <= 1e-4 * max(1.0, float(gin_eager.abs().max()))
    # autograd: the backward of a TRIG_FP32 forward recomputes the bin centres with TRIG_FP32
    Fg = F.clone().requires_grad_(True)
    y = _RRoiAlign(ph, pw, s, trig=ext.TRIG_FP32)(Fg, Rn)
    assert bool(((y == want_f[:n]) | (y.isnan() & want_f[:n].isnan())).all())
    gout = torch.randn_like(y)
    y.backward(gout)
    g_f32 = ext.backward(gout, Rn, f.shape, s, trig=ext.TRIG_FP32)
    g_dbl = ext.backward(gout, Rn, f.shape, s, trig=ext.TRIG_DOUBLE)
    scale = max(1.0, float(g_f32.abs().max()))
    assert float((Fg.grad - g_f32).abs().max()) <= 1e-5 * scale
    # (where the recipes place a bin half a pixel apart the two gradients differ by whole tap weights)
    if int((g_d[:n] != g_f[:n]).any(-1).sum()) > 0:
        assert float((g_dbl - g_f32).abs().max()) > 1e-3


def test_launcher_trig_environment(ref):
    """The reference-ABI launchers carry no `path`: RROI_ALIGN_LAUNCHER_TRIG=fp32 in the environment of the process
    (read once) makes them evaluate the device library's cosf / sinf, and then con_idx_x / con_idx_y and the crops
    equal the reference build's in EVERY bin of a tie-heavy draw; without it they are the default recipe's.  Run in
    a subprocess: the setting is a constant of the process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(REFDIR))
    code = r'''
import ctypes, os, sys, importlib.util
import numpy as np, torch
root = sys.argv[1]
sys.path[:0] = [root, os.path.join(root, "fots.pytorch_amd"), os.path.join(root, "tests")]
from rroi_align._ext import rroi_align as ext
spec = importlib.util.spec_from_file_location("fuzz_ref", os.path.join(root, "tools", "fuzz_ref.py"))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
ref = fz.load_ref()
rng = np.random.default_rng(11)
F = torch.from_numpy(rng.standard_normal((1, 2, 160, 160), dtype=np.float32)).cuda()
moved = 0
for n in (4096, 12):     # the tiled launcher path and the direct one
    R = torch.from_numpy(fz.random_rois(rng, n)).cuda()
    want, ix, iy = (torch.zeros((n, 2, 8, 64), device="cuda") for _ in range(3))
    ref.RROIAlignForwardLaucher(F.data_ptr(), 0.25, n, 160, 160, 2, 8, 64, R.data_ptr(), want.data_ptr(), ix.data_ptr(),
                                iy.data_ptr(), torch.cuda.current_stream().cuda_stream)
    out, ox, oy = (torch.empty_like(want) for _ in range(3))
    assert ext.rroi_align_forward_cuda(8, 64, 0.25, F, R, out, ox, oy) == 1
    moved += int(((ox != ix) | (oy != iy)).sum())
    if os.environ.get("RROI_ALIGN_LAUNCHER_TRIG") == "fp32":
        assert bool(((out == want) | (out.isnan() & want.isnan())).all())
print("MOVED", moved)
'''
    res = {}
    for mode in ("fp32", ""):
        env = dict(os.environ)
        env.pop("RROI_ALIGN_LAUNCHER_TRIG", None)
        if mode:
            env["RROI_ALIGN_LAUNCHER_TRIG"] = mode
        p = subprocess.run([sys.executable, "-c", code, root], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        res[mode] = int(p.stdout.strip().split("MOVED")[-1])
    assert res["fp32"] == 0 and res[""] > 0, res
