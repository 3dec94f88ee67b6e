"""GPU (MI355X): the HIP path against the oracle, through the C-ABI and the reference's
Python surface.  Forward: BIT-EXACT (north_star allows 1e-5; the arithmetic recipe is
shared, so any difference is a bug).  Backward: 1e-4 relative to the gradient scale
(fp32 atomics commute but do not associate; BASELINE.json configs[2])."""
import os

import numpy as np
import pytest
import torch

import workloads as Wk

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FWD_TOL = 0.0          # bit-exact
BWD_RTOL = 1e-4        # of max |grad|


@pytest.fixture(scope="module")
def ext():
    from rroi_align._ext import rroi_align as e
    return e


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def eq(a, b):
    return np.array_equal(a, b, equal_nan=True)


def run_fwd(ext, f, r, ph, pw, s, path):
    return ext.forward(dev(f), dev(r), ph, pw, s, path=path).cpu().numpy()


def mismatch(a, b):
    bad = ~((a == b) | (np.isnan(a) & np.isnan(b)))
    return int(bad.sum()), (float(np.nanmax(np.abs(a - b))) if bad.any() else 0.0)


SHAPES = {
    # name: (features/rois factory, ph, pw, scale)
    "cfg1": lambda: (*Wk.cfg1_inputs(), 8, 32, 1.0),                                  # BASELINE configs[0]
    "mid_c64": lambda: (*Wk.bench_inputs(R=40, C=64, seed=11), 8, 64, 0.25),
    "train_11xceil": lambda: (*Wk.bench_inputs(R=32, C=64, H=120, W=160, img=640, seed=5), 11, 83, 0.25),
    "image_c3_44x349": lambda: (*Wk.bench_inputs(R=5, C=3, H=276, W=500, img=500, seed=6), 44, 349, 1.0),
    "c70_odd": lambda: (*Wk.bench_inputs(R=9, C=70, H=50, W=70, img=280, seed=7), 8, 33, 0.25),
    "c5_pad": lambda: (*Wk.bench_inputs(R=17, C=5, H=64, W=64, img=256, seed=8), 32, 57, 0.25),
    "batch3": lambda: (*Wk.bench_inputs(R=48, C=32, H=64, W=64, img=256, seed=9, batch=3), 8, 40, 0.25),
    "r1_infer": lambda: (*Wk.bench_inputs(R=1, C=64, H=176, W=320, img=1280, seed=10), 11, 64, 0.25),
}


@pytest.mark.parametrize("path", ["direct", "tiled", "fused", "auto"])
@pytest.mark.parametrize("name", list(SHAPES))
def test_forward_bit_exact(ext, oracle, name, path):
    f, r, ph, pw, s = SHAPES[name]()
    want = oracle.forward_c(f, r, ph, pw, s, threads=8)
    got = run_fwd(ext, f, r, ph, pw, s, {"direct": ext.PATH_DIRECT, "tiled": ext.PATH_TILED, "fused": ext.PATH_FUSED,
                                         "auto": ext.PATH_AUTO}[path])
    n, d = mismatch(got, want)
    assert n == 0 and d <= FWD_TOL, f"{n} elements differ, max |d| = {d}"


@pytest.mark.parametrize("case", ["train_c64_b2", "c32_b3", "c40_b2_cl", "many_rois", "one_image_band", "more_images_than_buckets",
                                  "merge_11x83", "merge_11x82", "merge_11x85", "merge_c32_11x84"])
def test_forward_xcd_groups(ext, oracle, case):
    """Round 5: C <= 64 (one or two channel chunks) and >= 64 ROIs take the XCD-GROUP form of the tiled forward -- the
    ROIs are counting-sorted by (image, centre row) inside the prologue launch and each group of XCDs gathers one
    quantile of them.  The order only ever decides WHO computes a crop: every case is bit-exact against the oracle,
    including ROIs whose image index is invalid, centres at NaN and infinity (they sort to an end), more than
    1024 ROIs (the sort's ranks beyond its registers go through memory), more images than the sort has buckets,
    channels-last features consumed in place and channels-last crops.
    The merge_* cases are large enough (>= 64 MB of crops over a small map) for the MERGING form on rows that are not whole
    64-byte sectors: strided tiles whose partial sectors meet in the group's L2, 16-byte stores at dword alignment and a
    row's last one to three dwords stored singly (913, 902 and 935 bins a row: 1, 2 and 3 of them; 924: none)."""
    rng = np.random.default_rng(4242)
    cl = False
    if case == "train_c64_b2":
        B, C, H, W, R, ph, pw = 2, 64, 120, 160, 300, 11, 83
    elif case == "c32_b3":
        B, C, H, W, R, ph, pw = 3, 32, 64, 96, 200, 8, 64
    elif case == "c40_b2_cl":
        B, C, H, W, R, ph, pw, cl = 2, 40, 50, 70, 130, 8, 40, True
    elif case == "many_rois":
        B, C, H, W, R, ph, pw = 2, 8, 40, 60, 1500, 4, 16
    elif case == "one_image_band":
        B, C, H, W, R, ph, pw = 4, 64, 48, 64, 128, 11, 32
    elif case.startswith("merge_c32"):
        B, C, H, W, R, ph, pw, cl = 2, 32, 60, 80, 600, 11, 84, True   # (+ the merging form over channels-last features in place)
    elif case.startswith("merge_"):
        B, C, H, W, R, ph, pw = 2, 64, 60, 80, 300, 11, int(case[-2:])
        cl = pw == 85
    else:
        B, C, H, W, R, ph, pw = 1100, 4, 8, 8, 2300, 2, 8
    f = rng.standard_normal((B, C, H, W), dtype=np.float32)
    h = rng.uniform(8, 40, R)
    r = np.stack([rng.integers(0, B, R), rng.uniform(-10, 4 * W + 10, R), rng.uniform(-10, 4 * H + 10, R), h,
                  h * rng.uniform(1, 8, R), rng.uniform(-90, 90, R)], 1).astype(np.float32)
    if case == "one_image_band":          # every ROI in a thin band of image 2
        r[:, 0], r[:, 2] = 2, rng.uniform(40, 60, R)
    # invalid image indices (zeros here; the reference reads out of bounds, so the oracle gets a valid index and the
    # rows are zeroed) and wild centres: they sort to the ends, the op treats them as the reference does
    r[17, 2], r[19, 2], r[23, 2], r[29, 1] = np.nan, np.inf, -np.inf, np.nan
    want = oracle.forward_c(f, r, ph, pw, 0.25, threads=8)
    bad = [3, 7, 11]
    r[3, 0], r[7, 0], r[11, 0] = -1, B, 1e9
    want[bad] = 0
    got = run_fwd(ext, f, r, ph, pw, 0.25, ext.PATH_TILED)
    n, d = mismatch(got, want)
    assert n == 0, f"{case}: {n} elements differ, max |d| = {d}"
    if cl:
        F = dev(f).contiguous(memory_format=torch.channels_last)
        got_cl = ext.forward(F, dev(r), ph, pw, 0.25, path=ext.PATH_TILED)
        assert eq(got_cl.cpu().numpy(), want)
        out_cl = ext.forward(dev(f), dev(r), ph, pw, 0.25, path=ext.PATH_TILED, channels_last_out=True)
        assert out_cl.is_contiguous(memory_format=torch.channels_last) and eq(out_cl.cpu().numpy(), want)
    # the launches separately (the bench brackets them): the gather needs what THIS prologue wrote, the sort included
    Fd, Rd = dev(f), dev(r)
    out = torch.full((R, C, ph, pw), float("nan"), device="cuda")
    nb = ext._lib.rroi_align_forward_workspace_bytes(B, C, H, W, R, 0)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for stage in (ext.STAGE_PROLOGUE, ext.STAGE_GATHER):
        assert ext._lib.rroi_align_forward_stages_hip(Fd.data_ptr(), 0, 0.25, B, R, H, W, C, ph, pw, Rd.data_ptr(), out.data_ptr(),
                                                      ws.data_ptr(), nb, ext.PATH_TILED, stage, st) == 1
    assert eq(out.cpu().numpy(), want)


@pytest.mark.parametrize("pw", [96, 83, 50])
def test_forward_beyond_the_cache(ext, oracle, pw):
    """Crops beyond the 256 MB memory-side cache.  XCD groups are kept at ANY size (forward_groups: no size limit), for rows
    of whole sectors (11 x 96, 297 MB) and for rows that are not (11 x 83: 330 MB at C = 64; 11 x 50 at C = 136 -- a last
    chunk of eight channels --: 335 MB); the latter take, from 320 MB up, the LINE-aligned form (SHIFT == 2: a tile stores
    32 bins of the 64 it gathers, every store a whole 128-byte line).  The tiled path against the direct kernel on all of
    it, and against the oracle on a sample of the ROIs.  ADVICE r05: the ONE-LAUNCH form (RROI_PATH_FUSED) has no
    line-aligned instantiation -- on such crops it must run its SHIFT form (it used to launch SHIFT = 0 on a grid sized for
    32-bin tiles: a row's last 16-byte store ran into the next channel's first bins when NB % 4 != 0)."""
    rng = np.random.default_rng(99)
    B, C, H, W, R, ph = 2, 64, 60, 80, 1100 if pw == 96 else 1480, 11
    if pw == 50:
        C, R = 136, 1170
    f = rng.standard_normal((B, C, H, W), dtype=np.float32)
    h = rng.uniform(8, 40, R)
    r = np.stack([rng.integers(0, B, R), rng.uniform(-10, 4 * W + 10, R), rng.uniform(-10, 4 * H + 10, R), h,
                  h * rng.uniform(1, 9, R), rng.uniform(-90, 90, R)], 1).astype(np.float32)
    Fd, Rd = dev(f), dev(r)
    got = ext.forward(Fd, Rd, ph, pw, 0.25, path=ext.PATH_TILED)
    assert got.numel() * 4 > (256 << 20 if pw == 96 else 320 << 20)
    direct = ext.forward(Fd, Rd, ph, pw, 0.25, path=ext.PATH_DIRECT)
    assert torch.equal(got.view(torch.int32), direct.view(torch.int32))
    pick = np.sort(rng.choice(R, 48, replace=False))
    want = oracle.forward_c(f, r[pick], ph, pw, 0.25, threads=8)
    assert eq(got[torch.from_numpy(pick).cuda()].cpu().numpy(), want)
    del got
    fused = ext.forward(Fd, Rd, ph, pw, 0.25, path=ext.PATH_FUSED)
    assert torch.equal(fused.view(torch.int32), direct.view(torch.int32))


def test_forward_tiny_maps_every_path(ext, oracle):
    """Maps of one to five pixels a side (the direct path's row PAIRS -- 8-byte loads that start at x0, or at x0 - 1 on a
    row's last pixel -- have nowhere to go wrong but here; a map one pixel wide takes the thread-per-bin fallback), odd
    channel counts, ROIs larger than the map: every forward path bit-exact against the oracle."""
    rng = np.random.default_rng(77)
    for (H, W) in ((1, 1), (1, 2), (2, 1), (2, 2), (2, 3), (3, 2), (5, 4), (4, 5), (1, 7), (7, 1)):
        for C in (1, 3, 33):
            f = rng.standard_normal((2, C, H, W), dtype=np.float32)
            R = 12
            r = np.stack([rng.integers(0, 2, R), rng.uniform(-1, W + 1, R), rng.uniform(-1, H + 1, R), rng.uniform(0.5, 2 * H + 1, R),
                          rng.uniform(0.5, 3 * W + 1, R), rng.uniform(-90, 90, R)], 1).astype(np.float32)
            for (ph, pw) in ((2, 3), (3, 21), (8, 16)):
                want = oracle.forward_c(f, r, ph, pw, 1.0)
                for p in ext.FORWARD_PATHS:
                    got = run_fwd(ext, f, r, ph, pw, 1.0, p)
                    n, d = mismatch(got, want)
                    assert n == 0, f"{H}x{W} C={C} {ph}x{pw} path {p}: {n} elements differ, max |d| = {d}"


@pytest.mark.parametrize("path", ["direct", "tiled", "fused"])
def test_forward_edge_and_degenerate_rois(ext, oracle, path):
    rng = np.random.default_rng(1)
    f = rng.standard_normal((1, 36, 160, 160), dtype=np.float32)
    p = {"direct": ext.PATH_DIRECT, "tiled": ext.PATH_TILED, "fused": ext.PATH_FUSED}[path]
    for rois in (Wk.edge_rois(), Wk.degenerate_rois()):
        want = oracle.forward_c(f, rois, 8, 64, 0.25)
        got = run_fwd(ext, f, rois, 8, 64, 0.25, p)
        n, d = mismatch(got, want)
        assert n == 0, f"{n} elements differ (max {d})"


def test_forward_nonfinite_features(ext, oracle):
    """0 * inf = NaN in the reference's blend (kernel.cu:138-141): reproduced, not skipped."""
    f, r = Wk.bench_inputs(R=16, C=8, seed=12)
    f[0, :, 40:44, 50:54] = np.inf
    f[0, :, 90, 100] = np.nan
    want = oracle.forward_c(f, r, 8, 64, 0.25)
    for p in (ext.PATH_DIRECT, ext.PATH_TILED, ext.PATH_FUSED):
        assert mismatch(run_fwd(ext, f, r, 8, 64, 0.25, p), want)[0] == 0


def test_forward_nonfinite_scattered(ext, oracle):
    """+-inf, NaN and near-overflow values sprinkled over the whole map (3 % of the pixels), so that
    every blend class meets them: single-tap bins (weight 1), two-tap bins (1/2, 1/2) and four-tap
    bins (1/4 each), with the non-finite value on the weighted tap, on the zero-weight alias, or
    both.  The tiled kernel's two-term fast path must still equal the reference's four-term sum."""
    rng = np.random.default_rng(21)
    f, r = Wk.bench_inputs(R=96, C=40, seed=21)
    H, W = f.shape[2:]
    sel = rng.random((H, W))
    f[0, :, sel < 0.010] = np.inf
    f[0, :, (sel >= 0.010) & (sel < 0.018)] = -np.inf
    f[0, :, (sel >= 0.018) & (sel < 0.024)] = np.nan
    f[0, :, (sel >= 0.024) & (sel < 0.030)] = 3.0e38
    f[0, ::2, (sel >= 0.024) & (sel < 0.030)] = -3.0e38
    want = oracle.forward_c(f, r, 8, 64, 0.25)
    assert np.isnan(want).any() and np.isinf(want).any()
    for p in (ext.PATH_DIRECT, ext.PATH_TILED, ext.PATH_FUSED):
        n, d = mismatch(run_fwd(ext, f, r, 8, 64, 0.25, p), want)
        assert n == 0, f"path {p}: {n} elements differ"


@pytest.mark.parametrize("name", ["oracle_cfg1", "oracle_mid", "oracle_edge"])
def test_golden_fixtures(ext, name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    ph, pw = (int(v) for v in z["pooled"])
    s = float(z["scale"])
    for p in (ext.PATH_DIRECT, ext.PATH_TILED, ext.PATH_FUSED):
        assert eq(run_fwd(ext, z["features"], z["rois"], ph, pw, s, p), z["out"])
    H, W = z["features"].shape[2:]
    geom = ext.bin_centres(dev(z["rois"]), ph, pw, s, H, W).cpu().numpy()
    assert eq(geom, z["geom"])
    gout = (2.0 * np.nan_to_num(z["out"])).astype(np.float32)
    for p in (ext.PATH_DIRECT, ext.PATH_TILED, ext.PATH_TILED_LISTS, ext.PATH_TILED_BUCKETS, ext.PATH_TILED_INKERNEL, ext.PATH_TILED_ATOMIC):
        gin = ext.backward(dev(gout), dev(z["rois"]), z["features"].shape, s, path=p).cpu().numpy()
        scale = max(1.0, float(np.abs(z["grad_in"]).max()))
        assert np.abs(gin - z["grad_in"]).max() <= BWD_RTOL * scale


def test_sincos_recipe_matches_host_libm(ext):
    """The one library-dependent step: (float)cos((double)a), (float)sin((double)a).
    Device (ocml) and host (glibc) must round to the same fp32 for every angle tried."""
    rng = np.random.default_rng(0)
    deg = np.concatenate([
        rng.uniform(-180, 180, 2_000_000), rng.uniform(-720, 720, 500_000),
        np.arange(-360, 361, 0.25), rng.uniform(-1e-3, 1e-3, 100_000), rng.uniform(-1e6, 1e6, 100_000),
    ]).astype(np.float32)
    got = ext.sincos_probe(dev(deg)).cpu().numpy()
    ang = ((deg.astype(np.float64) / 180.0) * 3.1415926535).astype(np.float32).astype(np.float64)
    want = np.stack([np.cos(ang), np.sin(ang)], 1).astype(np.float32)
    bad = int((got.view(np.uint32) != want.view(np.uint32)).sum())
    assert bad == 0, f"{bad} of {2 * len(deg)} sin/cos values differ between ocml and glibc"


@pytest.mark.parametrize("path", ["direct", "tiled", "tiled_lists", "tiled_buckets", "tiled_inkernel", "tiled_atomic"])
@pytest.mark.parametrize("name", ["cfg1", "mid_c64", "c70_odd", "c5_pad", "batch3", "train_11xceil"])
def test_backward_vs_oracle(ext, oracle, name, path):
    f, r, ph, pw, s = SHAPES[name]()
    out = oracle.forward_c(f, r, ph, pw, s, threads=8)
    gout = (2 * out).astype(np.float32)
    want = oracle.backward_c(gout, r, f.shape, s)
    p = {"direct": ext.PATH_DIRECT, "tiled": ext.PATH_TILED, "tiled_lists": ext.PATH_TILED_LISTS,
         "tiled_inkernel": ext.PATH_TILED_INKERNEL, "tiled_atomic": ext.PATH_TILED_ATOMIC,
         "tiled_buckets": ext.PATH_TILED_BUCKETS}[path]
    got = ext.backward(dev(gout), dev(r), f.shape, s, path=p).cpu().numpy()
    # max-abs <= 1e-4 (BASELINE configs[2]) -- except c5_pad, the heavy-overlap case of this list: 17 ROIs of 32 x 57
    # bins on a 64 x 64 map put hundreds of terms on a pixel, max |grad| = 250, where ONE fp32 ulp is 3e-5 and the
    # order of the sum alone moves the result by 1e-4 (measured 0.8 - 1.1e-4 = 4e-7 of the scale on every path, the
    # reference's unordered atomicAdds included): relative bar there
    Wk.check_backward(got, want, f"{name} {path}", require_abs=name != "c5_pad")
    assert eq(got == 0, want == 0) or np.abs(got[(got == 0) != (want == 0)]).max() < 1e-30


def test_backward_edge_rois(ext, oracle):
    rng = np.random.default_rng(2)
    f = rng.standard_normal((1, 8, 160, 160), dtype=np.float32)
    rois = np.concatenate([Wk.edge_rois(), Wk.degenerate_rois()[[0, 1, 2, 3, 4]]])
    gout = rng.standard_normal((len(rois), 8, 8, 64), dtype=np.float32)
    want = oracle.backward_c(gout, rois, f.shape, 0.25)
    for p in (ext.PATH_DIRECT, ext.PATH_TILED, ext.PATH_TILED_LISTS, ext.PATH_TILED_BUCKETS, ext.PATH_TILED_INKERNEL, ext.PATH_TILED_ATOMIC):
        got = ext.backward(dev(gout), dev(rois), f.shape, 0.25, path=p).cpu().numpy()
        assert np.abs(got - want).max() <= BWD_RTOL * max(1.0, float(np.abs(want).max()))


def test_backward_heavy_overlap(ext, oracle):
    """Hundreds of ROIs on the same few pixels: lists of thousands of pairs per pixel, more than 256
    candidate ROIs and more than 512 row segments per map tile (the in-kernel gather processes them in
    several ROI batches, segment flushes and rounds), two images, tall pooled grids."""
    rng = np.random.default_rng(77)
    for (R, C, H, W, ph, pw, B) in ((600, 8, 24, 40, 8, 24, 1), (300, 36, 16, 16, 16, 9, 2)):
        f = rng.standard_normal((B, C, H, W), dtype=np.float32)
        rois = np.zeros((R, 6), np.float32)
        rois[:, 0] = rng.integers(0, B, R)
        rois[:, 1] = (W / 2 + rng.uniform(-3, 3, R)) * 4
        rois[:, 2] = (H / 2 + rng.uniform(-3, 3, R)) * 4
        rois[:, 3] = rng.uniform(8, 40, R)
        rois[:, 4] = rois[:, 3] * rng.uniform(1, 4, R)
        rois[:, 5] = rng.uniform(-90, 90, R)
        rois[: R // 3] = rois[0]                      # identical copies
        gout = rng.standard_normal((R, C, ph, pw), dtype=np.float32)
        want = oracle.backward_c(gout, rois, f.shape, 0.25)
        for p in (ext.PATH_TILED_LISTS, ext.PATH_TILED_BUCKETS, ext.PATH_TILED_INKERNEL, ext.PATH_TILED_ATOMIC):
            got = ext.backward(dev(gout), dev(rois), f.shape, 0.25, path=p).cpu().numpy()
            assert np.abs(got - want).max() <= BWD_RTOL * max(1.0, float(np.abs(want).max())), (R, p)


def test_more_than_256_channels(ext, oracle):
    """C = 300: ten channel chunks -- the forward's chunk loop leaves the one-chunk-per-XCD regime
    and the gather backward needs a second channel pass (64 lanes cover 8 chunks)."""
    f, r = Wk.bench_inputs(R=20, C=300, H=40, W=56, img=224, seed=41, batch=2)
    want = oracle.forward_c(f, r, 8, 32, 0.25, threads=8)
    for p in (ext.PATH_DIRECT, ext.PATH_TILED, ext.PATH_FUSED):
        assert mismatch(run_fwd(ext, f, r, 8, 32, 0.25, p), want)[0] == 0
    gout = np.random.default_rng(41).standard_normal(want.shape).astype(np.float32)
    gwant = oracle.backward_c(gout, r, f.shape, 0.25)
    for p in (ext.PATH_DIRECT, ext.PATH_TILED, ext.PATH_TILED_LISTS, ext.PATH_TILED_BUCKETS, ext.PATH_TILED_INKERNEL, ext.PATH_TILED_ATOMIC):
        g = ext.backward(dev(gout), dev(r), f.shape, 0.25, path=p).cpu().numpy()
        assert np.abs(g - gwant).max() <= BWD_RTOL * max(1.0, float(np.abs(gwant).max())), f"path {p}"


def test_backward_gather_writes_nchw_in_place(ext, oracle):
    """Round 4: for C <= 128, and beyond that with at most 16 bins per map pixel, the list gathers store the caller's NCHW gradient themselves
    (rroi_bwd_gather_kernel<kDstNchw / kDstNchwAdd>): whole 32-byte sectors of a map row per channel out of an LDS
    tile, no chunk-major scratch, no relayout launch.  Maps whose width is not a multiple of 4 (scalar stores) or
    of 8 / height of 4 (padded key tiles), channel counts that need several passes over blockIdx.y with a ragged last
    chunk (C = 300: 10 chunks in groups of 4), two images; every element of a NaN-filled gradient is written; the
    reference-ABI launcher's accumulating form adds k times the gradient in k calls."""
    stream = torch.cuda.current_stream().cuda_stream
    for (R, C, H, W, ph, pw, B) in ((40, 300, 37, 61, 8, 32, 2), (64, 96, 61, 77, 8, 64, 1), (20, 256, 30, 44, 11, 83, 1),
                                    (64, 20, 33, 50, 8, 64, 1), (32, 130, 40, 56, 8, 64, 3)):
        f, r = Wk.bench_inputs(R=R, C=C, H=H, W=W, img=4 * W, seed=R + C, batch=B)
        assert C <= 128 or R * ph * pw <= 16 * B * H * W
        gout = np.random.default_rng(C).standard_normal((R, C, ph, pw)).astype(np.float32)
        want = oracle.backward_c(gout, r, f.shape, 0.25)
        scale = max(1.0, float(np.abs(want).max()))
        G, Rr = dev(gout), dev(r)
        nb = ext._lib.rroi_align_backward_workspace_bytes(B, C, H, W, R, ph, pw)
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        for p in (ext.PATH_TILED_LISTS, ext.PATH_TILED_BUCKETS):
            got = torch.full(f.shape, float("nan"), device="cuda")
            assert ext._lib.rroi_align_backward_hip(G.data_ptr(), 0.25, B, R, H, W, C, ph, pw, Rr.data_ptr(),
                                                    got.data_ptr(), ws.data_ptr(), nb, p, stream) == 1
            g = got.cpu().numpy()
            assert np.abs(g - want).max() <= BWD_RTOL * scale, (R, C, H, W, p)      # (NaN left behind fails this too)
            assert np.array_equal(g == 0, want == 0)
        if R * C * ph * pw >= 200_000:    # the launcher takes its tiled form (no con_idx read): += on every call
            gin = torch.zeros(f.shape, device="cuda")
            dummy = torch.empty(1, device="cuda")
            for k in (1, 2):
                assert ext._lib.RROIAlignBackwardLaucher(G.data_ptr(), 0.25, B, R, H, W, C, ph, pw, Rr.data_ptr(),
                                                         gin.data_ptr(), dummy.data_ptr(), dummy.data_ptr(), stream) == 1
                assert np.abs(gin.cpu().numpy() - k * want).max() <= 3 * BWD_RTOL * scale, (R, C, k)


def test_backward_relayout_leaves_out_dead_bins_only(ext, oracle):
    """Round 4: the relayout of top_diff neither reads nor writes a bin that enters no pixel's list (all four taps fail
    kernel.cu:267-274: the parts of a ROI that hang over the map's edge, masked columns, foreign batch indices).  The
    workspace is filled with NaN bit patterns first: a single list entry that named a left-out bin would put a NaN
    into the gradient.  ROIs centred on and beyond every edge and corner, every angle, on all gather paths."""
    rng = np.random.default_rng(123)
    B, C, H, W, ph, pw = 2, 96, 37, 52, 8, 40
    R = 160
    rois = np.zeros((R, 6), np.float32)
    rois[:, 0] = rng.integers(-1, B + 1, R)                       # some foreign batch indices
    edge = rng.integers(0, 4, R)
    rois[:, 1] = np.where(edge == 0, rng.uniform(-30, 10, R), np.where(edge == 1, rng.uniform(4 * W - 10, 4 * W + 30, R),
                                                                    rng.uniform(0, 4 * W, R)))
    rois[:, 2] = np.where(edge == 2, rng.uniform(-30, 10, R), np.where(edge == 3, rng.uniform(4 * H - 10, 4 * H + 30, R),
                                                                    rng.uniform(0, 4 * H, R)))
    rois[:, 3] = rng.uniform(8, 48, R)
    rois[:, 4] = rois[:, 3] * rng.uniform(1, 6, R)
    rois[:, 5] = rng.uniform(-180, 180, R)
    gout = rng.standard_normal((R, C, ph, pw)).astype(np.float32)
    ok = (rois[:, 0] >= 0) & (rois[:, 0] < B)     # (the reference, and with it the oracle, reads out of bounds for the others;
    want = oracle.backward_c(gout[ok], rois[ok], (B, C, H, W), 0.25)      # the product gives them no gradient)
    scale = max(1.0, float(np.abs(want).max()))
    G, Rr = dev(gout), dev(rois)
    nb = ext._lib.rroi_align_backward_workspace_bytes(B, C, H, W, R, ph, pw)
    stream = torch.cuda.current_stream().cuda_stream
    for p in (ext.PATH_TILED_LISTS, ext.PATH_TILED_BUCKETS, ext.PATH_TILED_INKERNEL):
        ws = torch.full((nb,), 0xFF, dtype=torch.uint8, device="cuda")
        got = torch.full((B, C, H, W), float("nan"), device="cuda")
        assert ext._lib.rroi_align_backward_hip(G.data_ptr(), 0.25, B, R, H, W, C, ph, pw, Rr.data_ptr(), got.data_ptr(),
                                                ws.data_ptr(), nb, p, stream) == 1
        g = got.cpu().numpy()
        assert np.isfinite(g).all(), p
        assert np.abs(g - want).max() <= BWD_RTOL * scale, p


def test_backward_many_rois_on_one_pixel(ext, oracle):
    """200 identical ROIs: every touched pixel's list holds 200 x its pairs (long lists, the
    counters' hot spots) and untouched pixels must come out exactly zero."""
    f, r = Wk.bench_inputs(R=1, C=16, H=48, W=48, img=192, seed=43)
    r = np.repeat(r, 200, axis=0)
    gout = np.random.default_rng(43).standard_normal((200, 16, 8, 64)).astype(np.float32)
    gwant = oracle.backward_c(gout, r, f.shape, 0.25)
    for p in (ext.PATH_TILED_LISTS, ext.PATH_TILED_BUCKETS, ext.PATH_TILED_INKERNEL, ext.PATH_TILED_ATOMIC):
        g = ext.backward(dev(gout), dev(r), f.shape, 0.25, path=p).cpu().numpy()
        assert np.abs(g - gwant).max() <= BWD_RTOL * max(1.0, float(np.abs(gwant).max()))
        assert np.array_equal(g == 0, gwant == 0)


@pytest.mark.parametrize("R,C,ph,pw,B", [(300, 64, 11, 83, 1), (300, 64, 11, 100, 2), (700, 40, 7, 50, 1),
                                         (260, 96, 11, 85, 3), (2100, 32, 3, 21, 1)])
def test_forward_rows_that_are_not_whole_sectors(ext, oracle, R, C, ph, pw, B):
    """Crops whose rows are not multiples of 64 bytes (PH * PW % 16 != 0), with enough ROIs for the
    tiled path's SHIFT form (sector-aligned store windows over overlapped tiles): bit-exact like
    every other shape, also into a buffer that starts 4 bytes off a 16-byte boundary, and nothing written outside
    the crops."""
    f, r = Wk.bench_inputs(R=R, C=C, H=60, W=90, img=360, seed=R + pw, batch=B)
    r[7, 3] = 0.0               # a degenerate ROI
    ro = r.copy()
    r[5, 0] = float(B)          # an image index out of range: zeros (the oracle gets a valid one and a zeroed row)
    ro[5, 0] = 0.0
    want = oracle.forward_c(f, ro, ph, pw, 0.25, threads=8)
    want[5] = 0.0
    got = ext.forward(dev(f), dev(r), ph, pw, 0.25, path=ext.PATH_TILED).cpu().numpy()
    assert mismatch(got, want)[0] == 0
    n = want.size
    buf = torch.full((n + 9,), float("nan"), device="cuda")
    out = buf[1:1 + n].view(R, C, ph, pw)
    nb = ext._lib.rroi_align_forward_workspace_bytes(B, C, 60, 90, R, ext.LAYOUT_NCHW)
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device="cuda")
    F, Rt = dev(f), dev(r)
    assert ext._lib.rroi_align_forward_hip(F.data_ptr(), ext.LAYOUT_NCHW, 0.25, B, R, 60, 90, C, ph, pw, Rt.data_ptr(),
                                           out.data_ptr(), ws.data_ptr(), nb, ext.PATH_TILED,
                                           torch.cuda.current_stream().cuda_stream) == 1
    host = buf.cpu().numpy()
    assert mismatch(host[1:1 + n].reshape(want.shape), want)[0] == 0
    assert np.isnan(host[0]) and np.isnan(host[1 + n:]).all()


@pytest.mark.parametrize("name", ["mid_c64", "batch3", "train_11xceil", "c70_odd"])
def test_forward_channels_last_output(ext, oracle, name):
    """Crops written straight into channels_last storage: same values, element for element."""
    f, r, ph, pw, s = SHAPES[name]()
    C = f.shape[1]
    if C % 4:
        with pytest.raises(ValueError):
            ext.forward(dev(f), dev(r), ph, pw, s, channels_last_out=True)
        return
    want = oracle.forward_c(f, r, ph, pw, s, threads=8)
    for F in (dev(f), dev(f).contiguous(memory_format=torch.channels_last)):
        got = ext.forward(F, dev(r), ph, pw, s, channels_last_out=True)
        assert got.is_contiguous(memory_format=torch.channels_last) and got.shape == want.shape
        assert mismatch(got.cpu().numpy(), want)[0] == 0
    with pytest.raises(ValueError):
        ext.forward(dev(f), dev(r), ph, pw, s, path=ext.PATH_DIRECT, channels_last_out=True)


def test_calls_are_graph_capturable(ext):
    """No allocation, no synchronisation, no host read-back inside the library: forward and
    backward can be captured into a HIP graph and replayed."""
    f, r = Wk.bench_inputs(R=40, C=64, seed=23)
    F, R = dev(f), dev(r)
    B, C, H, W = f.shape
    out = torch.empty((40, C, 8, 64), device="cuda")
    gin = torch.empty(f.shape, device="cuda")
    nf = ext._lib.rroi_align_forward_workspace_bytes(B, C, H, W, 40, 0)
    nb = ext._lib.rroi_align_backward_workspace_bytes(B, C, H, W, 40, 8, 64)
    wf = torch.empty(nf, dtype=torch.uint8, device="cuda")
    wb = torch.empty(nb, dtype=torch.uint8, device="cuda")

    def calls():
        st = torch.cuda.current_stream().cuda_stream
        assert ext._lib.rroi_align_forward_hip(F.data_ptr(), 0, 0.25, B, 40, H, W, C, 8, 64, R.data_ptr(),
                                               out.data_ptr(), wf.data_ptr(), nf, ext.PATH_TILED, st) == 1
        assert ext._lib.rroi_align_backward_hip(out.data_ptr(), 0.25, B, 40, H, W, C, 8, 64, R.data_ptr(),
                                                gin.data_ptr(), wb.data_ptr(), nb, ext.PATH_TILED, st) == 1
    calls()
    torch.cuda.synchronize()
    want_out, want_gin = out.clone(), gin.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        calls()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        calls()
    out.zero_()
    gin.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want_out)
    assert float((gin - want_gin).abs().max()) <= BWD_RTOL * max(1.0, float(want_gin.abs().max()))


def test_channels_last_pipeline_through_autograd(ext, oracle):
    """channels_last features -> channels_last crops -> channels_last gradient, through the module:
    values and gradients equal the NCHW pipeline's (forward bit for bit)."""
    from rroi_align.modules.rroi_align import _RRoiAlign
    f, r, ph, pw, s = SHAPES["mid_c64"]()
    want = oracle.forward_c(f, r, ph, pw, s, threads=8)
    gwant = oracle.backward_c((2 * want).astype(np.float32), r, f.shape, s)
    feats = dev(f).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    pooled = _RRoiAlign(ph, pw, s, channels_last_out=True)(feats, dev(r))
    assert pooled.is_contiguous(memory_format=torch.channels_last)
    assert mismatch(pooled.detach().cpu().numpy(), want)[0] == 0
    pooled.pow(2).sum().backward()          # grad_output = 2 * pooled, channels_last
    assert feats.grad.is_contiguous(memory_format=torch.channels_last)   # written in place, no relayout
    got = feats.grad.cpu().numpy()
    assert np.abs(got - gwant).max() <= BWD_RTOL * max(1.0, float(np.abs(gwant).max()))
    # every combination of gradient layouts through the native entry
    gout = (2 * want).astype(np.float32)
    for G in (dev(gout), dev(gout).contiguous(memory_format=torch.channels_last)):
        for cl in (False, True):
            for p in (ext.PATH_AUTO, ext.PATH_TILED_LISTS, ext.PATH_TILED_BUCKETS, ext.PATH_TILED_INKERNEL):
                g = ext.backward(G, dev(r), f.shape, s, path=p, channels_last_grad=cl)
                assert g.is_contiguous(memory_format=torch.channels_last) == cl or g.is_contiguous()
                assert np.abs(g.cpu().numpy() - gwant).max() <= BWD_RTOL * max(1.0, float(np.abs(gwant).max()))


@pytest.mark.parametrize("name", ["mid_c64", "batch3", "train_11xceil"])
def test_mixed_layout_handoff(ext, oracle, name):
    """the hand-off the reference's pipeline can reach -- NCHW features and NCHW feature gradient
    (what its backbone emits and expects, tools/models.py:387-457) with channels-last crops and top_diff.  Through the
    C-ABI (rroi_align_forward_layout_hip(NCHW, NHWC), rroi_align_backward_layout_hip(NHWC, NCHW)) and through the
    module with a channels-last consumer behind it: crops bit-exact against the oracle, gradient max-abs <= 1e-4."""
    from rroi_align.modules.rroi_align import _RRoiAlign
    f, r, ph, pw, s = SHAPES[name]()
    B, C, H, W = f.shape
    R = len(r)
    want = oracle.forward_c(f, r, ph, pw, s, threads=8)
    gout = np.random.default_rng(4).standard_normal(want.shape).astype(np.float32)
    gwant = oracle.backward_c(gout, r, f.shape, s)
    stream = torch.cuda.current_stream().cuda_stream
    F, Rr = dev(f), dev(r)
    out = torch.empty((R, C, ph, pw), device="cuda").contiguous(memory_format=torch.channels_last)
    G = dev(gout).contiguous(memory_format=torch.channels_last)
    gin = torch.full(f.shape, float("nan"), device="cuda")
    nf = ext._lib.rroi_align_forward_workspace_bytes(B, C, H, W, R, ext.LAYOUT_NCHW)
    nb = ext._lib.rroi_align_backward_workspace_bytes(B, C, H, W, R, ph, pw)
    ws = torch.empty(max(nf, nb), dtype=torch.uint8, device="cuda")
    for path in (ext.PATH_AUTO, ext.PATH_TILED):
        out.fill_(float("nan"))
        assert ext._lib.rroi_align_forward_layout_hip(F.data_ptr(), ext.LAYOUT_NCHW, ext.LAYOUT_NHWC, s, B, R, H, W, C, ph, pw,
                                                      Rr.data_ptr(), out.data_ptr(), ws.data_ptr(), nf, path, stream) == 1
        assert out.is_contiguous(memory_format=torch.channels_last) and eq(out.cpu().numpy(), want)
    for path in (ext.PATH_AUTO, ext.PATH_TILED, ext.PATH_TILED_LISTS, ext.PATH_TILED_BUCKETS, ext.PATH_TILED_INKERNEL):
        gin.fill_(float("nan"))
        assert ext._lib.rroi_align_backward_layout_hip(G.data_ptr(), ext.LAYOUT_NHWC, ext.LAYOUT_NCHW, s, B, R, H, W, C, ph, pw,
                                                       Rr.data_ptr(), gin.data_ptr(), ws.data_ptr(), nb, path, stream) == 1
        Wk.check_backward(gin.cpu().numpy(), gwant, f"mixed {name} path {path}")
    # the module: NCHW features, channels_last_out=True, a consumer whose gradient arrives channels-last
    feats = dev(f).requires_grad_(True)
    crops = _RRoiAlign(ph, pw, s, channels_last_out=True)(feats, Rr)
    assert crops.is_contiguous(memory_format=torch.channels_last) and eq(crops.detach().cpu().numpy(), want)
    (crops * G).sum().backward()              # d/d crops = G, handed over in G's channels-last storage
    assert feats.grad.is_contiguous()         # NCHW, the features' own layout
    Wk.check_backward(feats.grad.cpu().numpy(), gwant, f"mixed {name} autograd")


@pytest.mark.parametrize("name", ["mid_c64", "batch3", "train_11xceil"])
def test_backward_channels_last_grad(ext, oracle, name):
    """grad_output in channels_last storage (a channels_last recognition head) is consumed in place."""
    f, r, ph, pw, s = SHAPES[name]()
    gout = np.random.default_rng(3).standard_normal((len(r), f.shape[1], ph, pw)).astype(np.float32)
    want = oracle.backward_c(gout, r, f.shape, s)
    G = dev(gout).contiguous(memory_format=torch.channels_last)
    assert not G.is_contiguous()
    for p in (ext.PATH_AUTO, ext.PATH_TILED, ext.PATH_TILED_LISTS, ext.PATH_TILED_BUCKETS, ext.PATH_TILED_INKERNEL):
        got = ext.backward(G, dev(r), f.shape, s, path=p).cpu().numpy()
        assert np.abs(got - want).max() <= BWD_RTOL * max(1.0, float(np.abs(want).max()))
    # paths that need NCHW fall back to a contiguous copy
    got = ext.backward(G, dev(r), f.shape, s, path=ext.PATH_DIRECT).cpu().numpy()
    assert np.abs(got - want).max() <= BWD_RTOL * max(1.0, float(np.abs(want).max()))
    # the C-ABI refuses the combinations it cannot serve
    R, C = gout.shape[:2]
    B, _, H, W = f.shape
    nb = ext._lib.rroi_align_backward_workspace_bytes(B, C, H, W, R, ph, pw)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    gin = torch.empty(f.shape, device="cuda")
    st = ext._lib.rroi_align_backward_layout_hip(G.data_ptr(), ext.LAYOUT_NHWC, ext.LAYOUT_NCHW, s, B, R, H, W, C, ph, pw,
                                                 dev(r).data_ptr(), gin.data_ptr(), ws.data_ptr(), nb,
                                                 ext.PATH_DIRECT, torch.cuda.current_stream().cuda_stream)
    assert st == 0


def test_backward_nonfinite_gradients(ext, oracle):
    """The reference sends w*g AND 0*g to a pixel that two taps of a bin alias (kernel.cu:260-274):
    an infinite g there makes the pixel NaN, not inf.  Same set of non-finite pixels on every path;
    the finite ones within tolerance."""
    f, r = Wk.bench_inputs(R=24, C=8, seed=17)
    rng = np.random.default_rng(17)
    gout = rng.standard_normal((24, 8, 8, 64), dtype=np.float32)
    sel = rng.random(gout.shape)
    gout[sel < 0.002] = np.inf
    gout[(sel >= 0.002) & (sel < 0.003)] = np.nan
    want = oracle.backward_c(gout, r, f.shape, 0.25)
    assert np.isnan(want).any()
    fin = np.isfinite(want)
    for p in (ext.PATH_DIRECT, ext.PATH_TILED, ext.PATH_TILED_LISTS, ext.PATH_TILED_BUCKETS, ext.PATH_TILED_INKERNEL, ext.PATH_TILED_ATOMIC):
        got = ext.backward(dev(gout), dev(r), f.shape, 0.25, path=p).cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(want)), f"path {p}"
        assert np.array_equal(np.isposinf(got), np.isposinf(want)) and np.array_equal(np.isneginf(got), np.isneginf(want))
        assert np.abs(got[fin] - want[fin]).max() <= BWD_RTOL * max(1.0, float(np.abs(want[fin]).max()))


def test_autograd_surface(ext, oracle):
    """The reference's demo flow (rroi_align/test2.py:70-75): module call,
    pooled.pow(2).sum().backward(), grad on the feature map only."""
    from rroi_align.modules.rroi_align import _RRoiAlign
    f, r = Wk.bench_inputs(R=24, C=16, seed=13)
    feats = dev(f).requires_grad_(True)
    rois = dev(r)
    pooled = _RRoiAlign(8, 64, 0.25)(feats, rois.view(-1, 6))
    assert pooled.shape == (24, 16, 8, 64) and pooled.dtype == torch.float32
    pooled.pow(2).sum().backward()
    out = oracle.forward_c(f, r, 8, 64, 0.25)
    assert eq(pooled.detach().cpu().numpy(), out)
    want = oracle.backward_c((2 * out).astype(np.float32), r, f.shape, 0.25)
    got = feats.grad.cpu().numpy()
    assert np.abs(got - want).max() <= BWD_RTOL * max(1.0, float(np.abs(want).max()))
    assert rois.grad is None


def test_legacy_function_methods(ext, oracle):
    from rroi_align.functions.rroi_align import RRoiAlignFunction
    f, r = Wk.cfg1_inputs()
    fn = RRoiAlignFunction(8, 32, 1.0)
    out = fn.forward(dev(f), dev(r))
    assert tuple(fn.feature_size) == f.shape
    gin, none = fn.backward(out * 2)
    assert none is None and gin.shape == f.shape
    want = oracle.backward_c((2 * oracle.forward_c(f, r, 8, 32, 1.0)).astype(np.float32), r, f.shape, 1.0)
    assert np.abs(gin.cpu().numpy() - want).max() <= BWD_RTOL * max(1.0, float(np.abs(want).max()))


def test_reference_ffi_names(ext, oracle):
    """_ext.rroi_align.rroi_align_forward_cuda / _backward_cuda with the reference's
    argument order (src/rroi_align_cuda.h:1-7), including idx_x / idx_y."""
    f, r = Wk.bench_inputs(R=12, C=6, seed=14)
    F, Rr = dev(f), dev(r)
    shape = (12, 6, 8, 64)
    out, ix, iy = (torch.full(shape, 7.0, device="cuda") for _ in range(3))  # not pre-zeroed on purpose
    assert ext.rroi_align_forward_cuda(8, 64, 0.25, F, Rr, out, ix, iy) == 1
    want, wx, wy = oracle.forward_literal_c(f, r, 8, 64, 0.25)
    assert eq(out.cpu().numpy(), want) and eq(ix.cpu().numpy(), wx) and eq(iy.cpu().numpy(), wy)
    gin = torch.zeros(f.shape, device="cuda")
    assert ext.rroi_align_backward_cuda(8, 64, 0.25, out * 2, Rr, gin, ix, iy) == 1
    wb = oracle.backward_literal_c((2 * want).astype(np.float32), r, wx, wy, f.shape, 0.25)
    assert np.abs(gin.cpu().numpy() - wb).max() <= BWD_RTOL * max(1.0, float(np.abs(wb).max()))
    assert ext.rroi_align_forward_cuda(8, 64, 0.25, F, Rr[:, :5].contiguous(), out, ix, iy) == 0


def test_reference_launchers_take_the_fast_paths(ext, oracle):
    """behind the reference's own symbols (`RROIAlignForwardLaucher` /
    `RROIAlignBackwardLaucher`, rroi_align_kernel.h:8-18) a large problem runs the tiled forward and
    the gather backward on stream-ordered scratch; ROIs of images >= 1 (the signature carries no
    batch count) and con_idx_x / con_idx_y are still produced.  Same results as the literal forms."""
    f, r = Wk.bench_inputs(R=96, C=64, seed=21, batch=2)   # 3.1 M output elements: above both crossovers
    assert set(r[:, 0]) == {0.0, 1.0}
    F, Rr = dev(f), dev(r)
    shape = (96, 64, 8, 64)
    out, ix, iy = (torch.full(shape, 7.0, device="cuda") for _ in range(3))
    assert ext.rroi_align_forward_cuda(8, 64, 0.25, F, Rr, out, ix, iy) == 1
    want, wx, wy = oracle.forward_literal_c(f, r, 8, 64, 0.25)
    assert eq(out.cpu().numpy(), want) and eq(ix.cpu().numpy(), wx) and eq(iy.cpu().numpy(), wy)
    gin = torch.zeros(f.shape, device="cuda")
    assert ext.rroi_align_backward_cuda(8, 64, 0.25, out * 2, Rr, gin, ix, iy) == 1
    wb = oracle.backward_literal_c((2 * want).astype(np.float32), r, wx, wy, f.shape, 0.25)
    assert np.abs(gin.cpu().numpy() - wb).max() <= BWD_RTOL * max(1.0, float(np.abs(wb).max()))
    # the launcher itself without con_idx (NULL is allowed): the whole call is the two tiled launches
    st = ext._lib.RROIAlignForwardLaucher(F.data_ptr(), 0.25, 96, 160, 160, 64, 8, 64, Rr.data_ptr(),
                                          out.fill_(3.0).data_ptr(), None, None, torch.cuda.current_stream().cuda_stream)
    assert st == 1 and eq(out.cpu().numpy(), want)


def test_reference_launchers_reuse_their_scratch_and_accumulate(ext, oracle):
    """VERDICT r02 #6 / ADVICE r02: the launchers keep one scratch buffer per (device, stream) inside the
    library -- no hipMallocAsync / hipFreeAsync per call, so a call whose buffer exists can be captured
    into a HIP graph -- and `RROIAlignBackwardLaucher` ADDS into bottom_diff on every path, like the
    reference's atomicAdds (kernel.cu:260-274): the result no longer depends on the problem size."""
    stream = lambda: torch.cuda.current_stream().cuda_stream
    for R, C in ((96, 64), (6, 8)):     # above the tiled crossovers / below them (literal kernels)
        f, r = Wk.bench_inputs(R=R, C=C, seed=31)
        F, Rr = dev(f), dev(r)
        shape = (R, C, 8, 64)
        out, ix, iy = (torch.empty(shape, device="cuda") for _ in range(3))
        assert ext.rroi_align_forward_cuda(8, 64, 0.25, F, Rr, out, ix, iy) == 1
        want, wx, wy = oracle.forward_literal_c(f, r, 8, 64, 0.25)
        assert eq(out.cpu().numpy(), want)
        gout = torch.randn(shape, device="cuda")
        wb = oracle.backward_literal_c(gout.cpu().numpy(), r, wx, wy, f.shape, 0.25)
        scale = max(1.0, float(np.abs(wb).max()))
        gin = torch.zeros(f.shape, device="cuda")
        for k in (1, 2, 3):             # three calls onto the same buffer: k times the gradient
            assert ext.rroi_align_backward_cuda(8, 64, 0.25, gout, Rr, gin, ix, iy) == 1
            assert np.abs(gin.cpu().numpy() - k * wb).max() <= 3 * BWD_RTOL * scale, (R, k)
    # capture: the buffers of this stream exist after a first call on it, so the calls enqueue kernels only
    f, r = Wk.bench_inputs(R=96, C=64, seed=32)
    F, Rr = dev(f), dev(r)
    out = torch.empty((96, 64, 8, 64), device="cuda")
    gin = torch.zeros(f.shape, device="cuda")
    gout = torch.randn_like(out)

    def calls():
        assert ext._lib.RROIAlignForwardLaucher(F.data_ptr(), 0.25, 96, 160, 160, 64, 8, 64, Rr.data_ptr(),
                                                out.data_ptr(), None, None, stream()) == 1
        assert ext._lib.RROIAlignBackwardLaucher(gout.data_ptr(), 0.25, 1, 96, 160, 160, 64, 8, 64, Rr.data_ptr(),
                                                 gin.data_ptr(), out.data_ptr(), out.data_ptr(), stream()) == 1
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        calls()                          # creates the side stream's scratch
        side.synchronize()
        want_out, g1 = out.clone(), gin.clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            calls()
    torch.cuda.current_stream().wait_stream(side)
    out.zero_()
    gin.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want_out)
    assert float((gin - g1).abs().max()) <= BWD_RTOL * max(1.0, float(g1.abs().max()))
    # the buffer the capture was handed is PINNED.  A later, larger call on the same stream (rounds 2-3
    # freed the small buffer in stream order to grow it), the release call and more streams than the table holds
    # all leave it alone: the graph replays correctly afterwards
    f2, r2 = Wk.bench_inputs(R=384, C=64, seed=33)
    F2, R2 = dev(f2), dev(r2)
    big = torch.empty((384, 64, 8, 64), device="cuda")
    with torch.cuda.stream(side):
        for _ in range(2):
            assert ext._lib.RROIAlignForwardLaucher(F2.data_ptr(), 0.25, 384, 160, 160, 64, 8, 64, R2.data_ptr(),
                                                    big.data_ptr(), None, None, side.cuda_stream) == 1
        side.synchronize()
    assert eq(big.cpu().numpy(), oracle.forward_c(f2, r2, 8, 64, 0.25))
    others = [torch.cuda.Stream() for _ in range(ext.launcher_scratch_stats()["capacity"] + 4)]   # more streams than the library keeps buffers for
    for s_ in others:
        with torch.cuda.stream(s_):
            assert ext._lib.RROIAlignForwardLaucher(F.data_ptr(), 0.25, 96, 160, 160, 64, 8, 64, Rr.data_ptr(),
                                                    big.data_ptr(), None, None, s_.cuda_stream) == 1
    torch.cuda.synchronize()
    ext.release_workspaces()
    scribble = [torch.full((64 << 18,), float("nan"), device="cuda") for _ in range(4)]   # reuse what was freed
    out.zero_()
    gin.zero_()
    g.replay()
    torch.cuda.synchronize()
    del scribble
    assert torch.equal(out, want_out)
    assert float((gin - g1).abs().max()) <= BWD_RTOL * max(1.0, float(g1.abs().max()))
    ext.release_workspaces()            # frees the library's unpinned buffers too; the next call re-creates them
    assert ext._lib.RROIAlignForwardLaucher(F.data_ptr(), 0.25, 96, 160, 160, 64, 8, 64, Rr.data_ptr(),
                                            out.data_ptr(), None, None, stream()) == 1
    torch.cuda.synchronize()
    assert torch.equal(out, want_out)


def test_launcher_scratch_is_reused_within_one_capture(ext, oracle):
    """a buffer handed out during a capture is pinned (graph-exclusive) -- but further launcher calls inside the
    SAME capture reuse it (same capture id) instead of taking hipMallocAsync / hipFreeAsync nodes each; the deprecated
    per-device trig setter is a shim that refuses what it can no longer do."""
    f, r = Wk.bench_inputs(R=160, C=64, seed=35)
    F, Rr = dev(f), dev(r)
    outs = [torch.empty((160, 64, 8, 64), device="cuda") for _ in range(3)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())

    def call(o):
        assert ext._lib.RROIAlignForwardLaucher(F.data_ptr(), 0.25, 160, 160, 160, 64, 8, 64, Rr.data_ptr(), o.data_ptr(), None, None,
                                                side.cuda_stream) == 1
    # (160 ROIs: above the forward's two-launch crossover of 3.8 M output elements -- fewer take the one-launch patch kernel, no scratch)
    with torch.cuda.stream(side):
        call(outs[0])                    # creates the side stream's scratch
        side.synchronize()
        before = ext.launcher_scratch_stats()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for o in outs:
                call(o)
        after = ext.launcher_scratch_stats()
    torch.cuda.current_stream().wait_stream(side)
    assert after["pinned"] == before["pinned"] + 1 and after["transient_calls"] == before["transient_calls"], (before, after)
    for o in outs:
        o.zero_()
    g.replay()
    torch.cuda.synchronize()
    want = oracle.forward_c(f, r, 8, 64, 0.25, threads=8)
    for o in outs:
        assert eq(o.cpu().numpy(), want)
    assert ext._lib.rroi_align_set_trig_recipe_hip(ext.TRIG_DOUBLE) == 1
    assert ext._lib.rroi_align_set_trig_recipe_hip(ext.TRIG_FP32) == 0 and ext._lib.rroi_align_get_trig_recipe_hip() == ext.TRIG_DOUBLE


def test_reference_launchers_from_several_threads(ext, oracle):
    """the launchers' scratch table has one lock per (device, stream) entry (held until the caller has enqueued
    its launches) and a table lock for look-up only: threads on DIFFERENT streams run concurrently, threads SHARING a
    stream serialise on its entry -- either way every call gets its own correct crops, through growth of the buffers (the
    ROI count alternates) and more streams than the table has entries."""
    import threading
    f, r = Wk.bench_inputs(R=160, C=64, seed=41)
    F = dev(f)
    sizes = (96, 160, 128)
    want = {n: oracle.forward_c(f, r[:n], 8, 64, 0.25, threads=8) for n in sizes}
    streams = [torch.cuda.Stream() for _ in range(6)]
    errors = []

    def worker(tid):
        try:
            st_ = streams[tid % len(streams)] if tid < 12 else streams[0]      # the last four threads share one stream
            torch.cuda.set_device(0)
            with torch.cuda.stream(st_):
                for it in range(12):
                    n = sizes[(tid + it) % len(sizes)]
                    Rr = dev(r[:n])
                    out = torch.empty((n, 64, 8, 64), device="cuda")
                    assert ext._lib.RROIAlignForwardLaucher(F.data_ptr(), 0.25, n, 160, 160, 64, 8, 64, Rr.data_ptr(),
                                                            out.data_ptr(), None, None, st_.cuda_stream) == 1
                    st_.synchronize()
                    if not eq(out.cpu().numpy(), want[n]):
                        errors.append((tid, it, n))
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))
    torch.cuda.synchronize()
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(16)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]
    ext.release_workspaces()


def test_channels_last_odd_chunk(ext, oracle):
    """C % 32 != 0 with channels-last features consumed in place -- the lanes of the
    channel quads beyond C must fetch nothing (their offset is an out-of-range sentinel that may not
    wrap when it meets an invalid tap's sentinel)."""
    for C in (36, 100):
        f, r = Wk.bench_inputs(R=24, C=C, seed=31 + C, batch=2)
        r = np.concatenate([r, Wk.edge_rois()[:12]], 0)
        want = oracle.forward_c(f, r, 8, 64, 0.25, threads=8)
        cl = dev(f).contiguous(memory_format=torch.channels_last)
        assert eq(ext.forward(cl, dev(r), 8, 64, 0.25, path=ext.PATH_TILED).cpu().numpy(), want)
        assert eq(ext.forward(dev(f), dev(r), 8, 64, 0.25, path=ext.PATH_TILED).cpu().numpy(), want)


def test_channels_last_is_consumed_in_place(ext, oracle):
    f, r = Wk.bench_inputs(R=20, C=64, seed=15, batch=2)
    want = oracle.forward_c(f, r, 8, 64, 0.25, threads=8)
    cl = dev(f).contiguous(memory_format=torch.channels_last)
    assert not cl.is_contiguous()
    assert eq(ext.forward(cl, dev(r), 8, 64, 0.25).cpu().numpy(), want)
    f3, r3 = Wk.bench_inputs(R=6, C=3, H=64, W=96, img=96, seed=16)
    cl3 = dev(f3).contiguous(memory_format=torch.channels_last)
    assert eq(ext.forward(cl3, dev(r3), 32, 100, 1.0, path=ext.PATH_TILED).cpu().numpy(),
              oracle.forward_c(f3, r3, 32, 100, 1.0))


def test_other_stream_and_reentrancy(ext, oracle):
    f, r = Wk.bench_inputs(R=32, C=32, seed=17)
    want = oracle.forward_c(f, r, 8, 64, 0.25, threads=8)
    F, Rr = dev(f), dev(r)
    torch.cuda.synchronize()
    s1 = torch.cuda.Stream()
    with torch.cuda.stream(s1):
        outs = [ext.forward(F, Rr, 8, 64, 0.25, path=ext.PATH_TILED) for _ in range(4)]
    s1.synchronize()
    for o in outs:
        assert eq(o.cpu().numpy(), want)


def test_two_calls_in_flight_on_two_streams(ext, oracle):
    """Round 6 (`extra.two_calls_in_flight`): consecutive calls alternate between two streams and run side by side on the
    device -- different ROIs per stream, so that a workspace (chunk-major copy, affine table, ROI order) shared between
    two calls in flight would show; forward through the two-launch path and the backward's list path, each against the
    oracle."""
    f, r0 = Wk.bench_inputs(R=96, C=64, H=64, W=96, img=384, seed=31)
    _, r1 = Wk.bench_inputs(R=96, C=64, H=64, W=96, img=384, seed=32)
    want = [oracle.forward_c(f, r, 8, 64, 0.25, threads=8) for r in (r0, r1)]
    g = np.random.default_rng(3).standard_normal(want[0].shape).astype(np.float32)
    gwant = [oracle.backward_c(g, r, f.shape, 0.25, threads=8) for r in (r0, r1)]
    F, G, Rr = dev(f), dev(g), [dev(r0), dev(r1)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs, grads = [], []
    for i in range(8):
        with torch.cuda.stream(streams[i % 2]):
            outs.append(ext.forward(F, Rr[i % 2], 8, 64, 0.25, path=ext.PATH_TILED))
            grads.append(ext.backward(G, Rr[i % 2], f.shape, 0.25, path=ext.PATH_TILED))
    for s in streams:
        s.synchronize()
    for i, (o, gi) in enumerate(zip(outs, grads)):
        assert eq(o.cpu().numpy(), want[i % 2]), "forward of call %d" % i
        scale = max(1.0, float(np.abs(gwant[i % 2]).max()))
        assert float(np.abs(gi.cpu().numpy() - gwant[i % 2]).max()) <= 1e-4 * scale, "backward of call %d" % i


def test_invalid_batch_index_yields_zeros(ext):
    f, r = Wk.bench_inputs(R=4, C=8, seed=18)
    r[1, 0] = 5
    r[2, 0] = -3
    for p in (ext.PATH_DIRECT, ext.PATH_TILED, ext.PATH_FUSED):
        out = run_fwd(ext, f, r, 8, 64, 0.25, p)
        assert not out[1].any() and not out[2].any() and out[0].any() and out[3].any()


def test_input_validation(ext):
    F = torch.zeros(1, 4, 16, 16, device="cuda")
    R = torch.zeros(2, 6, device="cuda")
    with pytest.raises(ValueError):
        ext.forward(F, R[:, :5], 8, 8, 1.0)
    with pytest.raises(TypeError):
        ext.forward(F.double(), R, 8, 8, 1.0)
    with pytest.raises(ValueError):
        ext.forward(F[0], R, 8, 8, 1.0)
    assert ext.forward(F, R[:0], 8, 8, 1.0).shape == (0, 4, 8, 8)


def test_random_shape_sweep(ext, oracle):
    """Seeded sweep over the parameter regimes of SURVEY.md 3.5 and beyond: channel counts that
    are not multiples of 4 or 32, pooled sizes whose product is not a multiple of 4 or 64, maps
    with odd sizes, several images, scales 1 / 0.5 / 0.25 -- every path, bit-exact."""
    rng = np.random.default_rng(2024)
    for trial in range(36):
        C = int(rng.choice([1, 3, 4, 5, 31, 32, 33, 64, 70, 96]))
        H, W = int(rng.integers(9, 90)), int(rng.integers(9, 120))
        B = int(rng.integers(1, 4))
        ph = int(rng.choice([1, 2, 8, 11, 13, 32]))
        pw = int(rng.integers(1, 130))
        s = float(rng.choice([1.0, 0.5, 0.25]))
        R = int(rng.integers(1, 40))
        f, r = Wk.bench_inputs(R=R, C=C, H=H, W=W, img=int(W / s), seed=1000 + trial, batch=B)
        r[:, 2] = rng.uniform(0, H / s, R)  # cy in the image space of a non-square map
        r[:, 3] = rng.uniform(2, 40, R) / (s * 4)
        r[:, 4] = r[:, 3] * rng.uniform(0.3, 12, R)
        want = oracle.forward_c(f, r, ph, pw, s, threads=8)
        for p in (ext.PATH_DIRECT, ext.PATH_TILED, ext.PATH_FUSED):
            got = run_fwd(ext, f, r, ph, pw, s, p)
            n, d = mismatch(got, want)
            assert n == 0, f"trial {trial} C={C} {H}x{W} B={B} {ph}x{pw} s={s} R={R} path={p}: {n} differ (max {d})"
        gout = np.random.default_rng(trial).standard_normal(want.shape).astype(np.float32)
        gwant = oracle.backward_c(gout, r, f.shape, s)
        for p in (ext.PATH_DIRECT, ext.PATH_TILED, ext.PATH_TILED_LISTS, ext.PATH_TILED_BUCKETS, ext.PATH_TILED_INKERNEL, ext.PATH_TILED_ATOMIC):
            g = ext.backward(dev(gout), dev(r), f.shape, s, path=p).cpu().numpy()
            assert np.abs(g - gwant).max() <= BWD_RTOL * max(1.0, float(np.abs(gwant).max())), f"trial {trial} bwd path={p}"


def test_views_with_storage_offsets(ext, oracle):
    """Batch slices and odd-sized maps: the features pointer need not be 16-byte aligned and
    H*W need not be a multiple of 4 (the prologue then reads scalars)."""
    f, r = Wk.bench_inputs(R=12, C=41, H=33, W=45, img=180, seed=31, batch=3)
    F = dev(f)
    r1 = r.copy()
    r1[:, 0] = np.minimum(r1[:, 0], 1)
    want = oracle.forward_c(f[1:], r1, 8, 40, 0.25)
    sub = F[1:]                      # contiguous view, storage offset 41*33*45 floats: not a multiple of 16 B
    assert sub.is_contiguous() and sub.data_ptr() % 16 != 0
    for p in (ext.PATH_TILED, ext.PATH_DIRECT):
        assert eq(ext.forward(sub, dev(r1), 8, 40, 0.25, path=p).cpu().numpy(), want)
    # a strided channel slice is made contiguous by the Python surface
    want2 = oracle.forward_c(f[:, 8:24], r, 8, 40, 0.25)
    assert eq(ext.forward(F[:, 8:24], dev(r), 8, 40, 0.25).cpu().numpy(), want2)


def test_autograd_surface_under_autocast():
    """torch.autocast around the module: half features are cast up on the way in, the op computes in fp32 (what the
    reference is), the crops are the fp32 op's crops of the rounded features bit for bit, and the feature gradient
    comes back in the features' own dtype."""
    from rroi_align.modules.rroi_align import _RRoiAlign
    f, r = Wk.bench_inputs(R=24, C=32, H=40, W=60, img=240, seed=77)
    F16 = dev(f).half().requires_grad_(True)
    R = dev(r)
    op = _RRoiAlign(8, 32, 0.25)
    with torch.autocast("cuda", dtype=torch.float16):
        out = op(F16, R)
    assert out.dtype == torch.float32
    ref = op(F16.detach().float(), R)
    assert torch.equal(out, ref)
    out.pow(2).sum().backward()
    assert F16.grad is not None and F16.grad.dtype == torch.float16 and F16.grad.shape == F16.shape
    F32 = F16.detach().float().requires_grad_(True)
    op(F32, R).pow(2).sum().backward()
    assert torch.allclose(F16.grad.float(), F32.grad, rtol=2e-3, atol=1e-3 * float(F32.grad.abs().max()))


def test_reference_launcher_on_rows_that_are_not_whole_sectors(ext):
    """The reference-ABI forward launcher (two launches: prologue with the extra blocks for ROIs of images >= 1,
    then the gather) takes the SHIFT kernels like the native call: same crops bit for bit, con_idx filled."""
    f, r = Wk.bench_inputs(R=300, C=64, H=60, W=90, img=360, seed=91)
    F, R = dev(f), dev(r)
    want = ext.forward(F, R, 11, 83, 0.25, path=ext.PATH_TILED)
    out, ix, iy = (torch.full((300, 64, 11, 83), float("nan"), device="cuda") for _ in range(3))
    st = torch.cuda.current_stream().cuda_stream
    assert ext._lib.RROIAlignForwardLaucher(F.data_ptr(), 0.25, 300, 60, 90, 64, 11, 83, R.data_ptr(), out.data_ptr(),
                                            None, None, st) == 1
    assert torch.equal(out, want)
    assert ext.rroi_align_forward_cuda(11, 83, 0.25, F, R, out, ix, iy) == 1
    assert torch.equal(out, want) and not ix.isnan().any() and not iy.isnan().any()
    # two images: the launcher's signature has no batch count, so the ROIs of image 1 are sampled by the prologue's
    # extra blocks and SKIPPED by the gather -- also by its SHIFT form, whose windows reach over row ends
    f2, r2 = Wk.bench_inputs(R=300, C=64, H=60, W=90, img=360, seed=92, batch=2)
    F2, R2 = dev(f2), dev(r2)
    want2 = ext.forward(F2, R2, 11, 83, 0.25, path=ext.PATH_TILED)
    out2 = torch.full((300, 64, 11, 83), float("nan"), device="cuda")
    assert ext._lib.RROIAlignForwardLaucher(F2.data_ptr(), 0.25, 300, 60, 90, 64, 11, 83, R2.data_ptr(), out2.data_ptr(),
                                            None, None, st) == 1
    assert int((r2[:, 0] == 1).sum()) > 50 and torch.equal(out2, want2)
