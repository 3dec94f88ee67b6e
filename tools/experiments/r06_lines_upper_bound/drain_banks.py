#!/usr/bin/env python3
"""The SHIFT forms' drain reads the bin-major tile with lane = (row, piece), piece fastest (what the TCP coalesces): the pieces
of a row sit 4 bins = 128 floats apart -- the same LDS bank (ds_read_b32: (a / 4) mod 32) -- a 4-way (SHIFT == 1) / 8-way
(SHIFT == 2) conflict on every read.  What would conflict-free reads be worth?  drain_banks.patch (exploration build, wrong
results on purpose): dbg 16 moves the piece from the row bits into the bank bits of the address -- the same instructions,
no conflicts.  us per call: whole call / gather alone; dbg 0 shipped, 16 the bound, 3 / 19 the same without any memory access."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_forward_stages_hip.argtypes = [vp, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, it, vp]
lib.rroi_align_forward_workspace_bytes.restype = sz
lib.rroi_align_forward_workspace_bytes.argtypes = [it] * 6
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, warm, iters):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
# (B, C, H, W, R, ph, pw, merge): beyond the cache (SHIFT == 2), inside it (SHIFT == 1; merge = 0 forces it where the merging form ships)
SHAPES = ((1, 256, 160, 160, 600, 11, 100, 1), (2, 64, 120, 160, 2048, 11, 83, 1), (1, 256, 160, 160, 238, 11, 100, 1),
          (1, 256, 160, 160, 512, 7, 50, 1), (2, 64, 120, 160, 128, 11, 83, 1), (2, 64, 120, 160, 512, 11, 100, 0), (2, 64, 120, 160, 32, 11, 100, 1))
for (B, C, H, W, R, ph, pw, merge) in SHAPES:
    rng = np.random.default_rng(1000 + R + pw)
    F = torch.from_numpy(rng.standard_normal((B, C, H, W), dtype=np.float32)).cuda()
    h = rng.uniform(16, 64, R)
    Rt = torch.from_numpy(np.stack([rng.integers(0, B, R), rng.uniform(0, 4 * W, R), rng.uniform(0, 4 * H, R), h,
                                    h * rng.uniform(2, pw / float(ph), R), rng.uniform(-45, 45, R)], 1).astype(np.float32)).cuda()
    nb = lib.rroi_align_forward_workspace_bytes(B, C, H, W, R, 0)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    top = torch.empty((R, C, ph, pw), device="cuda")
    def call(stages):
        assert lib.rroi_align_forward_stages_hip(F.data_ptr(), 0, 0.25, B, R, H, W, C, ph, pw, Rt.data_ptr(), top.data_ptr(), ws.data_ptr(), nb, 2, stages, st) == 1
    crops = R * C * ph * pw * 4
    lib.rroi_align_debug_set_fwd_merge(merge)
    n = 50 if crops > (200 << 20) else 300
    for rep in range(2):
        row = []
        for dbg in (0, 16, 3, 19):
            lib.rroi_align_debug_set_fwd_dbg(dbg)
            call(3)
            whole, alone = timeit(lambda: call(3), n // 2, n), timeit(lambda: call(2), n // 2, n)
            row.append(f"dbg{dbg:2d}: {whole:6.1f} / {alone:6.1f}")
        lib.rroi_align_debug_set_fwd_dbg(0)
        print(f"B={B} C={C} {H}x{W} R={R} {ph}x{pw:3d} {crops / 1e6:6.0f} MB merge={merge}  whole / gather alone  " + "  ".join(row), flush=True)
    lib.rroi_align_debug_set_fwd_merge(1)
