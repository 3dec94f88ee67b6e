#!/usr/bin/env python3
"""(experiment) SHIFT == 2, line-aligned store windows (32 own bins of 64 gathered), against the shipped forms: random problems
compared bit for bit, then crops beyond the 256 MB memory-side cache timed (whole call (gather alone))."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_forward_stages_hip.argtypes = [vp, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, it, vp]
lib.rroi_align_forward_workspace_bytes.restype = sz
lib.rroi_align_forward_workspace_bytes.argtypes = [it] * 6
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, warm=15, iters=50):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def problem(rng, B, C, H, W, R, ph, pw, wild=False):
    F = torch.from_numpy(rng.standard_normal((B, C, H, W), dtype=np.float32)).cuda()
    h = rng.uniform(4, 64, R) if wild else rng.uniform(16, 64, R)
    Rt = torch.from_numpy(np.stack([rng.integers(0, B, R), rng.uniform(-8 if wild else 0, 4 * W + (8 if wild else 0), R),
                                    rng.uniform(-8 if wild else 0, 4 * H + (8 if wild else 0), R), h,
                                    h * rng.uniform(1 if wild else 2, 9 if wild else pw / float(ph), R),
                                    rng.uniform(-90 if wild else -45, 90 if wild else 45, R)], 1).astype(np.float32)).cuda()
    nb = lib.rroi_align_forward_workspace_bytes(B, C, H, W, R, 0)
    return F, Rt, nb, torch.empty(nb, dtype=torch.uint8, device="cuda")
rng = np.random.default_rng(9)
bad = 0
for trial in range(int(os.environ.get("TRIALS", 200))):
    B, C, H, W = int(rng.integers(1, 4)), int(rng.integers(1, 100)), int(rng.integers(6, 90)), int(rng.integers(6, 110))
    R, ph, pw = int(rng.integers(1, 200)), int(rng.integers(1, 13)), int(rng.integers(1, 130))
    F, Rt, nb, ws = problem(rng, B, C, H, W, R, ph, pw, wild=True)
    outs = []
    for lines in (0, 2):
        lib.rroi_align_debug_set_fwd_shift_lines(lines, 0)
        # an odd base address now and then: the windows follow the crops' own alignment
        pad = int(rng.integers(0, 32)) if lines == 2 else 0
        buf = torch.full((R * C * ph * pw + 32,), float("nan"), device="cuda")
        top = buf[pad:pad + R * C * ph * pw].view(R, C, ph, pw)
        assert lib.rroi_align_forward_stages_hip(F.data_ptr(), 0, 0.25, B, R, H, W, C, ph, pw, Rt.data_ptr(), top.data_ptr(), ws.data_ptr(), nb, 2, 3, st) == 1
        outs.append(top.clone())
        if lines == 2 and (torch.isnan(buf[:pad]).sum() != pad or not torch.isnan(buf[pad + R * C * ph * pw:]).all()):
            bad += 1; print("WROTE OUTSIDE", B, C, H, W, R, ph, pw, pad)
    if not torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32)):
        bad += 1; print("MISMATCH", B, C, H, W, R, ph, pw, int((outs[0].view(torch.int32) != outs[1].view(torch.int32)).sum()))
lib.rroi_align_debug_set_fwd_shift_lines(0, 0)
print(f"line-aligned windows vs shipped forms on random problems: {bad} bad", flush=True)
shapes = ((8, 64, 160, 160, 2048, 11, 100), (2, 64, 120, 160, 2048, 11, 83), (1, 256, 160, 160, 1024, 11, 50), (1, 256, 160, 160, 600, 11, 100),
          (2, 64, 120, 160, 4096, 11, 83), (2, 64, 120, 160, 512, 11, 83))
for (B, C, H, W, R, ph, pw) in shapes:
    F, Rt, nb, ws = problem(np.random.default_rng(1000 + R + pw), B, C, H, W, R, ph, pw)
    top = torch.empty((R, C, ph, pw), device="cuda")
    def call(stages):
        assert lib.rroi_align_forward_stages_hip(F.data_ptr(), 0, 0.25, B, R, H, W, C, ph, pw, Rt.data_ptr(), top.data_ptr(), ws.data_ptr(), nb, 2, stages, st) == 1
    row, ref = [], None
    for (lines, wpc) in ((0, 0), (2, 10), (2, 8), (2, 12), (0, 0), (2, 10)):
        lib.rroi_align_debug_set_fwd_shift_lines(lines, wpc)
        call(3)
        if ref is None: ref = top.clone()
        same = torch.equal(ref.view(torch.int32), top.view(torch.int32))
        row.append(f"{'shipped' if lines == 0 else 'lines/%d' % wpc}: {timeit(lambda: call(3)):6.1f} ({timeit(lambda: call(2)):6.1f}){'' if same else ' DIFFERENT'}")
    lib.rroi_align_debug_set_fwd_shift_lines(0, 10)
    print(f"B={B} C={C:3d} {H}x{W} R={R:4d} {ph}x{pw:3d} {R * C * ph * pw * 4 / 2**20:6.0f} MB  " + "  ".join(row), flush=True)
