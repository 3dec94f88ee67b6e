#!/usr/bin/env python3
"""Per-wave start / end time stamps (100 MHz) of rroi_fwd_patch_kernel at R = 32, C = 64 (temporary instrumentation)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_forward_hip.argtypes = [vp, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, vp]
lib.rroi_align_debug_set_trace_ptr.argtypes = [vp]
st = torch.cuda.current_stream().cuda_stream
R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B, C, H, W, ph, pw = 2, 64, 120, 160, 11, 96
rng = np.random.default_rng(1000 + R + pw)
F = torch.from_numpy(rng.standard_normal((B, C, H, W), dtype=np.float32)).cuda()
h = rng.uniform(16, 64, R)
rois = np.stack([rng.integers(0, B, R), rng.uniform(0, 4 * W, R), rng.uniform(0, 4 * H, R), h,
                 h * rng.uniform(2, pw / float(ph), R), rng.uniform(-45, 45, R)], 1).astype(np.float32)
Rt = torch.from_numpy(rois).cuda()
top = torch.empty((R, C, ph, pw), device="cuda")
trace = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
lib.rroi_align_debug_set_trace_ptr(trace.data_ptr())
def call():
    assert lib.rroi_align_forward_hip(F.data_ptr(), 0, 0.25, B, R, H, W, C, ph, pw, Rt.data_ptr(), top.data_ptr(), None, 0, 1 | (0x100 if os.environ.get("TRIG") == "fp32" else 0), st) == 1
for _ in range(50): call()
lib.rroi_align_debug_set_fwd_patch_ablate(8 | int(os.environ.get("ABLATE", "0")))
trace.zero_(); torch.cuda.synchronize()
for _ in range(3): call()
torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(-1, 2)
t = t[t[:, 0] != 0]
t0 = t[:, 0].min()
s, e = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0
print("waves", len(t), "start us: p10 %.2f p50 %.2f p90 %.2f max %.2f | end us: p10 %.2f p50 %.2f p90 %.2f max %.2f | life us: p50 %.2f p90 %.2f" % (
    *np.percentile(s, [10, 50, 90]), s.max(), *np.percentile(e, [10, 50, 90]), e.max(), *np.percentile(e - s, [50, 90])))
hist, edges = np.histogram(s, bins=np.arange(0, e.max() + 0.5, 0.5))
print("starts per 0.5 us:", hist.tolist())
hist, edges = np.histogram(e, bins=np.arange(0, e.max() + 0.5, 0.5))
print("ends   per 0.5 us:", hist.tolist())
