#!/usr/bin/env python3
"""configs[2]'s backward 40 times through an OLDER build of the library (argv[1] = prev) or the product (new): run under
rocprofv3 --kernel-trace to compare per-kernel durations across builds on one box."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import workloads as Wk
from rroi_align._ext import rroi_align as ext
which = sys.argv[1] if len(sys.argv) > 1 else "new"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 512
lib = ext._lib
if which == "prev":
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_prev.so"))
    lib.rroi_align_backward_hip.argtypes = ext._lib.rroi_align_backward_hip.argtypes
    lib.rroi_align_backward_workspace_bytes.restype = ctypes.c_size_t
    lib.rroi_align_backward_workspace_bytes.argtypes = [ctypes.c_int] * 7
TRAIN = os.environ.get("TRAIN") == "1"   # the training shape instead of configs[2]
B, C, H, W, PH, PW = (2, 64, 120, 160, 11, 96) if TRAIN else (1, 256, 160, 160, 8, 64)
f, r = Wk.bench_inputs(R=R, C=C, H=H, W=W, img=4 * W, batch=B)
Rt = torch.from_numpy(r).cuda()
g = torch.randn(R, C, PH, PW, device="cuda")
gin = torch.empty(f.shape, device="cuda")
nb = lib.rroi_align_backward_workspace_bytes(B, C, H, W, R, PH, PW)
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(40):
    assert lib.rroi_align_backward_hip(g.data_ptr(), 0.25, B, R, H, W, C, PH, PW, Rt.data_ptr(), gin.data_ptr(), ws.data_ptr(), nb, 0, st) == 1
torch.cuda.synchronize()
