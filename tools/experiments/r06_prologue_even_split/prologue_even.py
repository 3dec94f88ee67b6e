#!/usr/bin/env python3
"""The forward prologue at configs[1] relays out 1600 tiles ([32 ch] x [128 px]) with 768 blocks (3 per CU): 2.08 tiles per block,
i.e. two full rounds and a third one in which 64 of the 768 blocks work.  Does an EVEN split pay (800 blocks x 2 tiles, 536 x 3,
400 x 4, 1600 x 1)?  Exploration build (rroi_align_debug_set_prologue_blocks_exact); prologue alone and the whole call, HIP
events, arms interleaved, four rounds."""
import ctypes, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import workloads as Wk
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_explore", "librroi_align_hip_explore.so"))
vp, fl, it, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t
lib.rroi_align_forward_stages_hip.argtypes = [vp, it, fl, it, it, it, it, it, it, it, vp, vp, vp, sz, it, it, vp]
lib.rroi_align_forward_workspace_bytes.restype = sz
lib.rroi_align_forward_workspace_bytes.argtypes = [it] * 6
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, warm=200, iters=500):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / iters * 1e3, 2)
def run(tag, B, C, H, W, n, ph, pw, arms):
    rng = np.random.default_rng(5)
    if tag == "configs[1]":
        f, r = Wk.bench_inputs()
    else:
        f = rng.standard_normal((B, C, H, W), dtype=np.float32)
        h = rng.uniform(16, 64, n)
        r = np.stack([rng.integers(0, B, n), rng.uniform(0, 4 * W, n), rng.uniform(0, 4 * H, n), h, h * rng.uniform(2, pw / float(ph), n), rng.uniform(-45, 45, n)], 1).astype(np.float32)
    F, R = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
    top = torch.empty((n, C, ph, pw), device="cuda")
    nb = lib.rroi_align_forward_workspace_bytes(B, C, H, W, n, 0)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    def call(stages):
        assert lib.rroi_align_forward_stages_hip(F.data_ptr(), 0, 0.25, B, n, H, W, C, ph, pw, R.data_ptr(), top.data_ptr(), ws.data_ptr(), nb, 2, stages, st) == 1
    ref = None
    res = {}
    for rnd in range(4):
        for blocks in arms:
            lib.rroi_align_debug_set_prologue_blocks_exact(blocks)
            call(3); torch.cuda.synchronize()
            if ref is None: ref = top.clone()
            same = bool(torch.equal(ref, top))
            res.setdefault(f"blocks_{blocks or 'rule'}", []).append([timeit(lambda: call(1), 100, 300), timeit(lambda: call(3)), same])
    lib.rroi_align_debug_set_prologue_blocks_exact(0)
    print(tag, json.dumps(res), flush=True)
run("configs[1]", 1, 256, 160, 160, 512, 8, 64, (0, 800, 536, 400, 1600, 1064, 0, 800))
run("C=256 R=1024", 1, 256, 160, 160, 1024, 8, 64, (0, 800, 536, 1600))
run("C=128 200x200", 1, 128, 200, 200, 512, 8, 64, (0, 632, 424, 1256))   # 313 ptiles x 4 chunks = 1252 tiles
