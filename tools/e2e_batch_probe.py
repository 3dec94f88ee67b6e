#!/usr/bin/env python3
"""Round 6: would a MULTI-IMAGE batch through the backbone pay?  ms per image of FOTSNet (fp32, NCHW, random weights) at
1280 x 704 for B = 1, 2, 4, 8 images per forward; `find` as first argument: with torch.backends.cudnn.benchmark = True
(MIOpen's find mode instead of its immediate-mode heuristics)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd")]
from fots_e2e.alphabet import ALPHABET
from fots_e2e.hostcpus import cap_torch_threads
from fots_e2e.model import FOTSNet
from fots_e2e.weights import deterministic_init
cap_torch_threads()
if len(sys.argv) > 1 and sys.argv[1] == "find":
    torch.backends.cudnn.benchmark = True
CL = len(sys.argv) > 1 and sys.argv[1] == "channels_last"   # the network and its input in channels_last
dev = torch.device("cuda", 0)
net = deterministic_init(FOTSNet(len(ALPHABET) + 1)).eval().to(dev)
if CL:
    net = net.to(memory_format=torch.channels_last)
with torch.no_grad():
    for B in (1, 2, 4, 8, 1):
        x = torch.randn(B, 3, 704, 1280, device=dev)
        if CL:
            x = x.contiguous(memory_format=torch.channels_last)
        t0 = time.perf_counter()
        for _ in range(3): net(x)
        torch.cuda.synchronize()
        warm = time.perf_counter() - t0
        ts = []
        for _ in range(8):
            t0 = time.perf_counter(); net(x); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"benchmark={torch.backends.cudnn.benchmark} channels_last={CL} B={B}: {np.median(ts) * 1e3:7.2f} ms per forward, {np.median(ts) * 1e3 / B:6.2f} ms per image (first three calls: {warm:.1f} s)", flush=True)
    crops = torch.randn(188, 64, 11, 64, device=dev)
    if CL:
        crops = crops.contiguous(memory_format=torch.channels_last)
    for _ in range(3): net.forward_ocr(crops)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): net.forward_ocr(crops)
    torch.cuda.synchronize(); print(f"head on 188 crops of 11 x 64: {(time.perf_counter() - t0) * 100:.2f} ms", flush=True)
