#!/usr/bin/env python3
"""Round 6: host-side profile of the recognition stage of one image (`pipeline.batched`, 24 words): where do its 1.8 ms go?"""
import cProfile, os, pstats, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
from e2e_inputs import synthetic_detector_maps
from fots_e2e.alphabet import ALPHABET
from fots_e2e.hostcpus import cap_torch_threads
from fots_e2e.model import FOTSNet
from fots_e2e.pipeline import batched, target_widths_host
from fots_e2e.weights import deterministic_init
from rroi_align.decode import CTCLabelConverter
from rroi_align.nms import get_boxes
cap_torch_threads()
dev = torch.device("cuda", 0)
net = deterministic_init(FOTSNet(len(ALPHABET) + 1)).eval().to(dev)
conv = CTCLabelConverter(ALPHABET)
maps = tuple(torch.from_numpy(a).to(dev) for a in synthetic_detector_maps(704, 1280, 24, seed=0))
with torch.no_grad():
    _, _, _, feats = net(torch.randn(1, 3, 704, 1280, device=dev))
    boxes = get_boxes(*maps, 0.5)
    gw = target_widths_host(boxes)
    for _ in range(20): batched(net, conv, feats, boxes, gw_host=gw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): batched(net, conv, feats, boxes, gw_host=gw)
    torch.cuda.synchronize(); print(f"batched(): {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per image, {len(boxes)} words, widths {sorted(set(gw))}")
    # GPU time alone: events around the same loop
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): batched(net, conv, feats, boxes, gw_host=gw)
    e1.record(); torch.cuda.synchronize(); print(f"  between events: {e0.elapsed_time(e1) / 200:.3f} ms")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): batched(net, conv, feats, boxes, gw_host=gw)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
