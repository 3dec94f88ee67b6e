#!/usr/bin/env python3
"""Round 6: where does a batch of images spend its time in `infer_batch`?  Stage by stage, device-synchronised."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
from bench_e2e import load_images, BOXES_PER_IMAGE
from e2e_inputs import synthetic_detector_maps
from fots_e2e.alphabet import ALPHABET
from fots_e2e.hostcpus import cap_torch_threads
from fots_e2e.model import FOTSNet
from fots_e2e.pipeline import batched, preprocess, target_widths_host, infer_batch
from fots_e2e.weights import deterministic_init
from rroi_align.decode import CTCLabelConverter
from rroi_align.nms import get_boxes_batch
cap_torch_threads()
dev = torch.device("cuda", 0)
net = deterministic_init(FOTSNet(len(ALPHABET) + 1)).eval().to(dev)
conv = CTCLabelConverter(ALPHABET)
ims, _ = load_images()
maps = [tuple(torch.from_numpy(a).to(dev) for a in synthetic_detector_maps(704, 1280, BOXES_PER_IMAGE, seed=i)) for i in range(len(ims))]
def sync(): torch.cuda.synchronize(dev)
for B in (8, 4, 1):
    g = list(range(B))
    stacked = tuple(torch.stack([maps[i][j] for i in g]) for j in range(3))
    with torch.no_grad():
        for rep in range(4):
            sync(); t = [time.perf_counter()]
            x = torch.cat([preprocess(ims[i], dev) for i in g], 0); sync(); t.append(time.perf_counter())
            score, rbox, angle, feats = net(x); sync(); t.append(time.perf_counter())
            per_image = get_boxes_batch(*stacked, 0.5); sync(); t.append(time.perf_counter())
            boxes = np.concatenate(per_image, 0)
            bidx = np.repeat(np.arange(B, dtype=np.float32), [len(b) for b in per_image])
            gw = target_widths_host(boxes); t.append(time.perf_counter())
            texts = batched(net, conv, feats, boxes, gw_host=gw, batch_index=bidx); sync(); t.append(time.perf_counter())
            d = np.diff(t) * 1e3
        print(f"B={B}: preprocess {d[0]:.2f}  net {d[1]:.2f}  get_boxes_batch {d[2]:.2f}  widths {d[3]:.2f}  recognition {d[4]:.2f}  "
              f"total {sum(d):.2f} ms = {sum(d) / B:.2f} ms per image; words {len(boxes)}, buckets {sorted(set(gw))}", flush=True)

# the bench leg's own loop: three different groups of eight, whole chain per call
order = [i % len(ims) for i in range(24)]
groups = [order[i:i + 8] for i in range(0, 24, 8)]
with torch.no_grad():
    for rep in range(4):
        row = []
        for g in groups:
            stacked = tuple(torch.stack([maps[i][j] for i in g]) for j in range(3))
            sync(); t0 = time.perf_counter()
            r = infer_batch(net, conv, [ims[i] for i in g], detector=lambda _x: stacked)
            sync(); row.append((time.perf_counter() - t0) * 1e3)
        print("groups of 8, ms per batch:", " ".join(f"{v:.1f}" for v in row), flush=True)
