#!/usr/bin/env python3
"""configs[4]: the network and the batched recognition in torch.contiguous_format against torch.channels_last (MIOpen's
preferred layout; the callers' modules then hand channels-last crops over by themselves): ms per image, median of 4 passes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from fots_e2e.alphabet import ALPHABET
from bench_e2e import load_images
from fots_e2e.model import FOTSNet
from fots_e2e.pipeline import batched, preprocess, resize_rule
from e2e_inputs import synthetic_boxes
from fots_e2e.weights import deterministic_init
from fots_e2e.hostcpus import cap_torch_threads
from rroi_align.decode import CTCLabelConverter
cap_torch_threads()
dev = torch.device("cuda", 0)
conv = CTCLabelConverter(ALPHABET)
ims, _ = load_images()
boxes = [synthetic_boxes(24, *resize_rule(720, 1280), seed=100 + i) for i in range(len(ims))]
texts = {}
for fmt_name, fmt in (("contiguous", torch.contiguous_format), ("channels_last", torch.channels_last)):
    net = deterministic_init(FOTSNet(87)).eval().to(dev).to(memory_format=fmt)
    t_net, t_rec = [], []
    with torch.no_grad():
        for rep in range(5):
            for i, im in enumerate(ims):
                x = preprocess(im, dev).contiguous(memory_format=fmt)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                _, _, _, feats = net(x)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                out = batched(net, conv, feats, boxes[i])
                torch.cuda.synchronize(); t2 = time.perf_counter()
                if rep:
                    t_net.append((t1 - t0) * 1e3); t_rec.append((t2 - t1) * 1e3)
                texts.setdefault(fmt_name, {})[i] = out
    print(f"{fmt_name:14s} net {np.median(t_net):6.2f} ms  recognition {np.median(t_rec):6.2f} ms  -> {1e3 / (np.median(t_net) + np.median(t_rec)):6.1f} images/s; features channels_last: {feats[1].is_contiguous(memory_format=torch.channels_last) and not feats[1].is_contiguous()}", flush=True)
same = all(str(texts["contiguous"][i]) == str(texts["channels_last"][i]) for i in texts["contiguous"])
print("decoded texts equal between the layouts:", same)
