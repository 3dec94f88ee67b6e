"""Per-image timing of the end-to-end pipeline stages (debugging aid for fots_e2e.bench_e2e)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import torch
from fots_e2e.alphabet import ALPHABET
from bench_e2e import load_images
from fots_e2e.model import FOTSNet
from fots_e2e.pipeline import batched, preprocess, resize_rule
from e2e_inputs import synthetic_boxes
from oracle.e2e_loop_oracle import per_box
from fots_e2e.weights import deterministic_init
from rroi_align.decode import CTCLabelConverter

from fots_e2e.hostcpus import cap_torch_threads
cap_torch_threads()
dev = torch.device("cuda", 0)
net = deterministic_init(FOTSNet(87)).eval().to(dev)
conv = CTCLabelConverter(ALPHABET)
ims, _ = load_images(4)
boxes = [synthetic_boxes(24, *resize_rule(720, 1280), seed=100 + i) for i in range(len(ims))]
with torch.no_grad():
    for order in (("batched", batched), ("per_box", per_box), ("batched", batched), ("per_box", per_box)):
        for rep in range(3):
            line = []
            for i, im in enumerate(ims):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                im_data = preprocess(im, dev)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                _, _, _, feats = net(im_data)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                order[1](net, conv, feats, boxes[i])
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                line.append("%.1f/%.1f/%.1f" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
            print(order[0], rep, "pre/net/rec ms:", " ".join(line), flush=True)
