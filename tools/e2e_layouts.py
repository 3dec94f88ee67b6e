"""End-to-end pipeline: where the occasional 40-80 ms per-image stalls come from (debugging aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fots.pytorch_amd"), os.path.join(ROOT, "tests")]
import torch
from fots_e2e.alphabet import ALPHABET
from bench_e2e import load_images
from fots_e2e.model import FOTSNet
from fots_e2e.pipeline import batched, preprocess, resize_rule
from e2e_inputs import synthetic_boxes
from fots_e2e.weights import deterministic_init
from rroi_align.decode import CTCLabelConverter

dev = torch.device("cuda", 0)
net = deterministic_init(FOTSNet(87)).eval().to(dev)
conv = CTCLabelConverter(ALPHABET)
ims, _ = load_images()
boxes = [synthetic_boxes(24, *resize_rule(720, 1280), seed=100 + i) for i in range(len(ims))]
st = lambda k: torch.cuda.memory_stats(dev)[k]
with torch.no_grad():
    for rep in range(4):
        for i, im in enumerate(ims):
            f0, a0 = st("num_device_free"), st("num_device_alloc")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            im_data = preprocess(im, dev)
            _, _, _, feats = net(im_data)
            batched(net, conv, feats, boxes[i])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            if dt > 20 or rep == 3:
                print("rep %d img %2d: %6.1f ms  device allocs +%d frees +%d  reserved %.0f MB" %
                      (rep, i, dt, st("num_device_alloc") - a0, st("num_device_free") - f0, st("reserved_bytes.all.current") / 1e6), flush=True)
