"""bench_e2e.py -- benchmark code (imported by bench.py for `extra.e2e`), not part of the product package.

BASELINE configs[4]: images/s of the end-to-end inference pipeline (`test.py:75-116`) on the
reference's own test images (`data/example_image/*.jpg`, 1280x720 -> 1280x704), random weights
(the checkpoint is not in the reference's repository), per-box vs batched recognition.

Timed per image, after warm-up, device-synchronised: upload of the decoded uint8 image, resize +
normalisation, backbone + heads, recognition of the image's boxes (RoIRotate + CRNN head + greedy
CTC) and the read-back of the strings.  JPEG decoding is done once, outside the timed region.
Boxes: `synthetic_boxes` (24 per image, seeded) -- with random detection weights the score map
carries no text regions, so the detector's post-processing is measured on its own (rroi_align.nms).
"""
import glob
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "fots.pytorch_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from fots_e2e.alphabet import ALPHABET  # noqa: E402
from fots_e2e.hostcpus import cap_torch_threads  # noqa: E402
from fots_e2e.model import FOTSNet  # noqa: E402
from fots_e2e.pipeline import batched, infer_batch, infer_image, infer_stream, preprocess, resize_rule  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
from e2e_inputs import synthetic_boxes, synthetic_detector_maps  # noqa: E402  (input generators, not product)
from fots_e2e.weights import deterministic_init  # noqa: E402
# the baseline leg: the reference's per-word loop (checker / baseline code, kept outside the product package)
from oracle.e2e_loop_oracle import infer_image_per_box, per_box  # noqa: E402

BOXES_PER_IMAGE = 24


def load_images(limit=None):
    """Decoded uint8 BGR arrays of the reference's example images (data fixtures under
    tests/golden/ref_data), or seeded noise of the same size when they are not there."""
    paths = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "ref_data", "example_image", "*.jpg")))
    ims, source = [], "data/example_image/*.jpg (11 images, 1280x720)"
    try:
        from PIL import Image
        for p in paths[:limit]:
            ims.append(np.ascontiguousarray(np.asarray(Image.open(p).convert("RGB"))[:, :, ::-1]))  # BGR like cv2.imread
    except Exception:
        ims = []
    if not ims:
        rng = np.random.default_rng(0)
        ims = [rng.integers(0, 256, (720, 1280, 3), dtype=np.uint8) for _ in range(limit or 11)]
        source = "seeded noise, 1280x720 (example images not found)"
    return ims, source


def measure(device, reps=5, channels_last=False):
    from rroi_align.decode import CTCLabelConverter
    host_threads = cap_torch_threads()  # hostcpus.py: an oversized intra-op pool gets the process throttled
    net = deterministic_init(FOTSNet(len(ALPHABET) + 1)).eval().to(device)
    if channels_last:
        net = net.to(memory_format=torch.channels_last)
    conv = CTCLabelConverter(ALPHABET)
    ims, source = load_images()
    boxes = []
    for i, im in enumerate(ims):
        h, w = resize_rule(im.shape[0], im.shape[1])
        boxes.append(synthetic_boxes(BOXES_PER_IMAGE, h, w, seed=100 + i))

    def run(recognise):
        """one pass over the images -> per-image (backbone seconds, recognition seconds), last texts"""
        samples, texts = [], None
        for i, im in enumerate(ims):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            im_data = preprocess(im, device)
            _, _, _, feats = net(im_data)
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            texts = recognise(net, conv, feats, boxes[i])
            torch.cuda.synchronize(device)
            samples.append((t1 - t0, time.perf_counter() - t1))
        return samples, texts

    out = {}
    with torch.no_grad():
        same = None
        for name, fn in (("per_box", per_box), ("batched", batched)):
            run(fn)  # warm-up: MIOpen picks its kernels for every crop width here
            samples = []
            t0 = time.perf_counter()
            for _ in range(reps):
                s, texts = run(fn)
                samples += s
            wall = time.perf_counter() - t0
            a = np.asarray(samples)
            med = float(np.median(a.sum(1)))
            # median per-image time and whole-pass mean: they agree once the intra-op pool is capped to
            # the cgroup's CPU quota (before: 70-80 ms throttling stalls on every third image or so)
            out[name] = {"images_per_s": round(1.0 / med, 2),
                         "images_per_s_mean": round(len(samples) / wall, 2),
                         "backbone_ms_per_image": round(float(np.median(a[:, 0])) * 1e3, 3),
                         "recognition_ms_per_image": round(float(np.median(a[:, 1])) * 1e3, 3)}
            same = texts if same is None else (same == texts)
    # the detector's post-processing on its own (see the module docstring): 24 words at 1280 x 704
    from rroi_align.nms import get_boxes
    maps = [tuple(torch.from_numpy(a).to(device) for a in synthetic_detector_maps(704, 1280, BOXES_PER_IMAGE, seed=i))
            for i in range(len(ims))]
    # ---- the whole chain of test.py:75-116 per image: preprocess, net, get_boxes ON the maps (the
    # synthetic trained-detector maps stand in for the three head outputs), recognition of ITS boxes
    with torch.no_grad():
        for name in ("per_box", "batched"):
            def chain(i, im):
                fn = infer_image_per_box if name == "per_box" else infer_image
                return fn(net, conv, im, detector=lambda _x, m=maps[i]: m)
            for i, im in enumerate(ims):
                chain(i, im)                                   # warm-up
            per_image, nbox = [], 0
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(reps):
                for i, im in enumerate(ims):
                    t1 = time.perf_counter()
                    b, _t = chain(i, im)
                    torch.cuda.synchronize(device)
                    per_image.append(time.perf_counter() - t1)
                    nbox += len(b)
            wall = time.perf_counter() - t0
            out[name]["chain_images_per_s"] = round(1.0 / float(np.median(per_image)), 2)
            out[name]["chain_images_per_s_mean"] = round(len(per_image) / wall, 2)
            out[name]["chain_boxes_per_image"] = round(nbox / len(per_image), 1)
    # ---- round 6: the same chain over BATCHES of images (`infer_batch`: one pass of the network, one RoIRotate launch and
    # one set of head launches for the words of all images of the batch) -- throughput mode; the reference's loop has none
    IMAGES_PER_BATCH = 8
    with torch.no_grad():
        order = [i % len(ims) for i in range(2 * IMAGES_PER_BATCH)]        # two different batches: every image at least once
        groups = [order[i:i + IMAGES_PER_BATCH] for i in range(0, len(order), IMAGES_PER_BATCH)]

        def chain_batch(g):
            stacked = tuple(torch.stack([maps[i][j] for i in g]) for j in range(3))
            return infer_batch(net, conv, [ims[i] for i in g], detector=lambda _x: stacked)
        for g in groups:
            chain_batch(g)                                     # warm-up
        per_batch, nbox = [], 0
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(max(1, min(reps, 3))):
            for g in groups:
                t1 = time.perf_counter()
                r = chain_batch(g)
                torch.cuda.synchronize(device)
                per_batch.append(time.perf_counter() - t1)
                nbox += sum(len(b) for b, _t in r)
        wall = time.perf_counter() - t0
        out["image_batch"] = {"images_per_batch": IMAGES_PER_BATCH,
                              "chain_images_per_s": round(IMAGES_PER_BATCH / float(np.median(per_batch)), 2),
                              "chain_images_per_s_mean": round(IMAGES_PER_BATCH * len(per_batch) / wall, 2),
                              "chain_boxes_per_image": round(nbox / (IMAGES_PER_BATCH * len(per_batch)), 1),
                              "what": "infer_batch: %d images per pass of the network (uploads and preprocessing included), "
                                      "get_boxes per image behind one synchronisation, ONE RoIRotate launch for the words of all "
                                      "images (the op's batch index), the head per pooled-width bucket across the images" % IMAGES_PER_BATCH}
        # ... and with two batches in flight (`infer_stream`): batch k's read-backs, host merges, recognition and strings on a
        # side stream beside the network pass of batch k + 1 -- wall clock over the whole sequence
        n_seq = max(2, 2 * min(reps, 3)) * len(groups)
        seq_groups = [groups[i % len(groups)] for i in range(n_seq)]
        seq_maps = [tuple(torch.stack([maps[i][j] for i in g]) for j in range(3)) for g in seq_groups]

        def run_stream():
            nb = 0
            for r in infer_stream(net, conv, ([ims[i] for i in g] for g in seq_groups), detector=lambda k, _x: seq_maps[k]):
                nb += sum(len(b) for b, _t in r)
            return nb
        run_stream()                                           # warm-up (the side stream's MIOpen handle and workspaces)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        nbox = run_stream()
        torch.cuda.synchronize(device)
        wall = time.perf_counter() - t0
        out["image_stream"] = {"images_per_batch": IMAGES_PER_BATCH, "batches": n_seq,
                               "chain_images_per_s": round(IMAGES_PER_BATCH * n_seq / wall, 2),
                               "chain_boxes_per_image": round(nbox / (IMAGES_PER_BATCH * n_seq), 1),
                               "what": "infer_stream: the image_batch chain with two batches in flight -- batch k's second half "
                                       "(read-backs, host merges, RoIRotate, head, strings) on a side stream beside the network "
                                       "pass of batch k + 1; wall clock over %d batches" % n_seq}
    out["chain"] = ("chain_*: preprocess + net + rroi_align.nms.get_boxes on the device maps (synthetic trained-detector "
                    "maps injected for the three head outputs: random weights pass no box) + recognition of the boxes "
                    "get_boxes returned; host synchronisations per image on the batched path: the read-back of the "
                    "passing pixels before the merge, then the decoded labels")
    times, found = [], 0
    for rep in range(reps + 1):
        for m in maps:
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            b = get_boxes(*m, 0.5)
            if rep:
                times.append(time.perf_counter() - t0)
                found += len(b)
    out["nms"] = {"ms_per_image": round(float(np.median(times)) * 1e3, 3), "boxes_per_image": round(found / len(times), 1),
                  "what": "rroi_align.nms.get_boxes on synthetic trained-detector maps (176 x 320, %d words): device "
                          "decode + read-back of the passing pixels + host merge" % BOXES_PER_IMAGE}
    out["what"] = ("configs[4]: %s -> 1280x704, FOTSNet (ModelResNetSep2 restated, random weights), %d seeded boxes per "
                   "image, RoIRotate 11 x target_gw on the 64-ch 1/4 map, CRNN head, greedy CTC; median per-image time over %d passes over "
                   "the images; per_box = the reference's loop (R = 1 launch, head and decode per word), batched = one "
                   "RoIRotate launch per image, head per width bucket" % (source, BOXES_PER_IMAGE, reps))
    out["last_image_texts_equal"] = bool(same) if isinstance(same, bool) else None
    out["speedup"] = round(out["batched"]["images_per_s"] / out["per_box"]["images_per_s"], 2)
    out["host_threads"] = host_threads
    return out


if __name__ == "__main__":
    import json
    print(json.dumps(measure(torch.device("cuda", 0))))
