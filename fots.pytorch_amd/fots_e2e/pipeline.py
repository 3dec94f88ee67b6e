"""The inference driver around RoIRotate, restated on PyTorch-ROCm (`test.py:44-127`).

    im (H0, W0, 3) uint8 BGR --resize_rule/preprocess--> im_data (1, 3, H, W) in [-1, 1)
      --net--> score, rbox, angle maps (1/4) and features [merged 256 ch, focr 64 ch]
      --boxes--> (N, 9) quads + score   (`nms.get_boxes`; rroi_align.nms here, or a seeded synthetic
                                          set while the detection heads carry random weights)
      --recognise--> N strings

  batched   ROI rows of all boxes built on the device (`rroi_align_quads_to_rois_hip`), ONE RoIRotate
            launch per image at the widest pooled width, `forward_ocr` once per width BUCKET on the
            crops' own first target_gw columns (InstanceNorm statistics are per sample and per
            width, so a word must see exactly the columns the per-box call gives it), one batched
            greedy-CTC launch per bucket: per IMAGE a handful of launches and one read-back.

The reference's own structure (`tools/ocr_utils.py:131-199`: per WORD a host-built ROI, an upload, an R = 1
launch, the head, `max(1)`, a Python decode) is not part of this package: it lives with the checkers
(`oracle/e2e_loop_oracle.py`), where the tests compare the two and `bench_e2e.py` times it as the baseline.
"""

import numpy as np
import torch
import torch.nn.functional as F

from rroi_align.batched import _crops_channels_last, rois_from_quads
from rroi_align.decode import ctc_greedy_decode
from rroi_align.modules.rroi_align import _RRoiAlign

TARGET_H = 11           # tools/ocr_utils.py:147
SPATIAL_SCALE = 1.0 / 4  # :151, features[1] is the 1/4-resolution map


def resize_rule(h0, w0, max_size=1585152, scale_up=False):
    """(resize_h, resize_w) of `resize_image` (test.py:25-41): multiples of 32, area capped."""
    f = 3 if scale_up else 1
    size = [w0 * f // 32 * 32, h0 * f // 32 * 32]
    while size[0] * size[1] > max_size:
        size[0] /= 1.2
        size[1] /= 1.2
        size[0] = int(size[0] // 32) * 32
        size[1] = int(size[1] // 32) * 32
    return int(size[1]), int(size[0])


def preprocess(im_u8, device):
    """(H0, W0, 3) uint8 -> (1, 3, H, W) fp32 = resized / 128 - 1 (test.py:77-83).  The resize is
    half-pixel bilinear like cv2.resize's default, done on the device."""
    # the BYTES go up (a quarter of the fp32 image) and are widened on the device: `.to(device, dtype)` in one step converts
    # on the host first -- a parallel region of torch's intra-op pool per image, whose spinning workers are what a
    # CPU-quota'd container gets throttled for (hostcpus.py; round 6: batches of images took 36 or 70 ms at random)
    t = torch.as_tensor(im_u8).to(device=device).to(torch.float32).permute(2, 0, 1).unsqueeze(0)
    h, w = resize_rule(t.shape[2], t.shape[3])
    if (h, w) != tuple(t.shape[2:]):
        t = F.interpolate(t, size=(h, w), mode="bilinear", align_corners=False)
    return t / 128 - 1


def target_widths_host(boxes):
    """The callers' pooled-width rule (tools/ocr_utils.py:146-150) for boxes that are already on the host:
    target_gw = int(w * (11 / max(1, h))) + 11 rounded down to a multiple of 32, at least 64 -- w, h from the fp32
    corner differences, square root in double.  The same numbers the device kernel
    (`rroi_align_quads_to_rois_hip`, mode 0) writes next to its ROI rows (asserted equal in
    tests/test_e2e_gpu.py); having them here saves `infer_image` a read-back before the head."""
    b = np.asarray(boxes, np.float32)[:, :8].reshape(-1, 4, 2)
    dw, dh = b[:, 2] - b[:, 1], b[:, 1] - b[:, 0]
    w = np.sqrt((dw[:, 0] * dw[:, 0] + dw[:, 1] * dw[:, 1]).astype(np.float64))
    h = np.sqrt((dh[:, 0] * dh[:, 0] + dh[:, 1] * dh[:, 1]).astype(np.float64))
    gw = (w * (TARGET_H / np.maximum(1.0, h))).astype(np.int64) + TARGET_H
    return (np.maximum(2, gw // 32) * 32).tolist()


def batched(net, converter, features, boxes, return_crops=False, gw_host=None, batch_index=None):
    """All words of an image at once.  `boxes`: (N, >= 8) tensor on the device (or array).
    `gw_host`: the boxes' pooled widths when the caller already has them on the host (`infer_image`:
    the boxes come out of the host-side merge: `target_widths_host`) -- then nothing is
    read back before the head.
    `batch_index` (N,): the image each box belongs to when `features` hold SEVERAL images (`infer_batch`) -- the op's
    own first ROI column (`tools/ocr_utils.py:151` writes 0 there: one image per call); still ONE RoIRotate launch."""
    focr = features[1]
    quads = torch.as_tensor(boxes, dtype=torch.float32, device=focr.device)[:, :8].contiguous()
    n = quads.shape[0]
    if n == 0:
        return ([], [], []) if return_crops else []
    if batch_index is not None:
        batch_index = torch.as_tensor(batch_index, dtype=torch.float32).to(focr.device).contiguous()
    rois, gw = rois_from_quads(quads, batch_index, False, TARGET_H)
    if gw_host is None:
        gw_host = gw.cpu()                              # the one read-back before the head
    else:
        gw_host = torch.as_tensor(gw_host, dtype=torch.int32)
        if gw_host.numel() != n:
            raise ValueError("gw_host must have one width per box")
    widths = sorted(set(int(v) for v in gw_host))
    # (the crops follow the features' layout: a channels_last network gets channels_last crops, no relayout on either side)
    crops_all = _RRoiAlign(TARGET_H, widths[-1], SPATIAL_SCALE, _crops_channels_last(focr, None))(focr, rois)
    texts = [None] * n
    crops, labels = [None] * n, [None] * n
    # every bucket's head and decode are enqueued before anything is read back: one wait for the image,
    # not two per bucket (the buckets' index lists come from the widths that are on the host already)
    pending = []
    gw_list = gw_host.tolist()
    for wdt in widths:                                  # buckets are multiples of 32: a handful
        idx = [i for i, v in enumerate(gw_list) if int(v) == wdt]
        idx_dev = torch.tensor(idx, dtype=torch.int64).to(focr.device, non_blocking=True)
        x = crops_all.index_select(0, idx_dev)[:, :, :, :wdt]
        logp = net.forward_ocr(x)
        decoded, dlen, lab = ctc_greedy_decode(logp, None, return_labels=True)
        pending.append((idx, decoded, dlen, x, lab))
    for idx, decoded, dlen, x, lab in pending:
        words = converter.to_texts(decoded, dlen)       # the first of these waits for the whole image
        for j, i in enumerate(idx):
            texts[i] = words[j]
            if return_crops:
                crops[i], labels[i] = x[j:j + 1], lab[j].to(torch.int64)
    return (texts, crops, labels) if return_crops else texts


def infer_image(net, converter, im, detector=None, segm_thresh=0.5, return_debug=False):
    """One image through the whole chain of `test.py:75-116`: preprocess -> net -> `get_boxes` on the maps
    where the network wrote them -> RoIRotate + recognition head + greedy CTC for every box ->
    (boxes (n, 9) numpy, texts); like the reference's loop, boxes whose text is empty are dropped
    (test.py:109-110).  `return_debug` appends (all boxes, the recogniser's (texts, crops, labels), the features).

    `detector`: optional hook `im_data -> (score (h, w), rbox (4, h, w), angle (2, h, w))` device tensors
    that stand in for the three head outputs -- random weights pass no box (or a hundred thousand)
    through the NMS, so tests and the benchmark inject the maps of `tests/e2e_inputs.py: synthetic_detector_maps` here.

    Host synchronisations per image: ONE before the head (`get_boxes` reads the number of passing pixels
    and their records: the merge is sequential host code) and the final read-back of the decoded labels.
    The pooled-width buckets need no second one: the boxes are on the host after the merge and the width
    rule is plain arithmetic (`target_widths_host`)."""
    from rroi_align.nms import get_boxes
    device = next(net.parameters()).device
    im_data = preprocess(im, device) if not isinstance(im, torch.Tensor) else im
    score, rbox, angle, feats = net(im_data)
    if detector is not None:
        s, r, a = detector(im_data)
    else:
        s, r, a = score[0][0, 0], rbox[0][0], angle[0][0]
    boxes = get_boxes(s, r, a, segm_thresh)
    out = batched(net, converter, feats, boxes, return_crops=return_debug, gw_host=target_widths_host(boxes))
    texts = out[0] if return_debug else out
    keep = [i for i, t in enumerate(texts) if len(t) > 0]
    res = (boxes[keep], [texts[i] for i in keep])
    return res + ((boxes, out, feats),) if return_debug else res


def _batch_front(net, ims, detector, segm_thresh):
    """The first half of `infer_batch`, ENQUEUED on the current stream and nothing read back: preprocessing, one pass of the
    network, one decode launch per image."""
    from rroi_align.nms import decode_batch
    device = next(net.parameters()).device
    if isinstance(ims, torch.Tensor):
        im_data = ims
    else:
        sizes = {resize_rule(im.shape[0], im.shape[1]) for im in ims}
        if len(sizes) != 1:
            raise ValueError("infer_batch: the images must resize to one size, got %s (group them by size)" % sorted(sizes))
        im_data = torch.cat([preprocess(im, device) for im in ims], 0)
    score, rbox, angle, feats = net(im_data)
    if detector is not None:
        s, r, a = detector(im_data)
    else:
        s, r, a = score[0][:, 0], rbox[0], angle[0]
    return im_data.shape[0], feats, decode_batch(s, r, a, segm_thresh)


def _batch_back(net, converter, front, return_debug=False):
    """The second half, on the current stream: the boxes of every image (`merge_decoded`: two synchronisations, host merges),
    ONE RoIRotate launch for all their words, the head per pooled-width bucket, the strings."""
    from rroi_align.nms import merge_decoded
    nimg, feats, decoded = front
    per_image = merge_decoded(decoded)
    counts = [len(b) for b in per_image]
    boxes = np.concatenate(per_image, 0) if nimg else np.zeros((0, 9), np.float32)
    bidx = np.repeat(np.arange(nimg, dtype=np.float32), counts)
    out = batched(net, converter, feats, boxes, return_crops=return_debug, gw_host=target_widths_host(boxes) if len(boxes) else [],
                  batch_index=bidx)
    texts = out[0] if return_debug else out
    res, at = [], 0
    for b in range(nimg):
        t = texts[at:at + counts[b]]
        keep = [i for i, x in enumerate(t) if len(x) > 0]
        res.append((per_image[b][keep], [t[i] for i in keep]))
        at += counts[b]
    return (res, (per_image, out, feats)) if return_debug else res


def infer_batch(net, converter, ims, detector=None, segm_thresh=0.5, return_debug=False):
    """SEVERAL images of one size through the chain of `test.py:75-116` at once (round 6; the reference's loop takes one
    image per pass, `test.py:62-75` -- its test set, ICDAR 2015, is 1280 x 720 throughout): ONE pass through the network
    for the batch, `get_boxes` per image (every image's device decode is enqueued before the first read-back: one wait
    for the batch), ONE RoIRotate launch for the words of ALL images (the op's batch index, `kernel.cu:46`), the
    recognition head once per pooled-width bucket across the images, one read-back of the decoded labels.
    -> a list of `infer_image`'s (boxes (n, 9) numpy, texts) per image, in order.  A word's crop is bit-identical to the
    one the per-image chain cuts out of the same feature map (`tests/test_e2e_gpu.py`).

    `ims`: a list of (H0, W0, 3) uint8 arrays that `resize_rule` maps to ONE size, or an (N, 3, H, W) tensor that is
    already preprocessed.  `detector`: as in `infer_image`, for the whole batch: `im_data -> (score (N, h, w),
    rbox (N, 4, h, w), angle (N, 2, h, w))`."""
    if len(ims) == 0:
        return ([], ([], ([], [], []), None)) if return_debug else []
    return _batch_back(net, converter, _batch_front(net, ims, detector, segm_thresh), return_debug)


def infer_stream(net, converter, batches, detector=None, segm_thresh=0.5):
    """`infer_batch` over a SEQUENCE of batches with two batches in flight (a generator: one list of (boxes, texts) per
    batch, in order): the network pass of batch k + 1 is enqueued on the caller's stream BEFORE batch k's boxes are read
    back, and batch k's second half -- read-backs, host merges, RoIRotate, head, strings -- runs on a second stream
    beside it.  The device never waits for the host's merges and string building, the host never waits for a network
    pass it does not need yet; results are those of `infer_batch` (same kernels on the same data).  `detector`: a callable
    `(k, im_data) -> maps` (the batch's index first), or None."""
    device = next(net.parameters()).device
    side = torch.cuda.Stream(device=device)

    def back(front, ready):
        with torch.cuda.stream(side):
            side.wait_event(ready)
            res = _batch_back(net, converter, front)      # its read-backs synchronise `side` only
        return res
    prev = None
    for k, ims in enumerate(batches):
        if len(ims) == 0:
            if prev is not None:
                yield back(*prev)
                prev = None
            yield []
            continue
        front = _batch_front(net, ims, None if detector is None else (lambda x, k=k: detector(k, x)), segm_thresh)
        ready = torch.cuda.Event()
        ready.record()
        if prev is not None:
            yield back(*prev)
        prev = (front, ready)
    if prev is not None:
        yield back(*prev)
