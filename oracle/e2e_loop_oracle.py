"""oracle/e2e_loop_oracle.py -- TEST / BENCHMARK INFRASTRUCTURE, not product.

The reference's inference loop over the detected words, restated: `test.py:102-116` calling
`tools/ocr_utils.py:131-199` (`align_ocr`) once per box -- ROI built on the host with Python floats
(`ocr_utils.py:133-150`), uploaded, `_RRoiAlign(11, target_gw, 1/4)` with R = 1, the recognition head, `max(1)`,
a Python CTC decode: ~6 launches + 1 upload + 1 read-back per WORD.

Used as the checker of the batched recognition path (`tests/test_e2e_gpu.py`: crops bit for bit, strings equal)
and as the baseline leg of `bench_e2e.py` (BASELINE configs[4]: images/s of the reference's structure next to
the batched one).  Only `tests/`, `bench_e2e.py` and `tools/` import it; the product package (`fots.pytorch_amd/`)
does not (`tests/test_abi.py::test_product_never_imports_the_oracle`).  The R = 1 launches go through the product's
own `_RRoiAlign` -- the loop's STRUCTURE is the reference's, the op under it is the thing being measured.
"""
import math

import numpy as np
import torch

TARGET_H = 11            # tools/ocr_utils.py:147
SPATIAL_SCALE = 1.0 / 4  # :151, features[1] is the 1/4-resolution map


def host_roi(box):
    """One box -> ([0, int(cx), int(cy), h, w, angle], target_gw) as `align_ocr` computes them on
    the host (tools/ocr_utils.py:133-150): numpy fp32 corner arithmetic, Python-float sqrt/atan2."""
    b = np.asarray(box[0:8], np.float32).reshape(-1, 2)
    center = (b[0, :] + b[1, :] + b[2, :] + b[3, :]) / 4
    dw, dh = b[2, :] - b[1, :], b[1, :] - b[0, :]
    w = math.sqrt(dw[0] * dw[0] + dw[1] * dw[1])
    h = math.sqrt(dh[0] * dh[0] + dh[1] * dh[1])
    angle = -math.atan2(b[2][1] - b[1][1], b[2][0] - b[1][0]) / 3.1415926535 * 180
    gw = int(w * (TARGET_H / max(1, h))) + TARGET_H
    return [0, int(center[0]), int(center[1]), h, w, angle], max(2, gw // 32) * 32


def per_box(net, converter, features, boxes, return_crops=False):
    """The reference's loop (test.py:102-116): one word at a time.  -> texts[, crops, labels]"""
    from rroi_align.modules.rroi_align import _RRoiAlign
    focr = features[1]
    texts, crops, labels = [], [], []
    for box in boxes:
        roi, gw = host_roi(box)
        rois = torch.tensor(roi).to(torch.float).to(focr.device)
        x = _RRoiAlign(TARGET_H, gw, SPATIAL_SCALE)(focr, rois.view(-1, 6))
        logp = net.forward_ocr(x)
        _, lab = logp.max(1)
        lab = lab.transpose(1, 0).contiguous().view(-1)
        texts.append(converter.decode(lab.cpu(), torch.IntTensor([lab.size(0)]), raw=False))
        if return_crops:
            crops.append(x)
            labels.append(lab)
    return (texts, crops, labels) if return_crops else texts


def infer_image_per_box(net, converter, im, detector=None, segm_thresh=0.5, return_debug=False):
    """`fots_e2e.pipeline.infer_image` with the recognition done by the reference's per-word loop: the same chain
    (preprocess -> net -> get_boxes -> recognition -> empty texts dropped, test.py:75-116)."""
    from fots_e2e.pipeline import preprocess
    from rroi_align.nms import get_boxes
    device = next(net.parameters()).device
    im_data = preprocess(im, device) if not isinstance(im, torch.Tensor) else im
    score, rbox, angle, feats = net(im_data)
    s, r, a = detector(im_data) if detector is not None else (score[0][0, 0], rbox[0][0], angle[0][0])
    boxes = get_boxes(s, r, a, segm_thresh)
    out = per_box(net, converter, feats, boxes, return_crops=return_debug)
    texts = out[0] if return_debug else out
    keep = [i for i, t in enumerate(texts) if len(t) > 0]
    res = (boxes[keep], [texts[i] for i in keep])
    return res + ((boxes, out, feats),) if return_debug else res
