"""oracle/roi_build_oracle.py -- CPU restatement of the callers' ROI construction
(SURVEY.md section 8 row a11 / section 8f ranks 1-2).  TEST INFRASTRUCTURE ONLY (see
rroi_align_oracle.py).  Citations relative to /root/reference.

mode 0, inference -- tools/ocr_utils.py:133-150 (`align_ocr`, one detected box):
    boxr = boxo[0:8].reshape(-1, 2)                      fp32 quad from nms.get_boxes
    center = (b0 + b1 + b2 + b3) / 4                     fp32
    dw = b2 - b1 ; dh = b1 - b0                          fp32
    w = math.sqrt(dw0*dw0 + dw1*dw1)                     fp32 products and sum, sqrt in double
    h = math.sqrt(dh0*dh0 + dh1*dh1)
    angle = -atan2(b2y - b1y, b2x - b1x) / 3.1415926535 * 180     double
    rroi = [0, int(cx), int(cy), h, w, angle] -> torch.float
    target_gw = max(2, (int(w * (11 / max(1, h))) + 11) // 32) * 32

mode 1, training ground truth -- src/ocr_process.py:196-206 (arrays are float64 there):
    center as above in double; w, h = sqrt of squared edge lengths in double
    angle = -( atan2(edge 1->2) + atan2(edge 0->3) ) / 2 / 3.1415926535 * 180
    rroi = [bid, cx, cy, h (+ caller's random jitter), w, angle] -> torch.float
    pooled_width = ceil(11 * max(w / h))                 (:259-263, on the fp32 rois)
"""
import math

import numpy as np


def rois_from_quads(quads, batch_idx=None, mode=0, target_h=11, jitter=None):
    """quads (N, 8) fp32 [x0,y0,...,x3,y3] -> (rois (N,6) fp32, target_gw (N,) int32).
    jitter (mode 1): the caller's random.randint(-2, 2) of ocr_process.py:204, scalar or per box."""
    q = np.ascontiguousarray(quads, np.float32).reshape(-1, 4, 2)
    n = q.shape[0]
    bidx = np.zeros(n, np.float32) if batch_idx is None else np.asarray(batch_idx, np.float32)
    rois = np.zeros((n, 6), np.float32)
    gw = np.zeros(n, np.int32)
    for i in range(n):
        b = q[i]
        if mode == 0:
            center = (b[0] + b[1] + b[2] + b[3]) / np.float32(4)          # fp32, ocr_utils.py:136
            dw, dh = b[2] - b[1], b[1] - b[0]                              # :138-139
            w = math.sqrt(dw[0] * dw[0] + dw[1] * dw[1])                   # :140 (fp32 inside, double sqrt)
            h = math.sqrt(dh[0] * dh[0] + dh[1] * dh[1])                   # :141
            angle = math.atan2(b[2][1] - b[1][1], b[2][0] - b[1][0])       # :143
            angle = -angle / 3.1415926535 * 180                            # :144
            cx, cy = int(center[0]), int(center[1])                        # :145
        else:
            d = b.astype(np.float64)
            center = (d[0] + d[1] + d[2] + d[3]) / 4                       # ocr_process.py:198
            dw, dh = d[2] - d[1], d[1] - d[0]                              # :199-200
            w = math.sqrt(dw[0] ** 2 + dw[1] ** 2)                         # :201-203
            h = math.sqrt(dh[0] ** 2 + dh[1] ** 2)                         # :204
            if jitter is not None:
                h = h + float(np.broadcast_to(np.asarray(jitter, np.float64), (n,))[i])  # + random.randint(-2, 2)
            angle = (math.atan2(d[2][1] - d[1][1], d[2][0] - d[1][0]) +
                     math.atan2(d[3][1] - d[0][1], d[3][0] - d[0][0])) / 2  # :205
            angle = -angle / 3.1415926535 * 180                            # :206
            cx, cy = center[0], center[1]
        rois[i] = np.asarray([bidx[i], cx, cy, h, w, angle], np.float64).astype(np.float32)
        scale = target_h / max(1, h)                                       # ocr_utils.py:148
        t = int(w * scale) + target_h                                      # :149
        gw[i] = max(2, t // 32) * 32                                       # :150
    return rois, gw


def train_pooled_width(rois, pooled_h=11):
    """ocr_process.py:259-263: ceil(pooled_height * max(w / h)) on the fp32 roi tensor."""
    r = np.asarray(rois, np.float32)
    return int(math.ceil(pooled_h * float((r[:, 4] / r[:, 3]).max())))
