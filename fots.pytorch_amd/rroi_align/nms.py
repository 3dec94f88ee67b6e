"""rroi_align.nms -- detection post-processing (`nms.get_boxes`, nms/__init__.py:20-29 of the
reference): score / RBOX / angle maps -> merged word quads, the input of ROI construction.

The reference pulls the three full-resolution maps to the host, transposes them with numpy and
runs everything in C++ on the CPU (test.py:86-96, nms/adaptor.cpp, nms/nms.h).  Here the per-pixel
half -- threshold, RBOX -> quad decode -- is a HIP kernel on the maps where the
network wrote them (channels-first, no transposes), and only the passing pixels' 64-byte records
(in raster order) travel to the host, where the reference's inherently sequential locality-aware
merge + polygon NMS runs on them (`rroi_nms_merge_host`, same arithmetic as nms.h).
"""
import ctypes

import numpy as np
import torch

from ._ext import rroi_align as _ext

# rdist = the pixel's raw RBOX distances r[0..3]; the host merge forms the corner confidences from them
CANDIDATE = np.dtype([("quad", "<i4", (8,)), ("score", "<f4"), ("rdist", "<f4", (4,)),
                      ("x", "<i4"), ("y", "<i4"), ("pad", "<i4")])
assert CANDIDATE.itemsize == 64


def decode(score, rbox, angle, segm_thresh=0.5):
    """score (h, w), rbox (4, h, w), angle (2, h, w) fp32 on the GPU -> (records uint8 (h*w, 64) on
    the GPU, count int32 (1,) on the GPU): the candidates of adaptor.cpp:76-117 in raster order."""
    for t, name in ((score, "score"), (rbox, "rbox"), (angle, "angle")):
        _ext._require_cuda_f32(t, name)
    h, w = score.shape[-2:]
    score, rbox, angle = score.reshape(h, w).contiguous(), rbox.reshape(4, h, w).contiguous(), angle.reshape(2, h, w).contiguous()
    slabs = (h * w + 1023) // 1024
    # maps beyond 256 K pixels: the library keeps its per-slab counts behind the h * w records
    cap = h * w + ((slabs * 4 + 63) // 64 if slabs > 256 else 0)
    with torch.cuda.device_of(score):
        rec = torch.empty((cap, 64), dtype=torch.uint8, device=score.device)
        cnt = torch.empty((1,), dtype=torch.int32, device=score.device)
        st = _ext._lib.rroi_rbox_decode_hip(score.data_ptr(), rbox.data_ptr(), angle.data_ptr(), h, w,
                                            float(segm_thresh), rec.data_ptr(), cap, cnt.data_ptr(), _ext._stream())
    _ext._check(st, "rroi_rbox_decode_hip")
    return rec, cnt


def merge(records, width, height, iou_threshold=0.4, iou_threshold2=0.2):
    """Candidate records (numpy, dtype CANDIDATE or (n, 64) uint8, host) -> (n, 9) fp32 boxes
    [x0,y0,..,x3,y3 in input-image pixels, score] (nms.h:149-213 then :116-146)."""
    rec = np.ascontiguousarray(records).view(np.uint8).reshape(-1, 64)
    n = rec.shape[0]
    # after the merge: at most two entries per candidate (nms.h:198,201)
    out = np.empty((max(1, 2 * n), 9), np.float32)
    k = _ext._lib.rroi_nms_merge_host(rec.ctypes.data_as(ctypes.c_void_p), n, int(width), int(height),
                                      float(iou_threshold), float(iou_threshold2),
                                      out.ctypes.data_as(ctypes.c_void_p), out.shape[0])
    if k < 0:
        raise ValueError("rroi_nms_merge_host: invalid argument")
    return out[:k].copy()


def get_boxes(iou_map, rbox, angle_pred, segm_thresh=0.5):
    """`nms.get_boxes` (nms/__init__.py:20-29).  Device tensors in the network's own layout --
    iou_map (h, w), rbox (4, h, w), angle_pred (2, h, w) -- or, for call-site compatibility, the
    reference's numpy arrays (rbox as (h, w, 4)), which are uploaded first.  -> (n, 9) numpy fp32."""
    if not isinstance(iou_map, torch.Tensor):
        dev = torch.device("cuda", torch.cuda.current_device())
        iou_map = torch.as_tensor(np.ascontiguousarray(iou_map, np.float32), device=dev)
        rbox = torch.as_tensor(np.ascontiguousarray(np.asarray(rbox, np.float32).transpose(2, 0, 1)), device=dev)
        angle_pred = torch.as_tensor(np.ascontiguousarray(angle_pred, np.float32), device=dev)
    h, w = iou_map.shape[-2:]
    rec, cnt = decode(iou_map, rbox, angle_pred, segm_thresh)
    n = int(cnt.item())                      # the one synchronisation: how many pixels passed
    host = rec[:n].cpu().numpy()
    return merge(host, w, h, 0.4, 0.2)


_merge_pool = None


def _merge_workers():
    """A few host threads for the merges of a batch (`rroi_nms_merge_host` keeps no state between calls and ctypes
    releases the GIL around it); created on first use, never more than four."""
    global _merge_pool
    if _merge_pool is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        try:
            cpus = len(os.sched_getaffinity(0))
        except AttributeError:  # pragma: no cover - non-Linux
            cpus = os.cpu_count() or 1
        _merge_pool = ThreadPoolExecutor(max_workers=max(1, min(4, cpus)), thread_name_prefix="rroi-nms-merge")
    return _merge_pool


def decode_batch(iou_map, rbox, angle_pred, segm_thresh=0.5):
    """The device half of `get_boxes_batch`: one decode launch per image, nothing read back -> what `merge_decoded` takes."""
    n_img = iou_map.shape[0]
    return [decode(iou_map[i], rbox[i], angle_pred[i], segm_thresh) for i in range(n_img)], tuple(iou_map.shape[-2:])


def merge_decoded(decoded):
    """The host half: the N counts come back together, then the passing pixels' records of all images in one copy (two
    synchronisations of the CURRENT stream for the batch); the merges -- sequential code per image, as in the reference --
    run on a few threads side by side.  -> a list of N (n_i, 9) numpy fp32 arrays."""
    pending, (h, w) = decoded
    if not pending:
        return []
    counts = torch.cat([c for _, c in pending]).cpu().tolist()      # synchronisation 1: how many pixels passed, per image
    host = torch.cat([rec[:n] for (rec, _), n in zip(pending, counts)]).cpu().numpy()   # 2: their records
    ends = np.cumsum(counts)
    parts = [host[e - n:e] for e, n in zip(ends, counts)]
    if len(parts) == 1:
        return [merge(parts[0], w, h, 0.4, 0.2)]
    return list(_merge_workers().map(lambda part: merge(part, w, h, 0.4, 0.2), parts))


def get_boxes_batch(iou_map, rbox, angle_pred, segm_thresh=0.5):
    """`get_boxes` for the maps of SEVERAL images -- iou_map (N, h, w), rbox (N, 4, h, w), angle_pred (N, 2, h, w) on the
    GPU -- with TWO host synchronisations for the batch instead of two per image: every image's decode launch is
    enqueued first (`decode_batch`), then `merge_decoded`.
    -> a list of N (n_i, 9) numpy fp32 arrays, each equal to `get_boxes` of that image's maps."""
    return merge_decoded(decode_batch(iou_map, rbox, angle_pred, segm_thresh))
