"""rroi_align.batched -- one RoIRotate launch per image instead of one per word.

The reference's inference loop (`test.py:102-116` -> `tools/ocr_utils.py:131-199`, `align_ocr`)
builds the ROI of each detected box on the host with numpy, uploads it, and calls the op with
R = 1: launch-latency bound.  Here the ROI rows of ALL boxes are built on the device
(`rroi_align_quads_to_rois_hip`, same arithmetic as `ocr_utils.py:133-150`) and the op runs once
with a common pooled width, the largest `target_gw` of the batch.  The first `target_gw[i]`
columns of crop i are bit-identical to what the per-box call returns (returned alongside, so a
caller can slice exactly the reference's crop); beyond them the crop continues up to the box's own
`roi_pooled_width = 11*w/h` (the per-box rule rounds the width DOWN to a multiple of 32 and can cut
a word short) and is zero after that by the op's `pw <= roi_pooled_width` mask (kernel.cu:107).
"""
import torch
from torch.nn.modules.module import Module

from ._ext import rroi_align as _ext
from .modules.rroi_align import _RRoiAlign


def _crops_channels_last(features, setting):
    """The hand-off layout of the crops.  None (the default of both callers' modules, round 5) = FOLLOW THE FEATURES:
    channels_last features with C % 4 == 0 -- a backbone that runs in MIOpen's preferred layout -- give channels_last
    crops (same values, element for element), so the recognition head's first convolution relays nothing out, the
    gradient comes back channels_last, and the op's backward consumes it and writes the feature gradient in place:
    no pixel-major copy of top_diff at all (configs[2]'s shape: 0.057 ms instead of 0.115 ms).  True / False force it."""
    if setting is not None:
        return bool(setting)
    return (features.dim() == 4 and features.shape[1] % 4 == 0 and not features.is_contiguous()
            and features.is_contiguous(memory_format=torch.channels_last))


def rois_from_quads(quads, batch_index=None, training=False, target_h=11):
    """(N, 8) fp32 quads [x0,y0,..,x3,y3] -> ((N, 6) rois, (N,) int32 pooled widths)."""
    return _ext.quads_to_rois(quads, batch_index, 1 if training else 0, target_h)


class BatchedRRoiAlign(Module):
    """forward(features, quads[, batch_index]) -> (crops (N, C, target_h, max_gw), target_gw (N,))."""

    def __init__(self, target_h=11, spatial_scale=1.0 / 4, pooled_width=None, channels_last_out=None):
        super(BatchedRRoiAlign, self).__init__()
        self.target_h = int(target_h)
        self.spatial_scale = float(spatial_scale)
        self.pooled_width = pooled_width  # fixed width avoids the one-int device->host read
        self.channels_last_out = channels_last_out  # None: follow the features' layout (see _crops_channels_last)

    def forward(self, features, quads, batch_index=None):
        rois, gw = rois_from_quads(quads, batch_index, False, self.target_h)
        width = self.pooled_width
        if width is None:
            width = int(gw.max().item()) if gw.numel() else 64
        crops = _RRoiAlign(self.target_h, width, self.spatial_scale,
                           _crops_channels_last(features, self.channels_last_out))(features, rois)
        return crops, gw


_warned_ambiguous_jitter = False


class GroundTruthRRoiAlign(Module):
    """The training caller's call (src/ocr_process.py:196-221, :253-267): ground-truth quads of a
    batch -> crops for the recognition loss, ROI rows and pooled width computed on the device.

    forward(features, quads, batch_index, height_jitter=None, keep=None, per_image_jitter=None)
        -> (crops (N, C, pooled_height, pooled_width), rois)
    with pooled_width = ceil(pooled_height * max(w / h)) (:260-263) read back once, as the
    reference does with `.item()`.

    What stays with the caller, as in the reference: WHICH boxes take part.  The reference drops boxes
    whose label starts with '##' and boxes that leave the image (:212-219) and only then truncates the
    list of the whole batch to its first 32 rows (:253-255).  Pass the surviving boxes, or all of them
    with `keep` (N,) bool: the filter is applied first, `max_rois` second -- the reference's order.
    `height_jitter` is the caller's random.randint(-2, 2) (:204): one value per box, or -- what the
    reference draws -- ONE PER IMAGE, a tensor of batch-size length that is then looked up through
    `batch_index`.  Which of the two it is is SAID, not guessed: `per_image_jitter=True` / `False`; left at None
    the length decides only while it can (a length that equals both the number of boxes and the batch
    size -- e.g. one box per image -- is read per box, as rounds 1-3 did, with one warning per process).  Rows whose jittered h is negative yield all-zero crops (the op's
    `pw <= roi_pooled_width` mask is false everywhere, kernel.cu:107); an h of exactly 0 makes the ratio
    infinite -- the reference's `math.ceil` raises there, and so does this module.  A maximal ratio
    <= 0 (every w = 0) would make the reference's pooled width 0 (and its launch fail); it is 1 here.
    """

    def __init__(self, pooled_height=11, spatial_scale=1.0 / 4, max_rois=32, channels_last_out=None):
        super(GroundTruthRRoiAlign, self).__init__()
        self.channels_last_out = channels_last_out  # None: follow the features' layout (see _crops_channels_last)
        self.pooled_height = int(pooled_height)
        self.spatial_scale = float(spatial_scale)
        self.max_rois = max_rois

    def forward(self, features, quads, batch_index=None, height_jitter=None, keep=None, per_image_jitter=None):
        import math
        n = quads.shape[0]
        if height_jitter is not None:
            m = height_jitter.numel()
            if per_image_jitter is None:
                # the length tells only when it cannot be read both ways
                # (rounds 1-3 read a length-n jitter per box; a ValueError here broke callers on exactly the
                # batches where the counts coincide -- e.g. 8 boxes in an 8-image batch.  The per-box reading stays the
                # default, with ONE warning per process.)
                if m == n and batch_index is not None and m == features.shape[0] and n > 1:
                    global _warned_ambiguous_jitter
                    if not _warned_ambiguous_jitter:
                        _warned_ambiguous_jitter = True
                        import warnings
                        warnings.warn("GroundTruthRRoiAlign: height_jitter has one entry per box AND one per image (%d); "
                                      "read per BOX -- pass per_image_jitter=True / False to say which is meant" % m,
                                      stacklevel=2)
                per_image_jitter = m != n
            if per_image_jitter:
                if batch_index is None:
                    raise ValueError("a per-image height_jitter needs batch_index")
                if m != features.shape[0]:
                    raise ValueError("per-image height_jitter: %d entries for %d images" % (m, features.shape[0]))
                height_jitter = height_jitter.reshape(-1)[batch_index.reshape(-1).long()]
            elif m != n:
                raise ValueError("per-box height_jitter: %d entries for %d boxes" % (m, n))
        if keep is not None:
            keep = keep.reshape(-1).bool()
            quads = quads[keep]
            batch_index = None if batch_index is None else batch_index.reshape(-1)[keep]
            height_jitter = None if height_jitter is None else height_jitter.reshape(-1)[keep]
        if self.max_rois is not None:
            quads = quads[: self.max_rois]
            batch_index = None if batch_index is None else batch_index[: self.max_rois]
            height_jitter = None if height_jitter is None else height_jitter[: self.max_rois]
        rois, ratio = _ext.gt_quads_to_rois(quads, batch_index, height_jitter)
        r = float(ratio.item())
        if not math.isfinite(r):
            raise ValueError("degenerate ground-truth box: max(w / h) is %r (a jittered height of 0, or NaN)" % r)
        pooled_width = max(1, math.ceil(self.pooled_height * r))
        return _RRoiAlign(self.pooled_height, pooled_width, self.spatial_scale,
                          _crops_channels_last(features, self.channels_last_out))(features, rois), rois
