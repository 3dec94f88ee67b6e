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


def rois_from_quads(quads, batch_index=None, training=False, target_h=11):
    """(N, 8) fp32 quads [x0,y0,..,x3,y3] -> ((N, 6) rois, (N,) int32 pooled widths)."""
    return _ext.quads_to_rois(quads, batch_index, 1 if training else 0, target_h)


class BatchedRRoiAlign(Module):
    """forward(features, quads[, batch_index]) -> (crops (N, C, target_h, max_gw), target_gw (N,))."""

    def __init__(self, target_h=11, spatial_scale=1.0 / 4, pooled_width=None, channels_last_out=False):
        super(BatchedRRoiAlign, self).__init__()
        self.target_h = int(target_h)
        self.spatial_scale = float(spatial_scale)
        self.pooled_width = pooled_width  # fixed width avoids the one-int device->host read
        self.channels_last_out = bool(channels_last_out)  # crops for a channels_last recognition head

    def forward(self, features, quads, batch_index=None):
        rois, gw = rois_from_quads(quads, batch_index, False, self.target_h)
        width = self.pooled_width
        if width is None:
            width = int(gw.max().item()) if gw.numel() else 64
        crops = _RRoiAlign(self.target_h, width, self.spatial_scale, self.channels_last_out)(features, rois)
        return crops, gw
