"""rroi_align.modules.rroi_align -- ``_RRoiAlign`` module.

Same constructor and call as ``rroi_align/modules/rroi_align.py:5-14`` of the
reference: ``_RRoiAlign(pooled_height, pooled_width, spatial_scale)(features, rois)``
with ``rois`` = ``(R, 6)`` rows ``[batch_idx, cx, cy, h, w, angle_deg]`` in
input-image pixels.
"""
from torch.nn.modules.module import Module

from ..functions.rroi_align import RRoiAlignFunction


class _RRoiAlign(Module):
    def __init__(self, pooled_height, pooled_width, spatial_scale):
        super(_RRoiAlign, self).__init__()
        self.pooled_width = int(pooled_width)
        self.pooled_height = int(pooled_height)
        self.spatial_scale = float(spatial_scale)

    def forward(self, features, rois):
        return RRoiAlignFunction(self.pooled_height, self.pooled_width,
                                 self.spatial_scale)(features, rois)

    def extra_repr(self):
        return "pooled_height={}, pooled_width={}, spatial_scale={}".format(
            self.pooled_height, self.pooled_width, self.spatial_scale)
