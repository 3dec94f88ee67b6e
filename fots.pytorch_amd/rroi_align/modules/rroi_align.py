"""rroi_align.modules.rroi_align -- ``_RRoiAlign`` module.

Same constructor and call as ``rroi_align/modules/rroi_align.py:5-14`` of the
reference: ``_RRoiAlign(pooled_height, pooled_width, spatial_scale)(features, rois)``
with ``rois`` = ``(R, 6)`` rows ``[batch_idx, cx, cy, h, w, angle_deg]`` in
input-image pixels.  One optional keyword beyond the reference: ``channels_last_out=True`` returns
the crops in channels_last storage (same values) for a recognition head that runs in channels_last
-- MIOpen's preferred layout -- so that its first convolution does not relay 256 MiB out again;
the gradient that comes back in channels_last is consumed in place as well.  ``trig=1`` evaluates the
angle's cosine / sine with the device library's fp32 functions (what the reference's sources do when built
for this GPU) instead of the oracle's correctly rounded recipe; forward and backward of a call use the same.
"""
from torch.nn.modules.module import Module

from ..functions.rroi_align import RRoiAlignFunction


class _RRoiAlign(Module):
    def __init__(self, pooled_height, pooled_width, spatial_scale, channels_last_out=False, trig=0):
        super(_RRoiAlign, self).__init__()
        self.trig = int(trig)   # 0 = TRIG_DOUBLE (default), 1 = TRIG_FP32: per call, see _ext.rroi_align
        self.pooled_width = int(pooled_width)
        self.pooled_height = int(pooled_height)
        self.spatial_scale = float(spatial_scale)
        self.channels_last_out = bool(channels_last_out)

    def forward(self, features, rois):
        return RRoiAlignFunction(self.pooled_height, self.pooled_width, self.spatial_scale,
                                 self.channels_last_out, self.trig)(features, rois)

    def extra_repr(self):
        return "pooled_height={}, pooled_width={}, spatial_scale={}".format(
            self.pooled_height, self.pooled_width, self.spatial_scale)
