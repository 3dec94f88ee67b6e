"""rroi_align.functions.rroi_align -- autograd surface of RoIRotate.

Mirrors ``rroi_align/functions/rroi_align.py:6-40`` of the reference: an object
constructed as ``RRoiAlignFunction(pooled_height, pooled_width, spatial_scale)``
and called with ``(features, rois)``; gradient w.r.t. ``features`` only
(``:40`` returns ``(grad_input, None)``).  The reference is a legacy
instance-style ``torch.autograd.Function`` (removed in torch >= 1.5); here the
same callable wraps a static Function.  Unlike the reference nothing
output-sized is kept for backward: the bin centres (``ctx.idx_x/idx_y``,
``:19-20``) are a pure function of the rois and are recomputed.
"""
import torch
from torch.autograd import Function

from .._ext import rroi_align


class _RRoiAlignOp(Function):
    # Under torch.autocast the op stays what the reference is -- an fp32 operator: half / bfloat16 features are
    # cast up on the way in (the reference's THCudaTensor signature would reject them), the crops come out fp32 and
    # autograd casts the feature gradient back to the features' dtype.
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, features, rois, pooled_height, pooled_width, spatial_scale, channels_last_out=False,
                trig=rroi_align.TRIG_DOUBLE):
        ctx.pooled_height = pooled_height
        ctx.pooled_width = pooled_width
        ctx.spatial_scale = spatial_scale
        ctx.feature_size = features.size()
        # a channels_last backbone gets its gradient back in channels_last storage
        ctx.channels_last_grad = (features.dim() == 4 and not features.is_contiguous()
                                  and features.is_contiguous(memory_format=torch.channels_last))
        ctx.trig = trig   # the backward recomputes the bin centres: with the forward's recipe
        ctx.save_for_backward(rois)
        return rroi_align.forward(features, rois, pooled_height, pooled_width, spatial_scale,
                                  channels_last_out=channels_last_out, trig=trig)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        (rois,) = ctx.saved_tensors
        grad_input = None
        if ctx.needs_input_grad[0]:
            grad_input = rroi_align.backward(grad_output, rois, ctx.feature_size, ctx.spatial_scale,
                                             channels_last_grad=ctx.channels_last_grad, trig=ctx.trig)
        return grad_input, None, None, None, None, None, None


class RRoiAlignFunction(object):
    """``RRoiAlignFunction(ph, pw, scale)(features, rois) -> (R, C, ph, pw)``."""

    def __init__(self, pooled_height, pooled_width, spatial_scale, channels_last_out=False,
                 trig=rroi_align.TRIG_DOUBLE):
        # extension: the recipe of cos / sin of the ROI angle (kernel.cu:73-74) for this object's calls, forward and
        # backward alike: TRIG_DOUBLE (the oracle's) or TRIG_FP32 (the reference's sources built for this GPU)
        self.trig = int(trig)
        self.pooled_width = pooled_width
        self.pooled_height = pooled_height
        self.spatial_scale = spatial_scale
        # extension: crops in channels_last storage for a channels_last recognition head
        self.channels_last_out = bool(channels_last_out)
        self.feature_size = None
        self.rois = None

    def __call__(self, features, rois):
        self.feature_size = features.size()
        self.rois = rois
        return _RRoiAlignOp.apply(features, rois, int(self.pooled_height), int(self.pooled_width),
                                  float(self.spatial_scale), self.channels_last_out, self.trig)

    # the legacy Function's two methods, callable by hand as in torch 0.4
    def forward(self, features, rois):
        self.feature_size = features.size()
        self.rois = rois
        return rroi_align.forward(features, rois, int(self.pooled_height), int(self.pooled_width),
                                  float(self.spatial_scale), channels_last_out=self.channels_last_out, trig=self.trig)

    def backward(self, grad_output):
        assert self.feature_size is not None and grad_output.is_cuda
        grad_input = rroi_align.backward(grad_output, self.rois, self.feature_size,
                                         float(self.spatial_scale), trig=self.trig)
        return grad_input, None
