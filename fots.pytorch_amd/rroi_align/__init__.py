"""rroi_align -- RoIRotate (rotated-ROI sampling) for FOTS on MI355X / gfx950.

Drop-in for the reference package of the same name (chenjun2hao/FOTS.pytorch,
``rroi_align/``): the import paths ``rroi_align.modules.rroi_align._RRoiAlign``
and ``rroi_align.functions.rroi_align.RRoiAlignFunction`` and their call
signatures are kept; underneath, a hand-written HIP library
(``_ext/rroi_align/librroi_align_hip.so``, C-ABI in ``include/rroi_align_hip.h``)
replaces the reference's CUDA kernels.  There is no CPU fallback.
"""
__all__ = ["functions", "modules", "sharded"]
