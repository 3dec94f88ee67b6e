"""rroi_align._ext.rroi_align -- ctypes binding of librroi_align_hip.so.

Stands where the reference's cffi extension stood (``rroi_align/build.py:7-37``
built ``_ext.rroi_align`` from ``src/rroi_align_cuda.c``): it exports the same
two tensor-taking functions, ``rroi_align_forward_cuda`` and
``rroi_align_backward_cuda`` (``src/rroi_align_cuda.h:1-7``), plus the native
entry points the autograd Function uses.

The shared library is mandatory: importing this module without it raises.
"""
from __future__ import annotations

import collections
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librroi_align_hip.so")

LAYOUT_NCHW, LAYOUT_NHWC = 0, 1
PATH_AUTO, PATH_DIRECT, PATH_TILED, PATH_TILED_ATOMIC, PATH_TILED_LISTS, PATH_TILED_INKERNEL = 0, 1, 2, 3, 4, 5
PATH_TILED_BUCKETS = 6  # backward only: the gather over pixel lists built in one pass (buckets + overflow chains)
PATH_FUSED = 7  # forward only: one launch for few ROIs, the tiled gather reading the NCHW map itself (no workspace)
# the gather formulations of the backward (they also read / write channels-last tensors in place), and every path
GATHER_PATHS = (PATH_TILED, PATH_TILED_LISTS, PATH_TILED_BUCKETS, PATH_TILED_INKERNEL)
BACKWARD_PATHS = (PATH_AUTO, PATH_DIRECT, PATH_TILED_ATOMIC) + GATHER_PATHS
STAGE_PROLOGUE, STAGE_GATHER, STAGE_ALL = 1, 2, 3
# every value `path` may take in forward() (the parity tests run them all)
FORWARD_PATHS = (PATH_AUTO, PATH_DIRECT, PATH_TILED, PATH_FUSED)

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C fots.pytorch_amd/csrc` "
        "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
        "rroi_align has no CPU or PyTorch fallback.")

# torch must be imported first so that libamdhip64.so.7 resolves to the runtime
# torch already loaded (same SONAME) and streams/pointers are interchangeable.
_lib = ctypes.CDLL(LIB_PATH)

_vp, _f, _i, _sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_size_t

_lib.rroi_align_hip_version.restype = ctypes.c_char_p
_lib.rroi_align_forward_workspace_bytes.restype = _sz
_lib.rroi_align_forward_workspace_bytes.argtypes = [_i] * 6
_lib.rroi_align_backward_workspace_bytes.restype = _sz
_lib.rroi_align_backward_workspace_bytes.argtypes = [_i] * 7
_lib.rroi_align_forward_hip.restype = _i
_lib.rroi_align_forward_hip.argtypes = [_vp, _i, _f, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _i, _vp]
_lib.rroi_align_forward_layout_hip.restype = _i
_lib.rroi_align_forward_layout_hip.argtypes = [_vp, _i, _i, _f, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _i, _vp]
_lib.rroi_align_forward_stages_hip.restype = _i
_lib.rroi_align_forward_stages_hip.argtypes = [_vp, _i, _f, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _i, _i, _vp]
_lib.rroi_align_backward_hip.restype = _i
_lib.rroi_align_backward_hip.argtypes = [_vp, _f, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _i, _vp]
_lib.rroi_align_backward_layout_hip.restype = _i
_lib.rroi_align_backward_layout_hip.argtypes = [_vp, _i, _i, _f, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _i, _vp]
_lib.rroi_align_bin_centres_hip.restype = _i
_lib.rroi_align_bin_centres_hip.argtypes = [_f, _i, _i, _i, _i, _i, _vp, _vp, _vp]
_lib.rroi_align_quads_to_rois_hip.restype = _i
_lib.rroi_align_quads_to_rois_hip.argtypes = [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]
_lib.rroi_align_gt_quads_to_rois_hip.restype = _i
_lib.rroi_align_gt_quads_to_rois_hip.argtypes = [_vp, _vp, _vp, _i, _vp, _vp, _vp]
_lib.rroi_rbox_decode_hip.restype = _i
_lib.rroi_rbox_decode_hip.argtypes = [_vp, _vp, _vp, _i, _i, _f, _vp, _i, _vp, _vp]
_lib.rroi_nms_merge_host.restype = _i
_lib.rroi_nms_merge_host.argtypes = [_vp, _i, _i, _i, _f, _f, _vp, _i]
_lib.rroi_ctc_greedy_decode_hip.restype = _i
_lib.rroi_ctc_greedy_decode_hip.argtypes = [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]
_lib.rroi_align_sincos_probe_hip.restype = _i
_lib.rroi_align_sincos_probe_hip.argtypes = [_vp, _i, _vp, _vp]
_lib.rroi_align_write_probe_hip.restype = _i
_lib.rroi_align_write_probe_hip.argtypes = [_vp, _sz, _vp]
_lib.rroi_align_bin_centres_trig_hip.restype = _i
_lib.rroi_align_bin_centres_trig_hip.argtypes = [_f, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp]
_lib.rroi_nms_record_format.restype = _i
_lib.rroi_nms_record_format.argtypes = []
_lib.RROIAlignForwardLaucher.restype = _i
_lib.RROIAlignForwardLaucher.argtypes = [_vp, _f, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]
_lib.RROIAlignBackwardLaucher.restype = _i
_lib.RROIAlignBackwardLaucher.argtypes = [_vp, _f, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]
_lib.rroi_align_release_launcher_scratch.restype = _i
_lib.rroi_align_release_launcher_scratch.argtypes = []
_lib.rroi_align_launcher_scratch_stats.restype = _i
_lib.rroi_align_launcher_scratch_stats.argtypes = [_vp, _vp, _vp, _vp]

EXPORTS = (
    "RROIAlignForwardLaucher", "RROIAlignBackwardLaucher", "rroi_align_forward_hip",
    "rroi_align_backward_hip", "rroi_align_forward_stages_hip", "rroi_align_forward_workspace_bytes",
    "rroi_align_backward_workspace_bytes", "rroi_align_bin_centres_hip",
    "rroi_align_sincos_probe_hip", "rroi_align_quads_to_rois_hip", "rroi_align_hip_version",
    "rroi_ctc_greedy_decode_hip", "rroi_align_backward_layout_hip", "rroi_align_forward_layout_hip",
    "rroi_align_gt_quads_to_rois_hip", "rroi_rbox_decode_hip", "rroi_nms_merge_host",
    "rroi_align_release_launcher_scratch", "rroi_align_bin_centres_trig_hip", "rroi_nms_record_format",
    "rroi_align_write_probe_hip", "rroi_align_launcher_scratch_stats",
    "rroi_align_set_trig_recipe_hip", "rroi_align_get_trig_recipe_hip",   # deprecated shims (refuse TRIG_FP32)
)


def version() -> str:
    return _lib.rroi_align_hip_version().decode()


# The one library-dependent step of the arithmetic (rroi_align_kernel.cu:73-74): TRIG_DOUBLE (default; the oracle's
# recipe, (float)cos((double)angle)) or TRIG_FP32 (the device library's cosf / sinf -- what the reference's own sources
# evaluate when built for this GPU; bit-exact against that build in every bin).  PER CALL since round 5: the `trig=`
# keyword of forward() / backward() / bin_centres() sets the RROI_PATH_TRIG_FP32 bit of the call's `path`; a kernel
# argument, so two streams may run different recipes at once and either is capturable into a graph.
TRIG_DOUBLE, TRIG_FP32 = 0, 1
PATH_TRIG_FP32 = 0x100


def _path_word(path: int, trig: int) -> int:
    if trig not in (TRIG_DOUBLE, TRIG_FP32):
        raise ValueError(f"trig must be TRIG_DOUBLE (0) or TRIG_FP32 (1), got {trig!r}")
    return int(path) | (PATH_TRIG_FP32 if trig == TRIG_FP32 else 0)


def _check(status: int, what: str) -> None:
    if status == 1:
        return
    if status == 0:
        raise ValueError(f"{what}: invalid argument (shape/layout/workspace)")
    raise RuntimeError(f"{what}: HIP error {-status}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# Scratch for the tiled paths, one buffer per (device, stream), grown on demand and reused: calls on
# one stream are ordered, so the next call may overwrite what the previous one left (the contents
# are dead after a call); another stream gets its own buffer.  Saves the allocator round trip per call.
# What is kept is bounded: a request above _SCRATCH_KEEP_BYTES (the backward of a 4096-ROI problem asks
# for 2.4 GB) is served by a plain allocation that goes back to torch's caching allocator with the call,
# at most _SCRATCH_MAX_STREAMS buffers are kept (least recently used first out; a buffer whose stream
# has been destroyed ages out this way), and the table is guarded by a lock.
_SCRATCH_KEEP_BYTES = 512 << 20
_SCRATCH_MAX_STREAMS = 8
_scratch = collections.OrderedDict()
_scratch_lock = threading.Lock()


def _workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    if nbytes > _SCRATCH_KEEP_BYTES or torch.cuda.is_current_stream_capturing():
        # too large to pin, or a graph owns its memory: nothing of it is cached
        return torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    with _scratch_lock:
        buf = _scratch.get(key)
        if buf is not None and buf.numel() >= nbytes:
            _scratch.move_to_end(key)
            return buf
        buf = None
        _scratch.pop(key, None)           # release the smaller buffer before asking for the larger one
        while len(_scratch) >= _SCRATCH_MAX_STREAMS:
            _scratch.popitem(last=False)
        buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        _scratch[key] = buf
        return buf


def release_workspaces() -> None:
    """Drop the cached scratch buffers -- the Python surface's and the library's own (the per-stream
    scratch of the reference-ABI launchers); all of them are re-created on demand."""
    with _scratch_lock:
        _scratch.clear()
    _check(_lib.rroi_align_release_launcher_scratch(), "rroi_align_release_launcher_scratch")


def launcher_scratch_stats() -> dict:
    """State of the library's scratch table for the reference-ABI launchers (see the header)."""
    u, p, c, t = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_ulonglong()
    _lib.rroi_align_launcher_scratch_stats(ctypes.byref(u), ctypes.byref(p), ctypes.byref(c), ctypes.byref(t))
    return {"in_use": u.value, "pinned": p.value, "capacity": c.value, "transient_calls": t.value}


def _require_cuda_f32(t: torch.Tensor, name: str) -> None:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on {t.device}: rroi_align runs on the GPU only (the reference's CPU "
            "branch, functions/rroi_align.py:22-25, is dead code and is not reproduced)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")


# --------------------------------------------------------------------------- native path
def forward(features: torch.Tensor, rois: torch.Tensor, pooled_height: int, pooled_width: int,
            spatial_scale: float, path: int = PATH_AUTO, channels_last_out: bool = False,
            trig: int = TRIG_DOUBLE) -> torch.Tensor:
    """(B,C,H,W) x (R,6) -> (R,C,PH,PW).  NCHW-contiguous or channels_last features.
    channels_last_out: return the crops in channels_last storage (same values) for a recognition
    head that runs in channels_last; needs C % 4 == 0 and the tiled path.
    trig: TRIG_DOUBLE / TRIG_FP32, for this call (pass the same to backward())."""
    word = _path_word(path, trig)
    _require_cuda_f32(features, "features")
    _require_cuda_f32(rois, "rois")
    if features.dim() != 4:
        raise ValueError(f"features must be (B,C,H,W), got {tuple(features.shape)}")
    if rois.dim() != 2 or rois.size(1) != 6:
        raise ValueError(f"rois must be (R,6) [batch,cx,cy,h,w,angle_deg], got {tuple(rois.shape)}")
    if rois.device != features.device:
        raise ValueError("features and rois must be on the same device")
    B, C, H, W = features.shape
    R = rois.size(0)
    ph, pw = int(pooled_height), int(pooled_width)
    if ph <= 0 or pw <= 0:
        raise ValueError("pooled_height and pooled_width must be positive")
    if features.is_contiguous():
        layout = LAYOUT_NCHW
    elif features.is_contiguous(memory_format=torch.channels_last) and C % 4 == 0 and path not in (PATH_DIRECT, PATH_FUSED):
        layout = LAYOUT_NHWC  # consumed in place: a pixel's channels are already contiguous
    else:
        features, layout = features.contiguous(), LAYOUT_NCHW
    rois = rois.contiguous()
    if channels_last_out and (C % 4 != 0 or path in (PATH_DIRECT, PATH_FUSED)):
        raise ValueError("channels_last_out needs C % 4 == 0 and the tiled path")
    with torch.cuda.device_of(features):
        out = torch.empty((R, C, ph, pw), dtype=torch.float32, device=features.device,
                          memory_format=torch.channels_last if channels_last_out else torch.contiguous_format)
        if R == 0 or out.numel() == 0:
            return out
        nbytes = 0 if path in (PATH_DIRECT, PATH_FUSED) else _lib.rroi_align_forward_workspace_bytes(B, C, H, W, R, layout)
        ws = _workspace(features.device, nbytes)
        st = _lib.rroi_align_forward_layout_hip(features.data_ptr(), layout,
                                                LAYOUT_NHWC if channels_last_out else LAYOUT_NCHW,
                                                float(spatial_scale), B, R, H, W, C, ph, pw, rois.data_ptr(),
                                                out.data_ptr(), ws.data_ptr(), nbytes, word, _stream())
    _check(st, "rroi_align_forward_hip")
    return out


def backward(grad_output: torch.Tensor, rois: torch.Tensor, feature_size, spatial_scale: float,
             path: int = PATH_AUTO, channels_last_grad: bool = False, trig: int = TRIG_DOUBLE) -> torch.Tensor:
    """(R,C,PH,PW) -> grad w.r.t. features (B,C,H,W): NCHW contiguous, or (channels_last_grad, for a
    channels_last backbone; needs C % 4 == 0 and the tiled path) in channels_last storage.
    trig: the recipe the forward of these crops ran with."""
    word = _path_word(path, trig)
    _require_cuda_f32(grad_output, "grad_output")
    _require_cuda_f32(rois, "rois")
    B, C, H, W = (int(v) for v in feature_size)
    if grad_output.dim() != 4 or grad_output.size(1) != C or grad_output.size(0) != rois.size(0):
        raise ValueError("grad_output must be (R,C,PH,PW) matching rois and the feature size")
    R, _, ph, pw = grad_output.shape
    # channels_last storage (R, PH, PW, C) -- what autograd hands over when the recognition head
    # runs in channels_last -- is consumed in place by the gather path: no .contiguous() copy of
    # the 256 MiB, no relayout pass
    layout = LAYOUT_NCHW
    if (not grad_output.is_contiguous() and grad_output.is_contiguous(memory_format=torch.channels_last)
            and C % 4 == 0 and path in (PATH_AUTO,) + GATHER_PATHS and R > 0):
        layout = LAYOUT_NHWC
    else:
        grad_output = grad_output.contiguous()
    rois = rois.contiguous()
    cl_grad = bool(channels_last_grad) and C % 4 == 0 and path in (PATH_AUTO,) + GATHER_PATHS and R > 0
    with torch.cuda.device_of(grad_output):
        grad_in = torch.empty((B, C, H, W), dtype=torch.float32, device=grad_output.device,
                              memory_format=torch.channels_last if cl_grad else torch.contiguous_format)
        if grad_in.numel() == 0:
            return grad_in
        nbytes = 0 if path == PATH_DIRECT else _lib.rroi_align_backward_workspace_bytes(B, C, H, W, R, ph, pw)
        ws = _workspace(grad_output.device, nbytes)
        st = _lib.rroi_align_backward_layout_hip(grad_output.data_ptr(), layout,
                                                 LAYOUT_NHWC if cl_grad else LAYOUT_NCHW, float(spatial_scale),
                                                 B, R, H, W, C, ph, pw, rois.data_ptr(), grad_in.data_ptr(),
                                                 ws.data_ptr(), nbytes, word, _stream())
    _check(st, "rroi_align_backward_hip")
    return grad_in


def bin_centres(rois: torch.Tensor, pooled_height: int, pooled_width: int, spatial_scale: float,
                height: int, width: int, trig: int = TRIG_DOUBLE) -> torch.Tensor:
    """(R,PH,PW,2) sample points (kernel.cu:86-107); zero outside the ROI's pooled width."""
    _require_cuda_f32(rois, "rois")
    rois = rois.contiguous()
    R = rois.size(0)
    with torch.cuda.device_of(rois):
        geom = torch.empty((R, int(pooled_height), int(pooled_width), 2), dtype=torch.float32,
                           device=rois.device)
        st = _lib.rroi_align_bin_centres_trig_hip(float(spatial_scale), R, int(height), int(width),
                                                  int(pooled_height), int(pooled_width),
                                                  rois.data_ptr(), geom.data_ptr(), int(trig), _stream())
    _check(st, "rroi_align_bin_centres_trig_hip")
    return geom


def ctc_greedy_decode(logits: torch.Tensor, lengths=None, return_labels: bool = False):
    """(N, nclass, T) fp32 logits -> (decoded (N, T) int32 zero-padded, decoded_len (N,) int32
    [, raw arg-max labels (N, T) int32]), all on the device; one launch for all sequences.
    Replaces `labels_pred.max(1)` + strLabelConverter.decode (tools/ocr_utils.py:183-186)."""
    _require_cuda_f32(logits, "logits")
    if logits.dim() != 3:
        raise ValueError("logits must be (N, nclass, T)")
    logits = logits.contiguous()
    N, K, T = logits.shape
    if K == 0:
        raise ValueError("logits must have at least one class")
    with torch.cuda.device_of(logits):
        dev = logits.device
        if lengths is not None:
            lengths = torch.as_tensor(lengths, dtype=torch.int32, device=dev).contiguous()
            if lengths.numel() != N:
                raise ValueError("lengths must have one entry per sequence")
        decoded = torch.empty((N, T), dtype=torch.int32, device=dev)
        dlen = torch.empty((N,), dtype=torch.int32, device=dev)
        labels = torch.empty((N, T), dtype=torch.int32, device=dev) if return_labels else None
        st = _lib.rroi_ctc_greedy_decode_hip(
            logits.data_ptr(), N, K, T, lengths.data_ptr() if lengths is not None else None,
            labels.data_ptr() if labels is not None else None, decoded.data_ptr(), dlen.data_ptr(),
            _stream())
    _check(st, "rroi_ctc_greedy_decode_hip")
    return (decoded, dlen, labels) if return_labels else (decoded, dlen)


def quads_to_rois(quads: torch.Tensor, batch_index=None, mode: int = 0, target_h: int = 11):
    """(N, 8) quads -> ((N, 6) rois, (N,) int32 pooled widths), both on the device."""
    _require_cuda_f32(quads, "quads")
    quads = quads.contiguous().view(-1, 8)
    n = quads.size(0)
    if batch_index is not None:
        _require_cuda_f32(batch_index, "batch_index")
        batch_index = batch_index.contiguous().view(-1)
        if batch_index.numel() != n:
            raise ValueError("batch_index must have one entry per quad")
    with torch.cuda.device_of(quads):
        rois = torch.empty((n, 6), dtype=torch.float32, device=quads.device)
        gw = torch.empty((n,), dtype=torch.int32, device=quads.device)
        st = _lib.rroi_align_quads_to_rois_hip(quads.data_ptr(),
                                               batch_index.data_ptr() if batch_index is not None else None,
                                               n, int(mode), int(target_h), rois.data_ptr(), gw.data_ptr(),
                                               _stream())
    _check(st, "rroi_align_quads_to_rois_hip")
    return rois, gw


def gt_quads_to_rois(quads: torch.Tensor, batch_index=None, height_jitter=None):
    """(N, 8) ground-truth quads -> ((N, 6) rois, (1,) max w/h over the fp32 rows), on the device
    (src/ocr_process.py:196-219, :259-263)."""
    _require_cuda_f32(quads, "quads")
    quads = quads.contiguous().view(-1, 8)
    n = quads.size(0)
    aux = []
    for t, name in ((batch_index, "batch_index"), (height_jitter, "height_jitter")):
        if t is not None:
            _require_cuda_f32(t, name)
            t = t.contiguous().view(-1)
            if t.numel() != n:
                raise ValueError(f"{name} must have one entry per quad")
        aux.append(t)
    with torch.cuda.device_of(quads):
        rois = torch.empty((n, 6), dtype=torch.float32, device=quads.device)
        ratio = torch.empty((1,), dtype=torch.float32, device=quads.device)
        st = _lib.rroi_align_gt_quads_to_rois_hip(quads.data_ptr(),
                                                  aux[0].data_ptr() if aux[0] is not None else None,
                                                  aux[1].data_ptr() if aux[1] is not None else None,
                                                  n, rois.data_ptr(), ratio.data_ptr(), _stream())
    _check(st, "rroi_align_gt_quads_to_rois_hip")
    return rois, ratio


def sincos_probe(angle_deg: torch.Tensor) -> torch.Tensor:
    _require_cuda_f32(angle_deg, "angle_deg")
    angle_deg = angle_deg.contiguous().view(-1)
    with torch.cuda.device_of(angle_deg):
        out = torch.empty((angle_deg.numel(), 2), dtype=torch.float32, device=angle_deg.device)
        st = _lib.rroi_align_sincos_probe_hip(angle_deg.data_ptr(), angle_deg.numel(),
                                              out.data_ptr(), _stream())
    _check(st, "rroi_align_sincos_probe_hip")
    return out


# --------------------------------------------------------------------------- reference FFI names
def rroi_align_forward_cuda(pooled_height, pooled_width, spatial_scale, features, rois, output,
                            idx_x, idx_y) -> int:
    """Same name, argument order and return value as the reference's FFI function
    (src/rroi_align_cuda.c:7-44): fills ``output``, ``idx_x``, ``idx_y`` in place;
    returns 0 when ``rois.size(1) != 6`` (:22-26), else 1."""
    for t, name in ((features, "features"), (rois, "rois"), (output, "output"), (idx_x, "idx_x"),
                    (idx_y, "idx_y")):
        _require_cuda_f32(t, name)
        if not t.is_contiguous():
            raise ValueError(f"{name} must be contiguous (the reference reads raw storage)")
    if rois.size(1) != 6:
        return 0
    num_rois = rois.size(0)
    _, C, H, W = features.shape
    with torch.cuda.device_of(features):
        st = _lib.RROIAlignForwardLaucher(features.data_ptr(), float(spatial_scale), num_rois, H, W,
                                          C, int(pooled_height), int(pooled_width),
                                          rois.data_ptr(), output.data_ptr(), idx_x.data_ptr(),
                                          idx_y.data_ptr(), _stream())
    _check(st, "RROIAlignForwardLaucher")
    return 1


def rroi_align_backward_cuda(pooled_height, pooled_width, spatial_scale, top_grad, rois,
                             bottom_grad, idx_x, idx_y) -> int:
    """Reference FFI signature (src/rroi_align_cuda.c:49-87).  ``bottom_grad`` must be
    zero on entry, as functions/rroi_align.py:35 guarantees in the reference."""
    for t, name in ((top_grad, "top_grad"), (rois, "rois"), (bottom_grad, "bottom_grad"),
                    (idx_x, "idx_x"), (idx_y, "idx_y")):
        _require_cuda_f32(t, name)
        if not t.is_contiguous():
            raise ValueError(f"{name} must be contiguous (the reference reads raw storage)")
    if rois.size(1) != 6:
        return 0
    num_rois = rois.size(0)
    B, C, H, W = bottom_grad.shape
    with torch.cuda.device_of(top_grad):
        st = _lib.RROIAlignBackwardLaucher(top_grad.data_ptr(), float(spatial_scale), B, num_rois, H,
                                           W, C, int(pooled_height), int(pooled_width),
                                           rois.data_ptr(), bottom_grad.data_ptr(),
                                           idx_x.data_ptr(), idx_y.data_ptr(), _stream())
    _check(st, "RROIAlignBackwardLaucher")
    return 1
