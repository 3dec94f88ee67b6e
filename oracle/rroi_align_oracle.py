"""oracle/rroi_align_oracle.py -- CPU oracle for the RoIRotate (rroi_align) hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product
(``fots.pytorch_amd/``) never does and fails loudly without its HIP library.

Two independent restatements of the reference algorithm live here:

* ``forward_c / backward_c / forward_literal_c / backward_literal_c`` -- ctypes
  bindings of ``rroi_align_oracle.c`` (gcc, ``-ffp-contract=off``), which follows
  ``/root/reference/rroi_align/src/rroi_align_kernel.cu:28-162`` (forward) and
  ``:193-278`` (backward) statement by statement.
* ``forward_np / backward_np`` -- a vectorised numpy-fp32 restatement of
  SURVEY.md Appendix A written separately from the C file; the two are required
  to agree bit for bit (tests/test_oracle.py), which guards against
  transcription slips in either.

Pinning: tests/test_oracle_kat.py checks the oracle against the only outputs of
the real CUDA op that the reference repository holds
(``rroi_align/data/res{0,1,2}.jpg`` and ``grad.jpg`` written by
``rroi_align/test2.py:87,98``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "librroi_oracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "librroi_ref_hip.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)


def build(force: bool = False) -> str:
    """Compile rroi_align_oracle.c with gcc (see oracle/Makefile)."""
    src = os.path.join(_HERE, "rroi_align_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "_build/librroi_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def build_ref() -> str | None:
    """Build oracle/_ref (reference kernels through hipify-perl + hipcc) where the
    reference checkout exists; returns the path or None."""
    if os.path.isdir("/root/reference/rroi_align/src"):
        subprocess.run(["make", "-C", _HERE, "ref"], check=True, stdout=subprocess.DEVNULL)
    return _REF_PATH if os.path.exists(_REF_PATH) else None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.rroi_oracle_touched_pixels.restype = ctypes.c_long
        _lib.rroi_oracle_max_threads.restype = ctypes.c_int
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(_f32p)


def _prep(features, rois):
    features = np.ascontiguousarray(features, dtype=np.float32)
    rois = np.ascontiguousarray(rois, dtype=np.float32).reshape(-1, 6)
    assert features.ndim == 4
    return features, rois


# --------------------------------------------------------------------------- C oracle
def forward_c(features, rois, pooled_h, pooled_w, scale, threads=1, return_geom=False):
    """Hoisted C oracle; returns out (R,C,PH,PW) [and geom (R,PH,PW,2)]."""
    features, rois = _prep(features, rois)
    B, C, H, W = features.shape
    R = rois.shape[0]
    out = np.empty((R, C, pooled_h, pooled_w), np.float32)
    geom = np.empty((R, pooled_h, pooled_w, 2), np.float32)
    lib().rroi_oracle_forward(_p(features), ctypes.c_float(scale), R, H, W, C, pooled_h, pooled_w,
                              _p(rois), _p(out), _p(geom), int(threads))
    return (out, geom) if return_geom else out


def forward_literal_c(features, rois, pooled_h, pooled_w, scale):
    """Per-element literal C oracle; returns (out, idx_x, idx_y), each (R,C,PH,PW)."""
    features, rois = _prep(features, rois)
    B, C, H, W = features.shape
    R = rois.shape[0]
    shape = (R, C, pooled_h, pooled_w)
    out, ix, iy = (np.zeros(shape, np.float32) for _ in range(3))
    lib().rroi_oracle_forward_literal(_p(features), ctypes.c_float(scale), R, H, W, C, pooled_h,
                                      pooled_w, _p(rois), _p(out), _p(ix), _p(iy))
    return out, ix, iy


def backward_literal_c(grad_out, rois, idx_x, idx_y, feat_shape, scale):
    """Literal backward in ascending index order, fp32 accumulation."""
    B, C, H, W = feat_shape
    grad_out = np.ascontiguousarray(grad_out, np.float32)
    rois = np.ascontiguousarray(rois, np.float32).reshape(-1, 6)
    R, _, PH, PW = grad_out.shape
    gin = np.zeros(feat_shape, np.float32)
    lib().rroi_oracle_backward_literal(_p(grad_out), ctypes.c_float(scale), B, R, H, W, C, PH, PW,
                                       _p(rois), _p(gin), _p(np.ascontiguousarray(idx_x)),
                                       _p(np.ascontiguousarray(idx_y)))
    return gin


def backward_c(grad_out, rois, feat_shape, scale, threads=None):
    """Hoisted backward, double accumulation (order-free reference value).  threads=None: the single-threaded
    statement-order form; threads=T: the channel-partitioned OpenMP form (bit-identical result; bench.py's
    CPU baseline for the backward)."""
    B, C, H, W = feat_shape
    grad_out = np.ascontiguousarray(grad_out, np.float32)
    rois = np.ascontiguousarray(rois, np.float32).reshape(-1, 6)
    R, _, PH, PW = grad_out.shape
    gin = np.empty(feat_shape, np.float32)
    if threads is None:
        lib().rroi_oracle_backward(_p(grad_out), ctypes.c_float(scale), B, R, H, W, C, PH, PW,
                                   _p(rois), _p(gin))
    else:
        lib().rroi_oracle_backward_mt(_p(grad_out), ctypes.c_float(scale), B, R, H, W, C, PH, PW,
                                      _p(rois), _p(gin), int(threads))
    return gin


def touched_pixels(rois, batch, H, W, pooled_h, pooled_w, scale) -> int:
    rois = np.ascontiguousarray(rois, np.float32).reshape(-1, 6)
    return int(lib().rroi_oracle_touched_pixels(ctypes.c_float(scale), batch, rois.shape[0], H, W,
                                                pooled_h, pooled_w, _p(rois)))


def max_threads() -> int:
    return int(lib().rroi_oracle_max_threads())


# --------------------------------------------------------------------------- numpy oracle
_F = np.float32


def _fmax(a, b):  # NaN-dropping like C fmaxf / CUDA max
    return np.fmax(a, b)


def _fmin(a, b):
    return np.fmin(a, b)


def _round_half_away(x):
    """roundf: half away from zero (np.round is half-to-even).  ``|x| - floor|x|`` is
    exact in fp32, so the comparison against 0.5 cannot be fooled by rounding."""
    with np.errstate(invalid="ignore"):
        ax = np.abs(x)
        fl = np.floor(ax)
        mag = np.where(ax - fl >= _F(0.5), fl + _F(1.0), fl)
        r = np.where(np.isfinite(x), np.copysign(mag, x), x)
    return r.astype(np.float32)


def _f2i_sat(x):
    with np.errstate(invalid="ignore"):
        y = np.where(np.isnan(x), 0.0, x.astype(np.float64))
    y = np.clip(y, -2147483648.0, 2147483647.0)
    return y.astype(np.int64)


def geometry_np(rois, pooled_h, pooled_w, scale, H, W):
    """SURVEY.md Appendix A in numpy fp32: returns (batch[R], active[R,PH,PW],
    bin_cx[R,PH,PW], bin_cy[R,PH,PW]); every * and + is one fp32 ufunc."""
    rois = np.ascontiguousarray(rois, np.float32).reshape(-1, 6)
    scale = _F(scale)
    with np.errstate(all="ignore"):
        batch = _f2i_sat(rois[:, 0])
        cx, cy, h, w = rois[:, 1], rois[:, 2], rois[:, 3], rois[:, 4]
        angle = ((rois[:, 5].astype(np.float64) / 180.0) * 3.1415926535).astype(np.float32)
        rpw = (_F(pooled_h) * w) / h
        dx = -rpw / _F(2.0)
        dy = _F(-pooled_h / 2.0)
        Sx = (w * scale) / rpw
        Sy = (h * scale) / _F(pooled_h)
        A = np.cos(angle.astype(np.float64)).astype(np.float32)
        Bt = np.sin(angle.astype(np.float64)).astype(np.float32)
        Dx = cx * scale
        Dy = cy * scale
        m00 = A * Sx
        m01 = Bt * Sy
        m02 = ((m00 * dx) + (m01 * dy)) + Dx
        m10 = (-Bt) * Sx
        m11 = A * Sy
        m12 = ((m10 * dx) + (m11 * dy)) + Dy

        pw = np.arange(pooled_w, dtype=np.float32)[None, None, :]
        ph = np.arange(pooled_h, dtype=np.float32)[None, :, None]
        pw1 = np.arange(1, pooled_w + 1, dtype=np.float32)[None, None, :]
        ph1 = np.arange(1, pooled_h + 1, dtype=np.float32)[None, :, None]

        def bc(v):
            return v[:, None, None]

        def X(a, b):
            return ((bc(m00) * a) + (bc(m01) * b)) + bc(m02)

        def Y(a, b):
            return ((bc(m10) * a) + (bc(m11) * b)) + bc(m12)

        P0, P2, P4, P6 = X(pw, ph), X(pw, ph1), X(pw1, ph), X(pw1, ph1)
        P1, P3, P5, P7 = Y(pw, ph), Y(pw, ph1), Y(pw1, ph), Y(pw1, ph1)
        left = _fmax(_round_half_away(_fmin(_fmin(P0, P2), _fmin(P4, P6))), _F(0.0))
        right = _fmin(_round_half_away(_fmax(_fmax(P0, P2), _fmax(P4, P6))), _F(W) - _F(1.0))
        top = _fmax(_round_half_away(_fmin(_fmin(P1, P3), _fmin(P5, P7))), _F(0.0))
        bottom = _fmin(_round_half_away(_fmax(_fmax(P1, P3), _fmax(P5, P7))), _F(H) - _F(1.0))
        bin_cx = ((left + right) / _F(2.0)).astype(np.float32)
        bin_cy = ((top + bottom) / _F(2.0)).astype(np.float32)
        active = np.broadcast_to(pw, bin_cx.shape) <= bc(rpw)
    return batch, active, bin_cx, bin_cy


def _taps(bin_cx, bin_cy):
    with np.errstate(all="ignore"):
        fx, fy = np.floor(bin_cx), np.floor(bin_cy)
        x0, x1 = _f2i_sat(fx), _f2i_sat(np.ceil(bin_cx))
        y0, y1 = _f2i_sat(fy), _f2i_sat(np.ceil(bin_cy))
        rx = (bin_cx - fx).astype(np.float64)
        ry = (bin_cy - fy).astype(np.float64)
        wlt = ((1.0 - rx) * (1.0 - ry)).astype(np.float32)
        wrt = (rx * (1.0 - ry)).astype(np.float32)
        wrb = (rx * ry).astype(np.float32)
        wlb = ((1.0 - rx) * ry).astype(np.float32)
    return (x0, x1, y0, y1), (wlt, wrt, wrb, wlb)


def forward_np(features, rois, pooled_h, pooled_w, scale):
    features, rois = _prep(features, rois)
    B, C, H, W = features.shape
    R = rois.shape[0]
    batch, active, bcx, bcy = geometry_np(rois, pooled_h, pooled_w, scale, H, W)
    (x0, x1, y0, y1), (wlt, wrt, wrb, wlb) = _taps(bcx, bcy)
    out = np.zeros((R, C, pooled_h, pooled_w), np.float32)

    def tap(n, y, x):
        ok = (y > 0) & (x > 0) & (y < H) & (x < W)
        yy = np.where(ok, y, 0)
        xx = np.where(ok, x, 0)
        v = features[batch[n]][:, yy, xx]  # (C,PH,PW)
        return np.where(ok[None], v, _F(0.0))

    with np.errstate(all="ignore"):
        for n in range(R):
            lt = tap(n, y0[n], x0[n])
            rt = tap(n, y0[n], x1[n])
            lb = tap(n, y1[n], x0[n])
            rb = tap(n, y1[n], x1[n])
            v = np.zeros_like(lt)
            v = v + lt * wlt[n][None]
            v = v + rt * wrt[n][None]
            v = v + rb * wrb[n][None]
            v = v + lb * wlb[n][None]
            out[n] = np.where(active[n][None], v, _F(0.0))
    idx_x = np.where(active, bcx, _F(0.0)).astype(np.float32)
    idx_y = np.where(active, bcy, _F(0.0)).astype(np.float32)
    return out, idx_x, idx_y


def backward_np(grad_out, rois, feat_shape, scale):
    """float64-accumulated backward with the reference's asymmetric bounds
    (kernel.cu:267-274)."""
    B, C, H, W = feat_shape
    grad_out = np.ascontiguousarray(grad_out, np.float32)
    rois = np.ascontiguousarray(rois, np.float32).reshape(-1, 6)
    R, _, PH, PW = grad_out.shape
    # scatter exactly where the forward's mask held: elsewhere the stored centre is (0,0),
    # which fails every bound below (see rroi_oracle_backward in the C file)
    batch, keep, bcx, bcy = geometry_np(rois, PH, PW, scale, H, W)
    (x0, x1, y0, y1), (wlt, wrt, wrb, wlb) = _taps(bcx, bcy)
    acc = np.zeros((B, C, H * W), np.float64)
    conds = [
        ((y0 > 0) & (x0 > 0) & (y0 < H - 1) & (x0 < W - 1), y0, x0, wlt),
        ((y0 > 0) & (x1 < W - 1) & (y0 < H - 1) & (x1 > 0), y0, x1, wrt),
        ((y1 < H - 1) & (x1 < W - 1) & (y1 > 0) & (x1 > 0), y1, x1, wrb),
        ((y1 < H - 1) & (x0 > 0) & (y1 > 0) & (x0 < W - 1), y1, x0, wlb),
    ]
    for n in range(R):
        g = grad_out[n].reshape(C, -1)
        for ok, y, x, wgt in conds:
            m = (ok[n] & keep[n]).reshape(-1)
            if not m.any():
                continue
            lin = (y[n] * W + x[n]).reshape(-1)[m]
            contrib = (wgt[n].reshape(-1)[m][None, :] * g[:, m]).astype(np.float32)
            for c in range(C):
                np.add.at(acc[batch[n], c], lin, contrib[c].astype(np.float64))
    return acc.reshape(B, C, H, W).astype(np.float32)
