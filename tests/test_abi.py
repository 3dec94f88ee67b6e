"""CPU: the C-ABI library loads and exports every symbol include/*.h declares; the
host-side argument checks work; the Python surface refuses to run without a GPU
instead of falling back.  No kernel is launched here."""
import ctypes
import glob
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names += re.findall(r"^\s*(?:const\s+)?(?:int|size_t|char\s*\*|void)\s*\*?\s*(\w+)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_header_lists_expected_entry_points():
    names = declared_functions()
    for n in ("RROIAlignForwardLaucher", "RROIAlignBackwardLaucher", "rroi_align_forward_hip",
              "rroi_align_backward_hip", "rroi_align_forward_workspace_bytes"):
        assert n in names, names


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: the header must compile as C99 (and as C++) with no torch / HIP types."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "h.c"
    src.write_text('#include "rroi_align_hip.h"\nint main(void) { return 0; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", str(src)],
                   check=True)
    if shutil.which("g++"):
        subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I", inc, "-fsyntax-only", "-x", "c++", str(src)],
                       check=True)
    text = open(os.path.join(inc, "rroi_align_hip.h")).read()
    assert "hipStream_t stream" not in text and "#include <hip" not in text and "at::Tensor" not in text


def test_backward_workspace_and_layout_entry_points_are_host_checked():
    from rroi_align._ext import rroi_align as ext
    wb = ext._lib.rroi_align_backward_workspace_bytes
    n = wb(1, 256, 160, 160, 512, 8, 64)
    assert n >= 512 * 8 * (8 * 64 + 1) * 128      # the relaid-out top_diff dominates
    assert wb(1, 256, 160, 160, 512, 0, 64) == 0
    b = ext._lib.rroi_align_backward_layout_hip
    # channels-last needs C % 4 == 0 and a tiled path; unknown layouts are refused -- before any launch
    assert b(None, 1, 0, 0.25, 1, 4, 16, 16, 6, 8, 8, None, None, None, 0, 0, None) == 0
    assert b(None, 1, 0, 0.25, 1, 4, 16, 16, 8, 8, 8, None, None, None, 0, 1, None) == 0
    assert b(None, 7, 0, 0.25, 1, 4, 16, 16, 8, 8, 8, None, None, None, 0, 0, None) == 0
    f = ext._lib.rroi_align_forward_layout_hip
    assert f(None, 0, 1, 0.25, 1, 4, 16, 16, 6, 8, 8, None, None, None, 0, 0, None) == 0
    assert f(None, 0, 1, 0.25, 1, 4, 16, 16, 8, 8, 8, None, None, None, 0, 1, None) == 0


def test_library_exports_every_declared_symbol():
    from rroi_align._ext import rroi_align as ext
    lib = ctypes.CDLL(ext.LIB_PATH)
    for n in declared_functions():
        assert hasattr(lib, n), f"{n} declared in include/ but not exported"
    assert set(declared_functions()) == set(ext.EXPORTS)
    assert "gfx950" in ext.version()


def test_code_object_is_gfx950():
    from rroi_align._ext import rroi_align as ext
    blob = open(ext.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"gfx942" not in blob and b"sm_" not in blob


def test_workspace_query_is_host_only():
    from rroi_align._ext import rroi_align as ext
    n = ext._lib.rroi_align_forward_workspace_bytes(1, 256, 160, 160, 512, ext.LAYOUT_NCHW)
    assert n >= 256 * 160 * 160 * 4 + 512 * 32
    assert n < 1.02 * (256 * 160 * 160 * 4) + 512 * 32 + 4096  # padded row pitch + a spare pixel per slice
    # channels_last with C % 4 == 0 is consumed in place: only the affine table
    assert ext._lib.rroi_align_forward_workspace_bytes(1, 256, 160, 160, 512, ext.LAYOUT_NHWC) < 32768
    assert ext._lib.rroi_align_forward_workspace_bytes(0, 256, 160, 160, 512, 0) == 0


def test_invalid_arguments_return_zero_without_touching_the_gpu():
    from rroi_align._ext import rroi_align as ext
    f = ext._lib.rroi_align_forward_hip
    # bad shape / layout / path -> 0 (reference convention, rroi_align_cuda.c:23-26)
    assert f(None, 0, 1.0, 1, 4, -1, 8, 3, 8, 32, None, None, None, 0, 0, None) == 0
    assert f(None, 7, 1.0, 1, 4, 8, 8, 3, 8, 32, None, None, None, 0, 0, None) == 0
    assert f(None, 0, 1.0, 1, 4, 8, 8, 3, 8, 32, None, None, None, 0, 9, None) == 0
    assert f(None, 0, 1.0, 1, 4, 8, 8, 3, 8, 32, None, None, None, 0, 0, None) == 0  # null pointers
    assert f(None, 0, 1.0, 1, 0, 8, 8, 3, 8, 32, None, None, None, 0, 0, None) == 1  # R == 0 is a no-op
    assert ext._lib.RROIAlignForwardLaucher(None, 1.0, 4, 8, 8, 3, 0, 32, None, None, None, None, None) == 0


def test_cpu_tensors_are_refused_not_emulated():
    from rroi_align.functions.rroi_align import RRoiAlignFunction
    from rroi_align.modules.rroi_align import _RRoiAlign
    feats, rois = torch.zeros(1, 3, 8, 8), torch.zeros(2, 6)
    with pytest.raises(RuntimeError, match="GPU only"):
        _RRoiAlign(8, 32, 1.0)(feats, rois)
    with pytest.raises(RuntimeError, match="GPU only"):
        RRoiAlignFunction(8, 32, 1.0)(feats, rois)


def test_module_surface_matches_reference():
    """rroi_align/modules/rroi_align.py:6-14 and functions/rroi_align.py:7-11."""
    from rroi_align.functions.rroi_align import RRoiAlignFunction
    from rroi_align.modules.rroi_align import _RRoiAlign
    m = _RRoiAlign(11.0, "64", 1 / 4)
    assert (m.pooled_height, m.pooled_width, m.spatial_scale) == (11, 64, 0.25)
    assert isinstance(m, torch.nn.Module) and not list(m.parameters())
    fn = RRoiAlignFunction(8, 64, 0.25)
    assert (fn.pooled_height, fn.pooled_width, fn.spatial_scale, fn.feature_size) == (8, 64, 0.25, None)
    from rroi_align._ext import rroi_align as ext
    assert callable(ext.rroi_align_forward_cuda) and callable(ext.rroi_align_backward_cuda)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under fots.pytorch_amd/ may import,
    link, load or execute it (mentions in comments are fine)."""
    pkg = os.path.join(ROOT, "fots.pytorch_amd")
    bad = re.compile(r"(^\s*(from|import)\s+\S*oracle)|librroi_oracle|oracle/_build|oracle/_ref|"
                     r"CDLL\([^)]*oracle|subprocess[^\n]*oracle", re.M)
    checked = 0
    for path in glob.glob(os.path.join(pkg, "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
            checked += 1
            assert not bad.search(open(path, errors="ignore").read()), path
    assert checked >= 6


def test_python_constants_are_the_headers():
    """PATH_* / LAYOUT_* of the ctypes binding are the #defines of include/rroi_align_hip.h, value for value."""
    import re
    from rroi_align._ext import rroi_align as ext
    text = open(os.path.join(ROOT, "include", "rroi_align_hip.h")).read()
    defs = {m.group(1): int(m.group(2), 0)
            for m in re.finditer(r"#define\s+RROI_((?:PATH|LAYOUT|TRIG)_\w+)\s+(0x[0-9a-fA-F]+|\d+)", text)}
    assert len(defs) >= 12 and "TRIG_FP32" in defs and defs["PATH_TRIG_FP32"] == 0x100
    for name, value in defs.items():
        assert getattr(ext, name) == value, name
    # path values live in the low byte, flags above it
    assert set(ext.BACKWARD_PATHS) | {ext.PATH_FUSED} == {v for k, v in defs.items() if k.startswith("PATH_") and v < 0x100}
    assert set(ext.FORWARD_PATHS) <= set(ext.BACKWARD_PATHS) | {ext.PATH_FUSED}   # (FUSED: forward only)


@pytest.mark.gpu
def test_write_probe_and_trig_recipe_hooks():
    """Round-4 entry points: the bench's write probe fills exactly the floats it is given (values that differ from
    store to store, nothing behind them), refuses misaligned / odd requests.  Round 5: the trig recipe travels in the
    call (`path | RROI_PATH_TRIG_FP32`); the device-wide setter is a deprecated shim without effect."""
    import torch
    from rroi_align._ext import rroi_align as ext
    buf = torch.full((4096 + 8,), float("nan"), device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert ext._lib.rroi_align_write_probe_hip(buf.data_ptr(), 4096, st) == 1
    h = buf.cpu().numpy()
    assert np.isfinite(h[:4096]).all() and np.isnan(h[4096:]).all() and len(np.unique(h[:4096:4])) > 900
    assert ext._lib.rroi_align_write_probe_hip(buf.data_ptr(), 4094, st) == 0       # not a multiple of 4 floats
    assert ext._lib.rroi_align_write_probe_hip(buf.data_ptr() + 4, 4096, st) == 0   # not 16-byte aligned
    assert ext._lib.rroi_align_write_probe_hip(buf.data_ptr(), 0, st) == 1
    # round 5: the trig recipe is a flag bit of the call's `path` (no device-wide setter any more); unknown flag bits
    # and unknown recipes are refused
    # (0.8.0, ADVICE r05: the setter / getter of 0.5-0.6 are back as DEPRECATED shims that refuse what they can no longer
    # do -- there is no device-wide state: TRIG_DOUBLE is acknowledged, TRIG_FP32 is refused, the getter says TRIG_DOUBLE)
    assert ext._lib.rroi_align_set_trig_recipe_hip(ext.TRIG_DOUBLE) == 1
    assert ext._lib.rroi_align_set_trig_recipe_hip(ext.TRIG_FP32) == 0
    assert ext._lib.rroi_align_set_trig_recipe_hip(7) == 0
    assert ext._lib.rroi_align_get_trig_recipe_hip() == ext.TRIG_DOUBLE
    F = torch.randn(1, 4, 16, 16, device="cuda")
    R = torch.tensor([[0, 30, 30, 10, 40, 20.0]], device="cuda")
    a = ext.forward(F, R, 4, 16, 0.25, trig=ext.TRIG_DOUBLE)
    b = ext.forward(F, R, 4, 16, 0.25, trig=ext.TRIG_FP32)
    assert a.shape == b.shape == (1, 4, 4, 16)
    with pytest.raises(ValueError):
        ext.forward(F, R, 4, 16, 0.25, trig=7)
    out = torch.empty(1, 4, 4, 16, device="cuda")
    for bad in (0x200, 0x1000 | ext.PATH_DIRECT):
        assert ext._lib.rroi_align_forward_hip(F.data_ptr(), 0, 0.25, 1, 1, 16, 16, 4, 4, 16, R.data_ptr(), out.data_ptr(),
                                               None, 0, bad, st) == 0
    assert ext._lib.rroi_align_forward_hip(F.data_ptr(), 0, 0.25, 1, 1, 16, 16, 4, 4, 16, R.data_ptr(), out.data_ptr(),
                                           None, 0, ext.PATH_DIRECT | ext.PATH_TRIG_FP32, st) == 1
    assert torch.equal(out, ext.forward(F, R, 4, 16, 0.25, path=ext.PATH_DIRECT, trig=ext.TRIG_FP32))
    geom = torch.empty(1, 4, 16, 2, device="cuda")
    assert ext._lib.rroi_align_bin_centres_trig_hip(0.25, 1, 16, 16, 4, 16, R.data_ptr(), geom.data_ptr(), 5, st) == 0
    assert ext._lib.rroi_nms_record_format() == 2
