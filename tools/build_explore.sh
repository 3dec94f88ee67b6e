#!/bin/sh
# The product library with the rroi_align_debug_set_* knobs exported (-DRROI_EXPLORE), built next to
# the tools (git-ignored); tools/staged_explore.py and friends load it instead of the product .so.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_explore
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude \
    -Wno-unused-function -DRROI_EXPLORE -o tools/_explore/librroi_align_hip_explore.so \
    fots.pytorch_amd/csrc/rroi_align_hip.hip
