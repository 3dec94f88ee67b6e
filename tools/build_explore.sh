#!/bin/sh
# The product translation unit with a mutable Tuning struct and the rroi_align_debug_set_* knobs
# (tools/rroi_align_hip_explore.hip), built next to
# the tools (git-ignored); tools/staged_explore.py and friends load it instead of the product .so.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_explore
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude \
    -Wno-unused-function -o tools/_explore/librroi_align_hip_explore.so \
    tools/rroi_align_hip_explore.hip
