// tools/rroi_align_hip_explore.hip -- the EXPLORATION build of the library: the product translation unit with a
// mutable `Tuning` struct, plus the rroi_align_debug_set_* setters.  Nothing of this is compiled into the product
// (fots.pytorch_amd/csrc/Makefile builds rroi_align_hip.hip itself).  tools/build_explore.sh builds
// tools/_explore/librroi_align_hip_explore.so from this file; tools/kbench.hip includes it.
#define RROI_TUNING_QUALIFIER
#include "../fots.pytorch_amd/csrc/rroi_align_hip.hip"

extern "C" {
#include "rroi_explore_setters.h"
}
