// tools/rroi_explore_setters.h -- the rroi_align_debug_set_* knobs of the EXPLORATION build of the library.
// Included by tools/rroi_align_hip_explore.hip BEHIND the product translation unit (which it includes with a mutable
// `Tuning`), inside an extern "C" block; tools/build_explore.sh builds tools/_explore/librroi_align_hip_explore.so.
// They write the `Tuning` struct that the product reads as compile-time constants: the product library neither
// contains nor exports any of this.  Every setter returns the previous value.
#pragma once

// workgroups per CU of the strided split kernel (0: leave)
int rroi_align_debug_set_split_wgs_per_cu(int v)
{
    const int old = g_tune.split_wgs_per_cu;
    if (v > 0) g_tune.split_wgs_per_cu = v;
    return old;
}
// v: 0 never / 1 where it pays / 2 always the SHIFT form; wgs_per_cu: 0 = the host's rule (the third argument, round 3's
// runs per block, is ignored since the tiles overlap: round 4)
int rroi_align_debug_set_fwd_shift(int v, int wgs_per_cu, int)
{
    const int old = g_tune.fwd_shift;
    if (v >= 0) g_tune.fwd_shift = v;
    if (wgs_per_cu >= 0) g_tune.shift_wgs_per_cu = wgs_per_cu;
    return old;
}
int rroi_align_debug_set_bwd_tile_run(int v)
{
    const int old = g_tune.bwd_tile_run;
    g_tune.bwd_tile_run = v;
    return old;
}
int rroi_align_debug_set_bwd_skip_dead(int v)
{
    const int old = g_tune.bwd_skip_dead;
    g_tune.bwd_skip_dead = v;
    return old;
}
int rroi_align_debug_set_bwd_nchw_direct(int v)
{
    const int old = g_tune.bwd_nchw_direct;
    g_tune.bwd_nchw_direct = v;
    return old;
}
int rroi_align_debug_set_bwd_buckets(int v)
{
    const int old = g_tune.bwd_buckets;
    g_tune.bwd_buckets = v;
    return old;
}
// ablations of the gather: 1 = output stores dropped, 2 = every tap out of range (the free-first-item bit 256 and the
// per-workgroup time stamps need tools/experiments/r04_wg_trace_instrumentation.patch)
int rroi_align_debug_set_fwd_dbg(int v)
{
    const int old = g_tune.fwd_dbg;
    g_tune.fwd_dbg = v;
    return old;
}
int rroi_align_debug_set_prologue_blocks(int v)
{
    const int old = g_tune.prologue_blocks_per_cu;
    if (v >= 1 && v <= 64) g_tune.prologue_blocks_per_cu = v;
    return old;
}
int rroi_align_debug_set_row_pad(int v)
{
    const int old = g_tune.row_pad;
    g_tune.row_pad = v;
    return old;
}
int rroi_align_debug_set_waves_per_cu(int v)
{
    const int old = g_tune.waves_per_cu;
    if (v >= 1 && v <= 64) g_tune.waves_per_cu = v;
    return old;
}
// XCD groups of the forward (round 5): 0 = one group (rounds 1-4), 1 = G = 8 / nchunks groups where the chunks divide the XCDs
int rroi_align_debug_set_fwd_groups(int v)
{
    const int old = g_tune.fwd_groups;
    g_tune.fwd_groups = v;
    return old;
}
int rroi_align_debug_set_fwd_groups_min_rois(int v)
{
    const int old = g_tune.fwd_groups_min_rois;
    g_tune.fwd_groups_min_rois = v;
    return old;
}
// AUTO's forward crossovers: one-launch fused form on / off, its lower bound and the two-launch path's, in output elements (<= 0: leave)
int rroi_align_debug_set_fwd_fused(int on, double fused_min, double tiled_min)
{
    const int old = g_tune.fwd_fused;
    g_tune.fwd_fused = on;
    if (fused_min > 0) g_tune.fwd_fused_min_elems = fused_min;
    if (tiled_min > 0) g_tune.fwd_tiled_min_elems = tiled_min;
    return old;
}
int rroi_align_debug_set_bwd_pair_blocks(int v) { const int old = g_tune.bwd_pair_blocks_per_cu; g_tune.bwd_pair_blocks_per_cu = v; return old; }
// the direct path's form: 1 = K2p (patch kernel, round 5), 0 = rounds 1-4's thread-per-bin kernel; waves / cwave: 0 = leave
int rroi_align_debug_set_fwd_patch(int on, int waves, int cwave)
{
    const int old = g_tune.fwd_patch;
    g_tune.fwd_patch = on;
    if (waves > 0) g_tune.fwd_patch_waves = waves;
    if (cwave > 0) g_tune.fwd_patch_cwave = cwave;
    return old;
}
int rroi_align_debug_set_bwd_pair_aggregate(int v) { const int old = g_tune.bwd_pair_aggregate; g_tune.bwd_pair_aggregate = v; return old; }
int rroi_align_debug_set_fwd_merge(int v) { const int old = g_tune.fwd_merge; g_tune.fwd_merge = v; return old; }
int rroi_align_debug_set_fwd_shift_lines(int v, int wgs_per_cu) { const int old = g_tune.fwd_shift_lines; g_tune.fwd_shift_lines = v; if (wgs_per_cu > 0) g_tune.shift_lines_wgs_per_cu = wgs_per_cu; return old; }
