/*
 * include/rroi_align_hip.h -- C-ABI of librroi_align_hip.so, the MI355X (gfx950)
 * implementation of the RoIRotate / rroi_align hot path of FOTS.pytorch.
 *
 * Plain pointers and sizes only; no torch types.  Every device pointer is
 * borrowed for the duration of the asynchronous launch; the library never
 * allocates, frees or synchronises.  `stream` is a hipStream_t passed as
 * void* (NULL = the null stream).  All citations below are relative to the
 * reference checkout (chenjun2hao/FOTS.pytorch).
 *
 * Return convention (extends rroi_align/src/rroi_align_cuda.c:23-26,43 and
 * rroi_align_kernel.cu:186): 1 = success, 0 = invalid argument, negative =
 * -(hipError_t) of a failed launch.  The reference's launchers call exit(-1)
 * on a launch error (kernel.cu:179-184); this library never exits.
 */
#ifndef RROI_ALIGN_HIP_H
#define RROI_ALIGN_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------- *
 * 1. The reference's own launcher ABI, symbol for symbol.
 *    Replaces rroi_align/src/rroi_align_kernel.h:8-18 (implemented in the
 *    reference by rroi_align_kernel.cu:164-187 and :280-312), so the
 *    reference's glue rroi_align/src/rroi_align_cuda.c:37-41,80-84 links
 *    against this library unchanged (cudaStream_t -> hipStream_t as void*).
 *
 *    bottom_data : (B, C, H, W) fp32, contiguous NCHW
 *    bottom_rois : (R, 6) fp32 rows [batch_idx, cx, cy, h, w, angle_deg]
 *    top_data, con_idx_x, con_idx_y : (R, C, PH, PW) fp32.  Every element is
 *        written (zeros where the reference leaves its memset untouched), so
 *        the buffers need not be zeroed first; con_idx_* may be NULL.
 * ------------------------------------------------------------------------- */
int RROIAlignForwardLaucher(const float* bottom_data, const float spatial_scale,
                            const int num_rois, const int height, const int width,
                            const int channels, const int pooled_height, const int pooled_width,
                            const float* bottom_rois, float* top_data, float* con_idx_x,
                            float* con_idx_y, void* stream);

/*  top_diff : (R, C, PH, PW) fp32;  bottom_diff : (B, C, H, W) fp32: the gradient is
 *  ADDED to it, whatever the problem size, as the reference's atomicAdds do -- zero it
 *  first for the plain gradient (functions/rroi_align.py:35 does);
 *  con_idx_x/y : the tensors the forward wrote (read per element, as
 *  kernel.cu:232-233 does). */
int RROIAlignBackwardLaucher(const float* top_diff, const float spatial_scale,
                             const int batch_size, const int num_rois, const int height,
                             const int width, const int channels, const int pooled_height,
                             const int pooled_width, const float* bottom_rois,
                             float* bottom_diff, const float* con_idx_x, const float* con_idx_y,
                             void* stream);

/* The two launchers above carry no workspace argument: the library keeps one scratch buffer per (device, stream),
 * grown on demand and reused by later calls on that stream (no allocation per call; a call whose buffer exists only
 * enqueues kernels and can be captured into a HIP graph).  Calls on different streams do not serialise each other.
 * Lifetime: a buffer handed out while its stream was CAPTURING is part of that graph and from then on GRAPH-EXCLUSIVE:
 * never freed afterwards, never handed out again -- later eager calls on the stream get a buffer of their own, later
 * captures take stream-ordered memory their own graph owns -- so a captured launcher call stays replayable for the
 * life of the process, on any stream, concurrently with eager calls (replays of ONE graph must of course be ordered
 * among themselves, as for any graph that writes its own scratch).  To capture with memory the GRAPH owns from the
 * start, capture the first call on a stream that has no buffer yet (or one that is too small): that call allocates
 * inside the capture.  Further launcher calls inside the SAME capture reuse the buffer that capture pinned when it is
 * large enough (0.8.0; before, each took stream-ordered memory of its own).
 * Limits: the table has 64 entries; a pinned buffer keeps its entry for the life of the process, an unpinned one is
 * evicted least-recently-used first.  With every entry pinned (64 captures on 64 streams) a call that finds nothing
 * cached takes stream-ordered memory for that call alone (hipMallocAsync / hipFreeAsync around its launches): correct,
 * slower, and not silent -- rroi_align_launcher_scratch_stats() reports the table's state.
 * A thread that waits for another thread's enqueue on the SAME (device, stream) entry does so without the table lock:
 * calls on different streams never serialise each other.
 * This frees every buffer no graph holds (synchronously); they are re-created on demand.
 * Returns 1 / -hipError. */
int rroi_align_release_launcher_scratch(void);
/* State of the launcher scratch table: *in_use = entries that hold a buffer, *pinned = those a graph owns,
 * *capacity = 64, *transient_calls = launcher calls served by stream-ordered memory of their own so far (no entry to
 * cache in, or a capture that found no buffer large enough).  Any pointer may be NULL.  Returns 1. */
int rroi_align_launcher_scratch_stats(int* in_use, int* pinned, int* capacity, unsigned long long* transient_calls);

/* ------------------------------------------------------------------------- *
 * 2. The MI355X-native entry points used by the Python surface
 *    (rroi_align.functions.rroi_align.RRoiAlignFunction, replacing
 *    rroi_align/functions/rroi_align.py:13-40).  No con_idx tensors: the bin
 *    centres are a pure function of the rois and are recomputed in backward.
 * ------------------------------------------------------------------------- */

/* feature_layout */
#define RROI_LAYOUT_NCHW 0 /* (B, C, H, W) contiguous -- the reference contract */
#define RROI_LAYOUT_NHWC 1 /* (B, H, W, C) contiguous (torch channels_last storage) */

/* path */
#define RROI_PATH_AUTO 0   /* pick by problem size                                  */
#define RROI_PATH_DIRECT 1 /* one kernel, NCHW gather, no workspace (small R)       */
#define RROI_PATH_TILED 2  /* relayout to pixel-major + wave-tiled gather (large R) */
#define RROI_PATH_TILED_ATOMIC 3 /* backward only: the tiled scatter with fp32 atomics (the
                                    default tiled backward is an atomic-free gather)         */
#define RROI_PATH_TILED_LISTS 4  /* backward only: the gather over per-pixel lists built in HBM by
                                    count / scan / fill launches.  AUTO / TILED fall back to it where
                                    the memory cap leaves a bucket below the mean list, unless the
                                    in-kernel gather's rule (below) takes the problem         */
#define RROI_PATH_TILED_BUCKETS 6 /* backward only (round 3): the gather over per-pixel lists built in ONE pass --
                                    fixed-capacity buckets per pixel plus overflow chains: no count pass, no
                                    scan.  What AUTO / TILED run (the bucket grows with the density, 16 ..
                                    4096 entries, under a memory cap); any density is accepted when named    */
#define RROI_PATH_FUSED 7   /* forward only (round 5): ONE launch for few ROIs -- the tiled gather reading the NCHW map
                                    itself (four dword loads per tap instead of one 16-byte load from a relaid-out copy)
                                    and evaluating the affines itself: no prologue launch, no workspace (NULL / 0 is
                                    accepted).  NCHW features and crops only.  What AUTO runs between the direct kernel
                                    (a handful of ROIs) and the two-launch path                                     */
#define RROI_PATH_TILED_INKERNEL 5 /* backward only: the gather that finds each map tile's bins
                                    inside the kernel, no lists in HBM.  Round 2's choice for C <= 64;
                                    since round 3 AUTO / TILED reach it only where the buckets are not
                                    to be had (capped below the mean list, or 32-bit bucket indices
                                    exceeded) AND its own rule holds: at most two channel chunks per
                                    lane, or four with <= 8 bins per map pixel, or eight with <= 1, and
                                    R x B <= 8192.  Named, it runs wherever its 32-bit offsets hold   */

/* path flags (OR into `path`; round 5).
 * RROI_PATH_TRIG_FP32: the one library-dependent step of the arithmetic (rroi_align_kernel.cu:73-74, `cos(angle)` /
 * `sin(angle)` of a float) for THIS call.  Default (bit clear) = RROI_TRIG_DOUBLE: (float)cos((double)angle), the
 * recipe of the oracle -- bit-exact against it and against the reference's sources evaluated with a correctly rounded
 * cosine.  Bit set = RROI_TRIG_FP32: cosf / sinf of the device library, what the reference's own sources call when they
 * are built for this GPU -- bit-exact, every bin, against that build (oracle/_ref/librroi_ref_hip_nofma.so).  The two
 * differ in 15-17 bins per million, at rounding ties.  The recipe is a kernel argument: stream-ordered, capturable
 * into a HIP graph, independent per call -- pass the SAME bit to the backward of a forward (the Python surface keeps
 * it in the autograd context).  Rounds 3-4 had a per-device setter (rroi_align_set_trig_recipe_hip): removed in 0.7.0.
 * The reference-ABI launchers of section 1 have no `path`: they use RROI_TRIG_DOUBLE, or RROI_TRIG_FP32 when the
 * process was started with RROI_ALIGN_LAUNCHER_TRIG=fp32 in its environment (read once, at the first launcher call). */
#define RROI_PATH_TRIG_FP32 0x100
#define RROI_TRIG_DOUBLE 0
#define RROI_TRIG_FP32 1

/* Bytes of scratch the tiled path needs for this problem (0 for the direct
 * path).  The caller owns the scratch; its contents are dead after the call. */
size_t rroi_align_forward_workspace_bytes(int batch_size, int channels, int height, int width,
                                          int num_rois, int feature_layout);
size_t rroi_align_backward_workspace_bytes(int batch_size, int channels, int height, int width,
                                           int num_rois, int pooled_height, int pooled_width);

/* Forward.  top_data (R, C, PH, PW) is fully written.  rois whose batch index
 * falls outside [0, batch_size) produce zeros (the reference reads out of
 * bounds there).  Returns 1 / 0 / -hipError. */
int rroi_align_forward_hip(const float* features, int feature_layout, float spatial_scale,
                           int batch_size, int num_rois, int height, int width, int channels,
                           int pooled_height, int pooled_width, const float* rois,
                           float* top_data, void* workspace, size_t workspace_bytes, int path,
                           void* stream);

/* The same with the layout of the crops stated: RROI_LAYOUT_NCHW (R, C, PH, PW) -- the reference
 * contract, what rroi_align_forward_hip writes -- or RROI_LAYOUT_NHWC = torch channels_last
 * storage (R, PH, PW, C), for a recognition head that runs in channels_last (MIOpen's preferred
 * layout): the first convolution then needs no 256 MiB relayout of its input.  Same values,
 * element for element.  Needs C % 4 == 0; tiled path only. */
int rroi_align_forward_layout_hip(const float* features, int feature_layout, int top_layout,
                                  float spatial_scale, int batch_size, int num_rois, int height,
                                  int width, int channels, int pooled_height, int pooled_width,
                                  const float* rois, float* top_data, void* workspace,
                                  size_t workspace_bytes, int path, void* stream);

/* The same call split into its launches, so a harness can bracket each kernel
 * with events on `stream`: RROI_STAGE_PROLOGUE = relayout + affine table,
 * RROI_STAGE_GATHER = the gather/blend/stream-out kernel (needs the workspace a
 * PROLOGUE call on the same arguments filled).  The direct path has one kernel,
 * run under RROI_STAGE_GATHER. */
#define RROI_STAGE_PROLOGUE 1
#define RROI_STAGE_GATHER 2
#define RROI_STAGE_ALL 3
int rroi_align_forward_stages_hip(const float* features, int feature_layout, float spatial_scale,
                                  int batch_size, int num_rois, int height, int width,
                                  int channels, int pooled_height, int pooled_width,
                                  const float* rois, float* top_data, void* workspace,
                                  size_t workspace_bytes, int path, int stages, void* stream);

/* Backward w.r.t. the features (kernel.cu:193-278 semantics, including its
 * asymmetric border tests :267-274).  bottom_diff (B, C, H, W) NCHW is fully
 * overwritten; it does not have to be zeroed. */
int rroi_align_backward_hip(const float* top_diff, float spatial_scale, int batch_size,
                            int num_rois, int height, int width, int channels,
                            int pooled_height, int pooled_width, const float* rois,
                            float* bottom_diff, void* workspace, size_t workspace_bytes, int path,
                            void* stream);

/* The same with the layouts stated.  top_diff_layout: RROI_LAYOUT_NCHW (R, C, PH, PW) as above, or
 * RROI_LAYOUT_NHWC = torch channels_last storage (R, PH, PW, C) -- what autograd hands over when
 * the recognition head runs in channels_last -- which the gather formulation consumes in place (no
 * relayout pass).  bottom_diff_layout: NCHW (B, C, H, W) as above, or NHWC storage (B, H, W, C),
 * written directly (no relayout back) for a channels_last backbone.  Either NHWC needs
 * C % 4 == 0 and path AUTO, TILED, TILED_LISTS, TILED_BUCKETS or TILED_INKERNEL. */
int rroi_align_backward_layout_hip(const float* top_diff, int top_diff_layout, int bottom_diff_layout,
                                   float spatial_scale, int batch_size, int num_rois, int height,
                                   int width, int channels, int pooled_height, int pooled_width,
                                   const float* rois, float* bottom_diff, void* workspace,
                                   size_t workspace_bytes, int path, void* stream);

/* ------------------------------------------------------------------------- *
 * 3. The callers' ROI construction, on the device.
 *    quads (n, 8) fp32 [x0,y0,x1,y1,x2,y2,x3,y3] -> rois (n, 6) fp32 rows for the
 *    op, plus (optionally) each box's pooled width by the inference rule.
 *      mode 0 replaces tools/ocr_utils.py:133-150 (align_ocr, per detected box)
 *      mode 1 replaces src/ocr_process.py:196-206 (ground-truth quads in training;
 *             the random height jitter of :204 stays with the caller)
 *    batch_index (n) fp32 may be NULL (= 0).  target_gw (n) int32 may be NULL.
 * ------------------------------------------------------------------------- */
int rroi_align_quads_to_rois_hip(const float* quads, const float* batch_index, int n, int mode,
                                 int target_h, float* rois, int* target_gw, void* stream);

/* The training caller's side, src/ocr_process.py:196-219 and :259-263: ground-truth quads (n, 8)
 * -> rois (n, 6) with the reference's double-precision arithmetic, `height_jitter` (n) -- the
 * integer the caller draws with random.randint(-2, 2), :204; NULL = 0 -- added to h in double
 * before the row is rounded to fp32, and *max_ratio = max over the fp32 rows of w / h (NaN if any
 * ratio is NaN; -inf for n = 0): pooled_width = ceil(pooled_height * max_ratio), :260-263.  One
 * launch for the whole batch; a jitter that makes h <= 0 yields the degenerate rows the op handles
 * as the reference does (h < 0: zeros; h == 0: an infinite ratio, at which the reference's own
 * math.ceil raises). */
int rroi_align_gt_quads_to_rois_hip(const float* quads, const float* batch_index, const float* height_jitter,
                                    int n, float* rois, float* max_ratio, void* stream);

/* ------------------------------------------------------------------------- *
 * 4. Detection post-processing, replacing
 *    nms/adaptor.cpp:40-120 + nms/nms.h:116-215 (`nms.get_boxes`, nms/__init__.py:20-29).
 *
 *    rroi_rbox_decode_hip    device: every pixel of the score map `segm` (h, w) above
 *        `segm_thresh` -> one 64-byte candidate record, in raster order: int32 quad[8] in
 *        1/10000 px, float score, float rdist[4] (the pixel's four raw RBOX distances r[0..3]; the corner
 *        confidences expf(-r / 9) of adaptor.cpp:97-100 are formed by rroi_nms_merge_host with the C
 *        library's expf, as the reference does), int32 x, y, pad (adaptor.cpp:76-117).
 *        rbox (4, h, w) and angle (2, h, w) are CHANNELS-FIRST, as the network emits them
 *        (the reference transposes on the host first).  *count receives the number of passing
 *        pixels even when it exceeds `capacity` (records beyond it are dropped), for any `capacity` >= 0.
 *        One workgroup per 1024 pixels.  On a map of more than 262144 pixels a buffer of `capacity` >=
 *        h * w + ceil(h * w / 1024 / 16) records lets the library keep per-slab counts behind the h * w
 *        records it can fill (two short launches); a smaller buffer is served by the one-launch form, in
 *        which every workgroup counts the pixels before its slab itself (slower on such maps, same result).
 *    rroi_nms_merge_host     host (no GPU work): locality-aware merge with `iou_threshold`, then
 *        polygon NMS with `iou_threshold2` (the reference passes 0.4 and 0.2) over `num_candidates`
 *        records in host memory -> boxes (n, 9) fp32 [x0,y0,..,x3,y3 in px, score]; returns the
 *        number of boxes found (writes at most max_boxes), or -1 on an invalid argument.
 * ------------------------------------------------------------------------- */
/* Format of the 64-byte candidate record: 1 = {quad[8], score, probs[4], x, y, pad} (versions <= 0.5),
 * 2 = {quad[8], score, rdist[4], x, y, pad} (0.6.0 on: raw distances, confidences formed by rroi_nms_merge_host).
 * A consumer that reads records itself checks this constant against the library's rroi_nms_record_format(). */
#define RROI_NMS_RECORD_FORMAT 2
int rroi_nms_record_format(void);
int rroi_rbox_decode_hip(const float* segm, const float* rbox, const float* angle, int height, int width,
                         float segm_thresh, void* candidates, int capacity, int* count, void* stream);
int rroi_nms_merge_host(const void* candidates, int num_candidates, int width, int height, float iou_threshold,
                        float iou_threshold2, float* boxes, int max_boxes);

/* Greedy CTC decode of the recognition logits computed from the crops: replaces the per-box
 * `labels_pred.max(1)` + Python loop of tools/ocr_utils.py:183-186 / src/utils.py:87-97
 * (strLabelConverter.decode, raw=False) for all boxes of an image in one launch.
 *   logits      (num_seqs, num_classes, num_steps) fp32, contiguous -- net.forward_ocr's layout
 *   lengths     (num_seqs) int32 valid time steps per sequence, or NULL (= num_steps)
 *   labels      (num_seqs, num_steps) int32 raw arg max per step, or NULL
 *   decoded     (num_seqs, num_steps) int32: label t is kept iff t < length, it is not the blank
 *               (0) and it differs from label t-1; kept labels first, then zeros
 *   decoded_len (num_seqs) int32 number of kept labels
 * arg max = first index of the largest value; NaN counts as largest (torch.max). */
int rroi_ctc_greedy_decode_hip(const float* logits, int num_seqs, int num_classes, int num_steps,
                               const int* lengths, int* labels, int* decoded, int* decoded_len,
                               void* stream);

/* Bin centres only: geom (R, PH, PW, 2) = (bin_cx, bin_cy), 0 where the bin is
 * outside the ROI's pooled width (kernel.cu:86-107).  Diagnostic / test hook. */
int rroi_align_bin_centres_hip(float spatial_scale, int num_rois, int height, int width,
                               int pooled_height, int pooled_width, const float* rois,
                               float* geom, void* stream);
/* ... with the trig recipe stated (RROI_TRIG_DOUBLE / RROI_TRIG_FP32; the function above = RROI_TRIG_DOUBLE). */
int rroi_align_bin_centres_trig_hip(float spatial_scale, int num_rois, int height, int width,
                                    int pooled_height, int pooled_width, const float* rois,
                                    float* geom, int trig_recipe, void* stream);

/* (float)cos((double)x), (float)sin((double)x) of n angles given in degrees,
 * through the same device code the kernels use (test hook for the one
 * library-dependent step of the arithmetic recipe). out = (n, 2). */
int rroi_align_sincos_probe_hip(const float* angle_deg, int n, float* out, void* stream);

/* Measurement hook: fills out[0 .. num_floats) (16-byte aligned, a multiple of 4 floats) with values that differ
 * from store to store, one 16-byte plain store per thread.  bench.py times it on the output buffer as the ceiling
 * a write of non-zero, non-constant data reaches on the box (zeros are written faster on this chip). */
int rroi_align_write_probe_hip(float* out, size_t num_floats, void* stream);

/* DEPRECATED shims of the per-device setter / getter of versions 0.5-0.6 (removed in 0.7.0 without one, ADVICE r05;
 * they go away in 0.9): the recipe travels in `path` now.  set: 1 for RROI_TRIG_DOUBLE (what every call without the
 * RROI_PATH_TRIG_FP32 bit runs: nothing to do), 0 for RROI_TRIG_FP32 or anything else -- the library has no device-wide
 * state left to change, so the request is REFUSED rather than silently ignored; get: RROI_TRIG_DOUBLE. */
int rroi_align_set_trig_recipe_hip(int recipe);
int rroi_align_get_trig_recipe_hip(void);

/* Identification: "rroi_align_hip <version> gfx950" (0.7.0: per-call trig recipe, device-wide setter removed; 0.8.0: launcher scratch reused within a capture, table of 64, stats). */
const char* rroi_align_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RROI_ALIGN_HIP_H */
