/*
 * oracle/rroi_align_oracle.c -- CPU restatement of the RoIRotate (rroi_align) hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (fots.pytorch_amd/) may
 * import, link or call this file; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, as the checker / reported baseline.
 *
 * What it restates (all citations relative to /root/reference):
 *   forward   rroi_align/src/rroi_align_kernel.cu:28-162   (RROIAlignForward)
 *   backward  rroi_align/src/rroi_align_kernel.cu:193-278  (RROIAlignBackward)
 *   host glue rroi_align/functions/rroi_align.py:13-40     (zero-filled buffers,
 *             grad only w.r.t. features)
 *
 * Arithmetic recipe: every fp32 `*` and `+` is a
 * separately rounded operation in source order -- build with
 * -ffp-contract=off, never -ffast-math.  The reference source's double
 * promotions are kept where they change bits (the degree->radian conversion,
 * kernel.cu:65) and dropped where they are exact (the /2.0 halvings, the
 * clamps against 0.0 / W-1.0).  cos/sin: the CUDA binary used cosf/sinf of
 * the CUDA math library (<=2 ulp, not reproducible off NVIDIA hardware); this
 * build's recipe for both oracle and HIP kernel is the correctly-rounded-in-
 * practice (float)cos((double)angle), (float)sin((double)angle).
 *
 * Pinning: (1) tests/test_oracle_kat.py -- the only outputs of the real CUDA op
 * the reference holds (rroi_align/data/res{0,1,2}.jpg, grad.jpg written by
 * rroi_align/test2.py:87,98); (2) tests/test_oracle_refhip.py -- outputs of the
 * reference's own kernels (rroi_align_kernel.cu through ROCm's hipify-perl,
 * oracle/Makefile: ref) run on an MI355X, reproduced bit for bit (forward,
 * con_idx_x/y); (3) an independent numpy restatement
 * (oracle/rroi_align_oracle.py) that must agree bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#if defined(__FAST_MATH__)
#error "the oracle must not be built with -ffast-math"
#endif
/* contraction is disabled by the build recipe (oracle/Makefile: -ffp-contract=off) */

/* float -> int conversion with the semantics of the device instruction the
 * reference's `(int)floor(x)` lowers to (cvt.rzi.s32.f32 on NVIDIA,
 * v_cvt_i32_f32 on CDNA): saturating, NaN -> 0.  Plain C casts are UB there. */
static inline int f2i_sat(float x)
{
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int)x;
}

/* CUDA max()/min() on (float,double) resolve to fmax/fmin: a NaN operand is
 * dropped (kernel.cu:97-100; SURVEY.md section 7 "Degenerate ROIs"). */
static inline float fmax_nan(float a, float b) { return fmaxf(a, b); }
static inline float fmin_nan(float a, float b) { return fminf(a, b); }

typedef struct {
    float M[2][3];          /* kernel.cu:78-84 */
    float roi_pooled_width; /* kernel.cu:68    */
    int batch;              /* kernel.cu:60    */
} rroi_affine_t;

/* kernel.cu:58-84 -- "TransformPrepare": 2x3 affine of one rotated ROI. */
void rroi_oracle_affine(const float* roi, int pooled_height, float spatial_scale,
                        rroi_affine_t* out)
{
    const int roi_batch_ind = f2i_sat(roi[0]);                            /* :60 */
    const float cx = roi[1], cy = roi[2], h = roi[3], w = roi[4];         /* :61-64 */
    const float angle = (float)(((double)roi[5] / 180.0) * 3.1415926535); /* :65 */

    const float rpw = ((float)pooled_height * w) / h;                     /* :68 */
    const float dx = -rpw / 2.0f;                                         /* :69 exact */
    const float dy = (float)(-pooled_height / 2.0);                       /* :70 exact */
    const float Sx = (w * spatial_scale) / rpw;                           /* :71 */
    const float Sy = (h * spatial_scale) / (float)pooled_height;          /* :72 */
    const float Alpha = (float)cos((double)angle);                        /* :73 */
    const float Beta = (float)sin((double)angle);                         /* :74 */
    const float Dx = cx * spatial_scale;                                  /* :75 */
    const float Dy = cy * spatial_scale;                                  /* :76 */

    const float m00 = Alpha * Sx;                                         /* :79 */
    const float m01 = Beta * Sy;                                          /* :80 */
    const float m10 = (-Beta) * Sx;                                       /* :82 */
    const float m11 = Alpha * Sy;                                         /* :83 */
    out->M[0][0] = m00;
    out->M[0][1] = m01;
    out->M[0][2] = ((m00 * dx) + (m01 * dy)) + Dx;                        /* :81 */
    out->M[1][0] = m10;
    out->M[1][1] = m11;
    out->M[1][2] = ((m10 * dx) + (m11 * dy)) + Dy;                        /* :84 */
    out->roi_pooled_width = rpw;
    out->batch = roi_batch_ind;
}

/* kernel.cu:86-107 -- sample point of bin (ph,pw): centre of the rounded,
 * clamped bounding box of the bin's four transformed corners.  Returns the
 * in_rroi mask (:107). */
static inline int bin_centre(const rroi_affine_t* A, int ph, int pw, int height, int width,
                             float* bin_cx, float* bin_cy)
{
    const float fpw = (float)pw, fph = (float)ph;
    const float fpw1 = (float)(pw + 1), fph1 = (float)(ph + 1);
    float P[8];
    P[0] = ((A->M[0][0] * fpw) + (A->M[0][1] * fph)) + A->M[0][2];        /* :87 */
    P[1] = ((A->M[1][0] * fpw) + (A->M[1][1] * fph)) + A->M[1][2];
    P[2] = ((A->M[0][0] * fpw) + (A->M[0][1] * fph1)) + A->M[0][2];
    P[3] = ((A->M[1][0] * fpw) + (A->M[1][1] * fph1)) + A->M[1][2];
    P[4] = ((A->M[0][0] * fpw1) + (A->M[0][1] * fph)) + A->M[0][2];
    P[5] = ((A->M[1][0] * fpw1) + (A->M[1][1] * fph)) + A->M[1][2];
    P[6] = ((A->M[0][0] * fpw1) + (A->M[0][1] * fph1)) + A->M[0][2];
    P[7] = ((A->M[1][0] * fpw1) + (A->M[1][1] * fph1)) + A->M[1][2];      /* :94 */

    const float leftMost = fmax_nan(roundf(fmin_nan(fmin_nan(P[0], P[2]), fmin_nan(P[4], P[6]))), 0.0f);
    const float rightMost = fmin_nan(roundf(fmax_nan(fmax_nan(P[0], P[2]), fmax_nan(P[4], P[6]))),
                                     (float)width - 1.0f);
    const float topMost = fmax_nan(roundf(fmin_nan(fmin_nan(P[1], P[3]), fmin_nan(P[5], P[7]))), 0.0f);
    const float bottomMost = fmin_nan(roundf(fmax_nan(fmax_nan(P[1], P[3]), fmax_nan(P[5], P[7]))),
                                      (float)height - 1.0f);              /* :97-100 */
    *bin_cx = (leftMost + rightMost) / 2.0f;                              /* :104 exact */
    *bin_cy = (topMost + bottomMost) / 2.0f;                              /* :105 */
    return fpw <= A->roi_pooled_width;                                    /* :107 */
}

/* kernel.cu:128-134 (forward) and :245-251 (backward): the weights are formed
 * in double (the 1.0 literals promote) and rounded to fp32 once.  rx, ry are
 * always 0 or 0.5 (bin centres are halves of integer sums) or NaN (infinite
 * centre), so this equals the plain fp32 evaluation bit for bit. */
static inline void tap_weights(float bin_cx, float bin_cy, float* wlt, float* wrt, float* wrb,
                               float* wlb)
{
    const float rx = bin_cx - floorf(bin_cx);
    const float ry = bin_cy - floorf(bin_cy);
    *wlt = (float)((1.0 - (double)rx) * (1.0 - (double)ry));
    *wrt = (float)((double)rx * (1.0 - (double)ry));
    *wrb = (float)((double)rx * (double)ry);
    *wlb = (float)((1.0 - (double)rx) * (double)ry);
}

/* kernel.cu:110-141 -- the 4-tap blend at (bin_cx, bin_cy) on one channel plane. */
static inline float blend(const float* plane, int height, int width, float bin_cx, float bin_cy)
{
    const int bin_l = f2i_sat(floorf(bin_cx));
    const int bin_r = f2i_sat(ceilf(bin_cx));
    const int bin_t = f2i_sat(floorf(bin_cy));
    const int bin_b = f2i_sat(ceilf(bin_cy));

    float lt = 0.0f, rt = 0.0f, lb = 0.0f, rb = 0.0f;
    if (bin_t > 0 && bin_l > 0 && bin_t < height && bin_l < width) lt = plane[bin_t * width + bin_l];
    if (bin_t > 0 && bin_r > 0 && bin_t < height && bin_r < width) rt = plane[bin_t * width + bin_r];
    if (bin_b > 0 && bin_l > 0 && bin_b < height && bin_l < width) lb = plane[bin_b * width + bin_l];
    if (bin_b > 0 && bin_r > 0 && bin_b < height && bin_r < width) rb = plane[bin_b * width + bin_r];

    float wlt, wrt, wrb, wlb;
    tap_weights(bin_cx, bin_cy, &wlt, &wrt, &wrb, &wlb);

    float v = 0.0f;
    v += lt * wlt;                                                        /* :138 */
    v += rt * wrt;
    v += rb * wrb;
    v += lb * wlb;                                                        /* :141 */
    return v;
}

/*
 * Literal per-element forward: the body of RROIAlignForward run for
 * index = 0..nthreads-1, geometry recomputed for every output element exactly
 * as the reference does, results ACCUMULATED into top_data / con_idx_x /
 * con_idx_y (the reference's atomicAdd onto buffers functions/rroi_align.py:17-20
 * zero-filled).  Caller passes zeroed buffers.  con_idx_* may be NULL.
 */
void rroi_oracle_forward_literal(const float* bottom_data, float spatial_scale, int num_rois,
                                 int height, int width, int channels, int pooled_height,
                                 int pooled_width, const float* bottom_rois, float* top_data,
                                 float* con_idx_x, float* con_idx_y)
{
    const long nthreads = (long)num_rois * pooled_height * pooled_width * channels;
    for (long index = 0; index < nthreads; ++index) {
        long n = index;
        const int pw = (int)(n % pooled_width);
        n /= pooled_width;
        const int ph = (int)(n % pooled_height);
        n /= pooled_height;
        const int c = (int)(n % channels);
        n /= channels;

        rroi_affine_t A;
        rroi_oracle_affine(bottom_rois + n * 6, pooled_height, spatial_scale, &A);
        float bin_cx, bin_cy;
        if (!bin_centre(&A, ph, pw, height, width, &bin_cx, &bin_cy)) continue; /* :151-159 */
        const float* plane = bottom_data + ((long)A.batch * channels + c) * height * width; /* :102 */
        top_data[index] += blend(plane, height, width, bin_cx, bin_cy);
        if (con_idx_x) con_idx_x[index] += bin_cx;
        if (con_idx_y) con_idx_y[index] += bin_cy;
    }
}

/*
 * Hoisted forward: geometry once per (roi, ph, pw), channel loop inside.
 * Bit-identical to the literal form (the geometry has no channel dependence).
 * Writes EVERY output element (zeros where the reference leaves its memset
 * untouched).  geom (optional) receives (R, PH, PW, 2) = (bin_cx, bin_cy),
 * zero where masked.  OpenMP over ROIs when built with -fopenmp and
 * num_threads > 1.
 */
void rroi_oracle_forward(const float* bottom_data, float spatial_scale, int num_rois, int height,
                         int width, int channels, int pooled_height, int pooled_width,
                         const float* bottom_rois, float* top_data, float* geom, int num_threads)
{
    const long bins = (long)pooled_height * pooled_width;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads > 0 ? num_threads : 1)
#endif
    for (int n = 0; n < num_rois; ++n) {
        rroi_affine_t A;
        rroi_oracle_affine(bottom_rois + (long)n * 6, pooled_height, spatial_scale, &A);
        float* out_n = top_data + (long)n * channels * bins;
        for (int ph = 0; ph < pooled_height; ++ph) {
            for (int pw = 0; pw < pooled_width; ++pw) {
                float bin_cx, bin_cy;
                const int in_rroi = bin_centre(&A, ph, pw, height, width, &bin_cx, &bin_cy);
                const long b = (long)ph * pooled_width + pw;
                if (geom) {
                    geom[((long)n * bins + b) * 2 + 0] = in_rroi ? bin_cx : 0.0f;
                    geom[((long)n * bins + b) * 2 + 1] = in_rroi ? bin_cy : 0.0f;
                }
                for (int c = 0; c < channels; ++c) {
                    float v = 0.0f;
                    if (in_rroi) {
                        const float* plane =
                            bottom_data + ((long)A.batch * channels + c) * height * width;
                        v = 0.0f + blend(plane, height, width, bin_cx, bin_cy);
                    }
                    out_n[(long)c * bins + b] = v;
                }
            }
        }
    }
    (void)num_threads;
}

/* kernel.cu:245-274 -- scatter of one output-gradient element. */
static inline void scatter(float* plane, int height, int width, float bin_cx, float bin_cy, float g)
{
    float wlt, wrt, wrb, wlb;
    tap_weights(bin_cx, bin_cy, &wlt, &wrt, &wrb, &wlb);
    const int min_x = f2i_sat(floorf(bin_cx));
    const int max_x = f2i_sat(ceilf(bin_cx));
    const int min_y = f2i_sat(floorf(bin_cy));
    const int max_y = f2i_sat(ceilf(bin_cy));
    const float v1 = wlt * g, v2 = wrt * g, v3 = wrb * g, v4 = wlb * g;
    /* the reference's asymmetric bounds: last row/column excluded too (:267-274) */
    if (min_y > 0 && min_x > 0 && min_y < height - 1 && min_x < width - 1)
        plane[min_y * width + min_x] += v1;
    if (min_y > 0 && max_x < width - 1 && min_y < height - 1 && max_x > 0)
        plane[min_y * width + max_x] += v2;
    if (max_y < height - 1 && max_x < width - 1 && max_y > 0 && max_x > 0)
        plane[max_y * width + max_x] += v3;
    if (max_y < height - 1 && min_x > 0 && max_y > 0 && min_x < width - 1)
        plane[max_y * width + min_x] += v4;
}

/*
 * Literal backward: body of RROIAlignBackward for index = 0..nthreads-1 in
 * ascending index order (the reference's atomicAdd order is unspecified; this
 * is one valid serialisation).  bottom_diff must be zeroed by the caller
 * (functions/rroi_align.py:35).  Reads the bin centres from con_idx_x/y as the
 * reference does (:232-233).
 */
void rroi_oracle_backward_literal(const float* top_diff, float spatial_scale, int batch_size,
                                  int num_rois, int height, int width, int channels,
                                  int pooled_height, int pooled_width, const float* bottom_rois,
                                  float* bottom_diff, const float* con_idx_x,
                                  const float* con_idx_y)
{
    (void)spatial_scale;
    (void)batch_size;
    const long nthreads = (long)num_rois * pooled_height * pooled_width * channels;
    for (long index = 0; index < nthreads; ++index) {
        long n = index;
        const int pw = (int)(n % pooled_width);
        n /= pooled_width;
        n /= pooled_height;
        const int c = (int)(n % channels);
        n /= channels;
        const float* roi = bottom_rois + n * 6;
        const int roi_batch_ind = f2i_sat(roi[0]);
        const float h = roi[3], w = roi[4];
        const float rpw = ((float)pooled_height * w) / h;                 /* :236 */
        if ((float)pw > rpw) continue;                                    /* :238 */
        float* plane = bottom_diff + ((long)roi_batch_ind * channels + c) * height * width;
        scatter(plane, height, width, con_idx_x[index], con_idx_y[index], top_diff[index]);
    }
}

/*
 * Hoisted backward from the rois alone (geometry recomputed, which yields the
 * same bin centres the forward stored).  Accumulates in double per feature
 * element to give a summation-order-free reference value; the final result is
 * rounded to fp32.  bottom_diff is overwritten.
 */
void rroi_oracle_backward(const float* top_diff, float spatial_scale, int batch_size, int num_rois,
                          int height, int width, int channels, int pooled_height,
                          int pooled_width, const float* bottom_rois, float* bottom_diff)
{
    const long plane_sz = (long)height * width;
    const long total = (long)batch_size * channels * plane_sz;
    const long bins = (long)pooled_height * pooled_width;
    double* acc = (double*)calloc((size_t)total, sizeof(double));
    for (int n = 0; n < num_rois; ++n) {
        rroi_affine_t A;
        rroi_oracle_affine(bottom_rois + (long)n * 6, pooled_height, spatial_scale, &A);
        for (int ph = 0; ph < pooled_height; ++ph) {
            for (int pw = 0; pw < pooled_width; ++pw) {
                float bin_cx, bin_cy;
                /* :232-242 -- the backward reads the centre the forward stored.  Where the
                 * forward's mask (pw <= roi_pooled_width, :107) was false nothing was stored,
                 * the zero-filled buffer yields centre (0,0), and (0,0) fails every bound of
                 * :267-274; where the mask was true, !(pw > roi_pooled_width) holds as well.
                 * So the scatter happens exactly on the forward's mask. */
                if (!bin_centre(&A, ph, pw, height, width, &bin_cx, &bin_cy)) continue;
                float wlt, wrt, wrb, wlb;
                tap_weights(bin_cx, bin_cy, &wlt, &wrt, &wrb, &wlb);
                const int min_x = f2i_sat(floorf(bin_cx)), max_x = f2i_sat(ceilf(bin_cx));
                const int min_y = f2i_sat(floorf(bin_cy)), max_y = f2i_sat(ceilf(bin_cy));
                const int ok1 = min_y > 0 && min_x > 0 && min_y < height - 1 && min_x < width - 1;
                const int ok2 = min_y > 0 && max_x < width - 1 && min_y < height - 1 && max_x > 0;
                const int ok3 = max_y < height - 1 && max_x < width - 1 && max_y > 0 && max_x > 0;
                const int ok4 = max_y < height - 1 && min_x > 0 && max_y > 0 && min_x < width - 1;
                const long b = (long)ph * pooled_width + pw;
                for (int c = 0; c < channels; ++c) {
                    const float g = top_diff[((long)n * channels + c) * bins + b];
                    double* plane = acc + ((long)A.batch * channels + c) * plane_sz;
                    if (ok1) plane[min_y * width + min_x] += (double)(wlt * g);
                    if (ok2) plane[min_y * width + max_x] += (double)(wrt * g);
                    if (ok3) plane[max_y * width + max_x] += (double)(wrb * g);
                    if (ok4) plane[max_y * width + min_x] += (double)(wlb * g);
                }
            }
        }
    }
    for (long i = 0; i < total; ++i) bottom_diff[i] = (float)acc[i];
    free(acc);
}

/*
 * The same hoisted backward on `threads` OpenMP threads (the CPU baseline of bench.py's backward leg,
 * BASELINE.md section 2 last row).  A thread owns the channels c = tid, tid + T, ...: no two threads touch the same
 * feature element, and every element still receives its contributions in the order (n, ph, pw) ascending, in
 * double -- the result is bit-identical to rroi_oracle_backward whatever the thread count.  Per ROI every thread
 * tabulates the bins' taps once (512 bins at 8 x 64) and then walks its channels with top_diff read contiguously.
 */
void rroi_oracle_backward_mt(const float* top_diff, float spatial_scale, int batch_size, int num_rois,
                             int height, int width, int channels, int pooled_height,
                             int pooled_width, const float* bottom_rois, float* bottom_diff, int threads)
{
    const long plane_sz = (long)height * width;
    const long total = (long)batch_size * channels * plane_sz;
    const long bins = (long)pooled_height * pooled_width;
    double* acc = (double*)calloc((size_t)total, sizeof(double));
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num(), nt = omp_get_num_threads();
#else
        const int tid = 0, nt = 1;
#endif
        int* off = (int*)malloc((size_t)bins * 4 * sizeof(int));     /* pixel index of each tap, -1: fails :267-274 */
        float* wgt = (float*)malloc((size_t)bins * 4 * sizeof(float));
        for (int n = 0; n < num_rois; ++n) {
            rroi_affine_t A;
            rroi_oracle_affine(bottom_rois + (long)n * 6, pooled_height, spatial_scale, &A);
            for (int ph = 0; ph < pooled_height; ++ph)
                for (int pw = 0; pw < pooled_width; ++pw) {
                    const long b = (long)ph * pooled_width + pw;
                    float bin_cx, bin_cy;
                    off[4 * b] = off[4 * b + 1] = off[4 * b + 2] = off[4 * b + 3] = -1;
                    if (!bin_centre(&A, ph, pw, height, width, &bin_cx, &bin_cy)) continue;
                    tap_weights(bin_cx, bin_cy, &wgt[4 * b], &wgt[4 * b + 1], &wgt[4 * b + 2], &wgt[4 * b + 3]);
                    const int min_x = f2i_sat(floorf(bin_cx)), max_x = f2i_sat(ceilf(bin_cx));
                    const int min_y = f2i_sat(floorf(bin_cy)), max_y = f2i_sat(ceilf(bin_cy));
                    if (min_y > 0 && min_x > 0 && min_y < height - 1 && min_x < width - 1) off[4 * b] = min_y * width + min_x;
                    if (min_y > 0 && max_x < width - 1 && min_y < height - 1 && max_x > 0) off[4 * b + 1] = min_y * width + max_x;
                    if (max_y < height - 1 && max_x < width - 1 && max_y > 0 && max_x > 0) off[4 * b + 2] = max_y * width + max_x;
                    if (max_y < height - 1 && min_x > 0 && max_y > 0 && min_x < width - 1) off[4 * b + 3] = max_y * width + min_x;
                }
            for (int c = tid; c < channels; c += nt) {
                const float* g = top_diff + ((long)n * channels + c) * bins;
                double* plane = acc + ((long)A.batch * channels + c) * plane_sz;
                for (long b = 0; b < bins; ++b) {
                    const int* o = off + 4 * b;
                    const float* w = wgt + 4 * b;
                    if (o[0] >= 0) plane[o[0]] += (double)(w[0] * g[b]);
                    if (o[1] >= 0) plane[o[1]] += (double)(w[1] * g[b]);
                    if (o[2] >= 0) plane[o[2]] += (double)(w[2] * g[b]);
                    if (o[3] >= 0) plane[o[3]] += (double)(w[3] * g[b]);
                }
            }
        }
        free(off);
        free(wgt);
    }
    for (long i = 0; i < total; ++i) bottom_diff[i] = (float)acc[i];
    free(acc);
}

/* Number of distinct feature elements the forward reads (for bytes_feat in
 * SURVEY.md section 8(d)): counts (batch, y, x) taps that pass the validity
 * test of at least one active bin; multiply by channels*4 for bytes. */
long rroi_oracle_touched_pixels(float spatial_scale, int batch_size, int num_rois, int height,
                                int width, int pooled_height, int pooled_width,
                                const float* bottom_rois)
{
    const long plane_sz = (long)height * width;
    unsigned char* seen = (unsigned char*)calloc((size_t)(batch_size * plane_sz), 1);
    for (int n = 0; n < num_rois; ++n) {
        rroi_affine_t A;
        rroi_oracle_affine(bottom_rois + (long)n * 6, pooled_height, spatial_scale, &A);
        if (A.batch < 0 || A.batch >= batch_size) continue;
        for (int ph = 0; ph < pooled_height; ++ph)
            for (int pw = 0; pw < pooled_width; ++pw) {
                float cx, cy;
                if (!bin_centre(&A, ph, pw, height, width, &cx, &cy)) continue;
                const int xs[2] = {f2i_sat(floorf(cx)), f2i_sat(ceilf(cx))};
                const int ys[2] = {f2i_sat(floorf(cy)), f2i_sat(ceilf(cy))};
                for (int j = 0; j < 2; ++j)
                    for (int i = 0; i < 2; ++i)
                        if (ys[j] > 0 && xs[i] > 0 && ys[j] < height && xs[i] < width)
                            seen[A.batch * plane_sz + (long)ys[j] * width + xs[i]] = 1;
            }
    }
    long cnt = 0;
    for (long i = 0; i < batch_size * plane_sz; ++i) cnt += seen[i];
    free(seen);
    return cnt;
}

int rroi_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
