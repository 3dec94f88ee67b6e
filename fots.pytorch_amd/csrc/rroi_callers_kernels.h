// rroi_callers_kernels.h -- the callers' side: quads -> ROI rows, bin centres, greedy CTC decode, sin/cos probe
// Part of the single translation unit rroi_align_hip.hip (included inside its anonymous
// namespace, in this order: rroi_device_common.h, rroi_forward_kernels.h,
// rroi_backward_kernels.h, rroi_callers_kernels.h); not a standalone header.
#pragma once

// ------------------------------------------------------------------------------------
// Callers' side of the path: detected / annotated quads -> the op's
// (R, 6) ROI rows, on the device, for a whole image batch at once -- so that inference can
// issue ONE RoIRotate launch per image instead of one per word (tools/ocr_utils.py:131-177).
//   mode 0  tools/ocr_utils.py:133-150: fp32 edge vectors, fp32 squared length, sqrt and atan2 in
//           double, centre truncated to int, angle of edge 1->2
//   mode 1  src/ocr_process.py:196-206: everything in double, angle = mean of edges 1->2 and 0->3
// Also emits each box's pooled width by the inference rule (ocr_utils.py:147-150).
// ------------------------------------------------------------------------------------
__global__ void rroi_quads_to_rois_kernel(const float* __restrict__ quads, const float* __restrict__ bidx,
                                          int n, int mode, int target_h, float* __restrict__ rois,
                                          int* __restrict__ gw)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* b = quads + (size_t)i * 8;
    const float x0 = b[0], y0 = b[1], x1 = b[2], y1 = b[3], x2 = b[4], y2 = b[5], x3 = b[6], y3 = b[7];
    double w, h, angle, cx, cy;
    if (mode == 0) {
        const float ccx = (((x0 + x1) + x2) + x3) / 4.0f, ccy = (((y0 + y1) + y2) + y3) / 4.0f;
        const float dwx = x2 - x1, dwy = y2 - y1, dhx = x1 - x0, dhy = y1 - y0;
        w = sqrt((double)((dwx * dwx) + (dwy * dwy)));
        h = sqrt((double)((dhx * dhx) + (dhy * dhy)));
        angle = atan2((double)(y2 - y1), (double)(x2 - x1));
        cx = (double)(int)ccx;  // int(center[0]): truncation toward zero
        cy = (double)(int)ccy;
    } else {
        const double X0 = x0, Y0 = y0, X1 = x1, Y1 = y1, X2 = x2, Y2 = y2, X3 = x3, Y3 = y3;
        cx = (((X0 + X1) + X2) + X3) / 4.0;
        cy = (((Y0 + Y1) + Y2) + Y3) / 4.0;
        const double dwx = X2 - X1, dwy = Y2 - Y1, dhx = X1 - X0, dhy = Y1 - Y0;
        w = sqrt(dwx * dwx + dwy * dwy);
        h = sqrt(dhx * dhx + dhy * dhy);
        angle = (atan2(Y2 - Y1, X2 - X1) + atan2(Y3 - Y0, X3 - X0)) / 2.0;
    }
    angle = -angle / 3.1415926535 * 180.0;
    float* r = rois + (size_t)i * 6;
    r[0] = bidx ? bidx[i] : 0.0f;
    r[1] = (float)cx;
    r[2] = (float)cy;
    r[3] = (float)h;
    r[4] = (float)w;
    r[5] = (float)angle;
    if (gw) {
        const double scale = (double)target_h / (h > 1.0 ? h : 1.0);  // max(1, h)
        const int t = (int)(w * scale) + target_h;
        const int g = t / 32;  // t >= target_h > 0: floor division == truncation
        gw[i] = (g > 2 ? g : 2) * 32;
    }
}

// Training side (src/ocr_process.py:196-219, :259-263): the ground-truth quads of a batch -> ROI
// rows with the caller's height jitter applied where the reference applies it (in double, before
// the row is rounded to fp32, :204), and the ratio the pooled width is chosen from,
// max(w / h) over the fp32 rows (:260-262; NaN wins, as torch.max has it).  One block: a training
// batch has at most a few hundred boxes.
__global__ __launch_bounds__(256) void rroi_gt_quads_to_rois_kernel(
    const float* __restrict__ quads, const float* __restrict__ bidx, const float* __restrict__ jitter, int n,
    float* __restrict__ rois, float* __restrict__ max_ratio)
{
    __shared__ float part[256];
    __shared__ int any_nan;
    if (threadIdx.x == 0) any_nan = 0;
    __syncthreads();
    float best = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float* b = quads + (size_t)i * 8;
        const double X0 = b[0], Y0 = b[1], X1 = b[2], Y1 = b[3], X2 = b[4], Y2 = b[5], X3 = b[6], Y3 = b[7];
        const double cx = (((X0 + X1) + X2) + X3) / 4.0, cy = (((Y0 + Y1) + Y2) + Y3) / 4.0;
        const double dwx = X2 - X1, dwy = Y2 - Y1, dhx = X1 - X0, dhy = Y1 - Y0;
        const double w = sqrt(dwx * dwx + dwy * dwy);
        const double h = sqrt(dhx * dhx + dhy * dhy) + (jitter ? (double)jitter[i] : 0.0);
        double angle = (atan2(Y2 - Y1, X2 - X1) + atan2(Y3 - Y0, X3 - X0)) / 2.0;
        angle = -angle / 3.1415926535 * 180.0;
        float* r = rois + (size_t)i * 6;
        const float hf = (float)h, wf = (float)w;
        r[0] = bidx ? bidx[i] : 0.0f;
        r[1] = (float)cx;
        r[2] = (float)cy;
        r[3] = hf;
        r[4] = wf;
        r[5] = (float)angle;
        const float ratio = wf / hf;
        if (ratio != ratio) any_nan = 1;
        else if (ratio > best) best = ratio;
    }
    part[threadIdx.x] = best;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s && part[threadIdx.x + s] > part[threadIdx.x]) part[threadIdx.x] = part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *max_ratio = any_nan ? NAN : part[0];
}

__global__ void rroi_bin_centres_kernel(const float* __restrict__ rois, float* __restrict__ geom,
                                        int num_rois, int height, int width, int pooled_height,
                                        int pooled_width, float spatial_scale, int trig)
{
    const int NB = pooled_height * pooled_width;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)num_rois * NB) return;
    const int n = (int)(gid / NB);
    const int bin = (int)(gid - (long)n * NB);
    const int ph = bin / pooled_width, pw = bin - ph * pooled_width;
    const Affine A = make_affine(rois + (size_t)n * 6, pooled_height, spatial_scale, trig);
    float bcx, bcy;
    const bool in_rroi = bin_centre(A, ph, pw, height, width, bcx, bcy);
    geom[gid * 2 + 0] = in_rroi ? bcx : 0.0f;
    geom[gid * 2 + 1] = in_rroi ? bcy : 0.0f;
}

// ------------------------------------------------------------------------------------
// Greedy CTC decode of the recognition logits that the crops turn into:
// tools/ocr_utils.py:183-186 takes `labels_pred.max(1)` -- arg max over the class axis of
// (N, nclass, T) -- and src/utils.py:87-97 keeps label t iff it is not the blank (0) and differs
// from label t-1.  One wave per sequence, lanes = time steps (the class loop reads rows that
// are contiguous in t); arg max = first index of the largest value, NaN counting as largest
// (torch.max); the kept labels are compacted with a ballot + popcount prefix.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void rroi_ctc_greedy_kernel(
    const float* __restrict__ logits, int nclass, int T, const int* __restrict__ lengths,
    int* __restrict__ labels, int* __restrict__ decoded, int* __restrict__ decoded_len)
{
    const unsigned n = blockIdx.x, lane = threadIdx.x;
    int len = lengths ? lengths[n] : T;
    len = len < 0 ? 0 : (len > T ? T : len);
    const float* row = logits + (size_t)n * nclass * T;
    int* lab = labels ? labels + (size_t)n * T : nullptr;
    int* dec = decoded + (size_t)n * T;
    unsigned out = 0;
    int prev_last = -1;  // label of time step t0 - 1 (none before the first)
    for (int t0 = 0; t0 < T; t0 += kWave) {
        const int t = t0 + (int)lane;
        int best = 0;
        if (t < T) {
            float bv = row[t];
            bool bnan = bv != bv;
            for (int k = 1; k < nclass; ++k) {
                const float v = row[(size_t)k * T + t];
                const bool vnan = v != v;
                if (!bnan && (vnan || v > bv)) {
                    bv = v;
                    best = k;
                    bnan = vnan;
                }
            }
            if (lab) lab[t] = best;
        }
        int prev = __shfl_up(best, 1, kWave);
        if (lane == 0) prev = prev_last;
        const bool keep = t < len && best != 0 && best != prev;
        const unsigned long long m = __ballot(keep);
        if (keep) dec[out + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = best;
        out += (unsigned)__popcll(m);
        prev_last = __shfl(best, kWave - 1, kWave);
    }
    for (unsigned i = out + lane; i < (unsigned)T; i += kWave) dec[i] = 0;  // padding
    if (lane == 0) decoded_len[n] = (int)out;
}

__global__ void rroi_sincos_probe_kernel(const float* __restrict__ deg, int n, float* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float angle = (float)(((double)deg[i] / 180.0) * 3.1415926535);
    out[2 * i + 0] = (float)cos((double)angle);
    out[2 * i + 1] = (float)sin((double)angle);
}

// ------------------------------------------------------------------------------------
// Measurement hook (bench.py `roofline.calibrated_nonzero`): a plain write of NON-CONSTANT data,
// one 16-byte store per thread -- the launch shape of a framework's elementwise kernel -- so that the ceiling of the
// gather's output stream is measured in the same run, on the same box, by the same clock.  Not on the hot path.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rroi_write_probe_kernel(float* __restrict__ out, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n4) return;
    const float f = (float)((unsigned)i * 2654435761u >> 8) * 1.1920929e-7f;   // [0, 2): a different value per store
    reinterpret_cast<v4f*>(out)[i] = v4f{f, f + 1.0f, f * 3.0f, -f};
}
