// Training labels of the Siamese (Ocean) head for a batch of boxes, in one launch
// (ltr/models/loss/ocean_label_gen.py:44-66,124-233; the reference builds the classification label with a Python loop
// over the batch, :219-231).  For clip b with box (x, y, w, h) on the search patch and response-map cell (r, c) whose
// centre on the patch is gx = (c - S/2) stride + search/2, gy = (r - S/2) stride + search/2:
//   reg[b][r][c]   = (gx - x1, gy - y1, x2 - gx, y2 - gy)          float64: the reference's grid is a float64 tensor,
//                                                                  so its subtraction promotes (:119-120,143-146)
//   reg_w[b][r][c] = min(reg) > 0
//   iou_w[b][r][c] = the same test against the box grown by `expand` of its (>= 1 px) size on every side (:156-190)
//   cls[b][r][c]   = |c - cx| + |r - cy| <= radius, (cx, cy) = S/2 + trunc((box centre - search/2) / stride) (:192-233)
// Built with -ffp-contract=off: the float32 steps round one by one as the reference's tensor ops do.
#include "common.h"
#include "../../include/spot_hip.h"

namespace {

__global__ __launch_bounds__(256) void ocean_labels_kernel(const float* __restrict__ anno, float* __restrict__ cls,
                                                           double* __restrict__ reg, float* __restrict__ reg_w,
                                                           float* __restrict__ iou_w, int B, int S, int stride,
                                                           int search, float expand, int radius) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * S * S) return;
    const int c = (int)(idx % S), r = (int)(idx / S % S), b = (int)(idx / ((long)S * S));
    const float x1 = anno[4 * b], y1 = anno[4 * b + 1];
    const float x2 = x1 + anno[4 * b + 2], y2 = y1 + anno[4 * b + 3];
    const double gx = (double)((c - S / 2) * stride + search / 2), gy = (double)((r - S / 2) * stride + search / 2);
    if (reg || reg_w) {
        const double l = gx - (double)x1, t = gy - (double)y1, rr = (double)x2 - gx, bb = (double)y2 - gy;
        if (reg) {
            double* o = reg + 4 * idx;
            o[0] = l; o[1] = t; o[2] = rr; o[3] = bb;
        }
        if (reg_w) reg_w[idx] = fmin(fmin(l, t), fmin(rr, bb)) > 0.0 ? 1.f : 0.f;
    }
    if (iou_w) {
        const float w = fmaxf(x2 - x1, 1.f), h = fmaxf(y2 - y1, 1.f);
        const float ew = expand * w, eh = expand * h;
        const float x1o = x1 - ew, x2o = x2 + ew, y1o = y1 - eh, y2o = y2 + eh;
        const double l = gx - (double)x1o, t = gy - (double)y1o, rr = (double)x2o - gx, bb = (double)y2o - gy;
        iou_w[idx] = fmin(fmin(l, t), fmin(rr, bb)) > 0.0 ? 1.f : 0.f;
    }
    if (cls) {
        const float half = (float)search / 2.f;
        const float sx = (x1 + x2) / 2.f - half, sy = (y1 + y2) / 2.f - half;
        const long cx = S / 2 + (long)(sx / (float)stride), cy = S / 2 + (long)(sy / (float)stride);   // truncation, as .long()
        cls[idx] = (labs((long)c - cx) + labs((long)r - cy) <= radius) ? 1.f : 0.f;
    }
}

// One frame of the Siamese online tracker, from the network's three maps to the winning cell
// (pysot_toolkit/trackers/ocean_tracker.py:71-104): score = ratio sigmoid(cls) + (1 - ratio) sigmoid(align) in
// float32; boxes, scale / ratio penalty, window blend and arg-max in float64 like the reference's numpy arithmetic
// (its grid is float64); the first maximum wins.  out[8] = (row, col, x1, y1, x2, y2, penalty, score) of that cell.
__global__ __launch_bounds__(256) void ocean_tracker_decode_kernel(const float* __restrict__ cls, const float* __restrict__ align,
                                                                   const float* __restrict__ ltrb, const double* __restrict__ window,
                                                                   const double* __restrict__ target_wh, float ratio,
                                                                   float one_minus_ratio,
                                                                   double penalty_k, double window_influence, int S, int stride,
                                                                   int instance, double* __restrict__ out) {
    __shared__ double bv[256];
    __shared__ int bi[256];
    const int n = S * S;
    const double tw = target_wh[0], th = target_wh[1];
    double best = -1.0 / 0.0;
    int besti = n;
    const double pad_t = (tw + th) * 0.5, sz_t = sqrt((tw + pad_t) * (th + pad_t));
    for (int i = threadIdx.x; i < n; i += 256) {
        const int r = i / S, c = i - r * S;
        const float sc = ratio * (1.f / (1.f + expf(-cls[i]))) + one_minus_ratio * (1.f / (1.f + expf(-align[i])));
        const double gx = (double)((c - S / 2) * stride + instance / 2), gy = (double)((r - S / 2) * stride + instance / 2);
        const double w = (gx + (double)ltrb[2 * n + i]) - (gx - (double)ltrb[i]);
        const double h = (gy + (double)ltrb[3 * n + i]) - (gy - (double)ltrb[n + i]);
        const double pad = (w + h) * 0.5;
        double sc_r = sqrt((w + pad) * (h + pad)) / sz_t;
        sc_r = fmax(sc_r, 1.0 / sc_r);
        double rc_r = (tw / th) / (w / h);
        rc_r = fmax(rc_r, 1.0 / rc_r);
        const double pen = exp(-(rc_r * sc_r - 1.0) * penalty_k);
        const double ps = pen * (double)sc * (1.0 - window_influence) + window[i] * window_influence;
        // numpy.argmax order: the first NaN wins over every number, else the first maximum
        if (best == best && (ps > best || ps != ps)) {
            best = ps;
            besti = i;
        }
    }
    bv[threadIdx.x] = best;
    bi[threadIdx.x] = besti;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const double v = bv[threadIdx.x + o];
            const int j = bi[threadIdx.x + o];
            const double mine = bv[threadIdx.x];
            const bool vn = v != v, mn = mine != mine;
            if ((vn && (!mn || j < bi[threadIdx.x])) || (!vn && !mn && (v > mine || (v == mine && j < bi[threadIdx.x])))) {
                bv[threadIdx.x] = v;
                bi[threadIdx.x] = j;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int i = bi[0] < n ? bi[0] : 0, r = i / S, c = i - r * S;
        const float sc = ratio * (1.f / (1.f + expf(-cls[i]))) + one_minus_ratio * (1.f / (1.f + expf(-align[i])));
        const double gx = (double)((c - S / 2) * stride + instance / 2), gy = (double)((r - S / 2) * stride + instance / 2);
        const double x1 = gx - (double)ltrb[i], y1 = gy - (double)ltrb[n + i];
        const double x2 = gx + (double)ltrb[2 * n + i], y2 = gy + (double)ltrb[3 * n + i];
        const double w = x2 - x1, h = y2 - y1, pad = (w + h) * 0.5;
        double sc_r = sqrt((w + pad) * (h + pad)) / sz_t;
        sc_r = fmax(sc_r, 1.0 / sc_r);
        double rc_r = (tw / th) / (w / h);
        rc_r = fmax(rc_r, 1.0 / rc_r);
        out[0] = r; out[1] = c; out[2] = x1; out[3] = y1; out[4] = x2; out[5] = y2;
        out[6] = exp(-(rc_r * sc_r - 1.0) * penalty_k);
        out[7] = (double)sc;
    }
}

}  // namespace

extern "C" int spot_ocean_tracker_decode(const float* cls_logits, const float* align_logits, const float* ltrb,
                                         const double* window, const double* target_wh, double ratio,
                                         double penalty_k, double window_influence, int S, int stride, int instance_size,
                                         double* out8, hipStream_t stream) {
    SPOT_CHECK_ARG(cls_logits && align_logits && ltrb && window && target_wh && out8, "ocean_tracker_decode: null pointer");
    SPOT_CHECK_ARG(S > 0 && stride > 0 && instance_size > 0, "ocean_tracker_decode: bad shape");
    ocean_tracker_decode_kernel<<<1, 256, 0, stream>>>(cls_logits, align_logits, ltrb, window, target_wh, (float)ratio,
                                                       (float)(1.0 - ratio), penalty_k, window_influence, S, stride,
                                                       instance_size, out8);
    SPOT_CHECK_LAUNCH("ocean_tracker_decode");
    return SPOT_OK;
}

extern "C" int spot_ocean_labels_f32(const float* anno_xywh, float* cls, double* reg, float* reg_w, float* iou_w, int B,
                                     int S, int stride, int search_sz, float expand, int radius, hipStream_t stream) {
    SPOT_CHECK_ARG(B >= 0 && S > 0 && stride > 0 && search_sz > 0, "ocean_labels: bad shape B=%d S=%d stride=%d", B, S, stride);
    SPOT_CHECK_ARG(anno_xywh || B == 0, "ocean_labels: null boxes");
    if (B == 0) return SPOT_OK;
    ocean_labels_kernel<<<spot_ceil_div((long)B * S * S, 256), 256, 0, stream>>>(anno_xywh, cls, reg, reg_w, iou_w, B, S,
                                                                                stride, search_sz, expand, radius);
    SPOT_CHECK_LAUNCH("ocean_labels");
    return SPOT_OK;
}
