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

}  // namespace

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
