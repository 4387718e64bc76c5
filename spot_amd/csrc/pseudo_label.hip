// Pseudo-label decode, IoU-confidence filter, box re-projection and batch
// compaction for the teacher -> student hand-over.
//
// One 256-thread workgroup per clip reduces the 1024-token score map with
// wavefront shuffles; lane 0 then does the dozen scalar box operations.  The
// arithmetic order (separate multiply and add, true division) follows the
// reference's tensor expressions so that the arg-max token, the boxes and the
// keep mask come out bit-identical to its fp32 path:
//   decode      ltr/actor/transt_sparse_actor.py:57-98
//   recover     ltr/data/processing_utils.py:205-229
//   keep mask   ltr/data/middleware.py:117  (conf >= tau)
//   re-project  ltr/data/processing_utils.py:180-202, middleware.py:149-152
//   compaction  ltr/trainers/ltr_sparse_trainer.py:334-357
#include "common.h"
#include "../../include/spot_hip.h"

// The reference evaluates these expressions as separate rounded multiplies and adds.  HIP's
// __fmul_rn/__fadd_rn are plain operators and hipcc fuses them by default, so this file is
// compiled with -ffp-contract=off (see the Makefile); explicit fmaf() calls stay fused.

namespace {

struct Best {
    float v;
    int i;
};

__device__ __forceinline__ Best better(Best a, Best b) {
    // larger value wins; on ties the smaller token index (first occurrence).  A NaN beats every number and the first
    // NaN beats later ones, as torch.max / numpy.argmax order them: the chosen index always lies inside the clip's map
    // (a diverged teacher gives conf = NaN, which fails the threshold, instead of an index no candidate ever set).
    const bool an = a.v != a.v, bn = b.v != b.v;
    if (an || bn) {
        if (an && bn) return a.i < b.i ? a : b;
        return an ? a : b;
    }
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}

template <bool DECODE, bool FILTER>
__global__ __launch_bounds__(256) void decode_filter_kernel(
    const float* __restrict__ logits,   // [B, L, 2]
    const float* __restrict__ boxes,    // [B, L, 4] cx,cy,w,h in [0,1]
    const float* __restrict__ ious,     // [B, L]
    const float* __restrict__ window,   // [L]
    float wp, float omwp, float img_w, float img_h,
    const float* __restrict__ unl_ext,  // [B,4]
    const float* __restrict__ unl_rf,   // [B]
    const float* __restrict__ aug_ext,  // [B,4]
    const float* __restrict__ aug_rf,   // [B]
    float half_crop,                    // (search_sz - 1) / 2
    float tau,
    float* __restrict__ best_bbox,      // [B,4] xywh px in the search crop  (in/out)
    float* __restrict__ best_score,     // [B]
    float* __restrict__ best_conf,      // [B]                               (in/out)
    int* __restrict__ best_idx,         // [B]
    float* __restrict__ recovered,      // [B,4] xywh px in the original image
    float* __restrict__ aug_anno,       // [B,4] xywh px in the aug crop
    unsigned char* __restrict__ keep,   // [B]
    int L) {
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    __shared__ Best s_best[4];

    float bx = 0.f, by = 0.f, bw = 0.f, bh = 0.f, conf = 0.f;
    if (DECODE) {
        const float2* lg = reinterpret_cast<const float2*>(logits) + (size_t)b * L;
        Best best{-INFINITY, 0x7fffffff};
        for (int t = tid; t < L; t += 256) {
            float2 l = lg[t];
            float m = fmaxf(l.x, l.y);
            float e0 = expf(l.x - m), e1 = expf(l.y - m);
            float s = e0 / (e0 + e1);
            float v = __fadd_rn(__fmul_rn(wp, window[t]), __fmul_rn(omwp, s));
            best = better(best, Best{v, t});
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            Best other{__shfl_xor(best.v, o), __shfl_xor(best.i, o)};
            best = better(best, other);
        }
        if ((tid & 63) == 0) s_best[tid >> 6] = best;
        __syncthreads();
        if (tid != 0) return;
        best = better(better(s_best[0], s_best[1]), better(s_best[2], s_best[3]));
        const float4 bb = reinterpret_cast<const float4*>(boxes)[(size_t)b * L + best.i];
        conf = ious[(size_t)b * L + best.i];
        bx = fmaxf(__fmul_rn(__fsub_rn(bb.x, __fmul_rn(bb.z, 0.5f)), img_w), 0.f);
        by = fmaxf(__fmul_rn(__fsub_rn(bb.y, __fmul_rn(bb.w, 0.5f)), img_h), 0.f);
        bw = fmaxf(__fmul_rn(bb.z, img_w), 1.f);
        bh = fmaxf(__fmul_rn(bb.w, img_h), 1.f);
        reinterpret_cast<float4*>(best_bbox)[b] = make_float4(bx, by, bw, bh);
        best_score[b] = best.v;
        best_conf[b] = conf;
        best_idx[b] = best.i;
    } else {
        if (tid != 0) return;
        const float4 bb = reinterpret_cast<const float4*>(best_bbox)[b];
        bx = bb.x; by = bb.y; bw = bb.z; bh = bb.w;
        conf = best_conf[b];
    }
    if (FILTER) {
        const float4 ue = reinterpret_cast<const float4*>(unl_ext)[b];
        const float4 ae = reinterpret_cast<const float4*>(aug_ext)[b];
        const float urf = unl_rf[b], arf = aug_rf[b];
        // crop -> original image
        float uex = __fadd_rn(ue.x, __fmul_rn(0.5f, ue.z)), uey = __fadd_rn(ue.y, __fmul_rn(0.5f, ue.w));
        float ccx = __fadd_rn(bx, __fmul_rn(0.5f, bw)), ccy = __fadd_rn(by, __fmul_rn(0.5f, bh));
        float ow = __fdiv_rn(bw, urf), oh = __fdiv_rn(bh, urf);
        float ocx = __fadd_rn(__fdiv_rn(__fsub_rn(ccx, half_crop), urf), uex);
        float ocy = __fadd_rn(__fdiv_rn(__fsub_rn(ccy, half_crop), urf), uey);
        float rx = fmaxf(__fsub_rn(ocx, __fmul_rn(0.5f, ow)), 0.f);
        float ry = fmaxf(__fsub_rn(ocy, __fmul_rn(0.5f, oh)), 0.f);
        float rw = fmaxf(ow, 0.f), rh = fmaxf(oh, 0.f);
        reinterpret_cast<float4*>(recovered)[b] = make_float4(rx, ry, rw, rh);
        keep[b] = conf >= tau ? 1 : 0;
        // original image -> aug crop
        float aex = __fadd_rn(ae.x, __fmul_rn(0.5f, ae.z)), aey = __fadd_rn(ae.y, __fmul_rn(0.5f, ae.w));
        float icx = __fadd_rn(rx, __fmul_rn(0.5f, rw)), icy = __fadd_rn(ry, __fmul_rn(0.5f, rh));
        float acx = __fadd_rn(half_crop, __fmul_rn(__fsub_rn(icx, aex), arf));
        float acy = __fadd_rn(half_crop, __fmul_rn(__fsub_rn(icy, aey), arf));
        float aw = __fmul_rn(rw, arf), ah = __fmul_rn(rh, arf);
        reinterpret_cast<float4*>(aug_anno)[b] =
            make_float4(fmaxf(__fsub_rn(acx, __fmul_rn(0.5f, aw)), 0.f),
                        fmaxf(__fsub_rn(acy, __fmul_rn(0.5f, ah)), 0.f), fmaxf(aw, 0.f), fmaxf(ah, 0.f));
    }
}

// Single workgroup: ordered compaction of the kept clip indices by wavefront
// ballot + prefix popcount, then cyclic refill to n_out rows.
__global__ __launch_bounds__(1024) void compact_kernel(const unsigned char* __restrict__ keep, int n_in,
                                                       int* __restrict__ gather_idx, int n_out,
                                                       int* __restrict__ n_keep_out) {
    extern __shared__ int s_kept[];  // [n_in] kept indices, then 16 wave counters
    __shared__ int s_wave[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int start = 0; start < n_in; start += 1024) {
        const int i = start + tid;
        const bool k = i < n_in && keep[i] != 0;
        const unsigned long long m = __ballot(k);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wv] = __popcll(m);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wv; ++w) off += s_wave[w];
        if (k) s_kept[off + before] = i;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < 16; ++w) t += s_wave[w];
            s_base += t;
        }
        __syncthreads();
    }
    const int n = s_base;
    if (tid == 0) *n_keep_out = n;
    for (int j = tid; j < n_out; j += 1024) gather_idx[j] = n > 0 ? s_kept[j % n] : -1;
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                          const int* __restrict__ idx, long row_vec) {
    const int r = blockIdx.y;
    const int s = idx[r];
    uint4* dp = dst + (size_t)r * row_vec;
    if (s < 0) {                 // no source row (nothing was kept): the row is defined, all zero bytes
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < row_vec; i += (long)gridDim.x * 256) dp[i] = make_uint4(0, 0, 0, 0);
        return;
    }
    const uint4* sp = src + (size_t)s * row_vec;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < row_vec; i += (long)gridDim.x * 256) dp[i] = sp[i];
}

__global__ __launch_bounds__(256) void gather_rows_bytes_kernel(const unsigned char* __restrict__ src,
                                                                unsigned char* __restrict__ dst,
                                                                const int* __restrict__ idx, long row_bytes) {
    const int r = blockIdx.y;
    const int s = idx[r];
    unsigned char* dp = dst + (size_t)r * row_bytes;
    if (s < 0) {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < row_bytes; i += (long)gridDim.x * 256) dp[i] = 0;
        return;
    }
    const unsigned char* sp = src + (size_t)s * row_bytes;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < row_bytes; i += (long)gridDim.x * 256) dp[i] = sp[i];
}

}  // namespace

extern "C" int spot_decode_teacher_f32(const float* pred_logits, const float* pred_boxes, const float* pred_ious,
                                       const float* window, double window_penalty, int img_w, int img_h,
                                       float* best_bbox, float* best_score, float* best_conf, int* best_idx,
                                       int B, int L, hipStream_t stream) {
    SPOT_CHECK_ARG(B >= 0 && L > 0, "decode_teacher: bad B=%d L=%d", B, L);
    if (B == 0) return SPOT_OK;
    SPOT_CHECK_ARG(pred_logits && pred_boxes && pred_ious && window && best_bbox && best_score && best_conf &&
                       best_idx, "decode_teacher: null pointer");
    decode_filter_kernel<true, false><<<B, 256, 0, stream>>>(
        pred_logits, pred_boxes, pred_ious, window, (float)window_penalty, (float)(1.0 - window_penalty),
        (float)img_w, (float)img_h, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, best_bbox, best_score,
        best_conf, best_idx, nullptr, nullptr, nullptr, L);
    SPOT_CHECK_LAUNCH("decode_teacher");
    return SPOT_OK;
}

extern "C" int spot_middleware_filter_f32(const float* best_bbox, const float* best_conf,
                                          const float* unlabeled_box_extract, const float* unlabeled_resize_factors,
                                          const float* aug_box_extract, const float* aug_resize_factors,
                                          int search_sz, float confidence_threshold, float* bbox_recovered,
                                          float* aug_anno, unsigned char* aug_success, int B, hipStream_t stream) {
    SPOT_CHECK_ARG(B >= 0, "middleware_filter: bad B=%d", B);
    if (B == 0) return SPOT_OK;
    SPOT_CHECK_ARG(best_bbox && best_conf && unlabeled_box_extract && unlabeled_resize_factors &&
                       aug_box_extract && aug_resize_factors && bbox_recovered && aug_anno && aug_success,
                   "middleware_filter: null pointer");
    decode_filter_kernel<false, true><<<B, 64, 0, stream>>>(
        nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, 0.f, unlabeled_box_extract, unlabeled_resize_factors,
        aug_box_extract, aug_resize_factors, (float)((search_sz - 1) / 2.0), confidence_threshold,
        const_cast<float*>(best_bbox), nullptr, const_cast<float*>(best_conf), nullptr, bbox_recovered, aug_anno,
        aug_success, 0);
    SPOT_CHECK_LAUNCH("middleware_filter");
    return SPOT_OK;
}

extern "C" int spot_decode_filter_f32(const float* pred_logits, const float* pred_boxes, const float* pred_ious,
                                      const float* window, double window_penalty, int img_w, int img_h,
                                      const float* unlabeled_box_extract, const float* unlabeled_resize_factors,
                                      const float* aug_box_extract, const float* aug_resize_factors, int search_sz,
                                      float confidence_threshold, float* best_bbox, float* best_score,
                                      float* best_conf, int* best_idx, float* bbox_recovered, float* aug_anno,
                                      unsigned char* aug_success, int B, int L, hipStream_t stream) {
    SPOT_CHECK_ARG(B >= 0 && L > 0, "decode_filter: bad B=%d L=%d", B, L);
    if (B == 0) return SPOT_OK;
    SPOT_CHECK_ARG(pred_logits && pred_boxes && pred_ious && window && unlabeled_box_extract &&
                       unlabeled_resize_factors && aug_box_extract && aug_resize_factors && best_bbox &&
                       best_score && best_conf && best_idx && bbox_recovered && aug_anno && aug_success,
                   "decode_filter: null pointer");
    decode_filter_kernel<true, true><<<B, 256, 0, stream>>>(
        pred_logits, pred_boxes, pred_ious, window, (float)window_penalty, (float)(1.0 - window_penalty),
        (float)img_w, (float)img_h, unlabeled_box_extract, unlabeled_resize_factors, aug_box_extract,
        aug_resize_factors, (float)((search_sz - 1) / 2.0), confidence_threshold, best_bbox, best_score, best_conf,
        best_idx, bbox_recovered, aug_anno, aug_success, L);
    SPOT_CHECK_LAUNCH("decode_filter");
    return SPOT_OK;
}

extern "C" int spot_compact_indices(const unsigned char* aug_success, int n_in, int* gather_idx, int n_out,
                                    int* n_keep, hipStream_t stream) {
    SPOT_CHECK_ARG(n_in > 0 && n_out > 0, "compact_indices: bad n_in=%d n_out=%d", n_in, n_out);
    SPOT_CHECK_ARG(n_in <= 12288, "compact_indices: n_in=%d exceeds 12288 clips per call", n_in);
    SPOT_CHECK_ARG(aug_success && gather_idx && n_keep, "compact_indices: null pointer");
    compact_kernel<<<1, 1024, (size_t)n_in * sizeof(int), stream>>>(aug_success, n_in, gather_idx, n_out, n_keep);
    SPOT_CHECK_LAUNCH("compact_indices");
    return SPOT_OK;
}

extern "C" int spot_gather_rows(const void* src, void* dst, const int* gather_idx, int n_rows, long row_bytes,
                                hipStream_t stream) {
    SPOT_CHECK_ARG(n_rows >= 0 && row_bytes > 0, "gather_rows: bad n_rows=%d row_bytes=%ld", n_rows, row_bytes);
    SPOT_CHECK_ARG(src && dst && gather_idx, "gather_rows: null pointer");
    if (n_rows == 0) return SPOT_OK;
    const bool vec = (row_bytes % 16 == 0) && (((uintptr_t)src | (uintptr_t)dst) % 16 == 0);
    if (vec) {
        long rv = row_bytes / 16;
        int gx = (int)((rv + 255) / 256);
        if (gx > 64) gx = 64;
        gather_rows_kernel<<<dim3(gx, n_rows), 256, 0, stream>>>((const uint4*)src, (uint4*)dst, gather_idx, rv);
    } else {
        int gx = (int)((row_bytes + 255) / 256);
        if (gx > 64) gx = 64;
        gather_rows_bytes_kernel<<<dim3(gx, n_rows), 256, 0, stream>>>((const unsigned char*)src,
                                                                       (unsigned char*)dst, gather_idx, row_bytes);
    }
    SPOT_CHECK_LAUNCH("gather_rows");
    return SPOT_OK;
}

// ---------------------------------------------------------------------------
// Matcher (ltr/models/loss/transt_matcher.py:24-109): tokens whose cell index lies in
// the ground-truth box are positives; a ring of width `gap` around them are the
// IoU-head negatives.  The reference does this arithmetic in Python floats
// (float64) on .item() values, hence the doubles, and cuts the token grid with numpy
// slices [a:b]: a bound below zero counts from the end (a box wholly left of or above
// the map gives a negative upper bound), a bound above n is n.  The four rectangles of
// the ring (:92-95) are disjoint for boxes on the map and may overlap once a bound has
// wrapped, so the ring is returned as a multiplicity.  One thread per token.
// ---------------------------------------------------------------------------
namespace {
__device__ __forceinline__ int np_bound(int i, int n) { return i < 0 ? max(i + n, 0) : min(i, n); }
__device__ __forceinline__ bool np_rect(int y, int x, int y0, int y1, int x0, int x1, int n) {
    return y >= np_bound(y0, n) && y < np_bound(y1, n) && x >= np_bound(x0, n) && x < np_bound(x1, n);
}

__global__ __launch_bounds__(256) void match_masks_kernel(const float* __restrict__ gt, int n, int gap,
                                                          unsigned char* __restrict__ pos,
                                                          unsigned char* __restrict__ neg, int B) {
    const int L = n * n;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * L) return;
    const int b = (int)(i / L), t = (int)(i % L), y = t / n, x = t % n;
    const double cx = gt[4 * b], cy = gt[4 * b + 1], w = gt[4 * b + 2], h = gt[4 * b + 3];
    int X0 = max((int)ceil((cx - w / 2) * n), 0), Y0 = max((int)ceil((cy - h / 2) * n), 0);
    int X1 = min((int)ceil((cx + w / 2) * n), n), Y1 = min((int)ceil((cy + h / 2) * n), n);
    if (X0 == X1) X1 += 1;
    if (Y0 == Y1) Y1 += 1;
    pos[i] = np_rect(y, x, Y0, Y1, X0, X1, n) ? 1 : 0;
    if (neg) {
        const int Yo0 = max(0, Y0 - gap), Yo1 = min(n, Y1 + gap), Xo0 = max(0, X0 - gap), Xo1 = min(n, X1 + gap);
        neg[i] = (unsigned char)((int)np_rect(y, x, Yo0, Yo1, Xo0, X0, n) + (int)np_rect(y, x, Yo0, Y0, X0, X1, n) +
                                 (int)np_rect(y, x, Y1, Yo1, X0, X1, n) + (int)np_rect(y, x, Yo0, Yo1, X1, Xo1, n));
    }
}
}  // namespace

extern "C" int spot_match_masks(const float* gt_cxcywh, int B, int feat_sz, int gap, unsigned char* pos_mask,
                                unsigned char* neg_mask, hipStream_t stream) {
    SPOT_CHECK_ARG(B >= 0 && feat_sz > 0 && gap >= 0, "match_masks: bad B=%d feat_sz=%d gap=%d", B, feat_sz, gap);
    if (B == 0) return SPOT_OK;
    SPOT_CHECK_ARG(gt_cxcywh && pos_mask, "match_masks: null pointer");
    const long total = (long)B * feat_sz * feat_sz;
    match_masks_kernel<<<(int)((total + 255) / 256), 256, 0, stream>>>(gt_cxcywh, feat_sz, gap, pos_mask, neg_mask, B);
    SPOT_CHECK_LAUNCH("match_masks");
    return SPOT_OK;
}

// ---------------------------------------------------------------------------
// Online tracker post-processing (pysot_toolkit/trackers/transt_tracker.py:21-32,164-171):
// softmax[:, 0] in fp32, pscore = score*(1-wp) + window*wp in float64 (the reference mixes a
// float32 score array with a float64 numpy window), arg-max, and the raw box of that token.
// One workgroup per frame; outputs stay on the device until the host needs the 4+1 numbers.
// ---------------------------------------------------------------------------
namespace {
struct BestD {
    double v;
    int i;
};
__device__ __forceinline__ BestD better_d(BestD a, BestD b) {       // numpy.argmax order: first NaN, else first maximum
    const bool an = a.v != a.v, bn = b.v != b.v;
    if (an || bn) {
        if (an && bn) return a.i < b.i ? a : b;
        return an ? a : b;
    }
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}
__global__ __launch_bounds__(256) void tracker_decode_kernel(const float* __restrict__ logits,
                                                             const float* __restrict__ boxes,
                                                             const double* __restrict__ window, float omwp, double wp,
                                                             double* __restrict__ out /* [B,6]: pscore,idx,cx,cy,w,h */,
                                                             double* __restrict__ pscore_map, int L) {
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ BestD s_best[4];
    const float2* lg = reinterpret_cast<const float2*>(logits) + (size_t)b * L;
    BestD best{-INFINITY, 0x7fffffff};
    for (int t = tid; t < L; t += 256) {
        const float2 l = lg[t];
        const float m = fmaxf(l.x, l.y);
        const float e0 = expf(l.x - m), e1 = expf(l.y - m);
        const float s = e0 / (e0 + e1);
        const double v = (double)(s * omwp) + window[t] * wp;
        if (pscore_map) pscore_map[(size_t)b * L + t] = v;
        best = better_d(best, BestD{v, t});
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        BestD other{__shfl_xor(best.v, o), __shfl_xor(best.i, o)};
        best = better_d(best, other);
    }
    if ((tid & 63) == 0) s_best[tid >> 6] = best;
    __syncthreads();
    if (tid != 0) return;
    best = better_d(better_d(s_best[0], s_best[1]), better_d(s_best[2], s_best[3]));
    const float4 bb = reinterpret_cast<const float4*>(boxes)[(size_t)b * L + best.i];
    double* o = out + 6 * (size_t)b;
    o[0] = best.v; o[1] = (double)best.i; o[2] = bb.x; o[3] = bb.y; o[4] = bb.z; o[5] = bb.w;
}
}  // namespace

extern "C" int spot_tracker_decode(const float* pred_logits, const float* pred_boxes, const double* window,
                                   double window_penalty, double* best6, double* pscore_map, int B, int L,
                                   hipStream_t stream) {
    SPOT_CHECK_ARG(B >= 0 && L > 0, "tracker_decode: bad B=%d L=%d", B, L);
    if (B == 0) return SPOT_OK;
    SPOT_CHECK_ARG(pred_logits && pred_boxes && window && best6, "tracker_decode: null pointer");
    tracker_decode_kernel<<<B, 256, 0, stream>>>(pred_logits, pred_boxes, window, (float)(1.0 - window_penalty),
                                                 window_penalty, best6, pscore_map, L);
    SPOT_CHECK_LAUNCH("tracker_decode");
    return SPOT_OK;
}
