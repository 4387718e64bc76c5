// HBM-bound elementwise kernels of the hot path: uint8 crop normalisation,
// teacher EMA, paired box IoU / GIoU (forward + backward).
#include "common.h"
#include "../../include/spot_hip.h"

// The reference evaluates these expressions as separate rounded multiplies and adds.  HIP's
// __fmul_rn/__fadd_rn are plain operators and hipcc fuses them by default, so this file is
// compiled with -ffp-contract=off (see the Makefile); explicit fmaf() calls stay fused.

namespace {

// ---------------------------------------------------------------------------
// uint8 -> normalised float (ltr/models/pre_input/necessary_preprocess.py:54-67,86-91)
//   out = (clamp(float(x) * scale[b], 0, 1) - mean[c]) / std[c]
// 16 pixels per lane: one 16-byte load, four 16-byte stores.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void preprocess_kernel(const uint4* __restrict__ img, const float* __restrict__ scale,
                                                         float3 mean, float3 stdv, float4* __restrict__ out,
                                                         long plane_vec /* H*W/16 */, int C, long total_vec) {
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < total_vec; v += (long)gridDim.x * 256) {
        const long plane = v / plane_vec;
        const int c = (int)(plane % C);
        const int b = (int)(plane / C);
        const float s = scale ? scale[b] : (1.0f / 255.0f);
        const float m = c == 0 ? mean.x : (c == 1 ? mean.y : mean.z);
        const float d = c == 0 ? stdv.x : (c == 1 ? stdv.y : stdv.z);
        const uint4 p = img[v];
        const unsigned w[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float4 o;
            o.x = __fdiv_rn(__fsub_rn(fminf(fmaxf(__fmul_rn((float)(w[k] & 0xff), s), 0.f), 1.f), m), d);
            o.y = __fdiv_rn(__fsub_rn(fminf(fmaxf(__fmul_rn((float)((w[k] >> 8) & 0xff), s), 0.f), 1.f), m), d);
            o.z = __fdiv_rn(__fsub_rn(fminf(fmaxf(__fmul_rn((float)((w[k] >> 16) & 0xff), s), 0.f), 1.f), m), d);
            o.w = __fdiv_rn(__fsub_rn(fminf(fmaxf(__fmul_rn((float)(w[k] >> 24), s), 0.f), 1.f), m), d);
            out[v * 4 + k] = o;
        }
    }
}

// The same map one byte per lane, for crops whose pixel count is not a multiple of 16 (the Siamese variant's 255^2
// and 127^2 crops, ltr/train_settings/spot/spot_ocean_003.py:86-87).
__global__ __launch_bounds__(256) void preprocess_bytes_kernel(const unsigned char* __restrict__ img,
                                                               const float* __restrict__ scale, float3 mean, float3 stdv,
                                                               float* __restrict__ out, long plane_px, int C, long total) {
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < total; v += (long)gridDim.x * 256) {
        const long plane = v / plane_px;
        const int c = (int)(plane % C);
        const int b = (int)(plane / C);
        const float s = scale ? scale[b] : (1.0f / 255.0f);
        const float m = c == 0 ? mean.x : (c == 1 ? mean.y : mean.z);
        const float d = c == 0 ? stdv.x : (c == 1 ? stdv.y : stdv.z);
        out[v] = __fdiv_rn(__fsub_rn(fminf(fmaxf(__fmul_rn((float)img[v], s), 0.f), 1.f), m), d);
    }
}

// ---------------------------------------------------------------------------
// teacher EMA (ltr/actor/transt_sparse_actor.py:11-33) on flat parameter storage
//   t = s * (1 - keep) + t * keep      (two rounded products, one rounded sum)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ema_kernel(float4* __restrict__ t, const float4* __restrict__ s, long n_vec,
                                                  float* __restrict__ t_tail, const float* __restrict__ s_tail,
                                                  int n_tail, float omk, float k) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (long)gridDim.x * 256) {
        float4 a = t[i];
        const float4 b = s[i];
        a.x = __fadd_rn(__fmul_rn(b.x, omk), __fmul_rn(a.x, k));
        a.y = __fadd_rn(__fmul_rn(b.y, omk), __fmul_rn(a.y, k));
        a.z = __fadd_rn(__fmul_rn(b.z, omk), __fmul_rn(a.z, k));
        a.w = __fadd_rn(__fmul_rn(b.w, omk), __fmul_rn(a.w, k));
        t[i] = a;
    }
    if (blockIdx.x == 0 && threadIdx.x < n_tail)
        t_tail[threadIdx.x] = __fadd_rn(__fmul_rn(s_tail[threadIdx.x], omk), __fmul_rn(t_tail[threadIdx.x], k));
}

// integer buffers (BatchNorm num_batches_tracked): the reference's expression
// promotes to float32 and load_state_dict truncates back to int64.
__global__ void ema_i64_kernel(long long* __restrict__ t, const long long* __restrict__ s, int n, float omk, float k) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) t[i] = (long long)__fadd_rn(__fmul_rn((float)s[i], omk), __fmul_rn((float)t[i], k));
}

// ---------------------------------------------------------------------------
// paired IoU / GIoU on cxcywh boxes (ltr/utils/box_ops.py:9-13,105-156)
// ---------------------------------------------------------------------------
struct XYXY {
    float x0, y0, x1, y1;
};
__device__ __forceinline__ XYXY to_xyxy(float4 b) {
    return XYXY{__fsub_rn(b.x, __fmul_rn(0.5f, b.z)), __fsub_rn(b.y, __fmul_rn(0.5f, b.w)),
                __fadd_rn(b.x, __fmul_rn(0.5f, b.z)), __fadd_rn(b.y, __fmul_rn(0.5f, b.w))};
}

template <bool CXCYWH>
__global__ __launch_bounds__(256) void giou_fwd_kernel(const float4* __restrict__ b1, const float4* __restrict__ b2,
                                                       float* __restrict__ giou, float* __restrict__ iou,
                                                       float* __restrict__ uni, int P) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    XYXY a, b;
    if (CXCYWH) {
        a = to_xyxy(b1[i]);
        b = to_xyxy(b2[i]);
    } else {
        float4 p = b1[i], q = b2[i];
        a = XYXY{p.x, p.y, p.z, p.w};
        b = XYXY{q.x, q.y, q.z, q.w};
    }
    const float a1 = __fmul_rn(__fsub_rn(a.x1, a.x0), __fsub_rn(a.y1, a.y0));
    const float a2 = __fmul_rn(__fsub_rn(b.x1, b.x0), __fsub_rn(b.y1, b.y0));
    const float iw = fmaxf(__fsub_rn(fminf(a.x1, b.x1), fmaxf(a.x0, b.x0)), 0.f);
    const float ih = fmaxf(__fsub_rn(fminf(a.y1, b.y1), fmaxf(a.y0, b.y0)), 0.f);
    const float inter = __fmul_rn(iw, ih);
    const float u = __fsub_rn(__fadd_rn(a1, a2), inter);
    const float io = __fdiv_rn(inter, u);
    if (iou) iou[i] = io;
    if (uni) uni[i] = u;
    if (giou) {
        const float hw = fmaxf(__fsub_rn(fmaxf(a.x1, b.x1), fminf(a.x0, b.x0)), 0.f);
        const float hh = fmaxf(__fsub_rn(fmaxf(a.y1, b.y1), fminf(a.y0, b.y0)), 0.f);
        const float area = __fmul_rn(hw, hh);
        giou[i] = __fsub_rn(io, __fdiv_rn(__fsub_rn(area, u), area));
    }
}

// Gradient of (giou, iou) w.r.t. the first box set (cxcywh), given upstream
// d_giou[P] and d_iou[P] (either may be null).  Sub-gradients follow ATen:
// max/min send the gradient to the selected operand and split it in half on
// ties; clamp(min=0) passes the gradient where the argument is >= 0.
__device__ __forceinline__ void dmax(float g, float p, float q, float& gp, float& gq) {
    if (p > q) gp += g;
    else if (p < q) gq += g;
    else { gp += 0.5f * g; gq += 0.5f * g; }
}
__device__ __forceinline__ void dmin(float g, float p, float q, float& gp, float& gq) {
    if (p < q) gp += g;
    else if (p > q) gq += g;
    else { gp += 0.5f * g; gq += 0.5f * g; }
}

__device__ __forceinline__ float4 giou_grad(float4 p, float4 q, float gg, float di) {
    const XYXY a = to_xyxy(p), b = to_xyxy(q);
    const float w1 = a.x1 - a.x0, h1 = a.y1 - a.y0;
    const float a1 = w1 * h1, a2 = (b.x1 - b.x0) * (b.y1 - b.y0);
    const float ix0 = fmaxf(a.x0, b.x0), iy0 = fmaxf(a.y0, b.y0), ix1 = fminf(a.x1, b.x1), iy1 = fminf(a.y1, b.y1);
    const float iwr = ix1 - ix0, ihr = iy1 - iy0;
    const float iw = fmaxf(iwr, 0.f), ih = fmaxf(ihr, 0.f);
    const float inter = iw * ih;
    const float u = a1 + a2 - inter;
    const float hx0 = fminf(a.x0, b.x0), hy0 = fminf(a.y0, b.y0), hx1 = fmaxf(a.x1, b.x1), hy1 = fmaxf(a.y1, b.y1);
    const float hwr = hx1 - hx0, hhr = hy1 - hy0;
    const float hw = fmaxf(hwr, 0.f), hh = fmaxf(hhr, 0.f);
    const float area = hw * hh;
    const float gi = di + gg;                            // giou = iou - (area-u)/area
    // iou = inter/u ; giou extra term -(area-u)/area = -1 + u/area
    float g_inter = gi / u;
    float g_u = -gi * inter / (u * u) + gg / area;
    float g_area = -gg * u / (area * area);
    // u = a1 + a2 - inter
    float g_a1 = g_u;
    g_inter -= g_u;
    // inter = iw*ih with clamps
    float g_iwr = (iwr >= 0.f) ? g_inter * ih : 0.f;
    float g_ihr = (ihr >= 0.f) ? g_inter * iw : 0.f;
    float g_hwr = (hwr >= 0.f) ? g_area * hh : 0.f;
    float g_hhr = (hhr >= 0.f) ? g_area * hw : 0.f;
    float gx0 = 0.f, gy0 = 0.f, gx1 = 0.f, gy1 = 0.f, dump = 0.f;
    dmin(g_iwr, a.x1, b.x1, gx1, dump);
    dmax(-g_iwr, a.x0, b.x0, gx0, dump);
    dmin(g_ihr, a.y1, b.y1, gy1, dump);
    dmax(-g_ihr, a.y0, b.y0, gy0, dump);
    dmax(g_hwr, a.x1, b.x1, gx1, dump);
    dmin(-g_hwr, a.x0, b.x0, gx0, dump);
    dmax(g_hhr, a.y1, b.y1, gy1, dump);
    dmin(-g_hhr, a.y0, b.y0, gy0, dump);
    // a1 = (x1-x0)*(y1-y0)
    gx1 += g_a1 * h1; gx0 -= g_a1 * h1;
    gy1 += g_a1 * w1; gy0 -= g_a1 * w1;
    // xyxy -> cxcywh
    return make_float4(gx0 + gx1, gy0 + gy1, 0.5f * (gx1 - gx0), 0.5f * (gy1 - gy0));
}

__global__ __launch_bounds__(256) void giou_bwd_kernel(const float4* __restrict__ b1, const float4* __restrict__ b2,
                                                       const float* __restrict__ d_giou, const float* __restrict__ d_iou,
                                                       float4* __restrict__ d_b1, int P) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    d_b1[i] = giou_grad(b1[i], b2[i], d_giou ? d_giou[i] : 0.f, d_iou ? d_iou[i] : 0.f);
}

// (iou, giou) of one pair of cxcywh boxes, rounded operation by operation as the element-wise reference is.
__device__ __forceinline__ void giou_pair(float4 p, float4 q, float& io, float& gio) {
    const XYXY a = to_xyxy(p), b = to_xyxy(q);
    const float a1 = __fmul_rn(__fsub_rn(a.x1, a.x0), __fsub_rn(a.y1, a.y0));
    const float a2 = __fmul_rn(__fsub_rn(b.x1, b.x0), __fsub_rn(b.y1, b.y0));
    const float iw = fmaxf(__fsub_rn(fminf(a.x1, b.x1), fmaxf(a.x0, b.x0)), 0.f);
    const float ih = fmaxf(__fsub_rn(fminf(a.y1, b.y1), fmaxf(a.y0, b.y0)), 0.f);
    const float inter = __fmul_rn(iw, ih);
    const float u = __fsub_rn(__fadd_rn(a1, a2), inter);
    io = __fdiv_rn(inter, u);
    const float hw = fmaxf(__fsub_rn(fmaxf(a.x1, b.x1), fminf(a.x0, b.x0)), 0.f);
    const float hh = fmaxf(__fsub_rn(fmaxf(a.y1, b.y1), fminf(a.y0, b.y0)), 0.f);
    const float area = __fmul_rn(hw, hh);
    gio = __fsub_rn(io, __fdiv_rn(__fsub_rn(area, u), area));
}

// out = keep(seed,row,col) ? x / (1-p) : 0 with the mask function the GEMM epilogue uses (row = token row,
// col = channel), so a backward pass can re-apply the forward's mask to the incoming gradient.
__global__ __launch_bounds__(256) void dropout_kernel(const float4* __restrict__ x, float4* __restrict__ out, long rows,
                                                      int cols4, unsigned thr, unsigned seed, float scale) {
    const long total = rows * cols4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / cols4;
        const int c = (int)(i - r * cols4) * 4;
        const unsigned rk = spot_row_key(seed, (unsigned)r);
        float4 v = x[i];
        v.x = spot_keep(rk, c, thr) ? v.x * scale : 0.f;
        v.y = spot_keep(rk, c + 1, thr) ? v.y * scale : 0.f;
        v.z = spot_keep(rk, c + 2, thr) ? v.z * scale : 0.f;
        v.w = spot_keep(rk, c + 3, thr) ? v.w * scale : 0.f;
        out[i] = v;
    }
}

}  // namespace

extern "C" int spot_dropout_f32(const float* x, float* out, long rows, int cols, float p, unsigned seed,
                                hipStream_t stream) {
    SPOT_CHECK_ARG(rows >= 0 && cols > 0 && cols % 4 == 0, "dropout: cols=%d must be a positive multiple of 4", cols);
    SPOT_CHECK_ARG(p >= 0.f && p < 1.f, "dropout: p=%f outside [0,1)", p);
    if (rows == 0) return SPOT_OK;
    SPOT_CHECK_ARG(x && out && (((uintptr_t)x | (uintptr_t)out) & 15) == 0, "dropout: buffers must be non-null and 16-byte aligned");
    const long total = rows * (cols / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    dropout_kernel<<<grid, 256, 0, stream>>>((const float4*)x, (float4*)out, rows, cols / 4, spot_drop_threshold(p), seed,
                                             1.f / (1.f - p));
    SPOT_CHECK_LAUNCH("dropout");
    return SPOT_OK;
}

extern "C" int spot_preprocess_u8_f32(const unsigned char* images, const float* scale, const float* mean3,
                                      const float* std3, float* out, int B, int C, int H, int W,
                                      hipStream_t stream) {
    SPOT_CHECK_ARG(images && out && mean3 && std3, "preprocess: null pointer");
    SPOT_CHECK_ARG(B >= 0 && C == 3 && H > 0 && W > 0, "preprocess: bad shape B=%d C=%d H=%d W=%d (C must be 3)", B, C,
                   H, W);
    if (B == 0) return SPOT_OK;
    if (((long)H * W) % 16 != 0 || (uintptr_t)images % 16 != 0 || (uintptr_t)out % 16 != 0) {
        const long total = (long)H * W * C * B;
        int grid = spot_ceil_div(total, 256);
        if (grid > 65535) grid = 65535;
        preprocess_bytes_kernel<<<grid, 256, 0, stream>>>(images, scale, make_float3(mean3[0], mean3[1], mean3[2]),
                                                          make_float3(std3[0], std3[1], std3[2]), out, (long)H * W, C, total);
        SPOT_CHECK_LAUNCH("preprocess");
        return SPOT_OK;
    }
    const long plane_vec = (long)H * W / 16;
    const long total = plane_vec * C * B;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    preprocess_kernel<<<grid, 256, 0, stream>>>((const uint4*)images, scale, make_float3(mean3[0], mean3[1], mean3[2]),
                                                make_float3(std3[0], std3[1], std3[2]), (float4*)out, plane_vec, C,
                                                total);
    SPOT_CHECK_LAUNCH("preprocess");
    return SPOT_OK;
}

extern "C" int spot_ema_lerp_f32(float* teacher, const float* student, long n, double keep_rate, hipStream_t stream) {
    SPOT_CHECK_ARG(teacher && student && n >= 0, "ema_lerp: bad arguments");
    SPOT_CHECK_ARG((uintptr_t)teacher % 16 == 0 && (uintptr_t)student % 16 == 0, "ema_lerp: buffers must be 16-byte aligned");
    if (n == 0) return SPOT_OK;
    const long n_vec = n / 4;
    const int n_tail = (int)(n - n_vec * 4);
    int grid = (int)((n_vec + 255) / 256);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    ema_kernel<<<grid, 256, 0, stream>>>((float4*)teacher, (const float4*)student, n_vec, teacher + n_vec * 4,
                                         student + n_vec * 4, n_tail, (float)(1.0 - keep_rate), (float)keep_rate);
    SPOT_CHECK_LAUNCH("ema_lerp");
    return SPOT_OK;
}

extern "C" int spot_ema_lerp_i64(long long* teacher, const long long* student, int n, double keep_rate,
                                 hipStream_t stream) {
    SPOT_CHECK_ARG(teacher && student && n >= 0, "ema_lerp_i64: bad arguments");
    if (n == 0) return SPOT_OK;
    ema_i64_kernel<<<(n + 255) / 256, 256, 0, stream>>>(teacher, student, n, (float)(1.0 - keep_rate), (float)keep_rate);
    SPOT_CHECK_LAUNCH("ema_lerp_i64");
    return SPOT_OK;
}

extern "C" int spot_box_iou_pair_f32(const float* boxes1, const float* boxes2, int cxcywh, float* giou, float* iou,
                                     float* union_area, int P, hipStream_t stream) {
    SPOT_CHECK_ARG(boxes1 && boxes2 && P >= 0, "box_iou_pair: bad arguments");
    SPOT_CHECK_ARG(giou || iou || union_area, "box_iou_pair: no output requested");
    if (P == 0) return SPOT_OK;
    const int grid = (P + 255) / 256;
    if (cxcywh)
        giou_fwd_kernel<true><<<grid, 256, 0, stream>>>((const float4*)boxes1, (const float4*)boxes2, giou, iou, union_area, P);
    else
        giou_fwd_kernel<false><<<grid, 256, 0, stream>>>((const float4*)boxes1, (const float4*)boxes2, giou, iou, union_area, P);
    SPOT_CHECK_LAUNCH("box_iou_pair");
    return SPOT_OK;
}

extern "C" int spot_box_giou_pair_bwd_f32(const float* boxes1_cxcywh, const float* boxes2_cxcywh, const float* d_giou,
                                          const float* d_iou, float* d_boxes1, int P, hipStream_t stream) {
    SPOT_CHECK_ARG(boxes1_cxcywh && boxes2_cxcywh && d_boxes1 && P >= 0, "box_giou_pair_bwd: bad arguments");
    SPOT_CHECK_ARG(d_giou || d_iou, "box_giou_pair_bwd: no upstream gradient");
    if (P == 0) return SPOT_OK;
    giou_bwd_kernel<<<(P + 255) / 256, 256, 0, stream>>>((const float4*)boxes1_cxcywh, (const float4*)boxes2_cxcywh,
                                                         d_giou, d_iou, (float4*)d_boxes1, P);
    SPOT_CHECK_LAUNCH("box_giou_pair_bwd");
    return SPOT_OK;
}

// ---------------------------------------------------------------------------
// AdamW over flat parameter storage: one launch for the whole network.
// torch.optim.AdamW (decoupled weight decay, no amsgrad) as the reference configures it
// (ltr/train_settings/spot/spot_transt_003.py:148-157: lr 1e-4, backbone 1e-5, weight decay 1e-4).
// The flat buffer also holds non-trainable slots (BatchNorm running statistics) and alignment
// gaps, so the work list is a table of chunks (offset, count, group); a workgroup takes one chunk.
// Per element: 16 bytes read (p, g, m, v) and 12 written: HBM-bound.
// ---------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    const spot_adamw_chunk* __restrict__ chunks,
                                                    spot_adamw_groups hp) {
    const spot_adamw_chunk c = chunks[blockIdx.x];
    const float lr = hp.lr[c.group], wd = hp.weight_decay[c.group];
    const float b1 = hp.beta1, b2 = hp.beta2, eps = hp.eps;
    const float step_size = lr / hp.bias_correction1, inv_sqrt_bc2 = 1.f / hp.sqrt_bias_correction2;
    const float decay = 1.f - lr * wd;
    const long base = c.offset;
    if ((base & 3) == 0 && (c.count & 3) == 0) {
        const int n4 = c.count >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) {
            float4 pp = reinterpret_cast<float4*>(p + base)[i];
            const float4 gg = reinterpret_cast<const float4*>(g + base)[i];
            float4 mm = reinterpret_cast<float4*>(m + base)[i];
            float4 vv = reinterpret_cast<float4*>(v + base)[i];
#define SPOT_ADAMW_1(x)                                              \
    pp.x *= decay;                                                   \
    mm.x = b1 * mm.x + (1.f - b1) * gg.x;                            \
    vv.x = b2 * vv.x + (1.f - b2) * gg.x * gg.x;                     \
    pp.x -= step_size * (mm.x / (sqrtf(vv.x) * inv_sqrt_bc2 + eps));
            SPOT_ADAMW_1(x) SPOT_ADAMW_1(y) SPOT_ADAMW_1(z) SPOT_ADAMW_1(w)
#undef SPOT_ADAMW_1
            reinterpret_cast<float4*>(p + base)[i] = pp;
            reinterpret_cast<float4*>(m + base)[i] = mm;
            reinterpret_cast<float4*>(v + base)[i] = vv;
        }
    } else {
        for (int i = threadIdx.x; i < c.count; i += 256) {
            float pp = p[base + i] * decay;
            const float gg = g[base + i];
            const float mm = b1 * m[base + i] + (1.f - b1) * gg;
            const float vv = b2 * v[base + i] + (1.f - b2) * gg * gg;
            pp -= step_size * (mm / (sqrtf(vv) * inv_sqrt_bc2 + eps));
            p[base + i] = pp;
            m[base + i] = mm;
            v[base + i] = vv;
        }
    }
}
}  // namespace

extern "C" int spot_adamw_f32(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                              const spot_adamw_chunk* chunks_dev, int n_chunks, const spot_adamw_groups* hyper,
                              hipStream_t stream) {
    SPOT_CHECK_ARG(n_chunks >= 0 && hyper, "adamw: bad arguments");
    if (n_chunks == 0) return SPOT_OK;
    SPOT_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && chunks_dev, "adamw: null pointer");
    SPOT_CHECK_ARG(hyper->bias_correction1 > 0.f && hyper->sqrt_bias_correction2 > 0.f, "adamw: bias corrections must be > 0");
    adamw_kernel<<<n_chunks, 256, 0, stream>>>(params, grads, exp_avg, exp_avg_sq, chunks_dev, *hyper);
    SPOT_CHECK_LAUNCH("adamw");
    return SPOT_OK;
}

// ---------------------------------------------------------------------------
// Sine position encoding (ltr/models/neck/position_encoding.py:27-45, normalize = True): per clip, the running count
// of unmasked cells down each column (y) and along each row (x), divided by the column's / row's last count + 1e-6
// and scaled; feature c of an axis is sin (c even) or cos (c odd) of that value / temperature^(2 floor(c/2) / F).
// Output as token rows [B, H*W, 2F]: F features of y then F of x.  One workgroup per clip; the two count maps live
// in LDS.
// ---------------------------------------------------------------------------
namespace {
constexpr int POSENC_MAX_CELLS = 4096;
__global__ __launch_bounds__(256) void posenc_sine_kernel(const unsigned char* __restrict__ mask, float* __restrict__ out,
                                                          int H, int W, int F, float temperature, float scale) {
    __shared__ float ye[POSENC_MAX_CELLS], xe[POSENC_MAX_CELLS];
    const int b = blockIdx.x, tid = threadIdx.x, HW = H * W;
    const unsigned char* m = mask ? mask + (long)b * HW : nullptr;
    for (int x = tid; x < W; x += 256) {            // counts down a column
        float run = 0.f;
        for (int y = 0; y < H; ++y) {
            run += (m && m[y * W + x]) ? 0.f : 1.f;
            ye[y * W + x] = run;
        }
    }
    for (int y = tid; y < H; y += 256) {            // counts along a row
        float run = 0.f;
        for (int x = 0; x < W; ++x) {
            run += (m && m[y * W + x]) ? 0.f : 1.f;
            xe[y * W + x] = run;
        }
    }
    __syncthreads();
    const int C = 2 * F;
    float* o = out + (long)b * HW * C;
    for (long i = tid; i < (long)HW * C; i += 256) {
        const int t = (int)(i / C), c = (int)(i % C), y = t / W, x = t % W;
        const int f = c < F ? c : c - F;
        const float e = c < F ? ye[t] / (ye[(H - 1) * W + x] + 1e-6f) * scale : xe[t] / (xe[y * W + W - 1] + 1e-6f) * scale;
        const float dim_t = powf(temperature, 2.f * (float)(f / 2) / (float)F);
        const float a = e / dim_t;
        o[i] = (f & 1) ? cosf(a) : sinf(a);
    }
}
}  // namespace

extern "C" int spot_posenc_sine_f32(const unsigned char* mask, float* out, int B, int H, int W, int num_pos_feats,
                                    float temperature, float scale, hipStream_t stream) {
    SPOT_CHECK_ARG(out && B >= 0 && H > 0 && W > 0 && num_pos_feats > 0, "posenc_sine: bad arguments");
    SPOT_CHECK_ARG((long)H * W <= POSENC_MAX_CELLS, "posenc_sine: H*W=%ld beyond %d cells", (long)H * W, POSENC_MAX_CELLS);
    if (B == 0) return SPOT_OK;
    posenc_sine_kernel<<<B, 256, 0, stream>>>(mask, out, H, W, num_pos_feats, temperature, scale);
    SPOT_CHECK_LAUNCH("posenc_sine");
    return SPOT_OK;
}

// ---------------------------------------------------------------------------
// Global gradient norm and clipping over the flat gradient buffer (torch.nn.utils.clip_grad_norm_(parameters, max_norm)
// as the trainer applies it to the Siamese variant, ltr/trainers/ltr_sparse_trainer.py:126-131): norm = sqrt(sum g^2)
// (block partials in float, their sum in double, no atomics: the same bits on every run and every rank),
// g *= max_norm / (norm + 1e-6) when that factor is below 1 and the norm is finite.
// ---------------------------------------------------------------------------
namespace {
constexpr int NORM_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float4* __restrict__ g, long n4, float* __restrict__ part) {
    __shared__ float s[4];
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = g[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}
__global__ __launch_bounds__(256) void norm_finish_kernel(const float* __restrict__ part, int nblocks, float* __restrict__ norm_out) {
    __shared__ double s[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) acc += (double)part[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) norm_out[0] = (float)sqrt(s[0]);
}
__global__ __launch_bounds__(256) void clip_scale_kernel(float4* __restrict__ g, long n4, const float* __restrict__ norm, float max_norm) {
    const float nv = norm[0];
    const float coef = max_norm / (nv + 1e-6f);
    if (!(nv == nv) || nv > 3.0e38f || !(coef < 1.f)) return;          // non-finite norm: the caller skips the step
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 v = g[i];
        v.x *= coef; v.y *= coef; v.z *= coef; v.w *= coef;
        g[i] = v;
    }
}
}  // namespace

extern "C" long spot_clip_grad_norm_ws_floats(void) { return NORM_BLOCKS; }

extern "C" int spot_clip_grad_norm_f32(float* grads, long n, float max_norm, float* norm_out, float* ws, hipStream_t stream) {
    SPOT_CHECK_ARG(grads && norm_out && ws && n >= 0, "clip_grad_norm: bad arguments");
    SPOT_CHECK_ARG(n % 4 == 0 && ((uintptr_t)grads & 15) == 0, "clip_grad_norm: the buffer must be 16-byte aligned with a length that is a multiple of 4");
    sumsq_partial_kernel<<<NORM_BLOCKS, 256, 0, stream>>>((const float4*)grads, n / 4, ws);
    SPOT_CHECK_LAUNCH("clip_grad_norm(partials)");
    norm_finish_kernel<<<1, 256, 0, stream>>>(ws, NORM_BLOCKS, norm_out);
    SPOT_CHECK_LAUNCH("clip_grad_norm(finish)");
    if (max_norm > 0.f) {
        clip_scale_kernel<<<2048, 256, 0, stream>>>((float4*)grads, n / 4, norm_out, max_norm);
        SPOT_CHECK_LAUNCH("clip_grad_norm(scale)");
    }
    return SPOT_OK;
}

// ---------------------------------------------------------------------------
// Set criterion of the transformer tracker in two launches (ltr/models/tracking/transt_criterion.py:40-109,136-170):
// class-weighted cross entropy over every token (positives are class 0, everything else the no-object class),
// L1 + (1 - GIoU) on the positive tokens, and the L1 of the IoU head against the detached IoU on the positive and the
// negative tokens.  Token sets come as dense maps (spot_match_masks); the target box of a token is its clip's box.
//   sums per workgroup: 0 w*nll, 1 w, 2 pos L1, 3 pos (1-giou), 4 pos iou, 5 pos |pred-iou|, 6 neg*|pred-iou|,
//                       7 #pos, 8 sum of neg multiplicities
// Partials in float, their sum in double and in a fixed order: the same bits on every run.
// ---------------------------------------------------------------------------
namespace {
constexpr int LOSS_TERMS = 9;
constexpr int LOSS_MAX_BLOCKS = 4096;

__global__ __launch_bounds__(256) void transt_loss_partial_kernel(
    const float2* __restrict__ logits, const float4* __restrict__ boxes, const float* __restrict__ iou_pred,
    const float4* __restrict__ gt, const unsigned char* __restrict__ pos, const unsigned char* __restrict__ neg,
    float w_obj, float w_eos, int L, long total, float* __restrict__ part) {
    __shared__ float s[4][LOSS_TERMS];
    float t[LOSS_TERMS];
#pragma unroll
    for (int k = 0; k < LOSS_TERMS; ++k) t[k] = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const bool is_pos = pos[i] != 0;
        const float nmul = neg ? (float)neg[i] : 0.f;
        const float2 x = logits[i];
        const float m = fmaxf(x.x, x.y);
        const float lse = m + logf(expf(x.x - m) + expf(x.y - m));
        const float w = is_pos ? w_obj : w_eos;
        t[0] += w * (lse - (is_pos ? x.x : x.y));
        t[1] += w;
        if (is_pos || nmul > 0.f) {
            const float4 b = boxes[i], g = gt[i / L];
            float io, gio;
            giou_pair(b, g, io, gio);
            const float err = iou_pred ? fabsf(iou_pred[i] - io) : 0.f;
            if (is_pos) {
                t[2] += (fabsf(b.x - g.x) + fabsf(b.y - g.y)) + (fabsf(b.z - g.z) + fabsf(b.w - g.w));
                t[3] += 1.f - gio;
                t[4] += io;
                t[5] += err;
                t[7] += 1.f;
            }
            if (nmul > 0.f) {
                t[6] += err * nmul;
                t[8] += nmul;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < LOSS_TERMS; ++k) {
        const float v = wave_sum(t[k]);
        if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < LOSS_TERMS)
        part[blockIdx.x * LOSS_TERMS + threadIdx.x] =
            (s[0][threadIdx.x] + s[1][threadIdx.x]) + (s[2][threadIdx.x] + s[3][threadIdx.x]);
}

// losses = (loss_ce, loss_bbox, loss_giou, iou, loss_iou_pred); norms = (sum of class weights, max(#pos,1), max(#neg,1))
__global__ __launch_bounds__(64) void transt_loss_finish_kernel(const float* __restrict__ part, int nblocks,
                                                                int has_iou_head, float* __restrict__ losses,
                                                                float* __restrict__ norms) {
    __shared__ double tot[LOSS_TERMS];
    if (threadIdx.x < LOSS_TERMS) {
        double acc = 0.0;
        for (int b = 0; b < nblocks; ++b) acc += (double)part[b * LOSS_TERMS + threadIdx.x];
        tot[threadIdx.x] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double n_pos = tot[7] > 1.0 ? tot[7] : 1.0, n_neg = tot[8] > 1.0 ? tot[8] : 1.0;
        losses[0] = (float)(tot[0] / tot[1]);
        losses[1] = (float)(tot[2] / n_pos);
        losses[2] = (float)(tot[3] / n_pos);
        losses[3] = (float)(tot[4] / n_pos);
        losses[4] = has_iou_head ? (float)(tot[5] / n_pos + tot[6] / n_neg) : 0.f;
        norms[0] = (float)tot[1];
        norms[1] = (float)n_pos;
        norms[2] = (float)n_neg;
    }
}

__device__ __forceinline__ float sgn(float v) { return (float)((v > 0.f) - (v < 0.f)); }

// Upstream gradients of (loss_ce, loss_bbox, loss_giou, loss_iou_pred) are read on the device (null = 0).  The IoU target
// of the IoU head is a constant, so the head's loss sends nothing to the boxes (transt_criterion.py:92,103).
__global__ __launch_bounds__(256) void transt_loss_bwd_kernel(
    const float2* __restrict__ logits, const float4* __restrict__ boxes, const float* __restrict__ iou_pred,
    const float4* __restrict__ gt, const unsigned char* __restrict__ pos, const unsigned char* __restrict__ neg,
    float w_obj, float w_eos, const float* __restrict__ norms, const float* __restrict__ g_ce,
    const float* __restrict__ g_bbox, const float* __restrict__ g_giou, const float* __restrict__ g_pred, int L,
    long total, float2* __restrict__ d_logits, float4* __restrict__ d_boxes, float* __restrict__ d_pred) {
    const float k_ce = (g_ce ? g_ce[0] : 0.f) / norms[0];
    const float k_l1 = (g_bbox ? g_bbox[0] : 0.f) / norms[1];
    const float k_gi = (g_giou ? g_giou[0] : 0.f) / norms[1];
    const float k_pp = (g_pred ? g_pred[0] : 0.f) / norms[1];
    const float k_pn = (g_pred ? g_pred[0] : 0.f) / norms[2];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const bool is_pos = pos[i] != 0;
        const float nmul = neg ? (float)neg[i] : 0.f;
        if (d_logits) {
            const float2 x = logits[i];
            const float m = fmaxf(x.x, x.y);
            const float e0 = expf(x.x - m), e1 = expf(x.y - m), inv = 1.f / (e0 + e1);
            const float k = k_ce * (is_pos ? w_obj : w_eos);
            d_logits[i] = make_float2(k * (e0 * inv - (is_pos ? 1.f : 0.f)), k * (e1 * inv - (is_pos ? 0.f : 1.f)));
        }
        float4 db = make_float4(0.f, 0.f, 0.f, 0.f);
        float dp = 0.f;
        if (is_pos || nmul > 0.f) {
            const float4 b = boxes[i], g = gt[i / L];
            if (is_pos && d_boxes) {
                db = giou_grad(b, g, -k_gi, 0.f);
                db.x += k_l1 * sgn(b.x - g.x);
                db.y += k_l1 * sgn(b.y - g.y);
                db.z += k_l1 * sgn(b.z - g.z);
                db.w += k_l1 * sgn(b.w - g.w);
            }
            if (d_pred) {
                float io, gio;
                giou_pair(b, g, io, gio);
                const float sg = sgn(iou_pred[i] - io);
                dp = (is_pos ? k_pp * sg : 0.f) + k_pn * nmul * sg;
            }
        }
        if (d_boxes) d_boxes[i] = db;
        if (d_pred) d_pred[i] = dp;
    }
}
}  // namespace

extern "C" long spot_transt_loss_ws_floats(void) { return (long)LOSS_MAX_BLOCKS * LOSS_TERMS; }

static int loss_grid(long total) {
    long g = (total + 255) / 256;
    return (int)(g > LOSS_MAX_BLOCKS ? LOSS_MAX_BLOCKS : g);
}

extern "C" int spot_transt_loss_f32(const float* logits, const float* boxes, const float* iou_pred, const float* gt,
                                    const unsigned char* pos, const unsigned char* neg, float w_obj, float w_eos,
                                    int B, int L, float* losses, float* norms, float* ws, hipStream_t stream) {
    SPOT_CHECK_ARG(logits && boxes && gt && pos && losses && norms && ws, "transt_loss: null pointer");
    SPOT_CHECK_ARG(B > 0 && L > 0, "transt_loss: bad shape B=%d L=%d", B, L);
    SPOT_CHECK_ARG((((uintptr_t)logits & 7) | ((uintptr_t)boxes & 15) | ((uintptr_t)gt & 15)) == 0,
                   "transt_loss: logits must be 8-byte aligned, boxes and targets 16-byte aligned");
    const long total = (long)B * L;
    const int grid = loss_grid(total);
    transt_loss_partial_kernel<<<grid, 256, 0, stream>>>((const float2*)logits, (const float4*)boxes, iou_pred,
                                                         (const float4*)gt, pos, neg, w_obj, w_eos, L, total, ws);
    SPOT_CHECK_LAUNCH("transt_loss(partials)");
    transt_loss_finish_kernel<<<1, 64, 0, stream>>>(ws, grid, iou_pred != nullptr, losses, norms);
    SPOT_CHECK_LAUNCH("transt_loss(finish)");
    return SPOT_OK;
}

extern "C" int spot_transt_loss_bwd_f32(const float* logits, const float* boxes, const float* iou_pred, const float* gt,
                                        const unsigned char* pos, const unsigned char* neg, float w_obj, float w_eos,
                                        const float* norms, const float* g_ce, const float* g_bbox, const float* g_giou,
                                        const float* g_iou_pred, int B, int L, float* d_logits, float* d_boxes,
                                        float* d_iou_pred, hipStream_t stream) {
    SPOT_CHECK_ARG(logits && boxes && gt && pos && norms, "transt_loss_bwd: null pointer");
    SPOT_CHECK_ARG(B > 0 && L > 0, "transt_loss_bwd: bad shape B=%d L=%d", B, L);
    SPOT_CHECK_ARG(d_logits || d_boxes || d_iou_pred, "transt_loss_bwd: no gradient requested");
    SPOT_CHECK_ARG(!d_iou_pred || iou_pred, "transt_loss_bwd: a gradient for the IoU head needs its predictions");
    SPOT_CHECK_ARG((((uintptr_t)logits & 7) | ((uintptr_t)boxes & 15) | ((uintptr_t)gt & 15) | ((uintptr_t)d_logits & 7) |
                    ((uintptr_t)d_boxes & 15)) == 0,
                   "transt_loss_bwd: logits must be 8-byte aligned, boxes and targets 16-byte aligned");
    const long total = (long)B * L;
    transt_loss_bwd_kernel<<<loss_grid(total), 256, 0, stream>>>(
        (const float2*)logits, (const float4*)boxes, iou_pred, (const float4*)gt, pos, neg, w_obj, w_eos, norms, g_ce,
        g_bbox, g_giou, g_iou_pred, L, total, (float2*)d_logits, (float4*)d_boxes, d_iou_pred);
    SPOT_CHECK_LAUNCH("transt_loss_bwd");
    return SPOT_OK;
}
