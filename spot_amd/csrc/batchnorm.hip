// BatchNorm2d fused with what follows it in the ResNet bottleneck
// (ltr/models/backbone/resnet.py:72-92: bn -> relu, and bn3 -> += residual -> relu;
// stem :196-198), NCHW fp32.
//
// All three are HBM-bound.  The reference's eager sequence reads and writes the
// activation once per op (batch_norm: 2 reads + 1 write, relu: 1 + 1, residual add:
// 2 + 1); here a layer is
//   training forward : stats pass (1 read)  +  apply pass (1 read [+ residual] , 1 write, 1/16 write of a sign byte)
//   training backward: reduce pass (dy, x, sign byte)  +  apply pass (dy, x, sign byte -> dx [, d_residual])
//   inference        : one pass
// Statistics are accumulated around a per-channel shift (the channel's first element),
// which keeps E[(x-K)^2] - E[x-K]^2 free of cancellation without a second read.
// One workgroup owns a (channel, slice of the batch) pair; slices of a channel are
// combined by every consumer workgroup from the per-slice partials (at most 64 pairs),
// so there is no finalize launch and no atomics: results are bit-reproducible.
#include "common.h"
#include "bn_train.h"
#include "../../include/spot_hip.h"

namespace {

using spot_bn::block_sum;
using spot_bn::combine;
using spot_bn::split_for;

struct Slice {
    int n0, n1;
};
// A plane is walked in packs of VW floats: VW = 4 (one 16-byte access) when H*W is a multiple of 4, VW = 1 otherwise
// (the 25x25 and 29x29 maps of the Siamese head, ltr/models/neck/ocean_fusion.py:56-85,126-158).  The sums are
// grouped (e0 + e1) + (e2 + e3) within a pack.
template <int VW>
struct Pack {
    float e[VW];
};
template <int VW>
__device__ __forceinline__ Pack<VW> ld(const float* __restrict__ p, size_t off) {
    Pack<VW> r;
    if constexpr (VW == 4) {
        const float4 v = reinterpret_cast<const float4*>(p)[off];
        r.e[0] = v.x; r.e[1] = v.y; r.e[2] = v.z; r.e[3] = v.w;
    } else {
        r.e[0] = p[off];
    }
    return r;
}
template <int VW>
__device__ __forceinline__ void st(float* __restrict__ p, size_t off, const Pack<VW>& r) {
    if constexpr (VW == 4) reinterpret_cast<float4*>(p)[off] = make_float4(r.e[0], r.e[1], r.e[2], r.e[3]);
    else p[off] = r.e[0];
}
template <int VW>
__device__ __forceinline__ float psum(const Pack<VW>& r) {
    if constexpr (VW == 4) return (r.e[0] + r.e[1]) + (r.e[2] + r.e[3]);
    else return r.e[0];
}
// thread's next pack of the (sample, position) walk: 256 packs further, wrapping into the next samples
__device__ __forceinline__ void advance(int& n, int& i, int HWV) {
    i += 256;
    while (i >= HWV) {
        i -= HWV;
        ++n;
    }
}
__device__ __forceinline__ Slice slice_of(int N, int S, int s) {
    const int base = N / S, rem = N % S;
    Slice r;
    r.n0 = s * base + min(s, rem);
    r.n1 = r.n0 + base + (s < rem ? 1 : 0);
    return r;
}

// ---- pass 1 (training): shifted sum and sum of squares per (channel, slice)
template <int VW>
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, float* __restrict__ partial, int N,
                                                       int C, int HWV, int S) {
    __shared__ float red[4];
    const int c = blockIdx.x, s = blockIdx.y;
    const Slice sl = slice_of(N, S, s);
    const float K = x[(size_t)c * HWV * VW];
    float s1 = 0.f, s2 = 0.f;
    for (int n = sl.n0 + (int)threadIdx.x / HWV, i = (int)threadIdx.x % HWV; n < sl.n1; advance(n, i, HWV)) {
        Pack<VW> v = ld<VW>(x, ((size_t)n * C + c) * HWV + i), q;
#pragma unroll
        for (int j = 0; j < VW; ++j) {
            v.e[j] -= K;
            q.e[j] = v.e[j] * v.e[j];
        }
        s1 += psum<VW>(v);
        s2 += psum<VW>(q);
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        partial[((size_t)c * S + s) * 2] = s1;
        partial[((size_t)c * S + s) * 2 + 1] = s2;
    }
}

// ---- pass 2 (training): normalise, affine, (+ residual), (relu), sign byte (bit j: element j of the pack was > 0)
template <bool RELU, bool RES, int VW>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                       const float* __restrict__ partial,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ run_mean, float* __restrict__ run_var,
                                                       long long* __restrict__ tracked, float* __restrict__ y,
                                                       float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                       unsigned char* __restrict__ sign, float* __restrict__ deferred,
                                                       int N, int C, int HWV, int S, float eps, float momentum) {
    __shared__ float red[4];
    const int c = blockIdx.x, s = blockIdx.y;
    const float M = (float)N * (float)(HWV * VW);
    const float K = x[(size_t)c * HWV * VW];
    const spot_bn::TrainStats ts = {partial, S, gamma, beta, run_mean, run_var, tracked, save_mean, save_rstd, deferred, eps, momentum};
    float a, b;
    spot_bn::channel_affine(ts, c, C, K, M, s == 0 && threadIdx.x == 0, red, a, b);
    const Slice sl = slice_of(N, S, s);
    for (int n = sl.n0 + (int)threadIdx.x / HWV, i = (int)threadIdx.x % HWV; n < sl.n1; advance(n, i, HWV)) {
        const size_t off = ((size_t)n * C + c) * HWV + i;
        Pack<VW> o = ld<VW>(x, off);
#pragma unroll
        for (int j = 0; j < VW; ++j) o.e[j] = fmaf(o.e[j], a, b);
        if (RES) {
            const Pack<VW> r = ld<VW>(res, off);
#pragma unroll
            for (int j = 0; j < VW; ++j) o.e[j] += r.e[j];
        }
        if (RELU) {
            unsigned m = 0;
#pragma unroll
            for (int j = 0; j < VW; ++j) {
                m |= (unsigned)(o.e[j] > 0.f) << j;
                o.e[j] = fmaxf(o.e[j], 0.f);
            }
            if (sign) sign[off] = (unsigned char)m;
        }
        st<VW>(y, off, o);
    }
}

// ---- inference: one pass with the running statistics
template <bool RELU, bool RES, int VW>
__global__ __launch_bounds__(256) void bn_infer_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ run_mean, const float* __restrict__ run_var,
                                                       float* __restrict__ y, int N, int C, int HWV, int S, float eps) {
    const int c = blockIdx.x, s = blockIdx.y;
    const float a = gamma[c] / sqrtf(run_var[c] + eps), b = beta[c] - run_mean[c] * a;
    const Slice sl = slice_of(N, S, s);
    for (int n = sl.n0 + (int)threadIdx.x / HWV, i = (int)threadIdx.x % HWV; n < sl.n1; advance(n, i, HWV)) {
        const size_t off = ((size_t)n * C + c) * HWV + i;
        Pack<VW> o = ld<VW>(x, off);
#pragma unroll
        for (int j = 0; j < VW; ++j) o.e[j] = fmaf(o.e[j], a, b);
        if (RES) {
            const Pack<VW> r = ld<VW>(res, off);
#pragma unroll
            for (int j = 0; j < VW; ++j) o.e[j] += r.e[j];
        }
        if (RELU) {
#pragma unroll
            for (int j = 0; j < VW; ++j) o.e[j] = fmaxf(o.e[j], 0.f);
        }
        st<VW>(y, off, o);
    }
}

template <int VW>
__device__ __forceinline__ Pack<VW> gated(const float* __restrict__ dy, const unsigned char* __restrict__ sign, size_t off) {
    Pack<VW> g = ld<VW>(dy, off);
    if (sign) {
        const unsigned m = sign[off];
#pragma unroll
        for (int j = 0; j < VW; ++j) g.e[j] = (m >> j & 1) ? g.e[j] : 0.f;
    }
    return g;
}

// ---- backward pass 1: sum g and sum g * xhat per (channel, slice), g = dy gated by the sign byte
template <int VW>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const unsigned char* __restrict__ sign,
                                                            const float* __restrict__ save_mean,
                                                            const float* __restrict__ save_rstd, float* __restrict__ partial,
                                                            int N, int C, int HWV, int S) {
    __shared__ float red[4];
    const int c = blockIdx.x, s = blockIdx.y;
    const float mean = save_mean[c], rstd = save_rstd[c];
    const Slice sl = slice_of(N, S, s);
    float sb = 0.f, sg = 0.f;
    for (int n = sl.n0 + (int)threadIdx.x / HWV, i = (int)threadIdx.x % HWV; n < sl.n1; advance(n, i, HWV)) {
        const size_t off = ((size_t)n * C + c) * HWV + i;
        const Pack<VW> g = gated<VW>(dy, sign, off);
        Pack<VW> v = ld<VW>(x, off);
#pragma unroll
        for (int j = 0; j < VW; ++j) v.e[j] = g.e[j] * (v.e[j] - mean);
        sb += psum<VW>(g);
        sg += psum<VW>(v);
    }
    sb = block_sum(sb, red);
    sg = block_sum(sg, red);
    if (threadIdx.x == 0) {
        partial[((size_t)c * S + s) * 2] = sb;
        partial[((size_t)c * S + s) * 2 + 1] = sg * rstd;
    }
}

// ---- backward pass 2: dx = gamma rstd (g - dbeta/M - xhat dgamma/M); d_residual = g
template <int VW>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const unsigned char* __restrict__ sign,
                                                           const float* __restrict__ save_mean,
                                                           const float* __restrict__ save_rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ partial,
                                                           float* __restrict__ dx, float* __restrict__ dres,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int N,
                                                           int C, int HWV, int S, int accumulate) {
    __shared__ float red[4];
    const int c = blockIdx.x, s = blockIdx.y;
    float db, dg;
    combine(partial, c, S, db, dg, red);
    if (s == 0 && threadIdx.x == 0) {
        // atomic when accumulating: two passes through the same layer (template and search crops) may run their
        // backward on two streams at once
        if (dgamma) {
            if (accumulate) atomicAdd(dgamma + c, dg);
            else dgamma[c] = dg;
        }
        if (dbeta) {
            if (accumulate) atomicAdd(dbeta + c, db);
            else dbeta[c] = db;
        }
    }
    const float mean = save_mean[c], rstd = save_rstd[c];
    const float M = (float)N * (float)(HWV * VW);
    const float k = gamma[c] * rstd, kb = db / M, kg = dg * rstd / M;
    const Slice sl = slice_of(N, S, s);
    for (int n = sl.n0 + (int)threadIdx.x / HWV, i = (int)threadIdx.x % HWV; n < sl.n1; advance(n, i, HWV)) {
        const size_t off = ((size_t)n * C + c) * HWV + i;
        const Pack<VW> g = gated<VW>(dy, sign, off);
        Pack<VW> v = ld<VW>(x, off);
        if (dres) st<VW>(dres, off, g);
#pragma unroll
        for (int j = 0; j < VW; ++j) v.e[j] = k * (g.e[j] - kb - (v.e[j] - mean) * kg);
        st<VW>(dx, off, v);
    }
}

bool shape_ok(int N, int C, int HW) { return N > 0 && C > 0 && HW > 0 && C <= 65535; }
// 16-byte packs need whole float4s per plane and aligned buffers; anything else is walked element by element
bool vec4(int HW, uintptr_t ptrs) { return HW % 4 == 0 && (ptrs & 15) == 0; }

}  // namespace

extern "C" long spot_batchnorm_ws_floats(int N, int C) { return (long)C * split_for(N > 0 ? N : 1, C > 0 ? C : 1) * 2; }

extern "C" int spot_batchnorm_stats_f32(const float* x, float* ws, int N, int C, int HW, hipStream_t stream) {
    SPOT_CHECK_ARG(shape_ok(N, C, HW), "batchnorm_stats: bad shape N=%d C=%d HW=%d", N, C, HW);
    SPOT_CHECK_ARG(x && ws && ((uintptr_t)x & 3) == 0, "batchnorm_stats: null or misaligned pointer");
    const int S = split_for(N, C);
    dim3 grid(C, S);
    if (vec4(HW, (uintptr_t)x)) bn_stats_kernel<4><<<grid, 256, 0, stream>>>(x, ws, N, C, HW / 4, S);
    else bn_stats_kernel<1><<<grid, 256, 0, stream>>>(x, ws, N, C, HW, S);
    SPOT_CHECK_LAUNCH("batchnorm_stats");
    return SPOT_OK;
}

extern "C" int spot_batchnorm_fwd_train_f32(const float* x, const float* residual, const float* gamma, const float* beta,
                                            float* running_mean, float* running_var, long long* num_batches_tracked,
                                            float* y, float* save_mean, float* save_rstd, unsigned char* relu_sign,
                                            float* deferred_stats, float* ws, int N, int C, int HW, float eps,
                                            float momentum, int relu, hipStream_t stream) {
    SPOT_CHECK_ARG(shape_ok(N, C, HW), "batchnorm_fwd_train: bad shape N=%d C=%d HW=%d", N, C, HW);
    SPOT_CHECK_ARG(x && gamma && beta && y && save_mean && save_rstd && ws, "batchnorm_fwd_train: null pointer");
    SPOT_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "batchnorm_fwd_train: running stats come in pairs");
    SPOT_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & 3) == 0, "batchnorm_fwd_train: 4-byte alignment");
    const bool v4 = vec4(HW, (uintptr_t)x | (uintptr_t)y | (uintptr_t)residual);
    SPOT_CHECK_ARG(relu || !relu_sign, "batchnorm_fwd_train: a sign buffer only makes sense with relu");
    const int S = split_for(N, C);
    dim3 grid(C, S);
    if (v4) bn_stats_kernel<4><<<grid, 256, 0, stream>>>(x, ws, N, C, HW / 4, S);
    else bn_stats_kernel<1><<<grid, 256, 0, stream>>>(x, ws, N, C, HW, S);
    SPOT_CHECK_LAUNCH("batchnorm_fwd_train(stats)");
#define SPOT_BN_APPLY_V(R, A, V)                                                                                        \
    bn_apply_kernel<R, A, V><<<grid, 256, 0, stream>>>(x, residual, ws, gamma, beta, running_mean, running_var,          \
                                                       num_batches_tracked, y, save_mean, save_rstd, relu_sign,         \
                                                       deferred_stats, N, C, HW / V, S, eps, momentum)
#define SPOT_BN_APPLY(R, A)                \
    do {                                   \
        if (v4) SPOT_BN_APPLY_V(R, A, 4);  \
        else SPOT_BN_APPLY_V(R, A, 1);     \
    } while (0)
    if (relu && residual) SPOT_BN_APPLY(true, true);
    else if (relu) SPOT_BN_APPLY(true, false);
    else if (residual) SPOT_BN_APPLY(false, true);
    else SPOT_BN_APPLY(false, false);
#undef SPOT_BN_APPLY
#undef SPOT_BN_APPLY_V
    SPOT_CHECK_LAUNCH("batchnorm_fwd_train(apply)");
    return SPOT_OK;
}

extern "C" int spot_batchnorm_fwd_infer_f32(const float* x, const float* residual, const float* gamma, const float* beta,
                                            const float* running_mean, const float* running_var, float* y, int N, int C,
                                            int HW, float eps, int relu, hipStream_t stream) {
    SPOT_CHECK_ARG(shape_ok(N, C, HW), "batchnorm_fwd_infer: bad shape N=%d C=%d HW=%d", N, C, HW);
    SPOT_CHECK_ARG(x && gamma && beta && running_mean && running_var && y, "batchnorm_fwd_infer: null pointer");
    SPOT_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & 3) == 0, "batchnorm_fwd_infer: 4-byte alignment");
    const bool v4 = vec4(HW, (uintptr_t)x | (uintptr_t)y | (uintptr_t)residual);
    const int S = split_for(N, C);
    dim3 grid(C, S);
#define SPOT_BN_INFER_V(R, A, V) \
    bn_infer_kernel<R, A, V><<<grid, 256, 0, stream>>>(x, residual, gamma, beta, running_mean, running_var, y, N, C, HW / V, S, eps)
#define SPOT_BN_INFER(R, A)                \
    do {                                   \
        if (v4) SPOT_BN_INFER_V(R, A, 4);  \
        else SPOT_BN_INFER_V(R, A, 1);     \
    } while (0)
    if (relu && residual) SPOT_BN_INFER(true, true);
    else if (relu) SPOT_BN_INFER(true, false);
    else if (residual) SPOT_BN_INFER(false, true);
    else SPOT_BN_INFER(false, false);
#undef SPOT_BN_INFER
#undef SPOT_BN_INFER_V
    SPOT_CHECK_LAUNCH("batchnorm_fwd_infer");
    return SPOT_OK;
}

extern "C" int spot_batchnorm_bwd_f32(const float* dy, const float* x, const unsigned char* relu_sign,
                                      const float* save_mean, const float* save_rstd, const float* gamma, float* dx,
                                      float* d_residual, float* dgamma, float* dbeta, float* ws, int N, int C, int HW,
                                      int accumulate, hipStream_t stream) {
    SPOT_CHECK_ARG(shape_ok(N, C, HW), "batchnorm_bwd: bad shape N=%d C=%d HW=%d", N, C, HW);
    SPOT_CHECK_ARG(dy && x && save_mean && save_rstd && gamma && dx && ws, "batchnorm_bwd: null pointer");
    SPOT_CHECK_ARG((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)d_residual) & 3) == 0,
                   "batchnorm_bwd: 4-byte alignment");
    const bool v4 = vec4(HW, (uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)d_residual);
    const int S = split_for(N, C);
    dim3 grid(C, S);
    if (v4) bn_bwd_reduce_kernel<4><<<grid, 256, 0, stream>>>(dy, x, relu_sign, save_mean, save_rstd, ws, N, C, HW / 4, S);
    else bn_bwd_reduce_kernel<1><<<grid, 256, 0, stream>>>(dy, x, relu_sign, save_mean, save_rstd, ws, N, C, HW, S);
    SPOT_CHECK_LAUNCH("batchnorm_bwd(reduce)");
    if (v4)
        bn_bwd_apply_kernel<4><<<grid, 256, 0, stream>>>(dy, x, relu_sign, save_mean, save_rstd, gamma, ws, dx, d_residual,
                                                         dgamma, dbeta, N, C, HW / 4, S, accumulate);
    else
        bn_bwd_apply_kernel<1><<<grid, 256, 0, stream>>>(dy, x, relu_sign, save_mean, save_rstd, gamma, ws, dx, d_residual,
                                                         dgamma, dbeta, N, C, HW, S, accumulate);
    SPOT_CHECK_LAUNCH("batchnorm_bwd(apply)");
    return SPOT_OK;
}

// Running-statistics updates that a forward pass left pending (deferred_stats of the training forward): one
// launch applies them for a whole network, in table order.  Used when two forward passes through the same
// BatchNorm layers run concurrently on two streams: the second pass in program order defers its update so the
// result equals the sequential one.
namespace {
__global__ __launch_bounds__(256) void bn_deferred_kernel(const spot_bn_deferred* __restrict__ table, int n) {
    const spot_bn_deferred e = table[blockIdx.x];
    for (int c = threadIdx.x; c < e.C; c += 256) {
        e.running_mean[c] = (1.f - e.momentum) * e.running_mean[c] + e.momentum * e.stats[c];
        e.running_var[c] = (1.f - e.momentum) * e.running_var[c] + e.momentum * e.stats[e.C + c];
    }
    if (threadIdx.x == 0 && e.num_batches_tracked) *e.num_batches_tracked += 1;
}
}  // namespace

extern "C" int spot_batchnorm_apply_deferred(const spot_bn_deferred* table_dev, int n, hipStream_t stream) {
    SPOT_CHECK_ARG(n >= 0 && (n == 0 || table_dev), "batchnorm_apply_deferred: bad arguments");
    if (n == 0) return SPOT_OK;
    bn_deferred_kernel<<<n, 256, 0, stream>>>(table_dev, n);
    SPOT_CHECK_LAUNCH("batchnorm_apply_deferred");
    return SPOT_OK;
}
