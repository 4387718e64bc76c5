// Pieces of the training-mode BatchNorm forward that a kernel outside batchnorm.hip needs when it takes the
// normalisation of its INPUT on the fly (the Winograd input transform behind bn1 -> relu -> conv2 of a bottleneck,
// ltr/models/backbone/resnet.py:77-81): how many batch slices the statistics pass used, how the per-slice partial
// sums of a channel combine, and what a channel's affine map and its side effects (saved statistics, running
// statistics) are.  batchnorm.hip uses the same functions, so both paths produce the same bits.
#pragma once
#include "common.h"

namespace spot_bn {

constexpr int MAX_SPLIT = 64;

inline int split_for(int N, int C) {
    int S = 2048 / C;
    if (S < 1) S = 1;
    if (S > N) S = N;
    if (S > MAX_SPLIT) S = MAX_SPLIT;
    return S;
}

__device__ __forceinline__ float block_sum(float v, float* red) {   // red: 4 floats of LDS; all 256 threads get the sum
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// combine the S partial pairs of channel c (every thread returns the totals)
__device__ __forceinline__ void combine(const float* __restrict__ partial, int c, int S, float& a, float& b, float* red) {
    float pa = 0.f, pb = 0.f;
    if ((int)threadIdx.x < S) {
        pa = partial[((size_t)c * S + threadIdx.x) * 2];
        pb = partial[((size_t)c * S + threadIdx.x) * 2 + 1];
    }
    a = block_sum(pa, red);
    b = block_sum(pb, red);
}

// Channel c's y = a x + b from the statistics pass's partial sums (shifted by K = the channel's first element, over
// M = N * H * W elements).  `writer` (one thread per channel) also stores the saved statistics and moves the running
// ones (or leaves the update pending in `deferred`).
struct TrainStats {
    const float* partial; int S;
    const float* gamma; const float* beta;
    float* run_mean; float* run_var; long long* tracked;
    float* save_mean; float* save_rstd; float* deferred;
    float eps, momentum;
};
__device__ __forceinline__ void channel_affine(const TrainStats& t, int c, int C, float K, float M, bool writer, float* red,
                                               float& a, float& b) {
    float s1, s2;
    combine(t.partial, c, t.S, s1, s2, red);
    const float dm = s1 / M;
    const float mean = K + dm;
    const float var = fmaxf(s2 / M - dm * dm, 0.f);
    const float rstd = rsqrtf(var + t.eps);
    if (writer) {
        t.save_mean[c] = mean;
        t.save_rstd[c] = rstd;
        if (t.run_mean) {
            t.run_mean[c] = (1.f - t.momentum) * t.run_mean[c] + t.momentum * mean;
            t.run_var[c] = (1.f - t.momentum) * t.run_var[c] + t.momentum * (var * (M / (M - 1.f)));
        }
        if (t.tracked && c == 0) *t.tracked += 1;
        if (t.deferred) {
            t.deferred[c] = mean;
            t.deferred[C + c] = var * (M / (M - 1.f));
        }
    }
    a = rstd * t.gamma[c];
    b = t.beta[c] - mean * a;
}

}  // namespace spot_bn
