// Depthwise cross-correlation head (Ocean / SiamRPN++ style).
//
// Reference: ltr/models/neck/ocean_fusion.py:238-247 (xcorr_depthwise: a grouped
// conv with B*C groups) and :211-235 (GroupDW: three correlations combined with
// softmax(weight)).  Here one 256-thread workgroup owns one (clip, channel)
// plane: the three search planes are staged once into LDS with 16-byte loads,
// the template taps sit in registers, each lane slides a register window down
// one output column (R rows per item), and the softmax-weighted sum of the
// three maps is formed before the single coalesced store.  HBM traffic is the
// algorithmic minimum: every input element read once, every output written once.
#include "common.h"
#include "../../include/spot_hip.h"

// The reference evaluates these expressions as separate rounded multiplies and adds.  HIP's
// __fmul_rn/__fadd_rn are plain operators and hipcc fuses them by default, so this file is
// compiled with -ffp-contract=off (see the Makefile); explicit fmaf() calls stay fused.

namespace {

constexpr int R = 8;  // output rows per work item

// Stage n floats starting at g (4-byte aligned) into LDS so that element e
// lands at dst[shift + e], shift = misalignment of g in floats; 16-byte loads
// for the aligned interior, guarded scalar loads for head and tail.
__device__ __forceinline__ int stage_plane(const float* __restrict__ g, int n, float* dst, int tid) {
    const int shift = (int)(((uintptr_t)g >> 2) & 3);
    const float* g0 = g - shift;  // 16-byte aligned
    const int nvec = (shift + n + 3) >> 2;
    for (int k = tid; k < nvec; k += 256) {
        const int e0 = 4 * k - shift;  // plane-relative index of the vector's first float
        if (e0 >= 0 && e0 + 3 < n) {
            *reinterpret_cast<float4*>(dst + 4 * k) = *reinterpret_cast<const float4*>(g0 + 4 * k);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = e0 + j;
                if (e >= 0 && e < n) dst[4 * k + j] = g[e];
            }
        }
    }
    return shift;
}

// r[0..R) += correlation of the LDS plane xs (row stride wx) with the HZ x WZ
// template z (global, workgroup-uniform) at output rows i0.., column col.
template <int HZ, int WZ>
__device__ __forceinline__ void corr_column(const float* xs, int wx, int hx, const float* __restrict__ z, int i0,
                                            int col, float (&r)[R]) {
    float zr[HZ * WZ];
#pragma unroll
    for (int k = 0; k < HZ * WZ; ++k) zr[k] = z[k];
#pragma unroll
    for (int k = 0; k < R; ++k) r[k] = 0.f;
#pragma unroll
    for (int v = 0; v < WZ; ++v) {
        float xc[R + HZ - 1];
#pragma unroll
        for (int k = 0; k < R + HZ - 1; ++k) {
            const int row = i0 + k;
            xc[k] = row < hx ? xs[row * wx + col + v] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < HZ; ++u)
#pragma unroll
            for (int k = 0; k < R; ++k) r[k] = fmaf(xc[k + u], zr[u * WZ + v], r[k]);
    }
}

// runtime-sized template (any hz, wz): taps read through the scalar/L1 path
__device__ __forceinline__ void corr_column_rt(const float* xs, int wx, int hx, const float* __restrict__ z, int hz,
                                               int wz, int i0, int col, float (&r)[R]) {
#pragma unroll
    for (int k = 0; k < R; ++k) r[k] = 0.f;
    for (int u = 0; u < hz; ++u)
        for (int v = 0; v < wz; ++v) {
            const float zv = z[u * wz + v];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int row = i0 + k + u;
                r[k] = fmaf(row < hx ? xs[row * wx + col + v] : 0.f, zv, r[k]);
            }
        }
}

struct Branch {
    const float* x;
    const float* z;
    int hx, wx, hz, wz;
};

// KIND 0: runtime template sizes; KIND 5: Ocean's (5x5, 3x5, 5x3) triple.
template <int NB, int KIND>
__global__ __launch_bounds__(256) void groupdw_kernel(Branch b0, Branch b1, Branch b2, const float* __restrict__ weight,
                                                      float* __restrict__ out, int Ho, int Wo) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int p = blockIdx.x;
    const int tid = threadIdx.x;
    const Branch br[3] = {b0, b1, b2};
    float* xs[3];
    int off = 0;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int n = br[i].hx * br[i].wx;
        const int sh = stage_plane(br[i].x + (size_t)p * n, n, smem + off, tid);
        xs[i] = smem + off + sh;
        off += (n + 3 + 3) & ~3;
    }
    float sw[3] = {1.f, 0.f, 0.f};
    if (NB == 3) {
        const float w0 = weight[0], w1 = weight[1], w2 = weight[2];
        const float m = fmaxf(w0, fmaxf(w1, w2));
        const float e0 = expf(w0 - m), e1 = expf(w1 - m), e2 = expf(w2 - m);
        const float s = e0 + e1 + e2;
        sw[0] = e0 / s; sw[1] = e1 / s; sw[2] = e2 / s;
    }
    __syncthreads();
    const int nstrip = (Ho + R - 1) / R;
    float* op = out + (size_t)p * Ho * Wo;
    for (int item = tid; item < nstrip * Wo; item += 256) {
        const int strip = item / Wo, col = item - strip * Wo;
        const int i0 = strip * R;
        float acc[R], r[R];
#pragma unroll
        for (int k = 0; k < R; ++k) acc[k] = 0.f;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const float* z = br[i].z + (size_t)p * br[i].hz * br[i].wz;
            if (KIND == 5) {
                if (i == 0) corr_column<5, 5>(xs[0], br[0].wx, br[0].hx, z, i0, col, r);
                if (i == 1) corr_column<3, 5>(xs[1], br[1].wx, br[1].hx, z, i0, col, r);
                if (i == 2) corr_column<5, 3>(xs[2], br[2].wx, br[2].hx, z, i0, col, r);
            } else {
                corr_column_rt(xs[i], br[i].wx, br[i].hx, z, br[i].hz, br[i].wz, i0, col, r);
            }
#pragma unroll
            for (int k = 0; k < R; ++k) acc[k] = NB == 1 ? r[k] : __fadd_rn(acc[k], __fmul_rn(sw[i], r[k]));
        }
#pragma unroll
        for (int k = 0; k < R; ++k)
            if (i0 + k < Ho) op[(i0 + k) * Wo + col] = acc[k];
    }
}

size_t lds_bytes(const Branch* b, int nb) {
    size_t f = 0;
    for (int i = 0; i < nb; ++i) f += ((size_t)b[i].hx * b[i].wx + 6) & ~(size_t)3;
    return f * sizeof(float);
}

}  // namespace

extern "C" int spot_xcorr_depthwise_f32(const float* x, const float* kernel, float* out, int planes, int hx, int wx,
                                        int hz, int wz, hipStream_t stream) {
    SPOT_CHECK_ARG(x && kernel && out, "xcorr_depthwise: null pointer");
    SPOT_CHECK_ARG(planes >= 0 && hx >= hz && wx >= wz && hz > 0 && wz > 0,
                   "xcorr_depthwise: bad shape planes=%d x=%dx%d kernel=%dx%d", planes, hx, wx, hz, wz);
    if (planes == 0) return SPOT_OK;
    Branch b[3] = {{x, kernel, hx, wx, hz, wz}, {}, {}};
    const size_t lds = lds_bytes(b, 1);
    SPOT_CHECK_ARG(lds <= 64 * 1024, "xcorr_depthwise: search plane %dx%d does not fit the 64 KB LDS stage", hx, wx);
    groupdw_kernel<1, 0><<<planes, 256, lds, stream>>>(b[0], b[1], b[2], nullptr, out, hx - hz + 1, wx - wz + 1);
    SPOT_CHECK_LAUNCH("xcorr_depthwise");
    return SPOT_OK;
}

extern "C" int spot_groupdw_f32(const float* x11, const float* x12, const float* x21, const float* z11,
                                const float* z12, const float* z21, const float* weight3, float* out, int planes,
                                int hx, int wx, int hz, int wz, hipStream_t stream) {
    SPOT_CHECK_ARG(x11 && x12 && x21 && z11 && z12 && z21 && weight3 && out, "groupdw: null pointer");
    SPOT_CHECK_ARG(planes >= 0 && hz >= 3 && wz >= 3 && hx >= hz && wx >= wz,
                   "groupdw: bad shape planes=%d x11=%dx%d z11=%dx%d", planes, hx, wx, hz, wz);
    if (planes == 0) return SPOT_OK;
    // branch shapes of ocean_fusion.py:117-171: (h,w), (h-2,w), (h,w-2) for both search and template
    Branch b[3] = {{x11, z11, hx, wx, hz, wz}, {x12, z12, hx - 2, wx, hz - 2, wz}, {x21, z21, hx, wx - 2, hz, wz - 2}};
    const size_t lds = lds_bytes(b, 3);
    SPOT_CHECK_ARG(lds <= 64 * 1024, "groupdw: search planes %dx%d do not fit the 64 KB LDS stage", hx, wx);
    const int Ho = hx - hz + 1, Wo = wx - wz + 1;
    if (hz == 5 && wz == 5)
        groupdw_kernel<3, 5><<<planes, 256, lds, stream>>>(b[0], b[1], b[2], weight3, out, Ho, Wo);
    else
        groupdw_kernel<3, 0><<<planes, 256, lds, stream>>>(b[0], b[1], b[2], weight3, out, Ho, Wo);
    SPOT_CHECK_LAUNCH("groupdw");
    return SPOT_OK;
}
