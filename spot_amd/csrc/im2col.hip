// The convolutions the Winograd path does not cover — the 7x7 / stride 2 stem on 3 input channels and layer2's one
// 3x3 / stride 2 (ltr/models/backbone/resnet.py:61,79,195-196), and the Siamese head's 3x3 convolutions on odd-sized
// maps, some dilated along one axis only (ltr/models/neck/ocean_fusion.py:56-85,126-158) — as explicit im2col + the
// library's GEMM:  cols[b][(ci,ky,kx)][oy*Wo+ox] = x[b][ci][oy*s - p + ky*dh][ox*s - p + kx*dw]  (zero outside the
// map; rows beyond Cin*kh*kw up to K_pad and columns beyond Ho*Wo up to the row pitch are zero, so that the GEMM's
// tiles are whole and 16-byte aligned whatever the map size).
//   forward          y[b]   = W[Cout, K] cols[b]                       (batched GEMM)
//   weight gradient  dW     = sum_b dy[b] cols[b]^T                    (one split-K GEMM over all clips; cols kept
//                                                                       from the forward pass)
//   input gradient   dcols[b] = W^T dy[b], then a gather of dcols into dx (col2im without atomics)
// Both kernels here are HBM-bound copies with index arithmetic; lanes run along the output positions, so every
// access to cols / dcols is a coalesced row segment.
#include "common.h"
#include "../../include/spot_hip.h"

namespace {

struct ConvGeo {
    int C, H, W, kh, kw, stride, pad, dil_h, dil_w, Ho, Wo, K, Kpad, pitch;
};

__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, float* __restrict__ cols, ConvGeo g) {
    const int N4 = g.pitch >> 2, N = g.Ho * g.Wo;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;      // (kidx, n4) within clip blockIdx.y
    if (idx >= (long)g.Kpad * N4) return;
    const int n4 = (int)(idx % N4), kidx = (int)(idx / N4);
    const int b = blockIdx.y;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kidx < g.K) {
        const int kx = kidx % g.kw, t = kidx / g.kw, ky = t % g.kh, ci = t / g.kh;
        const float* xp = x + ((size_t)b * g.C + ci) * g.H * g.W;
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = 4 * n4 + j, oy = n / g.Wo, ox = n - oy * g.Wo;
            const int iy = oy * g.stride - g.pad + ky * g.dil_h, ix = ox * g.stride - g.pad + kx * g.dil_w;
            o[j] = (n < N && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) ? xp[(size_t)iy * g.W + ix] : 0.f;
        }
        v = make_float4(o[0], o[1], o[2], o[3]);
    }
    reinterpret_cast<float4*>(cols + ((size_t)b * g.Kpad + kidx) * g.pitch)[n4] = v;
}

// dx[b][ci][iy][ix] = sum over the taps (ky,kx) whose window position (oy,ox) is integral and inside the output
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcols, float* __restrict__ dx, ConvGeo g) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;      // (ci, iy, ix) within clip blockIdx.y
    if (idx >= (long)g.C * g.H * g.W) return;
    const int ix = (int)(idx % g.W);
    const long t = idx / g.W;
    const int iy = (int)(t % g.H), ci = (int)(t / g.H);
    const int b = blockIdx.y;
    const size_t N = (size_t)g.pitch;
    const float* dc = dcols + (size_t)b * g.Kpad * N;
    float s = 0.f;
    for (int ky = 0; ky < g.kh; ++ky) {
        const int ty = iy + g.pad - ky * g.dil_h;
        if (ty < 0 || ty % g.stride) continue;
        const int oy = ty / g.stride;
        if (oy >= g.Ho) continue;
        for (int kx = 0; kx < g.kw; ++kx) {
            const int tx = ix + g.pad - kx * g.dil_w;
            if (tx < 0 || tx % g.stride) continue;
            const int ox = tx / g.stride;
            if (ox >= g.Wo) continue;
            s += dc[(size_t)((ci * g.kh + ky) * g.kw + kx) * N + (size_t)oy * g.Wo + ox];
        }
    }
    dx[((size_t)b * g.C + ci) * g.H * g.W + (size_t)iy * g.W + ix] = s;
}

int geo_ok(const ConvGeo& g, int B) {
    return B >= 0 && B <= 65535 && g.C > 0 && g.H > 0 && g.W > 0 && g.kh > 0 && g.kw > 0 && g.stride > 0 && g.dil_h > 0 &&
           g.dil_w > 0 && g.pad >= 0 && g.Ho > 0 && g.Wo > 0 && g.pitch >= g.Ho * g.Wo && g.pitch % 4 == 0 && g.Kpad >= g.K;
}

ConvGeo make_geo(int C, int H, int W, int kh, int kw, int stride, int pad, int dil_h, int dil_w, int Kpad, int pitch) {
    ConvGeo g;
    g.C = C; g.H = H; g.W = W; g.kh = kh; g.kw = kw; g.stride = stride; g.pad = pad; g.dil_h = dil_h; g.dil_w = dil_w;
    g.Ho = (H + 2 * pad - dil_h * (kh - 1) - 1) / stride + 1;
    g.Wo = (W + 2 * pad - dil_w * (kw - 1) - 1) / stride + 1;
    g.K = C * kh * kw;
    g.Kpad = Kpad;
    g.pitch = pitch;
    return g;
}

}  // namespace

extern "C" int spot_im2col_f32(const float* x, float* cols, int B, int C, int H, int W, int kh, int kw, int stride, int pad,
                               int dilation_h, int dilation_w, int Kpad, int pitch, hipStream_t stream) {
    SPOT_CHECK_ARG(x && cols, "im2col: null pointer");
    const ConvGeo g = make_geo(C, H, W, kh, kw, stride, pad, dilation_h, dilation_w, Kpad, pitch);
    SPOT_CHECK_ARG(geo_ok(g, B), "im2col: bad geometry (pitch >= output positions per map and a multiple of 4, Kpad >= C*kh*kw)");
    SPOT_CHECK_ARG(((uintptr_t)cols & 15) == 0, "im2col: cols must be 16-byte aligned");
    if (B == 0) return SPOT_OK;
    const long per = (long)g.Kpad * (g.pitch / 4);
    im2col_kernel<<<dim3(spot_ceil_div(per, 256), B), 256, 0, stream>>>(x, cols, g);
    SPOT_CHECK_LAUNCH("im2col");
    return SPOT_OK;
}

extern "C" int spot_col2im_f32(const float* dcols, float* dx, int B, int C, int H, int W, int kh, int kw, int stride, int pad,
                               int dilation_h, int dilation_w, int Kpad, int pitch, hipStream_t stream) {
    SPOT_CHECK_ARG(dcols && dx, "col2im: null pointer");
    const ConvGeo g = make_geo(C, H, W, kh, kw, stride, pad, dilation_h, dilation_w, Kpad, pitch);
    SPOT_CHECK_ARG(geo_ok(g, B), "col2im: bad geometry");
    if (B == 0) return SPOT_OK;
    col2im_kernel<<<dim3(spot_ceil_div((long)C * H * W, 256), B), 256, 0, stream>>>(dcols, dx, g);
    SPOT_CHECK_LAUNCH("col2im");
    return SPOT_OK;
}
