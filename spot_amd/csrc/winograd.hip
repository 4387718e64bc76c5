// 3x3 stride-1 convolutions of the ResNet bottlenecks (ltr/models/backbone/resnet.py:79-81: conv2 of every
// block that does not stride; dilation 1, or 2 as in layer3's first block, :61-62,166) by Winograd F(4x4, 3x3)
// in fp32: forward, input gradient and weight gradient.
//
// A 4x4 output tile needs 36 multiplications per (input channel, output channel) pair instead of 144, and in
// the transformed domain those are 36 independent GEMMs  M_xi[co][p] = sum_ci U_xi[co][ci] V_xi[ci][p]
// over p = (clip, tile), which run on the matrix cores through the library's GEMM:
//   1. weights  U = G g G^T          [36][Cout][Cin]          (tiny)
//   2. input    V = B^T d B          [36][Cin][P]             HBM-bound: reads the map once, writes 2.25x its size
//   3. 36 batched GEMMs              [36][Cout][P]            fp32 MFMA, a quarter of the direct convolution's FLOPs
//   4. output   y = A^T M A          [B][Cout][H][W]          HBM-bound: reads 2.25x the map, writes it once
// The input gradient is the same computation with the filter rotated by 180 degrees and the channel roles
// swapped (padding 1 stays padding 1), selected by `flip`.
// In steps 2 and 4 a thread owns one tile of one channel; lanes run along p, so every access to V / M is a
// coalesced 256-byte row segment per wave.
#include "common.h"
#include "bn_train.h"
#include "../../include/spot_hip.h"

namespace {

constexpr int XI = 36;

// column of B^T (6x6): interpolation points 0, +-1, +-2, inf
__device__ __forceinline__ void bt6(const float d0, const float d1, const float d2, const float d3, const float d4,
                                    const float d5, float (&o)[6]) {
    o[0] = 4.f * d0 - 5.f * d2 + d4;
    o[1] = -4.f * (d1 + d2) + d3 + d4;
    o[2] = 4.f * (d1 - d2) - d3 + d4;
    o[3] = 2.f * (d3 - d1) - d2 + d4;
    o[4] = 2.f * (d1 - d3) - d2 + d4;
    o[5] = 4.f * d1 - 5.f * d3 + d5;
}

// column of A^T (4x6)
__device__ __forceinline__ void at6(const float m0, const float m1, const float m2, const float m3, const float m4,
                                    const float m5, float (&o)[4]) {
    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    o[0] = m0 + s12 + s34;
    o[1] = d12 + 2.f * d34;
    o[2] = s12 + 4.f * s34;
    o[3] = d12 + 8.f * d34 + m5;
}

// U[xi][m][k] = (G g G^T)[xi];  flip = 0: g = w[m][k];  flip = 1: g = w[k][m] rotated by 180 degrees
__global__ __launch_bounds__(256) void wino_weights_kernel(const float* __restrict__ w, float* __restrict__ U, int Mo,
                                                           int Ki, int flip) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Mo * Ki) return;
    const int m = idx / Ki, k = idx - m * Ki;
    const float* gp = flip ? w + ((size_t)k * Mo + m) * 9 : w + ((size_t)m * Ki + k) * 9;
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) g[a][b] = flip ? gp[(2 - a) * 3 + (2 - b)] : gp[a * 3 + b];
    // rows of G: (1/4,0,0) (-1/6,-1/6,-1/6) (-1/6,1/6,-1/6) (1/24,1/12,1/6) (1/24,-1/12,1/6) (0,0,1)
    float t[6][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float a = g[0][j], b = g[1][j], c = g[2][j];
        t[0][j] = a * (1.f / 4.f);
        t[1][j] = -(a + b + c) * (1.f / 6.f);
        t[2][j] = (b - a - c) * (1.f / 6.f);
        t[3][j] = a * (1.f / 24.f) + b * (1.f / 12.f) + c * (1.f / 6.f);
        t[4][j] = a * (1.f / 24.f) - b * (1.f / 12.f) + c * (1.f / 6.f);
        t[5][j] = c;
    }
    const size_t plane = (size_t)Mo * Ki;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const float a = t[i][0], b = t[i][1], c = t[i][2];
        float* up = U + (size_t)(i * 6) * plane + idx;
        up[0] = a * (1.f / 4.f);
        up[plane] = -(a + b + c) * (1.f / 6.f);
        up[2 * plane] = (b - a - c) * (1.f / 6.f);
        up[3 * plane] = a * (1.f / 24.f) + b * (1.f / 12.f) + c * (1.f / 6.f);
        up[4 * plane] = a * (1.f / 24.f) - b * (1.f / 12.f) + c * (1.f / 6.f);
        up[5 * plane] = c;
    }
}

// Where tile p of channel c lives.  With dilation D the convolution splits into D*D independent undilated ones on
// the polyphase sub-maps x[D*i + a][D*j + b] (padding D becomes padding 1 on each), so a "map" here is one
// sub-map: Hs x Ws pixels, rs / D floats apart along rows / columns.
template <int D>
struct TileGeo {
    const float* base;    // pixel (0, 0) of the tile's sub-map
    int r0, c0, Hs, Ws;   // first output row / column of the tile within the sub-map
    size_t rs;
    __device__ __forceinline__ TileGeo(const float* x, int p, int c, int C, int H, int W) {
        Hs = H / D; Ws = W / D;
        const int TW = Ws >> 2, T = (Hs >> 2) * TW;
        const int sub = p / T, t = p - sub * T, ty = t / TW, tx = t - ty * TW;
        const int b = sub / (D * D), ph = sub - b * (D * D);
        base = x + ((size_t)b * C + c) * H * W + (size_t)(ph / D) * W + (ph % D);
        rs = (size_t)D * W;
        r0 = 4 * ty; c0 = 4 * tx;
    }
};

// V[xi][c][p] = (B^T d B)[xi], d = the 6x6 input patch of tile p (rows 4ty-1 .. 4ty+4, zero outside the map)
// INBN (D = 1): the input is the raw output of the 1x1 convolution in front; its training-mode BatchNorm and ReLU
// (bn1 -> relu of a bottleneck, resnet.py:77-79) are applied to every element on its way into the patch - the
// normalised map itself is needed by nobody else and is never written.  The workgroup (one channel) derives the
// channel's affine map from the statistics pass's partial sums exactly as the stand-alone apply kernel does; the
// thread that owns a 4x4 tile writes the sign bytes of its four 4-element packs for the BatchNorm backward.
struct InBn {
    spot_bn::TrainStats t;
    unsigned char* sign;
    int N;
};
template <int D, bool INBN>
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, float* __restrict__ V, int C, int H,
                                                         int W, int P, InBn bn) {
    const int p = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    float a = 1.f, b = 0.f;
    if (INBN) {
        __shared__ float red[4];
        const float M = (float)bn.N * (float)(H * W);
        spot_bn::channel_affine(bn.t, c, C, x[(size_t)c * H * W], M, blockIdx.x == 0 && threadIdx.x == 0, red, a, b);
    }
    if (p >= P) return;
    const TileGeo<D> g(x, p, c, C, H, W);
    float tmp[6][6];      // B^T d, built column by column after loading the patch rows
    float d[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int r = g.r0 - 1 + i;
        if (r >= 0 && r < g.Hs) {
            const float* rp = g.base + (size_t)r * g.rs + (size_t)g.c0 * D;
            if (D == 1) {
                const float4 mid = *reinterpret_cast<const float4*>(rp);
                d[i][1] = mid.x; d[i][2] = mid.y; d[i][3] = mid.z; d[i][4] = mid.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) d[i][1 + j] = rp[j * D];
            }
            d[i][0] = g.c0 > 0 ? rp[-D] : 0.f;
            d[i][5] = g.c0 + 4 < g.Ws ? rp[4 * D] : 0.f;
            if (INBN) {
                unsigned m = 0;
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const float t = fmaf(d[i][j], a, b);
                    if (j >= 1 && j <= 4) m |= (unsigned)(t > 0.f) << (j - 1);
                    d[i][j] = fmaxf(t, 0.f);
                }
                if (g.c0 == 0) d[i][0] = 0.f;                    // outside the map stays zero
                if (g.c0 + 4 >= g.Ws) d[i][5] = 0.f;
                if (bn.sign && i >= 1 && i <= 4) bn.sign[(size_t)(rp - x) >> 2] = (unsigned char)m;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 6; ++j) d[i][j] = 0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float o[6];
        bt6(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], o);
#pragma unroll
        for (int i = 0; i < 6; ++i) tmp[i][j] = o[i];
    }
    const size_t plane = (size_t)C * P;
    float* vp = V + (size_t)c * P + p;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float o[6];
        bt6(tmp[i][0], tmp[i][1], tmp[i][2], tmp[i][3], tmp[i][4], tmp[i][5], o);
#pragma unroll
        for (int j = 0; j < 6; ++j) vp[(size_t)(i * 6 + j) * plane] = o[j];
    }
}

// y tile = A^T M A
// inference-mode BatchNorm (+ ReLU) of the convolution's output channel, folded into the output transform
struct OutAffine {
    const float* gamma; const float* beta; const float* mean; const float* var;
    float eps;
    int relu;
};

template <int D>
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Mb, float* __restrict__ y, int C, int H,
                                                          int W, int P, OutAffine bn) {
    const int p = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (p >= P) return;
    float sc = 1.f, sh = 0.f;
    if (bn.gamma) {                              // same arithmetic as the stand-alone inference kernel
        sc = bn.gamma[c] / sqrtf(bn.var[c] + bn.eps);
        sh = bn.beta[c] - bn.mean[c] * sc;
    }
    const TileGeo<D> g(y, p, c, C, H, W);
    const size_t plane = (size_t)C * P;
    const float* mp = Mb + (size_t)c * P + p;
    float tmp[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float o[4];
        at6(mp[(size_t)j * plane], mp[(size_t)(6 + j) * plane], mp[(size_t)(12 + j) * plane], mp[(size_t)(18 + j) * plane],
            mp[(size_t)(24 + j) * plane], mp[(size_t)(30 + j) * plane], o);
#pragma unroll
        for (int i = 0; i < 4; ++i) tmp[i][j] = o[i];
    }
    float* yp = const_cast<float*>(g.base) + (size_t)g.r0 * g.rs + (size_t)g.c0 * D;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float o[4];
        at6(tmp[i][0], tmp[i][1], tmp[i][2], tmp[i][3], tmp[i][4], tmp[i][5], o);
        if (bn.gamma) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = fmaf(o[j], sc, sh);
                if (bn.relu) o[j] = fmaxf(o[j], 0.f);
            }
        }
        if (D == 1) {
            *reinterpret_cast<float4*>(yp + (size_t)i * g.rs) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) yp[(size_t)i * g.rs + j * D] = o[j];
        }
    }
}

// column of A (6x4): the output-gradient tile enters the transformed domain through the transpose of A^T
__device__ __forceinline__ void a4(const float y0, const float y1, const float y2, const float y3, float (&o)[6]) {
    const float e = y0 + y2, f = y1 + y3, g4 = y0 + 4.f * y2, h = 2.f * y1 + 8.f * y3;
    o[0] = y0;
    o[1] = e + f;
    o[2] = e - f;
    o[3] = g4 + h;
    o[4] = g4 - h;
    o[5] = y3;
}

// dM[xi][c][p] = (A dY A^T)[xi], dY = the 4x4 output-gradient tile p of channel c
template <int D>
__global__ __launch_bounds__(256) void wino_outgrad_kernel(const float* __restrict__ dy, float* __restrict__ dM, int C,
                                                           int H, int W, int P) {
    const int p = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (p >= P) return;
    const TileGeo<D> g(dy, p, c, C, H, W);
    const float* yp = g.base + (size_t)g.r0 * g.rs + (size_t)g.c0 * D;
    float4 r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (D == 1) r[i] = *reinterpret_cast<const float4*>(yp + (size_t)i * g.rs);
        else r[i] = make_float4(yp[(size_t)i * g.rs], yp[(size_t)i * g.rs + D], yp[(size_t)i * g.rs + 2 * D],
                                yp[(size_t)i * g.rs + 3 * D]);
    }
    float tmp[6][4];
    {
        float o[6];
        a4(r[0].x, r[1].x, r[2].x, r[3].x, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) tmp[i][0] = o[i];
        a4(r[0].y, r[1].y, r[2].y, r[3].y, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) tmp[i][1] = o[i];
        a4(r[0].z, r[1].z, r[2].z, r[3].z, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) tmp[i][2] = o[i];
        a4(r[0].w, r[1].w, r[2].w, r[3].w, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) tmp[i][3] = o[i];
    }
    const size_t plane = (size_t)C * P;
    float* mp = dM + (size_t)c * P + p;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float o[6];
        a4(tmp[i][0], tmp[i][1], tmp[i][2], tmp[i][3], o);
#pragma unroll
        for (int j = 0; j < 6; ++j) mp[(size_t)(i * 6 + j) * plane] = o[j];
    }
}

// row of G^T (3x6)
__device__ __forceinline__ void gt6(const float u0, const float u1, const float u2, const float u3, const float u4,
                                    const float u5, float (&o)[3]) {
    const float s12 = u1 + u2, s34 = u3 + u4;
    o[0] = u0 * (1.f / 4.f) - s12 * (1.f / 6.f) + s34 * (1.f / 24.f);
    o[1] = (u2 - u1) * (1.f / 6.f) + (u3 - u4) * (1.f / 12.f);
    o[2] = (s34 - s12) * (1.f / 6.f) + u5;
}

// dw[m][k][3][3] (+)= G^T dU[.][m][k] G
__global__ __launch_bounds__(256) void wino_wgrad_out_kernel(const float* __restrict__ dU, float* __restrict__ dw, int Mo,
                                                             int Ki, int accumulate) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Mo * Ki) return;
    const size_t plane = (size_t)Mo * Ki;
    const float* up = dU + idx;
    float tmp[3][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float o[3];
        gt6(up[(size_t)j * plane], up[(size_t)(6 + j) * plane], up[(size_t)(12 + j) * plane], up[(size_t)(18 + j) * plane],
            up[(size_t)(24 + j) * plane], up[(size_t)(30 + j) * plane], o);
#pragma unroll
        for (int a = 0; a < 3; ++a) tmp[a][j] = o[a];
    }
    float* wp = dw + (size_t)idx * 9;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float o[3];
        gt6(tmp[a][0], tmp[a][1], tmp[a][2], tmp[a][3], tmp[a][4], tmp[a][5], o);
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            // atomic: the template and the search pass may deliver a shared weight's gradient from two streams at once
            if (accumulate) atomicAdd(wp + a * 3 + b, o[b]);
            else wp[a * 3 + b] = o[b];
        }
    }
}

// ---- the same three transforms for any map size and padding (undilated): tiles cover the Ho x Wo output grid,
// TH x TW of them per clip, partial ones masked; the input patch of tile (ty, tx) starts at (4 ty - pad, 4 tx - pad).
// Used for the Siamese head's 25x25 ... 45x45 maps with padding 0 or 1 (ltr/models/neck/ocean_fusion.py:56-85,
// 126-135) and for the input gradient of such a convolution (padding 2 - pad on the output gradient).  The tile
// count is rounded up to a multiple of 32 (Pp) with zero columns so the products keep whole aligned tiles.
// Dilation (dh, dw), each 1 or 2, splits the map into dh*dw polyphase sub-maps (rows = a mod dh, columns = b mod dw)
// that are convolved independently without dilation; `pad` counts sub-map pixels (the convolution's padding divided
// by its dilation).  Sub-maps of an odd-sized map differ by one row / column; all use the tile grid of the largest.
struct GenGeo {
    int Hin, Win, Ho, Wo, pad, TH, TW, dh, dw;
};

struct GenTile {           // tile p: its clip, the sub-map's origin and extent in input and output, its tile position
    int b, ty, tx, hs_in, ws_in, hs_out, ws_out;
    size_t in_off, out_off;
    __device__ __forceinline__ GenTile(const GenGeo& g, int p) {
        const int T = g.TH * g.TW, S = g.dh * g.dw;
        b = p / (S * T);
        const int rem = p - b * (S * T), ph = rem / T, t = rem - ph * T;
        const int a = ph / g.dw, bb = ph - a * g.dw;
        ty = t / g.TW; tx = t - ty * g.TW;
        hs_in = (g.Hin - a + g.dh - 1) / g.dh; ws_in = (g.Win - bb + g.dw - 1) / g.dw;
        hs_out = (g.Ho - a + g.dh - 1) / g.dh; ws_out = (g.Wo - bb + g.dw - 1) / g.dw;
        in_off = (size_t)a * g.Win + bb; out_off = (size_t)a * g.Wo + bb;
    }
};

__global__ __launch_bounds__(256) void wino_input_gen_kernel(const float* __restrict__ x, float* __restrict__ V, int C,
                                                             GenGeo g, int P, int Pp) {
    const int p = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (p >= Pp) return;
    const size_t plane = (size_t)C * Pp;
    float* vp = V + (size_t)c * Pp + p;
    if (p >= P) {
#pragma unroll
        for (int k = 0; k < XI; ++k) vp[(size_t)k * plane] = 0.f;
        return;
    }
    const GenTile q(g, p);
    const float* base = x + ((size_t)q.b * C + c) * g.Hin * g.Win + q.in_off;
    const int r0 = 4 * q.ty - g.pad, c0 = 4 * q.tx - g.pad;
    const size_t rs = (size_t)g.dh * g.Win;
    float d[6][6], tmp[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int r = r0 + i;
        const bool rin = r >= 0 && r < q.hs_in;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int cc = c0 + j;
            d[i][j] = (rin && cc >= 0 && cc < q.ws_in) ? base[(size_t)r * rs + (size_t)cc * g.dw] : 0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float o[6];
        bt6(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], o);
#pragma unroll
        for (int i = 0; i < 6; ++i) tmp[i][j] = o[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float o[6];
        bt6(tmp[i][0], tmp[i][1], tmp[i][2], tmp[i][3], tmp[i][4], tmp[i][5], o);
#pragma unroll
        for (int j = 0; j < 6; ++j) vp[(size_t)(i * 6 + j) * plane] = o[j];
    }
}

__global__ __launch_bounds__(256) void wino_output_gen_kernel(const float* __restrict__ Mb, float* __restrict__ y,
                                                              const float* __restrict__ bias, int C, GenGeo g, int P, int Pp,
                                                              OutAffine bn) {
    const int p = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (p >= P) return;
    float sc = 1.f, sh = 0.f;
    if (bn.gamma) {                              // same arithmetic as the stand-alone inference kernel
        sc = bn.gamma[c] / sqrtf(bn.var[c] + bn.eps);
        sh = bn.beta[c] - bn.mean[c] * sc;
    }
    const GenTile q(g, p);
    const int ty = q.ty, tx = q.tx;
    const size_t plane = (size_t)C * Pp;
    const float* mp = Mb + (size_t)c * Pp + p;
    const float add = bias ? bias[c] : 0.f;
    float tmp[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float o[4];
        at6(mp[(size_t)j * plane], mp[(size_t)(6 + j) * plane], mp[(size_t)(12 + j) * plane], mp[(size_t)(18 + j) * plane],
            mp[(size_t)(24 + j) * plane], mp[(size_t)(30 + j) * plane], o);
#pragma unroll
        for (int i = 0; i < 4; ++i) tmp[i][j] = o[i];
    }
    float* yp = y + ((size_t)q.b * C + c) * g.Ho * g.Wo + q.out_off;
    const size_t rs = (size_t)g.dh * g.Wo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 4 * ty + i;
        if (r >= q.hs_out) break;
        float o[4];
        at6(tmp[i][0], tmp[i][1], tmp[i][2], tmp[i][3], tmp[i][4], tmp[i][5], o);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cc = 4 * tx + j;
            if (cc < q.ws_out) {
                float v = o[j] + add;
                if (bn.gamma) {
                    v = fmaf(v, sc, sh);
                    if (bn.relu) v = fmaxf(v, 0.f);
                }
                yp[(size_t)r * rs + (size_t)cc * g.dw] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void wino_outgrad_gen_kernel(const float* __restrict__ dy, float* __restrict__ dM, int C,
                                                               GenGeo g, int P, int Pp) {
    const int p = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (p >= Pp) return;
    const size_t plane = (size_t)C * Pp;
    float* mp = dM + (size_t)c * Pp + p;
    if (p >= P) {
#pragma unroll
        for (int k = 0; k < XI; ++k) mp[(size_t)k * plane] = 0.f;
        return;
    }
    const GenTile q(g, p);
    const float* yp = dy + ((size_t)q.b * C + c) * g.Ho * g.Wo + q.out_off;
    const size_t rs = (size_t)g.dh * g.Wo;
    float r[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rr = 4 * q.ty + i, cc = 4 * q.tx + j;
            r[i][j] = (rr < q.hs_out && cc < q.ws_out) ? yp[(size_t)rr * rs + (size_t)cc * g.dw] : 0.f;
        }
    float tmp[6][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float o[6];
        a4(r[0][j], r[1][j], r[2][j], r[3][j], o);
#pragma unroll
        for (int i = 0; i < 6; ++i) tmp[i][j] = o[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float o[6];
        a4(tmp[i][0], tmp[i][1], tmp[i][2], tmp[i][3], o);
#pragma unroll
        for (int j = 0; j < 6; ++j) mp[(size_t)(i * 6 + j) * plane] = o[j];
    }
}

struct GenPlan {
    GenGeo g;
    int P, Pp;
    long U, V, M;
    bool ok;
};
GenPlan gen_plan(int B, int Cin, int Cout, int Hin, int Win, int pad, int dh, int dw) {
    GenPlan q = {};
    q.g.Hin = Hin; q.g.Win = Win; q.g.pad = pad; q.g.dh = dh; q.g.dw = dw;
    q.g.Ho = Hin + (2 * pad - 2) * dh; q.g.Wo = Win + (2 * pad - 2) * dw;
    q.ok = B > 0 && Cin > 0 && Cout > 0 && Hin > 0 && Win > 0 && pad >= 0 && pad <= 2 && (dh == 1 || dh == 2) &&
           (dw == 1 || dw == 2) && q.g.Ho > 0 && q.g.Wo > 0 && Cin <= 65535 && Cout <= 65535;
    if (!q.ok) return q;
    q.g.TH = ((q.g.Ho + dh - 1) / dh + 3) / 4; q.g.TW = ((q.g.Wo + dw - 1) / dw + 3) / 4;
    const long P = (long)B * dh * dw * q.g.TH * q.g.TW, Pp = (P + 31) / 32 * 32;
    q.ok = Pp < (1L << 31) && Pp * (long)(Cin > Cout ? Cin : Cout) * XI < (1L << 40);
    q.P = (int)P; q.Pp = (int)Pp;
    q.U = (long)XI * Cout * Cin; q.V = (long)XI * Cin * Pp; q.M = (long)XI * Cout * Pp;
    return q;
}

struct Sizes {
    long U, V, M;
};
Sizes sizes(int B, int Cin, int Cout, int H, int W) {
    const long P = (long)B * (H / 4) * (W / 4);
    return {(long)XI * Cout * Cin, (long)XI * Cin * P, (long)XI * Cout * P};
}

}  // namespace

extern "C" long spot_conv3x3_winograd_ws_floats(int B, int Cin, int Cout, int H, int W) {
    const Sizes s = sizes(B, Cin, Cout, H, W);
    return s.U + s.V + s.M;
}

#define WINO_LAUNCH(kernel, dil, grid, ...)                                   \
    do {                                                                      \
        if ((dil) == 1) kernel<1><<<grid, 256, 0, stream>>>(__VA_ARGS__);     \
        else kernel<2><<<grid, 256, 0, stream>>>(__VA_ARGS__);                \
    } while (0)

#define WINO_LAUNCH_IN(dil, grid, ...)                                                        \
    do {                                                                                      \
        if ((dil) == 1) wino_input_kernel<1, false><<<grid, 256, 0, stream>>>(__VA_ARGS__, InBn{}); \
        else wino_input_kernel<2, false><<<grid, 256, 0, stream>>>(__VA_ARGS__, InBn{});           \
    } while (0)

extern "C" int spot_conv3x3_winograd_f32(const float* in, const float* w, float* out, float* ws,
                                         const spot_winograd_opts* opts, int B, int Cin, int Cout, int H, int W,
                                         int dilation, int flip, hipStream_t stream) {
    float* v_keep = opts ? opts->v_keep : nullptr;
    float* u_keep = opts ? opts->u_keep : nullptr;
    OutAffine bn = {nullptr, nullptr, nullptr, nullptr, 0.f, 0};
    if (opts && opts->bn_gamma) {
        SPOT_CHECK_ARG(opts->bn_beta && opts->bn_mean && opts->bn_var, "conv3x3_winograd: bn_gamma needs bn_beta, bn_mean, bn_var");
        bn = {opts->bn_gamma, opts->bn_beta, opts->bn_mean, opts->bn_var, opts->bn_eps, opts->relu};
    }
    SPOT_CHECK_ARG(in && w && out && ws, "conv3x3_winograd: null pointer");
    SPOT_CHECK_ARG(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "conv3x3_winograd: bad shape");
    SPOT_CHECK_ARG(dilation == 1 || dilation == 2, "conv3x3_winograd: dilation=%d (1 or 2)", dilation);
    SPOT_CHECK_ARG(H % (4 * dilation) == 0 && W % (4 * dilation) == 0,
                   "conv3x3_winograd: H=%d, W=%d must be multiples of %d", H, W, 4 * dilation);
    SPOT_CHECK_ARG((((uintptr_t)in | (uintptr_t)out | (uintptr_t)ws) & 15) == 0, "conv3x3_winograd: buffers must be 16-byte aligned");
    const long Pl = (long)B * (H / 4) * (W / 4);
    SPOT_CHECK_ARG(Pl * (long)(Cin > Cout ? Cin : Cout) * XI < (1L << 40) && Pl < (1L << 31), "conv3x3_winograd: problem too large");
    const int P = (int)Pl;
    const Sizes s = sizes(B, Cin, Cout, H, W);
    SPOT_CHECK_ARG((((uintptr_t)v_keep | (uintptr_t)u_keep) & 15) == 0, "conv3x3_winograd: v_keep / u_keep must be 16-byte aligned");
    float* U = u_keep ? u_keep : ws;
    float* V = v_keep ? v_keep : ws + s.U;
    float* Mb = ws + s.U + s.V;
    if (!(u_keep && opts->u_valid)) {
        wino_weights_kernel<<<spot_ceil_div((long)Cout * Cin, 256), 256, 0, stream>>>(w, U, Cout, Cin, flip);
        SPOT_CHECK_LAUNCH("conv3x3_winograd(weights)");
    }
    if (opts && opts->in_bn) {
        const spot_bn_train_in* ib = opts->in_bn;
        SPOT_CHECK_ARG(dilation == 1 && !flip, "conv3x3_winograd: in_bn needs dilation 1 and the forward direction");
        SPOT_CHECK_ARG(ib->stats_ws && ib->gamma && ib->beta && ib->save_mean && ib->save_rstd, "conv3x3_winograd: in_bn null pointer");
        SPOT_CHECK_ARG((ib->running_mean == nullptr) == (ib->running_var == nullptr), "conv3x3_winograd: in_bn running stats come in pairs");
        InBn bn_in = {{ib->stats_ws, spot_bn::split_for(B, Cin), ib->gamma, ib->beta, ib->running_mean, ib->running_var,
                       ib->num_batches_tracked, ib->save_mean, ib->save_rstd, ib->deferred_stats, ib->eps, ib->momentum},
                      ib->relu_sign, B};
        wino_input_kernel<1, true><<<dim3(spot_ceil_div(P, 256), Cin), 256, 0, stream>>>(in, V, Cin, H, W, P, bn_in);
    } else {
        WINO_LAUNCH_IN(dilation, dim3(spot_ceil_div(P, 256), Cin), in, V, Cin, H, W, P);
    }
    SPOT_CHECK_LAUNCH("conv3x3_winograd(input)");
    spot_gemm_desc d = {};
    d.A = U; d.B = V; d.C = Mb;
    d.M = Cout; d.N = P; d.K = Cin; d.batch = XI;
    d.transA = 0; d.transB = 1;
    d.lda = Cin; d.ldb = P; d.ldc = P;
    d.strideA = (long)Cout * Cin; d.strideB = (long)Cin * P; d.strideC = (long)Cout * P;
    d.splitk = 1; d.alpha = 1.f;
    const int rc = spot_gemm_f32(&d, stream);
    if (rc != SPOT_OK) return rc;
    WINO_LAUNCH(wino_output_kernel, dilation, dim3(spot_ceil_div(P, 256), Cout), Mb, out, Cout, H, W, P, bn);
    SPOT_CHECK_LAUNCH("conv3x3_winograd(output)");
    return SPOT_OK;
}

extern "C" long spot_conv3x3_winograd_wgrad_ws_floats(int B, int Cin, int Cout, int H, int W) {
    return spot_conv3x3_winograd_ws_floats(B, Cin, Cout, H, W);
}

extern "C" int spot_conv3x3_winograd_wgrad_f32(const float* x, const float* v_kept, const float* dy, float* dw, float* ws,
                                               int B, int Cin, int Cout, int H, int W, int dilation, int accumulate,
                                               hipStream_t stream) {
    SPOT_CHECK_ARG((x || v_kept) && dy && dw && ws, "conv3x3_winograd_wgrad: null pointer");
    SPOT_CHECK_ARG(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "conv3x3_winograd_wgrad: bad shape");
    SPOT_CHECK_ARG(dilation == 1 || dilation == 2, "conv3x3_winograd_wgrad: dilation=%d (1 or 2)", dilation);
    SPOT_CHECK_ARG(H % (4 * dilation) == 0 && W % (4 * dilation) == 0,
                   "conv3x3_winograd_wgrad: H=%d, W=%d must be multiples of %d", H, W, 4 * dilation);
    SPOT_CHECK_ARG((((uintptr_t)x | (uintptr_t)v_kept | (uintptr_t)dy | (uintptr_t)ws) & 15) == 0,
                   "conv3x3_winograd_wgrad: buffers must be 16-byte aligned");
    const long Pl = (long)B * (H / 4) * (W / 4);
    SPOT_CHECK_ARG(Pl * (long)(Cin > Cout ? Cin : Cout) * XI < (1L << 40) && Pl < (1L << 31), "conv3x3_winograd_wgrad: problem too large");
    const int P = (int)Pl;
    const Sizes s = sizes(B, Cin, Cout, H, W);
    float* dU = ws;
    float* Vown = dU + s.U;
    float* dM = Vown + s.V;
    const float* V = v_kept;
    if (!V) {
        WINO_LAUNCH_IN(dilation, dim3(spot_ceil_div(P, 256), Cin), x, Vown, Cin, H, W, P);
        SPOT_CHECK_LAUNCH("conv3x3_winograd_wgrad(input)");
        V = Vown;
    }
    WINO_LAUNCH(wino_outgrad_kernel, dilation, dim3(spot_ceil_div(P, 256), Cout), dy, dM, Cout, H, W, P);
    SPOT_CHECK_LAUNCH("conv3x3_winograd_wgrad(output gradient)");
    // dU[xi][co][ci] = sum_p dM[xi][co][p] V[xi][ci][p]: the tile index is the reduction; split it when the 36 output
    // matrices alone give too few workgroups
    const long wgs = (long)((Cout + 63) / 64) * ((Cin + 63) / 64) * XI;
    int sk = (int)((1536 + wgs - 1) / wgs);
    if (sk > P / 256) sk = P / 256;
    if (sk < 1) sk = 1;
    if (sk > 1 && hipMemsetAsync(dU, 0, (size_t)s.U * sizeof(float), stream) != hipSuccess) {
        spot_set_error("conv3x3_winograd_wgrad: clearing the workspace failed");
        return SPOT_ERR_LAUNCH;
    }
    spot_gemm_desc d = {};
    d.A = dM; d.B = V; d.C = dU;
    d.M = Cout; d.N = Cin; d.K = P; d.batch = XI;
    d.transA = 0; d.transB = 0;
    d.lda = P; d.ldb = P; d.ldc = Cin;
    d.strideA = (long)Cout * P; d.strideB = (long)Cin * P; d.strideC = (long)Cout * Cin;
    d.splitk = sk; d.accumulate = sk > 1; d.alpha = 1.f;
    const int rc = spot_gemm_f32(&d, stream);
    if (rc != SPOT_OK) return rc;
    wino_wgrad_out_kernel<<<spot_ceil_div((long)Cout * Cin, 256), 256, 0, stream>>>(dU, dw, Cout, Cin, accumulate);
    SPOT_CHECK_LAUNCH("conv3x3_winograd_wgrad(weights)");
    return SPOT_OK;
}


// ---- any map size, sub-map padding 0 / 1 / 2, dilation 1 or 2 per axis (see GenGeo) -----------------------------------------------------
extern "C" long spot_conv3x3_winograd_general_ws_floats(int B, int Cin, int Cout, int Hin, int Win, int pad, int dil_h,
                                                        int dil_w) {
    const GenPlan q = gen_plan(B, Cin, Cout, Hin, Win, pad, dil_h, dil_w);
    return q.ok ? q.U + q.V + q.M : 0;
}

extern "C" long spot_conv3x3_winograd_general_v_floats(int B, int Cin, int Hin, int Win, int pad, int dil_h, int dil_w) {
    const GenPlan q = gen_plan(B, Cin, 1, Hin, Win, pad, dil_h, dil_w);
    return q.ok ? q.V : 0;
}

extern "C" int spot_conv3x3_winograd_general_f32(const float* in, const float* w, const float* bias, float* out, float* ws,
                                                 const spot_winograd_opts* opts, int B, int Cin, int Cout, int Hin,
                                                 int Win, int pad, int dil_h, int dil_w, int flip, hipStream_t stream) {
    float* v_keep = opts ? opts->v_keep : nullptr;
    float* u_keep = opts ? opts->u_keep : nullptr;
    SPOT_CHECK_ARG(!(opts && opts->in_bn), "conv3x3_winograd_general: in_bn is taken by spot_conv3x3_winograd_f32 only");
    OutAffine bn = {nullptr, nullptr, nullptr, nullptr, 0.f, 0};
    if (opts && opts->bn_gamma) {
        SPOT_CHECK_ARG(opts->bn_beta && opts->bn_mean && opts->bn_var, "conv3x3_winograd_general: bn_gamma needs bn_beta, bn_mean, bn_var");
        bn = {opts->bn_gamma, opts->bn_beta, opts->bn_mean, opts->bn_var, opts->bn_eps, opts->relu};
    }
    SPOT_CHECK_ARG(in && w && out && ws, "conv3x3_winograd_general: null pointer");
    const GenPlan q = gen_plan(B, Cin, Cout, Hin, Win, pad, dil_h, dil_w);
    SPOT_CHECK_ARG(q.ok, "conv3x3_winograd_general: bad shape B=%d Cin=%d Cout=%d H=%d W=%d pad=%d", B, Cin, Cout, Hin, Win, pad);
    SPOT_CHECK_ARG((((uintptr_t)ws | (uintptr_t)v_keep | (uintptr_t)u_keep) & 15) == 0,
                   "conv3x3_winograd_general: ws / v_keep / u_keep must be 16-byte aligned");
    float* U = u_keep ? u_keep : ws;
    float* V = v_keep ? v_keep : ws + q.U;
    float* Mb = ws + q.U + q.V;
    if (!(u_keep && opts->u_valid)) {
        wino_weights_kernel<<<spot_ceil_div((long)Cout * Cin, 256), 256, 0, stream>>>(w, U, Cout, Cin, flip);
        SPOT_CHECK_LAUNCH("conv3x3_winograd_general(weights)");
    }
    wino_input_gen_kernel<<<dim3(spot_ceil_div(q.Pp, 256), Cin), 256, 0, stream>>>(in, V, Cin, q.g, q.P, q.Pp);
    SPOT_CHECK_LAUNCH("conv3x3_winograd_general(input)");
    spot_gemm_desc d = {};
    d.A = U; d.B = V; d.C = Mb;
    d.M = Cout; d.N = q.Pp; d.K = Cin; d.batch = XI;
    d.transA = 0; d.transB = 1;
    d.lda = Cin; d.ldb = q.Pp; d.ldc = q.Pp;
    d.strideA = (long)Cout * Cin; d.strideB = (long)Cin * q.Pp; d.strideC = (long)Cout * q.Pp;
    d.splitk = 1; d.alpha = 1.f;
    const int rc = spot_gemm_f32(&d, stream);
    if (rc != SPOT_OK) return rc;
    wino_output_gen_kernel<<<dim3(spot_ceil_div(q.P, 256), Cout), 256, 0, stream>>>(Mb, out, bias, Cout, q.g, q.P, q.Pp, bn);
    SPOT_CHECK_LAUNCH("conv3x3_winograd_general(output)");
    return SPOT_OK;
}

extern "C" int spot_conv3x3_winograd_general_wgrad_f32(const float* x, const float* v_kept, const float* dy, float* dw,
                                                       float* ws, int B, int Cin, int Cout, int Hin, int Win, int pad,
                                                       int dil_h, int dil_w, int accumulate, hipStream_t stream) {
    SPOT_CHECK_ARG((x || v_kept) && dy && dw && ws, "conv3x3_winograd_general_wgrad: null pointer");
    const GenPlan q = gen_plan(B, Cin, Cout, Hin, Win, pad, dil_h, dil_w);
    SPOT_CHECK_ARG(q.ok, "conv3x3_winograd_general_wgrad: bad shape");
    SPOT_CHECK_ARG((((uintptr_t)ws | (uintptr_t)v_kept) & 15) == 0, "conv3x3_winograd_general_wgrad: ws / v_kept must be 16-byte aligned");
    float* dU = ws;
    float* Vown = dU + q.U;
    float* dM = Vown + q.V;
    const float* V = v_kept;
    if (!V) {
        wino_input_gen_kernel<<<dim3(spot_ceil_div(q.Pp, 256), Cin), 256, 0, stream>>>(x, Vown, Cin, q.g, q.P, q.Pp);
        SPOT_CHECK_LAUNCH("conv3x3_winograd_general_wgrad(input)");
        V = Vown;
    }
    wino_outgrad_gen_kernel<<<dim3(spot_ceil_div(q.Pp, 256), Cout), 256, 0, stream>>>(dy, dM, Cout, q.g, q.P, q.Pp);
    SPOT_CHECK_LAUNCH("conv3x3_winograd_general_wgrad(output gradient)");
    const long wgs = (long)((Cout + 63) / 64) * ((Cin + 63) / 64) * XI;
    int sk = (int)((1536 + wgs - 1) / wgs);
    if (sk > q.Pp / 256) sk = q.Pp / 256;
    if (sk < 1) sk = 1;
    if (sk > 1) SPOT_CHECK_HIP(hipMemsetAsync(dU, 0, (size_t)q.U * sizeof(float), stream));
    spot_gemm_desc d = {};
    d.A = dM; d.B = V; d.C = dU;
    d.M = Cout; d.N = Cin; d.K = q.Pp; d.batch = XI;
    d.transA = 0; d.transB = 0;
    d.lda = q.Pp; d.ldb = q.Pp; d.ldc = Cin;
    d.strideA = (long)Cout * q.Pp; d.strideB = (long)Cin * q.Pp; d.strideC = (long)Cout * Cin;
    d.splitk = sk; d.accumulate = sk > 1; d.alpha = 1.f;
    const int rc = spot_gemm_f32(&d, stream);
    if (rc != SPOT_OK) return rc;
    wino_wgrad_out_kernel<<<spot_ceil_div((long)Cout * Cin, 256), 256, 0, stream>>>(dU, dw, Cout, Cin, accumulate);
    SPOT_CHECK_LAUNCH("conv3x3_winograd_general_wgrad(weights)");
    return SPOT_OK;
}
