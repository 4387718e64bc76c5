// 3x3 / stride 2 / padding 1 max-pool of the ResNet stem (ltr/models/backbone/resnet.py:113,199), NCHW fp32,
// forward and backward.  HBM-bound: the forward reads the map once and writes a quarter of it plus one byte per
// output (which of the nine window positions won; the first maximum in row-major window order, and any NaN in the
// window replaces the running maximum so that it reaches the output, both as ATen's `val > max || isnan(val)`);
// the backward is a gather — an input pixel lies in 1, 2 or 4 windows and collects dy from those whose winner
// it is — so it needs no atomics and writes dx exactly once.
#include "common.h"
#include "../../include/spot_hip.h"

namespace {

// one thread: one output row position pair (2 outputs) would under-use loads; here 4 consecutive outputs of one row
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          unsigned char* __restrict__ win, long planes, int H, int W) {
    const int Ho = H >> 1, Wo = W >> 1, Wq = Wo >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= planes * Ho * Wq) return;
    const int q = (int)(idx % Wq);
    const long t = idx / Wq;
    const int oy = (int)(t % Ho);
    const long pl = t / Ho;
    const float* xp = x + pl * H * W;
    const int ox0 = 4 * q, c0 = 2 * ox0 - 1;          // input columns c0 .. c0 + 8 feed the 4 windows
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int r = 2 * oy - 1 + ky;
        if (r < 0 || r >= H) continue;
        const float* rp = xp + (long)r * W;
        float v[9];
        v[0] = c0 >= 0 ? rp[c0] : -INFINITY;
        const float4 a = *reinterpret_cast<const float4*>(rp + c0 + 1);
        const float4 b = *reinterpret_cast<const float4*>(rp + c0 + 5);
        v[1] = a.x; v[2] = a.y; v[3] = a.z; v[4] = a.w; v[5] = b.x; v[6] = b.y; v[7] = b.z; v[8] = b.w;
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float val = v[2 * o + kx];
                const bool inside = kx > 0 || o > 0 || c0 >= 0;
                if (inside && (val > m[o] || val != val)) {
                    m[o] = val;
                    w[o] = ky * 3 + kx;
                }
            }
    }
    const long off = (pl * Ho + oy) * Wo + ox0;
    *reinterpret_cast<float4*>(y + off) = make_float4(m[0], m[1], m[2], m[3]);
    if (win) *reinterpret_cast<uchar4*>(win + off) = make_uchar4(w[0], w[1], w[2], w[3]);
}

// dx[y][4j .. 4j+3]: columns 4j and 4j+2 lie in one window column (2j, 2j+1), 4j+1 and 4j+3 in two
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ win,
                                                          float* __restrict__ dx, long planes, int H, int W) {
    const int Ho = H >> 1, Wo = W >> 1, Wq = W >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= planes * H * Wq) return;
    const int j = (int)(idx % Wq);
    const long t = idx / Wq;
    const int yy = (int)(t % H);
    const long pl = t / H;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    const int oy_lo = yy >> 1, n_oy = (yy & 1) ? 2 : 1;         // odd rows lie in windows oy_lo and oy_lo + 1
    for (int a = 0; a < n_oy; ++a) {
        const int oy = oy_lo + a;
        if (oy >= Ho) break;
        const int ky = yy - (2 * oy - 1);                         // row of this pixel inside window oy
        const float* gp = dy + (pl * Ho + oy) * Wo;
        const unsigned char* wp = win + (pl * Ho + oy) * Wo;
#pragma unroll
        for (int c = 0; c < 3; ++c) {                             // window columns 2j, 2j+1, 2j+2
            const int ox = 2 * j + c;
            if (ox >= Wo) break;
            const float g = gp[ox];
            const int wpos = wp[ox];
            // window ox covers input columns 2ox-1 .. 2ox+1, i.e. local columns (of 4j..4j+3): 2c-1, 2c, 2c+1
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int lc = 2 * c - 1 + kx;
                if (lc >= 0 && lc < 4 && wpos == ky * 3 + kx) o[lc] += g;
            }
        }
    }
    *reinterpret_cast<float4*>(dx + (pl * H + yy) * W + 4 * j) = make_float4(o[0], o[1], o[2], o[3]);
}

// Any map size (the Siamese variant's 61x61 stem output): one thread per output / per input element, scalar accesses.
__global__ __launch_bounds__(256) void maxpool_fwd_any_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              unsigned char* __restrict__ win, long planes, int H, int W,
                                                              int Ho, int Wo) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= planes * Ho * Wo) return;
    const int ox = (int)(idx % Wo);
    const long t = idx / Wo;
    const int oy = (int)(t % Ho);
    const float* xp = x + (t / Ho) * H * W;
    float m = -INFINITY;
    int w = 0;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int r = 2 * oy - 1 + ky;
        if (r < 0 || r >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int c = 2 * ox - 1 + kx;
            if (c < 0 || c >= W) continue;
            const float val = xp[(long)r * W + c];
            if (val > m || val != val) {
                m = val;
                w = ky * 3 + kx;
            }
        }
    }
    y[idx] = m;
    if (win) win[idx] = (unsigned char)w;
}

__global__ __launch_bounds__(256) void maxpool_bwd_any_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ win,
                                                              float* __restrict__ dx, long planes, int H, int W, int Ho,
                                                              int Wo) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= planes * H * W) return;
    const int xx = (int)(idx % W);
    const long t = idx / W;
    const int yy = (int)(t % H);
    const long pl = t / H;
    float o = 0.f;
    // pixel (yy, xx) lies in windows oy with 2oy - 1 <= yy <= 2oy + 1, i.e. oy in {yy/2, (yy+1)/2}; likewise ox
    for (int oy = yy >> 1; oy <= (yy + 1) >> 1 && oy < Ho; ++oy)
        for (int ox = xx >> 1; ox <= (xx + 1) >> 1 && ox < Wo; ++ox) {
            const int ky = yy - (2 * oy - 1), kx = xx - (2 * ox - 1);
            if (win[(pl * Ho + oy) * Wo + ox] == ky * 3 + kx) o += dy[(pl * Ho + oy) * Wo + ox];
        }
    dx[idx] = o;
}

}  // namespace

extern "C" int spot_maxpool3x3s2_fwd_f32(const float* x, float* y, unsigned char* winner, long planes, int H, int W,
                                         hipStream_t stream) {
    SPOT_CHECK_ARG(x && y, "maxpool_fwd: null pointer");
    SPOT_CHECK_ARG(planes >= 0 && H >= 1 && W >= 1, "maxpool_fwd: bad shape planes=%ld H=%d W=%d", planes, H, W);
    if (planes == 0) return SPOT_OK;
    if (H % 2 != 0 || W % 8 != 0 || (((uintptr_t)x | (uintptr_t)y) & 15) != 0 || ((uintptr_t)winner & 3) != 0) {
        const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
        const long n = planes * Ho * Wo;
        SPOT_CHECK_ARG((n + 255) / 256 < (1L << 31), "maxpool_fwd: too large");
        maxpool_fwd_any_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(x, y, winner, planes, H, W, Ho, Wo);
        SPOT_CHECK_LAUNCH("maxpool_fwd(any size)");
        return SPOT_OK;
    }
    const long n = planes * (H / 2) * (W / 8);
    SPOT_CHECK_ARG((n + 255) / 256 < (1L << 31), "maxpool_fwd: too large");
    maxpool_fwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(x, y, winner, planes, H, W);
    SPOT_CHECK_LAUNCH("maxpool_fwd");
    return SPOT_OK;
}

extern "C" int spot_maxpool3x3s2_bwd_f32(const float* dy, const unsigned char* winner, float* dx, long planes, int H, int W,
                                         hipStream_t stream) {
    SPOT_CHECK_ARG(dy && winner && dx, "maxpool_bwd: null pointer");
    SPOT_CHECK_ARG(planes >= 0 && H >= 1 && W >= 1, "maxpool_bwd: bad shape planes=%ld H=%d W=%d", planes, H, W);
    if (planes == 0) return SPOT_OK;
    if (H % 2 != 0 || W % 8 != 0 || ((uintptr_t)dx & 15) != 0) {
        const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
        const long n = planes * H * W;
        SPOT_CHECK_ARG((n + 255) / 256 < (1L << 31), "maxpool_bwd: too large");
        maxpool_bwd_any_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(dy, winner, dx, planes, H, W, Ho, Wo);
        SPOT_CHECK_LAUNCH("maxpool_bwd(any size)");
        return SPOT_OK;
    }
    const long n = planes * H * (W / 4);
    SPOT_CHECK_ARG((n + 255) / 256 < (1L << 31), "maxpool_bwd: too large");
    maxpool_bwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(dy, winner, dx, planes, H, W);
    SPOT_CHECK_LAUNCH("maxpool_bwd");
    return SPOT_OK;
}
