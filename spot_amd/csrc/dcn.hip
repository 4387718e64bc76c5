// Deformable convolution v1 (the aligned classification head of the Siamese variant, ltr/models/neck/ocean_fusion.py:
// 174-208; reference kernels ltr/models/neck/dcn/src/deform_conv_cuda_kernel.cu:82-113,189-241,277-333,371-434).
// The 3x3 taps of output position (oy, ox) are read at (oy*s - p + ky*d + off_y, ox*s - p + kx*d + off_x) by bilinear
// interpolation, off_y / off_x coming from offset[b][g*2*kh*kw + 2*(ky*kw + kx) (+1)][oy][ox] (g = deformable group of
// the input channel); samples at or beyond one pixel outside the map are zero, neighbours outside the map count as
// zero.  The sampled patch matrix has the layout of spot_im2col_f32, so forward, input-gradient and weight-gradient
// products are the same GEMMs as for an ordinary convolution:
//   forward   cols[b][(ci,ky,kx)][n] = bilinear(x[b][ci]; y, x);            y[b] = W cols[b]
//   backward  dcols[b] = W^T dy[b];  dx += bilinear weights * dcols (scatter, float atomics);
//             doffset[b][2 tap (+1)][n] = sum over ci of dcols * d bilinear / d y (d x)
// Work is laid out with the lanes along the output positions n: offsets, patch-matrix rows and dcols rows are then
// contiguous segments, and the interpolation set-up of a (tap, n) pair is shared by all input channels of the loop.
#include "common.h"
#include "../../include/spot_hip.h"

namespace {

struct DcnGeo {
    int C, H, W, kh, kw, stride, pad, dil, Ho, Wo, K, Kpad, pitch, dgroups;
};

struct Tap {          // one bilinear sample: the four neighbours' offsets in the plane (or -1) and weights
    int i00, i01, i10, i11;
    float w00, w01, w10, w11;
    float lh, lw;
    bool valid;
};

__device__ __forceinline__ Tap make_tap(float h, float w, int H, int W) {
    Tap t;
    t.valid = h > -1.f && w > -1.f && h < (float)H && w < (float)W;
    const int hl = (int)floorf(h), wl = (int)floorf(w);
    const int hh = hl + 1, wh = wl + 1;
    t.lh = h - (float)hl;
    t.lw = w - (float)wl;
    const float uh = 1.f - t.lh, uw = 1.f - t.lw;
    const bool top = t.valid && hl >= 0, bot = t.valid && hh <= H - 1, lef = wl >= 0, rig = wh <= W - 1;
    t.i00 = top && lef ? hl * W + wl : -1;
    t.i01 = top && rig ? hl * W + wh : -1;
    t.i10 = bot && lef ? hh * W + wl : -1;
    t.i11 = bot && rig ? hh * W + wh : -1;
    t.w00 = uh * uw; t.w01 = uh * t.lw; t.w10 = t.lh * uw; t.w11 = t.lh * t.lw;
    return t;
}

__device__ __forceinline__ Tap tap_at(const float* __restrict__ offset, const DcnGeo& g, int b, int dg, int tap, int n) {
    const int N = g.Ho * g.Wo;
    const int oy = n / g.Wo, ox = n - oy * g.Wo, ky = tap / g.kw, kx = tap - ky * g.kw;
    const float* op = offset + ((size_t)(b * g.dgroups + dg) * 2 * g.kh * g.kw + 2 * tap) * N + n;
    const float h = (float)(oy * g.stride - g.pad + ky * g.dil) + op[0];
    const float w = (float)(ox * g.stride - g.pad + kx * g.dil) + op[N];
    return make_tap(h, w, g.H, g.W);
}

constexpr int CI_CHUNK = 32;     // input channels per workgroup

// grid: (ceil(pitch / 256) * ceil(C / CI_CHUNK), kh*kw, B)
__global__ __launch_bounds__(256) void deform_im2col_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                                                            float* __restrict__ cols, DcnGeo g) {
    const int nblocks = (g.pitch + 255) / 256;
    const int n = (blockIdx.x % nblocks) * 256 + threadIdx.x, c0 = (blockIdx.x / nblocks) * CI_CHUNK;
    const int tap = blockIdx.y, b = blockIdx.z, N = g.Ho * g.Wo;
    if (n >= g.pitch) return;
    const int c1 = min(c0 + CI_CHUNK, g.C), cpg = g.C / g.dgroups;
    float* cp = cols + ((size_t)b * g.Kpad + (size_t)c0 * g.kh * g.kw + tap) * g.pitch + n;
    const size_t cstep = (size_t)g.kh * g.kw * g.pitch;
    if (n >= N) {                                     // padding columns of the row pitch
        for (int ci = c0; ci < c1; ++ci, cp += cstep) *cp = 0.f;
        return;
    }
    int dg = -1;
    Tap t{};
    for (int ci = c0; ci < c1; ++ci, cp += cstep) {
        if (ci / cpg != dg) {
            dg = ci / cpg;
            t = tap_at(offset, g, b, dg, tap, n);
        }
        const float* xp = x + ((size_t)b * g.C + ci) * g.H * g.W;
        const float v00 = t.i00 >= 0 ? xp[t.i00] : 0.f, v01 = t.i01 >= 0 ? xp[t.i01] : 0.f;
        const float v10 = t.i10 >= 0 ? xp[t.i10] : 0.f, v11 = t.i11 >= 0 ? xp[t.i11] : 0.f;
        *cp = t.w00 * v00 + t.w01 * v01 + t.w10 * v10 + t.w11 * v11;
    }
}

// Same grid.  dx must be zero on entry (atomic adds); doffset_part[chunk][b][dgroups*2*kh*kw][N] receives this
// channel chunk's share of the offset gradient (summed over the chunks afterwards, so no atomics there).
__global__ __launch_bounds__(256) void deform_col2im_kernel(const float* __restrict__ dcols, const float* __restrict__ x,
                                                            const float* __restrict__ offset, float* __restrict__ dx,
                                                            float* __restrict__ doffset_part, DcnGeo g, int B) {
    const int nblocks = (g.pitch + 255) / 256;
    const int n = (blockIdx.x % nblocks) * 256 + threadIdx.x, chunk = blockIdx.x / nblocks, c0 = chunk * CI_CHUNK;
    const int tap = blockIdx.y, b = blockIdx.z, N = g.Ho * g.Wo;
    if (n >= N) return;
    const int c1 = min(c0 + CI_CHUNK, g.C), cpg = g.C / g.dgroups;
    const float* dp = dcols + ((size_t)b * g.Kpad + (size_t)c0 * g.kh * g.kw + tap) * g.pitch + n;
    const size_t cstep = (size_t)g.kh * g.kw * g.pitch;
    const int och = g.dgroups * 2 * g.kh * g.kw;
    int dg = -1;
    Tap t{};
    float gh = 0.f, gw = 0.f;
    auto flush = [&]() {
        if (dg >= 0 && doffset_part) {
            float* o = doffset_part + (((size_t)chunk * B + b) * och + (size_t)dg * 2 * g.kh * g.kw + 2 * tap) * N + n;
            o[0] = gh;
            o[N] = gw;
        }
    };
    for (int ci = c0; ci < c1; ++ci, dp += cstep) {
        if (ci / cpg != dg) {
            flush();
            dg = ci / cpg;
            gh = gw = 0.f;
            t = tap_at(offset, g, b, dg, tap, n);
        }
        const float gcol = *dp;
        const size_t plane = ((size_t)b * g.C + ci) * g.H * g.W;
        if (dx) {
            if (t.i00 >= 0) atomicAdd(dx + plane + t.i00, t.w00 * gcol);
            if (t.i01 >= 0) atomicAdd(dx + plane + t.i01, t.w01 * gcol);
            if (t.i10 >= 0) atomicAdd(dx + plane + t.i10, t.w10 * gcol);
            if (t.i11 >= 0) atomicAdd(dx + plane + t.i11, t.w11 * gcol);
        }
        if (doffset_part) {
            const float* xp = x + plane;
            const float v00 = t.i00 >= 0 ? xp[t.i00] : 0.f, v01 = t.i01 >= 0 ? xp[t.i01] : 0.f;
            const float v10 = t.i10 >= 0 ? xp[t.i10] : 0.f, v11 = t.i11 >= 0 ? xp[t.i11] : 0.f;
            // d/dh and d/dw of the bilinear sample (deform_conv_cuda_kernel.cu:162-183)
            gh += gcol * ((1.f - t.lw) * (v10 - v00) + t.lw * (v11 - v01));
            gw += gcol * ((1.f - t.lh) * (v01 - v00) + t.lh * (v11 - v10));
        }
    }
    flush();
}

__global__ __launch_bounds__(256) void sum_chunks_kernel(const float* __restrict__ part, float* __restrict__ out, long n,
                                                         int chunks) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int c = 0; c < chunks; ++c) s += part[(size_t)c * n + i];
    out[i] = s;
}

DcnGeo make_geo(int C, int H, int W, int kh, int kw, int stride, int pad, int dil, int dgroups, int Kpad, int pitch) {
    DcnGeo g;
    g.C = C; g.H = H; g.W = W; g.kh = kh; g.kw = kw; g.stride = stride; g.pad = pad; g.dil = dil; g.dgroups = dgroups;
    g.Ho = (H + 2 * pad - dil * (kh - 1) - 1) / stride + 1;
    g.Wo = (W + 2 * pad - dil * (kw - 1) - 1) / stride + 1;
    g.K = C * kh * kw;
    g.Kpad = Kpad;
    g.pitch = pitch;
    return g;
}

bool geo_ok(const DcnGeo& g, int B) {
    return B >= 0 && B <= 65535 && g.C > 0 && g.H > 0 && g.W > 0 && g.kh > 0 && g.kw > 0 && g.kh * g.kw <= 65535 &&
           g.stride > 0 && g.dil > 0 && g.pad >= 0 && g.dgroups > 0 && g.C % g.dgroups == 0 && g.Ho > 0 && g.Wo > 0 &&
           g.pitch >= g.Ho * g.Wo && g.Kpad >= g.K;
}

int chunks_of(const DcnGeo& g) { return (g.C + CI_CHUNK - 1) / CI_CHUNK; }

}  // namespace

extern "C" int spot_deform_im2col_f32(const float* x, const float* offset, float* cols, int B, int C, int H, int W, int kh,
                                      int kw, int stride, int pad, int dilation, int deformable_groups, int Kpad,
                                      int pitch, hipStream_t stream) {
    SPOT_CHECK_ARG(x && offset && cols, "deform_im2col: null pointer");
    const DcnGeo g = make_geo(C, H, W, kh, kw, stride, pad, dilation, deformable_groups, Kpad, pitch);
    SPOT_CHECK_ARG(geo_ok(g, B), "deform_im2col: bad geometry");
    if (B == 0) return SPOT_OK;
    if (g.Kpad > g.K) {      // rows beyond C*kh*kw are zero so that the GEMM's k tiles are whole
        SPOT_CHECK_HIP(hipMemset2DAsync(cols + (size_t)g.K * g.pitch, (size_t)g.Kpad * g.pitch * sizeof(float), 0,
                                        (size_t)(g.Kpad - g.K) * g.pitch * sizeof(float), B, stream));
    }
    const int nblocks = (g.pitch + 255) / 256;
    deform_im2col_kernel<<<dim3(nblocks * chunks_of(g), kh * kw, B), 256, 0, stream>>>(x, offset, cols, g);
    SPOT_CHECK_LAUNCH("deform_im2col");
    return SPOT_OK;
}

extern "C" long spot_deform_col2im_ws_floats(int B, int C, int H, int W, int kh, int kw, int stride, int pad, int dilation,
                                             int deformable_groups) {
    const DcnGeo g = make_geo(C, H, W, kh, kw, stride, pad, dilation, deformable_groups > 0 ? deformable_groups : 1, 0, 0);
    if (g.Ho <= 0 || g.Wo <= 0 || B <= 0) return 0;
    return (long)chunks_of(g) * B * g.dgroups * 2 * kh * kw * g.Ho * g.Wo;
}

extern "C" int spot_deform_col2im_f32(const float* dcols, const float* x, const float* offset, float* dx, float* doffset,
                                      float* ws, int B, int C, int H, int W, int kh, int kw, int stride, int pad,
                                      int dilation, int deformable_groups, int Kpad, int pitch, hipStream_t stream) {
    SPOT_CHECK_ARG(dcols && x && offset, "deform_col2im: null pointer");
    SPOT_CHECK_ARG(dx || doffset, "deform_col2im: nothing to compute");
    SPOT_CHECK_ARG(!doffset || ws, "deform_col2im: the offset gradient needs the workspace");
    const DcnGeo g = make_geo(C, H, W, kh, kw, stride, pad, dilation, deformable_groups, Kpad, pitch);
    SPOT_CHECK_ARG(geo_ok(g, B), "deform_col2im: bad geometry");
    if (B == 0) return SPOT_OK;
    if (dx) SPOT_CHECK_HIP(hipMemsetAsync(dx, 0, (size_t)B * C * H * W * sizeof(float), stream));
    const int nblocks = (g.pitch + 255) / 256, chunks = chunks_of(g);
    if (doffset && g.dgroups > 1)     // a channel chunk then writes only the groups it touches
        SPOT_CHECK_HIP(hipMemsetAsync(ws, 0, (size_t)chunks * B * g.dgroups * 2 * kh * kw * g.Ho * g.Wo * sizeof(float), stream));
    deform_col2im_kernel<<<dim3(nblocks * chunks, kh * kw, B), 256, 0, stream>>>(dcols, x, offset, dx, doffset ? ws : nullptr,
                                                                              g, B);
    SPOT_CHECK_LAUNCH("deform_col2im");
    if (doffset) {
        const long n = (long)B * g.dgroups * 2 * kh * kw * g.Ho * g.Wo;
        sum_chunks_kernel<<<spot_ceil_div(n, 256), 256, 0, stream>>>(ws, doffset, n, chunks);
        SPOT_CHECK_LAUNCH("deform_col2im(sum)");
    }
    return SPOT_OK;
}
