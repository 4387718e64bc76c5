// In-network strong augmentation of the student's search crops (ltr/models/pre_input/transforms_in_net.py:121-216,
// 362-398), batched: the reference loops over the samples of a batch on the host, one flip + affine grid +
// grid_sample + fill per sample; here one launch transforms the whole batch.
//
//   rotate_flip_kernel   left/right flip, then rotation by an integer angle about the crop centre with
//                        nearest-neighbour sampling; pixels that fall outside take the sample's fill colour
//                        (:168-178, :240-331).  HBM-bound: one read, one write of the batch.
//   ellipse_boxes_kernel the box follows as the bounding box of its rotated inscribed ellipse, sampled at 360
//                        points (:180-214).
//   channel_scale_kernel per-(sample, channel) scaling of a crop batch (the GrayScale transform, :384-396).
// This file is compiled with -ffp-contract=off: the reference rounds every multiply and add separately.
#include "common.h"
#include "../../include/spot_hip.h"

namespace {

// theta: [721][6] inverse affine matrices for the integer angles -360..360, in centred pixel coordinates
__global__ __launch_bounds__(256) void rotate_flip_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          const float* __restrict__ theta, const float* __restrict__ angle,
                                                          const unsigned char* __restrict__ lr_flip,
                                                          const unsigned char* __restrict__ apply,
                                                          const float* __restrict__ fill, int C, int H, int W) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const size_t base = (size_t)b * C * H * W;
    if (!apply[b]) {
        for (int c = 0; c < C; ++c) out[base + (size_t)c * H * W + p] = in[base + (size_t)c * H * W + p];
        return;
    }
    const int y = p / W, x = p - y * W;
    int a = (int)angle[b];                       // int(angle[i]) of the reference
    a = a < -360 ? -360 : (a > 360 ? 360 : a);
    const float* t = theta + (size_t)(a + 360) * 6;
    const float hw = 0.5f * (float)W, hh = 0.5f * (float)H;
    // base grid: x - W/2 + 0.5 (a linspace of unit step), rescaled theta = theta / (0.5 w, 0.5 h)
    const float xb = (float)x - hw + 0.5f, yb = (float)y - hh + 0.5f;
    const float gx = xb * (t[0] / hw) + yb * (t[1] / hw) + (t[2] / hw);
    const float gy = xb * (t[3] / hh) + yb * (t[4] / hh) + (t[5] / hh);
    // grid_sample, align_corners = False, nearest (round half to even)
    const float fx = ((gx + 1.f) * (float)W - 1.f) / 2.f, fy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
    const float rx = rintf(fx), ry = rintf(fy);
    const bool inside = rx >= 0.f && rx < (float)W && ry >= 0.f && ry < (float)H;
    int sx = (int)rx;
    const int sy = (int)ry;
    if (lr_flip[b]) sx = W - 1 - sx;             // the sample was flipped before the rotation
    for (int c = 0; c < C; ++c) {
        const size_t pl = base + (size_t)c * H * W;
        out[pl + p] = inside ? in[pl + (size_t)sy * W + sx] : fill[b * C + c];
    }
}

__global__ __launch_bounds__(64) void ellipse_boxes_kernel(const float* __restrict__ bbox, const float* __restrict__ angle,
                                                           const unsigned char* __restrict__ lr_flip,
                                                           const unsigned char* __restrict__ apply, float* __restrict__ out,
                                                           int W, int H) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float bx = bbox[b * 4], by = bbox[b * 4 + 1], bw = bbox[b * 4 + 2], bh = bbox[b * 4 + 3];
    const float rw = 0.5f * bw, rh = 0.5f * bh;
    float cx = bx + rw;
    const float cy = by + rh;
    if (lr_flip[b]) cx = (float)(W - 1) - cx;
    const float d2r = (float)(3.14159265358979323846 / 180.0);
    const float ar = angle[b] * d2r;
    const float alpha = cosf(ar), beta = sinf(ar);
    const float x0 = (float)W * 0.5f, y0 = (float)H * 0.5f;
    const float m02 = (1.f - alpha) * x0 - beta * y0, m12 = beta * x0 + (1.f - alpha) * y0;
    float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
    for (int k = lane; k < 360; k += 64) {
        const float a = (float)k * d2r;
        const float ex = rw * sinf(a) + cx, ey = rh * cosf(a) + cy;
        const float tx = alpha * ex + beta * ey + m02;
        const float ty = -beta * ex + alpha * ey + m12;
        xmin = fminf(xmin, tx); xmax = fmaxf(xmax, tx);
        ymin = fminf(ymin, ty); ymax = fmaxf(ymax, ty);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        xmin = fminf(xmin, __shfl_xor(xmin, o)); xmax = fmaxf(xmax, __shfl_xor(xmax, o));
        ymin = fminf(ymin, __shfl_xor(ymin, o)); ymax = fmaxf(ymax, __shfl_xor(ymax, o));
    }
    if (lane == 0) {
        const bool ap = apply[b];
        out[b * 4] = fmaxf(ap ? xmin : bx, 0.f);
        out[b * 4 + 1] = fmaxf(ap ? ymin : by, 0.f);
        out[b * 4 + 2] = fmaxf(ap ? xmax - xmin : bw, 0.f);
        out[b * 4 + 3] = fmaxf(ap ? ymax - ymin : bh, 0.f);
    }
}

__global__ __launch_bounds__(256) void channel_scale_kernel(float* __restrict__ x, const float* __restrict__ scale, long planes,
                                                            int HW4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= planes * HW4) return;
    const float s = scale[i / HW4];
    float4 v = reinterpret_cast<float4*>(x)[i];
    v.x *= s; v.y *= s; v.z *= s; v.w *= s;
    reinterpret_cast<float4*>(x)[i] = v;
}

// element by element, for planes whose size is not a multiple of 4 (255x255 and 127x127 crops)
__global__ __launch_bounds__(256) void channel_scale_scalar_kernel(float* __restrict__ x, const float* __restrict__ scale,
                                                                   long planes, int HW) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= planes * HW) return;
    x[i] *= scale[i / HW];
}

}  // namespace

extern "C" int spot_rotate_flip_nearest_f32(const float* in, float* out, const float* theta_table, const float* angle,
                                            const unsigned char* lr_flip, const unsigned char* apply, const float* fill,
                                            int B, int C, int H, int W, hipStream_t stream) {
    SPOT_CHECK_ARG(in && out && theta_table && angle && lr_flip && apply && fill, "rotate_flip: null pointer");
    SPOT_CHECK_ARG(in != out, "rotate_flip: the transform gathers from its input and cannot run in place");
    SPOT_CHECK_ARG(B >= 0 && B <= 65535 && C > 0 && H > 0 && W > 0, "rotate_flip: bad shape");
    if (B == 0) return SPOT_OK;
    rotate_flip_kernel<<<dim3(spot_ceil_div((long)H * W, 256), B), 256, 0, stream>>>(in, out, theta_table, angle, lr_flip,
                                                                                     apply, fill, C, H, W);
    SPOT_CHECK_LAUNCH("rotate_flip");
    return SPOT_OK;
}

extern "C" int spot_ellipse_rotate_boxes_f32(const float* bbox, const float* angle, const unsigned char* lr_flip,
                                             const unsigned char* apply, float* out, int B, int W, int H,
                                             hipStream_t stream) {
    SPOT_CHECK_ARG(bbox && angle && lr_flip && apply && out, "ellipse_rotate_boxes: null pointer");
    SPOT_CHECK_ARG(B >= 0 && W > 0 && H > 0, "ellipse_rotate_boxes: bad shape");
    if (B == 0) return SPOT_OK;
    ellipse_boxes_kernel<<<B, 64, 0, stream>>>(bbox, angle, lr_flip, apply, out, W, H);
    SPOT_CHECK_LAUNCH("ellipse_rotate_boxes");
    return SPOT_OK;
}

extern "C" int spot_channel_scale_f32(float* x, const float* scale, long planes, int HW, hipStream_t stream) {
    SPOT_CHECK_ARG(x && scale, "channel_scale: null pointer");
    SPOT_CHECK_ARG(planes >= 0 && HW > 0, "channel_scale: bad shape HW=%d", HW);
    if (planes == 0) return SPOT_OK;
    const bool vec = HW % 4 == 0 && ((uintptr_t)x & 15) == 0;
    const long n = vec ? planes * (HW / 4) : planes * (long)HW;
    SPOT_CHECK_ARG((n + 255) / 256 < (1L << 31), "channel_scale: too large");
    if (vec) channel_scale_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(x, scale, planes, HW / 4);
    else channel_scale_scalar_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(x, scale, planes, HW);
    SPOT_CHECK_LAUNCH("channel_scale");
    return SPOT_OK;
}
