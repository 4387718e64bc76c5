// Square crop of a frame around a target, padded with the frame's channel average and resized to the network's
// input size, on the device: what the reference does per frame on the host with numpy slicing and cv2.resize
// (pysot_toolkit/trackers/transt_tracker.py:41-101 get_subwindow; ltr/data/processing_utils.py:8-91 sample_target).
// The frame is uint8 HWC as the loader delivers it; the crop comes out uint8 CHW, which is what the preprocess kernel
// reads.  Pixels of the crop window outside the frame take the pad colour (the channel average truncated to uint8, as
// numpy's assignment into a uint8 array does).  The resize restates OpenCV's INTER_LINEAR for 8-bit images: sample
// position (d + 0.5) * scale - 0.5, tap weights rounded to 1/2048, horizontal pass in integers, vertical pass
// ((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2; an exact 2:1 reduction is the 2x2 box average
// OpenCV substitutes for it.  One thread per output pixel (all three channels), lanes along the output row.
#include "common.h"
#include "../../include/spot_hip.h"

namespace {

struct Lin {
    int i0, i1;      // source indices (clamped into the crop window)
    int a0, a1;      // weights in 1/2048
};

__device__ __forceinline__ int round_short(float v) {      // saturate_cast<short>(float): round half to even
    const float r = rintf(v);
    return r > 32767.f ? 32767 : (r < -32768.f ? -32768 : (int)r);
}

__device__ __forceinline__ Lin horizontal_tap(int d, double scale, int n) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= n - 1) { f = 0.f; s = n - 1; }
    Lin t;
    t.i0 = s;
    t.i1 = min(s + 1, n - 1);
    t.a0 = round_short((1.f - f) * 2048.f);
    t.a1 = round_short(f * 2048.f);
    return t;
}

__device__ __forceinline__ Lin vertical_tap(int d, double scale, int n) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    const int s = (int)floorf(f);
    f -= (float)s;
    Lin t;
    t.i0 = min(max(s, 0), n - 1);
    t.i1 = min(max(s + 1, 0), n - 1);
    t.a0 = round_short((1.f - f) * 2048.f);
    t.a1 = round_short(f * 2048.f);
    return t;
}

__device__ __forceinline__ int src(const spot_crop_desc& c, int y, int x, int ch) {
    const int iy = c.y0 + y, ix = c.x0 + x;
    if (iy < 0 || iy >= c.H || ix < 0 || ix >= c.W) return c.pad[ch];
    return c.image[((size_t)iy * c.W + ix) * 3 + ch];
}

__device__ __forceinline__ void crop_pixel(const spot_crop_desc& c, unsigned char* __restrict__ out, int S, int dy, int dx) {
    const size_t plane = (size_t)S * S, o = (size_t)dy * S + dx;
    if (c.crop == S) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) out[ch * plane + o] = (unsigned char)src(c, dy, dx, ch);
        return;
    }
    if (c.crop == 2 * S) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
            out[ch * plane + o] = (unsigned char)((src(c, 2 * dy, 2 * dx, ch) + src(c, 2 * dy, 2 * dx + 1, ch) +
                                                   src(c, 2 * dy + 1, 2 * dx, ch) + src(c, 2 * dy + 1, 2 * dx + 1, ch) + 2) >> 2);
        return;
    }
    const double scale = (double)c.crop / (double)S;
    const Lin h = horizontal_tap(dx, scale, c.crop), v = vertical_tap(dy, scale, c.crop);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const int r0 = src(c, v.i0, h.i0, ch) * h.a0 + src(c, v.i0, h.i1, ch) * h.a1;
        const int r1 = src(c, v.i1, h.i0, ch) * h.a0 + src(c, v.i1, h.i1, ch) * h.a1;
        const int val = (((v.a0 * (r0 >> 4)) >> 16) + ((v.a1 * (r1 >> 4)) >> 16) + 2) >> 2;
        out[ch * plane + o] = (unsigned char)min(max(val, 0), 255);
    }
}

__global__ __launch_bounds__(256) void crop_resize_kernel(const spot_crop_desc* __restrict__ table, spot_crop_desc one,
                                                          unsigned char* __restrict__ out, int S) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= S * S) return;
    const spot_crop_desc c = table ? table[blockIdx.y] : one;
    crop_pixel(c, out + (size_t)blockIdx.y * 3 * S * S, S, idx / S, idx % S);
}

}  // namespace

extern "C" int spot_crop_resize_u8(const spot_crop_desc* crop /* host */, unsigned char* out, int out_sz, hipStream_t stream) {
    SPOT_CHECK_ARG(crop && out && out_sz > 0, "crop_resize: bad arguments");
    SPOT_CHECK_ARG(crop->image && crop->H > 0 && crop->W > 0 && crop->crop > 0, "crop_resize: bad frame or crop size");
    crop_resize_kernel<<<dim3(spot_ceil_div((long)out_sz * out_sz, 256), 1), 256, 0, stream>>>(nullptr, *crop, out, out_sz);
    SPOT_CHECK_LAUNCH("crop_resize");
    return SPOT_OK;
}

extern "C" int spot_crop_resize_batch_u8(const spot_crop_desc* table_dev, int B, unsigned char* out, int out_sz,
                                         hipStream_t stream) {
    SPOT_CHECK_ARG(B >= 0 && B <= 65535 && out_sz > 0 && (B == 0 || (table_dev && out)), "crop_resize_batch: bad arguments");
    if (B == 0) return SPOT_OK;
    spot_crop_desc none = {};
    crop_resize_kernel<<<dim3(spot_ceil_div((long)out_sz * out_sz, 256), B), 256, 0, stream>>>(table_dev, none, out, out_sz);
    SPOT_CHECK_LAUNCH("crop_resize_batch");
    return SPOT_OK;
}
