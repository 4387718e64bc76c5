// Shared helpers for the SPOT hot-path HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define SPOT_OK 0
#define SPOT_ERR_ARG (-1)
#define SPOT_ERR_LAUNCH (-2)
#define SPOT_ERR_UNSUPPORTED (-3)

// Thread-local last-error text, returned by spot_last_error().
void spot_set_error(const char* fmt, ...);

#define SPOT_CHECK_ARG(cond, ...)                  \
    do {                                           \
        if (!(cond)) {                             \
            spot_set_error(__VA_ARGS__);           \
            return SPOT_ERR_ARG;                   \
        }                                          \
    } while (0)

// Call right after a kernel launch: surfaces launch-configuration errors.
#define SPOT_CHECK_LAUNCH(name)                                                   \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            spot_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return SPOT_ERR_LAUNCH;                                               \
        }                                                                         \
    } while (0)

// Around a HIP runtime call (memset, copy) issued from an entry point.
#define SPOT_CHECK_HIP(call)                                                          \
    do {                                                                              \
        hipError_t e__ = (call);                                                      \
        if (e__ != hipSuccess) {                                                      \
            spot_set_error("%s failed: %s", #call, hipGetErrorString(e__));           \
            return SPOT_ERR_LAUNCH;                                                   \
        }                                                                             \
    } while (0)

static inline int spot_ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

#define WAVE 64

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ---------------------------------------------------------------------------
// Counter-based dropout mask.  keep(seed, row, col) is a pure function of its
// arguments, so forward and backward kernels regenerate identical masks without
// storing them.  p_drop is carried as a 32-bit threshold: keep iff hash >= thr.
// ---------------------------------------------------------------------------
__host__ __device__ __forceinline__ unsigned spot_mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ unsigned spot_row_key(unsigned seed, unsigned row) { return spot_mix32(row ^ seed); }
__host__ __device__ __forceinline__ bool spot_keep(unsigned row_key, unsigned col, unsigned thr) {
    return spot_mix32(col + row_key) >= thr;
}
static inline unsigned spot_drop_threshold(float p) {
    if (p <= 0.f) return 0u;
    double t = (double)p * 4294967296.0;
    return t >= 4294967295.0 ? 0xffffffffu : (unsigned)t;
}
