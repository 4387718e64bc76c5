// Does the fp32 matrix instruction overlap with vector ALU work of a partner wave on the same SIMD?
// 512-thread workgroups (waves w and w+4 share a SIMD), one workgroup per CU.
//   mode 0: waves 0-3 run MFMAs, waves 4-7 exit           mode 1: waves 4-7 run VALU, waves 0-3 exit
//   mode 2: waves 0-3 MFMA, waves 4-7 VALU (co-resident)  mode 3: all 8 waves MFMA     mode 4: all 8 waves VALU
// build: hipcc -O3 --offload-arch=gfx950 coexec_probe.hip -o coexec_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>   // 0: v_mfma_f32_32x32x2_f32, 1: v_mfma_f32_32x32x16_bf16
__device__ void mfma_loop(int iters, float* out) {
    f32x16 a0, a1;
    for (int e = 0; e < 16; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
    float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    bf16x8 bx, by;
    for (int e = 0; e < 8; ++e) { bx[e] = (__bf16)x; by[e] = (__bf16)y; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (KIND == 0) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
            } else {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(by, bx, a1, 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += a0[e] + a1[e];
    if (s == 123.456f) out[0] = s;
}

template <int VK>   // 0: v_fma_f32, 1: integer mul/xor hash, 2: v_exp_f32
__device__ void valu_loop(int iters, float* out) {
    float v[8];
    unsigned u[8];
    for (int e = 0; e < 8; ++e) { v[e] = threadIdx.x * 1e-3f + e; u[e] = threadIdx.x * 77u + e; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (VK == 0) v[e] = __builtin_fmaf(v[e], 1.0001f, 0.5f);
                else if (VK == 1) { u[e] ^= u[e] >> 15; u[e] *= 0x846ca68bU; }
                else v[e] = __builtin_amdgcn_exp2f(v[e]) * 0.5f;
            }
    }
    float s = 0.f;
    for (int e = 0; e < 8; ++e) s += v[e] + (float)u[e];
    if (s == 123.456f) out[0] = s;
}

template <int KIND, int VK>
__global__ __launch_bounds__(512) void probe(int mode, int mi, int vi, float* out) {
    const int grp = threadIdx.x >> 8;
    const bool do_m = (mode == 0 && grp == 0) || (mode == 2 && grp == 0) || mode == 3;
    const bool do_v = (mode == 1 && grp == 1) || (mode == 2 && grp == 1) || mode == 4;
    if (do_m) mfma_loop<KIND>(mi, out);
    if (do_v) valu_loop<VK>(vi, out);
}

template <int KIND, int VK>
void run(const char* name, int mi, int vi) {
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t s, e;
    hipEventCreate(&s);
    hipEventCreate(&e);
    printf("%s\n", name);
    for (int mode = 0; mode < 5; ++mode) {
        probe<KIND, VK><<<256, 512>>>(mode, mi, vi, out);
        hipDeviceSynchronize();
        hipEventRecord(s);
        for (int r = 0; r < 5; ++r) probe<KIND, VK><<<256, 512>>>(mode, mi, vi, out);
        hipEventRecord(e);
        hipEventSynchronize(e);
        float ms;
        hipEventElapsedTime(&ms, s, e);
        printf("  mode %d: %8.1f us\n", mode, ms * 1e3 / 5);
    }
    hipFree(out);
}

int main() {
    // loop lengths chosen so that the MFMA-only and VALU-only arms take about the same time
    run<0, 0>("f32 MFMA 32x32x2 vs v_fma_f32", 2000, 8000);
    run<0, 1>("f32 MFMA 32x32x2 vs int hash (shift, xor, v_mul_lo_u32)", 2000, 2000);
    run<0, 2>("f32 MFMA 32x32x2 vs v_exp_f32", 2000, 3000);
    run<1, 0>("bf16 MFMA 32x32x16 vs v_fma_f32", 4000, 8000);
    run<1, 1>("bf16 MFMA 32x32x16 vs int hash", 4000, 2000);
    return 0;
}
