// Lab for the fp32 MFMA GEMM (C[M,N] = A[M,K] W[N,K]^T + bias + residual, both operands k-contiguous): where does a
// launch's time go, and which schedule removes it?  Stand-alone (no torch): timed with HIP events, checked against a
// CPU sum on sampled elements.
//   build: hipcc -O3 --offload-arch=gfx950 gemm_lab.hip -o gemm_lab
//   run:   ./gemm_lab            (all variants over the step's main shapes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BK = 32, KPAD = BK + 4;

struct Args {
    const float* A; const float* B; float* C; const float* bias; const float* res;
    int M, N, K, prio_div; long long* clk = nullptr;
};

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// v0: the library's loop (register-staged single LDS buffer, two barriers per k tile), with knobs
//   PRIO 1: wave priority 3 - ((workgroup / 8 / prio_div) & 3) for the k loop, 0 for the epilogue
//   EPI  0: full epilogue   1: everything but the global stores   2: no epilogue at all
//   3: stores straight from the accumulators (dword per lane, 128-byte row segments), no LDS transpose
// ---------------------------------------------------------------------------------------------------------------------
template <int TS, int PRIO, int EPI>
__global__ __launch_bounds__(256, TS == 2 ? 3 : 6) void gemm_v0(Args g) {
    constexpr int BM = 64 * TS, BN = 64 * TS, NL = 2 * TS, TILE_F = BM * KPAD, CPAD = BN + 4, WT = 32 * TS;
    constexpr int RSH = TS == 2 ? 5 : 4, RMASK = (1 << RSH) - 1;
    __shared__ __attribute__((aligned(16))) float lds[2 * TILE_F];
    float* const As = lds;
    float* const Bs = lds + TILE_F;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, l31 = lane & 31;
    const int ntn = g.N / BN, ntm = g.M / BM, nwg = ntm * ntn;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, x = bid & 7, loc = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
    }
    const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;
    if (PRIO == 1) {
        const int p = ((blockIdx.x >> 3) / g.prio_div) & 3;
        if (p == 0) __builtin_amdgcn_s_setprio(3);
        else if (p == 1) __builtin_amdgcn_s_setprio(2);
        else if (p == 2) __builtin_amdgcn_s_setprio(1);
    }
    f32x16 acc[TS][TS];
#pragma unroll
    for (int t = 0; t < TS; ++t)
#pragma unroll
        for (int u = 0; u < TS; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;
    f32x4 sa[NL], sb[NL];
    unsigned offA[NL], offB[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int f = tid + 256 * i;
        offA[i] = (unsigned)((m0 + (f >> 3)) * g.K + ((f & 7) << 2));
        offB[i] = (unsigned)((n0 + (f >> 3)) * g.K + ((f & 7) << 2));
    }
    const int nkt = g.K / BK;
    int fk = 0;
    auto fetch = [&]() {
#pragma unroll
        for (int i = 0; i < NL; ++i) sa[i] = *reinterpret_cast<const f32x4*>(g.A + fk + offA[i]);
#pragma unroll
        for (int i = 0; i < NL; ++i) sb[i] = *reinterpret_cast<const f32x4*>(g.B + fk + offB[i]);
        fk += BK;
    };
    auto store = [&]() {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int f = tid + 256 * i;
            *reinterpret_cast<f32x4*>(As + (f >> 3) * KPAD + ((f & 7) << 2)) = sa[i];
            *reinterpret_cast<f32x4*>(Bs + (f >> 3) * KPAD + ((f & 7) << 2)) = sb[i];
        }
    };
    fetch();
    store();
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + 1 < nkt;
        if (more) fetch();
#pragma unroll
        for (int gk = 0; gk < BK / 8; ++gk) {
            float4 a[TS], b[TS];
#pragma unroll
            for (int t = 0; t < TS; ++t) a[t] = *reinterpret_cast<const float4*>(As + (wm * WT + t * 32 + l31) * KPAD + gk * 8 + 4 * h);
#pragma unroll
            for (int u = 0; u < TS; ++u) b[u] = *reinterpret_cast<const float4*>(Bs + (wn * WT + u * 32 + l31) * KPAD + gk * 8 + 4 * h);
#pragma unroll
            for (int t = 0; t < TS; ++t)
#pragma unroll
                for (int u = 0; u < TS; ++u) {
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].x, b[u].x, acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, b[u].y, acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].z, b[u].z, acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].w, b[u].w, acc[t][u], 0, 0, 0);
                }
        }
        __syncthreads();
        if (more) {
            store();
            __syncthreads();
        }
    }
    if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    if (EPI == 2) {
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < TS; ++t)
#pragma unroll
            for (int u = 0; u < TS; ++u)
#pragma unroll
                for (int e = 0; e < 16; ++e) sum += acc[t][u][e];
        if (sum == 12345.678f) g.C[0] = 1.f;
        return;
    }
    if (EPI == 3) {
#pragma unroll
        for (int t = 0; t < TS; ++t)
#pragma unroll
            for (int u = 0; u < TS; ++u) {
                const int n = n0 + wn * WT + u * 32 + l31;
                const float bv = g.bias[n];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm * WT + t * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    g.C[(long)m * g.N + n] = acc[t][u][e] + bv + g.res[(long)m * g.N + n];
                }
            }
        return;
    }
    float* Cs = lds;
    const int c4 = (tid & RMASK) << 2;
    const int n = n0 + c4;
    const float4 bv = *reinterpret_cast<const float4*>(g.bias + n);
    for (int half = 0; half < TS; ++half) {
        if (half) __syncthreads();
        if (TS == 1 || wm == half) {
            const int rbase = TS == 1 ? wm * 32 : 0;
#pragma unroll
            for (int t = 0; t < TS; ++t)
#pragma unroll
                for (int u = 0; u < TS; ++u)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        Cs[(rbase + t * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * CPAD + wn * WT + u * 32 + l31] = acc[t][u][e];
        }
        __syncthreads();
        const int mh = m0 + half * 64;
#pragma unroll 4
        for (int i = 0; i < (64 >> (8 - RSH)); ++i) {
            const int row = (i << (8 - RSH)) + (tid >> RSH);
            const int m = mh + row;
            float4 v = *reinterpret_cast<const float4*>(Cs + row * CPAD + c4);
            const float4 r4 = *reinterpret_cast<const float4*>(g.res + (long)m * g.N + n);
            v.x += bv.x + r4.x; v.y += bv.y + r4.y; v.z += bv.z + r4.z; v.w += bv.w + r4.w;
            if (EPI == 0 || v.x == 12345.678f) *reinterpret_cast<float4*>(g.C + (long)m * g.N + n) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ void fill_kernel(float* p, long n, unsigned seed) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = (float)(x >> 8) * (2.0f / 16777216.0f) - 1.0f;
}

static float* dev_rand(long n, unsigned seed) {
    float* p;
    CK(hipMalloc(&p, n * sizeof(float)));
    fill_kernel<<<(unsigned)((n + 255) / 256), 256>>>(p, n, seed);
    return p;
}

template <typename F>
static double time_us(F&& f, int iters = 10) {
    static hipEvent_t s = nullptr, e = nullptr;
    if (!s) { CK(hipEventCreate(&s)); CK(hipEventCreate(&e)); }
    f();
    CK(hipEventRecord(s));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(e));
    CK(hipEventSynchronize(e));
    float ms;
    CK(hipEventElapsedTime(&ms, s, e));
    CK(hipGetLastError());
    return ms * 1e3 / iters;
}

#include <functional>
#include <algorithm>
#include <string>
struct Variant { std::string tag; std::function<double()> run; std::vector<double> t; };
static void run_all(std::vector<Variant>& vs, double gflop, int rounds = 7) {
    for (int r = 0; r < rounds; ++r)
        for (auto& v : vs) v.t.push_back(v.run());
    for (auto& v : vs) {
        std::sort(v.t.begin(), v.t.end());
        const double mn = v.t.front(), md = v.t[v.t.size() / 2];
        printf("   %-46s min %8.1f us %6.1f TF | median %8.1f us %6.1f TF\n", v.tag.c_str(), mn, gflop / mn * 1e3, md, gflop / md * 1e3);
    }
}

static void check(const Args& a, const char* tag) {
    // sampled elements against a double sum on the host
    std::vector<float> hA((size_t)a.M * a.K), hB((size_t)a.N * a.K), hC((size_t)a.M * a.N), hb(a.N), hr((size_t)a.M * a.N);
    CK(hipMemcpy(hA.data(), a.A, hA.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hB.data(), a.B, hB.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hC.data(), a.C, hC.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), a.bias, hb.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), a.res, hr.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int s = 0; s < 4000; ++s) {
        const long m = (long)((s * 7919L + (s % 7) * 1237L) % a.M), n = (s * 104729L + 13) % a.N;
        double ref = hb[n] + hr[m * a.N + n];
        for (int k = 0; k < a.K; ++k) ref += (double)hA[m * a.K + k] * hB[n * a.K + k];
        worst = fmax(worst, fabs(ref - hC[m * a.N + n]));
    }
    printf("   check %-28s max |err| on 4000 samples %.3e %s\n", tag, worst, worst < 2e-5 * sqrt((double)a.K) ? "ok" : "WRONG");
}


// ---------------------------------------------------------------------------------------------------------------------
// Loop anatomy: the same grid, tile and MFMA count per wave, with parts of the k step removed
//   PART 0: matrix instructions on register operands only          (the matrix pipe's rate at this occupancy)
//   PART 1: + fragment reads from LDS (no barriers, no global loads; LDS holds whatever it holds)
//   PART 2: + the two barriers per k tile
//   PART 3: + the global loads (kept alive by a dummy use), no LDS stores
// ---------------------------------------------------------------------------------------------------------------------
template <int TS, int PART>
__global__ __launch_bounds__(256, TS == 2 ? 3 : 6) void loop_probe(Args g) {
    constexpr int BM = 64 * TS, BN = 64 * TS, NL = 2 * TS, TILE_F = BM * KPAD, WT = 32 * TS;
    __shared__ __attribute__((aligned(16))) float lds[2 * TILE_F];
    float* const As = lds;
    float* const Bs = lds + TILE_F;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, l31 = lane & 31;
    const int ntn = g.N / BN;
    const int bid = blockIdx.x;
    const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;
    f32x16 acc[TS][TS];
#pragma unroll
    for (int t = 0; t < TS; ++t)
#pragma unroll
        for (int u = 0; u < TS; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;
    for (int i = tid; i < 2 * TILE_F; i += 256) lds[i] = 1e-3f * (i & 127);
    __syncthreads();
    f32x4 sa[NL], sb[NL];
    unsigned offA[NL], offB[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int f = tid + 256 * i;
        offA[i] = (unsigned)((m0 + (f >> 3)) * g.K + ((f & 7) << 2));
        offB[i] = (unsigned)((n0 + (f >> 3)) * g.K + ((f & 7) << 2));
        sa[i] = 0.f; sb[i] = 0.f;
    }
    float keep = 0.f;
    const int nkt = g.K / BK;
    int fk = 0;
    const long long c0 = clock64(), w0 = wall_clock64();
    float4 ra = make_float4(1e-3f * lane, 2e-3f, 3e-3f, 4e-3f), rb = make_float4(1.f, 0.5f, 0.25f, 0.125f);
    for (int kt = 0; kt < nkt; ++kt) {
        if (PART >= 3) {
#pragma unroll
            for (int i = 0; i < NL; ++i) sa[i] = *reinterpret_cast<const f32x4*>(g.A + fk + offA[i]);
#pragma unroll
            for (int i = 0; i < NL; ++i) sb[i] = *reinterpret_cast<const f32x4*>(g.B + fk + offB[i]);
            fk += BK;
        }
#pragma unroll
        for (int gk = 0; gk < BK / 8; ++gk) {
            float4 a[TS], b[TS];
#pragma unroll
            for (int t = 0; t < TS; ++t) a[t] = PART >= 1 ? *reinterpret_cast<const float4*>(As + (wm * WT + t * 32 + l31) * KPAD + gk * 8 + 4 * h) : ra;
#pragma unroll
            for (int u = 0; u < TS; ++u) b[u] = PART >= 1 ? *reinterpret_cast<const float4*>(Bs + (wn * WT + u * 32 + l31) * KPAD + gk * 8 + 4 * h) : rb;
#pragma unroll
            for (int t = 0; t < TS; ++t)
#pragma unroll
                for (int u = 0; u < TS; ++u) {
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].x, b[u].x, acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, b[u].y, acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].z, b[u].z, acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].w, b[u].w, acc[t][u], 0, 0, 0);
                }
        }
        if (PART >= 2) {
            __syncthreads();
            __syncthreads();
        }
        if (PART >= 3) {
#pragma unroll
            for (int i = 0; i < NL; ++i) keep += sa[i].x + sb[i].w;
        }
    }
    float sum = keep;
#pragma unroll
    for (int t = 0; t < TS; ++t)
#pragma unroll
        for (int u = 0; u < TS; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) sum += acc[t][u][e];
    if (sum == 12345.678f) g.C[0] = 1.f;
    if (g.clk && tid == 0 && (blockIdx.x & 255) == 17) {
        const long long c1 = clock64(), w1 = wall_clock64();
        g.clk[2 * (blockIdx.x >> 8)] = c1 - c0;
        g.clk[2 * (blockIdx.x >> 8) + 1] = w1 - w0;
    }
}

template <int TS, int PART>
static double run_probe(const Args& a) {
    constexpr int BM = 64 * TS;
    dim3 grid((a.M / BM) * (a.N / BM));
    return time_us([&] { loop_probe<TS, PART><<<grid, 256>>>(a); });
}


// ---------------------------------------------------------------------------------------------------------------------
// v2: global -> LDS directly (global_load_lds_dwordx4, no register staging), a ring of S stages, ONE barrier per k tile,
// counted vmcnt so that S-2 tiles stay in flight across the barrier.  LDS rows are the operand's 128-byte k rows, unpadded
// (one wave instruction writes 8 whole rows); bank conflicts of the 16-byte fragment reads are avoided by storing 16-byte
// chunk c of row r at chunk position c ^ ((r >> 1) & 7), applied on the global source address.
// ---------------------------------------------------------------------------------------------------------------------
// One direct global -> LDS load of 16 bytes per lane, written as asm so that the compiler's own wait-count insertion does not
// know of it (it would otherwise drain every such load before each LDS read): lane's source = base + voff bytes, the
// wave's 1 KiB lands at LDS byte address lds_addr (wave-uniform) in lane order.
__device__ __forceinline__ void glds16(const float* base, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_addr) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int TS, int S, int EPI, int PRIO = 0>
__global__ __launch_bounds__(256) void gemm_v2(Args g) {
    constexpr int BM = 64 * TS, BN = 64 * TS, WT = 32 * TS, CPAD = BN + 4;
    constexpr int A_F = BM * 32, STAGE_F = (BM + BN) * 32;          // floats
    constexpr int GA = BM / 32;                                     // direct loads per wave per operand tile (8 rows each)
    constexpr int G = 2 * GA;
    constexpr int RSH = TS == 2 ? 5 : 4, RMASK = (1 << RSH) - 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, l31 = lane & 31;
    const int ntn = g.N / BN, ntm = g.M / BM, nwg = ntm * ntn;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, x = bid & 7, loc = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
    }
    const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;
    if (PRIO == 1) {
        if (((blockIdx.x >> 3) / g.prio_div) & 1) __builtin_amdgcn_s_setprio(2);
    }
    f32x16 acc[TS][TS];
#pragma unroll
    for (int t = 0; t < TS; ++t)
#pragma unroll
        for (int u = 0; u < TS; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;
    // source offsets of this lane's 16 bytes in each of the wave's direct loads
    unsigned offA[GA], offB[GA];
#pragma unroll
    for (int j = 0; j < GA; ++j) {
        const int row = 8 * (wave * GA + j) + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
        offA[j] = 4u * (unsigned)((m0 + row) * g.K + 4 * c);
        offB[j] = 4u * (unsigned)((n0 + row) * g.K + 4 * c);
    }
    // fragment read offsets (floats) inside a stage: row l31 of a 32-row tile, chunk 2 gk + h, swizzled
    int fo[4];
#pragma unroll
    for (int gk = 0; gk < 4; ++gk) fo[gk] = l31 * 32 + (((2 * gk + h) ^ ((l31 >> 1) & 7)) << 2);
    const int nkt = g.K / BK;
    const unsigned lds0 = (unsigned)(uintptr_t)lds + (unsigned)__builtin_amdgcn_readfirstlane(wave) * (GA * 1024u);
    auto issue = [&](int kt, int slot) {
        const unsigned dst = lds0 + (unsigned)slot * (STAGE_F * 4u);
        const float* Ab = g.A + kt * BK;
        const float* Bb = g.B + kt * BK;
#pragma unroll
        for (int j = 0; j < GA; ++j) glds16(Ab, offA[j], dst + j * 1024u);
#pragma unroll
        for (int j = 0; j < GA; ++j) glds16(Bb, offB[j], dst + A_F * 4u + j * 1024u);
    };
#pragma unroll
    for (int s = 0; s < S - 1; ++s)
        if (s < nkt) issue(s, s);
    int slot = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const int ahead = min(S - 2, nkt - 1 - kt);          // tiles issued after tile kt that may stay in flight
        if (ahead >= 2) wait_vm<2 * G>();
        else if (ahead == 1) wait_vm<G>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + S - 1 < nkt) issue(kt + S - 1, slot == 0 ? S - 1 : slot - 1);
        const float* As = lds + slot * STAGE_F;
        const float* Bs = As + A_F;
        if (PRIO == 2) {          // all four k groups' fragments requested up front: the matrix instructions of group g run
                                  // while the reads of the later groups are still in flight
            float4 a[4][TS], b[4][TS];
#pragma unroll
            for (int gk = 0; gk < 4; ++gk) {
#pragma unroll
                for (int t = 0; t < TS; ++t) a[gk][t] = *reinterpret_cast<const float4*>(As + (wm * WT + t * 32) * 32 + fo[gk]);
#pragma unroll
                for (int u = 0; u < TS; ++u) b[gk][u] = *reinterpret_cast<const float4*>(Bs + (wn * WT + u * 32) * 32 + fo[gk]);
            }
#pragma unroll
            for (int gk = 0; gk < 4; ++gk)
#pragma unroll
                for (int t = 0; t < TS; ++t)
#pragma unroll
                    for (int u = 0; u < TS; ++u) {
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gk][t].x, b[gk][u].x, acc[t][u], 0, 0, 0);
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gk][t].y, b[gk][u].y, acc[t][u], 0, 0, 0);
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gk][t].z, b[gk][u].z, acc[t][u], 0, 0, 0);
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gk][t].w, b[gk][u].w, acc[t][u], 0, 0, 0);
                    }
        } else {
#pragma unroll
        for (int gk = 0; gk < 4; ++gk) {
            float4 a[TS], b[TS];
#pragma unroll
            for (int t = 0; t < TS; ++t) a[t] = *reinterpret_cast<const float4*>(As + (wm * WT + t * 32) * 32 + fo[gk]);
#pragma unroll
            for (int u = 0; u < TS; ++u) b[u] = *reinterpret_cast<const float4*>(Bs + (wn * WT + u * 32) * 32 + fo[gk]);
#pragma unroll
            for (int t = 0; t < TS; ++t)
#pragma unroll
                for (int u = 0; u < TS; ++u) {
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].x, b[u].x, acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, b[u].y, acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].z, b[u].z, acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].w, b[u].w, acc[t][u], 0, 0, 0);
                }
        }
        }
        slot = slot + 1 == S ? 0 : slot + 1;
    }
    if (EPI == 2) {
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < TS; ++t)
#pragma unroll
            for (int u = 0; u < TS; ++u)
#pragma unroll
                for (int e = 0; e < 16; ++e) sum += acc[t][u][e];
        if (sum == 12345.678f) g.C[0] = 1.f;
        return;
    }
    if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    __syncthreads();                                            // every wave has read the last stage
    float* Cs = lds;
    const int c4 = (tid & RMASK) << 2;
    const int n = n0 + c4;
    const float4 bv = *reinterpret_cast<const float4*>(g.bias + n);
    for (int half = 0; half < TS; ++half) {
        if (half) __syncthreads();
        if (TS == 1 || wm == half) {
            const int rbase = TS == 1 ? wm * 32 : 0;
#pragma unroll
            for (int t = 0; t < TS; ++t)
#pragma unroll
                for (int u = 0; u < TS; ++u)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        Cs[(rbase + t * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * CPAD + wn * WT + u * 32 + l31] = acc[t][u][e];
        }
        __syncthreads();
        const int mh = m0 + half * 64;
#pragma unroll 4
        for (int i = 0; i < (64 >> (8 - RSH)); ++i) {
            const int row = (i << (8 - RSH)) + (tid >> RSH);
            const int m = mh + row;
            float4 v = *reinterpret_cast<const float4*>(Cs + row * CPAD + c4);
            if (EPI == 4) {
                v.x = fmaxf(v.x + bv.x, 0.f); v.y = fmaxf(v.y + bv.y, 0.f); v.z = fmaxf(v.z + bv.z, 0.f); v.w = fmaxf(v.w + bv.w, 0.f);
            } else {
                const float4 r4 = *reinterpret_cast<const float4*>(g.res + (long)m * g.N + n);
                v.x += bv.x + r4.x; v.y += bv.y + r4.y; v.z += bv.z + r4.z; v.w += bv.w + r4.w;
            }
            *reinterpret_cast<float4*>(g.C + (long)m * g.N + n) = v;
        }
    }
}

template <int TS, int S, int EPI, int PRIO = 0>
static double run_v2(const Args& a, int wg_per_cu_hint = 0) {
    constexpr int BM = 64 * TS;
    constexpr int lds_bytes = S * 2 * BM * 32 * 4;
    static_assert(lds_bytes >= 64 * (BM + 4) * 4, "epilogue staging must fit in the ring");
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_v2<TS, S, EPI, PRIO>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    dim3 grid((a.M / BM) * (a.N / BM));
    return time_us([&] { gemm_v2<TS, S, EPI, PRIO><<<grid, 256, lds_bytes>>>(a); });
}

template <int TS, int PRIO, int EPI>
static double run_v0(const Args& a) {
    constexpr int BM = 64 * TS;
    dim3 grid((a.M / BM) * (a.N / BM));
    return time_us([&] { gemm_v0<TS, PRIO, EPI><<<grid, 256>>>(a); });
}

int main(int argc, char** argv) {
    struct Shape { const char* name; int M, N, K; };
    const Shape shapes[] = {{"out-proj", 40960, 256, 256}, {"in-proj ", 40960, 768, 256}, {"ffn1    ", 40960, 2048, 256},
                            {"ffn2    ", 40960, 256, 2048}, {"dgrad768", 40960, 256, 768}, {"t-outprj", 10240, 256, 256}};
    {   // bring the clocks up before anything is timed
        Args a;
        a.M = 40960; a.N = 2048; a.K = 256; a.prio_div = 1;
        a.A = dev_rand((long)a.M * a.K, 1); a.B = dev_rand((long)a.N * a.K, 2); a.bias = dev_rand(a.N, 3);
        a.res = dev_rand((long)a.M * a.N, 4);
        float* C; CK(hipMalloc(&C, (size_t)a.M * a.N * 4)); a.C = C;
        for (int rep = 0; rep < 12; ++rep) run_v0<2, 0, 0>(a);
    }
    for (const Shape& s : shapes) {
        Args a;
        a.M = s.M; a.N = s.N; a.K = s.K; a.prio_div = 32;
        a.A = dev_rand((long)s.M * s.K, 1);
        a.B = dev_rand((long)s.N * s.K, 2);
        a.bias = dev_rand(s.N, 3);
        a.res = dev_rand((long)s.M * s.N, 4);
        float* C;
        CK(hipMalloc(&C, (size_t)s.M * s.N * 4));
        a.C = C;
        const double gf = 2.0 * s.M * s.N * s.K * 1e-9;
        printf("%s M=%d N=%d K=%d (%.2f GFLOP, %.1f us at 157.3 TF)\n", s.name, s.M, s.N, s.K, gf, gf / 157.3e3 * 1e6);
        std::vector<Variant> vs;
        auto add = [&](const char* tag, std::function<double()> f) { vs.push_back({tag, f, {}}); };
        CK(hipMemset(C, 0, (size_t)s.M * s.N * 4));
        run_v0<1, 0, 0>(a); check(a, "v0 64x64");
        CK(hipMemset(C, 0, (size_t)s.M * s.N * 4));
        run_v2<1, 2, 0>(a); check(a, "v2 64x64 S=2");
        CK(hipMemset(C, 0, (size_t)s.M * s.N * 4));
        run_v2<2, 2, 0>(a); check(a, "v2 128x128 S=2");
        add("v0 64x64   full", [&] { return run_v0<1, 0, 0>(a); });
        add("v0 128x128 full", [&] { return run_v0<2, 0, 0>(a); });
        add("v0 64x64   k loop only", [&] { return run_v0<1, 0, 2>(a); });
        add("v2 64x64   S=3", [&] { return run_v2<1, 3, 0>(a); });
        add("v2 64x64   S=2", [&] { return run_v2<1, 2, 0>(a); });
        add("v2 64x64   S=2 k loop only", [&] { return run_v2<1, 2, 2>(a); });
        add("v2 64x64   S=2 bias+relu", [&] { return run_v2<1, 2, 4>(a); });
        add("v2 64x64   S=2 full, reads up front", [&] { return run_v2<1, 2, 0, 2>(a); });
        add("v2 64x64   S=2 bias+relu, reads up front", [&] { return run_v2<1, 2, 4, 2>(a); });
        add("v2 64x64   S=2 k loop only, reads up front", [&] { return run_v2<1, 2, 2, 2>(a); });
        add("v2 64x64   S=2 prio/32", [&] { a.prio_div = 32; return run_v2<1, 2, 0, 1>(a); });
        add("v2 128x128 S=2", [&] { return run_v2<2, 2, 0>(a); });
        add("v2 128x128 S=2 k loop only", [&] { return run_v2<2, 2, 2>(a); });
        add("v2 128x128 S=2 bias+relu", [&] { return run_v2<2, 2, 4>(a); });
        add("v2 128x128 S=2 bias+relu prio/32", [&] { a.prio_div = 32; return run_v2<2, 2, 4, 1>(a); });
        if (argc > 1) {
            vs.clear();
            long long* clk; CK(hipMalloc(&clk, 4096 * 16)); 
            auto freq = [&](const char* tag, std::function<double()> f) {
                CK(hipMemset(clk, 0, 4096 * 16)); a.clk = clk; f(); CK(hipDeviceSynchronize()); a.clk = nullptr;
                std::vector<long long> h(8192); CK(hipMemcpy(h.data(), clk, 4096 * 16, hipMemcpyDeviceToHost));
                double cs = 0, ws = 0; int n = 0;
                for (int i = 0; i < 4096; ++i) if (h[2 * i + 1] > 0) { cs += h[2 * i]; ws += h[2 * i + 1]; ++n; }
                printf("   clock %-30s %d samples: shader cycles / 100 MHz ticks = %.3f (-> %.0f MHz), mean wg life %.1f us\n", tag, n, cs / ws, cs / ws * 100.0, ws / n / 100.0);
            };
            freq("64 part0", [&] { return run_probe<1, 0>(a); });
            freq("64 part1", [&] { return run_probe<1, 1>(a); });
            freq("64 part3", [&] { return run_probe<1, 3>(a); });
            freq("128 part0", [&] { return run_probe<2, 0>(a); });
            freq("128 part3", [&] { return run_probe<2, 3>(a); });
            add("probe 64  part0 mfma only", [&] { return run_probe<1, 0>(a); });
            add("probe 64  part1 +lds reads", [&] { return run_probe<1, 1>(a); });
            add("probe 64  part2 +barriers", [&] { return run_probe<1, 2>(a); });
            add("probe 64  part3 +global loads", [&] { return run_probe<1, 3>(a); });
            add("probe 128 part0 mfma only", [&] { return run_probe<2, 0>(a); });
            add("probe 128 part1 +lds reads", [&] { return run_probe<2, 1>(a); });
            add("probe 128 part2 +barriers", [&] { return run_probe<2, 2>(a); });
            add("probe 128 part3 +global loads", [&] { return run_probe<2, 3>(a); });
        }
        run_all(vs, gf);
        CK(hipFree((void*)a.A)); CK(hipFree((void*)a.B)); CK(hipFree((void*)a.bias)); CK(hipFree((void*)a.res)); CK(hipFree(C));
    }
    return 0;
}
