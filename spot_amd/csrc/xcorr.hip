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

// float4 registers per thread that hold one staged plane
int plane_vectors(int hx, int wx) { return ((hx * wx + 3 + 3) / 4 + 255) / 256; }

// LDS floats reserved for one staged plane: the plane (+3 alignment shift) plus the rows the last,
// partial strip of R output rows reads past the end (their results are discarded), 16-byte rounded.
__host__ __device__ inline int plane_floats(int hx, int wx, int hz) {
    const int ho = hx - hz + 1;
    const int rows = ((ho + R - 1) / R) * R + hz - 1;
    return ((rows > hx ? rows : hx) * wx + 3 + 3) & ~3;
}

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

// r[0..R) = correlation of the LDS plane xs (row stride wx) with the HZ x WZ template z (global,
// workgroup-uniform: the taps arrive through scalar loads) at output rows i0.., column col.
// The tap-column loop is kept rolled so only one (R+HZ-1)-deep register window is live: ~50 VGPRs,
// which lets six workgroups (24 waves) share a CU and keeps enough loads in flight to stream HBM.
// Rows past the plane are never guarded: the LDS plane is over-allocated and those outputs are dropped.
template <int HZ, int WZ, int RR = R>
__device__ __forceinline__ void corr_column(const float* xs, int wx, const float* __restrict__ z, int i0, int col,
                                            float (&r)[RR]) {
#pragma unroll
    for (int k = 0; k < RR; ++k) r[k] = 0.f;
    const float* xp = xs + i0 * wx + col;
#pragma unroll 1
    for (int v = 0; v < WZ; ++v) {
        float xc[RR + HZ - 1];
#pragma unroll
        for (int k = 0; k < RR + HZ - 1; ++k) xc[k] = xp[k * wx + v];
#pragma unroll
        for (int u = 0; u < HZ; ++u) {
            const float zv = z[u * WZ + v];
#pragma unroll
            for (int k = 0; k < RR; ++k) r[k] = fmaf(xc[k + u], zv, r[k]);
        }
    }
}

// runtime-sized template (any hz, wz)
__device__ __forceinline__ void corr_column_rt(const float* xs, int wx, const float* __restrict__ z, int hz, int wz,
                                               int i0, int col, float (&r)[R]) {
#pragma unroll
    for (int k = 0; k < R; ++k) r[k] = 0.f;
    const float* xp = xs + i0 * wx + col;
    for (int u = 0; u < hz; ++u)
        for (int v = 0; v < wz; ++v) {
            const float zv = z[u * wz + v];
#pragma unroll
            for (int k = 0; k < R; ++k) r[k] = fmaf(xp[(k + u) * wx + v], zv, r[k]);
        }
}

struct Branch {
    const float* x;
    const float* z;
    int hx, wx, hz, wz;
};

// Register-staged variant of stage_plane for the persistent loop below: fetch() issues the plane's 16-byte
// loads into NV float4 registers per thread (the loads stay in flight while the previous plane is being
// correlated), commit() drops them into LDS.  Vector k of the 16-byte-aligned stream that contains the plane
// lands at dst + 4k, i.e. plane element e at dst[shift + e].  Vectors that would cross the ends of the whole
// tensor are read element-wise.
template <int NV>
struct PlaneRegs {
    float4 v[NV];
    int shift, nvec;
    __device__ __forceinline__ void fetch(const float* __restrict__ tensor, long total, long first, int n, int tid) {
        const float* g = tensor + first;
        shift = (int)(((uintptr_t)g >> 2) & 3);
        nvec = (shift + n + 3) >> 2;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int k = tid + 256 * i;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < nvec) {
                const long e0 = first - shift + 4L * k;      // tensor-relative index of the vector's first float
                if (e0 >= 0 && e0 + 3 < total) {
                    v[i] = *reinterpret_cast<const float4*>(tensor + e0);
                } else {
                    if (e0 >= 0 && e0 < total) v[i].x = tensor[e0];
                    if (e0 + 1 >= 0 && e0 + 1 < total) v[i].y = tensor[e0 + 1];
                    if (e0 + 2 >= 0 && e0 + 2 < total) v[i].z = tensor[e0 + 2];
                    if (e0 + 3 >= 0 && e0 + 3 < total) v[i].w = tensor[e0 + 3];
                }
            }
        }
    }
    __device__ __forceinline__ void commit(float* dst, int tid) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int k = tid + 256 * i;
            if (k < nvec) *reinterpret_cast<float4*>(dst + 4 * k) = v[i];
        }
    }
};

// KIND 0: runtime template sizes; KIND 5: Ocean's (5x5, 3x5, 5x3) triple.
// Persistent workgroups: each walks planes blockIdx.x, blockIdx.x + gridDim.x, ...; the next plane's loads are
// issued before the current plane is correlated, so every CU keeps HBM requests in flight all the time instead
// of only during the load phase of whichever workgroups happen to be starting.
// workgroups per CU the register budget is sized for: the staged planes take 4 * NB * NV registers per thread
constexpr int wg_per_cu(int nb, int nv) { return nb * nv <= 2 ? 8 : (nb * nv <= 6 ? 4 : 2); }

template <int NB, int KIND, int NV>
__global__ __launch_bounds__(256, wg_per_cu(NB, NV)) void groupdw_kernel(Branch b0, Branch b1, Branch b2, const float* __restrict__ weight,
                                                      float* __restrict__ out, int Ho, int Wo, int planes) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const Branch br[3] = {b0, b1, b2};
    int offs[3];
    {
        int off = 0;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            offs[i] = off;
            off += plane_floats(br[i].hx, br[i].wx, br[i].hz);
        }
    }
    float sw[3] = {1.f, 0.f, 0.f};
    if (NB == 3) {
        const float w0 = weight[0], w1 = weight[1], w2 = weight[2];
        const float m = fmaxf(w0, fmaxf(w1, w2));
        const float e0 = expf(w0 - m), e1 = expf(w1 - m), e2 = expf(w2 - m);
        const float s = e0 + e1 + e2;
        sw[0] = e0 / s; sw[1] = e1 / s; sw[2] = e2 / s;
    }
    PlaneRegs<NV> regs[NB];
    int p = blockIdx.x;
    if (p < planes) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int n = br[i].hx * br[i].wx;
            regs[i].fetch(br[i].x, (long)planes * n, (long)p * n, n, tid);
        }
    }
    const int nstrip = (Ho + R - 1) / R;
    for (; p < planes; p += gridDim.x) {
        const float* xs[3];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            regs[i].commit(smem + offs[i], tid);
            xs[i] = smem + offs[i] + regs[i].shift;
        }
        __syncthreads();
        const int pn = p + gridDim.x;
        if (pn < planes) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int n = br[i].hx * br[i].wx;
                regs[i].fetch(br[i].x, (long)planes * n, (long)pn * n, n, tid);
            }
        }
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
                    if (i == 0) corr_column<5, 5>(xs[0], br[0].wx, z, i0, col, r);
                    if (i == 1) corr_column<3, 5>(xs[1], br[1].wx, z, i0, col, r);
                    if (i == 2) corr_column<5, 3>(xs[2], br[2].wx, z, i0, col, r);
                } else {
                    corr_column_rt(xs[i], br[i].wx, z, br[i].hz, br[i].wz, i0, col, r);
                }
#pragma unroll
                for (int k = 0; k < R; ++k) acc[k] = NB == 1 ? r[k] : __fadd_rn(acc[k], __fmul_rn(sw[i], r[k]));
            }
#pragma unroll
            for (int k = 0; k < R; ++k)
                if (i0 + k < Ho) op[(i0 + k) * Wo + col] = acc[k];
        }
        __syncthreads();       // everyone is done with this plane's LDS copy before the next commit
    }
}


template <int NB, int KIND>
int launch_fwd(const Branch* b, const float* weight, float* out, int planes, size_t lds, hipStream_t stream) {
    const int Ho = b[0].hx - b[0].hz + 1, Wo = b[0].wx - b[0].wz + 1;
    int nv = 0;
    for (int i = 0; i < NB; ++i) nv = plane_vectors(b[i].hx, b[i].wx) > nv ? plane_vectors(b[i].hx, b[i].wx) : nv;
    // as many workgroups as fit the chip at once (LDS- and register-limited), each walking several planes
    const int lds_cap = (int)((160 * 1024) / (lds > 0 ? lds : 1));
#define SPOT_XC_LAUNCH(NVV)                                                                                      \
    do {                                                                                                         \
        int per_cu = wg_per_cu(NB, NVV) < lds_cap ? wg_per_cu(NB, NVV) : lds_cap;                                \
        per_cu = per_cu < 1 ? 1 : per_cu;                                                                        \
        const int grid = planes < 256 * per_cu ? planes : 256 * per_cu;                                          \
        groupdw_kernel<NB, KIND, NVV><<<grid, 256, lds, stream>>>(b[0], b[1], b[2], weight, out, Ho, Wo, planes); \
    } while (0)
    if (nv <= 1) SPOT_XC_LAUNCH(1);
    else if (nv <= 2) SPOT_XC_LAUNCH(2);
    else if (nv <= 4) SPOT_XC_LAUNCH(4);
    else if (nv <= 8) SPOT_XC_LAUNCH(8);
    else SPOT_XC_LAUNCH(16);
#undef SPOT_XC_LAUNCH
    return 0;
}



// ---------------------------------------------------------------------------
// Register-tile forward for a single 5x5 template (xcorr_depthwise); written for the (5x5, 3x5, 5x3) family.
// The column walk above reads every input value WZ times from LDS (one 4-byte read per tap column), which
// makes LDS issue, not HBM, the limit.  Here a thread owns a 4-row x 2-column output patch and loads its
// (HZ+3) x (WZ+1) input patch once with 8-byte reads: 2.5x fewer LDS instructions per output.  8-byte reads
// need even column starts and an even row stride, so the planes are re-pitched while they are committed to
// LDS (stride wx rounded up to even) instead of being copied flat.
// ---------------------------------------------------------------------------
constexpr int TR = 4, TC = 2;

__host__ __device__ inline int tile_stride(int wx) { return (wx + 1) & ~1; }
// rows the last strip of TR output rows reads, at least the plane's own
__host__ __device__ inline int tile_rows(int hx, int hz) {
    const int ho = hx - hz + 1;
    const int rows = ((ho + TR - 1) / TR) * TR + hz - 1;
    return rows > hx ? rows : hx;
}

template <int NV>
__device__ __forceinline__ void commit_pitched(const PlaneRegs<NV>& regs, float* dst, int n, int wx, int S, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k = tid + 256 * i;
        if (k < regs.nvec) {
            const int e0 = 4 * k - regs.shift;        // plane-relative index of the vector's first float
            const float vals[4] = {regs.v[i].x, regs.v[i].y, regs.v[i].z, regs.v[i].w};
            const int ec = e0 < 0 ? 0 : e0;
            int r = ec / wx, c = ec - r * wx;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = e0 + j;
                if (e >= 0 && e < n) {
                    dst[r * S + c] = vals[j];
                    if (++c == wx) {
                        c = 0;
                        ++r;
                    }
                }
            }
        }
    }
}

template <int HZ, int WZ>
__device__ __forceinline__ void tile_corr(const float* xs, int S, const float* __restrict__ z, int i0, int j0,
                                          float (&acc)[TR][TC]) {
    float xp[HZ + TR - 1][WZ + TC - 1];
    const float* base = xs + i0 * S + j0;
#pragma unroll
    for (int r = 0; r < HZ + TR - 1; ++r)
#pragma unroll
        for (int c = 0; c < WZ + TC - 1; c += 2) {
            const float2 v = *reinterpret_cast<const float2*>(base + r * S + c);
            xp[r][c] = v.x;
            xp[r][c + 1] = v.y;
        }
#pragma unroll
    for (int r = 0; r < TR; ++r)
#pragma unroll
        for (int c = 0; c < TC; ++c) acc[r][c] = 0.f;
#pragma unroll
    for (int u = 0; u < HZ; ++u)
#pragma unroll
        for (int v = 0; v < WZ; ++v) {
            const float zv = z[u * WZ + v];
#pragma unroll
            for (int r = 0; r < TR; ++r)
#pragma unroll
                for (int c = 0; c < TC; ++c) acc[r][c] = fmaf(xp[r + u][c + v], zv, acc[r][c]);
        }
}

template <int NB, int NV>
__global__ __launch_bounds__(256, NB == 1 ? 6 : 4) void groupdw_tile_kernel(Branch b0, Branch b1, Branch b2,
                                                              const float* __restrict__ weight, float* __restrict__ out,
                                                              int Ho, int Wo, int planes) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const Branch br[3] = {b0, b1, b2};
    int offs[3], S[3];
    {
        int off = 0;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            offs[i] = off;
            S[i] = tile_stride(br[i].wx);
            off += (tile_rows(br[i].hx, br[i].hz) * S[i] + 3) & ~3;
        }
    }
    float sw[3] = {1.f, 0.f, 0.f};
    if (NB == 3) {
        const float w0 = weight[0], w1 = weight[1], w2 = weight[2];
        const float m = fmaxf(w0, fmaxf(w1, w2));
        const float e0 = expf(w0 - m), e1 = expf(w1 - m), e2 = expf(w2 - m);
        const float s = e0 + e1 + e2;
        sw[0] = e0 / s; sw[1] = e1 / s; sw[2] = e2 / s;
    }
    // PERSIST: the workgroup walks several planes and keeps the next plane's loads in flight in registers while
    // it correlates the current one.
    constexpr bool PERSIST = NB == 1;
    PlaneRegs<NV> regs[PERSIST ? NB : 1];
    int p = blockIdx.x;
    if (PERSIST && p < planes) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int n = br[i].hx * br[i].wx;
            regs[i].fetch(br[i].x, (long)planes * n, (long)p * n, n, tid);
        }
    }
    const int nsr = (Ho + TR - 1) / TR, nsc = (Wo + TC - 1) / TC;
    for (; p < planes; p += gridDim.x) {
        if (PERSIST) {
#pragma unroll
            for (int i = 0; i < NB; ++i)
                commit_pitched(regs[i], smem + offs[i], br[i].hx * br[i].wx, br[i].wx, S[i], tid);
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int n = br[i].hx * br[i].wx;
                PlaneRegs<NV> once;
                once.fetch(br[i].x, (long)planes * n, (long)p * n, n, tid);
                commit_pitched(once, smem + offs[i], n, br[i].wx, S[i], tid);
            }
        }
        __syncthreads();
        const int pn = p + gridDim.x;
        if (PERSIST && pn < planes) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int n = br[i].hx * br[i].wx;
                regs[i].fetch(br[i].x, (long)planes * n, (long)pn * n, n, tid);
            }
        }
        float* op = out + (size_t)p * Ho * Wo;
        for (int item = tid; item < nsr * nsc; item += 256) {
            const int sr = item / nsc, sc = item - sr * nsc;
            const int i0 = sr * TR, j0 = sc * TC;
            float acc[TR][TC], r[TR][TC];
#pragma unroll
            for (int a = 0; a < TR; ++a)
#pragma unroll
                for (int c = 0; c < TC; ++c) acc[a][c] = 0.f;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const float* z = br[i].z + (size_t)p * br[i].hz * br[i].wz;
                if (i == 0) tile_corr<5, 5>(smem + offs[0], S[0], z, i0, j0, r);
                if (i == 1) tile_corr<3, 5>(smem + offs[1], S[1], z, i0, j0, r);
                if (i == 2) tile_corr<5, 3>(smem + offs[2], S[2], z, i0, j0, r);
#pragma unroll
                for (int a = 0; a < TR; ++a)
#pragma unroll
                    for (int c = 0; c < TC; ++c) acc[a][c] = NB == 1 ? r[a][c] : __fadd_rn(acc[a][c], __fmul_rn(sw[i], r[a][c]));
                // one branch's input patch at a time: without a fence for memory operations the scheduler hoists
                // all three patches' LDS loads (116 registers) ahead of the arithmetic
                if (NB > 1) asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int a = 0; a < TR; ++a)
#pragma unroll
                for (int c = 0; c < TC; ++c)
                    if (i0 + a < Ho && j0 + c < Wo) op[(i0 + a) * Wo + j0 + c] = acc[a][c];
        }
        __syncthreads();
    }
}

template <int NB>
bool launch_tile(const Branch* b, const float* weight, float* out, int planes, hipStream_t stream) {
    const int Ho = b[0].hx - b[0].hz + 1, Wo = b[0].wx - b[0].wz + 1;
    size_t f = 0;
    int nv = 0;
    for (int i = 0; i < NB; ++i) {
        f += ((size_t)tile_rows(b[i].hx, b[i].hz) * tile_stride(b[i].wx) + 3) & ~(size_t)3;
        nv = plane_vectors(b[i].hx, b[i].wx) > nv ? plane_vectors(b[i].hx, b[i].wx) : nv;
    }
    // columns read past the plane: j0 + WZ + TC - 2 <= stride - 1 must hold for the last column patch
    for (int i = 0; i < NB; ++i)
        if (((Wo + TC - 1) / TC) * TC + b[i].wz - 1 > tile_stride(b[i].wx)) return false;
    const size_t lds = f * sizeof(float);
    if (lds > 40 * 1024 || nv > 2) return false;
    const int lds_cap = (int)((160 * 1024) / lds), reg_cap = NB == 1 ? 6 : 4;
    const int per_cu = lds_cap < reg_cap ? (lds_cap < 1 ? 1 : lds_cap) : reg_cap;
    const int grid = (NB > 1 || planes < 256 * per_cu) ? planes : 256 * per_cu;     // the triple: one plane per workgroup
    if (nv <= 1) groupdw_tile_kernel<NB, 1><<<grid, 256, lds, stream>>>(b[0], b[1], b[2], weight, out, Ho, Wo, planes);
    else groupdw_tile_kernel<NB, 2><<<grid, 256, lds, stream>>>(b[0], b[1], b[2], weight, out, Ho, Wo, planes);
    return true;
}

// ---------------------------------------------------------------------------
// Row-strip forward for the (5x5, 3x5, 5x3) triple (NB = 3) or a single 5x5 template (NB = 1).
//
// What the counters say about the column walk above at 384^2 / B = 128: HBM traffic is the algorithmic 1.0 GB, but
// every wave issues ~720 vector-ALU and ~150 4-byte LDS instructions per plane, a third of the LDS cycles are bank
// conflicts, and the vector ALU alone is busy for half of the kernel's time.  This kernel cuts both:
//   * a work item is 8 consecutive outputs of one row.  Its inputs are row segments of 12 floats, read as three
//     16-byte LDS loads (256 B/clk instead of 128 B/clk for 4-byte reads).  The planes are re-pitched on the way into
//     LDS (pitch = 4 * odd number of floats) and consecutive lanes take consecutive rows, so the 16 lanes the LDS
//     serves per cycle hit 16 distinct bank quads: no conflicts.
//   * the products are packed: v_pk_fma_f32 on register pairs as they come out of the loads.  Taps at an even column
//     offset v update the output pairs (0,1)(2,3)(4,5)(6,7) from input pairs (v,v+1)(v+2,v+3)...; taps at an odd v update
//     the shifted pairs (-1,0)(1,2)...(7,8) from input pairs (v-1,v)(v+1,v+2)...: both are aligned pairs, no moves.
//     The two partial sums are added once per branch.  55 taps cost 241 packed instead of 440 scalar multiply-adds.
//   * re-pitching costs no index arithmetic: a thread loads the 16 bytes at (row r, quad q) of the plane in global memory
//     (4-byte aligned; the last quad of a row runs into the next row, whose values land in the pitch padding) and
//     stores them with one 16-byte LDS write; its (r, q) slots are fixed for the whole launch.
// Persistent workgroups with the next plane's loads in flight in registers, as above.
// ---------------------------------------------------------------------------
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef int i4v __attribute__((ext_vector_type(4)));

// acc += x * (tap, tap): the tap is one half of a scalar register pair that holds two consecutive taps, picked by the
// instruction's half selectors (written as assembly because the compiler otherwise duplicates every tap into a pair of
// its own: 110 scalar registers for the 55 taps, spilled)
__device__ __forceinline__ void pk_fma_tap_lo(f2v& acc, f2v x, f2v taps) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(x), "s"(taps));
}
__device__ __forceinline__ void pk_fma_tap_hi(f2v& acc, f2v x, f2v taps) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(x), "s"(taps));
}

// acc += (one half of a vector register pair, broadcast) * x
__device__ __forceinline__ void pk_fma_g_lo(f2v& acc, f2v gpair, f2v x) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(gpair), "v"(x));
}
__device__ __forceinline__ void pk_fma_g_hi(f2v& acc, f2v gpair, f2v x) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(acc) : "v"(gpair), "v"(x));
}

// acc = (one half of a vector register pair, broadcast) * x: the first product of a sum, which then needs no zero fill
__device__ __forceinline__ void pk_mul_g_lo(f2v& acc, f2v gpair, f2v x) {
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(acc) : "v"(gpair), "v"(x));
}
__device__ __forceinline__ void pk_mul_g_hi(f2v& acc, f2v gpair, f2v x) {
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(acc) : "v"(gpair), "v"(x));
}

// the NT taps of one plane and branch as pairs (workgroup-uniform: scalar loads)
template <int NT>
__device__ __forceinline__ void load_taps(const float* __restrict__ z, f2v (&zt)[(NT + 1) / 2]) {
#pragma unroll
    for (int m = 0; m < NT / 2; ++m) zt[m] = f2v{z[2 * m], z[2 * m + 1]};
    if (NT & 1) zt[NT / 2] = f2v{z[NT - 1], 0.f};
}

// r[0..8) = the 8 outputs at (row, j0 .. j0+7) of the correlation of the pitched LDS plane xs with the HZ x WZ taps:
//   r[k] = sum_{u,v} xs[row + u][j0 + k + V0 + v] * tap(u, v),   tap(u, v) = taps[u WZ + v], or taps[HZ WZ - 1 - (u WZ + v)] with REV
// (V0 > 0 and REV serve the input gradient: a correlation of the zero-padded output gradient with the flipped taps, whose
// window starts V0 columns into the 16-byte aligned segment).  V0 + WZ <= 5.
// VT: the tap pairs sit in vector registers (every lane holds the same values) instead of scalar ones.
template <int HZ, int WZ, int PITCH, int V0 = 0, bool REV = false, bool VT = false>
__device__ __forceinline__ void row_strip(const float* xs, const f2v (&zt)[(HZ * WZ + 1) / 2], int row, int j0, float (&r)[8]) {
    f2v A[4], B[5];
#pragma unroll
    for (int c = 0; c < 4; ++c) A[c] = f2v{0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 5; ++c) B[c] = f2v{0.f, 0.f};
    const float* base = xs + row * PITCH + j0;
#pragma unroll
    for (int u = 0; u < HZ; ++u) {
        f4v x[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) x[q] = *reinterpret_cast<const f4v*>(base + u * PITCH + 4 * q);
        const f2v xp[6] = {x[0].xy, x[0].zw, x[1].xy, x[1].zw, x[2].xy, x[2].zw};
#pragma unroll
        for (int v = 0; v < WZ; ++v) {
            const int t = REV ? HZ * WZ - 1 - (u * WZ + v) : u * WZ + v;
            const int vv = v + V0;
            if ((vv & 1) == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (VT) {
                        if (t & 1) pk_fma_g_hi(A[c], zt[t >> 1], xp[c + vv / 2]);
                        else pk_fma_g_lo(A[c], zt[t >> 1], xp[c + vv / 2]);
                    } else {
                        if (t & 1) pk_fma_tap_hi(A[c], xp[c + vv / 2], zt[t >> 1]);
                        else pk_fma_tap_lo(A[c], xp[c + vv / 2], zt[t >> 1]);
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < 5; ++c) {
                    if (VT) {
                        if (t & 1) pk_fma_g_hi(B[c], zt[t >> 1], xp[c + (vv - 1) / 2]);
                        else pk_fma_g_lo(B[c], zt[t >> 1], xp[c + (vv - 1) / 2]);
                    } else {
                        if (t & 1) pk_fma_tap_hi(B[c], xp[c + (vv - 1) / 2], zt[t >> 1]);
                        else pk_fma_tap_lo(B[c], xp[c + (vv - 1) / 2], zt[t >> 1]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        r[2 * c] = A[c].x + B[c].y;
        r[2 * c + 1] = A[c].y + B[c + 1].x;
    }
}

constexpr int ROW_NV = 3;        // 16-byte vectors per thread and branch that hold a plane in flight (<= 768 quads)

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global access of
// the wave: here that would be the next plane's loads and this plane's stores, i.e. it would drain the memory pipeline
// twice per plane.  The kernel's global reads and writes never alias, so only the LDS image needs the barrier.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int NB, int PITCH>
__global__ __launch_bounds__(256, 4) void groupdw_row_kernel(Branch b0, Branch b1, Branch b2, const float* __restrict__ weight,
                                                             float* __restrict__ out, int Ho, int Wo, int planes) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const Branch br[3] = {b0, b1, b2};
    int offs[3], ostage;
    {
        int off = 0;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            offs[i] = off;
            off += br[i].hx * PITCH;
        }
        ostage = off;                  // the finished map, rows at pitch OP, until the next iteration stores it
    }
    constexpr int OP = PITCH;          // 4 * odd: the rows-fastest stage writes hit distinct bank quads
    float sw[3] = {1.f, 0.f, 0.f};
    if (NB == 3) {
        const float w0 = weight[0], w1 = weight[1], w2 = weight[2];
        const float m = fmaxf(w0, fmaxf(w1, w2));
        const float e0 = expf(w0 - m), e1 = expf(w1 - m), e2 = expf(w2 - m);
        const float s = e0 + e1 + e2;
        sw[0] = e0 / s; sw[1] = e1 / s; sw[2] = e2 / s;
    }
    // Staging map.  A thread owns quad column q = t % Q of rows r0 + k RPT (k = 0 .. passes-1), with Q = ceil(wx / 4)
    // quads per row, RPT = 256 / Q rows per pass and r0 = t / Q; threads past RPT * Q repeat the last one's work and
    // the last pass holds its row inside the plane (same data to the same LDS slot), so nothing branches per lane.
    // Only the first and the last pass's offsets live in registers: the passes between differ by workgroup-uniform steps.
    int passes[NB], gstep[NB], lstep[NB];
    unsigned g_first[NB], g_last[NB];
    int l_first[NB], l_last[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int Q = (br[i].wx + 3) >> 2, RPT = 256 / Q;
        passes[i] = (br[i].hx + RPT - 1) / RPT;
        const int t = min(tid, RPT * Q - 1), r0 = t / Q, q = t - r0 * Q;
        const int rl = min(r0 + (passes[i] - 1) * RPT, br[i].hx - 1);
        g_first[i] = (unsigned)(r0 * br[i].wx + 4 * q) * 4u;
        g_last[i] = (unsigned)(rl * br[i].wx + 4 * q) * 4u;
        l_first[i] = offs[i] + r0 * PITCH + 4 * q;
        l_last[i] = offs[i] + rl * PITCH + 4 * q;
        gstep[i] = RPT * br[i].wx * 4;
        lstep[i] = RPT * PITCH;
    }
    f4v regs[NB][ROW_NV];
    // plane p's quads into registers, by buffer loads: they take any 4-byte alignment, and the last quad of the last
    // plane's last row, which ends up to 3 floats past the tensor, reads zeros there (the descriptor bounds the range).
    __amdgpu_buffer_rsrc_t rx[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
        rx[i] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(br[i].x), 0, planes * br[i].hx * br[i].wx * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, planes * Ho * Wo * 4, 0x00020000);
    auto fetch = [&](int p) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int plane_bytes = p * (br[i].hx * br[i].wx * 4);
#pragma unroll
            for (int k = 0; k < ROW_NV; ++k) {
                if (k >= passes[i]) continue;
                const bool last = k == passes[i] - 1;
                regs[i][k] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(
                    rx[i], (int)(last ? g_last[i] : g_first[i]), plane_bytes + (last ? 0 : k * gstep[i]), 0));
            }
        }
    };
    // The finished map of plane q leaves through LDS so that a wave's lanes run along output rows: a row's Wo floats
    // go out as contiguous 16-byte pieces (a few whole-line write requests per row) instead of each work item's two
    // pieces 4 Wo bytes apart from its neighbours' (one partial-line request per lane).  It is stored one iteration
    // late, between the barriers that fence the staging image anyway.
    const int OQ = (Wo + 3) >> 2;                              // pieces per row; the last one may be partial
    auto store_out = [&](int q) {
        const int plane_bytes = q * (Ho * Wo * 4);
        for (int idx = tid; idx < Ho * OQ; idx += 256) {
            const int row = idx / OQ, c4 = (idx - row * OQ) * 4;
            const f4v v = *reinterpret_cast<const f4v*>(smem + ostage + row * OP + c4);
            const int obytes = (row * Wo + c4) * 4;
            if (c4 + 4 <= Wo) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4v, v), ro, obytes, plane_bytes, 0);
            } else {
                float* orow = out + (size_t)q * Ho * Wo + row * Wo + c4;
                orow[0] = v.x;
                if (c4 + 1 < Wo) orow[1] = v.y;
                if (c4 + 2 < Wo) orow[2] = v.z;
            }
        }
    };
    int p = blockIdx.x;
    if (p < planes) fetch(p);
    const int nstrip = (Wo + 7) >> 3;
    int prev = -1;
    for (; p < planes; p += gridDim.x) {
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int k = 0; k < ROW_NV; ++k) {
                if (k >= passes[i]) continue;
                const bool last = k == passes[i] - 1;
                *reinterpret_cast<f4v*>(smem + (last ? l_last[i] : l_first[i] + k * lstep[i])) = regs[i][k];
            }
        if (prev >= 0) store_out(prev);
        prev = p;
        lds_barrier();
        const int pn = p + gridDim.x;
        if (pn < planes) fetch(pn);
        // Items in chunks of 256 (one chunk for the 41x41 and 25x25 maps).  The taps are workgroup-uniform and come
        // through the scalar unit; each branch's are loaded right before its pass, in uniform control flow, so that
        // only one branch's taps occupy scalar registers at a time.
        const int nitems = nstrip * Ho;
        for (int base = 0; base < nitems; base += 256) {
            const int item = base + tid;
            const bool active = item < nitems;
            const int strip = item / Ho, row = item - strip * Ho, j0 = strip * 8;      // rows fastest: conflict-free LDS reads
            float acc[8], r[8];
            {
                f2v zt[13];
                load_taps<25>(br[0].z + (size_t)p * 25, zt);
                if (active) {
                    row_strip<5, 5, PITCH>(smem + offs[0], zt, row, j0, r);
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] = NB == 1 ? r[k] : sw[0] * r[k];
                }
            }
            if (NB == 3) {
                f2v zt[8];
                load_taps<15>(br[1].z + (size_t)p * 15, zt);
                if (active) {
                    row_strip<3, 5, PITCH>(smem + offs[1], zt, row, j0, r);
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] = acc[k] + sw[1] * r[k];
                }
                load_taps<15>(br[2].z + (size_t)p * 15, zt);
                if (active) {
                    row_strip<5, 3, PITCH>(smem + offs[2], zt, row, j0, r);
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] = acc[k] + sw[2] * r[k];
                }
            }
            if (active) {
                float* st = smem + ostage + row * OP + j0;
                *reinterpret_cast<f4v*>(st) = f4v{acc[0], acc[1], acc[2], acc[3]};
                *reinterpret_cast<f4v*>(st + 4) = f4v{acc[4], acc[5], acc[6], acc[7]};
            }
        }
        lds_barrier();         // everyone is done with this plane's LDS image before the next commit
    }
    if (prev >= 0) store_out(prev);
}

// Shapes the row-strip kernel takes: 5x5 base template; the widest row (+ the 12-float segment of the last strip) fits the
// pitch; a plane's quads fit ROW_NV vectors per thread.  Returns false when the shape is left to the kernels above.
template <int NB>
bool launch_row(const Branch* b, const float* weight, float* out, int planes, hipStream_t stream) {
    const int Ho = b[0].hx - b[0].hz + 1, Wo = b[0].wx - b[0].wz + 1;
    if (b[0].hz != 5 || b[0].wz != 5) return false;
    int pitch = 0, rows = 0;
    for (int i = 0; i < NB; ++i) {
        const int need = ((Wo + 7) / 8 - 1) * 8 + 12;                  // last strip's segment end
        const int wpad = ((b[i].wx + 3) / 4) * 4;
        const int w = need > wpad ? need : wpad;
        pitch = w > pitch ? w : pitch;
        rows += b[i].hx;
        const int Q = (b[i].wx + 3) / 4;
        if (Q > 256 || (b[i].hx + 256 / Q - 1) / (256 / Q) > ROW_NV) return false;
    }
    const int P = pitch <= 36 ? 36 : (pitch <= 52 ? 52 : 0);           // 4 * odd: rows 0..15 fall on distinct bank quads
    if (P == 0) return false;
    for (int i = 0; i < NB; ++i)                                        // 32-bit byte offsets inside each tensor
        if ((long)planes * b[i].hx * b[i].wx * 4 >= (1L << 31)) return false;
    const size_t lds = ((size_t)rows * P + (size_t)Ho * P) * sizeof(float);
    if (lds > 40 * 1024) return false;
    const int lds_cap = (int)((160 * 1024) / lds);
    const int reg_cap = NB == 1 ? 8 : 4;                               // 64 / 121 registers per lane
    const int per_cu = lds_cap < reg_cap ? (lds_cap < 1 ? 1 : lds_cap) : reg_cap;
    const int grid = planes < 256 * per_cu ? planes : 256 * per_cu;
    if (P == 36) groupdw_row_kernel<NB, 36><<<grid, 256, lds, stream>>>(b[0], b[1], b[2], weight, out, Ho, Wo, planes);
    else groupdw_row_kernel<NB, 52><<<grid, 256, lds, stream>>>(b[0], b[1], b[2], weight, out, Ho, Wo, planes);
    return true;
}

size_t lds_bytes(const Branch* b, int nb) {
    size_t f = 0;
    for (int i = 0; i < nb; ++i) f += plane_floats(b[i].hx, b[i].wx, b[i].hz);
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
    SPOT_CHECK_ARG(((uintptr_t)x & 15) == 0, "xcorr_depthwise: x must be 16-byte aligned");
    if (!launch_row<1>(b, nullptr, out, planes, stream) && !(hz == 5 && wz == 5 && launch_tile<1>(b, nullptr, out, planes, stream)))
        launch_fwd<1, 0>(b, nullptr, out, planes, lds, stream);
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
    SPOT_CHECK_ARG((((uintptr_t)x11 | (uintptr_t)x12 | (uintptr_t)x21) & 15) == 0, "groupdw: search planes must be 16-byte aligned");
    // (the register-tile kernel is not used for the triple: three input patches plus the weighted sum do not fit
    //  128 registers without spilling, and at 168 registers it loses to the column walk)
    if (launch_row<3>(b, weight3, out, planes, stream)) {
    } else if (hz == 5 && wz == 5) launch_fwd<3, 5>(b, weight3, out, planes, lds, stream);
    else launch_fwd<3, 0>(b, weight3, out, planes, lds, stream);
    SPOT_CHECK_LAUNCH("groupdw");
    return SPOT_OK;
}

// ===========================================================================
// Backward of xcorr_depthwise / GroupDW.  With g = dL/dout [planes, Ho, Wo] and
// s_i = softmax(weight)_i:
//   dx_i[a,b] = s_i * sum_{u,v} g[a-u, b-v] z_i[u,v]     (full correlation = valid correlation of the
//                                                          zero-padded g with the flipped template)
//   dz_i[u,v] = s_i * sum_{i,j} g[i,j] x_i[i+u, j+v]
//   ds_i      = sum over planes of sum_{u,v} z_i[u,v] * (dz_i[u,v] / s_i)   (= <g, xcorr(x_i, z_i)>)
// One workgroup per plane: g is staged zero-padded in LDS, the x planes beside it.  dx reuses the
// forward's register-window column walk; each dz tap is reduced by one wave.  ds is written as a
// per-plane partial [planes, 3] (summed by the caller) to keep the result order-independent.
// ===========================================================================
namespace {

constexpr int GPAD_MAX = 8;   // largest template extent - 1 the padded-g staging supports

__global__ __launch_bounds__(256) void groupdw_bwd_kernel(Branch b0, Branch b1, Branch b2, int nb,
                                                          const float* __restrict__ weight,
                                                          const float* __restrict__ g, int Ho, int Wo,
                                                          float* __restrict__ dx0, float* __restrict__ dx1,
                                                          float* __restrict__ dx2, float* __restrict__ dz0,
                                                          float* __restrict__ dz1, float* __restrict__ dz2,
                                                          float* __restrict__ dsw_partial, int pad, int gp_rows,
                                                          int gp_cols) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Branch br[3] = {b0, b1, b2};
    float* dxs[3] = {dx0, dx1, dx2};
    float* dzs[3] = {dz0, dz1, dz2};
    // ---- stage: zero-padded g, then the x planes
    float* gp = smem;
    for (int i = tid; i < gp_rows * gp_cols; i += 256) gp[i] = 0.f;
    __syncthreads();
    const float* gsrc = g + (size_t)p * Ho * Wo;
    for (int i = tid; i < Ho * Wo; i += 256) {
        const int r = i / Wo, c = i - r * Wo;
        gp[(r + pad) * gp_cols + c + pad] = gsrc[i];
    }
    int off = (gp_rows * gp_cols + 3) & ~3;
    float* xs[3];
    for (int i = 0; i < nb; ++i) {
        const int n = br[i].hx * br[i].wx;
        const int sh = stage_plane(br[i].x + (size_t)p * n, n, smem + off, tid);
        xs[i] = smem + off + sh;
        off += (n + 6) & ~3;
    }
    float sw[3] = {1.f, 0.f, 0.f};
    if (nb == 3) {
        const float w0 = weight[0], w1 = weight[1], w2 = weight[2];
        const float m = fmaxf(w0, fmaxf(w1, w2));
        const float e0 = expf(w0 - m), e1 = expf(w1 - m), e2 = expf(w2 - m);
        const float s = e0 + e1 + e2;
        sw[0] = e0 / s; sw[1] = e1 / s; sw[2] = e2 / s;
    }
    __syncthreads();
    for (int i = 0; i < nb; ++i) {
        const int hx = br[i].hx, wx = br[i].wx, hz = br[i].hz, wz = br[i].wz;
        const float* z = br[i].z + (size_t)p * hz * wz;
        // ---- dx_i: R-row register window down one column of the padded g
        if (dxs[i]) {
            float* op = dxs[i] + (size_t)p * hx * wx;
            const int nstrip = (hx + R - 1) / R;
            const int o0 = pad - (hz - 1), o1 = pad - (wz - 1);     // origin of this branch inside the padding
            for (int item = tid; item < nstrip * wx; item += 256) {
                const int strip = item / wx, col = item - strip * wx, a0 = strip * R;
                float r[R];
#pragma unroll
                for (int k = 0; k < R; ++k) r[k] = 0.f;
                for (int u = 0; u < hz; ++u)
                    for (int v = 0; v < wz; ++v) {
                        const float zv = z[(hz - 1 - u) * wz + (wz - 1 - v)];
#pragma unroll
                        for (int k = 0; k < R; ++k) r[k] = fmaf(gp[(a0 + k + u + o0) * gp_cols + col + v + o1], zv, r[k]);
                    }
#pragma unroll
                for (int k = 0; k < R; ++k)
                    if (a0 + k < hx) op[(a0 + k) * wx + col] = sw[i] * r[k];
            }
        }
        // ---- dz_i (and its contribution to ds_i): one wave per tap
        float ds_acc = 0.f;
        for (int tap = wave; tap < hz * wz; tap += 4) {
            const int u = tap / wz, v = tap - u * wz;
            float acc = 0.f;
            for (int q = lane; q < Ho * Wo; q += 64) {
                const int r = q / Wo, c = q - r * Wo;
                acc = fmaf(gp[(r + pad) * gp_cols + c + pad], xs[i][(r + u) * wx + c + v], acc);
            }
            acc = wave_sum(acc);
            if (lane == 0) {
                if (dzs[i]) dzs[i][(size_t)p * hz * wz + tap] = sw[i] * acc;
                ds_acc = fmaf(z[tap], acc, ds_acc);
            }
        }
        if (dsw_partial && nb == 3) {
            __shared__ float ds_red[4];
            if (lane == 0) ds_red[wave] = ds_acc;
            __syncthreads();
            if (tid == 0) dsw_partial[(size_t)p * 3 + i] = (ds_red[0] + ds_red[1]) + (ds_red[2] + ds_red[3]);
            __syncthreads();
        }
    }
}


// ---------------------------------------------------------------------------
// Fast backward for the (5x5, 3x5, 5x3) template triple (NB = 3) or a single 5x5 template (NB = 1).
//   dx: the forward's register-window walk over the zero-padded g with the reversed taps (9-row strips, so the
//       45-row planes take one pass of the 256 threads);
//   dz: every thread owns one (8-row strip, column) of g, keeps all HZ*WZ tap sums in registers while the
//       window slides, and the 256 partial vectors are summed by a butterfly that halves the number of live
//       values per lane at every step (32 shuffles per wave instead of 6 per tap), then across the four waves
//       through LDS.
// ---------------------------------------------------------------------------
constexpr int RD = 9;       // dx rows per work item
constexpr int TAPS = 32;    // tap slots of the reduction (25 used at most)

// value of lane (l ^ 1) / (l ^ 2): data-parallel-primitive moves inside the quad (vector ALU; the general lane
// exchange goes through the LDS crossbar, which is this kernel's busiest unit)
__device__ __forceinline__ float quad_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float quad_xor2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
}

// v[0..32) per lane -> lane l holds the 64-lane total of slot 16 b0 + 8 b1 + 4 b2 + 2 b3 + b4 (b_s = bit s of l).
// The first two steps (24 of the 31 exchanges) stay inside the quad.
__device__ __forceinline__ float reduce_slots(float (&v)[TAPS], int lane) {
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int half = (TAPS >> s) >> 1;
        const bool hi = (lane >> s) & 1;
#pragma unroll
        for (int i = 0; i < half; ++i) {
            float lo_v = v[i], hi_v = v[half + i];
            // keep the two elements in registers: folded into one lane-indexed load, the select would move the
            // whole array to scratch memory
            asm volatile("" : "+v"(lo_v), "+v"(hi_v));
            const float keep = hi ? hi_v : lo_v;
            const float send = hi ? lo_v : hi_v;
            v[i] = keep + (s == 0 ? quad_xor1(send) : (s == 1 ? quad_xor2(send) : __shfl_xor(send, 1 << s)));
        }
    }
    return v[0] + __shfl_xor(v[0], 32);
}
__device__ __forceinline__ int slot_of_lane(int lane) {
    return ((lane & 1) << 4) | ((lane & 2) << 2) | (lane & 4) | ((lane & 8) >> 2) | ((lane & 16) >> 4);
}

template <int HZ, int WZ>
__device__ __forceinline__ void dz_strip(const float* gcol, int gp_cols, const float* xcol, int wx, float (&acc)[TAPS]) {
    float gk[R];
#pragma unroll
    for (int k = 0; k < R; ++k) gk[k] = gcol[k * gp_cols];
#pragma unroll
    for (int v = 0; v < WZ; ++v) {
        float xc[R + HZ - 1];
#pragma unroll
        for (int k = 0; k < R + HZ - 1; ++k) xc[k] = xcol[k * wx + v];
#pragma unroll
        for (int u = 0; u < HZ; ++u) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < R; ++k) t = fmaf(gk[k], xc[k + u], t);
            acc[u * WZ + v] = t;
        }
    }
}

// LDS floats of a staged x plane for the dz walk: rows up to (last strip) + R - 1 + hz - 1 are read
__host__ __device__ inline int xplane_floats(int hx, int wx, int hz, int Ho) {
    const int rows = ((Ho + R - 1) / R) * R + hz - 1;
    return ((rows > hx ? rows : hx) * wx + 3 + 3) & ~3;
}

template <int NB>
__global__ __launch_bounds__(256, 4) void groupdw_bwd5_kernel(Branch b0, Branch b1, Branch b2,
                                                              const float* __restrict__ weight,
                                                              const float* __restrict__ g, int Ho, int Wo,
                                                              float* __restrict__ dx0, float* __restrict__ dx1,
                                                              float* __restrict__ dx2, float* __restrict__ dz0,
                                                              float* __restrict__ dz1, float* __restrict__ dz2,
                                                              float* __restrict__ dsw_partial, int gp_rows, int gp_cols) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float zf[3][TAPS];          // reversed taps
    __shared__ float red[4][TAPS];
    constexpr int pad = 4;
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Branch br[3] = {b0, b1, b2};
    float* dxs[3] = {dx0, dx1, dx2};
    float* dzs[3] = {dz0, dz1, dz2};
    float* gp = smem;
    int off = (gp_rows * gp_cols + 3) & ~3;
    float* xs[3];
    // zero everything that is read but not loaded (padding of g, tails of the x planes), then stage
    int total = off;
#pragma unroll
    for (int i = 0; i < NB; ++i) total += xplane_floats(br[i].hx, br[i].wx, br[i].hz, Ho);
    for (int i = tid; i < total; i += 256) smem[i] = 0.f;
    if (tid < TAPS) {          // static branch index: a lane-dependent one would push the Branch array to scratch
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int nt = i < NB ? br[i].hz * br[i].wz : 0;
            zf[i][tid] = tid < nt ? br[i].z[(size_t)p * nt + nt - 1 - tid] : 0.f;
        }
    }
    __syncthreads();
    const float* gsrc = g + (size_t)p * Ho * Wo;
    for (int i = tid; i < Ho * Wo; i += 256) {
        const int r = i / Wo, c = i - r * Wo;
        gp[(r + pad) * gp_cols + c + pad] = gsrc[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int n = br[i].hx * br[i].wx;
        const int sh = stage_plane(br[i].x + (size_t)p * n, n, smem + off, tid);
        xs[i] = smem + off + sh;
        off += xplane_floats(br[i].hx, br[i].wx, br[i].hz, Ho);
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
    // ---- dx_i = s_i * corr(padded g, reversed z_i)
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        if (!dxs[i]) continue;
        const int hx = br[i].hx, wx = br[i].wx;
        float* op = dxs[i] + (size_t)p * hx * wx;
        const int nstrip = (hx + RD - 1) / RD;
        const float* origin = gp + (pad - (br[i].hz - 1)) * gp_cols + (pad - (br[i].wz - 1));
        for (int item = tid; item < nstrip * wx; item += 256) {
            const int strip = item / wx, col = item - strip * wx, a0 = strip * RD;
            float r[RD];
            if (i == 0) corr_column<5, 5, RD>(origin, gp_cols, zf[0], a0, col, r);
            if (i == 1) corr_column<3, 5, RD>(origin, gp_cols, zf[1], a0, col, r);
            if (i == 2) corr_column<5, 3, RD>(origin, gp_cols, zf[2], a0, col, r);
#pragma unroll
            for (int k = 0; k < RD; ++k)
                if (a0 + k < hx) op[(a0 + k) * wx + col] = sw[i] * r[k];
        }
    }
    // ---- dz_i and ds_i
    const int nstrip = (Ho + R - 1) / R;
    const bool active = tid < nstrip * Wo;
    const int strip = tid / Wo, col = tid - strip * Wo, i0 = strip * R;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        float acc[TAPS];
#pragma unroll
        for (int k = 0; k < TAPS; ++k) acc[k] = 0.f;
        if (active) {
            const float* gcol = gp + (i0 + pad) * gp_cols + col + pad;
            const float* xcol = xs[i] + i0 * br[i].wx + col;
            if (i == 0) dz_strip<5, 5>(gcol, gp_cols, xcol, br[0].wx, acc);
            if (i == 1) dz_strip<3, 5>(gcol, gp_cols, xcol, br[1].wx, acc);
            if (i == 2) dz_strip<5, 3>(gcol, gp_cols, xcol, br[2].wx, acc);
        }
        const float tot = reduce_slots(acc, lane);
        if (lane < 32) red[wave][slot_of_lane(lane)] = tot;
        __syncthreads();
        if (tid < 64) {
            const int nt = br[i].hz * br[i].wz;
            float t = 0.f, zs = 0.f;
            if (tid < nt) {
                t = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
                if (dzs[i]) dzs[i][(size_t)p * nt + tid] = sw[i] * t;
                zs = zf[i][nt - 1 - tid] * t;          // z_i[tap] * raw dz
            }
            zs = wave_sum(zs);
            if (NB == 3 && dsw_partial && tid == 0) dsw_partial[(size_t)p * 3 + i] = zs;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// Row-strip backward for the (5x5, 3x5, 5x3) triple (NB = 3) or a single 5x5 template (NB = 1): the forward row
// kernel's machinery (re-pitched LDS images filled by buffer loads, 16-byte conflict-free LDS reads, packed
// multiply-adds on register pairs as loaded, LDS-only barriers, persistent workgroups with the next plane in flight
// in registers, maps leaving through an LDS stage so that lanes run along rows) applied to both gradients.
//   dz_i[u][v] = s_i sum_{i,j} g[i][j] x_i[i+u][j+v]: a work item is 8 consecutive positions of one row of g.  For
//     tap row u it holds the 12-float segment of x row i+u and adds g[k] * (x[k+v], x[k+v+1]) into pairs of taps:
//     even k feed the pairs (dz0,dz1)(dz2,dz3)(dz4,-), odd k the pairs (-,dz0)(dz1,dz2)(dz3,dz4): again only aligned
//     register pairs, g[k] picked from its pair by the instruction's half selectors.  120 packed instead of 200
//     scalar multiply-adds for 5x5.  The per-thread tap sums are reduced by the slot butterfly below.
//   dx_i = s_i * correlation of the zero-padded g with the flipped taps: the forward's row_strip on the padded image.
// ---------------------------------------------------------------------------

// dzr[u WZ + v] += sum_{k<8} g[row][j0+k] * xs[row+u][j0+k+v]; grow points at g[row][j0] in the padded image
// FIRST: the item is the thread's first of this plane and branch: its sums are stored instead of added
template <int HZ, int WZ, int PITCH, bool FIRST>
__device__ __forceinline__ void dz_item(const float* xs, const float* grow, int row, int j0, float (&dzr)[TAPS]) {
    constexpr int NE = (WZ + 1) / 2, NO = WZ / 2 + 1;
    const f4v g0 = *reinterpret_cast<const f4v*>(grow), g1 = *reinterpret_cast<const f4v*>(grow + 4);
    const f2v gp[4] = {g0.xy, g0.zw, g1.xy, g1.zw};
    const float* base = xs + row * PITCH + j0;
#pragma unroll
    for (int u = 0; u < HZ; ++u) {
        f4v x[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) x[q] = *reinterpret_cast<const f4v*>(base + u * PITCH + 4 * q);
        const f2v xp[6] = {x[0].xy, x[0].zw, x[1].xy, x[1].zw, x[2].xy, x[2].zw};
        f2v E[NE], O[NO];          // E[m] = (dz[2m], dz[2m+1]); O[m] = (dz[2m-1], dz[2m])
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if ((k & 1) == 0) {
#pragma unroll
                for (int m = 0; m < NE; ++m) {
                    if (k == 0) pk_mul_g_lo(E[m], gp[0], xp[m]);
                    else pk_fma_g_lo(E[m], gp[k >> 1], xp[k / 2 + m]);
                }
            } else {
#pragma unroll
                for (int m = 0; m < NO; ++m) {
                    if (k == 1) pk_mul_g_hi(O[m], gp[0], xp[m]);
                    else pk_fma_g_hi(O[m], gp[k >> 1], xp[(k - 1) / 2 + m]);
                }
            }
        }
#pragma unroll
        for (int v = 0; v < WZ; ++v) {
            const float t = (v & 1) ? E[v / 2].y + O[v / 2 + 1].x : E[v / 2].x + O[v / 2].y;
            dzr[u * WZ + v] = FIRST ? t : dzr[u * WZ + v] + t;
        }
    }
}

// Shapes are compile-time (square base plane of side HX, 5x5 base template: branch planes HX x HX, (HX-2) x HX,
// HX x (HX-2)); with run-time shapes the three unrolled branch bodies keep more derived scalars alive than there
// are scalar registers.
struct BranchC {
    int hx, wx, hz, wz;
};
__host__ __device__ constexpr BranchC branch_shape(int HX, int i) {
    return i == 0 ? BranchC{HX, HX, 5, 5} : (i == 1 ? BranchC{HX - 2, HX, 3, 5} : BranchC{HX, HX - 2, 5, 3});
}
// pitch of the LDS images: covers the widest row, the last strip's 12-float segment and the padded g; 4 * odd
__host__ __device__ constexpr int row_pitch(int HX) {
    const int Wo = HX - 4;
    int need = ((HX + 7) / 8 - 1) * 8 + 12;
    const int a = ((Wo + 7) / 8 - 1) * 8 + 12, c = Wo + 8, d = ((HX + 3) / 4) * 4;
    need = a > need ? a : need;
    need = c > need ? c : need;
    need = d > need ? d : need;
    int p = (need + 3) / 4;
    if (p % 2 == 0) ++p;
    return 4 * p;
}
__host__ __device__ constexpr int bwd_row_lds_floats(int NB, int HX) {
    int rows = HX - 4 + 8;
    for (int i = 0; i < NB; ++i) rows += branch_shape(HX, i).hx;
    return rows * row_pitch(HX);
}

// PART: 0 = both gradients in one launch, 1 = dz / ds only, 2 = dx only (stages g and the taps alone).  Two launches read g
// twice, but each half needs fewer registers and less LDS than the whole: more workgroups per CU to hide the latencies of
// the many short dependent phases of a plane.
template <int NB, int HX, int PART>
__global__ __launch_bounds__(256, PART == 0 ? 3 : 4) void groupdw_bwd_row_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                                 const float* __restrict__ x2, const float* __restrict__ z0,
                                                                 const float* __restrict__ z1, const float* __restrict__ z2,
                                                                 const float* __restrict__ weight, const float* __restrict__ g,
                                                                 float* __restrict__ dx0, float* __restrict__ dx1,
                                                                 float* __restrict__ dx2, float* __restrict__ dz0,
                                                                 float* __restrict__ dz1, float* __restrict__ dz2,
                                                                 float* __restrict__ dsw_partial, int planes) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float red[4][TAPS];
    __shared__ __attribute__((aligned(16))) float zl[3][TAPS];          // this plane's taps (staged with the planes)
    constexpr int PAD = 4, PITCH = row_pitch(HX), Ho = HX - 4, Wo = HX - 4, lds_floats = bwd_row_lds_floats(PART == 2 ? 0 : NB, HX);
    constexpr bool DZ = PART != 2, DX = PART != 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr BranchC bc[3] = {branch_shape(HX, 0), branch_shape(HX, 1), branch_shape(HX, 2)};
    struct BranchP {
        const float* x;
        const float* z;
        int hx, wx, hz, wz;
    };
    const BranchP br[3] = {{x0, z0, bc[0].hx, bc[0].wx, 5, 5}, {x1, z1, bc[1].hx, bc[1].wx, 3, 5}, {x2, z2, bc[2].hx, bc[2].wx, 5, 3}};
    float* dxs[3] = {dx0, dx1, dx2};
    float* dzs[3] = {dz0, dz1, dz2};
    int offs[3], goffs;
    {
        int off = 0;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            offs[i] = off;
            if (DZ) off += br[i].hx * PITCH;
        }
        goffs = off;                   // zero-padded g: (Ho + 2 PAD) rows, g[r][c] at row r + PAD, column c + PAD
    }
    float sw[3] = {1.f, 0.f, 0.f};
    if (NB == 3) {
        const float w0 = weight[0], w1 = weight[1], w2 = weight[2];
        const float m = fmaxf(w0, fmaxf(w1, w2));
        const float e0 = expf(w0 - m), e1 = expf(w1 - m), e2 = expf(w2 - m);
        const float s = e0 + e1 + e2;
        sw[0] = e0 / s; sw[1] = e1 / s; sw[2] = e2 / s;
    }
    // everything that is read but never staged (the padding of g, the pitch columns of the x images) starts at zero
    for (int i = tid; i < lds_floats; i += 256) smem[i] = 0.f;
    // staging maps of the x planes (as in the forward kernel) and of g
    int passes[NB], gstep[NB], lstep[NB];
    unsigned g_first[NB], g_last[NB];
    int l_first[NB], l_last[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int Q = (br[i].wx + 3) >> 2, RPT = 256 / Q;
        passes[i] = (br[i].hx + RPT - 1) / RPT;
        const int t = min(tid, RPT * Q - 1), r0 = t / Q, q = t - r0 * Q;
        const int rl = min(r0 + (passes[i] - 1) * RPT, br[i].hx - 1);
        g_first[i] = (unsigned)(r0 * br[i].wx + 4 * q) * 4u;
        g_last[i] = (unsigned)(rl * br[i].wx + 4 * q) * 4u;
        l_first[i] = offs[i] + r0 * PITCH + 4 * q;
        l_last[i] = offs[i] + rl * PITCH + 4 * q;
        gstep[i] = RPT * br[i].wx * 4;
        lstep[i] = RPT * PITCH;
    }
    const int GQ = (Wo + 3) >> 2, GRPT = 256 / GQ, gpasses = (Ho + GRPT - 1) / GRPT;     // gpasses <= 2 (host)
    unsigned gg[2];
    int gl[2], gvalid;                 // floats of this thread's quad that lie inside a row of g (the rest must stay 0)
    {
        const int t = min(tid, GRPT * GQ - 1), r0 = t / GQ, q = t - r0 * GQ;
        const int r1 = min(r0 + GRPT, Ho - 1);
        gg[0] = (unsigned)(r0 * Wo + 4 * q) * 4u;
        gg[1] = (unsigned)(r1 * Wo + 4 * q) * 4u;
        gl[0] = goffs + (r0 + PAD) * PITCH + PAD + 4 * q;
        gl[1] = goffs + (r1 + PAD) * PITCH + PAD + 4 * q;
        gvalid = min(4, Wo - 4 * q);
    }
    // Buffer descriptors are made per plane (base = the plane, range = its bytes: the 3 floats a row's last quad runs
    // past the plane read as zeros) right where they are used: kept for the whole launch they would crowd the
    // scalar registers.
    f4v regs[NB][ROW_NV], gregs[2];
    float zreg[NB];
    auto fetch = [&](int p) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int n = br[i].hx * br[i].wx;
            __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(br[i].x) + (size_t)p * n, 0, n * 4, 0x00020000);
#pragma unroll
            for (int k = 0; k < ROW_NV; ++k) {
                if (!DZ || k >= passes[i]) continue;
                const bool last = k == passes[i] - 1;
                regs[i][k] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(
                    rx, (int)(last ? g_last[i] : g_first[i]), last ? 0 : k * gstep[i], 0));
            }
            const int nt = br[i].hz * br[i].wz;
            zreg[i] = br[i].z[(size_t)p * nt + min(tid, nt - 1)];
        }
        __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g) + (size_t)p * Ho * Wo, 0, Ho * Wo * 4, 0x00020000);
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (k < gpasses) gregs[k] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rg, (int)gg[k], 0, 0));
    };
    int p = blockIdx.x;
    if (p < planes) fetch(p);
    lds_barrier();                     // the zero fill is done before the first commit
    for (; p < planes; p += gridDim.x) {
        // ---- commit the staged planes
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int k = 0; k < ROW_NV; ++k) {
                if (!DZ || k >= passes[i]) continue;
                const bool last = k == passes[i] - 1;
                *reinterpret_cast<f4v*>(smem + (last ? l_last[i] : l_first[i] + k * lstep[i])) = regs[i][k];
            }
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (k < gpasses) {
                f4v v = gregs[k];
                if (gvalid < 2) v.y = 0.f;
                if (gvalid < 3) v.z = 0.f;
                if (gvalid < 4) v.w = 0.f;
                *reinterpret_cast<f4v*>(smem + gl[k]) = v;
            }
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (tid < TAPS) zl[i][tid] = tid < br[i].hz * br[i].wz ? zreg[i] : 0.f;
        lds_barrier();
        const int pn = p + gridDim.x;
        if (pn < planes) fetch(pn);
        // ---- dz_i and ds_i
        const int nstrip_g = (Wo + 7) >> 3, nitems_g = nstrip_g * Ho;
#pragma unroll
        for (int i = 0; i < (DZ ? NB : 0); ++i) {
            float acc[TAPS];
#pragma unroll
            for (int k = 0; k < TAPS; ++k) acc[k] = 0.f;
#pragma unroll
            for (int base = 0; base < nitems_g; base += 256) {
                const int item = base + tid;
                if (item < nitems_g) {
                    const int strip = item / Ho, row = item - strip * Ho, j0 = strip * 8;
                    const float* grow = smem + goffs + (row + PAD) * PITCH + PAD + j0;
                    if (base == 0) {
                        if (i == 0) dz_item<5, 5, PITCH, true>(smem + offs[0], grow, row, j0, acc);
                        if (i == 1) dz_item<3, 5, PITCH, true>(smem + offs[1], grow, row, j0, acc);
                        if (i == 2) dz_item<5, 3, PITCH, true>(smem + offs[2], grow, row, j0, acc);
                    } else {
                        if (i == 0) dz_item<5, 5, PITCH, false>(smem + offs[0], grow, row, j0, acc);
                        if (i == 1) dz_item<3, 5, PITCH, false>(smem + offs[1], grow, row, j0, acc);
                        if (i == 2) dz_item<5, 3, PITCH, false>(smem + offs[2], grow, row, j0, acc);
                    }
                }
            }
            const float tot = reduce_slots(acc, lane);
            if (lane < 32) red[wave][slot_of_lane(lane)] = tot;
            lds_barrier();
            if (tid < 64) {
                const int nt = br[i].hz * br[i].wz;
                float zs = 0.f;
                if (tid < nt) {
                    const float t = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
                    if (dzs[i]) dzs[i][(size_t)p * nt + tid] = sw[i] * t;
                    zs = zl[i][tid] * t;
                }
                zs = wave_sum(zs);
                if (NB == 3 && dsw_partial && tid == 0) dsw_partial[(size_t)p * 3 + i] = zs;
            }
            lds_barrier();             // red is free again; after the last branch: nobody reads the x images any more
        }
        // ---- dx_i = s_i * corr(padded g, flipped taps)
#pragma unroll
        for (int i = 0; i < (DX ? NB : 0); ++i) {
            if (!dxs[i]) continue;
            const int hx = br[i].hx, wx = br[i].wx;
            const int nitems = ((wx + 7) >> 3) * hx;
            // window origin of this branch inside the padding: rows PAD - (hz - 1), columns PAD - (wz - 1)
            const float* origin = smem + goffs + (PAD - (br[i].hz - 1)) * PITCH;
            for (int base = 0; base < nitems; base += 256) {
                // the items past the first 256 (a few rows of the last strip) go to a different wave for every branch
                const int item = base + (base == 0 ? tid : ((tid + 192 - 64 * i) & 255));
                const bool active = item < nitems;
                // strips fastest: a wave's stores then cover whole consecutive rows of dx (1 KB per store instruction in
                // 8 lines instead of 64 pieces in 64 lines); the price is a 2-way bank conflict on 2 of 16 lanes of the
                // LDS reads
                const int nst = (wx + 7) >> 3;
                const int row = item / nst, strip = item - row * nst, j0 = strip * 8;
                float r[8];
                if (active) {          // the taps: broadcast 8-byte LDS reads, every lane gets the same pairs
                    if (i == 0) {
                        f2v zt[13];
#pragma unroll
                        for (int m = 0; m < 13; ++m) zt[m] = *reinterpret_cast<const f2v*>(&zl[0][2 * m]);
                        row_strip<5, 5, PITCH, 0, true, true>(origin, zt, row, j0, r);
                    }
                    if (i == 1) {
                        f2v zt[8];
#pragma unroll
                        for (int m = 0; m < 8; ++m) zt[m] = *reinterpret_cast<const f2v*>(&zl[1][2 * m]);
                        row_strip<3, 5, PITCH, 0, true, true>(origin, zt, row, j0, r);
                    }
                    if (i == 2) {
                        f2v zt[8];
#pragma unroll
                        for (int m = 0; m < 8; ++m) zt[m] = *reinterpret_cast<const f2v*>(&zl[2][2 * m]);
                        row_strip<5, 3, PITCH, 2, true, true>(origin, zt, row, j0, r);
                    }
                }
                if (active) {
                    // straight from registers: the LDS pipe is this kernel's busiest unit (a stage for coalesced rows, as
                    // in the forward kernel, costs it more than the partial-line writes cost the L2)
                    const f4v lo = f4v{sw[i] * r[0], sw[i] * r[1], sw[i] * r[2], sw[i] * r[3]};
                    const f4v hi = f4v{sw[i] * r[4], sw[i] * r[5], sw[i] * r[6], sw[i] * r[7]};
                    if (j0 + 8 <= wx) {
                        __amdgpu_buffer_rsrc_t rdx = __builtin_amdgcn_make_buffer_rsrc(dxs[i] + (size_t)p * hx * wx, 0, hx * wx * 4, 0x00020000);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4v, lo), rdx, (row * wx + j0) * 4, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4v, hi), rdx, (row * wx + j0) * 4 + 16, 0, 0);
                    } else {
                        float* orow = dxs[i] + (size_t)p * hx * wx + row * wx + j0;
                        const float vals[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            if (j0 + k < wx) orow[k] = vals[k];
                    }
                }
            }
        }
        lds_barrier();                 // the padded g is no longer needed: the next commit may overwrite it
    }
}

// Shapes the backward row kernel is built for: square base plane of side 45 (384^2 search crop) or 29 (255^2), 5x5
// base template, tensors below 2 GiB; false leaves the call to the kernels above.
template <int NB, int HX, int PART>
void launch_bwd_row_part(const Branch* b, const float* weight, const float* g, float** dx, float** dz, float* dsw, int planes,
                         hipStream_t stream) {
    constexpr size_t lds = (size_t)bwd_row_lds_floats(PART == 2 ? 0 : NB, HX) * sizeof(float);
    constexpr int lds_cap = (int)((160 * 1024) / (lds + 2048));         // + the static arrays and allocation granularity
    constexpr int reg_cap = PART == 0 ? 3 : 4;
    constexpr int per_cu = lds_cap < reg_cap ? (lds_cap < 1 ? 1 : lds_cap) : reg_cap;
    const int grid = planes < 256 * per_cu ? planes : 256 * per_cu;
    groupdw_bwd_row_kernel<NB, HX, PART><<<grid, 256, lds, stream>>>(b[0].x, b[1].x, b[2].x, b[0].z, b[1].z, b[2].z, weight, g,
                                                                     dx[0], dx[1], dx[2], dz[0], dz[1], dz[2], dsw, planes);
}

// One launch for both gradients.  Measured on MI355X at 384^2 / B = 128: the whole (PART 0) 646 us; dz alone (PART 1) 345 us
// plus dx alone (PART 2) 375 us = 715 us, because g is read twice and the dx half is bound by its write stream either way.
template <int NB, int HX>
void launch_bwd_row_hx(const Branch* b, const float* weight, const float* g, float** dx, float** dz, float* dsw, int planes,
                       hipStream_t stream) {
    launch_bwd_row_part<NB, HX, 0>(b, weight, g, dx, dz, dsw, planes, stream);
}

template <int NB>
bool launch_bwd_row(const Branch* b, const float* weight, const float* g, float** dx, float** dz, float* dsw, int planes,
                    hipStream_t stream) {
    const int HX = b[0].hx;
    if (b[0].hz != 5 || b[0].wz != 5 || b[0].wx != HX || (HX != 45 && HX != 29)) return false;
    if ((long)planes * HX * HX * 4 >= (1L << 31)) return false;
    if (HX == 45) launch_bwd_row_hx<NB, 45>(b, weight, g, dx, dz, dsw, planes, stream);
    else launch_bwd_row_hx<NB, 29>(b, weight, g, dx, dz, dsw, planes, stream);
    return true;
}

int launch_bwd(const Branch* b, int nb, const float* weight, const float* g, float** dx, float** dz, float* dsw,
               int planes, hipStream_t stream) {
    const int Ho = b[0].hx - b[0].hz + 1, Wo = b[0].wx - b[0].wz + 1;
    int pad = 0;
    for (int i = 0; i < nb; ++i) {
        pad = b[i].hz - 1 > pad ? b[i].hz - 1 : pad;
        pad = b[i].wz - 1 > pad ? b[i].wz - 1 : pad;
    }
    SPOT_CHECK_ARG(pad <= GPAD_MAX, "xcorr backward: template extent %d exceeds %d", pad + 1, GPAD_MAX + 1);
    if (nb == 3 ? launch_bwd_row<3>(b, weight, g, dx, dz, dsw, planes, stream)
                : (nb == 1 && launch_bwd_row<1>(b, weight, g, dx, dz, dsw, planes, stream))) {
        SPOT_CHECK_LAUNCH("xcorr backward");
        return SPOT_OK;
    }
    const bool five = b[0].hz == 5 && b[0].wz == 5 && ((Ho + R - 1) / R) * Wo <= 256;   // one dz item per thread
    if (five) {
        int rows = 0;
        for (int i = 0; i < nb; ++i) {
            const int need = ((b[i].hx + RD - 1) / RD) * RD + b[i].hz - 1 + (4 - (b[i].hz - 1));
            rows = need > rows ? need : rows;
        }
        const int dzrows = ((Ho + R - 1) / R) * R + 2 * 4;
        rows = dzrows > rows ? dzrows : rows;
        const int gp_cols5 = Wo + 8;
        size_t f5 = ((size_t)rows * gp_cols5 + 3) & ~(size_t)3;
        for (int i = 0; i < nb; ++i) f5 += xplane_floats(b[i].hx, b[i].wx, b[i].hz, Ho);
        if (f5 * sizeof(float) <= 64 * 1024) {
            if (nb == 3)
                groupdw_bwd5_kernel<3><<<planes, 256, f5 * sizeof(float), stream>>>(b[0], b[1], b[2], weight, g, Ho, Wo, dx[0],
                                                                                   dx[1], dx[2], dz[0], dz[1], dz[2], dsw,
                                                                                   rows, gp_cols5);
            else
                groupdw_bwd5_kernel<1><<<planes, 256, f5 * sizeof(float), stream>>>(b[0], b[1], b[2], weight, g, Ho, Wo, dx[0],
                                                                                   dx[1], dx[2], dz[0], dz[1], dz[2], dsw,
                                                                                   rows, gp_cols5);
            SPOT_CHECK_LAUNCH("xcorr backward");
            return SPOT_OK;
        }
    }
    int max_hx = 0;
    for (int i = 0; i < nb; ++i) max_hx = b[i].hx > max_hx ? b[i].hx : max_hx;
    // rows: every R-strip of dx may read R + hz - 1 rows starting at its origin
    const int gp_rows = ((max_hx + R - 1) / R) * R + 2 * pad, gp_cols = Wo + 2 * pad;
    size_t f = ((size_t)gp_rows * gp_cols + 3) & ~(size_t)3;
    for (int i = 0; i < nb; ++i) f += ((size_t)b[i].hx * b[i].wx + 6) & ~(size_t)3;
    SPOT_CHECK_ARG(f * sizeof(float) <= 64 * 1024, "xcorr backward: planes of %dx%d do not fit the 64 KB LDS stage", b[0].hx,
                   b[0].wx);
    groupdw_bwd_kernel<<<planes, 256, f * sizeof(float), stream>>>(b[0], b[1], b[2], nb, weight, g, Ho, Wo, dx[0], dx[1],
                                                                  dx[2], dz[0], dz[1], dz[2], dsw, pad, gp_rows, gp_cols);
    SPOT_CHECK_LAUNCH("xcorr backward");
    return SPOT_OK;
}

}  // namespace

extern "C" int spot_xcorr_depthwise_bwd_f32(const float* x, const float* kernel, const float* d_out, float* dx,
                                            float* dkernel, int planes, int hx, int wx, int hz, int wz,
                                            hipStream_t stream) {
    SPOT_CHECK_ARG(planes >= 0 && hx >= hz && wx >= wz && hz > 0 && wz > 0,
                   "xcorr_depthwise_bwd: bad shape planes=%d x=%dx%d kernel=%dx%d", planes, hx, wx, hz, wz);
    if (planes == 0) return SPOT_OK;
    SPOT_CHECK_ARG(x && kernel && d_out && (dx || dkernel), "xcorr_depthwise_bwd: null pointer");
    Branch b[3] = {{x, kernel, hx, wx, hz, wz}, {}, {}};
    float* dxs[3] = {dx, nullptr, nullptr};
    float* dzs[3] = {dkernel, nullptr, nullptr};
    return launch_bwd(b, 1, nullptr, d_out, dxs, dzs, nullptr, planes, stream);
}

extern "C" int spot_groupdw_bwd_f32(const float* x11, const float* x12, const float* x21, const float* z11,
                                    const float* z12, const float* z21, const float* weight3, const float* d_out,
                                    float* dx11, float* dx12, float* dx21, float* dz11, float* dz12, float* dz21,
                                    float* dsoftmax_partial, int planes, int hx, int wx, int hz, int wz,
                                    hipStream_t stream) {
    SPOT_CHECK_ARG(planes >= 0 && hz >= 3 && wz >= 3 && hx >= hz && wx >= wz,
                   "groupdw_bwd: bad shape planes=%d x11=%dx%d z11=%dx%d", planes, hx, wx, hz, wz);
    if (planes == 0) return SPOT_OK;
    SPOT_CHECK_ARG(x11 && x12 && x21 && z11 && z12 && z21 && weight3 && d_out, "groupdw_bwd: null pointer");
    Branch b[3] = {{x11, z11, hx, wx, hz, wz}, {x12, z12, hx - 2, wx, hz - 2, wz}, {x21, z21, hx, wx - 2, hz, wz - 2}};
    float* dxs[3] = {dx11, dx12, dx21};
    float* dzs[3] = {dz11, dz12, dz21};
    return launch_bwd(b, 3, weight3, d_out, dxs, dzs, dsoftmax_partial, planes, stream);
}

// Gradient of the three mixing logits from the per-plane sums the backward kernel leaves (GroupDW.forward mixes with
// softmax(weight), ltr/models/neck/ocean_fusion.py:217-236): ds_k = sum over planes, dw = s * (ds - <s, ds>).
// One workgroup, planes summed in double in a fixed order.
namespace {
__global__ __launch_bounds__(256) void groupdw_weight_grad_kernel(const float* __restrict__ part, int planes,
                                                                  const float* __restrict__ weight, float* __restrict__ dw) {
    __shared__ double red[3][256];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int p = threadIdx.x; p < planes; p += 256) {
        a0 += (double)part[(size_t)p * 3];
        a1 += (double)part[(size_t)p * 3 + 1];
        a2 += (double)part[(size_t)p * 3 + 2];
    }
    red[0][threadIdx.x] = a0; red[1][threadIdx.x] = a1; red[2][threadIdx.x] = a2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float m = fmaxf(weight[0], fmaxf(weight[1], weight[2]));
        const float e0 = expf(weight[0] - m), e1 = expf(weight[1] - m), e2 = expf(weight[2] - m);
        const float inv = 1.f / (e0 + e1 + e2);
        const float s[3] = {e0 * inv, e1 * inv, e2 * inv};
        const float ds[3] = {(float)red[0][0], (float)red[1][0], (float)red[2][0]};
        const float dot = s[0] * ds[0] + s[1] * ds[1] + s[2] * ds[2];
        for (int k = 0; k < 3; ++k) dw[k] = s[k] * (ds[k] - dot);
    }
}
}  // namespace

extern "C" int spot_groupdw_weight_grad_f32(const float* dsoftmax_partial, int planes, const float* weight3, float* d_weight3,
                                            hipStream_t stream) {
    SPOT_CHECK_ARG(dsoftmax_partial && weight3 && d_weight3 && planes > 0, "groupdw_weight_grad: bad arguments");
    groupdw_weight_grad_kernel<<<1, 256, 0, stream>>>(dsoftmax_partial, planes, weight3, d_weight3);
    SPOT_CHECK_LAUNCH("groupdw_weight_grad");
    return SPOT_OK;
}
