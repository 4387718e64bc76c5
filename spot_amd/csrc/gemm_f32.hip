// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32,
// bit-identical to an fmaf chain, 64 FLOP/clk/SIMD).  It carries every dense
// contraction of the fusion network and the heads, forward and backward:
//   nn.Linear / in-proj / out-proj / FFN / MLP heads / 1x1 input_proj
//   (ltr/models/neck/transt_fusion.py:121-133,172-199; transt_model.py:47-56,296-310).
//
//   C[M,N] (+)= alpha * opA(A) * opB(B)  (+ bias[n]) (+ rowbias[m % rb_rows][n]) (-> inference BatchNorm over rows m)
//               -> relu? -> (* mask>0)? (+ residual) -> relu (relu = 2)?
//
// Operand layouts (row-major storage, leading dimension in elements):
//   A: "MK" = A[m*lda + k]   or "KM" = A[k*lda + m]   (a transposed activation / NCHW feature map)
//   B: "NK" = B[n*ldb + k]   (an nn.Linear weight)  or "KN" = B[k*ldb + n]
// Tiling: 128x128x32 per 256-thread workgroup, 4 waves as 2x2, each wave a 2x2
// grid of 32x32 MFMA tiles (64 accumulator VGPRs); problems that would leave most
// of the chip idle at that size (or whose M or N is 64) take a 64x64x32 tile, one
// MFMA tile per wave.  Tiles are staged
// global -> registers -> LDS: the next tile's global loads are issued before the
// 64 MFMAs of the current one, the LDS writes after them.  LDS keeps the operand's own storage order
// (k-contiguous rows padded to 36 floats for conflict-free 16-byte fragment
// reads; m/n-contiguous rows read with 4-byte reads), so no transposes happen
// anywhere.  Within an 8-wide k group lane half h takes k = 4h..4h+3: the
// matrix instruction's two k slots then cover {j, 4+j}, j = 0..3, on both
// operands, which is a pure re-ordering of the k sum.
#include "common.h"
#include "../../include/spot_hip.h"
#include <type_traits>
#include <cstdint>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---- "bf16x3" products (precision = 1, an experiment kept apart from the exact path) ---------------------------------
// An fp32 value is cut, by truncation, into three bf16 pieces that sum to it exactly (8 + 8 + 8 mantissa bits):
//   hi = x & 0xffff0000, mid = (x - hi) & 0xffff0000, lo = (x - hi) - mid   (lo has at most 8 significant bits).
// a*b is then taken as hi*hi' + hi*mid' + mid*hi' + hi*lo' + lo*hi' + mid*mid' on v_mfma_f32_32x32x16_bf16 (each
// product exact in the fp32 accumulator); the three dropped terms are below 2^-23 |a||b|.  The pieces are made once per
// element, when a tile goes from registers to LDS, and LDS holds three bf16 planes per operand:
//   plane p, row r, k: byte p*PLANE + r*B3_ROW + 2k,  B3_ROW = 80 (64 bytes of k + 16 pad: 16-byte fragment reads of
//   32 consecutive rows fall on distinct banks).  An operand stored with k as the slow index is gathered four k at a time
//   on its way into registers, so all four layouts share the LDS image and the matrix loop.
constexpr int B3_ROW = 80;
__device__ __forceinline__ void split3(float4 v, uint2& hi, uint2& mid, uint2& lo) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    unsigned a[4], b[4], c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] = __float_as_uint(x[j]);
        const float r = x[j] - __uint_as_float(a[j] & 0xffff0000u);
        b[j] = __float_as_uint(r);
        c[j] = __float_as_uint(r - __uint_as_float(b[j] & 0xffff0000u));
    }
    // v_perm_b32: the upper halves of two dwords side by side, the first value in the lower half
    hi = make_uint2(__builtin_amdgcn_perm(a[1], a[0], 0x07060302u), __builtin_amdgcn_perm(a[3], a[2], 0x07060302u));
    mid = make_uint2(__builtin_amdgcn_perm(b[1], b[0], 0x07060302u), __builtin_amdgcn_perm(b[3], b[2], 0x07060302u));
    lo = make_uint2(__builtin_amdgcn_perm(c[1], c[0], 0x07060302u), __builtin_amdgcn_perm(c[3], c[2], 0x07060302u));
}

// TS = MFMA tiles per wave along m and along n: 2 -> 128x128 workgroup tile, 1 -> 64x64
constexpr int BK = 32;
constexpr int KPAD = BK + 4;    // k-contiguous LDS row stride (floats)
template <int TS> struct Tile {
    static constexpr int BM = 64 * TS, BN = 64 * TS;
    static constexpr int NL = 2 * TS;          // 16-byte loads per thread per operand tile
    static constexpr int MPAD = BM + 4;        // m/n-contiguous LDS row stride (floats)
    static constexpr int TILE_F = BM * KPAD;   // floats per operand tile buffer (>= 32*MPAD)
    static constexpr int CPAD = BN + 4;        // row stride of the epilogue's C staging tile (64 x CPAD floats <= 2*TILE_F)
    static constexpr int RSH = TS == 2 ? 5 : 4;   // log2(float4 per m/n-contiguous tile row)
};

struct GemmArgs {
    const float* A; const float* B; float* C;
    int M, N, K;
    long lda, ldb, ldc;
    long strideA, strideB, strideC;   // per batch (grid.z / splitk)
    int kseg; long segA, segB;        // K is cut in segments of kseg; segment s adds s*segA / s*segB
    int splitk;
    float alpha;
    int accumulate;                   // C += result (plain read-modify-write; atomic when splitk > 1)
    const float* bias;                // [N] or null
    const float* rowbias; int rb_rows; long ldrb;   // [rb_rows, rb_cols] indexed by m % rb_rows, columns n < rb_cols
    int rb_cols;
    int relu;
    const float* mask; long ldmask;   // multiply by (mask[m][n] > 0)
    const float* residual; long ldres; long strideRes;
    int cvec;                         // every epilogue operand allows 16-byte accesses
    int fast;                         // operands allow the unguarded interior-tile loader (see gemm_kernel)
    int precision;                    // 0 exact fp32, 1 bf16x3
    unsigned drop_thr, drop_seed; float drop_scale;   // dropout after relu, before residual (0 = off)
    float* rowsum;                    // [M] += sum_k opA(A)[m][k] (atomic), or null
    // inference-mode BatchNorm over the rows (output channels of a 1x1 convolution), or bn_gamma = null:
    // v = v * a + (beta[m] - mean[m] * a), a = gamma[m] / sqrt(var[m] + eps)
    const float* bn_gamma; const float* bn_beta; const float* bn_mean; const float* bn_var; float bn_eps;
};

template <int LAYOUT_KCONTIG, bool ALIGNED, int TS>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, long ld, int r0, int k0, int R, int K,
                                          float4 (&st)[Tile<TS>::NL], int tid) {
    // LAYOUT_KCONTIG: element (r,k) at P[r*ld + k], tile BM rows x 32 k, float4 along k
    // else          : element (r,k) at P[k*ld + r], tile 32 k x BM rows, float4 along r
    constexpr int RSH = Tile<TS>::RSH, RMASK = (1 << RSH) - 1;
#pragma unroll
    for (int i = 0; i < Tile<TS>::NL; ++i) {
        const int f = tid + 256 * i;
        int r, k;
        if (LAYOUT_KCONTIG) { r = r0 + (f >> 3); k = k0 + ((f & 7) << 2); }
        else                { k = k0 + (f >> RSH); r = r0 + ((f & RMASK) << 2); }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (LAYOUT_KCONTIG) {
            if (r < R) {
                const float* p = P + (long)r * ld + k;
                if (ALIGNED && k + 3 < K) v = *reinterpret_cast<const float4*>(p);
                else {
                    if (k < K) v.x = p[0];
                    if (k + 1 < K) v.y = p[1];
                    if (k + 2 < K) v.z = p[2];
                    if (k + 3 < K) v.w = p[3];
                }
            }
        } else {
            if (k < K) {
                const float* p = P + (long)k * ld + r;
                if (ALIGNED && r + 3 < R) v = *reinterpret_cast<const float4*>(p);
                else {
                    if (r < R) v.x = p[0];
                    if (r + 1 < R) v.y = p[1];
                    if (r + 2 < R) v.z = p[2];
                    if (r + 3 < R) v.w = p[3];
                }
            }
        }
        st[i] = v;
    }
}

template <int LAYOUT_KCONTIG, int TS>
__device__ __forceinline__ void store_tile(float* lds, const float4 (&st)[Tile<TS>::NL], int tid) {
    constexpr int RSH = Tile<TS>::RSH, RMASK = (1 << RSH) - 1, MPAD = Tile<TS>::MPAD;
#pragma unroll
    for (int i = 0; i < Tile<TS>::NL; ++i) {
        const int f = tid + 256 * i;
        if (LAYOUT_KCONTIG) *reinterpret_cast<float4*>(lds + (f >> 3) * KPAD + ((f & 7) << 2)) = st[i];
        else                *reinterpret_cast<float4*>(lds + (f >> RSH) * MPAD + ((f & RMASK) << 2)) = st[i];
    }
}

// fragment for one 32-row tile and one 8-wide k group: 4 floats, element j <-> k = g*8 + 4h + j
template <int LAYOUT_KCONTIG, int MPAD>
__device__ __forceinline__ float4 read_frag(const float* lds, int row, int g, int h) {
    if (LAYOUT_KCONTIG) return *reinterpret_cast<const float4*>(lds + row * KPAD + g * 8 + 4 * h);
    const float* p = lds + (g * 8 + 4 * h) * MPAD + row;
    return make_float4(p[0], p[MPAD], p[2 * MPAD], p[3 * MPAD]);
}

// What follows the k loop, shared by the workgroup-tile kernels: the bias gradient's row sums, then the epilogue.
// `lds` is the kernel's whole LDS allocation (free after the k loop; at least 64 x CPAD floats), `zb` / `zs` the batch
// and split-K indices of this workgroup, (m0, n0) its tile origin.
template <int TS, bool ROWSUM>
__device__ __forceinline__ void finish_tile(const GemmArgs& g, f32x16 (&acc)[TS][TS], float (&rs)[TS], bool do_rs, float* lds,
                                            int m0, int n0, int zb, int zs, int tid) {
    using T = Tile<TS>;
    constexpr int BN = T::BN, CPAD = T::CPAD, RSH = T::RSH, RMASK = (1 << RSH) - 1, WT = 32 * TS;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, l31 = lane & 31;
    if (ROWSUM && do_rs) {
        // a lane holds the k = 4h..4h+3 (mod 8) share of row l31 of each of its MFMA row tiles
#pragma unroll
        for (int t = 0; t < TS; ++t) {
            const float v = rs[t] + __shfl_xor(rs[t], 32);
            const int m = m0 + wm * WT + t * 32 + l31;
            if (h == 0 && m < g.M) atomicAdd(g.rowsum + m, v);
        }
    }

    // epilogue.  Per-lane dword stores of the accumulator layout (32 floats per row segment) are
    // store-issue bound, so the tile is transposed through LDS (free after the k loop) and leaves
    // the chip as 16-byte stores of whole 512-byte rows; bias / row-bias / mask / residual are
    // read the same way.  Accumulator element e of tile (t,u): row (e&3)+8*(e>>2)+4h, col lane&31.
    float* Cs = lds;
    float* C = g.C + zb * g.strideC;
    const bool first = zs == 0;                      // additive terms enter once under split-K
    const float* res = (g.residual && first) ? g.residual + zb * g.strideRes : nullptr;
    const float* rowbias = first ? g.rowbias : nullptr;
    const float* bias = first ? g.bias : nullptr;
    constexpr int CSH = RSH;                          // log2(float4 per staged C row)
    const int c4 = (tid & RMASK) << 2;
    const int n = n0 + c4;
    const bool full = g.cvec && n + 3 < g.N;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias && g.splitk == 1 && n < g.N) {
        if (full) bv = *reinterpret_cast<const float4*>(bias + n);
        else {
            bv.x = bias[n];
            if (n + 1 < g.N) bv.y = bias[n + 1];
            if (n + 2 < g.N) bv.z = bias[n + 2];
            if (n + 3 < g.N) bv.w = bias[n + 3];
        }
    }
    // the staging tile holds 64 rows at a time: for the 128-row tile, rows of the wm = 0 waves first, then wm = 1
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
        if (g.splitk > 1) {
            // split-K partial sums: one dword per lane, 64 consecutive floats per wave-instruction
            // (the shape float atomics run at full rate in)
            const int col = tid & (BN - 1), nn = n0 + col;
            if (nn < g.N) {
                const float b1 = bias ? bias[nn] : 0.f;
                for (int i = 0; i < 64 / (256 / BN); ++i) {
                    const int row = i * (256 / BN) + tid / BN, m = mh + row;
                    if (m >= g.M) break;
                    float x = Cs[row * CPAD + col] * g.alpha + b1;
                    if (rowbias && nn < g.rb_cols) x += rowbias[(long)(m % g.rb_rows) * g.ldrb + nn];
                    if (res) x += res[(long)m * g.ldres + nn];
                    atomicAdd(C + (long)m * g.ldc + nn, x);
                }
            }
            continue;
        }
        if (n >= g.N) continue;
#pragma unroll 4
        for (int i = 0; i < (64 >> (8 - CSH)); ++i) {
            const int row = (i << (8 - CSH)) + (tid >> CSH);
            const int m = mh + row;
            if (m >= g.M) break;
            float4 v = *reinterpret_cast<const float4*>(Cs + row * CPAD + c4);
            v.x = v.x * g.alpha + bv.x; v.y = v.y * g.alpha + bv.y; v.z = v.z * g.alpha + bv.z; v.w = v.w * g.alpha + bv.w;
            float* cp = C + (long)m * g.ldc + n;
            if (full) {
                if (rowbias && n < g.rb_cols) {
                    const float4 r4 = *reinterpret_cast<const float4*>(rowbias + (long)(m % g.rb_rows) * g.ldrb + n);
                    v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
                }
                if (g.bn_gamma) {
                    const float sc = g.bn_gamma[m] / sqrtf(g.bn_var[m] + g.bn_eps), sh = g.bn_beta[m] - g.bn_mean[m] * sc;
                    v.x = fmaf(v.x, sc, sh); v.y = fmaf(v.y, sc, sh); v.z = fmaf(v.z, sc, sh); v.w = fmaf(v.w, sc, sh);
                }
                if (g.relu == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (g.drop_thr) {
                    const unsigned rk = spot_row_key(g.drop_seed, (unsigned)(zb * g.M + m));
                    v.x = spot_keep(rk, n, g.drop_thr) ? v.x * g.drop_scale : 0.f;
                    v.y = spot_keep(rk, n + 1, g.drop_thr) ? v.y * g.drop_scale : 0.f;
                    v.z = spot_keep(rk, n + 2, g.drop_thr) ? v.z * g.drop_scale : 0.f;
                    v.w = spot_keep(rk, n + 3, g.drop_thr) ? v.w * g.drop_scale : 0.f;
                }
                if (g.mask) {
                    const float4 k4 = *reinterpret_cast<const float4*>(g.mask + (long)m * g.ldmask + n);
                    v.x = k4.x > 0.f ? v.x : 0.f; v.y = k4.y > 0.f ? v.y : 0.f;
                    v.z = k4.z > 0.f ? v.z : 0.f; v.w = k4.w > 0.f ? v.w : 0.f;
                }
                if (res) {
                    const float4 r4 = *reinterpret_cast<const float4*>(res + (long)m * g.ldres + n);
                    v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
                }
                if (g.relu == 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (g.accumulate) {
                    float4 o = *reinterpret_cast<float4*>(cp);
                    o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
                    *reinterpret_cast<float4*>(cp) = o;
                } else {
                    *reinterpret_cast<float4*>(cp) = v;
                }
            } else {
                float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (n + j >= g.N) break;
                    float x = vv[j];
                    if (rowbias && n + j < g.rb_cols) x += rowbias[(long)(m % g.rb_rows) * g.ldrb + n + j];
                    if (g.bn_gamma) {
                        const float sc = g.bn_gamma[m] / sqrtf(g.bn_var[m] + g.bn_eps);
                        x = fmaf(x, sc, g.bn_beta[m] - g.bn_mean[m] * sc);
                    }
                    if (g.relu == 1) x = fmaxf(x, 0.f);
                    if (g.drop_thr)
                        x = spot_keep(spot_row_key(g.drop_seed, (unsigned)(zb * g.M + m)), n + j, g.drop_thr) ? x * g.drop_scale : 0.f;
                    if (g.mask) x = g.mask[(long)m * g.ldmask + n + j] > 0.f ? x : 0.f;
                    if (res) x += res[(long)m * g.ldres + n + j];
                    if (g.relu == 2) x = fmaxf(x, 0.f);
                    if (g.accumulate) cp[j] += x;
                    else cp[j] = x;
                }
            }
        }
    }
}

// ROWSUM: the workgroups of the first tile column also add the row sums of opA(A) into g.rowsum (the bias
// gradient of a linear layer when A is the transposed output gradient: its tiles pass through here anyway).
// PREC: 0 = exact fp32 products, 1 = bf16x3 (interior tiles of k-contiguous operands only; the host checks)
// DEEP: exact fp32 with the loads of four k tiles in flight per workgroup, for grids of at most a few workgroups per CU
//       (one clip: the online trackers), where nothing else hides the memory latency of a k step
template <int LA, int LB, bool ALIGNED, int TS, bool ROWSUM, int PREC = 0, bool DEEP = false>   // LA: 1 = MK (k-contiguous), 0 = KM;  LB: 1 = NK, 0 = KN
__global__ __launch_bounds__(256, DEEP ? 2 : (PREC ? (TS == 2 ? 2 : 4) : (TS == 2 ? 3 : 6))) void gemm_kernel(GemmArgs g) {
    using T = Tile<TS>;
    constexpr int BM = T::BM, BN = T::BN, NL = T::NL, MPAD = T::MPAD, TILE_F = T::TILE_F;
    constexpr int RSH = T::RSH, RMASK = (1 << RSH) - 1, WT = 32 * TS;   // WT: rows (cols) of a wave's sub-tile
    constexpr int B3_PLANE = BM * B3_ROW;                               // bytes per bf16 plane of one operand
    __shared__ __attribute__((aligned(16))) float lds[PREC ? 6 * B3_PLANE / 4 : 2 * TILE_F];
    float* const As = lds;
    float* const Bs = lds + TILE_F;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, l31 = lane & 31;

    // XCD-aware tile order: workgroups that land on one XCD (ids equal mod 8) walk a contiguous
    // range of tiles, n fastest, so the A row panel a tile needs is already in that XCD's L2.
    const int ntn = (g.N + BN - 1) / BN, ntm = (g.M + BM - 1) / BM, nwg = ntm * ntn;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, x = bid & 7, loc = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
    }
    const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;
    const int zb = blockIdx.z / g.splitk, zs = blockIdx.z % g.splitk;

    const float* A = g.A + zb * g.strideA;
    const float* B = g.B + zb * g.strideB;
    const int nkt = (g.K + BK - 1) / BK;
    const int per = (nkt + g.splitk - 1) / g.splitk;
    const int kt0 = zs * per, kt1 = min(nkt, kt0 + per);

    f32x16 acc[TS][TS];
#pragma unroll
    for (int t = 0; t < TS; ++t)
#pragma unroll
        for (int u = 0; u < TS; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;

    float rs[TS];
#pragma unroll
    for (int t = 0; t < TS; ++t) rs[t] = 0.f;
    const bool do_rs = ROWSUM && n0 == 0 && wn == 0;

    float4 sa[NL], sb[NL];
    auto fetch = [&](int kt) {
        const int k = kt * BK;
        const int seg = k / g.kseg, kin = k - seg * g.kseg;
        load_tile<LA, ALIGNED, TS>(A + seg * g.segA, g.lda, m0, kin, g.M, min(g.kseg, g.K - seg * g.kseg), sa, tid);
        load_tile<LB, ALIGNED, TS>(B + seg * g.segB, g.ldb, n0, kin, g.N, min(g.kseg, g.K - seg * g.kseg), sb, tid);
    };
    // Interior tiles (whole 128x128 tile and every k tile in bounds, 16-byte aligned operands): the guarded
    // loader above costs a chain of ~90 branchy basic blocks per k tile; here a thread keeps four 32-bit
    // element offsets per operand and the k walk is one scalar base-pointer bump per tile.
    unsigned offA[NL], offB[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int f = tid + 256 * i;
        offA[i] = LA ? (unsigned)((m0 + (f >> 3)) * g.lda + ((f & 7) << 2)) : (unsigned)((f >> RSH) * g.lda + m0 + ((f & RMASK) << 2));
        offB[i] = LB ? (unsigned)((n0 + (f >> 3)) * g.ldb + ((f & 7) << 2)) : (unsigned)((f >> RSH) * g.ldb + n0 + ((f & RMASK) << 2));
    }
    int fseg = (kt0 * BK) / g.kseg, fkin = kt0 * BK - fseg * g.kseg;     // position of the next tile to fetch
    auto fetch_fast = [&]() {
        const float* Ab = A + fseg * g.segA + (LA ? (long)fkin : (long)fkin * g.lda);
        const float* Bb = B + fseg * g.segB + (LB ? (long)fkin : (long)fkin * g.ldb);
#pragma unroll
        for (int i = 0; i < NL; ++i) sa[i] = *reinterpret_cast<const float4*>(Ab + offA[i]);
#pragma unroll
        for (int i = 0; i < NL; ++i) sb[i] = *reinterpret_cast<const float4*>(Bb + offB[i]);
        fkin += BK;
        if (fkin >= g.kseg) {
            fkin = 0;
            ++fseg;
        }
    };
    const bool fast = ALIGNED && g.fast && m0 + BM <= g.M && n0 + BN <= g.N;

    // One LDS buffer per operand (36.9 KB per workgroup, so three workgroups share a CU and the
    // prologue / epilogue of one hides behind the matrix work of the others).  The next tile's
    // global loads are in flight during the 64 MFMAs of the current tile and land in LDS between
    // two barriers.
    auto k_loop = [&](auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        if (kt0 < kt1) {
            if (FAST) fetch_fast();
            else fetch(kt0);
            store_tile<LA, TS>(As, sa, tid);
            store_tile<LB, TS>(Bs, sb, tid);
        }
        __syncthreads();
        for (int kt = kt0; kt < kt1; ++kt) {
            const bool more = kt + 1 < kt1;
            if (more) {
                if (FAST) fetch_fast();
                else fetch(kt + 1);
            }
#pragma unroll
            for (int gk = 0; gk < BK / 8; ++gk) {
                float4 a[TS], b[TS];
#pragma unroll
                for (int t = 0; t < TS; ++t) a[t] = read_frag<LA, MPAD>(As, wm * WT + t * 32 + l31, gk, h);
#pragma unroll
                for (int u = 0; u < TS; ++u) b[u] = read_frag<LB, MPAD>(Bs, wn * WT + u * 32 + l31, gk, h);
                if (ROWSUM && do_rs) {
#pragma unroll
                    for (int t = 0; t < TS; ++t) rs[t] += (a[t].x + a[t].y) + (a[t].z + a[t].w);
                }
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
                store_tile<LA, TS>(As, sa, tid);
                store_tile<LB, TS>(Bs, sb, tid);
                __syncthreads();
            }
        }
    };
    // bf16x3 main loop.  LDS holds tile kt (bf16 pieces); one register set holds the raw fp32 tile kt+1, the other
    // has the loads of tile kt+2 in flight.  Between the two barriers of an iteration a wave issues its 48 matrix
    // instructions and, in their shadow, cuts tile kt+1 into pieces and writes them to LDS (the fragments of tile kt
    // are all in registers by then), then re-issues that register set's loads for tile kt+3.  Past the last k tile the
    // loader stays on it: a duplicate tile is staged and never read.
    auto k_loop_bf16x3 = [&]() {
        unsigned char* const Ab = reinterpret_cast<unsigned char*>(lds);
        unsigned char* const Bb = Ab + 3 * B3_PLANE;
        float4 ra[NL], rb[NL];
        // A thread's share of an operand tile is NL groups of four consecutive k of one row: one 16-byte load when k
        // is the contiguous index, four 4-byte loads ld apart (lanes along the row index) when it is not.  Either
        // way the group lands in LDS as 3 x 8 bytes at (row, k / 4).
        int rowA[NL], kqA[NL], rowB[NL], kqB[NL];
        unsigned oA[NL], oB[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int f = tid + 256 * i;
            if (LA) { rowA[i] = f >> 3; kqA[i] = f & 7; oA[i] = offA[i]; }
            else    { rowA[i] = f & (BM - 1); kqA[i] = f / BM; oA[i] = (unsigned)(4 * kqA[i] * g.lda + m0 + rowA[i]); }
            if (LB) { rowB[i] = f >> 3; kqB[i] = f & 7; oB[i] = offB[i]; }
            else    { rowB[i] = f & (BN - 1); kqB[i] = f / BN; oB[i] = (unsigned)(4 * kqB[i] * g.ldb + n0 + rowB[i]); }
        }
        int left = kt1 - kt0 - 1;                      // tiles after the one the loader points at
        auto fetch_next = [&](float4 (&da)[NL], float4 (&db)[NL]) {
            const float* Ap = A + fseg * g.segA + (LA ? (long)fkin : (long)fkin * g.lda);
            const float* Bp = B + fseg * g.segB + (LB ? (long)fkin : (long)fkin * g.ldb);
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                if (LA) da[i] = *reinterpret_cast<const float4*>(Ap + oA[i]);
                else da[i] = make_float4(Ap[oA[i]], (Ap + g.lda)[oA[i]], (Ap + 2 * g.lda)[oA[i]], (Ap + 3 * g.lda)[oA[i]]);
            }
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                if (LB) db[i] = *reinterpret_cast<const float4*>(Bp + oB[i]);
                else db[i] = make_float4(Bp[oB[i]], (Bp + g.ldb)[oB[i]], (Bp + 2 * g.ldb)[oB[i]], (Bp + 3 * g.ldb)[oB[i]]);
            }
            const int adv = left > 0, nk = fkin + BK, wrap = nk >= g.kseg;
            left -= adv;
            fkin = adv ? (wrap ? 0 : nk) : fkin;
            fseg += adv & wrap;
        };
        auto stage = [&](const float4 (&da)[NL], const float4 (&db)[NL]) {
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                uint2 p0, p1, p2;
                const int atA = rowA[i] * B3_ROW + (kqA[i] << 3), atB = rowB[i] * B3_ROW + (kqB[i] << 3);
                split3(da[i], p0, p1, p2);
                *reinterpret_cast<uint2*>(Ab + atA) = p0;
                *reinterpret_cast<uint2*>(Ab + B3_PLANE + atA) = p1;
                *reinterpret_cast<uint2*>(Ab + 2 * B3_PLANE + atA) = p2;
                split3(db[i], p0, p1, p2);
                *reinterpret_cast<uint2*>(Bb + atB) = p0;
                *reinterpret_cast<uint2*>(Bb + B3_PLANE + atB) = p1;
                *reinterpret_cast<uint2*>(Bb + 2 * B3_PLANE + atB) = p2;
            }
        };
        auto body = [&](float4 (&xa)[NL], float4 (&xb)[NL]) {
            // lane half h holds k = 16 gg + 8h .. + 7 of its row on both operands
            bf16x8 a[2][TS][3], b[2][TS][3];
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
#pragma unroll
                for (int t = 0; t < TS; ++t)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        a[gg][t][p] = *reinterpret_cast<const bf16x8*>(Ab + p * B3_PLANE + (wm * WT + t * 32 + l31) * B3_ROW + gg * 32 + h * 16);
#pragma unroll
                for (int u = 0; u < TS; ++u)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        b[gg][u][p] = *reinterpret_cast<const bf16x8*>(Bb + p * B3_PLANE + (wn * WT + u * 32 + l31) * B3_ROW + gg * 32 + h * 16);
            }
            __syncthreads();                           // every wave holds its fragments: LDS may be overwritten
#pragma unroll
            for (int gg = 0; gg < 2; ++gg)
#pragma unroll
                for (int t = 0; t < TS; ++t)
#pragma unroll
                    for (int u = 0; u < TS; ++u) {
                        // smallest terms first
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[gg][t][1], b[gg][u][1], acc[t][u], 0, 0, 0);
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[gg][t][2], b[gg][u][0], acc[t][u], 0, 0, 0);
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[gg][t][0], b[gg][u][2], acc[t][u], 0, 0, 0);
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[gg][t][1], b[gg][u][0], acc[t][u], 0, 0, 0);
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[gg][t][0], b[gg][u][1], acc[t][u], 0, 0, 0);
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[gg][t][0], b[gg][u][0], acc[t][u], 0, 0, 0);
                    }
            stage(xa, xb);
            fetch_next(xa, xb);
            // issue order between the barriers: one matrix instruction, then a few of the vector / LDS-store
            // instructions of the staging, so that they run while the matrix pipe is busy
#pragma unroll
            for (int i = 0; i < 12 * TS * TS; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, TS == 2 ? 4 : 8, 0);
                if ((i & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x200, TS == 2 ? 1 : 2, 0);
            }
            __syncthreads();                           // the staged tile is visible
        };
        if (kt0 < kt1) {        // an empty k range of a split-K launch starts past the end of the operands: no loads
            fetch_next(sa, sb);
            stage(sa, sb);
            fetch_next(sa, sb);
            fetch_next(ra, rb);
        }
        __syncthreads();
        for (int kt = kt0; kt < kt1; kt += 2) {
            body(sa, sb);
            if (kt + 1 < kt1) body(ra, rb);
        }
    };
    // Deep-prefetch loop.  With one workgroup per CU the 16 matrix instructions of a k step (0.5 us) cannot cover a
    // global load (1-2 us), and the loop above keeps only the next tile in flight: a k step then costs a memory
    // latency.  Here four tiles are in flight in four register sets; the barriers order LDS traffic only (a full
    // __syncthreads would wait for every outstanding load and undo the prefetch).
    auto k_loop_deep = [&]() {
        static_assert(!DEEP || TS == 1, "the deep-prefetch loop is written for the 64x64 tile (two 16-byte loads per thread and operand)");
        constexpr int D = 4;
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        struct Set { f32x4 a0, a1, b0, b1; };           // native vectors (the float4 class is copied by memcpy, which pins it in scratch)
        Set s0, s1, s2, s3;
        auto lds_barrier = [] {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        };
        // LDS slots of this thread's two vectors per operand (as store_tile places them)
        const int f0 = tid, f1 = tid + 256;
        float* const la0 = As + (LA ? (f0 >> 3) * KPAD + ((f0 & 7) << 2) : (f0 >> RSH) * MPAD + ((f0 & RMASK) << 2));
        float* const la1 = As + (LA ? (f1 >> 3) * KPAD + ((f1 & 7) << 2) : (f1 >> RSH) * MPAD + ((f1 & RMASK) << 2));
        float* const lb0 = Bs + (LB ? (f0 >> 3) * KPAD + ((f0 & 7) << 2) : (f0 >> RSH) * MPAD + ((f0 & RMASK) << 2));
        float* const lb1 = Bs + (LB ? (f1 >> 3) * KPAD + ((f1 & 7) << 2) : (f1 >> RSH) * MPAD + ((f1 & RMASK) << 2));
        int left = kt1 - kt0 - 1;                      // tiles after the one the loader points at
        // (written as macros over the named sets: handed to a lambda by reference the sets end up in scratch memory)
        // the loader always loads: past the last k tile it stays on it (a duplicate that is never staged)
#define SPOT_DEEP_FETCH(RS)                                                                        \
        {                                                                                         \
            const float* Ab = A + fseg * g.segA + (LA ? (long)fkin : (long)fkin * g.lda);         \
            const float* Bb = B + fseg * g.segB + (LB ? (long)fkin : (long)fkin * g.ldb);         \
            RS.a0 = *reinterpret_cast<const f32x4*>(Ab + offA[0]);                                 \
            RS.a1 = *reinterpret_cast<const f32x4*>(Ab + offA[1]);                                \
            RS.b0 = *reinterpret_cast<const f32x4*>(Bb + offB[0]);                                \
            RS.b1 = *reinterpret_cast<const f32x4*>(Bb + offB[1]);                                \
            const int adv = left > 0, nk = fkin + BK, wrap = nk >= g.kseg;                        \
            left -= adv;                                                                          \
            fkin = adv ? (wrap ? 0 : nk) : fkin;                                                  \
            fseg += adv & wrap;                                                                   \
        }
#define SPOT_DEEP_STEP(RS)                                                                         \
        {                                                                                         \
            *reinterpret_cast<f32x4*>(la0) = RS.a0;                                               \
            *reinterpret_cast<f32x4*>(la1) = RS.a1;                                               \
            *reinterpret_cast<f32x4*>(lb0) = RS.b0;                                               \
            *reinterpret_cast<f32x4*>(lb1) = RS.b1;                                               \
            lds_barrier();                                                                        \
            SPOT_DEEP_FETCH(RS)                                                                    \
            _Pragma("unroll") for (int gk = 0; gk < BK / 8; ++gk) {                               \
                const float4 a = read_frag<LA, MPAD>(As, wm * WT + l31, gk, h);                   \
                const float4 b = read_frag<LB, MPAD>(Bs, wn * WT + l31, gk, h);                   \
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[0][0], 0, 0, 0);   \
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[0][0], 0, 0, 0);   \
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[0][0], 0, 0, 0);   \
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[0][0], 0, 0, 0);   \
            }                                                                                     \
            lds_barrier();                                                                        \
        }
        // (a split-K launch can hand a workgroup an empty k range, which starts past the end of the operands: nothing
        //  may be loaded then)
        if (kt0 < kt1) {
        SPOT_DEEP_FETCH(s0)
        SPOT_DEEP_FETCH(s1)
        SPOT_DEEP_FETCH(s2)
        SPOT_DEEP_FETCH(s3)
        // whole rounds of four steps without a branch inside (the wait counts for "this set's loads, not the twelve
        // younger ones" need every path to have issued the same number of loads), then the 0-3 steps that are left
        int kt = kt0;
        for (; kt + D <= kt1; kt += D) {
            SPOT_DEEP_STEP(s0)
            SPOT_DEEP_STEP(s1)
            SPOT_DEEP_STEP(s2)
            SPOT_DEEP_STEP(s3)
        }
        if (kt < kt1) SPOT_DEEP_STEP(s0)
        if (kt + 1 < kt1) SPOT_DEEP_STEP(s1)
        if (kt + 2 < kt1) SPOT_DEEP_STEP(s2)
        }
#undef SPOT_DEEP_STEP
#undef SPOT_DEEP_FETCH
        __syncthreads();
    };
    if constexpr (DEEP) {
        k_loop_deep();
    } else if constexpr (PREC == 1) {
        k_loop_bf16x3();
    } else {
        if (fast) k_loop(std::true_type{});
        else k_loop(std::false_type{});
    }

    finish_tile<TS, ROWSUM>(g, acc, rs, do_rs, lds, m0, n0, zb, zs, tid);
}

// ---------------------------------------------------------------------------
// Ring kernel: the same workgroup tiles, products and summation order as gemm_kernel (so the same bits), for problems
// made of whole tiles of aligned operands.  Operand tiles go global -> LDS directly (global_load_lds_dwordx4: no
// register staging, no LDS store instructions) into a ring of two stages with ONE barrier per k tile: the loads of
// tile k+1 are issued right behind the barrier that retires tile k and land while tile k is multiplied.
//   LDS image of a k-contiguous operand: the tile's 128-byte k rows, unpadded (a wave instruction writes 8 whole rows in
//     lane order), 16-byte chunk c of row r stored at chunk position c ^ ((r >> 1) & 7): the 16-byte fragment reads of
//     16 consecutive rows then cover all 64 banks once.
//   LDS image of an m/n-contiguous operand: 32 k rows of BM floats, unpadded, chunk c of k row kk at position
//     c ^ (((kk >> 2) & 1) << 3): lane half h = 1 (k + 4) reads the other 128-byte half of the bank row than h = 0.
//   Either swizzle is applied on the per-lane SOURCE address (the LDS destination of a direct load is lane-linear).
// The direct loads are written as asm: the compiler's wait-count insertion would otherwise drain every outstanding
// direct load before each LDS read; the k loop therefore holds no compiler-visible vector-memory load at all and
// waits with its own s_waitcnt.
// ---------------------------------------------------------------------------
template <int GA> __device__ __forceinline__ void direct_load_pieces(const float* base, const unsigned (&voff)[GA], unsigned lds_addr);
// GA consecutive 1 KiB pieces: lane's 16 bytes from base + voff[j] bytes, piece j at LDS byte address lds_addr + 1024 j
template <> __device__ __forceinline__ void direct_load_pieces<2>(const float* base, const unsigned (&voff)[2], unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff[0]), "v"(voff[1]), "s"(base), "s"(lds_addr) : "memory", "scc");
}
template <> __device__ __forceinline__ void direct_load_pieces<4>(const float* base, const unsigned (&voff)[4], unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(base), "s"(lds_addr) : "memory", "scc");
}

template <int LA, int LB, int TS, bool ROWSUM>
__global__ __launch_bounds__(256) void gemm_ring_kernel(GemmArgs g) {
    using T = Tile<TS>;
    constexpr int BM = T::BM, BN = T::BN, WT = 32 * TS;
    constexpr int A_F = BM * BK, STAGE_F = (BM + BN) * BK;        // floats per operand tile / per stage
    constexpr int GA = BM / 32;                                    // 1 KiB pieces per wave per operand tile
    static_assert(2 * STAGE_F >= 64 * T::CPAD, "the epilogue's staging tile must fit in the ring");
    __shared__ __attribute__((aligned(1024))) float lds[2 * STAGE_F];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, l31 = lane & 31;
    const int ntn = g.N / BN, ntm = g.M / BM, nwg = ntm * ntn;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, x = bid & 7, loc = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
    }
    const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;
    const int zb = blockIdx.z / g.splitk, zs = blockIdx.z % g.splitk;
    const int nkt = g.K / BK;
    const int per = (nkt + g.splitk - 1) / g.splitk;
    const int kt0 = zs * per, kt1 = min(nkt, kt0 + per);

    // tile origins at k = 0 (scalar), and the lane's byte offset inside a tile for each of the wave's pieces
    const float* Ao = g.A + zb * g.strideA + (LA ? (long)m0 * g.lda : (long)m0);
    const float* Bo = g.B + zb * g.strideB + (LB ? (long)n0 * g.ldb : (long)n0);
    unsigned voA[GA], voB[GA];
#pragma unroll
    for (int j = 0; j < GA; ++j) {
        const int i = wave * GA + j;                               // piece index inside the operand tile
        if (LA) {
            const int row = 8 * i + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
            voA[j] = 4u * (unsigned)(row * g.lda + 4 * c);
        } else {
            constexpr int CH = BM / 4;                             // 16-byte chunks per k row
            const int kk = i * (256 / BM) + lane / CH, c = (lane % CH) ^ (((kk >> 2) & 1) << 3);
            voA[j] = 4u * (unsigned)(kk * g.lda + 4 * c);
        }
        if (LB) {
            const int row = 8 * i + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
            voB[j] = 4u * (unsigned)(row * g.ldb + 4 * c);
        } else {
            constexpr int CH = BN / 4;
            const int kk = i * (256 / BN) + lane / CH, c = (lane % CH) ^ (((kk >> 2) & 1) << 3);
            voB[j] = 4u * (unsigned)(kk * g.ldb + 4 * c);
        }
    }
    // fragment read offsets (floats inside an operand tile).  k-contiguous: one per 8-wide k group (16-byte read of row
    // l31 of the wave's sub-tile 0; sub-tile t is 32 rows further).  m/n-contiguous: one per sub-tile (k row 4h, the
    // lane's column swizzled); k group and element are whole k rows further.
    int fak[4], fbk[4], fam[TS], fbm[TS];
#pragma unroll
    for (int gk = 0; gk < 4; ++gk) {
        fak[gk] = (wm * WT + l31) * BK + (((2 * gk + h) ^ ((l31 >> 1) & 7)) << 2);
        fbk[gk] = (wn * WT + l31) * BK + (((2 * gk + h) ^ ((l31 >> 1) & 7)) << 2);
    }
#pragma unroll
    for (int t = 0; t < TS; ++t) {
        fam[t] = 4 * h * BM + ((wm * WT + t * 32 + l31) ^ (h << 5));
        fbm[t] = 4 * h * BN + ((wn * WT + t * 32 + l31) ^ (h << 5));
    }

    f32x16 acc[TS][TS];
#pragma unroll
    for (int t = 0; t < TS; ++t)
#pragma unroll
        for (int u = 0; u < TS; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;
    float rs[TS];
#pragma unroll
    for (int t = 0; t < TS; ++t) rs[t] = 0.f;
    const bool do_rs = ROWSUM && n0 == 0 && wn == 0;

    const unsigned lds0 = (unsigned)(uintptr_t)lds + (unsigned)__builtin_amdgcn_readfirstlane(wave) * (GA * 1024u);
    int fseg = (kt0 * BK) / g.kseg, fkin = kt0 * BK - fseg * g.kseg;     // position of the next tile to fetch
    auto issue = [&](int slot) {
        const float* Ap = Ao + fseg * g.segA + (LA ? (long)fkin : (long)fkin * g.lda);
        const float* Bp = Bo + fseg * g.segB + (LB ? (long)fkin : (long)fkin * g.ldb);
        const unsigned dst = lds0 + (unsigned)slot * (STAGE_F * 4u);
        direct_load_pieces<GA>(Ap, voA, dst);
        direct_load_pieces<GA>(Bp, voB, dst + A_F * 4u);
        fkin += BK;
        if (fkin >= g.kseg) {
            fkin = 0;
            ++fseg;
        }
    };
    if (kt0 < kt1) issue(0);       // an empty k range of a split-K launch starts past the end of the operands: no loads
    int slot = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of tile kt have landed
        __builtin_amdgcn_s_barrier();                              // everyone's have; everyone is done reading the other slot
        if (kt + 1 < kt1) issue(slot ^ 1);
        const float* As = lds + slot * STAGE_F;
        const float* Bs = As + A_F;
#pragma unroll
        for (int gk = 0; gk < BK / 8; ++gk) {
            float4 a[TS], b[TS];
#pragma unroll
            for (int t = 0; t < TS; ++t) {
                if (LA) a[t] = *reinterpret_cast<const float4*>(As + fak[gk] + t * 32 * BK);
                else { const float* q = As + fam[t] + gk * 8 * BM; a[t] = make_float4(q[0], q[BM], q[2 * BM], q[3 * BM]); }
            }
#pragma unroll
            for (int u = 0; u < TS; ++u) {
                if (LB) b[u] = *reinterpret_cast<const float4*>(Bs + fbk[gk] + u * 32 * BK);
                else { const float* q = Bs + fbm[u] + gk * 8 * BN; b[u] = make_float4(q[0], q[BN], q[2 * BN], q[3 * BN]); }
            }
            if (ROWSUM && do_rs) {
#pragma unroll
                for (int t = 0; t < TS; ++t) rs[t] += (a[t].x + a[t].y) + (a[t].z + a[t].w);
            }
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
        slot ^= 1;
    }
    __syncthreads();                                               // every wave has read the last stage
    finish_tile<TS, ROWSUM>(g, acc, rs, do_rs, lds, m0, n0, zb, zs, tid);
}

// ---------------------------------------------------------------------------
// Small grids (one clip: the online trackers' 256- and 1024-token layers).  With 16 to 64 tiles of 64x64 most CUs idle and
// a launch lasts as long as one wave's chain of K/2 dependent matrix instructions (52 us at M = 1024, N = 256, K = 2048).
// Here a workgroup owns a 32x32 output tile and its four waves split every 128-wide k step between them: four times as
// many workgroups, a chain a quarter as long.  The four partial tiles are summed through LDS in a fixed order (no
// atomics: the same bits on every run), then the usual epilogue runs on one float4 per thread.
// Whole tiles of aligned operands only (M, N multiples of 32, K and every K segment multiples of 128); the host checks.
// ---------------------------------------------------------------------------
constexpr int KS_TM = 32, KS_K = 128, KS_KP = KS_K + 4, KS_MP = KS_TM + 4, KS_TILE_F = KS_K * KS_MP;   // 4608 >= 32 * 132

// (bid, zb): the workgroup's tile index and batch index (a kernel launched for two problems hands each workgroup
//  the problem it belongs to and its index inside that problem)
template <int LA, int LB>
__device__ __forceinline__ void gemm_ks4_body(const GemmArgs& g, int bid, int zb, float* lds) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    float* const As = lds;
    float* const Bs = lds + KS_TILE_F;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int ntn = g.N / KS_TM;
    const int m0 = (bid / ntn) * KS_TM, n0 = (bid % ntn) * KS_TM;
    const float* A = g.A + zb * g.strideA;
    const float* B = g.B + zb * g.strideB;

    // a thread's four 16-byte pieces of each operand tile: global element offsets and LDS slots
    unsigned offA[4], offB[4];
    int ldsA[4], ldsB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + 256 * i;
        if (LA) { offA[i] = (unsigned)((m0 + (f >> 5)) * g.lda + ((f & 31) << 2)); ldsA[i] = (f >> 5) * KS_KP + ((f & 31) << 2); }
        else    { offA[i] = (unsigned)((f >> 3) * g.lda + m0 + ((f & 7) << 2));    ldsA[i] = (f >> 3) * KS_MP + ((f & 7) << 2); }
        if (LB) { offB[i] = (unsigned)((n0 + (f >> 5)) * g.ldb + ((f & 31) << 2)); ldsB[i] = (f >> 5) * KS_KP + ((f & 31) << 2); }
        else    { offB[i] = (unsigned)((f >> 3) * g.ldb + n0 + ((f & 7) << 2));    ldsB[i] = (f >> 3) * KS_MP + ((f & 7) << 2); }
    }
    f32x4 ra[4], rb[4];
    int fseg = 0, fkin = 0;
    auto fetch = [&]() {
        const float* Ab = A + fseg * g.segA + (LA ? (long)fkin : (long)fkin * g.lda);
        const float* Bb = B + fseg * g.segB + (LB ? (long)fkin : (long)fkin * g.ldb);
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f32x4*>(Ab + offA[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = *reinterpret_cast<const f32x4*>(Bb + offB[i]);
        fkin += KS_K;
        if (fkin >= g.kseg) {
            fkin = 0;
            ++fseg;
        }
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int nst = g.K / KS_K;
    fetch();
    for (int st = 0; st < nst; ++st) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(As + ldsA[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(Bs + ldsB[i]) = rb[i];
        __syncthreads();
        if (st + 1 < nst) fetch();
        const int kw = 32 * wave;                     // this wave's quarter of the k step
#pragma unroll
        for (int gk = 0; gk < 4; ++gk) {
            const int k = kw + gk * 8 + 4 * h;        // lane half h takes k .. k+3 of the 8-wide group, on both operands
            float4 a, b;
            if (LA) a = *reinterpret_cast<const float4*>(As + l31 * KS_KP + k);
            else { const float* p = As + k * KS_MP + l31; a = make_float4(p[0], p[KS_MP], p[2 * KS_MP], p[3 * KS_MP]); }
            if (LB) b = *reinterpret_cast<const float4*>(Bs + l31 * KS_KP + k);
            else { const float* p = Bs + k * KS_MP + l31; b = make_float4(p[0], p[KS_MP], p[2 * KS_MP], p[3 * KS_MP]); }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // the four partial tiles: accumulator element e sits at row (e&3) + 8 (e>>2) + 4h, column lane & 31
    float* red = lds + wave * (KS_TM * KS_MP);
#pragma unroll
    for (int e = 0; e < 16; ++e) red[((e & 3) + 8 * (e >> 2) + 4 * h) * KS_MP + l31] = acc[e];
    __syncthreads();
    const int row = tid >> 3, c4 = (tid & 7) << 2, m = m0 + row, n = n0 + c4;
    float4 v = *reinterpret_cast<const float4*>(lds + row * KS_MP + c4);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const float4 p = *reinterpret_cast<const float4*>(lds + w * (KS_TM * KS_MP) + row * KS_MP + c4);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    // epilogue, in the order of gemm_kernel's
    v.x *= g.alpha; v.y *= g.alpha; v.z *= g.alpha; v.w *= g.alpha;
    if (g.bias) {
        const float4 b4 = *reinterpret_cast<const float4*>(g.bias + n);
        v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
    }
    if (g.rowbias && n < g.rb_cols) {
        const float4 r4 = *reinterpret_cast<const float4*>(g.rowbias + (long)(m % g.rb_rows) * g.ldrb + n);
        v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
    }
    if (g.bn_gamma) {
        const float sc = g.bn_gamma[m] / sqrtf(g.bn_var[m] + g.bn_eps), sh = g.bn_beta[m] - g.bn_mean[m] * sc;
        v.x = fmaf(v.x, sc, sh); v.y = fmaf(v.y, sc, sh); v.z = fmaf(v.z, sc, sh); v.w = fmaf(v.w, sc, sh);
    }
    if (g.relu == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (g.drop_thr) {
        const unsigned rk = spot_row_key(g.drop_seed, (unsigned)(zb * g.M + m));
        v.x = spot_keep(rk, n, g.drop_thr) ? v.x * g.drop_scale : 0.f;
        v.y = spot_keep(rk, n + 1, g.drop_thr) ? v.y * g.drop_scale : 0.f;
        v.z = spot_keep(rk, n + 2, g.drop_thr) ? v.z * g.drop_scale : 0.f;
        v.w = spot_keep(rk, n + 3, g.drop_thr) ? v.w * g.drop_scale : 0.f;
    }
    if (g.mask) {
        const float4 k4 = *reinterpret_cast<const float4*>(g.mask + (long)m * g.ldmask + n);
        v.x = k4.x > 0.f ? v.x : 0.f; v.y = k4.y > 0.f ? v.y : 0.f; v.z = k4.z > 0.f ? v.z : 0.f; v.w = k4.w > 0.f ? v.w : 0.f;
    }
    if (g.residual) {
        const float4 r4 = *reinterpret_cast<const float4*>(g.residual + zb * g.strideRes + (long)m * g.ldres + n);
        v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
    }
    if (g.relu == 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    float* cp = g.C + zb * g.strideC + (long)m * g.ldc + n;
    if (g.accumulate) {
        const float4 o = *reinterpret_cast<const float4*>(cp);
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    *reinterpret_cast<float4*>(cp) = v;
}

template <int LA, int LB>
__global__ __launch_bounds__(256, 2) void gemm_ks4_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[2 * KS_TILE_F];
    gemm_ks4_body<LA, LB>(g, blockIdx.x, blockIdx.z, lds);
}

// Two independent problems of the same operand layouts in one launch (the template and the search stream of a fusion
// layer run the same sequence of small products on different weights: one clip's launches last as long as their
// latencies, so two side by side cost what one costs).  Workgroups 0 .. n0-1 belong to the first problem.
template <int LA, int LB>
__global__ __launch_bounds__(256, 2) void gemm_ks4_pair_kernel(GemmArgs g0, GemmArgs g1, int n0) {
    __shared__ __attribute__((aligned(16))) float lds[2 * KS_TILE_F];
    if ((int)blockIdx.x < n0) gemm_ks4_body<LA, LB>(g0, blockIdx.x, 0, lds);
    else gemm_ks4_body<LA, LB>(g1, (int)blockIdx.x - n0, 0, lds);
}

// Workgroup tile for a problem.  Measured on MI355X the two tiles reach the same rate per FLOP (the 64x64 one
// keeps six workgroups per CU in flight against three), so the choice is about quantisation: MFMA work is handed
// out in workgroups, the last round of a grid leaves CUs idle, and a 64x64 workgroup is a quarter of the work.
// The 128x128 tile (fewer epilogues, half the LDS traffic per FLOP) is kept for grids large enough that the tail
// does not matter.
int pick_tile(const GemmArgs& a, int batch, int forced) {
    if (forced == 128) return 2;
    if (forced == 64 || forced == 32) return 1;
    // whole 64x64 tiles of aligned operands run on the ring kernel, where the small tile is never slower (five
    // workgroups per CU hide each other's epilogues; measured equal on N = 2048 outputs, 10-25 % faster on N = 256 and
    // wherever the epilogue reads a mask or a residual)
    if (a.fast && a.precision == 0 && a.M % 64 == 0 && a.N % 64 == 0) return 1;
    auto wgs = [&](int b) { return (long)((a.M + b - 1) / b) * ((a.N + b - 1) / b) * batch * a.splitk; };
    const long CUS = 256;
    const long w128 = wgs(128);
    const long cost128 = 4 * ((w128 + CUS - 1) / CUS), cost64 = (wgs(64) + CUS - 1) / CUS;
    return (cost64 < cost128 || w128 < 2048) ? 1 : 2;
}

template <int LA, int LB>
int launch(const GemmArgs& a, int batch, int tile, bool aligned, hipStream_t stream) {
    const int ts = pick_tile(a, batch, tile), b = 64 * ts;
    const int ntn = (a.N + b - 1) / b, ntm = (a.M + b - 1) / b;
    dim3 grid(ntm * ntn, 1, batch * a.splitk);
    // whole tiles of aligned operands (every k tile whole and inside one segment): the ring kernel, unless the grid is
    // so small that the deep-prefetch loop below is the better fit
    const int nkt_split = ((a.K + BK - 1) / BK + a.splitk - 1) / a.splitk;
    const bool deep = ts == 1 && aligned && a.fast && a.M % 64 == 0 && a.N % 64 == 0 && (long)grid.x * grid.z <= 512 &&
                      nkt_split >= 3 && !a.rowsum;
    const bool ring = a.fast && a.precision == 0 && tile != 32 && !deep && a.M % b == 0 && a.N % b == 0 &&
                      a.lda < (1L << 22) && a.ldb < (1L << 22);
    if (LA == 0 && LB == 0 && a.rowsum) {      // the weight-gradient layout is the only one that asks for row sums
        constexpr bool RS = LA == 0 && LB == 0;
        if (ring) {
            if (ts == 2) gemm_ring_kernel<LA, LB, 2, RS><<<grid, 256, 0, stream>>>(a);
            else gemm_ring_kernel<LA, LB, 1, RS><<<grid, 256, 0, stream>>>(a);
        } else if (ts == 2) {
            if (aligned) gemm_kernel<LA, LB, true, 2, RS><<<grid, 256, 0, stream>>>(a);
            else gemm_kernel<LA, LB, false, 2, RS><<<grid, 256, 0, stream>>>(a);
        } else {
            if (aligned) gemm_kernel<LA, LB, true, 1, RS><<<grid, 256, 0, stream>>>(a);
            else gemm_kernel<LA, LB, false, 1, RS><<<grid, 256, 0, stream>>>(a);
        }
        return 0;
    }
    if (a.precision == 1) {
        if (ts == 2) gemm_kernel<LA, LB, true, 2, false, 1><<<grid, 256, 0, stream>>>(a);
        else gemm_kernel<LA, LB, true, 1, false, 1><<<grid, 256, 0, stream>>>(a);
        return 0;
    }
    // tile = 32 (asked for by the online trackers for one clip's layers): 32x32 tiles with the k step split over the waves
    if (tile == 32) {
        dim3 gks((a.M / KS_TM) * (a.N / KS_TM), 1, batch);
        gemm_ks4_kernel<LA, LB><<<gks, 256, 0, stream>>>(a);
        return 0;
    }
    // a grid of at most two workgroups per CU whose tiles are all interior ones: the deep-prefetch loop (same
    // products in the same order, so the same bits)
    if (deep) {
        gemm_kernel<LA, LB, true, 1, false, 0, true><<<grid, 256, 0, stream>>>(a);
        return 0;
    }
    if (ring) {
        if (ts == 2) gemm_ring_kernel<LA, LB, 2, false><<<grid, 256, 0, stream>>>(a);
        else gemm_ring_kernel<LA, LB, 1, false><<<grid, 256, 0, stream>>>(a);
        return 0;
    }
    if (ts == 2) {
        if (aligned) gemm_kernel<LA, LB, true, 2, false><<<grid, 256, 0, stream>>>(a);
        else gemm_kernel<LA, LB, false, 2, false><<<grid, 256, 0, stream>>>(a);
    } else {
        if (aligned) gemm_kernel<LA, LB, true, 1, false><<<grid, 256, 0, stream>>>(a);
        else gemm_kernel<LA, LB, false, 1, false><<<grid, 256, 0, stream>>>(a);
    }
    return 0;
}

}  // namespace

// Checks a descriptor and fills the kernel arguments.  Returns SPOT_OK with a.M == 0 || a.N == 0 for an empty problem.
static int gemm_prepare(const spot_gemm_desc* d, GemmArgs& a, int& ta, int& tb, bool& aligned) {
    SPOT_CHECK_ARG(d, "gemm: null descriptor");
    SPOT_CHECK_ARG(d->A && d->B && d->C, "gemm: null operand");
    SPOT_CHECK_ARG(d->M >= 0 && d->N >= 0 && d->K >= 0 && d->batch >= 1, "gemm: bad shape M=%d N=%d K=%d batch=%d",
                   d->M, d->N, d->K, d->batch);
    a.M = d->M; a.N = d->N;
    if (d->M == 0 || d->N == 0) return SPOT_OK;
    a.A = d->A; a.B = d->B; a.C = d->C;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc;
    a.strideA = d->strideA; a.strideB = d->strideB; a.strideC = d->strideC;
    a.kseg = d->kseg > 0 ? d->kseg : (d->K > 0 ? d->K : 1);
    a.segA = d->segA; a.segB = d->segB;
    a.splitk = d->splitk > 1 ? d->splitk : 1;
    a.alpha = d->alpha;
    a.accumulate = d->accumulate;
    a.bias = d->bias;
    a.rowbias = d->rowbias; a.rb_rows = d->rb_rows > 0 ? d->rb_rows : 1; a.ldrb = d->ldrb;
    a.rb_cols = d->rb_cols > 0 ? d->rb_cols : d->N;
    a.relu = d->relu;
    a.mask = d->mask; a.ldmask = d->ldmask;
    a.residual = d->residual; a.ldres = d->ldres; a.strideRes = d->strideRes;
    a.rowsum = d->rowsum;
    a.bn_gamma = d->bn_gamma; a.bn_beta = d->bn_beta; a.bn_mean = d->bn_mean; a.bn_var = d->bn_var; a.bn_eps = d->bn_eps;
    SPOT_CHECK_ARG(!a.bn_gamma || (a.bn_beta && a.bn_mean && a.bn_var), "gemm: bn_gamma needs bn_beta, bn_mean and bn_var");
    SPOT_CHECK_ARG(a.relu >= 0 && a.relu <= 2, "gemm: relu=%d (0 off, 1 before the residual, 2 after it)", a.relu);
    SPOT_CHECK_ARG(!a.rowsum || (d->transA && d->transB && d->batch == 1),
                   "gemm: rowsum is for the weight-gradient layout (transA = transB = 1, batch = 1)");
    SPOT_CHECK_ARG(a.kseg % BK == 0 || a.kseg >= a.K, "gemm: kseg=%d must be a multiple of %d", a.kseg, BK);
    SPOT_CHECK_ARG(d->tile == 0 || d->tile == 32 || d->tile == 64 || d->tile == 128, "gemm: tile=%d (0, 32, 64 or 128)", d->tile);
    SPOT_CHECK_ARG(d->drop_p >= 0.f && d->drop_p < 1.f, "gemm: drop_p=%f outside [0,1)", d->drop_p);
    a.drop_thr = spot_drop_threshold(d->drop_p);
    a.drop_seed = d->drop_seed;
    a.drop_scale = 1.f / (1.f - d->drop_p);
    if (a.splitk > 1) {
        SPOT_CHECK_ARG(!a.relu && !a.mask && !a.drop_thr && !a.bn_gamma,
                       "gemm: split-K cannot be combined with relu/mask/dropout/BatchNorm epilogues");
        SPOT_CHECK_ARG(a.accumulate, "gemm: split-K accumulates atomically into C; zero C and set accumulate=1");
    }
    ta = d->transA ? 0 : 1, tb = d->transB ? 0 : 1;   // template arg 1 = k-contiguous
    // 16-byte vector loads need aligned bases/strides and a contiguous extent that is a multiple of 4
    auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    aligned = al(a.A) && al(a.B) && (a.lda % 4 == 0) && (a.ldb % 4 == 0) && (a.strideA % 4 == 0) &&
                   (a.strideB % 4 == 0) && (a.segA % 4 == 0) && (a.segB % 4 == 0);
    aligned = aligned && (ta ? (a.K % 4 == 0 && a.kseg % 4 == 0) : (a.M % 4 == 0));
    aligned = aligned && (tb ? (a.K % 4 == 0 && a.kseg % 4 == 0) : (a.N % 4 == 0));
    a.cvec = al(a.C) && a.ldc % 4 == 0 && a.strideC % 4 == 0 && a.N % 4 == 0 &&
             (!a.bias || al(a.bias)) && (!a.rowbias || (al(a.rowbias) && a.ldrb % 4 == 0 && a.rb_cols % 4 == 0)) &&
             (!a.mask || (al(a.mask) && a.ldmask % 4 == 0)) &&
             (!a.residual || (al(a.residual) && a.ldres % 4 == 0 && a.strideRes % 4 == 0));
    // interior-tile loader: every k tile whole and inside one segment, element offsets within 32 bits
    const long extA = ta ? (long)a.M * a.lda : (long)a.kseg * a.lda + a.M;
    const long extB = tb ? (long)a.N * a.ldb : (long)a.kseg * a.ldb + a.N;
    a.fast = aligned && a.K % BK == 0 && a.kseg % BK == 0 && extA < (1L << 31) && extB < (1L << 31);
    if (d->tile == 32) {
        SPOT_CHECK_ARG(aligned && a.fast && a.cvec && a.splitk == 1 && !a.rowsum && d->precision == 0 && a.M % KS_TM == 0 &&
                           a.N % KS_TM == 0 && a.K % KS_K == 0 && (a.kseg % KS_K == 0 || a.kseg >= a.K),
                       "gemm: tile=32 needs 16-byte aligned operands and epilogue terms, M and N multiples of 32, K (and "
                       "every K segment) a multiple of 128, no split-K, no row sums (M=%d N=%d K=%d)", a.M, a.N, a.K);
    }
    a.precision = d->precision;
    SPOT_CHECK_ARG(a.precision == 0 || a.precision == 1, "gemm: precision=%d (0 exact fp32, 1 bf16x3)", a.precision);
    if (a.precision == 1) {
        // whole tiles only: the bf16x3 loop has no edge handling (either tile size must divide M and N)
        SPOT_CHECK_ARG(a.fast && a.M % 128 == 0 && a.N % 128 == 0 && !a.rowsum,
                       "gemm: bf16x3 products need 16-byte aligned operands, K (and every K segment) a multiple of 32, "
                       "M and N multiples of 128 and no row sums (M=%d N=%d K=%d)", a.M, a.N, a.K);
    }
    return SPOT_OK;
}

extern "C" int spot_gemm_f32(const spot_gemm_desc* d, hipStream_t stream) {
    GemmArgs a;
    int ta, tb;
    bool aligned;
    const int rc = gemm_prepare(d, a, ta, tb, aligned);
    if (rc != SPOT_OK || a.M == 0 || a.N == 0) return rc;
    if (ta && tb) launch<1, 1>(a, d->batch, d->tile, aligned, stream);
    else if (ta && !tb) launch<1, 0>(a, d->batch, d->tile, aligned, stream);
    else if (!ta && tb) launch<0, 1>(a, d->batch, d->tile, aligned, stream);
    else launch<0, 0>(a, d->batch, d->tile, aligned, stream);
    SPOT_CHECK_LAUNCH("gemm");
    return SPOT_OK;
}

extern "C" int spot_gemm_pair_f32(const spot_gemm_desc* d0, const spot_gemm_desc* d1, hipStream_t stream) {
    SPOT_CHECK_ARG(d0 && d1, "gemm_pair: null descriptor");
    GemmArgs a0, a1;
    int ta0, tb0, ta1, tb1;
    bool al0, al1;
    int rc = gemm_prepare(d0, a0, ta0, tb0, al0);
    if (rc != SPOT_OK) return rc;
    rc = gemm_prepare(d1, a1, ta1, tb1, al1);
    if (rc != SPOT_OK) return rc;
    const bool one_launch = d0->tile == 32 && d1->tile == 32 && d0->batch == 1 && d1->batch == 1 && ta0 == ta1 && tb0 == tb1 &&
                            a0.M > 0 && a0.N > 0 && a1.M > 0 && a1.N > 0;
    if (!one_launch) {
        rc = spot_gemm_f32(d0, stream);
        return rc != SPOT_OK ? rc : spot_gemm_f32(d1, stream);
    }
    const int n0 = (a0.M / KS_TM) * (a0.N / KS_TM), n1 = (a1.M / KS_TM) * (a1.N / KS_TM);
    if (ta0 && tb0) gemm_ks4_pair_kernel<1, 1><<<n0 + n1, 256, 0, stream>>>(a0, a1, n0);
    else if (ta0 && !tb0) gemm_ks4_pair_kernel<1, 0><<<n0 + n1, 256, 0, stream>>>(a0, a1, n0);
    else if (!ta0 && tb0) gemm_ks4_pair_kernel<0, 1><<<n0 + n1, 256, 0, stream>>>(a0, a1, n0);
    else gemm_ks4_pair_kernel<0, 0><<<n0 + n1, 256, 0, stream>>>(a0, a1, n0);
    SPOT_CHECK_LAUNCH("gemm_pair");
    return SPOT_OK;
}
