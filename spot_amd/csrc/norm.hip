// LayerNorm over the channel dimension of token rows, forward and backward, and
// the column sum that yields bias gradients.  HBM-bound row kernels: one wave
// per token row, 16 bytes per lane per access.
//
// Reference: the nn.LayerNorm instances norm11..norm23 / norm1, norm2 /
// decoder.norm of ltr/models/neck/transt_fusion.py:127-128,185-190,26.  The
// residual sum that precedes each of them is produced by the GEMM epilogue
// (gemm_f32.hip, `residual`), so this kernel reads the sum once.
#include "common.h"
#include "../../include/spot_hip.h"

namespace {

constexpr int MAXV = 4;   // float4 per lane: C <= 1024

// (bid, nblocks): this workgroup's index among the nblocks that walk the rows (a launch for two tensors hands each
//  workgroup its tensor and its index there)
__device__ __forceinline__ void ln_fwd_body(const float* __restrict__ x, const float* __restrict__ gamma,
                                            const float* __restrict__ beta, float* __restrict__ y,
                                            float* __restrict__ mean, float* __restrict__ rstd, long rows,
                                            int C, float eps, int bid, int nblocks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = C >> 2;
    for (long row = (long)bid * 4 + wave; row < rows; row += (long)nblocks * 4) {
        const float4* xp = reinterpret_cast<const float4*>(x + row * C);
        float4 v[MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int idx = lane + 64 * i;
            v[i] = idx < nv ? xp[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        const float mu = wave_sum(s) / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            if (lane + 64 * i < nv) {
                const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
                q += (a * a + b * b) + (c * c + d * d);
            }
        }
        const float rs = rsqrtf(wave_sum(q) / C + eps);
        float4* yp = reinterpret_cast<float4*>(y + row * C);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int idx = lane + 64 * i;
            if (idx < nv) {
                const float4 g = reinterpret_cast<const float4*>(gamma)[idx];
                const float4 b = reinterpret_cast<const float4*>(beta)[idx];
                yp[idx] = make_float4((v[i].x - mu) * rs * g.x + b.x, (v[i].y - mu) * rs * g.y + b.y,
                                      (v[i].z - mu) * rs * g.z + b.z, (v[i].w - mu) * rs * g.w + b.w);
            }
        }
        if (lane == 0) {
            if (mean) mean[row] = mu;
            if (rstd) rstd[row] = rs;
        }
    }
}

__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, long rows,
                                                     int C, float eps) {
    ln_fwd_body(x, gamma, beta, y, mean, rstd, rows, C, eps, blockIdx.x, gridDim.x);
}

// two independent tensors in one launch (inference: no saved statistics); workgroups 0 .. n0-1 take the first
__global__ __launch_bounds__(256) void ln_fwd_pair_kernel(const float* __restrict__ x0, const float* __restrict__ g0,
                                                          const float* __restrict__ b0, float* __restrict__ y0, long rows0,
                                                          const float* __restrict__ x1, const float* __restrict__ g1,
                                                          const float* __restrict__ b1, float* __restrict__ y1, long rows1,
                                                          int C, float eps0, float eps1, int n0) {
    if ((int)blockIdx.x < n0) ln_fwd_body(x0, g0, b0, y0, nullptr, nullptr, rows0, C, eps0, blockIdx.x, n0);
    else ln_fwd_body(x1, g1, b1, y1, nullptr, nullptr, rows1, C, eps1, (int)blockIdx.x - n0, (int)gridDim.x - n0);
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma;  dgamma = sum dy * xhat;  dbeta = sum dy
// dxd (optional): a second copy of dx behind the dropout mask of (drop_seed, row, column) that a GEMM epilogue applied
// to the block that produced this LayerNorm's input: the gradient that block's weight and input products start from,
// written here instead of by a pass of its own over dx.
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, float* __restrict__ dx,
                                                     float* __restrict__ partial, long rows, int C,
                                                     float* __restrict__ dxd, unsigned drop_thr, unsigned drop_seed,
                                                     float drop_scale) {
    __shared__ float4 red[2][4][64 * MAXV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = C >> 2;
    float4 gm[MAXV], ag[MAXV], ab[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + 64 * i;
        gm[i] = idx < nv ? reinterpret_cast<const float4*>(gamma)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        ab[i] = ag[i];
    }
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const float4* xp = reinterpret_cast<const float4*>(x + row * C);
        const float4* dp = reinterpret_cast<const float4*>(dy + row * C);
        const float mu = mean[row], rs = rstd[row];
        float4 xh[MAXV], g[MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int idx = lane + 64 * i;
            if (idx < nv) {
                const float4 xv = xp[idx], dv = dp[idx];
                xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                g[i] = make_float4(dv.x * gm[i].x, dv.y * gm[i].y, dv.z * gm[i].z, dv.w * gm[i].w);
                s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
                s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
                ag[i].x += dv.x * xh[i].x; ag[i].y += dv.y * xh[i].y; ag[i].z += dv.z * xh[i].z; ag[i].w += dv.w * xh[i].w;
                ab[i].x += dv.x; ab[i].y += dv.y; ab[i].z += dv.z; ab[i].w += dv.w;
            } else {
                xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                g[i] = xh[i];
            }
        }
        const float c1 = wave_sum(s1) / C, c2 = wave_sum(s2) / C;
        float4* op = reinterpret_cast<float4*>(dx + row * C);
        float4* od = dxd ? reinterpret_cast<float4*>(dxd + row * C) : nullptr;
        const unsigned rk = spot_row_key(drop_seed, (unsigned)row);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int idx = lane + 64 * i;
            if (idx < nv) {
                const float4 v = make_float4(rs * (g[i].x - c1 - xh[i].x * c2), rs * (g[i].y - c1 - xh[i].y * c2),
                                             rs * (g[i].z - c1 - xh[i].z * c2), rs * (g[i].w - c1 - xh[i].w * c2));
                op[idx] = v;
                if (od) {
                    const unsigned n = 4u * idx;
                    od[idx] = make_float4(spot_keep(rk, n, drop_thr) ? v.x * drop_scale : 0.f,
                                          spot_keep(rk, n + 1, drop_thr) ? v.y * drop_scale : 0.f,
                                          spot_keep(rk, n + 2, drop_thr) ? v.z * drop_scale : 0.f,
                                          spot_keep(rk, n + 3, drop_thr) ? v.w * drop_scale : 0.f);
                }
            }
        }
    }
    if (!partial) return;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        red[0][wave][lane + 64 * i] = ag[i];
        red[1][wave][lane + 64 * i] = ab[i];
    }
    __syncthreads();
    // per-workgroup partial sums [gridDim.x][2][C]; a second kernel adds them (no atomics: the 2C
    // destination words would be hit by every workgroup)
    float4* pg = reinterpret_cast<float4*>(partial + (size_t)blockIdx.x * 2 * C);
    for (int idx = threadIdx.x; idx < nv; idx += 256) {
        float4 a = red[0][0][idx], b = red[1][0][idx];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 a2 = red[0][w][idx], b2 = red[1][w][idx];
            a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
            b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
        }
        pg[idx] = a;
        pg[nv + idx] = b;
    }
}

// dgamma|dbeta [2C] (+)= sum over the nblk partial rows.  64 columns x 64 rows per workgroup (the 4 waves split
// the rows, 4 loads in flight per lane); row groups are combined by atomic adds (at most nblk/64 per word), so the
// destination is cleared beforehand unless it accumulates.
constexpr int LN_RED_ROWS = 64;
__global__ __launch_bounds__(256) void ln_reduce_kernel(const float* __restrict__ partial, int nblk, int C2,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * LN_RED_ROWS, r1 = min(nblk, r0 + LN_RED_ROWS);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (col < C2) {
        int r = r0 + wave;
        for (; r + 12 < r1; r += 16) {
            s0 += partial[(size_t)r * C2 + col];
            s1 += partial[(size_t)(r + 4) * C2 + col];
            s2 += partial[(size_t)(r + 8) * C2 + col];
            s3 += partial[(size_t)(r + 12) * C2 + col];
        }
        for (; r < r1; r += 4) s0 += partial[(size_t)r * C2 + col];
    }
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && col < C2) {
        const float t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        const int C = C2 >> 1;
        if (col < C) { if (dgamma) atomicAdd(dgamma + col, t); }
        else if (dbeta) atomicAdd(dbeta + col - C, t);
    }
}

// out[n] += sum_m x[m][n]; lanes along n (coalesced rows), waves and blocks along m.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long ld, float* __restrict__ out,
                                                     long rows, int N, long rows_per_block) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float s0 = 0.f, s1 = 0.f;
    if (n < N) {
        long r = r0 + wave;
        for (; r + 4 < r1; r += 8) {
            s0 += x[r * ld + n];
            s1 += x[(r + 4) * ld + n];
        }
        for (; r < r1; r += 4) s0 += x[r * ld + n];
    }
    red[wave][lane] = s0 + s1;
    __syncthreads();
    if (wave == 0 && n < N) atomicAdd(out + n, (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
}

// out[r][c] = sum_b x[(b*R + r)*ld + c]: the sum over clips of a [nb*R, C] row block (what the shared positional
// rows receive from every clip); one thread per four adjacent columns.
__global__ __launch_bounds__(256) void batch_sum_kernel(const float* __restrict__ x, long ld, int nb, long R, int C4,
                                                        float4* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * C4) return;
    const long r = i / C4;
    const int c = (int)(i - r * C4) * 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = 0; b < nb; ++b) {
        const float4 v = *(const float4*)(x + ((long)b * R + r) * ld + c);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    out[i] = a;
}

// out[c] = sum over clips and pixels of x[b][c][:]: one workgroup per channel, a fixed order (no atomics).
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, int B, int C, long HW,
                                                          float* __restrict__ out) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* p = x + ((long)b * C + c) * HW;
        for (long i = threadIdx.x; i < HW; i += 256) acc += p[i];
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[c] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace

extern "C" int spot_batch_sum_f32(const float* x, long ld, int nb, long R, int C, float* out, hipStream_t stream) {
    SPOT_CHECK_ARG(x && out && nb > 0 && R > 0 && C > 0, "batch_sum: bad arguments");
    SPOT_CHECK_ARG(C % 4 == 0 && ld % 4 == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0,
                   "batch_sum: C=%d and the row stride must be multiples of 4, buffers 16-byte aligned", C);
    const long n = R * (C / 4);
    batch_sum_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(x, ld, nb, R, C / 4, (float4*)out);
    SPOT_CHECK_LAUNCH("batch_sum");
    return SPOT_OK;
}

extern "C" int spot_channel_sum_f32(const float* x, int B, int C, long HW, float* out, hipStream_t stream) {
    SPOT_CHECK_ARG(x && out && B > 0 && C > 0 && HW > 0, "channel_sum: bad arguments");
    channel_sum_kernel<<<C, 256, 0, stream>>>(x, B, C, HW, out);
    SPOT_CHECK_LAUNCH("channel_sum");
    return SPOT_OK;
}

extern "C" int spot_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                      float* rstd, long rows, int C, float eps, hipStream_t stream) {
    SPOT_CHECK_ARG(x && gamma && beta && y, "layernorm_fwd: null pointer");
    SPOT_CHECK_ARG(rows >= 0 && C > 0 && C % 4 == 0 && C <= 256 * MAXV, "layernorm_fwd: C=%d must be a multiple of 4, <= %d",
                   C, 256 * MAXV);
    SPOT_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0,
                   "layernorm_fwd: buffers must be 16-byte aligned");
    if (rows == 0) return SPOT_OK;
    long grid = (rows + 3) / 4;
    if (grid > 8192) grid = 8192;
    ln_fwd_kernel<<<(int)grid, 256, 0, stream>>>(x, gamma, beta, y, mean, rstd, rows, C, eps);
    SPOT_CHECK_LAUNCH("layernorm_fwd");
    return SPOT_OK;
}

extern "C" int spot_layernorm_fwd_pair_f32(const float* x0, const float* gamma0, const float* beta0, float* y0, long rows0,
                                           float eps0, const float* x1, const float* gamma1, const float* beta1, float* y1,
                                           long rows1, float eps1, int C, hipStream_t stream) {
    SPOT_CHECK_ARG(x0 && gamma0 && beta0 && y0 && x1 && gamma1 && beta1 && y1, "layernorm_fwd_pair: null pointer");
    SPOT_CHECK_ARG(rows0 > 0 && rows1 > 0 && C > 0 && C % 4 == 0 && C <= 256 * MAXV,
                   "layernorm_fwd_pair: rows must be positive and C=%d a multiple of 4, <= %d", C, 256 * MAXV);
    SPOT_CHECK_ARG((((uintptr_t)x0 | (uintptr_t)y0 | (uintptr_t)gamma0 | (uintptr_t)beta0 | (uintptr_t)x1 | (uintptr_t)y1 |
                     (uintptr_t)gamma1 | (uintptr_t)beta1) & 15) == 0, "layernorm_fwd_pair: buffers must be 16-byte aligned");
    long n0 = (rows0 + 3) / 4, n1 = (rows1 + 3) / 4;
    if (n0 > 4096) n0 = 4096;
    if (n1 > 4096) n1 = 4096;
    ln_fwd_pair_kernel<<<(int)(n0 + n1), 256, 0, stream>>>(x0, gamma0, beta0, y0, rows0, x1, gamma1, beta1, y1, rows1, C, eps0,
                                                          eps1, (int)n0);
    SPOT_CHECK_LAUNCH("layernorm_fwd_pair");
    return SPOT_OK;
}

static int ln_bwd_grid(long rows) {
    long grid = (rows + 3) / 4;
    return (int)(grid > 1024 ? 1024 : grid);
}

extern "C" long spot_layernorm_bwd_ws_floats(long rows, int C) { return (long)ln_bwd_grid(rows) * 2 * C; }

extern "C" int spot_layernorm_bwd_f32(const float* dy, const float* x, const float* mean, const float* rstd,
                                      const float* gamma, float* dx, float* dgamma, float* dbeta, float* ws,
                                      long rows, int C, int accumulate, float* dx_dropped, float drop_p, unsigned drop_seed,
                                      hipStream_t stream) {
    SPOT_CHECK_ARG(rows >= 0 && C > 0 && C % 4 == 0 && C <= 256 * MAXV, "layernorm_bwd: C=%d must be a multiple of 4, <= %d",
                   C, 256 * MAXV);
    if (rows == 0) {
        if (accumulate) return SPOT_OK;
        if (dgamma) (void)hipMemsetAsync(dgamma, 0, C * sizeof(float), stream);
        if (dbeta) (void)hipMemsetAsync(dbeta, 0, C * sizeof(float), stream);
        return SPOT_OK;
    }
    SPOT_CHECK_ARG(dy && x && mean && rstd && gamma && dx, "layernorm_bwd: null pointer");
    SPOT_CHECK_ARG(!(dgamma || dbeta) || ws, "layernorm_bwd: dgamma/dbeta need the partial-sum workspace");
    SPOT_CHECK_ARG((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)gamma | (uintptr_t)ws) & 15) == 0,
                   "layernorm_bwd: buffers must be 16-byte aligned");
    const int grid = ln_bwd_grid(rows);
    const bool want = dgamma || dbeta;
    SPOT_CHECK_ARG(!dx_dropped || (drop_p > 0.f && drop_p < 1.f && rows < (1L << 32) && ((uintptr_t)dx_dropped & 15) == 0),
                   "layernorm_bwd: the dropped copy needs 0 < drop_p < 1, fewer than 2^32 rows and a 16-byte aligned buffer");
    ln_bwd_kernel<<<grid, 256, 0, stream>>>(dy, x, mean, rstd, gamma, dx, want ? ws : nullptr, rows, C, dx_dropped,
                                            dx_dropped ? spot_drop_threshold(drop_p) : 0u, drop_seed,
                                            dx_dropped ? 1.f / (1.f - drop_p) : 1.f);
    SPOT_CHECK_LAUNCH("layernorm_bwd");
    if (want) {
        if (!accumulate) {
            if (dgamma) (void)hipMemsetAsync(dgamma, 0, C * sizeof(float), stream);
            if (dbeta) (void)hipMemsetAsync(dbeta, 0, C * sizeof(float), stream);
        }
        ln_reduce_kernel<<<dim3((2 * C + 63) / 64, (grid + LN_RED_ROWS - 1) / LN_RED_ROWS), 256, 0, stream>>>(
            ws, grid, 2 * C, dgamma, dbeta);
        SPOT_CHECK_LAUNCH("layernorm_bwd(reduce)");
    }
    return SPOT_OK;
}

extern "C" int spot_colsum_f32(const float* x, long ld, float* out, long rows, int N, hipStream_t stream) {
    SPOT_CHECK_ARG(x && out && rows >= 0 && N > 0, "colsum: bad arguments");
    if (rows == 0) return SPOT_OK;
    const int gx = (N + 63) / 64;
    long gy = (rows + 255) / 256;
    const long cap = 1024 / gx > 0 ? 1024 / gx : 1;
    if (gy > cap) gy = cap;
    const long rpb = (rows + gy - 1) / gy;
    gy = (rows + rpb - 1) / rpb;
    colsum_kernel<<<dim3(gx, (unsigned)gy), 256, 0, stream>>>(x, ld, out, rows, N, rpb);
    SPOT_CHECK_LAUNCH("colsum");
    return SPOT_OK;
}
