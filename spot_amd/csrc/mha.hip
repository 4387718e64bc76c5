// Fused multi-head attention core on the matrix cores, head dim 32: exact fp32 products, or bf16
// products with fp32 softmax and accumulation (the *_bf16 entry points).
//
// Replaces the QK^T -> key-padding mask -> softmax -> (dropout) -> PV middle of
// nn.MultiheadAttention as the fusion layers call it
// (ltr/models/neck/transt_fusion.py:145-148,212-213,218-219,223-230): ECA
// self-attention (256x256, 1024x1024 tokens) and CFA template x search
// cross-attention (256x1024, 1024x256).  The in/out projections are GEMMs
// (gemm_f32.hip); the B*8*Lq*Lk score tensor the reference materialises never
// leaves the chip.
//
// One 256-thread workgroup = 128 queries of one (clip, head); each wave owns 32
// queries.  Keys/values stream through LDS in 64-key tiles (register-staged,
// double-buffered).  Per 32-key block a wave issues
//   S^T = K Q^T   16 x v_mfma_f32_32x32x2_f32  (keys on the accumulator rows, queries on the lanes)
//   O^T += V^T P^T 16 x v_mfma_f32_32x32x2_f32 (the S^T accumulator IS the B operand: no shuffle, no LDS)
// With queries on the lanes the online-softmax state (running max, sum, rescale
// factor) is one scalar per lane and the row reductions are 15 in-register ops
// plus one exchange with lane^32.
#include <mutex>

#include "common.h"
#include "../../include/spot_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// One 32x32 (+)= [32 x 32] x [32 x 32] product per call.  A lane's operand fragment is 16 values
// of its row/column: with the lane-half h owning reduction indices 16h .. 16h+15 on BOTH operands
// (the MFMA only needs the two operands to agree on which index sits where), the fp32 path issues
// 16 v_mfma_f32_32x32x2_f32 (step j pairs index j of either half) and the bf16 path converts the
// same 16 values to two 8-wide bf16 operands and issues 2 v_mfma_f32_32x32x16_bf16; accumulation
// is fp32 in both.
template <bool BF16> struct Frag;
template <> struct Frag<false> {
    float f[16];
    __device__ __forceinline__ void set(int i, float x) { f[i] = x; }
};
template <> struct Frag<true> {
    bf16x8 v[2];
    __device__ __forceinline__ void set(int i, float x) { v[i >> 3][i & 7] = (__bf16)x; }
};
__device__ __forceinline__ void mma(const Frag<false>& a, const Frag<false>& b, f32x16& acc) {
#pragma unroll
    for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.f[j], b.f[j], acc, 0, 0, 0);
}
__device__ __forceinline__ void mma(const Frag<true>& a, const Frag<true>& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v[0], b.v[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v[1], b.v[1], acc, 0, 0, 0);
}
// fragment = 16 consecutive floats of an LDS / global row (16-byte aligned), times mul
template <bool BF16>
__device__ __forceinline__ void load_frag(Frag<BF16>& fr, const float* p, float mul = 1.f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(p + 4 * i);
        fr.set(4 * i, v.x * mul); fr.set(4 * i + 1, v.y * mul); fr.set(4 * i + 2, v.z * mul); fr.set(4 * i + 3, v.w * mul);
    }
}
// fragment = the lane's column of a row-major LDS tile at the 16 rows its accumulator registers map to
template <bool BF16>
__device__ __forceinline__ void load_frag_rows(Frag<BF16>& fr, const float* tile, int ld, int h, int col) {
#pragma unroll
    for (int e = 0; e < 16; ++e) fr.set(e, tile[((e & 3) + 8 * (e >> 2) + 4 * h) * ld + col]);
}
template <bool BF16>
__device__ __forceinline__ void acc_frag(Frag<BF16>& fr, const f32x16& a) {
#pragma unroll
    for (int e = 0; e < 16; ++e) fr.set(e, a[e]);
}

constexpr int DH = 32;        // head dim
constexpr int KT = 64;        // keys per LDS tile
constexpr int KP = DH + 4;    // padded LDS row (floats)

__device__ __forceinline__ int acc_row(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

// attention-probability dropout: keep(q, key) for (clip, head) stream key sbh; Lq, Lk <= 65536.
// The fp32 matrix instruction does not overlap with vector ALU work on its SIMD, so every vector
// instruction in the attention loops costs wall time: the hash is built from full-rate operations only
// (24-bit multiply-add, shifts, xors; two rounds) and ONE hash serves the two keys 2j, 2j+1 of a query:
// its low 16 bits decide the even key, its high 16 bits the odd one (keep iff half >= thr16, thr16 =
// round-down of p * 65536).  In the forward kernels a lane holds both keys of a pair, so the hash work per
// element halves; in the backward kernel (keys on the lanes) a lane shifts its half into place.  Drop rate,
// neighbour correlations and row/column count dispersion of the masks match a 32-bit mixer's per-element draw.
__device__ __forceinline__ unsigned attn_stream_key(unsigned seed, unsigned bh) { return spot_mix32(bh * 0x9E3779B1U ^ seed); }
__device__ __forceinline__ unsigned attn_mix(unsigned x) {
    x ^= x >> 15;
    x = (x & 0xffffffu) * 0x966bafu + (x >> 10);
    x ^= x >> 15;
    x = (x & 0xffffffu) * 0xf9341du + (x >> 10);
    x ^= x >> 13;
    return x;
}
__device__ __forceinline__ bool attn_keep(unsigned sbh, unsigned q, unsigned key, unsigned thr16) {
    const unsigned bits = attn_mix(((q << 16) | (key & ~1u)) ^ sbh);
    return ((key & 1u) ? bits >> 16 : bits & 0xffffu) >= thr16;
}
// Keep bits for the backward pass.  A forward pass that is followed by a backward writes the mask it drew as one bit
// per (query, key): 32-bit word [clip*head][query tile of 32][key tile of 32][slot] holds the 32 keys of the tile for
// one query, key k of the tile at bit k.  The query's slot is attn_bits_slot(query in tile): the order in which the
// backward's accumulator rows meet the queries, so that words 2e and 2e + 1 together are the 64-lane mask (keys on the
// lanes, lane half h on query acc_row(e, h)) of its element e.  The backward reads them with scalar loads and spends
// ONE select per element, with the mask in scalar registers, instead of the hash.
__device__ __forceinline__ int attn_bits_slot(int q31) { return 2 * ((q31 & 3) + 4 * (q31 >> 3)) + ((q31 >> 2) & 1); }
__device__ __forceinline__ long attn_bits_words(int Lq, int Lk) { return (long)((Lq + 31) / 32) * ((Lk + 31) / 32) * 32; }
// 16-bit threshold of the pair hash from the library-wide 32-bit one (0 = dropout off; a rate below 2^-16 rounds up)
static inline unsigned attn_threshold(float p) {
    const unsigned t = spot_drop_threshold(p);
    return t == 0 ? 0u : ((t >> 16) ? (t >> 16) : 1u);
}

typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

// Online-softmax step on one 32-key score block held as S^T (keys on the accumulator rows, the
// lane's query fixed).  Scores arrive in the log2 domain (Q was pre-multiplied by scale*log2 e).
// The reference value m_ref that the exponentials are taken against is NOT the exact running maximum:
// it moves (and the running sums are rescaled) only when some query of the wave meets a score more than
// LAZY_TAU above its reference, so that step is rare after the first blocks and the common path is
// subtract, exp2, add.  Probabilities relative to the reference stay below 2^LAZY_TAU and the sums
// below 2^(LAZY_TAU + 16): the normalised output and the log-sum-exp do not depend on the reference.
// check = true (workgroup-uniform) applies the bounds / key-padding tests first.
constexpr float LAZY_TAU = 8.f;
// packed = 2 * packed + (this lane's bit of mask): one add with a comparison's lane mask as carry-in (the compiler's
// own sequence for packing comparison results is a select, a shift and an or per bit)
__device__ __forceinline__ void push_bit(unsigned& packed, unsigned long long mask) {
    unsigned long long carry_out;
    asm("v_addc_co_u32 %0, %1, %0, %0, %2" : "+v"(packed), "=s"(carry_out) : "s"(mask));
}
// DROPC: what the caller knows at compile time about dropout: -1 nothing (decided per call from drop_thr / keep_out),
// 0 off, 1 on, 2 on with keep bits (keep_out: the word of the lane's query in the first key tile; null, for the whole
// wave, when its 32 queries all lie past the end)
template <int DROPC = -1>
__device__ __forceinline__ void softmax_block(bool check, f32x16& s, f32x16& o, float& m_ref, float& l_run, int key0, int h, int Lk,
                                              const unsigned char* mb, unsigned qs, unsigned drop_thr,
                                              unsigned* keep_out = nullptr) {
    if (check) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int key = key0 + acc_row(e, h);
            const bool dead = key >= Lk || (mb && mb[key]);
            s[e] = dead ? -INFINITY : s[e];
        }
    }
    float mx = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
    for (int e = 3; e < 15; e += 2) mx = fmaxf(fmaxf(mx, s[e]), s[e + 1]);
    mx = fmaxf(mx, s[15]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (__builtin_amdgcn_ballot_w64(mx > m_ref + LAZY_TAU)) {      // (-inf + tau = -inf: the first live block always enters)
        const float m_new = fmaxf(m_ref, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_ref - (m_new == -INFINITY ? 0.f : m_new));   // m_ref = -inf -> 0
        l_run *= alpha;
        o *= alpha;
        m_ref = m_new;
    }
    const float m_use = m_ref == -INFINITY ? 0.f : m_ref;
    s = s - m_use;                                  // packed subtracts
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = __builtin_amdgcn_exp2f(s[e]);
    {                                               // row sum as a tree of packed adds
        const f32x8 a8 = s.lo + s.hi;
        const f32x4v a4 = a8.lo + a8.hi;
        const f32x2v a2 = a4.lo + a4.hi;
        l_run += a2.x + a2.y;
    }
    // dropout acts on the normalised probabilities: the row sum above stays undropped, and the factor 1 / (1 - p) of the
    // kept ones is applied once, to the finished output rows
    if (DROPC < 0 ? drop_thr != 0 : DROPC > 0) {
        // keys of accumulator elements (e, e + 1), e even: key0 + 4h + r, r = acc_row(e, 0) even, and r + 1
        const unsigned x0 = qs ^ (unsigned)(key0 + 4 * h);   // low 16 bits: key
        unsigned long long keep[16];               // lane masks of the comparisons
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            const unsigned bits = attn_mix(x0 ^ (unsigned)acc_row(e, 0));
            const bool k0 = (bits & 0xffffu) >= drop_thr, k1 = (bits >> 16) >= drop_thr;
            keep[e] = __builtin_amdgcn_ballot_w64(k0);
            keep[e + 1] = __builtin_amdgcn_ballot_w64(k1);
            s[e] = k0 ? s[e] : 0.f;
            s[e + 1] = k1 ? s[e + 1] : 0.f;
        }
        if ((DROPC < 0 || DROPC == 2) && keep_out) {   // a backward follows
            unsigned packed = 0;                       // element e at bit e: nibble j holds keys 8j + 4h .. + 3
#pragma unroll
            for (int e = 15; e >= 0; --e) push_bit(packed, keep[e]);
            packed = (packed | (packed << 8)) & 0x00ff00ffu;
            packed = (packed | (packed << 4)) & 0x0f0f0f0fu;       // nibble j now in the low half of byte j
            packed <<= 4 * h;
            packed |= __shfl_xor(packed, 32);          // the keys of the other lane half
            if (h == 0) keep_out[(key0 >> 5) * 32] = packed;
        }
    }
}

// One call of the forward kernels' arguments (a launch may carry two, see mha_fwd_pair_kernel)
struct MhaFwdArgs {
    const float* Q; long ldq; const float* K; long ldk; const float* V; long ldv;
    float* O; long ldo; float* LSE; const unsigned char* kpm;
    int Lq, Lk, H; float scale; unsigned drop_thr, drop_seed; float drop_scale;
    int nsplit; float* part_o; float* part_ml;
    unsigned* keep_bits;      // dropout keep bits for the backward pass, or null (layout: attn_bits_words)
};

// (bx, head, b, nb): the workgroup's index along the (query tile, key range) axis, its head, clip and the clip count
template <bool BF16>
__device__ __forceinline__ void mha_fwd_body(const MhaFwdArgs& A, int bx, int head, int b, int nb, float* lds) {
    const float* __restrict__ Q = A.Q; const float* __restrict__ K = A.K; const float* __restrict__ V = A.V;
    float* __restrict__ O = A.O; float* __restrict__ LSE = A.LSE; const unsigned char* __restrict__ kpm = A.kpm;
    const long ldq = A.ldq, ldk = A.ldk, ldv = A.ldv, ldo = A.ldo;
    const int Lq = A.Lq, Lk = A.Lk, H = A.H, nsplit = A.nsplit;
    const float scale = A.scale, drop_scale = A.drop_scale;
    const unsigned drop_thr = A.drop_thr, drop_seed = A.drop_seed;
    float* __restrict__ part_o = A.part_o; float* __restrict__ part_ml = A.part_ml;
    auto Ks = [&](int buf) { return lds + buf * 2 * KT * KP; };
    auto Vs = [&](int buf) { return lds + buf * 2 * KT * KP + KT * KP; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    // with nsplit > 1 the keys are cut in nsplit ranges of whole tiles, one workgroup each: bx = tile * nsplit + s
    const int split = bx % nsplit;
    const int q = (bx / nsplit) * 128 + wave * 32 + l31;
    const bool qok = q < Lq;
    const float c = scale * 1.4426950408889634f;   // Q is pre-multiplied by c: scores live in the log2 domain
    const unsigned qs = ((unsigned)q << 16) ^ attn_stream_key(drop_seed, (unsigned)(b * H + head));

    // Q fragment: lane holds c * Q[q][16h .. 16h+15]
    Frag<BF16> qf;
    load_frag(qf, Q + ((long)b * Lq + (qok ? q : 0)) * ldq + head * DH + 16 * h, c);
    f32x16 o;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    // the keep-bit word of this lane's query in the first key tile (null: no backward follows, or the wave's queries all
    // lie past the end)
    unsigned* const keep_lane = A.keep_bits && (q & ~31) < Lq
        ? A.keep_bits + ((long)b * H + head) * attn_bits_words(Lq, Lk) + (long)(q >> 5) * ((Lk + 31) / 32) * 32 +
              attn_bits_slot(q & 31)
        : nullptr;

    const float* Kb = K + (long)b * Lk * ldk + head * DH;
    const float* Vb = V + (long)b * Lk * ldv + head * DH;
    const unsigned char* mb = kpm ? kpm + (long)b * Lk : nullptr;

    float4 sk[2], sv[2];
    auto fetch = [&](int t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i;
            const int key = t * KT + (f >> 3), d4 = (f & 7) << 2;
            if (key < Lk) {
                sk[i] = *reinterpret_cast<const float4*>(Kb + (long)key * ldk + d4);
                sv[i] = *reinterpret_cast<const float4*>(Vb + (long)key * ldv + d4);
            } else {
                sk[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                sv[i] = sk[i];
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i;
            *reinterpret_cast<float4*>(Ks(buf) + (f >> 3) * KP + ((f & 7) << 2)) = sk[i];
            *reinterpret_cast<float4*>(Vs(buf) + (f >> 3) * KP + ((f & 7) << 2)) = sv[i];
        }
    };

    const int ntile_all = (Lk + KT - 1) / KT, per = (ntile_all + nsplit - 1) / nsplit;
    const int t0 = split * per, ntile = min(ntile_all, t0 + per);
    if (t0 < ntile) {
        fetch(t0);
        stash(0);
    }
    __syncthreads();
    for (int t = t0; t < ntile; ++t) {
        const int cur = (t - t0) & 1;
        const bool more = t + 1 < ntile;
        if (more) fetch(t + 1);
#pragma unroll
        for (int sub = 0; sub < KT / 32; ++sub) {
            const int key0 = t * KT + sub * 32;
            if (key0 >= Lk) break;
            const float* ks = Ks(cur) + sub * 32 * KP;
            const float* vs = Vs(cur) + sub * 32 * KP;
            // ---- S^T = K Q^T
            f32x16 s;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = 0.f;
            {
                Frag<BF16> kf;
                load_frag(kf, ks + l31 * KP + 16 * h);
                mma(kf, qf, s);
            }
            // ---- mask, online softmax (per lane = per query)
            softmax_block(mb || key0 + 32 > Lk, s, o, m_run, l_run, key0, h, Lk, mb, qs, drop_thr, keep_lane);
            // ---- O^T += V^T P^T   (reduction slot e of half h is key acc_row(e,h): the S^T accumulator IS the operand)
            {
                Frag<BF16> vf, pf;
                load_frag_rows(vf, vs, KP, h, l31);
                acc_frag(pf, s);
                mma(vf, pf, o);
            }
        }
        if (more) stash(cur ^ 1);
        __syncthreads();
    }
    // each half summed its own 16 keys per block
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float kept_scale = drop_thr ? drop_scale : 1.f;
    if (part_o) {                // this key range's unnormalised sums, its running maximum (log2 domain) and row sum
        if (qok) {
            float* op = part_o + (((long)split * nb + b) * Lq + q) * ((long)H * DH) + head * DH;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *reinterpret_cast<float4*>(op + 8 * gq + 4 * h) =
                    make_float4(o[4 * gq] * kept_scale, o[4 * gq + 1] * kept_scale, o[4 * gq + 2] * kept_scale, o[4 * gq + 3] * kept_scale);
            if (h == 0) {
                float* mp = part_ml + ((((long)split * nb + b) * H + head) * Lq + q) * 2;
                mp[0] = m_run;
                mp[1] = l_tot;
            }
        }
        return;
    }
    const float inv = kept_scale / l_tot;
    if (qok) {
        float* op = O + ((long)b * Lq + q) * ldo + head * DH;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            *reinterpret_cast<float4*>(op + 8 * gq + 4 * h) =
                make_float4(o[4 * gq] * inv, o[4 * gq + 1] * inv, o[4 * gq + 2] * inv, o[4 * gq + 3] * inv);
        if (LSE && h == 0) LSE[((long)b * H + head) * Lq + q] = m_run * 0.6931471805599453f + logf(l_tot);
    }
}

template <bool BF16>
__global__ __launch_bounds__(256, 3) void mha_fwd_kernel(MhaFwdArgs A) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * KT * KP];
    mha_fwd_body<BF16>(A, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.z, lds);
}

// Two independent calls of the same head count in one launch (one clip each: the template and the search stream of a
// fusion layer); workgroups 0 .. n0-1 along x belong to the first.
__global__ __launch_bounds__(256, 3) void mha_fwd_pair_kernel(MhaFwdArgs A0, MhaFwdArgs A1, int n0) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * KT * KP];
    if ((int)blockIdx.x < n0) mha_fwd_body<false>(A0, blockIdx.x, blockIdx.y, 0, 1, lds);
    else mha_fwd_body<false>(A1, (int)blockIdx.x - n0, blockIdx.y, 0, 1, lds);
}

// ---------------------------------------------------------------------------
// Two-group variant of the forward kernel.  A 512-thread workgroup owns 256
// queries; waves w and w+4 sit on the same SIMD.  Every wave runs the same
// alternating program
//     M(-1): S(0)      V(0): softmax(0)      M(0): O += P(0) V(0), S(1)      V(1): softmax(1)   ...
// with a workgroup barrier after each phase, and waves 4..7 run it one phase
// late.  So on every SIMD one wave is always in a matrix phase while its partner
// is in a vector (softmax / dropout-hash) phase: the matrix pipe and the VALU
// work concurrently instead of the two waves drifting into lockstep and taking
// turns idling both.  K/V stream through a ring of three 32-key LDS tiles: tile
// j is fetched to registers in global phase 2j-2, written to LDS in 2j-1 and
// first read in 2j; its slot's previous tenant j-3 was last read in phase 2j-3.
// ---------------------------------------------------------------------------
template <int DROPC>
__global__ __launch_bounds__(512) void mha_fwd_pp_kernel(
    const float* __restrict__ Q, long ldq, const float* __restrict__ K, long ldk, const float* __restrict__ V, long ldv,
    float* __restrict__ O, long ldo, float* __restrict__ LSE, const unsigned char* __restrict__ kpm,
    int Lq, int Lk, int H, float scale, unsigned drop_thr, unsigned drop_seed, float drop_scale, unsigned* keep_bits) {
    __shared__ __attribute__((aligned(16))) float lds[3 * 2 * 32 * KP];
    auto Ks = [&](int j) { return lds + (j % 3) * 2 * 32 * KP; };
    auto Vs = [&](int j) { return lds + (j % 3) * 2 * 32 * KP + 32 * KP; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = wave >> 2, h = lane >> 5, l31 = lane & 31;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q = blockIdx.x * 256 + wave * 32 + l31;
    const bool qok = q < Lq;
    const float c = scale * 1.4426950408889634f;
    const unsigned qs = ((unsigned)q << 16) ^ attn_stream_key(drop_seed, (unsigned)(b * H + head));

    // Q fragment: lane holds c * Q[q][16h .. 16h+15]
    Frag<false> qf;
    load_frag(qf, Q + ((long)b * Lq + (qok ? q : 0)) * ldq + head * DH + 16 * h, c);
    f32x16 o, sA, sB;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    unsigned* const keep_lane = keep_bits && (q & ~31) < Lq
        ? keep_bits + ((long)b * H + head) * attn_bits_words(Lq, Lk) + (long)(q >> 5) * ((Lk + 31) / 32) * 32 +
              attn_bits_slot(q & 31)
        : nullptr;

    const unsigned char* mb = kpm ? kpm + (long)b * Lk : nullptr;
    const int n = (Lk + 31) / 32;

    // loader role of this thread: threads 0..255 carry K rows, 256..511 carry V rows, one float4 each per tile
    const int ld_row = (tid & 255) >> 3, ld_d4 = (tid & 7) << 2;
    const bool ld_v = tid >= 256;
    const float* ld_base = ld_v ? V + (long)b * Lk * ldv + head * DH + ld_d4 : K + (long)b * Lk * ldk + head * DH + ld_d4;
    const long ld_stride = ld_v ? ldv : ldk;
    float4 st = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](int j) {
        const int key = j * 32 + ld_row;
        st = key < Lk ? *reinterpret_cast<const float4*>(ld_base + (long)key * ld_stride) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto stash = [&](int j) { *reinterpret_cast<float4*>((ld_v ? Vs(j) : Ks(j)) + ld_row * KP + ld_d4) = st; };
    auto duty = [&](int gp) {     // global phase gp
        if ((gp & 1) == 0) {
            const int j = (gp >> 1) + 1;
            if (j >= 2 && j < n) fetch(j);
        } else {
            const int j = (gp + 1) >> 1;
            if (j >= 2 && j < n) stash(j);
        }
    };
    auto score = [&](int j, f32x16& s) {      // S^T(j) = K_j Q^T
        Frag<false> kf;
        load_frag(kf, Ks(j) + l31 * KP + 16 * h);
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
        mma(kf, qf, s);
    };
    auto accumulate = [&](int j, const f32x16& s) {   // O^T += V_j^T P^T
        Frag<false> vf, pf;
        load_frag_rows(vf, Vs(j), KP, h, l31);
        acc_frag(pf, s);
        mma(vf, pf, o);
    };
    auto softmax = [&](int j, f32x16& s) {
        const int key0 = j * 32;
        softmax_block<DROPC>(mb || key0 + 32 > Lk, s, o, m_run, l_run, key0, h, Lk, mb, qs, drop_thr, keep_lane);
    };

    fetch(0);
    stash(0);
    if (n > 1) {
        fetch(1);
        stash(1);
    }
    __syncthreads();
    if (grp) __syncthreads();                 // waves 4..7 run one phase behind
    int gp = grp;
    duty(gp++);
    score(0, sA);
    __syncthreads();
    for (int i = 0; i < n; i += 2) {
        duty(gp++);
        softmax(i, sA);
        __syncthreads();
        duty(gp++);
        accumulate(i, sA);
        if (i + 1 < n) score(i + 1, sB);
        __syncthreads();
        if (i + 1 < n) {
            duty(gp++);
            softmax(i + 1, sB);
            __syncthreads();
            duty(gp++);
            accumulate(i + 1, sB);
            if (i + 2 < n) score(i + 2, sA);
            __syncthreads();
        }
    }
    if (!grp) __syncthreads();
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = (DROPC ? drop_scale : 1.f) / l_tot;
    if (qok) {
        float* op = O + ((long)b * Lq + q) * ldo + head * DH;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            *reinterpret_cast<float4*>(op + 8 * gq + 4 * h) =
                make_float4(o[4 * gq] * inv, o[4 * gq + 1] * inv, o[4 * gq + 2] * inv, o[4 * gq + 3] * inv);
        if (LSE && h == 0) LSE[((long)b * H + head) * Lq + q] = m_run * 0.6931471805599453f + logf(l_tot);
    }
}

}  // namespace

static int mha_fwd_impl(bool bf16, const float* q, long ldq, const float* k, long ldk, const float* v, long ldv,
                        float* out, long ldo, float* lse, const unsigned char* key_padding_mask, int B, int Lq, int Lk,
                        int nhead, int head_dim, float scale, float drop_p, unsigned drop_seed, unsigned* drop_bits,
                        hipStream_t stream) {
    SPOT_CHECK_ARG(q && k && v && out, "mha_fwd: null pointer");
    SPOT_CHECK_ARG(head_dim == DH, "mha_fwd: head_dim=%d unsupported (this build handles 32)", head_dim);
    SPOT_CHECK_ARG(B >= 0 && Lq > 0 && Lk > 0 && nhead > 0, "mha_fwd: bad shape B=%d Lq=%d Lk=%d heads=%d", B, Lq, Lk, nhead);
    SPOT_CHECK_ARG(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0 &&
                       (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0,
                   "mha_fwd: operands must be 16-byte aligned with row strides that are multiples of 4");
    SPOT_CHECK_ARG(B <= 65535 && nhead <= 65535, "mha_fwd: B and heads must fit a grid dimension");
    SPOT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "mha_fwd: drop_p=%f outside [0,1)", drop_p);
    SPOT_CHECK_ARG(drop_p == 0.f || (Lq <= 65536 && Lk <= 65536), "mha_fwd: dropout needs Lq, Lk <= 65536");
    if (B == 0) return SPOT_OK;
    const unsigned thr = attn_threshold(drop_p);
    const float dscale = 1.f / (1.f - drop_p);
    if (bf16) {
        // the matrix work shrinks 16-fold: the loop is vector-bound and gains nothing from pairing matrix with vector phases
        dim3 grid((Lq + 127) / 128, nhead, B);
        mha_fwd_kernel<true><<<grid, 256, 0, stream>>>(MhaFwdArgs{q, ldq, k, ldk, v, ldv, out, ldo, lse, key_padding_mask, Lq, Lk,
                                                       nhead, scale, thr, drop_seed, dscale, 1, nullptr, nullptr, drop_bits});
    } else if (Lq >= 512 && (long)((Lq + 255) / 256) * nhead * B >= 512) {
        // 256-query workgroups need enough of them to fill the chip; short query streams keep the 128-query kernel
        dim3 grid((Lq + 255) / 256, nhead, B);
#define SPOT_MHA_PP(D)                                                                                                    \
    mha_fwd_pp_kernel<D><<<grid, 512, 0, stream>>>(q, ldq, k, ldk, v, ldv, out, ldo, lse, key_padding_mask, Lq, Lk, nhead, \
                                                   scale, thr, drop_seed, dscale, drop_bits)
        if (!thr) SPOT_MHA_PP(0);
        else if (drop_bits) SPOT_MHA_PP(2);
        else SPOT_MHA_PP(1);
#undef SPOT_MHA_PP
    } else {
        dim3 grid((Lq + 127) / 128, nhead, B);
        mha_fwd_kernel<false><<<grid, 256, 0, stream>>>(MhaFwdArgs{q, ldq, k, ldk, v, ldv, out, ldo, lse, key_padding_mask, Lq, Lk,
                                                        nhead, scale, thr, drop_seed, dscale, 1, nullptr, nullptr, drop_bits});
    }
    SPOT_CHECK_LAUNCH("mha_fwd");
    return SPOT_OK;
}

extern "C" long spot_mha_drop_bits_words(int B, int nhead, int Lq, int Lk) {
    if (B <= 0 || nhead <= 0 || Lq <= 0 || Lk <= 0) return 0;
    return (long)B * nhead * ((long)((Lq + 31) / 32) * ((Lk + 31) / 32) * 32);
}

extern "C" int spot_mha_fwd_f32(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv,
                                float* out, long ldo, float* lse, const unsigned char* key_padding_mask, int B,
                                int Lq, int Lk, int nhead, int head_dim, float scale, float drop_p, unsigned drop_seed,
                                unsigned* drop_bits, hipStream_t stream) {
    return mha_fwd_impl(false, q, ldq, k, ldk, v, ldv, out, ldo, lse, key_padding_mask, B, Lq, Lk, nhead, head_dim, scale,
                        drop_p, drop_seed, drop_bits, stream);
}

extern "C" int spot_mha_fwd_bf16(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv,
                                 float* out, long ldo, float* lse, const unsigned char* key_padding_mask, int B,
                                 int Lq, int Lk, int nhead, int head_dim, float scale, float drop_p, unsigned drop_seed,
                                 unsigned* drop_bits, hipStream_t stream) {
    return mha_fwd_impl(true, q, ldq, k, ldk, v, ldv, out, ldo, lse, key_padding_mask, B, Lq, Lk, nhead, head_dim, scale,
                        drop_p, drop_seed, drop_bits, stream);
}

// ===========================================================================
// Backward.  One workgroup owns 128 keys of one (clip, head): each wave keeps
// the dK and dV tiles of its 32 keys in accumulators for the whole sweep over
// the queries, so dK/dV need no cross-workgroup sum.  Scores are recomputed
// from Q, K and the forward's log-sum-exp.  With the KEY on the lanes
//   S  = Q K^T (init -LSE/scale)  and  dP = dO V^T (init -delta)
// come out with queries on the accumulator rows, which makes P and dS directly
// the A operands of  dV += P^T dO  and  dK += dS^T Q  (sums over the row index).
// Only dS crosses LDS once (a per-wave 32x32 transpose) for  dQ += dS K,  which
// leaves the chip as fp32 atomic adds shaped as two 128-byte row segments per
// wave-instruction.
// ===========================================================================
namespace {

constexpr int QP = DH + 4;   // padded LDS row

__global__ __launch_bounds__(256) void mha_clear_dq_kernel(float* __restrict__ dQ, long lddq, int width4, long rows) {
    // dQ is accumulated with atomic adds: its [rows, 4 * width4] window (row stride lddq) starts at zero
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * width4) return;
    const long r = i / width4;
    const int c4 = (int)(i % width4);
    *reinterpret_cast<float4*>(dQ + r * lddq + 4 * c4) = make_float4(0.f, 0.f, 0.f, 0.f);
}

// DROP: 0 = no attention dropout (the loop carries none of it), 1 = the mask is regenerated from the pair hash,
// 2 = the mask is read from the keep bits the forward pass wrote (scalar loads and one select per element instead of
// the hash).  The factor 1 / (1 - p) of the kept probabilities rides on the V fragment (dP) and on the final dV.
template <bool BF16, int DROP>
__global__ __launch_bounds__(256, 2) void mha_bwd_kernel(
    const float* __restrict__ Q, long ldq, const float* __restrict__ K, long ldk, const float* __restrict__ V, long ldv,
    const float* __restrict__ O, long ldo, const float* __restrict__ dO, long lddo, const float* __restrict__ LSE,
    const unsigned char* __restrict__ kpm, float* __restrict__ dQ, long lddq, float* __restrict__ dK, long lddk,
    float* __restrict__ dV, long lddv, int Lq, int Lk, int H, float scale, unsigned drop_thr, unsigned drop_seed,
    float drop_scale, const unsigned* __restrict__ keep_bits) {
    // LDS: 2 x (Q tile, dO tile) [32][QP], 2 x (lse, delta) [32], K tile [128][QP], 2 x 4 x dS/dQ patch [32][QP]
    __shared__ __attribute__((aligned(16))) float lds[4 * 32 * QP + 4 * 32 + 128 * QP + 2 * 4 * 32 * QP];
    auto Qs = [&](int buf) { return lds + buf * 32 * QP; };
    auto Gs = [&](int buf) { return lds + (2 + buf) * 32 * QP; };
    auto Ls = [&](int buf) { return lds + 4 * 32 * QP + buf * 64; };
    auto Ds = [&](int buf) { return lds + 4 * 32 * QP + buf * 64 + 32; };
    float* Kt = lds + 4 * 32 * QP + 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    float* const patches = Kt + 128 * QP;       // [2 sets][4 waves][32][QP]

    const int head = blockIdx.y, b = blockIdx.z;
    const int key_w0 = blockIdx.x * 128 + wave * 32;
    const int key = key_w0 + l31;
    const float c = scale * 1.4426950408889634f;   // folded into the K fragment: scores come out in the log2 domain
    const unsigned ks = attn_stream_key(drop_seed, (unsigned)(b * H + head)) ^ (unsigned)(key & ~1);
    const unsigned half_shift = (key & 1) ? 0u : 16u, thr_hi = drop_thr << 16;
    // keep bits: the words of this wave's key tile (see attn_bits_words), two per accumulator element and query tile;
    // a wave whose keys all lie past the end reads the last tile's (its probabilities are zero anyway)
    const int nkt = (Lk + 31) / 32;
    const unsigned long long* const Mw = DROP == 2
        ? reinterpret_cast<const unsigned long long*>(
              keep_bits + ((long)b * H + head) * attn_bits_words(Lq, Lk) +
              (long)min((int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(wave), nkt - 1) * 32)
        : nullptr;
    unsigned long long km[16];
    auto load_masks = [&](int t) {
#pragma unroll
        for (int e = 0; e < 16; ++e) km[e] = Mw[(long)t * nkt * 16 + e];
    };

    const float* Qb = Q + (long)b * Lq * ldq + head * DH;
    const float* Gb = dO + (long)b * Lq * lddo + head * DH;
    const float* Ob = O + (long)b * Lq * ldo + head * DH;
    const float* Lb = LSE + ((long)b * H + head) * Lq;

    // K tile of the workgroup into LDS (zero rows past Lk); V fragment straight to registers
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + 256 * i;
        const int kr = blockIdx.x * 128 + (f >> 3), d4 = (f & 7) << 2;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kr < Lk) v = *reinterpret_cast<const float4*>(K + ((long)b * Lk + kr) * ldk + head * DH + d4);
        *reinterpret_cast<float4*>(Kt + (f >> 3) * QP + d4) = v;
    }
    Frag<BF16> vf, kf;
    const bool kvalid = key < Lk;
    const bool kdead = !kvalid || (kpm && kpm[(long)b * Lk + key]);
    load_frag(vf, V + ((long)b * Lk + (kvalid ? key : 0)) * ldv + head * DH + 16 * h, kvalid ? (DROP ? drop_scale : 1.f) : 0.f);
    float4 sq, sg, so;
    float sl = 0.f;
    auto fetch = [&](int t) {
        const int qr = t * 32 + (tid >> 3), d4 = (tid & 7) << 2;
        if (qr < Lq) {
            sq = *reinterpret_cast<const float4*>(Qb + (long)qr * ldq + d4);
            sg = *reinterpret_cast<const float4*>(Gb + (long)qr * lddo + d4);
            so = *reinterpret_cast<const float4*>(Ob + (long)qr * ldo + d4);
        } else {
            sq = make_float4(0.f, 0.f, 0.f, 0.f);
            sg = sq;
            so = sq;
        }
        if (tid < 32) {
            const int qq = t * 32 + tid;
            sl = qq < Lq ? -(Lb[qq] * 1.4426950408889634f) : -INFINITY;      // the score accumulator starts at -LSE
        }
    };
    auto stash = [&](int buf) {
        *reinterpret_cast<float4*>(Qs(buf) + (tid >> 3) * QP + ((tid & 7) << 2)) = sq;
        *reinterpret_cast<float4*>(Gs(buf) + (tid >> 3) * QP + ((tid & 7) << 2)) = sg;
        if (tid < 32) Ls(buf)[tid] = sl;
        // delta[q] = sum_c O[q][c] dO[q][c] of this head: the eight lanes that hold a query row's 32 channels add up
        // (every workgroup of the (clip, head) computes the same value in the same order)
        float dsum = (so.x * sg.x + so.y * sg.y) + (so.z * sg.z + so.w * sg.w);
        dsum += __shfl_xor(dsum, 1);
        dsum += __shfl_xor(dsum, 2);
        dsum += __shfl_xor(dsum, 4);
        if ((tid & 7) == 0) Ds(buf)[tid >> 3] = DROP ? dsum : -dsum;          // without dropout dP starts at -delta
    };
    fetch(0);
    stash(0);
    __syncthreads();
    // K fragment (B operand of S): lane holds c * K[key][16h .. 16h+15]
    load_frag(kf, Kt + (wave * 32 + l31) * QP + 16 * h, c);
    // From here on the LDS copy of K only feeds dQ = dS K * scale: fold the scale into it once.
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + 256 * i;
        float4* kp = reinterpret_cast<float4*>(Kt + (f >> 3) * QP + ((f & 7) << 2));
        float4 v = *kp;
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        *kp = v;
    }
    __syncthreads();
    const bool any_dead = __builtin_amdgcn_ballot_w64(kdead) != 0;
    // B operand of dQ = dS K: rows 16h .. 16h+15 of this wave's (scaled) keys, fixed for the whole sweep
    Frag<BF16> kq;
#pragma unroll
    for (int j = 0; j < 16; ++j) kq.set(j, Kt[(wave * 32 + 16 * h + j) * QP + l31]);
    f32x16 dk, dv;
#pragma unroll
    for (int e = 0; e < 16; ++e) { dk[e] = 0.f; dv[e] = 0.f; }

    const int ntile = (Lq + 31) / 32;
    for (int t = 0; t < ntile; ++t) {
        const int cur = t & 1;
        const bool more = t + 1 < ntile;
        if (more) fetch(t + 1);
        const float* qs = Qs(cur);
        const float* gs = Gs(cur);
        // ---- S' = Q K^T - LSE/scale ; dP' = dO V^T - delta      (rows q, cols key)
        // (with dropout the mask multiplies dO V^T before delta is subtracted, so delta is applied later)
        f32x16 s, dp;
        float dl[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 l4 = *reinterpret_cast<const float4*>(Ls(cur) + 8 * g4 + 4 * h);
            const float4 d4 = *reinterpret_cast<const float4*>(Ds(cur) + 8 * g4 + 4 * h);
            s[4 * g4] = l4.x; s[4 * g4 + 1] = l4.y; s[4 * g4 + 2] = l4.z; s[4 * g4 + 3] = l4.w;
            dl[4 * g4] = d4.x; dl[4 * g4 + 1] = d4.y; dl[4 * g4 + 2] = d4.z; dl[4 * g4 + 3] = d4.w;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) dp[e] = DROP ? 0.f : dl[e];
        {
            Frag<BF16> qf, gf;
            load_frag(qf, qs + l31 * QP + 16 * h);
            load_frag(gf, gs + l31 * QP + 16 * h);
            if (DROP == 2) {
                // The scalar loads of this tile's masks go out once the LDS reads above have landed and come back
                // under the 32 matrix instructions below: scalar loads and LDS reads share a counter that cannot tell
                // them apart, so a scalar load in flight would turn every wait for an LDS read into a wait for it.
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_sched_barrier(0);
                load_masks(t);
            }
            mma(qf, kf, s);
            mma(gf, vf, dp);
        }
        // ---- P = exp2(S'), dS = P * dP' (the 1/sqrt(d) factor rides on K and on the final dK); masked or
        //      out-of-range keys contribute nothing
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = __builtin_amdgcn_exp2f(s[e]);
        if (any_dead) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = kdead ? 0.f : s[e];
        }
        if (DROP) {
            f32x16 pm;                                  // the kept probabilities (their factor 1 / (1 - p) is in dP and dV)
            if (DROP == 2) {
#pragma unroll
                for (int e = 0; e < 16; ++e) pm[e] = __builtin_amdgcn_inverse_ballot_w64(km[e]) ? s[e] : 0.f;
            } else {
                const unsigned x0 = ks ^ ((unsigned)(t * 32 + 4 * h) << 16);     // query index sits in the high half
                // the pair hash of (query, key & ~1): an odd key reads its high half; an even key's low half is moved up,
                // and the comparison runs on the upper 16 bits (the threshold's lower 16 are zero)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    pm[e] = (attn_mix(x0 ^ ((unsigned)acc_row(e, 0) << 16)) << half_shift) >= thr_hi ? s[e] : 0.f;
            }
            // dropped probabilities feed dV; dS = P (M dP - delta) = (P M) dP - P delta, in packed operations
            f32x16 dlv;
#pragma unroll
            for (int e = 0; e < 16; ++e) dlv[e] = dl[e];
            const f32x16 pd = s * dlv;
            dp = pm * dp - pd;
            s = pm;
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) dp[e] *= s[e];
        }
        // ---- dV += P^T dO ; dK += dS^T Q     (A = the accumulator's 16 query rows, B = those rows of the LDS tile)
        {
            Frag<BF16> pf, gf;
            acc_frag(pf, s);
            load_frag_rows(gf, gs, QP, h, l31);
            mma(pf, gf, dv);
            Frag<BF16> df, qf;
            acc_frag(df, dp);
            load_frag_rows(qf, qs, QP, h, l31);
            mma(df, qf, dk);
        }
        // ---- dQ tile = dS K : transpose dS through this wave's LDS patch (patch sets alternate per tile, so a
        //      wave that runs ahead never overwrites a patch the others still reduce from)
        float* const pset = patches + (t & 1) * 4 * 32 * QP;
        float* const dSs = pset + wave * 32 * QP;
#pragma unroll
        for (int e = 0; e < 16; ++e) dSs[acc_row(e, h) * QP + l31] = dp[e];
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes have landed
        __builtin_amdgcn_wave_barrier();
        f32x16 dq;
#pragma unroll
        for (int e = 0; e < 16; ++e) dq[e] = 0.f;
        {
            Frag<BF16> af;
            load_frag(af, dSs + l31 * QP + 16 * h);
            mma(af, kq, dq);
        }
        // The four waves hold partial dQ tiles over their 32 keys each.  Float atomics are served at the memory
        // side at a fixed rate, so the partials are summed through the (now free) patches first: one atomic per
        // element per workgroup instead of four.
#pragma unroll
        for (int e = 0; e < 16; ++e) dSs[acc_row(e, h) * QP + l31] = dq[e];
        if (more) stash(cur ^ 1);
        __syncthreads();
        {
            const int col = tid & 31;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (tid >> 5) + 8 * i;
                const float* pp = pset + row * QP + col;
                const float v = (pp[0] + pp[32 * QP]) + (pp[2 * 32 * QP] + pp[3 * 32 * QP]);
                const int qr = t * 32 + row;
                if (qr < Lq) atomicAdd(dQ + ((long)b * Lq + qr) * lddq + head * DH + col, v);
            }
        }
    }
    // dK / dV: rows = this wave's keys, cols = channel on the lanes
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int kr = key_w0 + acc_row(e, h);
        if (kr < Lk) {
            dK[((long)b * Lk + kr) * lddk + head * DH + l31] = dk[e] * scale;
            dV[((long)b * Lk + kr) * lddv + head * DH + l31] = DROP ? dv[e] * drop_scale : dv[e];
        }
    }
}

}  // namespace

static int mha_bwd_impl(bool bf16, const float* q, long ldq, const float* k, long ldk, const float* v, long ldv,
                        const float* out, long ldo, const float* d_out, long lddo, const float* lse,
                        const unsigned char* key_padding_mask, float* delta_ws, float* dq, long lddq, float* dk,
                        long lddk, float* dv, long lddv, int B, int Lq, int Lk, int nhead, int head_dim, float scale,
                        float drop_p, unsigned drop_seed, const unsigned* drop_bits, hipStream_t stream) {
    (void)delta_ws;      // kept in the signature; the kernel computes delta itself
    SPOT_CHECK_ARG(q && k && v && out && d_out && lse && dq && dk && dv, "mha_bwd: null pointer");
    SPOT_CHECK_ARG(head_dim == DH, "mha_bwd: head_dim=%d unsupported (this build handles 32)", head_dim);
    SPOT_CHECK_ARG(B >= 0 && Lq > 0 && Lk > 0 && nhead > 0, "mha_bwd: bad shape B=%d Lq=%d Lk=%d heads=%d", B, Lq, Lk, nhead);
    SPOT_CHECK_ARG(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0 && lddo % 4 == 0 &&
                       (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out | (uintptr_t)d_out) & 15) == 0,
                   "mha_bwd: operands must be 16-byte aligned with row strides that are multiples of 4");
    SPOT_CHECK_ARG(B <= 65535 && nhead <= 65535, "mha_bwd: B and heads must fit a grid dimension");
    SPOT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "mha_bwd: drop_p=%f outside [0,1)", drop_p);
    SPOT_CHECK_ARG(drop_p == 0.f || (Lq <= 65536 && Lk <= 65536), "mha_bwd: dropout needs Lq, Lk <= 65536");
    SPOT_CHECK_ARG(((uintptr_t)drop_bits & 7) == 0, "mha_bwd: drop_bits must be 8-byte aligned");
    if (B == 0) return SPOT_OK;
    // dQ is accumulated atomically: its [B*Lq, nhead*32] window is cleared first
    SPOT_CHECK_ARG(lddq % 4 == 0 && ((uintptr_t)dq & 15) == 0, "mha_bwd: dq must be 16-byte aligned with a row stride that is a multiple of 4");
    const long rows = (long)B * Lq, cells = rows * (nhead * DH / 4);
    mha_clear_dq_kernel<<<(int)((cells + 255) / 256), 256, 0, stream>>>(dq, lddq, nhead * DH / 4, rows);
    SPOT_CHECK_LAUNCH("mha_bwd(clear)");
    dim3 grid((Lk + 127) / 128, nhead, B);
    const unsigned thr = attn_threshold(drop_p);
#define SPOT_MHA_BWD(BF, DR)                                                                                                   \
    mha_bwd_kernel<BF, DR><<<grid, 256, 0, stream>>>(q, ldq, k, ldk, v, ldv, out, ldo, d_out, lddo, lse, key_padding_mask, dq, \
                                                     lddq, dk, lddk, dv, lddv, Lq, Lk, nhead, scale, thr, drop_seed,           \
                                                     1.f / (1.f - drop_p), drop_bits)
    if (bf16) {
        if (thr && drop_bits) SPOT_MHA_BWD(true, 2);
        else if (thr) SPOT_MHA_BWD(true, 1);
        else SPOT_MHA_BWD(true, 0);
    } else {
        if (thr && drop_bits) SPOT_MHA_BWD(false, 2);
        else if (thr) SPOT_MHA_BWD(false, 1);
        else SPOT_MHA_BWD(false, 0);
    }
#undef SPOT_MHA_BWD
    SPOT_CHECK_LAUNCH("mha_bwd");
    return SPOT_OK;
}

extern "C" int spot_mha_bwd_f32(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv,
                                const float* out, long ldo, const float* d_out, long lddo, const float* lse,
                                const unsigned char* key_padding_mask, float* delta_ws, float* dq, long lddq,
                                float* dk, long lddk, float* dv, long lddv, int B, int Lq, int Lk, int nhead,
                                int head_dim, float scale, float drop_p, unsigned drop_seed, const unsigned* drop_bits,
                                hipStream_t stream) {
    return mha_bwd_impl(false, q, ldq, k, ldk, v, ldv, out, ldo, d_out, lddo, lse, key_padding_mask, delta_ws, dq, lddq,
                        dk, lddk, dv, lddv, B, Lq, Lk, nhead, head_dim, scale, drop_p, drop_seed, drop_bits, stream);
}

extern "C" int spot_mha_bwd_bf16(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv,
                                 const float* out, long ldo, const float* d_out, long lddo, const float* lse,
                                 const unsigned char* key_padding_mask, float* delta_ws, float* dq, long lddq,
                                 float* dk, long lddk, float* dv, long lddv, int B, int Lq, int Lk, int nhead,
                                 int head_dim, float scale, float drop_p, unsigned drop_seed, const unsigned* drop_bits,
                                 hipStream_t stream) {
    return mha_bwd_impl(true, q, ldq, k, ldk, v, ldv, out, ldo, d_out, lddo, lse, key_padding_mask, delta_ws, dq, lddq,
                        dk, lddk, dv, lddv, B, Lq, Lk, nhead, head_dim, scale, drop_p, drop_seed, drop_bits, stream);
}


// The attention dropout mask as bytes [B, nhead, Lq, Lk] (1 = kept): lets a host reference apply the very
// mask the fused kernels regenerate from (seed, clip, head, query, key).
namespace {
__global__ __launch_bounds__(256) void attn_mask_kernel(unsigned char* __restrict__ out, int Lq, int Lk, int BH,
                                                        unsigned thr, unsigned seed) {
    const long total = (long)BH * Lq * Lk;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const unsigned key = (unsigned)(i % Lk);
        const unsigned q = (unsigned)((i / Lk) % Lq);
        const unsigned bh = (unsigned)(i / ((long)Lk * Lq));
        out[i] = attn_keep(attn_stream_key(seed, bh), q, key, thr) ? 1 : 0;
    }
}
}  // namespace

extern "C" int spot_mha_dropout_mask(unsigned char* mask, int B, int nhead, int Lq, int Lk, float drop_p,
                                     unsigned drop_seed, hipStream_t stream) {
    SPOT_CHECK_ARG(mask && B > 0 && nhead > 0 && Lq > 0 && Lk > 0 && Lq <= 65536 && Lk <= 65536, "mha_dropout_mask: bad arguments");
    SPOT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "mha_dropout_mask: drop_p=%f outside [0,1)", drop_p);
    attn_mask_kernel<<<2048, 256, 0, stream>>>(mask, Lq, Lk, B * nhead, attn_threshold(drop_p), drop_seed);
    SPOT_CHECK_LAUNCH("mha_dropout_mask");
    return SPOT_OK;
}


// ---------------------------------------------------------------------------
// Forward for grids too small to fill the chip (the online tracker: one clip, 8 heads, 2-8 query tiles): the keys
// are cut in nsplit ranges, one workgroup each, and a second kernel merges the partial softmax sums
//   m = max_s m_s,  out = sum_s 2^(m_s - m) o_s / sum_s 2^(m_s - m) l_s.
// Inference only: no dropout, no log-sum-exp output.
// ---------------------------------------------------------------------------
namespace {
__device__ __forceinline__ void mha_merge_body(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                               float* __restrict__ O, long ldo, int B, int Lq, int H, int nsplit, int bid) {
    const long idx = (long)bid * 256 + threadIdx.x;                 // (b, q, head, group of 4 features)
    const long total = (long)B * Lq * H * (DH / 4);
    if (idx >= total) return;
    const int g4 = (int)(idx % (DH / 4)), head = (int)(idx / (DH / 4) % H);
    const long row = idx / ((DH / 4) * H);                            // b * Lq + q
    const int b = (int)(row / Lq), q = (int)(row - (long)b * Lq);
    float m = -INFINITY;
    for (int s = 0; s < nsplit; ++s) m = fmaxf(m, part_ml[((((long)s * B + b) * H + head) * Lq + q) * 2]);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float den = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float* ml = part_ml + ((((long)s * B + b) * H + head) * Lq + q) * 2;
        const float w = ml[0] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(ml[0] - m);
        const float4 o = *reinterpret_cast<const float4*>(part_o + (((long)s * B + b) * Lq + q) * ((long)H * DH) + head * DH + 4 * g4);
        acc.x += w * o.x; acc.y += w * o.y; acc.z += w * o.z; acc.w += w * o.w;
        den += w * ml[1];
    }
    const float inv = 1.f / den;
    *reinterpret_cast<float4*>(O + row * ldo + head * DH + 4 * g4) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}
__global__ __launch_bounds__(256) void mha_merge_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                        float* __restrict__ O, long ldo, int B, int Lq, int H, int nsplit) {
    mha_merge_body(part_o, part_ml, O, ldo, B, Lq, H, nsplit, blockIdx.x);
}
__global__ __launch_bounds__(256) void mha_merge_pair_kernel(const float* __restrict__ po0, const float* __restrict__ pm0,
                                                             float* __restrict__ O0, long ldo0, int Lq0, int ns0,
                                                             const float* __restrict__ po1, const float* __restrict__ pm1,
                                                             float* __restrict__ O1, long ldo1, int Lq1, int ns1, int H, int n0) {
    if ((int)blockIdx.x < n0) mha_merge_body(po0, pm0, O0, ldo0, 1, Lq0, H, ns0, blockIdx.x);
    else mha_merge_body(po1, pm1, O1, ldo1, 1, Lq1, H, ns1, (int)blockIdx.x - n0);
}
}  // namespace

extern "C" long spot_mha_fwd_splitk_ws_floats(int B, int Lq, int nhead, int nsplit) {
    if (B <= 0 || Lq <= 0 || nhead <= 0 || nsplit <= 0) return 0;
    return (long)nsplit * B * Lq * nhead * (DH + 2);
}

extern "C" int spot_mha_fwd_splitk_f32(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv,
                                       float* out, long ldo, const unsigned char* key_padding_mask, float* ws, int nsplit,
                                       int B, int Lq, int Lk, int nhead, int head_dim, float scale, hipStream_t stream) {
    SPOT_CHECK_ARG(q && k && v && out && ws, "mha_fwd_splitk: null pointer");
    SPOT_CHECK_ARG(head_dim == DH, "mha_fwd_splitk: head_dim=%d unsupported (this build handles 32)", head_dim);
    SPOT_CHECK_ARG(B >= 0 && Lq > 0 && Lk > 0 && nhead > 0 && nsplit >= 1 && nsplit <= 64,
                   "mha_fwd_splitk: bad shape B=%d Lq=%d Lk=%d heads=%d nsplit=%d", B, Lq, Lk, nhead, nsplit);
    SPOT_CHECK_ARG(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0 &&
                       (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out | (uintptr_t)ws) & 15) == 0,
                   "mha_fwd_splitk: operands must be 16-byte aligned with row strides that are multiples of 4");
    SPOT_CHECK_ARG(B <= 65535 && nhead <= 65535, "mha_fwd_splitk: B and heads must fit a grid dimension");
    if (B == 0) return SPOT_OK;
    float* part_o = ws;
    float* part_ml = ws + (long)nsplit * B * Lq * nhead * DH;
    dim3 grid(((Lq + 127) / 128) * nsplit, nhead, B);
    mha_fwd_kernel<false><<<grid, 256, 0, stream>>>(MhaFwdArgs{q, ldq, k, ldk, v, ldv, out, ldo, nullptr, key_padding_mask, Lq, Lk,
                                                    nhead, scale, 0u, 0u, 1.f, nsplit, part_o, part_ml});
    SPOT_CHECK_LAUNCH("mha_fwd_splitk");
    const long total = (long)B * Lq * nhead * (DH / 4);
    mha_merge_kernel<<<spot_ceil_div(total, 256), 256, 0, stream>>>(part_o, part_ml, out, ldo, B, Lq, nhead, nsplit);
    SPOT_CHECK_LAUNCH("mha_fwd_splitk(merge)");
    return SPOT_OK;
}

// Two split-key forwards of one clip each in two launches instead of four (forward pair, merge pair): the template and
// the search stream of a fusion layer.  Same arithmetic as two spot_mha_fwd_splitk_f32 calls.
extern "C" int spot_mha_fwd_splitk_pair_f32(const float* q0, long ldq0, const float* k0, long ldk0, const float* v0, long ldv0,
                                            float* out0, long ldo0, const unsigned char* kpm0, float* ws0, int nsplit0, int Lq0,
                                            int Lk0, const float* q1, long ldq1, const float* k1, long ldk1, const float* v1,
                                            long ldv1, float* out1, long ldo1, const unsigned char* kpm1, float* ws1, int nsplit1,
                                            int Lq1, int Lk1, int nhead, int head_dim, float scale, hipStream_t stream) {
    SPOT_CHECK_ARG(q0 && k0 && v0 && out0 && ws0 && q1 && k1 && v1 && out1 && ws1, "mha_fwd_splitk_pair: null pointer");
    SPOT_CHECK_ARG(head_dim == DH, "mha_fwd_splitk_pair: head_dim=%d unsupported (this build handles 32)", head_dim);
    SPOT_CHECK_ARG(Lq0 > 0 && Lk0 > 0 && Lq1 > 0 && Lk1 > 0 && nhead > 0 && nhead <= 65535 && nsplit0 >= 1 && nsplit0 <= 64 &&
                       nsplit1 >= 1 && nsplit1 <= 64, "mha_fwd_splitk_pair: bad shape");
    SPOT_CHECK_ARG(((ldq0 | ldk0 | ldv0 | ldo0 | ldq1 | ldk1 | ldv1 | ldo1) % 4) == 0 &&
                       (((uintptr_t)q0 | (uintptr_t)k0 | (uintptr_t)v0 | (uintptr_t)out0 | (uintptr_t)ws0 | (uintptr_t)q1 |
                         (uintptr_t)k1 | (uintptr_t)v1 | (uintptr_t)out1 | (uintptr_t)ws1) & 15) == 0,
                   "mha_fwd_splitk_pair: operands must be 16-byte aligned with row strides that are multiples of 4");
    float* po0 = ws0; float* pm0 = ws0 + (long)nsplit0 * Lq0 * nhead * DH;
    float* po1 = ws1; float* pm1 = ws1 + (long)nsplit1 * Lq1 * nhead * DH;
    const int n0 = ((Lq0 + 127) / 128) * nsplit0, n1 = ((Lq1 + 127) / 128) * nsplit1;
    mha_fwd_pair_kernel<<<dim3(n0 + n1, nhead, 1), 256, 0, stream>>>(
        MhaFwdArgs{q0, ldq0, k0, ldk0, v0, ldv0, out0, ldo0, nullptr, kpm0, Lq0, Lk0, nhead, scale, 0u, 0u, 1.f, nsplit0, po0, pm0},
        MhaFwdArgs{q1, ldq1, k1, ldk1, v1, ldv1, out1, ldo1, nullptr, kpm1, Lq1, Lk1, nhead, scale, 0u, 0u, 1.f, nsplit1, po1, pm1}, n0);
    SPOT_CHECK_LAUNCH("mha_fwd_splitk_pair");
    const int m0 = (int)spot_ceil_div((long)Lq0 * nhead * (DH / 4), 256), m1 = (int)spot_ceil_div((long)Lq1 * nhead * (DH / 4), 256);
    mha_merge_pair_kernel<<<m0 + m1, 256, 0, stream>>>(po0, pm0, out0, ldo0, Lq0, nsplit0, po1, pm1, out1, ldo1, Lq1, nsplit1,
                                                      nhead, m0);
    SPOT_CHECK_LAUNCH("mha_fwd_splitk_pair(merge)");
    return SPOT_OK;
}

// ---------------------------------------------------------------------------
// The split-key forward for callers that may not allocate (a frame replayed under HIP graph capture, serving loops
// that forbid hipMalloc): the partial sums live in a file-scope device array instead of a caller workspace.
// Every GPU the library runs on holds its own copy of the array until the process ends; its address on a device is
// resolved once, with that device current, on the entry point's first call there.  Every (range, clip, head, query)
// slot the merge kernel reads is written by the forward kernel of the same call (an empty key range stores max = -inf,
// sum = 0), so nothing has to be cleared between calls; calls that overlap on one GPU share the array and race.
// ---------------------------------------------------------------------------
static_assert(SPOT_MHA_SPLITK_NOALLOC_MAX_FLOATS == 8L * 8 * 1024 * (DH + 2), "8 key ranges x 8 heads x 1024 queries");
#define SPOT_SPLITK_SCRATCH_FLOATS SPOT_MHA_SPLITK_NOALLOC_MAX_FLOATS
static __device__ __align__(256) float scratch[SPOT_SPLITK_SCRATCH_FLOATS];

namespace {
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
float* g_scratch_ptr[MAX_DEVICES] = {};

// Device address of `scratch` on the current device (nullptr + error text on failure).
float* scratch_on_current_device() {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= MAX_DEVICES) {
        spot_set_error("mha_fwd_splitk_noalloc: hipGetDevice failed or device %d beyond %d: %s", dev, MAX_DEVICES,
                       hipGetErrorString(e));
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        float* p = nullptr;
        e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            spot_set_error("mha_fwd_splitk_noalloc: hipGetSymbolAddress(scratch) failed on device %d: %s", dev,
                           hipGetErrorString(e));
            return nullptr;
        }
        g_scratch_ptr[dev] = p;
    }
    return g_scratch_ptr[dev];
}
}  // namespace

extern "C" int spot_mha_fwd_splitk_noalloc_f32(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv,
                                               float* out, long ldo, const unsigned char* key_padding_mask, int nsplit,
                                               int B, int Lq, int Lk, int nhead, int head_dim, float scale,
                                               hipStream_t stream) {
    const long need = spot_mha_fwd_splitk_ws_floats(B, Lq, nhead, nsplit);
    SPOT_CHECK_ARG(need <= SPOT_SPLITK_SCRATCH_FLOATS,
                   "mha_fwd_splitk_noalloc: nsplit*B*Lq*heads*(head_dim+2) = %ld floats exceeds the static scratch of %ld "
                   "floats (nsplit=%d B=%d Lq=%d heads=%d); use spot_mha_fwd_splitk_f32 with a caller workspace",
                   need, (long)SPOT_SPLITK_SCRATCH_FLOATS, nsplit, B, Lq, nhead);
    float* ws = scratch_on_current_device();
    if (!ws) return SPOT_ERR_LAUNCH;
    return spot_mha_fwd_splitk_f32(q, ldq, k, ldk, v, ldv, out, ldo, key_padding_mask, ws, nsplit, B, Lq, Lk, nhead,
                                   head_dim, scale, stream);
}
