// Multi-head softmax attention (forward + backward) for the Point-MAE token sequences
// (T <= 128 tokens: 64 patch tokens in pre-training, cls + 64 in fine-tuning; head_dim 64, 6 heads) on gfx950
// matrix cores.
//
// Beneath: timm-0.4.5 Attention.forward -- in-tree twin
//   Point-MAE_SA3D/models/Point_MAE.py:113-125 (qkv reshape :115, softmax(q k^T * scale) :118-119,
//   attn @ v :122), used by every Block of TransformerEncoder/TransformerDecoder
//   (models_mae_learn_loss.py:901-917,959-990).
//
// Design (MI355X): a whole (batch, head) problem -- Q,K,V of (<=128)x64 -- fits one CU, so one
// workgroup owns one (b,h) and nothing but qkv in / out (+lse) ever crosses HBM: the
// T x T score matrix lives in MFMA accumulators.  Scores are computed TRANSPOSED
// (S^T = K Q^T) so that a query is a lane and its keys are registers: the softmax row
// reduction is register-local plus one lane^32 exchange, and the probability tile is
// already laid out as the A operand of the P.V product (accumulator-as-operand, k order
// permuted by the C layout).  Two precisions share the structure:
//   GM3D_BF16: v_mfma_f32_32x32x16_bf16, f32 softmax and accumulation (throughput mode);
//   GM3D_F32 : v_mfma_f32_32x32x2_f32, bit-exact f32 FMA chains (parity mode).
// Backward recomputes P from the saved log-sum-exp; NT waves each own one 32-key tile (dK/dV) and NT more waves each own
// one 32-query tile (dQ), side by side in the same workgroup: no atomics, no cross-workgroup sums.
// Kernels are templated on MT, the number of 32-row tiles the LDS buffers hold (2: T <= 64, 4: T <= 128).
#include "common.hpp"

namespace gm3d {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HD = 64;        // head dim
constexpr int VLD = 72;       // bf16 LDS row pitch (144 B: 16-B aligned rows)

// C/D layout of a 32x32 MFMA: column = lane & 31, row = crow(reg, lane >> 5).
__device__ __forceinline__ int crow(int g, int hh) { return (g & 3) + 8 * (g >> 2) + 4 * hh; }
// k index (0..31) carried by element j of lane half hh at k-step s when accumulator
// registers 8s..8s+7 are reused as a bf16 operand: equals crow(8s + j, hh).
__device__ __forceinline__ int permk(int s, int j, int hh) { return 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3); }

__device__ __forceinline__ bf16x8 ld8(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (bf16_t)0.0f;
    return z;
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int g = 0; g < 16; ++g) z[g] = 0.f;
    return z;
}
// 8 elements M[row][k], k = permk order of k-step s inside the 32-wide tile starting at k0
__device__ __forceinline__ bf16x8 ld8_perm(const bf16_t* row, int k0, int s, int hh) {
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(row + k0 + 16 * s + 4 * hh);
    const bf16x4 b = *reinterpret_cast<const bf16x4*>(row + k0 + 16 * s + 8 + 4 * hh);
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
    return v;
}
// 8 elements M[k][col] gathered down a column, k in permk order
__device__ __forceinline__ bf16x8 ld8_col_perm(const bf16_t* M, int ld, int k0, int s, int hh, int col) {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = M[(size_t)(k0 + permk(s, j, hh)) * ld + col];
    return v;
}
__device__ __forceinline__ bf16x8 cvt8(const f32x16& x, int s, float mul) {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16_t)(x[8 * s + j] * mul);
    return v;
}

#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define MFMA_F32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

// Cooperative copy of one head's (T,64) slice of qkv into a zero-padded `rows`-row LDS tile.
__device__ __forceinline__ void stage_bf16(bf16_t* dst /*[rows][VLD]*/, const bf16_t* src, size_t row_stride, int T, int rows) {
    for (int c = threadIdx.x; c < rows * 8; c += blockDim.x) {
        const int row = c >> 3, col = (c & 7) * 8;
        *reinterpret_cast<bf16x8*>(dst + row * VLD + col) = row < T ? ld8(src + (size_t)row * row_stride + col) : zero8();
    }
}
// f32 tiles use a 64-float pitch with a per-row rotation so that both row-wise and
// column-wise lane patterns are bank-conflict free: element (r,c) at r*64 + ((c + r) & 63).
__device__ __forceinline__ int rot(int r, int c) { return r * 64 + ((c + r) & 63); }
__device__ __forceinline__ void stage_f32(float* dst /*[rows*64]*/, const float* src, size_t row_stride, int T, int rows) {
    for (int c = threadIdx.x; c < rows * 64; c += blockDim.x) {
        const int row = c >> 6, col = c & 63;
        dst[rot(row, col)] = row < T ? src[(size_t)row * row_stride + col] : 0.f;
    }
}

// ===================================================================== forward, bf16
template <int MT>
__global__ __launch_bounds__(64 * MT) void attn_fwd_bf16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                float* __restrict__ lse, int T, int H, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t Vs[32 * MT * VLD];
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int lane = threadIdx.x & 63, qb = threadIdx.x >> 6;  // wave = 32-query block
    const int r = lane & 31, hh = lane >> 5;
    const size_t rs = (size_t)3 * H * HD;
    const bf16_t* Qg = qkv + (size_t)b * T * rs + (size_t)h * HD;
    const bf16_t* Kg = Qg + (size_t)H * HD;
    const bf16_t* Vg = Kg + (size_t)H * HD;
    const int NK = (T + 31) >> 5;

    stage_bf16(Vs, Vg, rs, T, 32 * MT);
    __syncthreads();

    // S^T tiles: rows = keys (regs), cols = queries (lanes)
    f32x16 st[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) st[t] = zero16();
    const int qrow = 32 * qb + r;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const bf16x8 bq = qrow < T ? ld8(Qg + (size_t)qrow * rs + 16 * s + 8 * hh) : zero8();
#pragma unroll
        for (int t = 0; t < MT; ++t)
            if (t < NK) {
                const bf16x8 a = 32 * t + r < T ? ld8(Kg + (size_t)(32 * t + r) * rs + 16 * s + 8 * hh) : zero8();
                st[t] = MFMA_BF16(a, bq, st[t]);
            }
    }
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int key = 32 * t + crow(g, hh);
            st[t][g] = key < T ? st[t][g] * scale : -INFINITY;
            m = fmaxf(m, st[t][g]);
        }
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            st[t][g] = __expf(st[t][g] - m);
            sum += st[t][g];
        }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;

    f32x16 o0 = zero16(), o1 = zero16();
#pragma unroll
    for (int t = 0; t < MT; ++t)
        if (t < NK) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 p = cvt8(st[t], s, inv);
                o0 = MFMA_BF16(p, ld8_col_perm(Vs, VLD, 32 * t, s, hh, r), o0);
                o1 = MFMA_BF16(p, ld8_col_perm(Vs, VLD, 32 * t, s, hh, 32 + r), o1);
            }
        }
    // O tile: rows = queries (regs), cols = d (lanes)
    bf16_t* og = out + (size_t)b * T * H * HD + (size_t)h * HD;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const int q = 32 * qb + crow(g, hh);
        if (q < T) {
            og[(size_t)q * H * HD + r] = (bf16_t)o0[g];
            og[(size_t)q * H * HD + 32 + r] = (bf16_t)o1[g];
        }
    }
    if (lse && hh == 0 && qrow < T) lse[((size_t)b * H + h) * T + qrow] = m + logf(sum);
}

// ===================================================================== forward, f32
template <int MT>
__global__ __launch_bounds__(64 * MT) void attn_fwd_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                               float* __restrict__ lse, int T, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int ROWS = 32 * MT;
    float* Qs = sm;
    float* Ks = Qs + ROWS * 64;
    float* Vs = Ks + ROWS * 64;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int lane = threadIdx.x & 63, qb = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const size_t rs = (size_t)3 * H * HD;
    const float* Qg = qkv + (size_t)b * T * rs + (size_t)h * HD;
    const int NK = (T + 31) >> 5;
    stage_f32(Qs, Qg, rs, T, ROWS);
    stage_f32(Ks, Qg + (size_t)H * HD, rs, T, ROWS);
    stage_f32(Vs, Qg + (size_t)2 * H * HD, rs, T, ROWS);
    __syncthreads();

    f32x16 st[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) st[t] = zero16();
    const int qrow = 32 * qb + r;
    for (int s = 0; s < 32; ++s) {
        const int d = 2 * s + hh;
        const float bq = Qs[rot(qrow, d)];
#pragma unroll
        for (int t = 0; t < MT; ++t)
            if (t < NK) st[t] = MFMA_F32(Ks[rot(32 * t + r, d)], bq, st[t]);
    }
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int key = 32 * t + crow(g, hh);
            st[t][g] = key < T ? st[t][g] * scale : -INFINITY;
            m = fmaxf(m, st[t][g]);
        }
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            st[t][g] = expf(st[t][g] - m);
            sum += st[t][g];
        }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;

    f32x16 o0 = zero16(), o1 = zero16();
#pragma unroll
    for (int s = 0; s < 16; ++s) {  // k-step s consumes accumulator register s (key crow(s,hh))
        const int key = crow(s, hh);
#pragma unroll
        for (int t = 0; t < MT; ++t)
            if (t < NK) {
                const float p = st[t][s] * inv;
                o0 = MFMA_F32(p, Vs[rot(32 * t + key, r)], o0);
                o1 = MFMA_F32(p, Vs[rot(32 * t + key, 32 + r)], o1);
            }
    }
    float* og = out + (size_t)b * T * H * HD + (size_t)h * HD;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const int q = 32 * qb + crow(g, hh);
        if (q < T) {
            og[(size_t)q * H * HD + r] = o0[g];
            og[(size_t)q * H * HD + 32 + r] = o1[g];
        }
    }
    if (lse && hh == 0 && qrow < T) lse[((size_t)b * H + h) * T + qrow] = m + logf(sum);
}

// ===================================================================== backward, bf16
// LDS (dynamic): Q,K,V,dO tiles (ROWS x VLD bf16 each) + lse[ROWS] + delta[ROWS].
template <int MT>
__global__ __launch_bounds__(128 * MT) void attn_bwd_bf16_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                                const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                                bf16_t* __restrict__ dqkv, int T, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int ROWS = 32 * MT;
    bf16_t* Qs = reinterpret_cast<bf16_t*>(sm);
    bf16_t* Ks = Qs + ROWS * VLD;
    bf16_t* Vs = Ks + ROWS * VLD;
    bf16_t* Ds = Vs + ROWS * VLD;
    float* Ls = reinterpret_cast<float*>(Ds + ROWS * VLD);
    float* Del = Ls + ROWS;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const size_t rs = (size_t)3 * H * HD, os = (size_t)H * HD;
    const bf16_t* Qg = qkv + (size_t)b * T * rs + (size_t)h * HD;
    const bf16_t* Og = out + (size_t)b * T * os + (size_t)h * HD;
    const bf16_t* Dg = dout + (size_t)b * T * os + (size_t)h * HD;
    const int NT = (T + 31) >> 5;  // number of 32-row tiles (= waves)

    stage_bf16(Qs, Qg, rs, T, ROWS);
    stage_bf16(Ks, Qg + os, rs, T, ROWS);
    stage_bf16(Vs, Qg + 2 * os, rs, T, ROWS);
    stage_bf16(Ds, Dg, os, T, ROWS);
    {   // delta[q] = sum_d dO[q][d] * O[q][d]; two threads per row
        const int q = tid >> 1, half = tid & 1;
        float acc = 0.f;
        if (q < T) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bf16x8 o8 = ld8(Og + (size_t)q * os + half * 32 + c * 8);
                const bf16x8 d8 = ld8(Dg + (size_t)q * os + half * 32 + c * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += (float)o8[j] * (float)d8[j];
            }
        }
        acc += __shfl_xor(acc, 1);
        if (half == 0 && q < ROWS) {
            Del[q] = acc;
            Ls[q] = q < T ? lse[((size_t)b * H + h) * T + q] : 0.f;
        }
    }
    __syncthreads();

    bf16_t* dQg = dqkv + (size_t)b * T * rs + (size_t)h * HD;
    bf16_t* dKg = dQg + os;
    bf16_t* dVg = dKg + os;

    // The workgroup has 2*NT waves: waves 0..NT-1 each own one key tile (phase A: dK, dV), waves NT..2NT-1 each own one query
    // tile (phase B: dQ) -- the two phases of a (b,h) problem run side by side instead of one after the other.
    // ---- phase A: this wave owns key tile kt.  Tiles X[q][key]: rows = q (regs), cols = key (lanes).
    if (w < NT) {
        const int kt = w;
        const int key = 32 * kt + r;
        f32x16 dv0 = zero16(), dv1 = zero16(), dk0 = zero16(), dk1 = zero16();
        for (int qt = 0; qt < NT; ++qt) {
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int d0 = 16 * ks + 8 * hh;
                s = MFMA_BF16(ld8(Qs + (32 * qt + r) * VLD + d0), ld8(Ks + key * VLD + d0), s);
                dp = MFMA_BF16(ld8(Ds + (32 * qt + r) * VLD + d0), ld8(Vs + key * VLD + d0), dp);
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int q = 32 * qt + crow(g, hh);
                const float p = (q < T && key < T) ? __expf(s[g] * scale - Ls[q]) : 0.f;
                s[g] = p;                       // P[q][key]
                dp[g] = p * (dp[g] - Del[q]);   // dS[q][key]
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {  // sum over q (X rows): X^T . B, B[k=q][col=d]
                const bf16x8 pa = cvt8(s, ks, 1.0f);
                const bf16x8 da = cvt8(dp, ks, scale);
                dv0 = MFMA_BF16(pa, ld8_col_perm(Ds, VLD, 32 * qt, ks, hh, r), dv0);
                dv1 = MFMA_BF16(pa, ld8_col_perm(Ds, VLD, 32 * qt, ks, hh, 32 + r), dv1);
                dk0 = MFMA_BF16(da, ld8_col_perm(Qs, VLD, 32 * qt, ks, hh, r), dk0);
                dk1 = MFMA_BF16(da, ld8_col_perm(Qs, VLD, 32 * qt, ks, hh, 32 + r), dk1);
            }
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {  // result rows = keys (regs), cols = d (lanes)
            const int kk = 32 * kt + crow(g, hh);
            if (kk < T) {
                dVg[(size_t)kk * rs + r] = (bf16_t)dv0[g];
                dVg[(size_t)kk * rs + 32 + r] = (bf16_t)dv1[g];
                dKg[(size_t)kk * rs + r] = (bf16_t)dk0[g];
                dKg[(size_t)kk * rs + 32 + r] = (bf16_t)dk1[g];
            }
        }
    }
    // ---- phase B: this wave owns query tile qt.  Tiles X'[key][q]: rows = key (regs), cols = q (lanes).
    else {
        const int qt = w - NT;
        const int q = 32 * qt + r;
        const float lq = Ls[q], dq_ = Del[q];
        f32x16 dq0 = zero16(), dq1 = zero16();
        for (int kt = 0; kt < NT; ++kt) {
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int d0 = 16 * ks + 8 * hh;
                s = MFMA_BF16(ld8(Ks + (32 * kt + r) * VLD + d0), ld8(Qs + q * VLD + d0), s);
                dp = MFMA_BF16(ld8(Vs + (32 * kt + r) * VLD + d0), ld8(Ds + q * VLD + d0), dp);
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int key = 32 * kt + crow(g, hh);
                const float p = (q < T && key < T) ? __expf(s[g] * scale - lq) : 0.f;
                dp[g] = p * (dp[g] - dq_);  // dS^T[key][q]
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {  // sum over key (X' rows): X'^T . B, B[k=key][col=d]
                const bf16x8 da = cvt8(dp, ks, scale);
                dq0 = MFMA_BF16(da, ld8_col_perm(Ks, VLD, 32 * kt, ks, hh, r), dq0);
                dq1 = MFMA_BF16(da, ld8_col_perm(Ks, VLD, 32 * kt, ks, hh, 32 + r), dq1);
            }
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int qq = 32 * qt + crow(g, hh);
            if (qq < T) {
                dQg[(size_t)qq * rs + r] = (bf16_t)dq0[g];
                dQg[(size_t)qq * rs + 32 + r] = (bf16_t)dq1[g];
            }
        }
    }
}

// ===================================================================== backward, f32
template <int MT>
__global__ __launch_bounds__(128 * MT) void attn_bwd_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                               const float* __restrict__ dout, const float* __restrict__ lse,
                                                               float* __restrict__ dqkv, int T, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int ROWS = 32 * MT;
    float* Qs = sm;
    float* Ks = Qs + ROWS * 64;
    float* Vs = Ks + ROWS * 64;
    float* Ds = Vs + ROWS * 64;
    float* Ls = Ds + ROWS * 64;
    float* Del = Ls + ROWS;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const size_t rs = (size_t)3 * H * HD, os = (size_t)H * HD;
    const float* Qg = qkv + (size_t)b * T * rs + (size_t)h * HD;
    const float* Og = out + (size_t)b * T * os + (size_t)h * HD;
    const float* Dg = dout + (size_t)b * T * os + (size_t)h * HD;
    const int NT = (T + 31) >> 5;

    stage_f32(Qs, Qg, rs, T, ROWS);
    stage_f32(Ks, Qg + os, rs, T, ROWS);
    stage_f32(Vs, Qg + 2 * os, rs, T, ROWS);
    stage_f32(Ds, Dg, os, T, ROWS);
    {
        const int q = tid >> 1, half = tid & 1;
        float acc = 0.f;
        if (q < T)
            for (int c = 0; c < 32; ++c) acc += Og[(size_t)q * os + half * 32 + c] * Dg[(size_t)q * os + half * 32 + c];
        acc += __shfl_xor(acc, 1);
        if (half == 0 && q < ROWS) {
            Del[q] = acc;
            Ls[q] = q < T ? lse[((size_t)b * H + h) * T + q] : 0.f;
        }
    }
    __syncthreads();

    float* dQg = dqkv + (size_t)b * T * rs + (size_t)h * HD;
    float* dKg = dQg + os;
    float* dVg = dKg + os;

    if (w < NT) {   // phase A (waves 0..NT-1): key tile kt; X[q][key]
        const int kt = w;
        const int key = 32 * kt + r;
        f32x16 dv0 = zero16(), dv1 = zero16(), dk0 = zero16(), dk1 = zero16();
        for (int qt = 0; qt < NT; ++qt) {
            f32x16 s = zero16(), dp = zero16();
            for (int ks = 0; ks < 32; ++ks) {
                const int d = 2 * ks + hh;
                s = MFMA_F32(Qs[rot(32 * qt + r, d)], Ks[rot(key, d)], s);
                dp = MFMA_F32(Ds[rot(32 * qt + r, d)], Vs[rot(key, d)], dp);
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int q = 32 * qt + crow(g, hh);
                const float p = (q < T && key < T) ? expf(s[g] * scale - Ls[q]) : 0.f;
                s[g] = p;
                dp[g] = p * (dp[g] - Del[q]) * scale;
            }
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {  // k-step ks consumes register ks: q = 32*qt + crow(ks,hh)
                const int q = 32 * qt + crow(ks, hh);
                dv0 = MFMA_F32(s[ks], Ds[rot(q, r)], dv0);
                dv1 = MFMA_F32(s[ks], Ds[rot(q, 32 + r)], dv1);
                dk0 = MFMA_F32(dp[ks], Qs[rot(q, r)], dk0);
                dk1 = MFMA_F32(dp[ks], Qs[rot(q, 32 + r)], dk1);
            }
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int kk = 32 * kt + crow(g, hh);
            if (kk < T) {
                dVg[(size_t)kk * rs + r] = dv0[g];
                dVg[(size_t)kk * rs + 32 + r] = dv1[g];
                dKg[(size_t)kk * rs + r] = dk0[g];
                dKg[(size_t)kk * rs + 32 + r] = dk1[g];
            }
        }
    }
    else {   // phase B (waves NT..2NT-1): query tile qt; X'[key][q]
        const int qt = w - NT;
        const int q = 32 * qt + r;
        const float lq = Ls[q], dq_ = Del[q];
        f32x16 dq0 = zero16(), dq1 = zero16();
        for (int kt = 0; kt < NT; ++kt) {
            f32x16 s = zero16(), dp = zero16();
            for (int ks = 0; ks < 32; ++ks) {
                const int d = 2 * ks + hh;
                s = MFMA_F32(Ks[rot(32 * kt + r, d)], Qs[rot(q, d)], s);
                dp = MFMA_F32(Vs[rot(32 * kt + r, d)], Ds[rot(q, d)], dp);
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int key = 32 * kt + crow(g, hh);
                const float p = (q < T && key < T) ? expf(s[g] * scale - lq) : 0.f;
                dp[g] = p * (dp[g] - dq_) * scale;
            }
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const int key = 32 * kt + crow(ks, hh);
                dq0 = MFMA_F32(dp[ks], Ks[rot(key, r)], dq0);
                dq1 = MFMA_F32(dp[ks], Ks[rot(key, 32 + r)], dq1);
            }
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int qq = 32 * qt + crow(g, hh);
            if (qq < T) {
                dQg[(size_t)qq * rs + r] = dq0[g];
                dQg[(size_t)qq * rs + 32 + r] = dq1[g];
            }
        }
    }
}

static int attn_check(const void* a, const void* b, int B, int T, int H, int dtype) {
    if (!a || !b || B < 0 || T < 1 || H < 1) return GM3D_EINVAL;
    if (T > 128) return GM3D_EUNSUPPORTED;
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    if ((long long)B * H > 0x7fffffffLL) return GM3D_EUNSUPPORTED;
    return GM3D_OK;
}

}  // namespace gm3d

template <class K, class... A>
static int launch_attn(K kernel, int grid, int threads, size_t lds, hipStream_t st, A... args) {
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return GM3D_ELAUNCH;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds, st, args...);
    return hipGetLastError() == hipSuccess ? GM3D_OK : GM3D_ELAUNCH;
}

extern "C" int gm3d_attention_fwd(const void* qkv, void* out, float* lse, int B, int T, int H, float scale,
                                  int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    int rc = attn_check(qkv, out, B, T, H, dtype);
    if (rc != GM3D_OK) return rc;
    if (B == 0) return GM3D_OK;
    hipStream_t st = (hipStream_t)stream;
    const int threads = 64 * ((T + 31) / 32);
    const int rows = T <= 64 ? 64 : 128;
    if (dtype == GM3D_BF16) {
        return T <= 64 ? launch_attn(attn_fwd_bf16_kernel<2>, B * H, threads, 0, st, (const bf16_t*)qkv, (bf16_t*)out, lse, T, H, scale)
                       : launch_attn(attn_fwd_bf16_kernel<4>, B * H, threads, 0, st, (const bf16_t*)qkv, (bf16_t*)out, lse, T, H, scale);
    }
    const size_t lds = sizeof(float) * 3 * rows * 64;
    return T <= 64 ? launch_attn(attn_fwd_f32_kernel<2>, B * H, threads, lds, st, (const float*)qkv, (float*)out, lse, T, H, scale)
                   : launch_attn(attn_fwd_f32_kernel<4>, B * H, threads, lds, st, (const float*)qkv, (float*)out, lse, T, H, scale);
}

extern "C" int gm3d_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                                  int B, int T, int H, float scale, int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    int rc = attn_check(qkv, out, B, T, H, dtype);
    if (rc != GM3D_OK) return rc;
    if (!dout || !lse || !dqkv) return GM3D_EINVAL;
    if (B == 0) return GM3D_OK;
    hipStream_t st = (hipStream_t)stream;
    const int threads = 128 * ((T + 31) / 32);   // NT key-tile waves + NT query-tile waves
    const int rows = T <= 64 ? 64 : 128;
    if (dtype == GM3D_BF16) {
        const size_t lds = (size_t)4 * rows * VLD * sizeof(bf16_t) + 2 * rows * sizeof(float);
        return T <= 64 ? launch_attn(attn_bwd_bf16_kernel<2>, B * H, threads, lds, st, (const bf16_t*)qkv, (const bf16_t*)out,
                                     (const bf16_t*)dout, lse, (bf16_t*)dqkv, T, H, scale)
                       : launch_attn(attn_bwd_bf16_kernel<4>, B * H, threads, lds, st, (const bf16_t*)qkv, (const bf16_t*)out,
                                     (const bf16_t*)dout, lse, (bf16_t*)dqkv, T, H, scale);
    }
    const size_t lds = sizeof(float) * (4 * rows * 64 + 2 * rows);
    return T <= 64 ? launch_attn(attn_bwd_f32_kernel<2>, B * H, threads, lds, st, (const float*)qkv, (const float*)out,
                                 (const float*)dout, lse, (float*)dqkv, T, H, scale)
                   : launch_attn(attn_bwd_f32_kernel<4>, B * H, threads, lds, st, (const float*)qkv, (const float*)out,
                                 (const float*)dout, lse, (float*)dqkv, T, H, scale);
}
