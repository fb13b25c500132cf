// Masked multi-head attention for the hierarchical (Point-M2AE) encoder: up to 512 tokens per sample, head_dim 16 / 32 / 64,
// a per-sample boolean mask (local-radius mask of the centres + padding of the variable-length visible sets), forward and backward.
//
// Beneath: the Block / Attention of Point-M2AE's H_Encoder (hyper-parameters
// Point-M2AE_SA3D/cfgs/config_Point_M2AE.yaml:57-99: dims 96/192/384, 6 heads, local_radius 0.32/0.64/1.28).  The reference ships
// no source for this model (Point-M2AE_SA3D/README.md:1); the operator follows the published block -- softmax((q k^T) * scale
// with masked pairs removed) v on the timm qkv layout (B,T,3,H,hd) -- and is checked against oracle/hier_ref.py.
//
// mask: bitset (B, T, W = ceil(T/32)) of uint32, bit (j & 31) of word j >> 5 of row i set = pair (i, j) NOT allowed; it must be
// SYMMETRIC (radius masks and padding masks are), which lets the key-major phase of the backward read it row-wise as well.
// Keys >= T count as masked.  A query whose every key is masked (a padding row) gets a zero output and zero gradients.
//
// bf16 (throughput mode), flash-style on the matrix cores: one wave per 32-query tile walks the key tiles with a running
// max / sum per query (a query is a lane, its keys are registers: S^T = K Q^T), probabilities go into the P.V product as the B
// operand straight from the accumulators, V^T / Q^T / dO^T / K^T fragments come out of row-major LDS images by
// ds_read_b64_tr_b16.  K and V (forward) or Q, K, V, dO (backward) of the whole head sit in LDS (<= 512 x 64 bf16 each).
// f32 (parity mode): plain one-thread-per-query / one-thread-per-key kernels -- exact fp32 dot products in a fixed order.
#include "common.hpp"

namespace gm3d {

typedef __bf16 mbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 mbf16x4 __attribute__((ext_vector_type(4)));
typedef float mf32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) mbf16x4 lds_mbf16x4;

#define MMFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

__device__ __forceinline__ int mcrow(int g, int hh) { return (g & 3) + 8 * (g >> 2) + 4 * hh; }
__device__ __forceinline__ mf32x16 mzero16() {
    mf32x16 z;
#pragma unroll
    for (int g = 0; g < 16; ++g) z[g] = 0.f;
    return z;
}
__device__ __forceinline__ mbf16x8 mzero8() {
    mbf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (bf16_t)0.0f;
    return z;
}
__device__ __forceinline__ mbf16x8 mcvt8(const mf32x16& x, int s, float mul) {
    mbf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16_t)(x[8 * s + j] * mul);
    return v;
}

// LDS image of a (rows, HD) bf16 matrix: row pitch HD*2 + 16 bytes (48 / 80 / 144: 16-byte row reads of 16 consecutive rows hit
// 16 different bank quads; transposed reads are at worst 2-way)
template <int HD> struct MImg {
    static constexpr int PITCH = HD * 2 + 16;
    static __device__ __forceinline__ mbf16x8 row(const unsigned char* img, int r, int ch) {     // 8 consecutive d starting at 8*ch
        return *reinterpret_cast<const mbf16x8*>(img + r * PITCH + ch * 16);
    }
    // fragment A[i = column][k] in accumulator-as-operand k order: rows k0 + 16 s + 4 hh + {0..3, 8..11}, column colblk + (lane&15)
    static __device__ __forceinline__ mbf16x8 tr8(const unsigned char* img, int k0, int s, int hh, int colblk, int lane) {
        const int l16 = lane & 15, q = l16 >> 2, p = l16 & 3;
        int col = colblk + 4 * p;
        if (col >= HD) col -= HD * (col / HD);            // lanes past the head dim read valid (ignored) data: pad, don't mask
        const int r0 = k0 + 16 * s + 4 * hh + q;
        const mbf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_mbf16x4*)(img + r0 * PITCH + col * 2));
        const mbf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_mbf16x4*)(img + (r0 + 8) * PITCH + col * 2));
        mbf16x8 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
        return v;
    }
    // cooperative copy of one head's (T, HD) slice into a zero-padded `rows`-row image
    static __device__ __forceinline__ void stage(unsigned char* img, const bf16_t* src, size_t row_stride, int T, int rows, int tid,
                                                 int nthreads) {
        constexpr int CH = HD / 8;
        for (int c = tid; c < rows * CH; c += nthreads) {
            const int r = c / CH, ch = c - r * CH;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r < T) v = *reinterpret_cast<const uint4*>(src + (size_t)r * row_stride + ch * 8);
            *reinterpret_cast<uint4*>(img + r * PITCH + ch * 16) = v;
        }
    }
};

// Workgroup -> (cloud * head, tile block) with the workgroups of ONE cloud on ONE XCD.  The hardware deals consecutive workgroup ids to
// the 8 XCDs round-robin, and the qkv layout (B,T,3,H,hd) interleaves the heads of a cloud within every 2 H hd-byte row (32-byte
// pieces at hd = 16): with id = cloud * H + head the six heads of a cloud pull the same lines into six different L2s (PMC, T = 512,
// hd = 16: 527 MB fetched for 63 MB of operands in the backward).  Remapped, id' = (id % 8) * (total / 8) + id / 8 runs through the
// (tile block, head, cloud) triples in order on each XCD.
__device__ __forceinline__ void mattn_block(int& bh, int& yb) {
    const int nx = gridDim.x, ny = gridDim.y, total = nx * ny;
    int lin = blockIdx.x + nx * blockIdx.y;
    if ((total & 7) == 0) lin = (lin & 7) * (total >> 3) + (lin >> 3);
    yb = lin % ny;
    bh = lin / ny;
}

// word `kt` of mask row `i` with the keys >= T of that word forced on; rows >= T: everything masked
__device__ __forceinline__ unsigned mask_word(const unsigned* mrow /* row i, or nullptr */, bool row_ok, int kt, int T) {
    unsigned w = mrow ? mrow[kt] : 0u;
    const int rem = T - 32 * kt;                    // valid keys in this word
    if (rem < 32) w |= rem <= 0 ? 0xffffffffu : (0xffffffffu << rem);
    return row_ok ? w : 0xffffffffu;
}

// ===================================================================== forward, bf16
// grid (B*H, ceil(T / (32 NWV))), 64 NWV threads: wave w owns query tile NWV*blockIdx.y + w.  NWV = 8 where the images fit twice per
// CU (HD <= 32): the K / V images of a head are staged once per 256 queries and a CU holds 16 waves -- a wave's key-tile loop is one
// dependent chain (MFMA -> mask -> max -> exp -> sum -> MFMA), so what hides its latency is other waves (136 -> see profiles/ at T = 512).
template <int HD, int NWV>
__global__ __launch_bounds__(64 * NWV) void mattn_fwd_bf16_kernel(const bf16_t* __restrict__ qkv, const unsigned* __restrict__ mask,
                                                             bf16_t* __restrict__ out, float* __restrict__ lse, int T, int H, int W,
                                                             float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char msm[];
    using I = MImg<HD>;
    constexpr int KS = HD / 16;                     // k-steps of the q.k product
    constexpr int DT = (HD + 31) / 32;              // 32-row tiles of the transposed output
    const int NK = (T + 31) >> 5, rowsK = 32 * NK;
    unsigned char* Ki = msm;
    unsigned char* Vi = Ki + rowsK * I::PITCH;
    unsigned char* Qi = Vi + rowsK * I::PITCH;
    int bh_, yb_;
    mattn_block(bh_, yb_);
    const int b = bh_ / H, h = bh_ % H;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const size_t rs = (size_t)3 * H * HD, os = (size_t)H * HD;
    const bf16_t* Qg = qkv + (size_t)b * T * rs + (size_t)h * HD;
    const int q0 = 32 * NWV * yb_;
    I::stage(Ki, Qg + os, rs, T, rowsK, tid, 64 * NWV);
    I::stage(Vi, Qg + 2 * os, rs, T, rowsK, tid, 64 * NWV);
    I::stage(Qi, Qg + (size_t)q0 * rs, rs, T - q0, 32 * NWV, tid, 64 * NWV);
    __syncthreads();
    const int q = q0 + 32 * w + r;                  // this lane's query
    if (q0 + 32 * w >= T) return;                   // whole tile past the end (wave-uniform)
    const bool q_ok = q < T;
    const unsigned* mrow = mask ? mask + ((size_t)b * T + (q_ok ? q : 0)) * W : nullptr;
    mbf16x8 fq[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) fq[s] = I::row(Qi, 32 * w + r, 2 * s + hh);
    float m = -INFINITY, l = 0.f;
    mf32x16 o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) o[d] = mzero16();
    const int cb = 16 * ((lane >> 4) & 1);
    for (int kt = 0; kt < NK; ++kt) {
        // a key tile that is blocked for every query of this wave's tile contributes exp(-inf) = 0 to every sum and leaves the
        // running maximum alone: skipping it changes no bit.  With the visible tokens moved to the front of the cloud
        // (gm3d_partition_visible) the filler rows behind them form such tiles, as keys and as queries.
        const unsigned mw = mask_word(mrow, q_ok, kt, T);
        if (__ballot(mw != 0xffffffffu) == 0ull) continue;
        mf32x16 st = mzero16();
#pragma unroll
        for (int s = 0; s < KS; ++s) st = MMFMA(I::row(Ki, 32 * kt + r, 2 * s + hh), fq[s], st);
        float tm = -INFINITY;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            st[g] = ((mw >> mcrow(g, hh)) & 1u) ? -INFINITY : st[g] * scale;
            tm = fmaxf(tm, st[g]);
        }
        tm = fmaxf(tm, __shfl_xor(tm, 32));
        const float mn = fmaxf(m, tm);
        const float alpha = mn == -INFINITY ? 1.f : __expf(m - mn);       // m = -inf: exp(-inf) = 0 (nothing accumulated yet)
        float ts = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            st[g] = mn == -INFINITY ? 0.f : __expf(st[g] - mn);
            ts += st[g];
        }
        ts += __shfl_xor(ts, 32);
        l = l * alpha + ts;
        m = mn;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 16; ++g) o[d][g] *= alpha;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const mbf16x8 p = mcvt8(st, s, 1.0f);
#pragma unroll
            for (int d = 0; d < DT; ++d) o[d] = MMFMA(I::tr8(Vi, 32 * kt, s, hh, 32 * d + cb, lane), p, o[d]);
        }
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    if (q_ok) {
        bf16_t* og = out + ((size_t)b * T + q) * os + (size_t)h * HD;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d0 = 32 * d + 8 * g4 + 4 * hh;
                if (d0 < HD) {
                    mbf16x4 pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk[e] = (bf16_t)(o[d][4 * g4 + e] * inv);
                    *reinterpret_cast<mbf16x4*>(og + d0) = pk;
                }
            }
        if (hh == 0) lse[((size_t)b * H + h) * T + q] = l > 0.f ? m + logf(l) : 0.f;
    }
}

// ===================================================================== backward, bf16
// grid (B*H, ceil(T / (32 NTP))), 128 NTP threads: waves 0..NTP-1 own key tiles NTP c + w (dK, dV), waves NTP..2 NTP-1 own query
// tiles NTP c + w - NTP (dQ).  NTP = 8 (16 waves, one workgroup per CU) for HD <= 32: the four images of a head are staged half as
// often and twice as many waves share a CU.
template <int HD, int NTP>
__global__ __launch_bounds__(128 * NTP) void mattn_bwd_bf16_kernel(const bf16_t* __restrict__ qkv, const unsigned* __restrict__ mask,
                                                             const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout,
                                                             const float* __restrict__ lse, bf16_t* __restrict__ dqkv, int T, int H,
                                                             int W, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char msm[];
    using I = MImg<HD>;
    constexpr int KS = HD / 16, DT = (HD + 31) / 32;
    const int NT = (T + 31) >> 5, rows = 32 * NT;
    unsigned char* Qi = msm;
    unsigned char* Ki = Qi + rows * I::PITCH;
    unsigned char* Vi = Ki + rows * I::PITCH;
    unsigned char* Di = Vi + rows * I::PITCH;
    float* Ls = reinterpret_cast<float*>(Di + rows * I::PITCH);
    float* Del = Ls + rows;
    int bh_, yb_;
    mattn_block(bh_, yb_);
    const int b = bh_ / H, h = bh_ % H;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const size_t rs = (size_t)3 * H * HD, os = (size_t)H * HD;
    const bf16_t* Qg = qkv + (size_t)b * T * rs + (size_t)h * HD;
    const bf16_t* Og = out + (size_t)b * T * os + (size_t)h * HD;
    const bf16_t* Dg = dout + (size_t)b * T * os + (size_t)h * HD;
    constexpr int NTH = 128 * NTP;
    I::stage(Qi, Qg, rs, T, rows, tid, NTH);
    I::stage(Ki, Qg + os, rs, T, rows, tid, NTH);
    I::stage(Vi, Qg + 2 * os, rs, T, rows, tid, NTH);
    I::stage(Di, Dg, os, T, rows, tid, NTH);
    for (int i = tid; i < rows; i += NTH) {         // delta[q] = sum_d dO[q][d] * O[q][d]
        float acc = 0.f;
        if (i < T) {
#pragma unroll
            for (int c = 0; c < HD / 8; ++c) {
                const mbf16x8 o8 = *reinterpret_cast<const mbf16x8*>(Og + (size_t)i * os + c * 8);
                const mbf16x8 d8 = *reinterpret_cast<const mbf16x8*>(Dg + (size_t)i * os + c * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += (float)o8[j] * (float)d8[j];
            }
        }
        Del[i] = acc;
        Ls[i] = i < T ? lse[((size_t)b * H + h) * T + i] : 0.f;
    }
    __syncthreads();
    bf16_t* dQg = dqkv + (size_t)b * T * rs + (size_t)h * HD;
    const int cb = 16 * ((lane >> 4) & 1);
    const int tile = NTP * yb_ + (w < NTP ? w : w - NTP);
    if (32 * tile >= T) return;                     // wave-uniform; no barrier follows
    const int row = 32 * tile + r;                  // this lane's key (waves 0..3) or query (waves 4..7)
    const bool row_ok = row < T;
    const unsigned* mrow = mask ? mask + ((size_t)b * T + (row_ok ? row : 0)) * W : nullptr;
    mf32x16 x[DT], y[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) { x[d] = mzero16(); y[d] = mzero16(); }
    if (w < NTP) {
        // phase A: X[q][key] tiles (rows = q in registers, cols = key on the lane); the mask is symmetric: row `key`, word qt
        mbf16x8 fk[KS], fv[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) { fk[s] = I::row(Ki, row, 2 * s + hh); fv[s] = I::row(Vi, row, 2 * s + hh); }
        for (int qt = 0; qt < NT; ++qt) {
            const unsigned mw = mask_word(mrow, row_ok, qt, T);
            if (__ballot(mw != 0xffffffffu) == 0ull) continue;        // every pair of this tile blocked: p = 0 throughout (exact)
            mf32x16 s = mzero16(), dp = mzero16();
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s = MMFMA(I::row(Qi, 32 * qt + r, 2 * ks + hh), fk[ks], s);
                dp = MMFMA(I::row(Di, 32 * qt + r, 2 * ks + hh), fv[ks], dp);
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int q = 32 * qt + mcrow(g, hh);
                const float p = ((mw >> mcrow(g, hh)) & 1u) ? 0.f : __expf(s[g] * scale - Ls[q]);
                s[g] = p;
                dp[g] = p * (dp[g] - Del[q]);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const mbf16x8 pa = mcvt8(s, ks, 1.0f), da = mcvt8(dp, ks, scale);
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    x[d] = MMFMA(I::tr8(Di, 32 * qt, ks, hh, 32 * d + cb, lane), pa, x[d]);      // dV^T[d][key]
                    y[d] = MMFMA(I::tr8(Qi, 32 * qt, ks, hh, 32 * d + cb, lane), da, y[d]);      // dK^T[d][key]
                }
            }
        }
    } else {
        // phase B: X'[key][q] tiles (rows = key in registers, cols = q on the lane)
        mbf16x8 fq[KS], fd[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) { fq[s] = I::row(Qi, row, 2 * s + hh); fd[s] = I::row(Di, row, 2 * s + hh); }
        const float lq = Ls[row], dq_ = Del[row];
        for (int kt = 0; kt < NT; ++kt) {
            const unsigned mw = mask_word(mrow, row_ok, kt, T);
            if (__ballot(mw != 0xffffffffu) == 0ull) continue;
            mf32x16 s = mzero16(), dp = mzero16();
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s = MMFMA(I::row(Ki, 32 * kt + r, 2 * ks + hh), fq[ks], s);
                dp = MMFMA(I::row(Vi, 32 * kt + r, 2 * ks + hh), fd[ks], dp);
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float p = ((mw >> mcrow(g, hh)) & 1u) ? 0.f : __expf(s[g] * scale - lq);
                dp[g] = p * (dp[g] - dq_);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const mbf16x8 da = mcvt8(dp, ks, scale);
#pragma unroll
                for (int d = 0; d < DT; ++d) x[d] = MMFMA(I::tr8(Ki, 32 * kt, ks, hh, 32 * d + cb, lane), da, x[d]);   // dQ^T[d][q]
            }
        }
    }
    if (row_ok) {
        bf16_t* g0 = dQg + (size_t)row * rs + (w < NTP ? 2 * os : 0);     // dV (key waves) / dQ (query waves)
        bf16_t* g1 = dQg + (size_t)row * rs + os;                         // dK
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d0 = 32 * d + 8 * g4 + 4 * hh;
                if (d0 < HD) {
                    mbf16x4 pk, pk2;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { pk[e] = (bf16_t)x[d][4 * g4 + e]; pk2[e] = (bf16_t)y[d][4 * g4 + e]; }
                    *reinterpret_cast<mbf16x4*>(g0 + d0) = pk;
                    if (w < NTP) *reinterpret_cast<mbf16x4*>(g1 + d0) = pk2;
                }
            }
    }
}

// ===================================================================== f32 (parity mode): plain kernels
__device__ __forceinline__ bool mbit(const unsigned* mask, size_t row_base, int W, int j) {
    return mask && ((mask[row_base * W + (j >> 5)] >> (j & 31)) & 1u);
}

// one thread per (b, h, query): two passes over the keys (max, then exp-sum and P.V)
template <int HD>
__global__ __launch_bounds__(128) void mattn_fwd_f32_kernel(const float* __restrict__ qkv, const unsigned* __restrict__ mask,
                                                            float* __restrict__ out, float* __restrict__ lse, int B, int T, int H, int W,
                                                            float scale) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)B * H * T) return;
    const int q = (int)(gid % T), h = (int)((gid / T) % H), b = (int)(gid / ((long long)T * H));
    const size_t rs = (size_t)3 * H * HD, os = (size_t)H * HD;
    const float* base = qkv + (size_t)b * T * rs + (size_t)h * HD;
    float qv[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) qv[d] = base[(size_t)q * rs + d];
    const size_t mr = (size_t)b * T + q;
    float m = -INFINITY;
    for (int j = 0; j < T; ++j) {
        if (mbit(mask, mr, W, j)) continue;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) s = fmaf(qv[d], base[(size_t)j * rs + os + d], s);
        m = fmaxf(m, s * scale);
    }
    float l = 0.f, o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
    if (m > -INFINITY)
        for (int j = 0; j < T; ++j) {
            if (mbit(mask, mr, W, j)) continue;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) s = fmaf(qv[d], base[(size_t)j * rs + os + d], s);
            const float p = expf(s * scale - m);
            l += p;
#pragma unroll
            for (int d = 0; d < HD; ++d) o[d] = fmaf(p, base[(size_t)j * rs + 2 * os + d], o[d]);
        }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) out[((size_t)b * T + q) * os + (size_t)h * HD + d] = o[d] * inv;
    lse[((size_t)b * H + h) * T + q] = l > 0.f ? m + logf(l) : 0.f;
}

// one thread per (b, h, row): `which` = 0: the row is a query (dQ); 1: the row is a key (dK, dV; the mask is symmetric)
template <int HD>
__global__ __launch_bounds__(128) void mattn_bwd_f32_kernel(const float* __restrict__ qkv, const unsigned* __restrict__ mask,
                                                            const float* __restrict__ out, const float* __restrict__ dout,
                                                            const float* __restrict__ lse, float* __restrict__ dqkv, int B, int T, int H,
                                                            int W, float scale) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)B * H * T) return;
    const int i = (int)(gid % T), h = (int)((gid / T) % H), b = (int)(gid / ((long long)T * H));
    const int which = blockIdx.y;
    const size_t rs = (size_t)3 * H * HD, os = (size_t)H * HD;
    const float* base = qkv + (size_t)b * T * rs + (size_t)h * HD;
    const float* ob = out + (size_t)b * T * os + (size_t)h * HD;
    const float* db = dout + (size_t)b * T * os + (size_t)h * HD;
    const float* lb = lse + ((size_t)b * H + h) * T;
    float* gb = dqkv + (size_t)b * T * rs + (size_t)h * HD;
    const size_t mr = (size_t)b * T + i;
    if (which == 0) {
        float qv[HD], dv[HD], acc[HD];
        float delta = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            qv[d] = base[(size_t)i * rs + d];
            dv[d] = db[(size_t)i * os + d];
            delta = fmaf(dv[d], ob[(size_t)i * os + d], delta);
            acc[d] = 0.f;
        }
        const float li = lb[i];
        for (int j = 0; j < T; ++j) {
            if (mbit(mask, mr, W, j)) continue;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                s = fmaf(qv[d], base[(size_t)j * rs + os + d], s);
                dp = fmaf(dv[d], base[(size_t)j * rs + 2 * os + d], dp);
            }
            const float ds = expf(s * scale - li) * (dp - delta) * scale;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] = fmaf(ds, base[(size_t)j * rs + os + d], acc[d]);
        }
#pragma unroll
        for (int d = 0; d < HD; ++d) gb[(size_t)i * rs + d] = acc[d];
    } else {
        float kv[HD], vv[HD], ak[HD], av[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            kv[d] = base[(size_t)i * rs + os + d];
            vv[d] = base[(size_t)i * rs + 2 * os + d];
            ak[d] = av[d] = 0.f;
        }
        for (int q = 0; q < T; ++q) {
            if (mbit(mask, mr, W, q)) continue;            // symmetric mask: (key i, query q) == (query q, key i)
            float s = 0.f, dp = 0.f, delta = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                const float qd = base[(size_t)q * rs + d], dd = db[(size_t)q * os + d];
                s = fmaf(qd, kv[d], s);
                dp = fmaf(dd, vv[d], dp);
                delta = fmaf(dd, ob[(size_t)q * os + d], delta);
            }
            const float p = expf(s * scale - lb[q]);
            const float ds = p * (dp - delta) * scale;
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                ak[d] = fmaf(ds, base[(size_t)q * rs + d], ak[d]);
                av[d] = fmaf(p, db[(size_t)q * os + d], av[d]);
            }
        }
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            gb[(size_t)i * rs + os + d] = ak[d];
            gb[(size_t)i * rs + 2 * os + d] = av[d];
        }
    }
}

// mask bitset of one level straight from its centres: bit j of row i set iff token i or j is not visible, or their centres are
// at squared distance >= radius^2 (same separately rounded fp32 expression as common.hpp::sqdist3 / the oracle).
// grid (ceil(G W / 256), B): one thread per (i, word) of cloud blockIdx.y, the cloud's centres and visibility staged in LDS (the 32
// partners of a word were 32 dependent global loads per thread: 110 us at B = 128, G = 512).
__global__ __launch_bounds__(256) void radius_mask_bits_kernel(const float* __restrict__ center, const unsigned char* __restrict__ vis,
                                                               float radius, int B, int G, int W, unsigned* __restrict__ bits, int inv) {
    extern __shared__ float rsm[];               // x[G] | y[G] | z[G] | vis[G] (as float flags)
    float* cx = rsm;
    float* cy = cx + G;
    float* cz = cy + G;
    float* cv = cz + G;
    const int b = blockIdx.y;
    const float* c = center + (size_t)b * G * 3;
    for (int j = threadIdx.x; j < G; j += 256) {
        cx[j] = c[3 * j]; cy[j] = c[3 * j + 1]; cz[j] = c[3 * j + 2];
        cv[j] = (!vis || ((vis[(size_t)b * G + j] != 0) != (inv != 0))) ? 1.f : 0.f;      // inv: the flags say "masked"
    }
    __syncthreads();
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= G * W) return;
    const int wd = id / G, i = id - wd * G;         // lanes = consecutive rows i of ONE word: the partner j is the same for the whole wave (LDS
    //                                                  broadcast); with lanes = consecutive words the 16 partners 32 floats apart share a bank
    const float r2 = __fmul_rn(radius, radius);
    const float ax = cx[i], ay = cy[i], az = cz[i];
    const bool vi = cv[i] != 0.f;
    unsigned w = 0;
    for (int t = 0; t < 32; ++t) {
        const int j = 32 * wd + t;
        bool blocked = true;
        if (j < G && vi && cv[j] != 0.f) blocked = radius > 0.f && sqdist3(ax, ay, az, cx[j], cy[j], cz[j]) >= r2;
        w |= blocked ? (1u << t) : 0u;
    }
    bits[((size_t)b * G + i) * W + wd] = w;
}

static int MATTN_WIDE = 1;      // 8 tiles per workgroup for HD <= 32, T > 128 (gm3d_attention_masked_set_wide: the A/B knob)

static int mattn_check(const void* a, const void* b, int B, int T, int H, int HDv, int dtype) {
    if (!a || !b || B < 0 || T < 1 || H < 1) return GM3D_EINVAL;
    if (HDv != 16 && HDv != 32 && HDv != 64) return GM3D_EUNSUPPORTED;
    if (T > 512) return GM3D_EUNSUPPORTED;
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    if ((long long)B * H > 0x7fffffffLL / 4) return GM3D_EUNSUPPORTED;
    return GM3D_OK;
}

template <class K>
static int mattn_attr(K kernel, size_t lds) {
    if (lds > 160 * 1024) return GM3D_EUNSUPPORTED;
    if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return GM3D_ELAUNCH;
    return GM3D_OK;
}

}  // namespace gm3d

extern "C" int gm3d_attention_masked_set_wide(int on) {
    gm3d::MATTN_WIDE = on ? 1 : 0;
    return GM3D_OK;
}

extern "C" int gm3d_radius_mask_bits(const float* center, const unsigned char* vis, float radius, int B, int G, unsigned* bits,
                                     gm3d_stream_t stream) {
    return gm3d_radius_mask_bits_m(center, vis, 0, radius, B, G, bits, stream);
}

extern "C" int gm3d_radius_mask_bits_m(const float* center, const unsigned char* flags, int flags_are_masked, float radius, int B, int G,
                                       unsigned* bits, gm3d_stream_t stream) {
    using namespace gm3d;
    const unsigned char* vis = flags;
    const int inv = flags_are_masked ? 1 : 0;
    if (!center || !bits || B < 0 || G < 1) return GM3D_EINVAL;
    if (B == 0) return GM3D_OK;
    const int W = (G + 31) / 32;
    if (G > 8192 || B > 65535) return GM3D_EUNSUPPORTED;
    hipLaunchKernelGGL(radius_mask_bits_kernel, dim3((unsigned)((G * W + 255) / 256), (unsigned)B), dim3(256), (size_t)4 * G * sizeof(float),
                       (hipStream_t)stream, center, vis, radius, B, G, W, bits, inv);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_attention_masked_fwd(const void* qkv, const unsigned* mask, void* out, float* lse, int B, int T, int H, int HD,
                                         float scale, int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    int rc = mattn_check(qkv, out, B, T, H, HD, dtype);
    if (rc != GM3D_OK) return rc;
    if (!lse) return GM3D_EINVAL;
    if (B == 0) return GM3D_OK;
    hipStream_t st = (hipStream_t)stream;
    const int W = (T + 31) / 32;
    if (dtype == GM3D_BF16) {
        const int rowsK = 32 * W;
        const int nwv = (HD <= 32 && T > 128 && MATTN_WIDE) ? 8 : 4;          // query tiles (= waves) per workgroup
        const size_t lds = (size_t)(2 * rowsK + 32 * nwv) * (HD * 2 + 16);
        const dim3 grid(B * H, (T + 32 * nwv - 1) / (32 * nwv));
#define GM3D_MF(HD_, NWV_)                                                                                             \
    {                                                                                                                  \
        rc = mattn_attr(mattn_fwd_bf16_kernel<HD_, NWV_>, lds);                                                        \
        if (rc != GM3D_OK) return rc;                                                                                  \
        hipLaunchKernelGGL((mattn_fwd_bf16_kernel<HD_, NWV_>), grid, dim3(64 * NWV_), lds, st, (const bf16_t*)qkv, mask, (bf16_t*)out, lse, \
                           T, H, W, scale);                                                                           \
    }
        if (HD == 16) { if (nwv == 8) GM3D_MF(16, 8) else GM3D_MF(16, 4) }
        else if (HD == 32) { if (nwv == 8) GM3D_MF(32, 8) else GM3D_MF(32, 4) }
        else GM3D_MF(64, 4)
#undef GM3D_MF
    } else {
        const long long n = (long long)B * H * T;
        const dim3 grid((unsigned)((n + 127) / 128));
#define GM3D_MF(HD_) hipLaunchKernelGGL(mattn_fwd_f32_kernel<HD_>, grid, dim3(128), 0, st, (const float*)qkv, mask, (float*)out, lse, B, T, H, W, scale);
        if (HD == 16) GM3D_MF(16) else if (HD == 32) GM3D_MF(32) else GM3D_MF(64)
#undef GM3D_MF
    }
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_attention_masked_bwd(const void* qkv, const unsigned* mask, const void* out, const void* dout, const float* lse,
                                         void* dqkv, int B, int T, int H, int HD, float scale, int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    int rc = mattn_check(qkv, out, B, T, H, HD, dtype);
    if (rc != GM3D_OK) return rc;
    if (!dout || !lse || !dqkv) return GM3D_EINVAL;
    if (B == 0) return GM3D_OK;
    hipStream_t st = (hipStream_t)stream;
    const int W = (T + 31) / 32;
    if (dtype == GM3D_BF16) {
        const int rows = 32 * W;
        const size_t lds = (size_t)4 * rows * (HD * 2 + 16) + (size_t)2 * rows * sizeof(float);
        const int ntp = (HD <= 32 && T > 128 && MATTN_WIDE) ? 8 : 4;          // key tiles (= query tiles) per workgroup
        const dim3 grid(B * H, (T + 32 * ntp - 1) / (32 * ntp));
#define GM3D_MB(HD_, NTP_)                                                                                             \
    {                                                                                                                  \
        rc = mattn_attr(mattn_bwd_bf16_kernel<HD_, NTP_>, lds);                                                        \
        if (rc != GM3D_OK) return rc;                                                                                  \
        hipLaunchKernelGGL((mattn_bwd_bf16_kernel<HD_, NTP_>), grid, dim3(128 * NTP_), lds, st, (const bf16_t*)qkv, mask, (const bf16_t*)out, \
                           (const bf16_t*)dout, lse, (bf16_t*)dqkv, T, H, W, scale);                                  \
    }
        if (HD == 16) { if (ntp == 8) GM3D_MB(16, 8) else GM3D_MB(16, 4) }
        else if (HD == 32) { if (ntp == 8) GM3D_MB(32, 8) else GM3D_MB(32, 4) }
        else GM3D_MB(64, 4)
#undef GM3D_MB
    } else {
        const long long n = (long long)B * H * T;
        const dim3 grid((unsigned)((n + 127) / 128), 2);
#define GM3D_MB(HD_) hipLaunchKernelGGL(mattn_bwd_f32_kernel<HD_>, grid, dim3(128), 0, st, (const float*)qkv, mask, (const float*)out, (const float*)dout, lse, (float*)dqkv, B, T, H, W, scale);
        if (HD == 16) GM3D_MB(16) else if (HD == 32) GM3D_MB(32) else GM3D_MB(64)
#undef GM3D_MB
    }
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}
