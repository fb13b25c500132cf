// C[M,N] = A[M,K] . W[N,K]^T (+ bias[N])  -- bf16 operands, fp32 accumulation, bf16 result -- for the skinny GEMMs of the
// transformer blocks and the mini-PointNet (K = 128..1536, N = 256..1536, M = 3200..262144 rows).
//
// Beneath: every nn.Linear / Conv1d(k=1) of the path (timm Block: qkv, proj, fc1, fc2 -- in-tree twin
// Point-MAE_SA3D/models/Point_MAE.py:82-125; Encoder convs models_mae_learn_loss.py:873-882), forward (W = the weight as
// stored, (out,in) row-major) and input-gradient (W = the transposed bf16 shadow the optimizer maintains).
//
// Why (MI355X): these GEMMs are one wave of tiles over 256 CUs with only K/64 = 6..24 pipeline stages, and the library
// kernels spend most of their ~12-20 us in prologue/epilogue (tools/gemm_profile.py: 75-610 TFLOP/s).  This kernel is
// shaped for exactly that regime: 128x128 tile per 256-thread workgroup (2 per CU), BK = 64, register-prefetched
// double-buffered LDS stages with ONE barrier per stage, v_mfma_f32_32x32x16_bf16 with the output computed transposed
// (lanes = rows of C, registers = 4 consecutive columns) and an LDS-staged epilogue that adds the bias in fp32 and
// writes 16-byte row-contiguous pieces.
#include "common.hpp"
#include <stdlib.h>

namespace gm3d {

typedef __bf16 gbf16x8 __attribute__((ext_vector_type(8)));
typedef float gf32x16 __attribute__((ext_vector_type(16)));

constexpr int GBM = 128, GBN = 128, GBK = 64;
constexpr int GPITCH = 72;                     // bf16 elements per LDS row (64 + 8 pad: conflict-free 16-byte operand reads)
constexpr int GCP = 132;                       // floats per row of the fp32 epilogue tile
constexpr int GSTAGE = (GBM + GBN) * GPITCH;   // bf16 elements per stage

// KTT: K / 64 when it is one of the path's values (fully unrolled: exact s_waitcnt counts), 0 = run-time loop.
// WMI: 32-row MFMA tiles per wave in M -- 2: 128x128 workgroup tile; 1: 64x128 (the 3200-row token streams, where 128-row tiles
// leave 40 % of the workgroup slots empty)
// LNA: the A operand is LayerNorm(U) of the residual stream U (M x 384), read from the bf16 copy U16 the producer wrote beside the
// fp32 stream; the per-row statistics (exact, from the fp32 values) arrive as three (mean, sum of squared deviations) pairs over
// 128-column tiles (gm3d_gemm_tn_bf16_res): every thread combines the pairs of its rows once (Chan's formula), then normalises,
// scales and shifts each 8-element chunk in registers on its way from global memory to the LDS stage -- models_mae_learn_loss.py's norm1 / norm2 (timm Block, Point-MAE_SA3D/models/Point_MAE.py:128-146) without a pass of
// their own.  The workgroups of column tile 0 also write the normalised rows (bf16, the weight-gradient GEMM's operand) and the
// row mean / rstd (the LayerNorm backward's inputs) when asked to.
// RAG (with KTT = 0, plain C (+ bias) output only): ragged shapes -- N and K any multiples of 8 (the 96 / 192 / 288 / 576-wide layers of
// Point-M2AE).  W rows past N are clamped (their columns are never stored), the last K-stage's chunks past K are replaced by zeros on
// their way into LDS (the load itself is redirected to the row's last valid chunk: always in bounds), column chunks past N are not stored.
template <int KTT, int WMI, bool LNA, bool RAG = false>
__global__ __launch_bounds__(256, 2) void gemm_tn_bf16_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                              const float* __restrict__ bias, bf16_t* __restrict__ C, int M,
                                                              int N, int K, int lda, int ldw, int ldc, int tiles_n,
                                                              int total_tiles, bf16_t* __restrict__ G, int ldg,
                                                              bf16_t* __restrict__ P, uint8_t* __restrict__ ARG, int ldp,
                                                              int bias_after_pool, const bf16_t* __restrict__ Fpre, int ldfp,
                                                              float* __restrict__ colpart, const float* __restrict__ Af,
                                                              const float* __restrict__ lnstats, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps, bf16_t* __restrict__ Hout,
                                                              float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    bf16_t* sm = reinterpret_cast<bf16_t*>(gsm);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    // XCD-aware tile order: the hardware deals workgroups round-robin over the 8 XCDs (each with its own L2), so workgroup b
    // takes logical tile (b % 8) * per_xcd + b / 8 -- every XCD walks a contiguous run of tiles in (tile_m, tile_n) order and
    // the tiles_n workgroups that share one block of A rows hit the same L2 (A comes from HBM once, not tiles_n times).
    const int per_xcd = gridDim.x >> 3;
    const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (logical >= total_tiles) return;
    const int tile_m = logical / tiles_n, tile_n = logical - tile_m * tiles_n;
    constexpr int BM = 64 * WMI, ACH = 2 * WMI;           // tile rows; 16-byte A chunks per thread per stage
    constexpr int STAGE = (BM + GBN) * GPITCH;
    const int m0 = tile_m * BM, n0 = tile_n * GBN;
    const int wm = (w >> 1) * 32 * WMI, wn = (w & 1) * 64;

    // global -> register prefetch, TWO stages deep: 4 chunks of A and 4 of W per thread per stage (16 bytes each), two register
    // sets.  Iteration kt issues the loads of stage kt+2, multiplies stage kt, and only then parks stage kt+1 (issued one
    // iteration ago) in the LDS buffer that stage kt-1 vacated: a load has a whole iteration plus a multiply to arrive.
    gbf16x8 pa[2][ACH], pw[2][4];
    float4 pg[2][2], pb[2][2];                                // LNA: the gamma / beta chunk of the stage
    float mu[ACH], rsd[ACH];
    const int crow = tid >> 3, ckc = (tid & 7) * 8;           // chunk c = tid + 256*i -> row = crow + 32*i, k offset ckc
    size_t aoff[ACH], woff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = crow + 32 * i;
        // rows past M are clamped, not predicated: a branch per load makes the compiler serialise the loads behind
        // s_waitcnt at every join; the clamped rows only feed output rows the epilogue never stores
        if (i < ACH) {
            const int am = m0 + row < M ? m0 + row : M - 1;
            aoff[i] = (size_t)am * lda + ckc;
            if (LNA) {
                // statistics of the 384-wide row from its three 128-wide tiles: mean of means, M2 = sum M2_t + 128 sum (m_t - m)^2
                const float m0_ = lnstats[((size_t)0 * M + am) * 2], q0 = lnstats[((size_t)0 * M + am) * 2 + 1];
                const float m1_ = lnstats[((size_t)1 * M + am) * 2], q1 = lnstats[((size_t)1 * M + am) * 2 + 1];
                const float m2_ = lnstats[((size_t)2 * M + am) * 2], q2 = lnstats[((size_t)2 * M + am) * 2 + 1];
                const float mean = (m0_ + m1_ + m2_) * (1.0f / 3.0f);
                const float d0 = m0_ - mean, d1 = m1_ - mean, d2 = m2_ - mean;
                const float m2sum = (q0 + q1 + q2) + 128.0f * (d0 * d0 + d1 * d1 + d2 * d2);
                mu[i] = mean;
                rsd[i] = rsqrtf(m2sum * (1.0f / 384.0f) + eps);
                if (mean_out && tile_n == 0 && (tid & 7) == 0 && m0 + row < M) {
                    mean_out[am] = mean;
                    rstd_out[am] = rsd[i];
                }
            }
        }
        woff[i] = (size_t)((RAG && n0 + row >= N) ? N - 1 : n0 + row) * ldw + ckc;
    }
    // RAG: K-stages incl. a partial last one; `dead`: this thread's chunk of the LAST stage lies past K
    const int KT = KTT > 0 ? KTT : (RAG ? (K + GBK - 1) / GBK : K / GBK);
    const int klast = (KT - 1) * GBK;
    const bool dead = RAG && ckc >= K - klast;
    const int kredir = K - 8 - ckc;                            // column offset that lands this thread on the row's last valid chunk
#define GM3D_LOAD_STAGE(SET, K0)                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                       \
        const int ko_ = (RAG && dead && (K0) == klast) ? kredir : (K0);                   \
        if (i < ACH) pa[SET][i] = *reinterpret_cast<const gbf16x8*>(A + aoff[i] + ko_);   \
        pw[SET][i] = *reinterpret_cast<const gbf16x8*>(W + woff[i] + ko_);                \
    }                                                                                     \
    if (LNA) {                                                                            \
        pg[SET][0] = *reinterpret_cast<const float4*>(gamma + ckc + (K0));                \
        pg[SET][1] = *reinterpret_cast<const float4*>(gamma + ckc + (K0) + 4);            \
        pb[SET][0] = *reinterpret_cast<const float4*>(beta + ckc + (K0));                 \
        pb[SET][1] = *reinterpret_cast<const float4*>(beta + ckc + (K0) + 4);             \
    }
#define GM3D_STORE_STAGE(SET, ST, K0)                                                     \
    {                                                                                     \
        bf16_t* as_ = sm + (ST) * STAGE;                                                  \
        bf16_t* ws_ = as_ + BM * GPITCH;                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                   \
            const int row = crow + 32 * i;                                                \
            if (i < ACH) {                                                                \
                if (LNA) {                                                                \
                    const gbf16x8 xv = pa[SET][i];                                        \
                    const float4 g0 = pg[SET][0], g1 = pg[SET][1], b0 = pb[SET][0], b1 = pb[SET][1];      \
                    const float m_ = mu[i], r_ = rsd[i];                                  \
                    gbf16x8 hv;                                                           \
                    hv[0] = (bf16_t)(((float)xv[0] - m_) * r_ * g0.x + b0.x);             \
                    hv[1] = (bf16_t)(((float)xv[1] - m_) * r_ * g0.y + b0.y);             \
                    hv[2] = (bf16_t)(((float)xv[2] - m_) * r_ * g0.z + b0.z);             \
                    hv[3] = (bf16_t)(((float)xv[3] - m_) * r_ * g0.w + b0.w);             \
                    hv[4] = (bf16_t)(((float)xv[4] - m_) * r_ * g1.x + b1.x);             \
                    hv[5] = (bf16_t)(((float)xv[5] - m_) * r_ * g1.y + b1.y);             \
                    hv[6] = (bf16_t)(((float)xv[6] - m_) * r_ * g1.z + b1.z);             \
                    hv[7] = (bf16_t)(((float)xv[7] - m_) * r_ * g1.w + b1.w);             \
                    *reinterpret_cast<gbf16x8*>(as_ + row * GPITCH + ckc) = hv;           \
                    if (Hout && tile_n == 0 && m0 + row < M)                              \
                        *reinterpret_cast<gbf16x8*>(Hout + (size_t)(m0 + row) * K + (K0) + ckc) = hv;     \
                } else if (RAG && dead && (K0) == klast) {                                \
                    gbf16x8 z_;                                                           \
                    _Pragma("unroll") for (int e = 0; e < 8; ++e) z_[e] = (bf16_t)0.f;    \
                    *reinterpret_cast<gbf16x8*>(as_ + row * GPITCH + ckc) = z_;           \
                } else                                                                    \
                    *reinterpret_cast<gbf16x8*>(as_ + row * GPITCH + ckc) = pa[SET][i];   \
            }                                                                             \
            if (RAG && dead && (K0) == klast) {                                           \
                gbf16x8 z_;                                                               \
                _Pragma("unroll") for (int e = 0; e < 8; ++e) z_[e] = (bf16_t)0.f;        \
                *reinterpret_cast<gbf16x8*>(ws_ + row * GPITCH + ckc) = z_;               \
            } else                                                                        \
                *reinterpret_cast<gbf16x8*>(ws_ + row * GPITCH + ckc) = pw[SET][i];       \
        }                                                                                 \
    }

    gf32x16 acc[WMI][2];
#pragma unroll
    for (int i = 0; i < WMI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

    GM3D_LOAD_STAGE(0, 0)
    if (KT > 1) GM3D_LOAD_STAGE(1, GBK)
    GM3D_STORE_STAGE(0, 0, 0)
    __syncthreads();
    // the loop is unrolled by two so that register-set indices are compile-time constants; with a compile-time KT it is
    // fully unrolled, which lets the compiler count outstanding loads exactly (vmcnt(8): stage kt+2 stays in flight while
    // stage kt+1 is parked) instead of draining them at every loop back-edge
#pragma unroll
    for (int kt = 0; kt < (KTT > 0 ? KTT : KT); kt += 2) {
#define GM3D_ITER(KT_, CUR, NXT)                                                                                           \
    if ((KT_) < KT) {                                                                                                      \
        if ((KT_) + 2 < KT) GM3D_LOAD_STAGE(CUR, ((KT_) + 2) * GBK) /* stage KT_+2 reuses register set CUR */              \
        const bf16_t* as = sm + (CUR) * STAGE;                                                                             \
        const bf16_t* ws = as + BM * GPITCH;                                                                               \
        /* all 16 operand fragments of the stage are requested from LDS before the first MFMA: one LDS round trip per */   \
        /* stage on the critical path instead of one per k-step (the compiler then drains lgkmcnt progressively)       */   \
        gbf16x8 fa[4][WMI], fw[4][2];                                                                                      \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                                    \
            const int ko = 16 * s + 8 * hh;                                                                                \
            _Pragma("unroll") for (int i = 0; i < WMI; ++i)                                                                \
                fa[s][i] = *reinterpret_cast<const gbf16x8*>(as + (wm + 32 * i + r) * GPITCH + ko);                        \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                  \
                fw[s][j] = *reinterpret_cast<const gbf16x8*>(ws + (wn + 32 * j + r) * GPITCH + ko);                        \
        }                                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);   /* keep the scheduler from sinking the reads back between the MFMAs */        \
        /* transposed product: rows (registers) = n, columns (lanes) = m */                                                \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                                      \
            _Pragma("unroll") for (int i = 0; i < WMI; ++i)                                                                \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                              \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[s][j], fa[s][i], acc[i][j], 0, 0, 0);           \
        if ((KT_) + 1 < KT) GM3D_STORE_STAGE(NXT, NXT, ((KT_) + 1) * GBK)   /* stage KT_+1 (register set NXT) -> LDS buffer NXT */ \
        __syncthreads();                                                                                                   \
    }
        GM3D_ITER(kt, 0, 1)
        GM3D_ITER(kt + 1, 1, 0)
    }
#undef GM3D_ITER
#undef GM3D_LOAD_STAGE
#undef GM3D_STORE_STAGE

    // epilogue: acc -> fp32 tile in LDS (row = m, 4 consecutive n per register quad) -> + bias -> bf16, 16-byte stores
    float* cs = reinterpret_cast<float*>(gsm);
#pragma unroll
    for (int i = 0; i < WMI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = wm + 32 * i + r, n = wn + 32 * j + 8 * q + 4 * hh;
                *reinterpret_cast<float4*>(cs + m * GCP + n) =
                    make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
            }
    __syncthreads();
    if (P) {
        // max-pool epilogue (mini-PointNet: max over the 32 points of a group, models_mae_learn_loss.py:893,897): the tile
        // holds 4 whole groups; thread = one column of two groups.  Values are rounded to bf16 BEFORE the comparison (first
        // maximum wins), which reproduces the separate GEMM -> gm3d_group_max_fwd path decision for decision.
        const int c = tid & 127, gh = tid >> 7;
        const float b = bias ? bias[n0 + c] : 0.f;
        const float pre = bias_after_pool ? 0.f : b, post = bias_after_pool ? b : 0.f;
#pragma unroll
        for (int gi = 0; gi < WMI; ++gi) {                       // BM/32 = 2*WMI groups per tile, WMI per thread-half
            const int rowbase = (gh * WMI + gi) * 32;
            if (m0 + rowbase < M) {
                float best = -INFINITY;
                int bk = 0;
#pragma unroll 8
                for (int k = 0; k < 32; ++k) {
                    const float v = (float)(bf16_t)(cs[(rowbase + k) * GCP + c] + pre);
                    if (v > best) { best = v; bk = k; }
                }
                const size_t o = (size_t)((m0 + rowbase) >> 5) * ldp + n0 + c;
                P[o] = (bf16_t)(best + post);
                ARG[o] = (uint8_t)bk;
            }
        }
        if (!C) return;
    }
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4 * WMI; ++i) {
        const int c = tid + 256 * i;                 // BM rows x 16 chunks of 8 columns
        const int row = c >> 4, nc = (c & 15) * 8;
        if (m0 + row < M && (!RAG || n0 + nc < N)) {
            float v[8];
            const float4 x = *reinterpret_cast<const float4*>(cs + row * GCP + nc), y = *reinterpret_cast<const float4*>(cs + row * GCP + nc + 4);
            v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
            float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (bias) {
                const float4 b0 = *reinterpret_cast<const float4*>(bias + n0 + nc), b1 = *reinterpret_cast<const float4*>(bias + n0 + nc + 4);
                bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
            }
            if (Fpre) {
                // fc2 input-gradient epilogue: C = bf16(dg) * GELU'(f + bias) (dg rounded to bf16 first, like a stored GEMM
                // result); per-tile column sums of the fp32 products -> colpart (the fc1 bias gradient, finished later)
                float fv[8];
                V8<bf16_t>::load(Fpre + (size_t)(m0 + row) * ldfp + n0 + nc, fv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] = (float)(bf16_t)v[e] * gelu_grad_f<bf16_t>(fv[e] + bv[e]);
                    csum[e] += v[e];
                }
                V8<bf16_t>::store(C + (size_t)(m0 + row) * ldc + n0 + nc, v);
            } else if (G) {
                // fc1 epilogue: C (optional) = the pre-activation WITHOUT bias, rounded to bf16 -- what the GELU backward
                // re-reads; G = GELU(bf16(f) + bias), from the rounded value so forward and backward see the same f
                if (C) V8<bf16_t>::store(C + (size_t)(m0 + row) * ldc + n0 + nc, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_f<bf16_t>((float)(bf16_t)v[e] + bv[e]);
                V8<bf16_t>::store(G + (size_t)(m0 + row) * ldg + n0 + nc, v);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bv[e];
                V8<bf16_t>::store(C + (size_t)(m0 + row) * ldc + n0 + nc, v);
            }
        }
    }
    if (Fpre) {      // the 16 threads that share a column chunk meet in LDS (the fp32 tile is no longer needed)
        __syncthreads();
        const int nc = (tid & 15) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[(tid >> 4) * GBN + nc + e] = csum[e];
        __syncthreads();
        if (tid < GBN) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += cs[k * GBN + tid];
            colpart[(size_t)tile_m * N + n0 + tid] = t;
        }
    }
}

// Batched bf16 transpose dst[b][c][r] = src[b][r][c] (64x64 LDS tiles, 16-byte global accesses on both sides): the per-step
// transposed weight shadows of the input-gradient GEMMs (20 blocks x (fc2 + proj) per step; the generic strided-copy kernel
// runs this at ~1 TB/s).
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int rows,
                                                             int cols, size_t src_bstride) {
    __shared__ bf16_t tile[64][66];
    const bf16_t* s = src + (size_t)blockIdx.z * src_bstride;
    bf16_t* d = dst + (size_t)blockIdx.z * rows * cols;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tr = threadIdx.x >> 3, tc = (threadIdx.x & 7) * 8;       // 32 rows x 8 chunks per pass
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int rr = tr + 32 * p;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (r0 + rr < rows && c0 + tc < cols) V8<bf16_t>::load(s + (size_t)(r0 + rr) * cols + c0 + tc, v);     // rows, cols % 8 == 0: a chunk is
#pragma unroll                                                                                                 // inside or outside as a whole
        for (int e = 0; e < 8; ++e) tile[rr][tc + e] = (bf16_t)v[e];
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int cc = tr + 32 * p;                                     // output row = source column
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)tile[tc + e][cc];
        if (c0 + cc < cols && r0 + tc < rows) V8<bf16_t>::store(d + (size_t)(c0 + cc) * rows + r0 + tc, v);
    }
}

// several batched transposes (the four transposed weight shadows a block stack's backward reads) in ONE launch: problem j owns blocks
// [first_j, first_j + (cols_j / 64) (rows_j / 64) batch_j); descriptors by value in the kernel arguments
struct TrProblem {
    const bf16_t* src;
    bf16_t* dst;
    long long src_bstride;
    int rows, cols, first;
};
constexpr int TR_MAXP = 8;
struct TrMulti {
    int count;
    TrProblem p[TR_MAXP];
};

__global__ __launch_bounds__(256) void transpose_bf16_multi_kernel(TrMulti m) {
    __shared__ bf16_t tile[64][66];
    TrProblem q = m.p[0];
#pragma unroll
    for (int j = 1; j < TR_MAXP; ++j)
        if (j < m.count && (int)blockIdx.x >= m.p[j].first) q = m.p[j];
    int local = (int)blockIdx.x - q.first;
    const int nx = (q.cols + 63) / 64, ny = (q.rows + 63) / 64;
    const int bx = local % nx; local /= nx;
    const int by = local % ny;
    const int bz = local / ny;
    const bf16_t* s = q.src + (size_t)bz * q.src_bstride;
    bf16_t* d = q.dst + (size_t)bz * q.rows * q.cols;
    const int r0 = by * 64, c0 = bx * 64;
    const int tr = threadIdx.x >> 3, tc = (threadIdx.x & 7) * 8;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int rr = tr + 32 * p;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (r0 + rr < q.rows && c0 + tc < q.cols) V8<bf16_t>::load(s + (size_t)(r0 + rr) * q.cols + c0 + tc, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[rr][tc + e] = (bf16_t)v[e];
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int cc = tr + 32 * p;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)tile[tc + e][cc];
        if (c0 + cc < q.cols && r0 + tc < q.rows) V8<bf16_t>::store(d + (size_t)(c0 + cc) * q.rows + r0 + tc, v);
    }
}

}  // namespace gm3d

// 64-row tiles for the short token streams (3200 rows x N/128 column tiles is 75-300 workgroups for 512 slots), 128 otherwise
static int gemm_tile_height(int M) {
    return M <= 4096 ? 64 : 128;
}

struct GemmLn {      // LayerNorm-on-load arguments (gm3d_gemm_tn_bf16_lna*)
    const float* U = nullptr; const float* stats = nullptr; const float* gamma = nullptr; const float* beta = nullptr; float eps = 0.f;
    void* H = nullptr; float* mean = nullptr; float* rstd = nullptr;
};

static int gemm_launch(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int lda, int ldw, int ldc,
                       void* G, int ldg, gm3d_stream_t stream, void* P = nullptr, uint8_t* ARG = nullptr, int ldp = 0,
                       int bias_after_pool = 0, const void* Fpre = nullptr, int ldfp = 0, float* colpart = nullptr,
                       const GemmLn* ln = nullptr) {
    using namespace gm3d;
    if (ln && (!ln->stats || !ln->gamma || !ln->beta || K != 384 || (ln->mean && !ln->rstd))) return GM3D_EINVAL;
    if (!A || !W || (!C && !G && !P) || M < 0 || N < 1 || K < 1) return GM3D_EINVAL;
    if (P && (!ARG || M % 32 || ldp < N)) return GM3D_EINVAL;
    const bool ragged = N % GBN || K % GBK;
    if (ragged && (ln || G || P || Fpre || !C || N % 8 || K % 8 || K < 8)) return GM3D_EUNSUPPORTED;   // ragged: the plain product only
    if (lda % 8 || ldw % 8 || lda < K || ldw < K) return GM3D_EUNSUPPORTED;
    if ((C && (ldc % 8 || ldc < N)) || (G && (ldg % 8 || ldg < N))) return GM3D_EUNSUPPORTED;
    if (M == 0) return GM3D_OK;
    const int bm = ln ? 64 : gemm_tile_height(M);     // LayerNorm-on-load: 64-row tiles (the 128-row form runs out of registers)
    const int tiles_m = (M + bm - 1) / bm, tiles_n = (N + GBN - 1) / GBN;
    if ((long long)tiles_m * tiles_n > 0x7fffffffLL) return GM3D_EUNSUPPORTED;
    const size_t lds_ab = (size_t)2 * (bm + GBN) * GPITCH * sizeof(bf16_t), lds_c = (size_t)bm * GCP * sizeof(float);
    const size_t lds = lds_ab > lds_c ? lds_ab : lds_c;
    const int total = tiles_m * tiles_n, grid = (total + 7) / 8 * 8;
#define GM3D_GEMM_LAUNCH_(KTT, WMI, LNA)                                                                                       \
    {                                                                                                                         \
        static LdsAttr attr;                                                                                            \
        if (!attr.ensure((const void*)gemm_tn_bf16_kernel<KTT, WMI, LNA>, lds)) return GM3D_ELAUNCH;                    \
        hipLaunchKernelGGL((gemm_tn_bf16_kernel<KTT, WMI, LNA>), dim3(grid), dim3(256), lds, (hipStream_t)stream,             \
                           (const bf16_t*)A, (const bf16_t*)W, bias, (bf16_t*)C, M, N, K, lda, ldw, ldc, tiles_n, total,      \
                           (bf16_t*)G, ldg, (bf16_t*)P, ARG, ldp, bias_after_pool, (const bf16_t*)Fpre, ldfp, colpart,        \
                           nullptr, ln ? ln->stats : nullptr, ln ? ln->gamma : nullptr, ln ? ln->beta : nullptr,  \
                           ln ? ln->eps : 0.f, ln ? (bf16_t*)ln->H : nullptr, ln ? ln->mean : nullptr, ln ? ln->rstd : nullptr); \
    }
#define GM3D_GEMM_LAUNCH(KTT, WMI) GM3D_GEMM_LAUNCH_(KTT, WMI, false)
    if (ragged) {
#define GM3D_GEMM_RAG(WMI)                                                                                                    \
    {                                                                                                                         \
        static LdsAttr attr;                                                                                                  \
        if (!attr.ensure((const void*)gemm_tn_bf16_kernel<0, WMI, false, true>, lds)) return GM3D_ELAUNCH;                    \
        hipLaunchKernelGGL((gemm_tn_bf16_kernel<0, WMI, false, true>), dim3(grid), dim3(256), lds, (hipStream_t)stream,       \
                           (const bf16_t*)A, (const bf16_t*)W, bias, (bf16_t*)C, M, N, K, lda, ldw, ldc, tiles_n, total,      \
                           nullptr, 0, nullptr, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f,  \
                           nullptr, nullptr, nullptr);                                                                        \
    }
        if (bm == 64) GM3D_GEMM_RAG(1) else GM3D_GEMM_RAG(2)
#undef GM3D_GEMM_RAG
        GM3D_CHECK_LAUNCH();
        return GM3D_OK;
    }
    if (ln) {       // K = 384: six stages
        GM3D_GEMM_LAUNCH_(6, 1, true)
        GM3D_CHECK_LAUNCH();
        return GM3D_OK;
    }
#define GM3D_GEMM_CASE(KTT)                                                                                                   \
    case KTT:                                                                                                                 \
        if (bm == 64) GM3D_GEMM_LAUNCH(KTT, 1) else GM3D_GEMM_LAUNCH(KTT, 2)                                                  \
        break;
    switch (K / GBK) {
        GM3D_GEMM_CASE(2) GM3D_GEMM_CASE(4) GM3D_GEMM_CASE(6) GM3D_GEMM_CASE(8) GM3D_GEMM_CASE(16) GM3D_GEMM_CASE(18)
        GM3D_GEMM_CASE(24)
        default:
            if (bm == 64) GM3D_GEMM_LAUNCH(0, 1) else GM3D_GEMM_LAUNCH(0, 2)
    }
#undef GM3D_GEMM_LAUNCH
#undef GM3D_GEMM_LAUNCH_
#undef GM3D_GEMM_CASE
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_gemm_tn_bf16(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int lda,
                                 int ldw, int ldc, gm3d_stream_t stream) {
    if (!C) return GM3D_EINVAL;
    return gemm_launch(A, W, bias, C, M, N, K, lda, ldw, ldc, nullptr, 0, stream);
}

extern "C" int gm3d_gemm_tn_bf16_gelu(const void* A, const void* W, const float* bias, void* F, void* G, int M, int N, int K,
                                      int lda, int ldw, int ldf, int ldg, gm3d_stream_t stream) {
    if (!G || !bias) return GM3D_EINVAL;
    return gemm_launch(A, W, bias, F, M, N, K, lda, ldw, ldf, G, ldg, stream);
}

extern "C" int gm3d_gemm_tn_bf16_pool(const void* A, const void* W, const float* bias, void* C, void* P, uint8_t* arg, int M, int N,
                                      int K, int lda, int ldw, int ldc, int ldp, int bias_after_pool, gm3d_stream_t stream) {
    if (!P || !arg) return GM3D_EINVAL;
    return gemm_launch(A, W, bias, C, M, N, K, lda, ldw, ldc, nullptr, 0, stream, P, arg, ldp, bias_after_pool);
}

extern "C" int gm3d_gemm_tn_bf16_gelu_bwd(const void* dO, const void* Wt, const void* F, const float* bias, void* dF, float* colpart,
                                          int M, int N, int K, int lda, int ldw, int ldf, int lddf, gm3d_stream_t stream) {
    if (!F || !bias || !dF || !colpart || ldf % 8 || ldf < N) return GM3D_EINVAL;
    return gemm_launch(dO, Wt, bias, dF, M, N, K, lda, ldw, lddf, nullptr, 0, stream, nullptr, nullptr, 0, 0, F, ldf, colpart);
}

extern "C" int gm3d_gemm_tn_bf16_lna(const void* U16, const float* stats, const float* gamma, const float* beta, float eps, const void* W,
                                     const float* bias, void* C, void* G, void* H, float* mean, float* rstd, int M, int N, int K, int ldu,
                                     int ldw, int ldc, int ldg, gm3d_stream_t stream) {
    GemmLn ln;
    ln.stats = stats; ln.gamma = gamma; ln.beta = beta; ln.eps = eps; ln.H = H; ln.mean = mean; ln.rstd = rstd;
    if (G && !bias) return GM3D_EINVAL;
    return gemm_launch(U16, W, bias, C, M, N, K, ldu, ldw, ldc, G, ldg, stream, nullptr, nullptr, 0, 0, nullptr, 0, nullptr, &ln);
}

extern "C" int gm3d_gemm_tile_rows(int M) { return M < 1 ? 0 : (M + gemm_tile_height(M) - 1) / gemm_tile_height(M); }

// `count` (<= 8) batched transposes dst[j][b][c][r] = src[j][b][r][c] in ONE launch (rows, cols multiples of 8; src batches at
// src_batch_stride[j] elements, dst dense).
extern "C" int gm3d_transpose_bf16_multi(int count, const void* const* src, void* const* dst, const int* batch, const int* rows, const int* cols,
                                         const long long* src_batch_stride, gm3d_stream_t stream) {
    using namespace gm3d;
    if (count < 1 || count > TR_MAXP || !src || !dst || !batch || !rows || !cols || !src_batch_stride) return GM3D_EINVAL;
    TrMulti m;
    m.count = count;
    long long first = 0;
    for (int j = 0; j < count; ++j) {
        if (!src[j] || !dst[j] || batch[j] < 1 || rows[j] < 1 || cols[j] < 1 || src_batch_stride[j] < (long long)rows[j] * cols[j]) return GM3D_EINVAL;
        if (rows[j] % 8 || cols[j] % 8) return GM3D_EUNSUPPORTED;
        m.p[j].src = (const bf16_t*)src[j]; m.p[j].dst = (bf16_t*)dst[j]; m.p[j].src_bstride = src_batch_stride[j];
        m.p[j].rows = rows[j]; m.p[j].cols = cols[j]; m.p[j].first = (int)first;
        first += (long long)((cols[j] + 63) / 64) * ((rows[j] + 63) / 64) * batch[j];
        if (first > 0x7ffffff0LL) return GM3D_EUNSUPPORTED;
    }
    for (int j = count; j < TR_MAXP; ++j) m.p[j] = m.p[0];
    hipLaunchKernelGGL(transpose_bf16_multi_kernel, dim3((unsigned)first), dim3(256), 0, (hipStream_t)stream, m);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_transpose_bf16_batched(const void* src, void* dst, int batch, int rows, int cols, long long src_batch_stride,
                                           gm3d_stream_t stream) {
    using namespace gm3d;
    if (!src || !dst || batch < 0 || rows < 1 || cols < 1 || src_batch_stride < (long long)rows * cols) return GM3D_EINVAL;
    if (rows % 8 || cols % 8 || batch > 65535) return GM3D_EUNSUPPORTED;
    if (batch == 0) return GM3D_OK;
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((cols + 63) / 64, (rows + 63) / 64, batch), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                       (bf16_t*)dst, rows, cols, (size_t)src_batch_stride);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}
