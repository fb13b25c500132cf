// C[M,N] = A[M,K] . W[N,K]^T (+ bias[N]), bf16 in / fp32 accumulate / bf16 out -- the LONG-K, FEW-TILE members of the path's GEMM
// family: fc2 (K = 1536 -> N = 384), the input gradients of fc1 (K = 1536) and qkv (K = 1152), at 3200 .. 8192 rows.
//
// Beneath: the same nn.Linear layers as csrc/gemm.hip (timm Block mlp.fc2 forward, mlp.fc1 / attn.qkv input gradients; in-tree twin
// Point-MAE_SA3D/models/Point_MAE.py:82-125).
//
// Why a second kernel (MI355X).  With N = 384 a launch has only 3 column tiles: 150 .. 192 workgroups for 256 CUs, ONE per CU,
// each walking 18 .. 24 K-stages.  A CU then has only its own loads in flight, and the register-prefetch pipeline of gemm.hip
// (two stages = 48 KB) gets about half the ~70 GB/s a CU can pull from L2 (DESIGN 3b'): the tuned library wins 0.65x there.
// This kernel keeps FOUR stages in an LDS ring filled by LDS-DMA (global_load_lds_dwordx4, no staging registers), two of them in
// flight behind a counted s_waitcnt vmcnt, one raw s_barrier per stage.  The LDS image is the conflict-free 128-byte-row image of
// csrc/attention.hip (16-byte chunk ch of row r at 128 r + 16 (ch ^ f(r))); LDS-DMA writes lane-linear, so the XOR is applied
// to the per-lane SOURCE address.  MFMA loop and epilogue as in gemm.hip (transposed product: lanes = rows of C).
#include "common.hpp"

namespace gm3d {

typedef __bf16 rbf16x8 __attribute__((ext_vector_type(8)));
typedef float rf32x16 __attribute__((ext_vector_type(16)));

constexpr int RBN = 128, RBK = 64, RNBUF = 4, RCP = 132;

__device__ __forceinline__ int ring_f(int row) { return (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1); }
__device__ __forceinline__ int ring_off(int row, int ch) { return row * 128 + ((ch ^ ring_f(row)) << 4); }

__device__ __forceinline__ void ring_glds16(const void* gsrc, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
}

// WMI: 32-row MFMA tiles per wave in M (1: 64-row workgroup tile, 2: 128-row)
// RES: residual epilogue of a transformer block's proj / fc2 (N = 384 = the LayerNorm width): instead of C the kernel writes the
//   fp32 residual stream  U = res + rowscale[row / rows_per_sample] * (A.W^T + bias) (+ add)   -- what gm3d_residual_ln_fwd
//   computes before it normalises --, a bf16 copy U16 of it (the consumer's A operand) and, per row and 128-column tile, the (mean, sum of squared deviations) of those 128
//   values: stats[tile_n][row][2].  The LayerNorm itself is applied by the consumer GEMM while it stages its A operand
//   (gm3d_gemm_tn_bf16_lna), so the normalised rows never make an HBM round trip of their own.
// DEPTH: K-stages requested ahead of the one being multiplied (ring of DEPTH + 2 slots).  What bounds this kernel is the bytes a CU
// has in flight towards L2 (about 1 us of loaded latency: 48 KB in flight = 50 GB/s per CU at DEPTH 2), so deeper is faster until
// the ring no longer fits: DEPTH 4 (144 KB) for 64-row tiles, DEPTH 3 (160 KB) for 128-row tiles.
template <int WMI, bool RES, int DEPTH>
__global__ __launch_bounds__(256) void gemm_tn_ring_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                           const float* __restrict__ bias, bf16_t* __restrict__ C, int M, int N, int K,
                                                           int lda, int ldw, int ldc, int tiles_n, int total_tiles,
                                                           const float* __restrict__ res, const float* __restrict__ rowscale,
                                                           int rows_per_sample, const bf16_t* __restrict__ add, float* __restrict__ U,
                                                           float* __restrict__ stats, bf16_t* __restrict__ U16) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];
    constexpr int BM = 64 * WMI;
    constexpr int STAGE = (BM + RBN) * 128;                 // bytes
    constexpr int PA = BM / 8, PIECES = (PA + 16) / 4;      // 1-KiB pieces of the A tile; pieces per wave per stage (6 or 8)
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int per_xcd = gridDim.x >> 3;
    const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (logical >= total_tiles) return;
    const int tile_m = logical / tiles_n, tile_n = logical - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * RBN;
    const int wm = (w >> 1) * 32 * WMI, wn = (w & 1) * 64;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)rsm;

    // this lane's share of a stage: pieces p = w + 4 i (i < PIECES); piece p < PA: A rows 8p .. 8p+7, else W rows 8(p-PA) ..
    const int prow = lane >> 3, pslot = lane & 7;
    const bf16_t* src[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int p = w + 4 * i;
        if (p < PA) {
            const int row = 8 * p + prow;
            const int am = m0 + row < M ? m0 + row : M - 1;          // rows past M: clamped (their outputs are never stored)
            src[i] = A + (size_t)am * lda + ((pslot ^ ring_f(row)) << 3);
        } else {
            const int row = 8 * (p - PA) + prow;
            const int wn_ = n0 + row < N ? n0 + row : N - 1;          // columns past N (a ragged last tile): clamped, never stored
            src[i] = W + (size_t)wn_ * ldw + ((pslot ^ ring_f(row)) << 3);
        }
    }
    constexpr int NBUF = DEPTH + 2;
#define GM3D_RING_STAGE(ST)                                                                          \
    {                                                                                                \
        const unsigned base = lds0 + ((ST) % NBUF) * STAGE;                                          \
        _Pragma("unroll") for (int i = 0; i < PIECES; ++i)                                           \
            ring_glds16(src[i] + (size_t)(ST) * RBK, base + 1024 * (w + 4 * i));                     \
    }

    rf32x16 acc[WMI][2];
#pragma unroll
    for (int i = 0; i < WMI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

    const int KT = K / RBK;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < KT) GM3D_RING_STAGE(d)
    for (int kt = 0; kt < KT; ++kt) {
        // counted waits: stage kt has landed when at most (stages issued after it) * PIECES loads of this wave are outstanding
        if (kt + DEPTH < KT) {
            GM3D_RING_STAGE(kt + DEPTH)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH * PIECES) : "memory");
        } else {
            const int rem = KT - 1 - kt;
            if (DEPTH > 3 && rem == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PIECES) : "memory");
            else if (DEPTH > 2 && rem == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
            else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const unsigned char* as = rsm + (kt % NBUF) * STAGE;
        const unsigned char* ws = as + BM * 128;
        rbf16x8 fa[4][WMI], fw[4][2];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int i = 0; i < WMI; ++i) fa[s][i] = *reinterpret_cast<const rbf16x8*>(as + ring_off(wm + 32 * i + r, 2 * s + hh));
#pragma unroll
            for (int j = 0; j < 2; ++j) fw[s][j] = *reinterpret_cast<const rbf16x8*>(ws + ring_off(wn + 32 * j + r, 2 * s + hh));
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < WMI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[s][j], fa[s][i], acc[i][j], 0, 0, 0);
    }
#undef GM3D_RING_STAGE
    __syncthreads();                 // every wave is done with the ring: it becomes the fp32 staging tile of the epilogue
    float* cs = reinterpret_cast<float*>(rsm);
#pragma unroll
    for (int i = 0; i < WMI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = wm + 32 * i + r, n = wn + 32 * j + 8 * q + 4 * hh;
                *reinterpret_cast<float4*>(cs + m * RCP + n) =
                    make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
            }
    __syncthreads();
    if (RES) {
#pragma unroll
        for (int i = 0; i < 4 * WMI; ++i) {
            const int c = tid + 256 * i;
            const int row = c >> 4, nc = (c & 15) * 8;          // the 16 lanes that hold one row's 128 columns are neighbours
            const bool valid = m0 + row < M;
            const int gr = valid ? m0 + row : M - 1;
            float v[8], rv[8];
            const float4 x = *reinterpret_cast<const float4*>(cs + row * RCP + nc), y = *reinterpret_cast<const float4*>(cs + row * RCP + nc + 4);
            v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
            // the product enters the residual stream rounded to bf16, exactly like the stored product of the two-kernel path
            // (GEMM -> gm3d_residual_ln_fwd): u = res + rowscale * (bf16(acc) + bias) + add
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)(bf16_t)v[e];
            if (bias) {
                const float4 b0 = *reinterpret_cast<const float4*>(bias + n0 + nc), b1 = *reinterpret_cast<const float4*>(bias + n0 + nc + 4);
                v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
            }
            const float rs = rowscale ? rowscale[gr / rows_per_sample] : 1.0f;
            V8<float>::load(res + (size_t)gr * N + n0 + nc, rv);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = rv[e] + rs * v[e];
            if (add) {
                float av[8];
                V8<bf16_t>::load(add + (size_t)gr * N + n0 + nc, av);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += av[e];
            }
            if (valid) {
                V8<float>::store(U + (size_t)gr * N + n0 + nc, v);
                V8<bf16_t>::store(U16 + (size_t)gr * N + n0 + nc, v);          // what the consumer GEMM stages (half the bytes)
            }
            float sm = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) sm += v[e];
            sm += __shfl_xor(sm, 1); sm += __shfl_xor(sm, 2); sm += __shfl_xor(sm, 4); sm += __shfl_xor(sm, 8);
            const float mu = sm * (1.0f / 128.0f);
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[e] - mu; q += d * d; }
            q += __shfl_xor(q, 1); q += __shfl_xor(q, 2); q += __shfl_xor(q, 4); q += __shfl_xor(q, 8);
            if (valid && (c & 15) == 0) {
                float* sp = stats + ((size_t)tile_n * M + gr) * 2;
                sp[0] = mu;
                sp[1] = q;
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4 * WMI; ++i) {
        const int c = tid + 256 * i;
        const int row = c >> 4, nc = (c & 15) * 8;
        if (m0 + row < M && n0 + nc < N) {
            float v[8];
            const float4 x = *reinterpret_cast<const float4*>(cs + row * RCP + nc), y = *reinterpret_cast<const float4*>(cs + row * RCP + nc + 4);
            v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
            if (bias) {
                const float4 b0 = *reinterpret_cast<const float4*>(bias + n0 + nc), b1 = *reinterpret_cast<const float4*>(bias + n0 + nc + 4);
                v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
            }
            V8<bf16_t>::store(C + (size_t)(m0 + row) * ldc + n0 + nc, v);
        }
    }
}


// ---- 96-column form: N = 384 as FOUR column tiles ----------------------------------------------------------------------------
// With 128-column tiles an N = 384 product has 3 tiles per row block: 150 (3200 rows) / 192 (4096, or 8192 in 128-row tiles)
// workgroups for 256 CUs, and what bounds each of them is the bytes its CU pulls from L2 (DESIGN 3b').  Four 96-column tiles per
// row block give 200 / 256 / 256 workgroups that each pull (BM + 96) instead of (BM + 128) operand rows per K-step: every CU
// busy AND 17 % (64-row) / 12.5 % (128-row) fewer bytes per workgroup, with no split-K fix-up and no cross-workgroup hand-off.
// Three waves (192 threads), wave w owns columns 32 w .. 32 w + 31 of the tile and all its rows; the same four-stage LDS-DMA ring,
// the same swizzled image and the same MFMA order along K as above -> results bit-identical to the 128-column kernel.
constexpr int R96 = 96, RCP96 = 100;

template <int WMI>
__global__ __launch_bounds__(192) void gemm_tn_ring96_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                             const float* __restrict__ bias, bf16_t* __restrict__ C, int M, int N, int K,
                                                             int lda, int ldw, int ldc, int tiles_n, int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];
    constexpr int BM = 64 * WMI, NMT = 2 * WMI;             // 32-row MFMA tiles per wave
    constexpr int STAGE = (BM + R96) * 128;                 // bytes
    constexpr int PA = BM / 8, NP = PA + 12;                // 1-KiB pieces per stage: 20 / 28
    constexpr int PMAX = (NP + 2) / 3;                      // pieces of wave 0 (7 / 10); waves 1, 2 may have one fewer
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int per_xcd = gridDim.x >> 3;
    const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (logical >= total_tiles) return;
    const int tile_m = logical / tiles_n, tile_n = logical - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * R96;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)rsm;
    const int mine = (NP - w + 2) / 3;                      // pieces p = w + 3 i < NP

    const int prow = lane >> 3, pslot = lane & 7;
    const bf16_t* src[PMAX];
#pragma unroll
    for (int i = 0; i < PMAX; ++i) {
        const int p = w + 3 * i < NP ? w + 3 * i : w;       // (a piece past NP is never issued; keep the pointer valid)
        if (p < PA) {
            const int row = 8 * p + prow;
            const int am = m0 + row < M ? m0 + row : M - 1;
            src[i] = A + (size_t)am * lda + ((pslot ^ ring_f(row)) << 3);
        } else {
            const int row = 8 * (p - PA) + prow;
            src[i] = W + (size_t)(n0 + row) * ldw + ((pslot ^ ring_f(row)) << 3);
        }
    }
#define GM3D_R96_STAGE(ST)                                                                           \
    {                                                                                                \
        const unsigned base = lds0 + ((ST) % RNBUF) * STAGE;                                         \
        _Pragma("unroll") for (int i = 0; i < PMAX; ++i)                                             \
            if (i < mine) ring_glds16(src[i] + (size_t)(ST) * RBK, base + 1024 * (w + 3 * i));       \
    }
    rf32x16 acc[NMT];
#pragma unroll
    for (int i = 0; i < NMT; ++i)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[i][g] = 0.f;

    const int KT = K / RBK;
    GM3D_R96_STAGE(0)
    if (KT > 1) GM3D_R96_STAGE(1)
    for (int kt = 0; kt < KT; ++kt) {
        // counted waits: this wave has `mine` loads per stage in flight (PMAX or PMAX - 1; uniform per wave)
        if (kt + 2 < KT) {
            GM3D_R96_STAGE(kt + 2)
            if (mine == PMAX) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PMAX) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (PMAX - 1)) : "memory");
        } else if (kt + 1 < KT) {
            if (mine == PMAX) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PMAX) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PMAX - 1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const unsigned char* as = rsm + (kt % RNBUF) * STAGE;
        const unsigned char* ws = as + BM * 128;
        rbf16x8 fa[4][NMT], fw[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int i = 0; i < NMT; ++i) fa[s][i] = *reinterpret_cast<const rbf16x8*>(as + ring_off(32 * i + r, 2 * s + hh));
            fw[s] = *reinterpret_cast<const rbf16x8*>(ws + ring_off(32 * w + r, 2 * s + hh));
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < NMT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[s], fa[s][i], acc[i], 0, 0, 0);
    }
#undef GM3D_R96_STAGE
    __syncthreads();
    float* cs = reinterpret_cast<float*>(rsm);              // [BM][RCP96] fp32 staging tile (25 / 50 KiB of the ring)
#pragma unroll
    for (int i = 0; i < NMT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = 32 * i + r, n = 32 * w + 8 * q + 4 * hh;
            *reinterpret_cast<float4*>(cs + m * RCP96 + n) = make_float4(acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]);
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < BM * 12 / 192; ++i) {                // 12 chunks of 8 columns per row
        const int c = tid + 192 * i;
        const int row = c / 12, nc = (c - row * 12) * 8;
        if (m0 + row < M) {
            float v[8];
            const float4 x = *reinterpret_cast<const float4*>(cs + row * RCP96 + nc), y = *reinterpret_cast<const float4*>(cs + row * RCP96 + nc + 4);
            v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
            if (bias) {
                const float4 b0 = *reinterpret_cast<const float4*>(bias + n0 + nc), b1 = *reinterpret_cast<const float4*>(bias + n0 + nc + 4);
                v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
            }
            V8<bf16_t>::store(C + (size_t)(m0 + row) * ldc + n0 + nc, v);
        }
    }
}

}  // namespace gm3d

// K-stages in flight ahead of the multiplied one, per tile height (64, 128); gm3d_gemm_ring_set_depth is the measurement knob of
// tools/gemm_kbench.py (values outside the supported range are clamped)
static int RING_DEPTH[2] = {2, 2};

extern "C" int gm3d_gemm_ring_set_depth(int bm, int depth) {
    if (bm != 64 && bm != 128) return GM3D_EINVAL;
    const int hi = bm == 64 ? 4 : 3;
    RING_DEPTH[bm == 64 ? 0 : 1] = depth < 2 ? 2 : (depth > hi ? hi : depth);
    return GM3D_OK;
}

static int ring_launch(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int lda, int ldw, int ldc, int bm,
                       gm3d_stream_t stream, const float* res, const float* rowscale, int rows_per_sample, const void* add, float* U,
                       float* stats, void* U16) {
    using namespace gm3d;
    if (!A || !W || (!C && !U) || M < 0 || N < 1 || K < 1) return GM3D_EINVAL;
    // N % 128 != 0 (the 96-wide reconstruction head, P/models_mae_learn_loss.py:169-176): the last column tile is ragged -- W rows
    // past N are clamped, columns past N never stored; the residual epilogue needs whole tiles
    if (N % 8 || (U && N % RBN) || K % RBK || lda % 8 || ldw % 8 || lda < K || ldw < K || (C && (ldc % 8 || ldc < N)))
        return GM3D_EUNSUPPORTED;
    if ((((size_t)A | (size_t)W) & 15) || (bm != 64 && bm != 128)) return GM3D_EUNSUPPORTED;
    if (M == 0) return GM3D_OK;
    const int tiles_m = (M + bm - 1) / bm, tiles_n = (N + RBN - 1) / RBN;
    if ((long long)tiles_m * tiles_n > 0x7ffffff0LL) return GM3D_EUNSUPPORTED;
    const int total = tiles_m * tiles_n, grid = (total + 7) / 8 * 8;
    const int depth = U ? 2 : RING_DEPTH[bm == 64 ? 0 : 1];
    const size_t lds = (size_t)(depth + 2) * (bm + RBN) * 128;
#define GM3D_RING_LAUNCH(WMI, RES, DEPTH)                                                                                 \
    {                                                                                                                    \
        static LdsAttr attr;                                                                                             \
        if (!attr.ensure((const void*)gemm_tn_ring_kernel<WMI, RES, DEPTH>, lds)) return GM3D_ELAUNCH;                   \
        hipLaunchKernelGGL((gemm_tn_ring_kernel<WMI, RES, DEPTH>), dim3(grid), dim3(256), lds, (hipStream_t)stream,      \
                           (const bf16_t*)A, (const bf16_t*)W, bias, (bf16_t*)C, M, N, K, lda, ldw, ldc, tiles_n, total, \
                           res, rowscale, rows_per_sample, (const bf16_t*)add, U, stats, (bf16_t*)U16);                  \
    }
    if (U) {
        if (bm == 64) GM3D_RING_LAUNCH(1, true, 2) else GM3D_RING_LAUNCH(2, true, 2)
    } else if (bm == 64) {
        if (depth == 2) GM3D_RING_LAUNCH(1, false, 2) else if (depth == 3) GM3D_RING_LAUNCH(1, false, 3) else GM3D_RING_LAUNCH(1, false, 4)
    } else {
        if (depth == 2) GM3D_RING_LAUNCH(2, false, 2) else GM3D_RING_LAUNCH(2, false, 3)
    }
#undef GM3D_RING_LAUNCH
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_gemm_tn_bf16_ring(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int lda, int ldw,
                                      int ldc, int bm, gm3d_stream_t stream) {
    if (!C) return GM3D_EINVAL;
    return ring_launch(A, W, bias, C, M, N, K, lda, ldw, ldc, bm, stream, nullptr, nullptr, 1, nullptr, nullptr, nullptr, nullptr);
}

extern "C" int gm3d_gemm_tn_bf16_ring96(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int lda, int ldw,
                                        int ldc, int bm, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!A || !W || !C || M < 0 || N < 1 || K < 1) return GM3D_EINVAL;
    if (N % R96 || K % RBK || lda % 8 || ldw % 8 || lda < K || ldw < K || ldc % 8 || ldc < N) return GM3D_EUNSUPPORTED;
    if ((((size_t)A | (size_t)W | (size_t)C) & 15) || (bm != 64 && bm != 128)) return GM3D_EUNSUPPORTED;
    if (M == 0) return GM3D_OK;
    const int tiles_m = (M + bm - 1) / bm, tiles_n = N / R96;
    if ((long long)tiles_m * tiles_n > 0x7ffffff0LL) return GM3D_EUNSUPPORTED;
    const int total = tiles_m * tiles_n, grid = (total + 7) / 8 * 8;
    const size_t lds = (size_t)RNBUF * (bm + R96) * 128;
    if (bm == 64) {
        static LdsAttr attr;
        if (!attr.ensure((const void*)gemm_tn_ring96_kernel<1>, lds)) return GM3D_ELAUNCH;
        hipLaunchKernelGGL((gemm_tn_ring96_kernel<1>), dim3(grid), dim3(192), lds, (hipStream_t)stream, (const bf16_t*)A, (const bf16_t*)W,
                           bias, (bf16_t*)C, M, N, K, lda, ldw, ldc, tiles_n, total);
    } else {
        static LdsAttr attr;
        if (!attr.ensure((const void*)gemm_tn_ring96_kernel<2>, lds)) return GM3D_ELAUNCH;
        hipLaunchKernelGGL((gemm_tn_ring96_kernel<2>), dim3(grid), dim3(192), lds, (hipStream_t)stream, (const bf16_t*)A, (const bf16_t*)W,
                           bias, (bf16_t*)C, M, N, K, lda, ldw, ldc, tiles_n, total);
    }
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_gemm_tn_bf16_res(const void* A, const void* W, const float* bias, const float* res, const float* rowscale,
                                     int rows_per_sample, const void* add, float* U, void* U16, float* stats, int M, int N, int K,
                                     int lda, int ldw, int bm, gm3d_stream_t stream) {
    if (!res || !U || !U16 || !stats || N != 384 || (rowscale && rows_per_sample < 1)) return GM3D_EINVAL;
    return ring_launch(A, W, bias, nullptr, M, N, K, lda, ldw, 0, bm, stream, res, rowscale, rows_per_sample, add, U, stats, U16);
}
