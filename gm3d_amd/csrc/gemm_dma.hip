// C[M,N] = A[M,K] . W[N,K]^T (+ bias[N]), bf16 in / fp32 accumulate / bf16 out -- the SHORT-K, MANY-TILE members of the path's GEMM
// family: qkv (384 -> 1152), fc1 (384 -> 1536, with the GELU epilogue), proj and its input gradient (384 -> 384), the fc2 input
// gradient (384 -> 1536, GELU-backward epilogue) at 3200 .. 8192 rows.
//
// Beneath: the same nn.Linear layers as csrc/gemm.hip (timm Block attn.qkv / attn.proj / mlp.fc1 forward, attn.proj / mlp.fc2
// input gradients; in-tree twin Point-MAE_SA3D/models/Point_MAE.py:82-125).
//
// Why a third kernel (MI355X).  With K = 384 a tile has six K-stages: prologue, epilogue and the fill of the load pipeline are as
// long as the steady state, so what counts is (a) bytes per flop a workgroup pulls from L2 and (b) how many workgroups a CU
// overlaps.  This kernel takes 64 (or 128) x 192 tiles -- 192 divides all three widths, 25 % fewer operand bytes per flop than
// 64 x 128 -- in stages of 32 (40) KiB filled by LDS-DMA (global_load_lds_dwordx4: no staging registers, no LDS store
// instructions), double-buffered, ONE stage in flight per workgroup and two workgroups per CU (64 / 80 KiB of LDS each): while
// one multiplies, the other's loads and epilogue stores run.  Measured inside the fused qkv + attention kernel
// (csrc/attention.hip) this shape ran the 4096 x 1152 x 384 product in about 6.5 us against 11.3 (library) / 12.4 (gemm.hip).
// Accumulation order is that of gemm.hip / gemm_ring.hip (k ascending in steps of 16 inside 32x32x16 MFMAs): identical bits.
//
// Round 3: the tile WIDTH is a template parameter (NJ 32-column MFMA tiles per wave: 128 / 192 / 256 columns).  The 256-wide form
// serves the mini-PointNet convolutions over the 262,144 point rows (Encoder.second_conv, P/models_mae_learn_loss.py:878-883, and
// their input gradients: N = 512 / 256), which are HBM-bound products (A read once + C written once = 400 MB at N = 512): a
// 128 x 256 tile moves 1 byte from L2 per 85 flops (128 x 128: 64), so the L2 -> CU stream stays below the HBM stream.
#include "common.hpp"

namespace gm3d {

typedef __bf16 dbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 dbf16x4 __attribute__((ext_vector_type(4)));
typedef float df32x16 __attribute__((ext_vector_type(16)));

constexpr int DBK = 64;

__device__ __forceinline__ int dma_f(int row) { return (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1); }
__device__ __forceinline__ int dma_off(int row, int ch) { return row * 128 + ((ch ^ dma_f(row)) << 4); }

__device__ __forceinline__ void dma_glds16(const void* gsrc, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
}

// WMI: 32-row MFMA tiles per wave in M (1: 64-row workgroup tile, 2: 128-row)
// EPI: 0  C = A.W^T (+ bias)
//      1  fc1:  C (optional) = bf16(A.W^T) -- the pre-activation WITHOUT bias, what the GELU backward re-reads --,
//               G = GELU(bf16(A.W^T) + bias)                                   (as gm3d_gemm_tn_bf16_gelu)
//      2  fc2 input gradient: C = bf16(A.W^T) * GELU'(Fpre + bias), colpart[tile_m][n] = the tile's column sums of the fp32
//               products (the fc1 bias gradient, finished later)                  (as gm3d_gemm_tn_bf16_gelu_bwd)
//      3  mini-PointNet Conv1d(k=1) + max over each group's 32 rows (+ argmax): G (= P) [group][n] = max_k bf16(A.W^T (+ bias)),
//               ARG the winning row (first maximum), bias before or after the pool; C (optional) = the rows
//               (as gm3d_gemm_tn_bf16_pool: same roundings, same decisions)
template <int WMI, int EPI, int NJ>
__global__ __launch_bounds__(256, 2) void gemm_tn_dma_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                             const float* __restrict__ bias, bf16_t* __restrict__ C, int M, int N, int K,
                                                             int lda, int ldw, int ldc, int tiles_n, int total_tiles,
                                                             bf16_t* __restrict__ G, int ldg, const bf16_t* __restrict__ Fpre,
                                                             int ldfp, float* __restrict__ colpart, uint8_t* __restrict__ ARG,
                                                             int bias_after_pool) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    constexpr int BM = 64 * WMI, DBN = 64 * NJ;
    constexpr int STAGE = (BM + DBN) * 128;                 // bytes
    constexpr int PA = BM / 8, PIECES = (PA + DBN / 8) / 4; // 1-KiB pieces of the A tile; pieces per wave per stage (6 .. 12)
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int per_xcd = gridDim.x >> 3;
    const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (logical >= total_tiles) return;
    const int tile_m = logical / tiles_n, tile_n = logical - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * DBN;
    const int wm = (w >> 1) * 32 * WMI, wn = (w & 1) * 32 * NJ;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)dsm;

    const int prow = lane >> 3, pslot = lane & 7;
    const bf16_t* src[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int p = w + 4 * i;
        if (p < PA) {
            const int row = 8 * p + prow;
            const int am = m0 + row < M ? m0 + row : M - 1;          // rows past M: clamped (their outputs are never stored)
            src[i] = A + (size_t)am * lda + ((pslot ^ dma_f(row)) << 3);
        } else {
            const int row = 8 * (p - PA) + prow;
            src[i] = W + (size_t)(n0 + row) * ldw + ((pslot ^ dma_f(row)) << 3);
        }
    }
#define GM3D_DMA_STAGE(ST)                                                                          \
    {                                                                                               \
        const unsigned base = lds0 + ((ST) & 1) * STAGE;                                            \
        _Pragma("unroll") for (int i = 0; i < PIECES; ++i)                                          \
            dma_glds16(src[i] + (size_t)(ST) * DBK, base + 1024 * (w + 4 * i));                     \
    }
    df32x16 acc[WMI][NJ];
#pragma unroll
    for (int i = 0; i < WMI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

    const int KT = K / DBK;
    GM3D_DMA_STAGE(0)
    for (int kt = 0; kt < KT; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's pieces of stage kt have landed ...
        __builtin_amdgcn_s_barrier();                             // ... everybody's have, and nobody still reads the other buffer
        if (kt + 1 < KT) GM3D_DMA_STAGE(kt + 1)
        const unsigned char* as = dsm + (kt & 1) * STAGE;
        const unsigned char* ws = as + BM * 128;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            dbf16x8 fa[WMI];
#pragma unroll
            for (int i = 0; i < WMI; ++i) fa[i] = *reinterpret_cast<const dbf16x8*>(as + dma_off(wm + 32 * i + r, 2 * s + hh));
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const dbf16x8 fw = *reinterpret_cast<const dbf16x8*>(ws + dma_off(wn + 32 * j + r, 2 * s + hh));
#pragma unroll
                for (int i = 0; i < WMI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fa[i], acc[i][j], 0, 0, 0);
            }
        }
    }
#undef GM3D_DMA_STAGE
    __syncthreads();                 // every wave is done with the stages: they become the bf16 staging images of the epilogue
    // acc[i][j][g]: row wm + 32 i + r, column wn + 32 j + crow(g) (4 consecutive per quad: 8 q + 4 hh + e).  Staged as NJ
    // [BM][64] bf16 images (swizzled 128-byte rows, the layout of the stages), read back as whole 16-byte chunks of a row.
#pragma unroll
    for (int i = 0; i < WMI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int row = wm + 32 * i + r, col = wn + 32 * j;
            unsigned char* img = dsm + (col >> 6) * (BM * 128);
            const int cbase = (col & 63) >> 3;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4] = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                if ((EPI == 0 || (EPI == 3 && !bias_after_pool)) && bias) {
                    const float4 b = *reinterpret_cast<const float4*>(bias + n0 + col + 8 * q + 4 * hh);
                    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                }
                dbf16x4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = (bf16_t)v[e];
                *reinterpret_cast<dbf16x4*>(img + dma_off(row, cbase + q) + 8 * hh) = pk;
            }
        }
    __syncthreads();
    if (EPI == 3) {
        // the tile holds BM / 32 whole groups; one thread = 8 columns of one group: 32 16-byte reads down the rows of the image,
        // values compared as the bf16 numbers they are stored as (first maximum wins: the decision of GEMM -> gm3d_group_max_fwd)
        constexpr int CH = DBN / 8;
        if (tid < (BM / 32) * CH) {
            const int g = tid / CH, chunk = tid - g * CH;
            if (m0 + 32 * g < M) {
                const unsigned char* img = dsm + (chunk >> 3) * (BM * 128);
                float best[8];
                int bk[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bk[e] = 0; }
#pragma unroll 8
                for (int k = 0; k < 32; ++k) {
                    const dbf16x8 f = *reinterpret_cast<const dbf16x8*>(img + dma_off(32 * g + k, chunk & 7));
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float v = (float)f[e];
                        if (v > best[e]) { best[e] = v; bk[e] = k; }
                    }
                }
                const int n = n0 + 8 * chunk;
                if (bias_after_pool && bias) {
                    float bv[8];
                    V8<float>::load(bias + n, bv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) best[e] += bv[e];
                }
                const size_t o = (size_t)((m0 + 32 * g) >> 5) * ldg + n;
                V8<bf16_t>::store(G + o, best);
                unsigned long long pk = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) pk |= (unsigned long long)(bk[e] & 0xff) << (8 * e);
                *reinterpret_cast<unsigned long long*>(ARG + o) = pk;
            }
        }
        if (!C) return;
    }
    // thread -> chunk ch = tid & 7 of rows (tid >> 3) + 32 k of image t: i = 2 WMI t + k
    float csum[NJ][8];
#pragma unroll
    for (int t = 0; t < NJ; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[t][e] = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * NJ * WMI; ++i) {                     // NJ images x BM rows x 8 chunks = 2 NJ WMI x 256 threads
        const int c = tid + 256 * i;
        const int t = i / (2 * WMI), row = (c - t * (BM * 8)) >> 3, ch = tid & 7;
        if (m0 + row < M) {
            const uint4 raw = *reinterpret_cast<const uint4*>(dsm + t * (BM * 128) + dma_off(row, ch));
            const int n = n0 + 64 * t + 8 * ch;
            if (EPI == 0 || EPI == 3) {
                *reinterpret_cast<uint4*>(C + (size_t)(m0 + row) * ldc + n) = raw;
            } else {
                const dbf16x8 f = *reinterpret_cast<const dbf16x8*>(&raw);
                float bv[8], v[8];
                V8<float>::load(bias + n, bv);
                if (EPI == 1) {
                    if (C) *reinterpret_cast<uint4*>(C + (size_t)(m0 + row) * ldc + n) = raw;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = gelu_f<bf16_t>((float)f[e] + bv[e]);
                    V8<bf16_t>::store(G + (size_t)(m0 + row) * ldg + n, v);
                } else {
                    float fv[8];
                    V8<bf16_t>::load(Fpre + (size_t)(m0 + row) * ldfp + n, fv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v[e] = (float)f[e] * gelu_grad_f<bf16_t>(fv[e] + bv[e]);
                        csum[t][e] += v[e];
                    }
                    V8<bf16_t>::store(C + (size_t)(m0 + row) * ldc + n, v);
                }
            }
        }
    }
    if (EPI == 2) {      // the 32 threads that share a column chunk meet in LDS (the images are no longer needed)
        __syncthreads();
        float* cs = reinterpret_cast<float*>(dsm);
#pragma unroll
        for (int t = 0; t < NJ; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) cs[(tid >> 3) * DBN + 64 * t + 8 * (tid & 7) + e] = csum[t][e];
        __syncthreads();
        if (tid < DBN) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k) s += cs[k * DBN + tid];
            colpart[(size_t)tile_m * N + n0 + tid] = s;
        }
    }
}

}  // namespace gm3d

static int dma_launch(const void* A, const void* W, const float* bias, void* C, void* G, int M, int N, int K, int lda, int ldw, int ldc,
                      int ldg, int bm, gm3d_stream_t stream, const void* Fpre = nullptr, int ldfp = 0, float* colpart = nullptr,
                      int bn = 192, uint8_t* ARG = nullptr, int bias_after_pool = 0) {
    using namespace gm3d;
    if (!A || !W || (!C && !G) || M < 0 || N < 1 || K < 1) return GM3D_EINVAL;
    if ((G || Fpre) && !bias && !ARG) return GM3D_EINVAL;
    if (ARG && (!G || Fpre || M % 32 || ldg % 8 || ((size_t)ARG & 7) || (bn != 128 && bn != 192))) return GM3D_EINVAL;
    if (Fpre && (!C || !colpart || G || ldfp % 8 || ldfp < N || ((size_t)Fpre & 15))) return GM3D_EINVAL;
    if (bn != 128 && bn != 192 && bn != 256) return GM3D_EUNSUPPORTED;
    if ((G || Fpre) && !ARG && bn != 192) return GM3D_EUNSUPPORTED;  // the GELU epilogues exist for the 192-column tile only
    if (N % bn || K % DBK || lda % 8 || ldw % 8 || lda < K || ldw < K || (C && (ldc % 8 || ldc < N)) || (G && (ldg % 8 || ldg < N)))
        return GM3D_EUNSUPPORTED;
    if ((((size_t)A | (size_t)W | (size_t)C | (size_t)G) & 15) || (bm != 64 && bm != 128)) return GM3D_EUNSUPPORTED;
    if (M == 0) return GM3D_OK;
    const int tiles_m = (M + bm - 1) / bm, tiles_n = N / bn;
    if ((long long)tiles_m * tiles_n > 0x7ffffff0LL) return GM3D_EUNSUPPORTED;
    const int total = tiles_m * tiles_n, grid = (total + 7) / 8 * 8;
    const size_t lds = (size_t)2 * (bm + bn) * 128;
#define GM3D_DMA_LAUNCH(WMI, EPI, NJ)                                                                                    \
    {                                                                                                                    \
        static LdsAttr attr;                                                                                             \
        if (!attr.ensure((const void*)gemm_tn_dma_kernel<WMI, EPI, NJ>, lds)) return GM3D_ELAUNCH;                       \
        hipLaunchKernelGGL((gemm_tn_dma_kernel<WMI, EPI, NJ>), dim3(grid), dim3(256), lds, (hipStream_t)stream,          \
                           (const bf16_t*)A, (const bf16_t*)W, bias, (bf16_t*)C, M, N, K, lda, ldw, ldc, tiles_n, total, \
                           (bf16_t*)G, ldg, (const bf16_t*)Fpre, ldfp, colpart, ARG, bias_after_pool);                   \
    }
    if (ARG) {
        if (bn == 128) {
            if (bm == 64) GM3D_DMA_LAUNCH(1, 3, 2) else GM3D_DMA_LAUNCH(2, 3, 2)
        } else {
            if (bm == 64) GM3D_DMA_LAUNCH(1, 3, 3) else GM3D_DMA_LAUNCH(2, 3, 3)
        }
    } else if (Fpre) {
        if (bm == 64) GM3D_DMA_LAUNCH(1, 2, 3) else GM3D_DMA_LAUNCH(2, 2, 3)
    } else if (G) {
        if (bm == 64) GM3D_DMA_LAUNCH(1, 1, 3) else GM3D_DMA_LAUNCH(2, 1, 3)
    } else if (bn == 192) {
        if (bm == 64) GM3D_DMA_LAUNCH(1, 0, 3) else GM3D_DMA_LAUNCH(2, 0, 3)
    } else if (bn == 128) {
        if (bm == 64) GM3D_DMA_LAUNCH(1, 0, 2) else GM3D_DMA_LAUNCH(2, 0, 2)
    } else {
        if (bm == 64) GM3D_DMA_LAUNCH(1, 0, 4) else GM3D_DMA_LAUNCH(2, 0, 4)
    }
#undef GM3D_DMA_LAUNCH
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_gemm_tn_bf16_dma(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int lda, int ldw,
                                     int ldc, int bm, gm3d_stream_t stream) {
    if (!C) return GM3D_EINVAL;
    return dma_launch(A, W, bias, C, nullptr, M, N, K, lda, ldw, ldc, 0, bm, stream);
}

extern "C" int gm3d_gemm_tn_bf16_dmaw(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int lda, int ldw,
                                      int ldc, int bm, int bn, gm3d_stream_t stream) {
    if (!C) return GM3D_EINVAL;
    return dma_launch(A, W, bias, C, nullptr, M, N, K, lda, ldw, ldc, 0, bm, stream, nullptr, 0, nullptr, bn);
}

extern "C" int gm3d_gemm_tn_bf16_dma_pool(const void* A, const void* W, const float* bias, void* C, void* P, uint8_t* arg, int M, int N,
                                          int K, int lda, int ldw, int ldc, int ldp, int bias_after_pool, int bm, int bn,
                                          gm3d_stream_t stream) {
    if (!P || !arg) return GM3D_EINVAL;
    return dma_launch(A, W, bias, C, P, M, N, K, lda, ldw, ldc, ldp, bm, stream, nullptr, 0, nullptr, bn, arg, bias_after_pool);
}

extern "C" int gm3d_gemm_tn_bf16_dma_gelu(const void* A, const void* W, const float* bias, void* F, void* G, int M, int N, int K,
                                          int lda, int ldw, int ldf, int ldg, int bm, gm3d_stream_t stream) {
    if (!G || !bias) return GM3D_EINVAL;
    return dma_launch(A, W, bias, F, G, M, N, K, lda, ldw, ldf, ldg, bm, stream);
}

extern "C" int gm3d_gemm_tn_bf16_dma_gelu_bwd(const void* dO, const void* Wt, const void* F, const float* bias, void* dF, float* colpart,
                                              int M, int N, int K, int lda, int ldw, int ldf, int lddf, int bm, gm3d_stream_t stream) {
    if (!F || !bias || !dF || !colpart) return GM3D_EINVAL;
    return dma_launch(dO, Wt, bias, dF, nullptr, M, N, K, lda, ldw, lddf, 0, bm, stream, F, ldf, colpart);
}
