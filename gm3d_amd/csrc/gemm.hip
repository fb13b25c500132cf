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

namespace gm3d {

typedef __bf16 gbf16x8 __attribute__((ext_vector_type(8)));
typedef float gf32x16 __attribute__((ext_vector_type(16)));

constexpr int GBM = 128, GBN = 128, GBK = 64;
constexpr int GPITCH = 72;                     // bf16 elements per LDS row (64 + 8 pad: conflict-free 16-byte operand reads)
constexpr int GCP = 132;                       // floats per row of the fp32 epilogue tile
constexpr int GSTAGE = (GBM + GBN) * GPITCH;   // bf16 elements per stage

__global__ __launch_bounds__(256, 2) void gemm_tn_bf16_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                              const float* __restrict__ bias, bf16_t* __restrict__ C, int M,
                                                              int N, int K, int lda, int ldw, int ldc, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    bf16_t* sm = reinterpret_cast<bf16_t*>(gsm);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    // consecutive workgroups share the A rows (same tile_m, all tile_n) so the activation tile is read once from HBM
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
    const int m0 = tile_m * GBM, n0 = tile_n * GBN;
    const int wm = (w >> 1) * 64, wn = (w & 1) * 64;

    // global -> register prefetch: 4 chunks of A and 4 of W per thread per stage (16 bytes each)
    gbf16x8 pa[4], pw[4];
    const int crow = tid >> 3, ckc = (tid & 7) * 8;           // chunk c = tid + 256*i -> row = crow + 32*i, k offset ckc
    auto load_stage = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = crow + 32 * i;
            const int am = m0 + row;
            pa[i] = am < M ? *reinterpret_cast<const gbf16x8*>(A + (size_t)am * lda + k0 + ckc) : gbf16x8{};
            pw[i] = *reinterpret_cast<const gbf16x8*>(W + (size_t)(n0 + row) * ldw + k0 + ckc);
        }
    };
    auto store_stage = [&](int st) {
        bf16_t* as = sm + st * GSTAGE;
        bf16_t* ws = as + GBM * GPITCH;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = crow + 32 * i;
            *reinterpret_cast<gbf16x8*>(as + row * GPITCH + ckc) = pa[i];
            *reinterpret_cast<gbf16x8*>(ws + row * GPITCH + ckc) = pw[i];
        }
    };

    gf32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

    const int KT = K / GBK;
    load_stage(0);
    store_stage(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) load_stage((kt + 1) * GBK);
        const bf16_t* as = sm + (kt & 1) * GSTAGE;
        const bf16_t* ws = as + GBM * GPITCH;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int ko = 16 * s + 8 * hh;
            gbf16x8 fa[2], fw[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const gbf16x8*>(as + (wm + 32 * i + r) * GPITCH + ko);
#pragma unroll
            for (int j = 0; j < 2; ++j) fw[j] = *reinterpret_cast<const gbf16x8*>(ws + (wn + 32 * j + r) * GPITCH + ko);
            // transposed product: rows (registers) = n, columns (lanes) = m
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[j], fa[i], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < KT) store_stage((kt + 1) & 1);
        __syncthreads();
    }

    // epilogue: acc -> fp32 tile in LDS (row = m, 4 consecutive n per register quad) -> + bias -> bf16, 16-byte stores
    float* cs = reinterpret_cast<float*>(gsm);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = wm + 32 * i + r, n = wn + 32 * j + 8 * q + 4 * hh;
                *reinterpret_cast<float4*>(cs + m * GCP + n) =
                    make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
            }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = tid + 256 * i;                 // 128 rows x 16 chunks of 8 columns
        const int row = c >> 4, nc = (c & 15) * 8;
        if (m0 + row < M) {
            float v[8];
            const float4 x = *reinterpret_cast<const float4*>(cs + row * GCP + nc), y = *reinterpret_cast<const float4*>(cs + row * GCP + nc + 4);
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

extern "C" int gm3d_gemm_tn_bf16(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int lda,
                                 int ldw, int ldc, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!A || !W || !C || M < 0 || N < 1 || K < 1) return GM3D_EINVAL;
    if (N % GBN || K % GBK || lda % 8 || ldw % 8 || ldc % 8 || lda < K || ldw < K || ldc < N) return GM3D_EUNSUPPORTED;
    if (M == 0) return GM3D_OK;
    const int tiles_m = (M + GBM - 1) / GBM, tiles_n = N / GBN;
    if ((long long)tiles_m * tiles_n > 0x7fffffffLL) return GM3D_EUNSUPPORTED;
    const size_t lds_ab = (size_t)2 * GSTAGE * sizeof(bf16_t), lds_c = (size_t)GBM * GCP * sizeof(float);
    const size_t lds = lds_ab > lds_c ? lds_ab : lds_c;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)gemm_tn_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return GM3D_ELAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL(gemm_tn_bf16_kernel, dim3(tiles_m * tiles_n), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)A,
                       (const bf16_t*)W, bias, (bf16_t*)C, M, N, K, lda, ldw, ldc, tiles_n);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}
