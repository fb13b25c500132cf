// C[M,N] = A[M,K] . W[N,K]^T (+ bias[N]), bf16 in / fp32 accumulate / bf16 out -- the TALL members of the path's GEMM family: the
// mini-PointNet convolutions over the B*G*k = 262,144 point rows (102,400 for the student's visible groups) and their input
// gradients, K <= 512, N <= 512.
//
// Beneath: Encoder.first_conv.3 / second_conv.0 / second_conv.3 (Conv1d(k=1): Point-MAE_SA3D/models_mae_learn_loss.py:876-882) with the
// max over each group's 32 points (:893,897), and the input gradients autograd derives for them.
//
// Why a kernel of their own (MI355X).  These products are HBM streams: A is read once and C written once (400 MB at 256 -> 512),
// the weights are 64 .. 384 KiB.  The tiled kernels re-stage a W tile with every output tile, so what each CU pulls from L2 is
// dominated by W (1.07 GB for 400 MB of HBM traffic at 128 x 128 tiles), and with four K-stages per tile the load latency is
// exposed tile after tile: 2.8 TB/s (profiles/r03_gemm_kbench.txt: 142 us; the tuned library 115 us).  Here the weights are
// STATIONARY IN REGISTERS: a workgroup is persistent, each of its NW compute waves keeps the fragments of 32 output columns for
// the whole K (K/4 VGPRs), and only A moves -- 32-row tiles ([32][K] bf16 = one group of 32 points) through an LDS ring filled
// by LDS-DMA from dedicated loader waves (their vmcnt counts nothing but those loads, so the ring runs on counted waits while the
// compute waves' stores drain at their own pace; DEPTH tiles in flight).  Per tile every compute wave multiplies the shared A
// tile with its resident W fragments (K/16 MFMA 32x32x16, k ascending: the accumulation order of csrc/gemm.hip -> identical
// bits), the 32 x BN result is staged in LDS as bf16 and leaves as whole rows; the max-pool epilogue reduces the staged tile --
// exactly one group -- down its 32 rows.  L2 -> CU traffic: A once per column block of 32 NW columns, nothing else.
#include "common.hpp"

namespace gm3d {

typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wbf16x4 __attribute__((ext_vector_type(4)));
typedef float wf32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int ws_f(int row) { return (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1); }
__device__ __forceinline__ int ws_off(int row, int ch) { return row * 128 + ((ch ^ ws_f(row)) << 4); }

__device__ __forceinline__ void ws_glds16(const void* gsrc, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n <= 48 (the immediate must be a constant: one uniform branch)
__device__ __forceinline__ void ws_wait_vm(int n) {
#define GM3D_WS_CASE(N_) case N_: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); break;
    switch (n) {
        GM3D_WS_CASE(0) GM3D_WS_CASE(4) GM3D_WS_CASE(8) GM3D_WS_CASE(12) GM3D_WS_CASE(16) GM3D_WS_CASE(24) GM3D_WS_CASE(32)
        GM3D_WS_CASE(36) GM3D_WS_CASE(48)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef GM3D_WS_CASE
}

// KT = K / 64; NW compute waves (32 output columns each: BN = 32 NW); NL loader waves; TM 32-row sub-tiles per ring slot (two: each
// wave carries two independent accumulator chains -- the K / 16 MFMAs of ONE chain are dependent, ~64 cycles each -- and the two
// barriers of an iteration are shared by 64 rows); EPI 0: C = A.W^T (+ bias); EPI 3: max over each sub-tile's 32 rows (+ argmax,
// bias before or after the pool) into P / ARG, C rows optional (as gm3d_gemm_tn_bf16_pool).
// EPI 4 / 5 (the second BatchNorm of the mini-PointNet around second_conv.0, 32-row tiles = one group each; T (M / 32, N) is the per-group
// term of the split concat product): 4 = eval-mode BatchNorm + ReLU applied to the bf16-rounded product in the epilogue,
// C = act((bf16(A.W^T) + T[group]) * scale + shift) -- the arithmetic of bn_bcast_apply_relu_kernel on the stored product, bit for bit;
// 5 = train mode: C = bf16(A.W^T) as EPI 0, plus this workgroup's column sums of y = C + T[group] and y^2 (bn_bcast_stats_kernel's
// statistics) -> partial[first][0 | 1][N], one row per workgroup of a column block, accumulated in a fixed order.
struct WsBn {
    const bf16_t* T;
    const float* scale;
    const float* shift;
    float* partial;
    int ldt;
    float slope;
};

constexpr int ws_depth(int ppl, int tile, int stage) {
    int d = 48 / ppl < 4 ? 48 / ppl : 4;                // vmcnt is a 6-bit counter
    while (d > 1 && (d + 1) * tile + stage > 152 * 1024) --d;
    return d;
}

template <int KT, int NW, int NL, int EPI, int TM>
__global__ __launch_bounds__(64 * (NW + NL)) void gemm_tn_ws_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                                     const float* __restrict__ bias, bf16_t* __restrict__ C, int M, int N,
                                                                     int lda, int ldw, int ldc, int tiles_n, bf16_t* __restrict__ P,
                                                                     uint8_t* __restrict__ ARG, int ldp, int bias_after_pool, WsBn bn,
                                                                     int Kreal, int pool16) {
    // pool16 (EPI 3): groups of 16 rows -- a 32-row tile holds two groups, rows 0..15 and 16..31 (the hierarchical model's level-0
    // groups of 16 points); P / ARG then have M / 16 rows and ARG counts within its group.
    // Kreal <= 64 KT: the true K (a multiple of 8) when it is not a multiple of the 64-column images (the hierarchical model's 96- and
    // 288-wide products).  Columns past it are never read: the loader's pieces there re-read column 0 of the same row (finite wherever
    // the row is), and the resident W fragments are zero from Kreal on, so those k contribute exactly 0.
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
    constexpr int BN = 32 * NW, NIMG = BN / 64 > 0 ? (BN + 63) / 64 : 1;
    constexpr int HALF = KT * 4096;                 // bytes of one [32][K] sub-tile: KT swizzled [32][64] images
    constexpr int TILE = TM * HALF;
    constexpr int NP = TM * 4 * KT;                 // 1-KiB LDS-DMA pieces per slot
    constexpr int PPL = NP / NL;                    // pieces per loader wave and slot
    constexpr int STAGE = TM * NIMG * 4096;         // [32 TM][BN] bf16 as swizzled [32][64] images
    constexpr int DEPTH = ws_depth(PPL, TILE, STAGE);          // slots in flight
    constexpr int NT = DEPTH + 1;                   // ring slots
    static_assert(NP % NL == 0 && PPL <= 48, "loader split");
    unsigned char* stage = wsm + NT * TILE;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int per_xcd = gridDim.x >> 3;
    const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);      // column blocks of one row stream share an XCD's L2
    const int tile_n = logical % tiles_n, first = logical / tiles_n, stride = gridDim.x / tiles_n;
    const int tiles_m = (M + 32 * TM - 1) / (32 * TM);
    const int nmine = first < tiles_m ? (tiles_m - first + stride - 1) / stride : 0;
    const int n0 = tile_n * BN;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)wsm;

    if (w >= NW) {
        // ------------------------------------------------------------------ loader wave(s): nothing but LDS-DMA and barriers
        const int lw = w - NW, prow = lane >> 3, pslot = lane & 7;
        auto issue = [&](int i) {
            const int m0 = (first + i * stride) * (32 * TM);
            const unsigned base = lds0 + (i % NT) * TILE;
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                const int p = lw * PPL + q, t = p / (4 * KT), pp = p - t * (4 * KT), kt = pp >> 2, row = 8 * (pp & 3) + prow;
                const int gr = m0 + 32 * t + row;
                const int am = gr < M ? gr : M - 1;                  // rows past M: clamped (their outputs are never stored)
                int col = 64 * kt + ((pslot ^ ws_f(row)) << 3);
                col = col < Kreal ? col : 0;
                ws_glds16(A + (size_t)am * lda + col, base + 1024 * p);
            }
        };
        for (int d = 0; d < DEPTH && d < nmine; ++d) issue(d);
        for (int i = 0; i < nmine; ++i) {
            // the slot of tile i + DEPTH is the one tile i - 1 was read from: every compute wave finished with it before the
            // second barrier of iteration i - 1, which this wave has passed
            if (i + DEPTH < nmine) issue(i + DEPTH);
            const int ahead = nmine - 1 - i < DEPTH ? nmine - 1 - i : DEPTH;
            ws_wait_vm(ahead * PPL);
            __builtin_amdgcn_s_barrier();            // B1: tile i has landed (all loader waves) -> visible to the compute waves
            __builtin_amdgcn_s_barrier();            // B2: the compute waves are done READING tile i (and have staged its rows)
        }
        return;
    }

    // ---------------------------------------------------------------------- compute waves: W fragments resident in registers
    wbf16x8 wreg[KT][4];
    {
        const bf16_t* wp = W + (size_t)(n0 + 32 * w + r) * ldw + 8 * hh;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (64 * kt + 16 * s + 8 * hh < Kreal) wreg[kt][s] = *reinterpret_cast<const wbf16x8*>(wp + 64 * kt + 16 * s);
                else
#pragma unroll
                    for (int e = 0; e < 8; ++e) wreg[kt][s][e] = (bf16_t)0.0f;
            }
    }
    float bq[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) bq[q][e] = (EPI != 3 && bias) ? bias[n0 + 32 * w + 8 * q + 4 * hh + e] : 0.f;
    const float bcol = (EPI == 3 && bias) ? bias[n0 + 32 * w + r] : 0.f;
    const bool bias_in_tile = EPI != 3 || !bias_after_pool;
    // EPI 4: this lane's 16 columns of (scale, shift).  EPI 5: their running sums of y and y^2 over this workgroup's tiles.
    float bn_a[4][4], bn_b[4][4];
    if (EPI == 4 || EPI == 5) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = n0 + 32 * w + 8 * q + 4 * hh + e;
                bn_a[q][e] = EPI == 4 ? bn.scale[col] : 0.f;
                bn_b[q][e] = EPI == 4 ? bn.shift[col] : 0.f;
            }
    }
    constexpr int CTHREADS = 64 * NW;

    for (int i = 0; i < nmine; ++i) {
        const int m0 = (first + i * stride) * (32 * TM);
        // EPI 4 / 5: the groups' row of T by SCALAR loads (uniform address, constant address space: lgkmcnt, not vmcnt -- a vector load
        // here would make the wave wait for the previous tile's global stores as well); 32 columns of this wave = 16 dwords
        float tq[TM][4][4];
        if (EPI == 4 || EPI == 5) {
            typedef const __attribute__((address_space(4))) unsigned int* sptr_t;
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                // pool16 (groups of 16 rows: two per tile): rows 0..15 of the tile take T row 2 g, rows 16..31 T row 2 g + 1 -- both rows
                // come by scalar loads, the lane (= tile row r) picks its own
                const size_t trow = pool16 ? (size_t)2 * ((m0 >> 5) + t) : (size_t)((m0 >> 5) + t);
                sptr_t tp = (sptr_t)(const void*)(bn.T + trow * bn.ldt + n0 + 32 * w);
                unsigned tw[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) tw[j] = tp[j];
                if (pool16) {
                    sptr_t tp2 = (sptr_t)(const void*)(bn.T + (trow + 1) * bn.ldt + n0 + 32 * w);
                    unsigned tw2[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) tw2[j] = tp2[j];
                    if (r & 16) {
#pragma unroll
                        for (int j = 0; j < 16; ++j) tw[j] = tw2[j];
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e2 = 0; e2 < 2; ++e2) {
                        const unsigned u = hh ? tw[4 * q + 2 + e2] : tw[4 * q + e2];      // columns 8 q + 4 hh + 2 e2 (+1)
                        tq[t][q][2 * e2] = __builtin_bit_cast(float, u << 16);
                        tq[t][q][2 * e2 + 1] = __builtin_bit_cast(float, u & 0xffff0000u);
                    }
            }
        }
        __builtin_amdgcn_s_barrier();                // B1
        const unsigned char* as = wsm + (i % NT) * TILE;
        wf32x16 acc[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            wbf16x8 fa[TM][4];
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s) fa[t][s] = *reinterpret_cast<const wbf16x8*>(as + t * HALF + kt * 4096 + ws_off(r, 2 * s + hh));
            // EPI 0: the transposed tile (lane = row, registers = 4 x 4 consecutive columns: whole-row staging).  EPI 3: operands
            // swapped -> lane = COLUMN (lane & 31), registers = 16 of the 32 rows (8 (g >> 2) + 4 hh + (g & 3)), the other 16 in lane ^ 32:
            // the max over rows is 15 register maxima + one cross-lane step.  Same products, same k order: same bits.
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < TM; ++t)
                    acc[t] = EPI == 3 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t][s], wreg[kt][s], acc[t], 0, 0, 0)
                                      : __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[kt][s], fa[t][s], acc[t], 0, 0, 0);
        }
        if (EPI == 3 && !C) __builtin_amdgcn_s_barrier();      // B2 (no rows to stage): tile i's slot may be refilled from here on
        if (EPI == 3) {
            // acc[t][g]: row 32 t + 8 (g >> 2) + 4 hh + (g & 3), column 32 w + r.  Rounded to bf16 (+ bias where it belongs before the
            // rounding); ONE 32-bit key per element -- (order-preserving image of the bf16 value) << 16 | (31 - row) -- so that a plain
            // unsigned max picks the largest value and, among equal values, the lowest row: the "first maximum wins" of
            // gm3d_group_max_fwd (-0 counts as +0).
            unsigned short rowbits[TM][16];
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                unsigned best = 0, best2 = 0;           // best2: rows 16..31 when the tile holds two 16-row groups (g >= 8 <=> row >= 16)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const bf16_t o = (bf16_t)(acc[t][g] + (bias_in_tile ? bcol : 0.f));
                    rowbits[t][g] = __builtin_bit_cast(unsigned short, o);
                    unsigned u = (unsigned)rowbits[t][g] << 16;
                    u = u == 0x80000000u ? 0u : u;
                    const unsigned k = (u ^ (unsigned)(((int)u >> 31) | (int)0x80000000)) & 0xffff0000u;
                    const unsigned key = k | (unsigned)(31 - (8 * (g >> 2) + 4 * hh + (g & 3)));
                    if (pool16 && g >= 8) best2 = key > best2 ? key : best2;
                    else best = key > best ? key : best;
                }
                {   // the other rows of this column sit in lane ^ 32
                    const unsigned other = (unsigned)__shfl_xor((int)best, 32);
                    best = other > best ? other : best;
                    const unsigned other2 = (unsigned)__shfl_xor((int)best2, 32);
                    best2 = other2 > best2 ? other2 : best2;
                }
                if (hh == 0 && m0 + 32 * t < M) {
                    const unsigned k = best & 0xffff0000u;
                    const unsigned u = (k & 0x80000000u) ? (k & 0x7fff0000u) : (~k & 0xffff0000u);
                    float v = __builtin_bit_cast(float, u);
                    if (bias_after_pool) v += bcol;
                    if (!pool16) {
                        const size_t o = (size_t)((m0 >> 5) + t) * ldp + n0 + 32 * w + r;
                        P[o] = (bf16_t)v;
                        ARG[o] = (uint8_t)(31u - (best & 0xffu));
                    } else {
                        const size_t o = (size_t)((m0 >> 4) + 2 * t) * ldp + n0 + 32 * w + r;
                        P[o] = (bf16_t)v;
                        ARG[o] = (uint8_t)(31u - (best & 0xffu));                 // rows 0..15
                        if (m0 + 32 * t + 16 < M) {
                            const unsigned k2 = best2 & 0xffff0000u;
                            const unsigned u2 = (k2 & 0x80000000u) ? (k2 & 0x7fff0000u) : (~k2 & 0xffff0000u);
                            float v2 = __builtin_bit_cast(float, u2);
                            if (bias_after_pool) v2 += bcol;
                            P[o + ldp] = (bf16_t)v2;
                            ARG[o + ldp] = (uint8_t)(15u - (best2 & 0xffu));      // 31 - (best2 & 0xff) = row in 16..31 -> minus 16
                        }
                    }
                }
            }
            if (!C) continue;                        // no rows wanted (second_conv.3): nothing is staged
            // rows wanted (first_conv.3): this lane's 16 values of column 32 w + r into the staging images, 2 bytes at a time
            const int ch = (((32 * w) & 63) + r) >> 3, sub = 2 * (r & 7);
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                unsigned char* img = stage + (t * NIMG + ((32 * w) >> 6)) * 4096;
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int row = 8 * (g >> 2) + 4 * hh + (g & 3);
                    *reinterpret_cast<unsigned short*>(img + ws_off(row, ch) + sub) = rowbits[t][g];
                }
            }
        } else {
            // acc[t][4 q + e]: row 32 t + r, column 32 w + 8 q + 4 hh + e, rounded to bf16 (+ bias) into the staging images
            const int cbase = ((32 * w) & 63) >> 3;
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                unsigned char* img = stage + (t * NIMG + ((32 * w) >> 6)) * 4096;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    wbf16x4 pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pk[e] = (bf16_t)(acc[t][4 * q + e] + bq[q][e]);
                        if (EPI == 4) {
                            const float h = ((float)pk[e] + tq[t][q][e]) * bn_a[q][e] + bn_b[q][e];
                            pk[e] = (bf16_t)(h > 0.f ? h : bn.slope * h);
                        }
                        if (EPI == 5) {
                            const float y = (float)pk[e] + tq[t][q][e];
                            bn_a[q][e] += y;
                            bn_b[q][e] += y * y;
                        }
                    }
                    *reinterpret_cast<wbf16x4*>(img + ws_off(r, cbase + q) + 8 * hh) = pk;
                }
            }
        }
        // B2: the tile's rows are staged.  NOT __syncthreads(): its workgroup-scope release also waits for vmcnt(0), i.e. for the global
        // stores of the previous tile's rows.  Only the LDS writes must land.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        {
            // 32 TM rows x BN / 8 chunks of 16 bytes: whole rows leave in 16-byte pieces (BN / 8 lanes per row)
            constexpr int CH = BN / 8;
#pragma unroll
            for (int c0 = 0; c0 < 32 * TM * CH; c0 += CTHREADS) {
                const int c = c0 + tid;
                if (c < 32 * TM * CH) {
                    const int row = c / CH, chunk = c - row * CH;
                    if (m0 + row < M) {
                        const uint4 raw = *reinterpret_cast<const uint4*>(stage + ((row >> 5) * NIMG + (chunk >> 3)) * 4096 + ws_off(row & 31, chunk & 7));
                        *reinterpret_cast<uint4*>(C + (size_t)(m0 + row) * ldc + n0 + 8 * chunk) = raw;
                    }
                }
            }
        }
    }
    if (EPI == 5) {
        // the 32 lanes of a half (same hh: same 16 columns, 32 different rows) fold their sums in a fixed butterfly order
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int m = 1; m < 32; m <<= 1) {
                    bn_a[q][e] += __shfl_xor(bn_a[q][e], m);
                    bn_b[q][e] += __shfl_xor(bn_b[q][e], m);
                }
                if (r == 0) {
                    const int col = n0 + 32 * w + 8 * q + 4 * hh + e;
                    bn.partial[(size_t)first * 2 * N + col] = bn_a[q][e];
                    bn.partial[(size_t)first * 2 * N + N + col] = bn_b[q][e];
                }
            }
    }
}

}  // namespace gm3d

// grid: one persistent workgroup per CU (a multiple of 8 and of tiles_n; never more than there are tiles)
static int WS_WG_PER_CU = 2;      // measurement knob (gm3d_gemm_ws_set_occupancy): persistent workgroups per CU where LDS allows two

extern "C" int gm3d_gemm_ws_set_occupancy(int wg_per_cu) {
    WS_WG_PER_CU = wg_per_cu < 1 ? 1 : (wg_per_cu > 2 ? 2 : wg_per_cu);
    return GM3D_OK;
}

using gm3d::ws_depth;

static int ws_grid(int tiles_m, int tiles_n, size_t lds) {
    long long want = (long long)tiles_m * tiles_n;
    const int cap = (WS_WG_PER_CU == 2 && 2 * lds <= 160 * 1024) ? 512 : 256;
    int g = want < cap ? (int)want : cap;
    g = (g + 7) / 8 * 8;
    while (g % tiles_n) g += 8;
    return g;
}

static int ws_launch(const void* A, const void* W, const float* bias, void* C, void* P, uint8_t* ARG, int M, int N, int K, int lda,
                     int ldw, int ldc, int ldp, int bias_after_pool, gm3d_stream_t stream, int bn_mode = 0, gm3d::WsBn bn = gm3d::WsBn(),
                     int pool16 = 0) {
    using namespace gm3d;
    if (!A || !W || (!C && !P) || M < 0 || N < 1 || K < 1) return GM3D_EINVAL;
    if (P && (!ARG || M % (pool16 ? 16 : 32) || ldp % 8 || ldp < N || ((size_t)ARG & 7) || ((size_t)P & 15))) return GM3D_EINVAL;
    if (K % 8 || K > 512 || lda % 8 || ldw % 8 || lda < K || ldw < K || (C && (ldc % 8 || ldc < N))) return GM3D_EUNSUPPORTED;
    if (((size_t)A | (size_t)W | (size_t)C) & 15) return GM3D_EUNSUPPORTED;
    if (M == 0) return GM3D_OK;
    hipStream_t st = (hipStream_t)stream;
#define GM3D_WS_LAUNCH(KT, NW, NL, EPI, TM)                                                                               \
    {                                                                                                                    \
        constexpr int TILE_ = TM * KT * 4096, STAGE_ = TM * ((32 * NW + 63) / 64) * 4096;                                \
        constexpr int DEPTH_ = ws_depth(TM * 4 * KT / NL, TILE_, STAGE_);                                                \
        const size_t lds = (size_t)(DEPTH_ + 1) * TILE_ + STAGE_;                                                        \
        const int tiles_m = (M + 32 * TM - 1) / (32 * TM);                                                               \
        const int tiles_n = N / (32 * NW), grid = ws_grid(tiles_m, tiles_n, lds);                                        \
        static LdsAttr attr;                                                                                             \
        if (!attr.ensure((const void*)gemm_tn_ws_kernel<KT, NW, NL, EPI, TM>, lds)) return GM3D_ELAUNCH;                 \
        hipLaunchKernelGGL((gemm_tn_ws_kernel<KT, NW, NL, EPI, TM>), dim3(grid), dim3(64 * (NW + NL)), lds, st,          \
                           (const bf16_t*)A, (const bf16_t*)W, bias, (bf16_t*)C, M, N, lda, ldw, ldc, tiles_n, (bf16_t*)P, ARG, ldp, \
                           bias_after_pool, bn, K, pool16);                                                              \
        GM3D_CHECK_LAUNCH();                                                                                             \
        return GM3D_OK;                                                                                                  \
    }
    // the shapes of the mini-PointNet (models_mae_learn_loss.py:872-883) and of its backward; anything else: EUNSUPPORTED
    if (P) {
        // (TM = 2 -- two sub-tiles per slot, two accumulator chains per wave -- measured no faster on any shape: TM = 1 everywhere)
        if (K == 128 && N == 256) GM3D_WS_LAUNCH(2, 8, 1, 3, 1)       // first_conv.3 + max-pool
        if (K == 512 && N == 384) GM3D_WS_LAUNCH(8, 6, 2, 3, 1)       // second_conv.3 + max-pool (two column blocks of 192)
        if (K == 512 && N == 96) GM3D_WS_LAUNCH(8, 3, 2, 3, 1)        // ... of the hierarchical model's level 0 (96-wide tokens)
        return GM3D_EUNSUPPORTED;
    }
    if (bn_mode) {                                                    // second_conv.0 with the BatchNorm that follows it (EPI 4 / 5)
        if (M % 32 || !bn.T || bn.ldt % 4 || bn.ldt < N || ((size_t)bn.T & 7)) return GM3D_EINVAL;      // (pool16: T has M / 16 rows)
        if (K == 256 && N == 512 && bn_mode == 4) GM3D_WS_LAUNCH(4, 8, 1, 4, 1)
        if (K == 256 && N == 512 && bn_mode == 5) GM3D_WS_LAUNCH(4, 8, 1, 5, 1)
        return GM3D_EUNSUPPORTED;
    }
    if (K == 256 && N == 512) GM3D_WS_LAUNCH(4, 8, 1, 0, 1)           // second_conv.0 on the local half (two column blocks)
    if (K == 512 && N == 256) GM3D_WS_LAUNCH(8, 4, 2, 0, 1)           // its input gradient (128 VGPRs of W per wave: 6 waves per CU)
    if (K == 384 && N == 512) GM3D_WS_LAUNCH(6, 8, 2, 0, 1)           // second_conv.3's input gradient
    if (K == 256 && N == 128) GM3D_WS_LAUNCH(4, 4, 1, 0, 1)           // first_conv.3's input gradient
    if (K == 128 && N == 256) GM3D_WS_LAUNCH(2, 8, 1, 0, 1)
    if (K == 512 && N == 384) GM3D_WS_LAUNCH(8, 6, 2, 0, 1)
    // level 0 of the hierarchical encoder (Point-M2AE_SA3D/cfgs/config_Point_M2AE.yaml:57-99: 96-wide blocks over B * 512 = 65,536 token
    // rows): qkv / proj / fc1 / fc2 and their input gradients -- ragged K (96, 288) through the Kreal path
    if (K == 96 && N == 288) GM3D_WS_LAUNCH(2, 9, 1, 0, 1)           // qkv
    if (K == 96 && N == 96) GM3D_WS_LAUNCH(2, 3, 1, 0, 1)            // proj, and its input gradient
    if (K == 96 && N == 512) GM3D_WS_LAUNCH(2, 8, 1, 0, 1)           // level-0 embed: second_conv.3's input gradient (1,048,576 rows)
    if (K == 96 && N == 192) GM3D_WS_LAUNCH(2, 6, 1, 0, 1)           // level-1 token embed: second_conv.0 on the local half (262,144 rows)
    if (K == 192 && N == 96) GM3D_WS_LAUNCH(3, 3, 1, 0, 1)           // ... and its input gradient
    if (K == 96 && N == 384) GM3D_WS_LAUNCH(2, 6, 1, 0, 1)           // fc1, fc2's input gradient (two column blocks of 192)
    if (K == 384 && N == 96) GM3D_WS_LAUNCH(6, 3, 1, 0, 1)           // fc2, fc1's input gradient
    if (K == 288 && N == 96) GM3D_WS_LAUNCH(5, 3, 1, 0, 1)           // qkv's input gradient
#undef GM3D_WS_LAUNCH
    return GM3D_EUNSUPPORTED;
}

extern "C" int gm3d_gemm_tn_bf16_ws(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int lda, int ldw,
                                    int ldc, gm3d_stream_t stream) {
    if (!C) return GM3D_EINVAL;
    return ws_launch(A, W, bias, C, nullptr, nullptr, M, N, K, lda, ldw, ldc, 0, 0, stream);
}

extern "C" int gm3d_gemm_tn_bf16_ws_pool(const void* A, const void* W, const float* bias, void* C, void* P, uint8_t* arg, int M, int N,
                                         int K, int lda, int ldw, int ldc, int ldp, int bias_after_pool, gm3d_stream_t stream) {
    if (!P || !arg) return GM3D_EINVAL;
    return ws_launch(A, W, bias, C, P, arg, M, N, K, lda, ldw, ldc, ldp, bias_after_pool, stream);
}

// the same with groups of `group_rows` = 16 or 32 rows: P / arg (M / group_rows, N)
extern "C" int gm3d_gemm_tn_bf16_ws_poolg(const void* A, const void* W, const float* bias, void* C, void* P, uint8_t* arg, int M, int N,
                                          int K, int lda, int ldw, int ldc, int ldp, int bias_after_pool, int group_rows,
                                          gm3d_stream_t stream) {
    if (!P || !arg || (group_rows != 16 && group_rows != 32)) return GM3D_EINVAL;
    return ws_launch(A, W, bias, C, P, arg, M, N, K, lda, ldw, ldc, ldp, bias_after_pool, stream, 0, gm3d::WsBn(), group_rows == 16);
}

// C = act((bf16(A.W^T) + T[row / 32]) * scale + shift), act(h) = h > 0 ? h : slope h: the product, the per-group term of the split
// concat and an eval-mode BatchNorm + ReLU in one launch (T (M / 32, N) bf16, scale / shift (N) f32).  M % 32 == 0.
extern "C" int gm3d_gemm_tn_bf16_ws_bn_apply(const void* A, const void* W, const void* T, const float* scale, const float* shift, float slope,
                                             void* C, int M, int N, int K, int lda, int ldw, int ldt, int ldc, gm3d_stream_t stream) {
    if (!C || !T || !scale || !shift) return GM3D_EINVAL;
    gm3d::WsBn bn;
    bn.T = (const gm3d::bf16_t*)T; bn.scale = scale; bn.shift = shift; bn.partial = nullptr; bn.ldt = ldt; bn.slope = slope;
    return ws_launch(A, W, nullptr, C, nullptr, nullptr, M, N, K, lda, ldw, ldc, 0, 0, stream, 4, bn);
}

// C = bf16(A.W^T) and, per workgroup, the column sums of y = C + T[row / 32] and of y^2: partial (gm3d_gemm_ws_stats_rows, 2 N) f32,
// every row written (no memset), to be summed over the rows in order (gm3d_colsum_finish) -- the train-mode statistics of the
// BatchNorm behind the product without another pass over it.
extern "C" int gm3d_gemm_tn_bf16_ws_bn_stats(const void* A, const void* W, const void* T, void* C, float* partial, int M, int N, int K,
                                             int lda, int ldw, int ldt, int ldc, gm3d_stream_t stream) {
    if (!C || !T || !partial) return GM3D_EINVAL;
    gm3d::WsBn bn;
    bn.T = (const gm3d::bf16_t*)T; bn.scale = nullptr; bn.shift = nullptr; bn.partial = partial; bn.ldt = ldt; bn.slope = 0.f;
    return ws_launch(A, W, nullptr, C, nullptr, nullptr, M, N, K, lda, ldw, ldc, 0, 0, stream, 5, bn);
}

// the two BatchNorm epilogues for groups of `group_rows` = 16 or 32 rows (T (M / group_rows, N)): Point-M2AE's level-0 groups hold 16 points
extern "C" int gm3d_gemm_tn_bf16_ws_bn_apply_g(const void* A, const void* W, const void* T, const float* scale, const float* shift, float slope,
                                               void* C, int M, int N, int K, int lda, int ldw, int ldt, int ldc, int group_rows,
                                               gm3d_stream_t stream) {
    if (!C || !T || !scale || !shift || (group_rows != 16 && group_rows != 32)) return GM3D_EINVAL;
    gm3d::WsBn bn;
    bn.T = (const gm3d::bf16_t*)T; bn.scale = scale; bn.shift = shift; bn.partial = nullptr; bn.ldt = ldt; bn.slope = slope;
    return ws_launch(A, W, nullptr, C, nullptr, nullptr, M, N, K, lda, ldw, ldc, 0, 0, stream, 4, bn, group_rows == 16);
}

extern "C" int gm3d_gemm_tn_bf16_ws_bn_stats_g(const void* A, const void* W, const void* T, void* C, float* partial, int M, int N, int K,
                                               int lda, int ldw, int ldt, int ldc, int group_rows, gm3d_stream_t stream) {
    if (!C || !T || !partial || (group_rows != 16 && group_rows != 32)) return GM3D_EINVAL;
    gm3d::WsBn bn;
    bn.T = (const gm3d::bf16_t*)T; bn.scale = nullptr; bn.shift = nullptr; bn.partial = partial; bn.ldt = ldt; bn.slope = 0.f;
    return ws_launch(A, W, nullptr, C, nullptr, nullptr, M, N, K, lda, ldw, ldc, 0, 0, stream, 5, bn, group_rows == 16);
}

// rows of gm3d_gemm_tn_bf16_ws_bn_stats' partial buffer for this shape (0: unsupported)
extern "C" int gm3d_gemm_ws_stats_rows(int M, int N, int K) {
    if (!(K == 256 && N == 512) || M % 32 || M < 32) return 0;
    constexpr int KT = 4, NW = 8, NL = 1;
    constexpr int TILE_ = KT * 4096, STAGE_ = ((32 * NW + 63) / 64) * 4096;
    const size_t lds = (size_t)(ws_depth(4 * KT / NL, TILE_, STAGE_) + 1) * TILE_ + STAGE_;
    const int tiles_n = N / (32 * NW);
    return ws_grid(M / 32, tiles_n, lds) / tiles_n;
}

// 1 when gm3d_gemm_tn_bf16_ws[_pool] has an instantiation for (N, K)
extern "C" int gm3d_gemm_ws_supported(int N, int K, int pool) {
    if (pool) return (K == 128 && N == 256) || (K == 512 && N == 384) || (K == 512 && N == 96);
    return (K == 256 && N == 512) || (K == 512 && N == 256) || (K == 384 && N == 512) || (K == 256 && N == 128) ||
           (K == 128 && N == 256) || (K == 512 && N == 384) || (K == 96 && (N == 288 || N == 96 || N == 384 || N == 192 || N == 512)) || (K == 384 && N == 96) ||
           (K == 288 && N == 96) || (K == 192 && N == 96);
}
