// dW[b][n][k] = sum_r dY[b][r][n] * X[b][r][k]  -- the weight-gradient GEMMs of the path: bf16 operands that are BOTH row-major
// with the reduction index r (token / point rows: 2048 .. 262,144 of them) as the slow dimension, fp32 result (N, K = 128 .. 1536)
// written straight into the parameters' slots of the optimizer's flat gradient buffer, batched over the blocks of a stack.
//
// Beneath: the .grad of every nn.Linear / Conv1d(k=1) weight on the path -- timm Block qkv / proj / fc1 / fc2 (in-tree twin
// Point-MAE_SA3D/models/Point_MAE.py:82-125) and the mini-PointNet convs (models_mae_learn_loss.py:873-882) -- which the
// reference leaves to autograd (addmm backward: one cuBLAS NT GEMM per weight per step).
//
// Design (MI355X).  Neither operand has the reduction index contiguous, which is what an MFMA fragment wants (8 consecutive k
// per lane).  Both are therefore staged as they lie in memory -- [32 rows][128 columns] bf16 tiles with coalesced 16-byte
// pieces, by LDS-DMA (global_load_lds_dwordx4: no staging registers, a ring of three stages, two in flight behind a COUNTED
// s_waitcnt vmcnt) -- and every fragment is read with the hardware-transposing ds_read_b64_tr_b16 (4 rows x 16 columns per
// 16-lane group, column-major out).  The LDS image is the 256-byte-row image with the chunk XOR
// ((row & 3) << 2 | (row >> 2) & 3) (conflict-free transposed reads); LDS-DMA writes lane-linear, so the XOR sits on the
// per-lane SOURCE address.  128 x 128 output tile per 256-thread workgroup (4 waves x (64 x 64) = 2 x 2 MFMA 32x32x16 tiles),
// 48 KiB of LDS -> three workgroups per CU, or two beside a workgroup of the backward chain the launch runs next to.  Long reductions over few output tiles (the 4-block decoders: 27 tiles per
// weight kind and block) are split over row ranges into fp32 partial slabs that gm3d_sum_few_rows adds in a fixed order
// (deterministic; no atomics).
#include "common.hpp"

namespace gm3d {

typedef __bf16 nbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 nbf16x4 __attribute__((ext_vector_type(4)));
typedef float nf32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) nbf16x4 lds_nbf16x4;

constexpr int NT_BR = 32;                 // reduction rows per stage
constexpr int NT_TILE = NT_BR * 256;      // bytes of one [32][128] bf16 tile
constexpr int NT_STAGE = 2 * NT_TILE;     // dY tile + X tile
constexpr int NT_NBUF = 3;

__device__ __forceinline__ int nt_f(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// one LDS-DMA piece: 64 lanes x 16 bytes -> 1 KiB at the wave-uniform LDS byte address `dst`; the compiler does not see this
// load (no automatic s_waitcnt vmcnt(0) before the next LDS read): completion is counted by hand below
__device__ __forceinline__ void glds16(const void* gsrc, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
}

// fragment for mfma_32x32x16 out of a [32 rows][128 cols] image: element j of this lane = M[k = kk0 + 8 hh + j][col], the lane's
// column being colblk + (lane & 15), colblk a multiple of 16 that already includes this lane's 16-lane group
__device__ __forceinline__ nbf16x8 nt_frag(const unsigned char* img, int kk0, int hh, int colblk, int lane) {
    const int l16 = lane & 15, q = l16 >> 2, p = l16 & 3;
    const int col = colblk + 4 * p, ch = col >> 3, sub = (col >> 2) & 1;
    const int r0 = kk0 + 8 * hh + q, r1 = r0 + 4;
    const nbf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_nbf16x4*)(img + 256 * r0 + 16 * (ch ^ nt_f(r0)) + 8 * sub));
    const nbf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_nbf16x4*)(img + 256 * r1 + 16 * (ch ^ nt_f(r1)) + 8 * sub));
    nbf16x8 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
    return v;
}

// grid.x = batch * splits * tiles_n * tiles_k (rounded up to a multiple of 8, XCD-aware order).  Rows of split s:
// [s * rows_split, (s + 1) * rows_split); rows_split % 32 == 0.
// one 128 x 128 output tile of one (batch, split): `logical` = this workgroup's tile number inside its problem (XCD-aware order already applied)
__device__ __forceinline__ void nt_tile(unsigned char* nsm, const bf16_t* __restrict__ Y, const bf16_t* __restrict__ X, float* __restrict__ O,
                                        int rows_split, int ldy, int ldx, int ldo, long long sY, long long sX, long long sO, long long sOs,
                                        int splits, int tiles_n, int tiles_k, int logical, int Nfull, int Kfull, int* __restrict__ counters,
                                        float* __restrict__ Ofin, long long sOfin, int ldofin, int tn_fastest = 0) {
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    int t = logical;
    int tk, tn;
    if (tn_fastest) { tn = t % tiles_n; t /= tiles_n; tk = t % tiles_k; t /= tiles_k; }     // measurement knob (gm3d_gemm_nt_set_order)
    else { tk = t % tiles_k; t /= tiles_k; tn = t % tiles_n; t /= tiles_n; }
    const int sp = t % splits;
    const int b = t / splits;
    const int n0 = tn * 128, k0 = tk * 128;
    const bf16_t* Yb = Y + (size_t)b * sY + (size_t)sp * rows_split * ldy + n0;
    const bf16_t* Xb = X + (size_t)b * sX + (size_t)sp * rows_split * ldx + k0;
    float* Ob = O + (size_t)b * sO + (size_t)sp * sOs;
    const int nst = rows_split / NT_BR;

    // staging role of this lane: each wave moves pieces w and w + 4 of both tiles (a piece = 4 rows x 256 B = 1 KiB)
    const int prow = lane >> 4, pcs = lane & 15;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)nsm;
    size_t yoff[2], xoff[2];
    // ragged N / K (multiples of 8: the 96 / 192 / 288 / 576-wide layers of Point-M2AE): chunks past the matrix edge are redirected to
    // the tile's last valid chunk (in bounds; they only feed output elements that are never stored)
    const int vy = (Nfull - n0) / 8 < 16 ? (Nfull - n0) / 8 : 16, vx = (Kfull - k0) / 8 < 16 ? (Kfull - k0) / 8 : 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 4 * (w + 4 * i) + prow;                    // row inside the stage's 32
        const int ch = pcs ^ nt_f(row);                            // the chunk that belongs at LDS slot pcs of this row
        yoff[i] = (size_t)row * ldy + (ch < vy ? ch : vy - 1) * 8;
        xoff[i] = (size_t)row * ldx + (ch < vx ? ch : vx - 1) * 8;
    }
#define GM3D_NT_STAGE(ST)                                                                                    \
    {                                                                                                        \
        const unsigned base = lds0 + ((ST) % NT_NBUF) * NT_STAGE;                                            \
        const size_t ro = (size_t)(ST) * NT_BR;                                                              \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                      \
            glds16(Yb + ro * ldy + yoff[i], base + 1024 * (w + 4 * i));                                      \
            glds16(Xb + ro * ldx + xoff[i], base + NT_TILE + 1024 * (w + 4 * i));                            \
        }                                                                                                    \
    }

    nf32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;
    const int wn = (w >> 1) * 64, wk = (w & 1) * 64;
    const int cb = 16 * ((lane >> 4) & 1);

    GM3D_NT_STAGE(0)
    if (nst > 1) GM3D_NT_STAGE(1)
    for (int st = 0; st < nst; ++st) {
        // every wave issues 4 pieces per stage: stage st has landed (this wave's share) when at most the pieces of the one stage
        // issued after it are outstanding
        if (st + 1 < nst) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                   // ... and everybody else's share; everybody is past its reads of stage st-1,
        if (st + 2 < nst) GM3D_NT_STAGE(st + 2)         // whose buffer (three in the ring) stage st+2 now refills
        const unsigned char* ys = nsm + (st % NT_NBUF) * NT_STAGE;
        const unsigned char* xs = ys + NT_TILE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            nbf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = nt_frag(ys, 16 * ks, hh, wn + 32 * i + cb, lane);      // A[i = n][k = row]
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = nt_frag(xs, 16 * ks, hh, wk + 32 * j + cb, lane);      // B[k = row][j = k col]
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
#undef GM3D_NT_STAGE
    // result: rows (registers) = n, columns (lanes) = k: 128 contiguous bytes per half-wave and register
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int n = n0 + wn + 32 * i + (g & 3) + 8 * (g >> 2) + 4 * hh;
                if (n < Nfull && k0 + wk + 32 * j + r < Kfull) Ob[(size_t)n * ldo + k0 + wk + 32 * j + r] = acc[i][j][g];
            }
    if (counters) {
        // The slab sum inside this launch (instead of gm3d_sum_few_rows): the workgroup that completes a tile's LAST slab -- whichever
        // it is -- adds the slabs of the tile in slab order 0, 1, ... (the order of gm3d_sum_few_rows: same bits, independent of the
        // arrival order) and writes the result.  Release: slab stores -> fence -> counter; acquire: counter -> fence -> slab loads.
        // The counter resets itself for the next launch that is handed this slice.
        __shared__ int s_last;
        __threadfence();
        __syncthreads();
        if (tid == 0) {
            int* cnt = counters + ((size_t)b * tiles_n + tn) * tiles_k + tk;
            const int prev = atomicAdd(cnt, 1);
            s_last = prev == splits - 1;
            if (s_last) atomicExch(cnt, 0);
        }
        __syncthreads();
        if (s_last) {
            __threadfence();
            const float* P0 = O + (size_t)b * sO + (size_t)n0 * ldo + k0;
            float* F0 = Ofin + (size_t)b * sOfin + (size_t)n0 * ldofin + k0;
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                const int e = tid + 256 * i, row = e >> 5, c4 = (e & 31) * 4;
                if (n0 + row < Nfull && k0 + c4 < Kfull) {
                    const float* p = P0 + (size_t)row * ldo + c4;
                    typedef float nf32x4 __attribute__((ext_vector_type(4)));
                    nf32x4 a = __builtin_nontemporal_load(reinterpret_cast<const nf32x4*>(p));
                    for (int sp2 = 1; sp2 < splits; ++sp2) a += __builtin_nontemporal_load(reinterpret_cast<const nf32x4*>(p + (size_t)sp2 * sOs));
                    *reinterpret_cast<nf32x4*>(F0 + (size_t)row * ldofin + c4) = a;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256, 2) void gemm_nt_bf16_kernel(const bf16_t* __restrict__ Y, const bf16_t* __restrict__ X,
                                                              float* __restrict__ O, int rows_split, int ldy, int ldx, int ldo,
                                                              long long sY, long long sX, long long sO, long long sOs, int splits,
                                                              int tiles_n, int tiles_k, int total, int Nfull, int Kfull,
                                                              int* __restrict__ counters, float* __restrict__ Ofin, long long sOfin,
                                                              int ldofin) {
    extern __shared__ __attribute__((aligned(16))) unsigned char nsm[];
    const int per_xcd = gridDim.x >> 3;
    const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (logical >= total) return;
    nt_tile(nsm, Y, X, O, rows_split, ldy, ldx, ldo, sY, sX, sO, sOs, splits, tiles_n, tiles_k, logical, Nfull, Kfull, counters, Ofin, sOfin,
            ldofin);
}

// ---- several weight-gradient problems in ONE launch ------------------------------------------------------------------------------------
// The twelve batched products of a step's three block stacks (4 weight kinds each) used to be twelve launches of 60-100 us, each with its
// own ramp and its own partly filled last wave of tiles, on a stream where that time is ~90 % exposed.  Here they are one grid: problem j
// owns workgroups [first_j, first_j + grid_j) (grid_j = its tile count rounded up to 8, so blockIdx & 7 -- the XCD -- and the order of
// tiles inside a problem are exactly those of its own launch); workgroups are dealt in order, so the next problem's tiles fill the CUs
// the previous one's tail leaves idle.  The descriptors travel by value in the kernel arguments (captured with the node in a hipGraph).
struct NtProblem {
    const bf16_t* Y;
    const bf16_t* X;
    float* O;
    long long sY, sX, sO, sOs;
    int rows_split, ldy, ldx, ldo, splits, tiles_n, tiles_k, total, N, K, first, grid;
};
constexpr int NT_MAXP = 16;
struct NtMulti {
    int count;
    int tn_fastest;
    NtProblem p[NT_MAXP];
};

__global__ __launch_bounds__(256, 2) void gemm_nt_multi_kernel(NtMulti m) {
    extern __shared__ __attribute__((aligned(16))) unsigned char nsm[];
    const int bid = blockIdx.x;
    // the problem of this workgroup: a scan with compile-time indices (a run-time index into a by-value struct would go through scratch)
    NtProblem q = m.p[0];
#pragma unroll
    for (int j = 1; j < NT_MAXP; ++j)
        if (j < m.count && bid >= m.p[j].first) q = m.p[j];
    const int local = bid - q.first;
    const int per_xcd = q.grid >> 3;
    const int logical = (local & 7) * per_xcd + (local >> 3);
    if (logical >= q.total) return;
    nt_tile(nsm, q.Y, q.X, q.O, q.rows_split, q.ldy, q.ldx, q.ldo, q.sY, q.sX, q.sO, q.sOs, q.splits, q.tiles_n, q.tiles_k, logical, q.N, q.K,
            nullptr, nullptr, 0, 0, m.tn_fastest);
}

// ---- 128 x 384 output tiles (the block stacks' weight gradients: every (N, K) there is a multiple of (128, 384)) ----------------------
// The 128 x 128 kernel above reads its fragments and runs its MFMAs in alternating phases (after a stage's barrier every wave is in the
// same phase) and stages 16 KiB from L2 per 1 MFLOP.  Here: 512 threads, 8 waves x (64 x 96) = 2 x 3 MFMA tiles each -> 32 KiB per
// 3.1 MFLOP stage (1.5 x the flops per L2 byte), 5 fragments per 6 MFMAs, and a SOFTWARE PIPELINE over the two k-steps of a stage (below).
// Same stage (32 rows), same order of the sum over rows: results equal to the 128 x 128 kernel's for equal row splits.
// LDS: per stage four [32][128] images -- dY columns 0..127, X columns 0..383 -- three stages in the ring (96 KiB).  Wave w stages rows
// 4 w .. 4 w + 3 of every image (4 pieces / stage).
constexpr int NB_IMG = NT_BR * 256;
constexpr int NB_STAGE = 4 * NB_IMG;
constexpr int NB_NBUF = 3;

__global__ __launch_bounds__(512, 1) void gemm_nt384_bf16_kernel(const bf16_t* __restrict__ Y, const bf16_t* __restrict__ X,
                                                                 float* __restrict__ O, int rows_split, int ldy, int ldx, int ldo,
                                                                 long long sY, long long sX, long long sO, long long sOs, int splits,
                                                                 int tiles_n, int tiles_k, int total) {
    extern __shared__ __attribute__((aligned(16))) unsigned char nsm[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int per_xcd = gridDim.x >> 3;
    const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (logical >= total) return;
    int t = logical;
    const int tk = t % tiles_k; t /= tiles_k;
    const int tn = t % tiles_n; t /= tiles_n;
    const int sp = t % splits;
    const int b = t / splits;
    const int n0 = tn * 128, k0 = tk * 384;
    const bf16_t* Yb = Y + (size_t)b * sY + (size_t)sp * rows_split * ldy + n0;
    const bf16_t* Xb = X + (size_t)b * sX + (size_t)sp * rows_split * ldx + k0;
    float* Ob = O + (size_t)b * sO + (size_t)sp * sOs;
    const int nst = rows_split / NT_BR;

    const int prow = lane >> 4, pcs = lane & 15;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)nsm;
    const int srow = 4 * w + prow;                              // the row of every image this lane stages
    const int sch = pcs ^ nt_f(srow);                           // ... and the source chunk that belongs at its LDS slot
    const size_t yoff = (size_t)srow * ldy + sch * 8, xoff = (size_t)srow * ldx + sch * 8;
#define GM3D_NB_STAGE(ST)                                                                                    \
    {                                                                                                        \
        const unsigned base = lds0 + ((ST) % NB_NBUF) * NB_STAGE + 1024 * w;                                 \
        const size_t ro = (size_t)(ST) * NT_BR;                                                              \
        glds16(Yb + ro * ldy + yoff, base);                                                                  \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) glds16(Xb + ro * ldx + xoff + 128 * i, base + (1 + i) * NB_IMG); \
    }

    nf32x16 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;
    const int wn = (w >> 2) * 64, wk = (w & 3) * 96;
    const int cb = 16 * ((lane >> 4) & 1);

    // Software pipeline.  Two fragment sets in registers: while k-step 0's MFMAs run, k-step 1's fragments are read, and while those
    // run, the NEXT stage's k-step 0 fragments.  Each block is  2 MFMAs | 10 transposed reads of the next set | 4 MFMAs: the first
    // MFMAs wait for reads issued a block ago (nothing newer is outstanding, so the wait is exact), the reads then issue between MFMAs.
    // Ring of three stage buffers: passing the barrier of stage st + 1 means every wave has finished READING stage st (each waits for
    // its own LDS reads first), whose buffer stage st + 3 then refills.
    // Fragment addresses: 10 per lane (2 + 3 fragments x the two 4-row halves of a transposed read), kept in registers and advanced by
    // one stage buffer per stage; k-step 1 is the same address + 16 rows (the chunk XOR repeats every 16 rows).
    unsigned fad[10];
    {
        const int l16 = lane & 15, q = l16 >> 2, pp = l16 & 3;
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            const int c = (f < 2 ? wn + 32 * f : wk + 32 * (f - 2)) + cb;
            const int col = (c & 127) + 4 * pp, ch = col >> 3, sub = (col >> 2) & 1;
            const int img = f < 2 ? 0 : 1 + (c >> 7);
            const int r0 = 8 * hh + q, r1 = r0 + 4;
            fad[2 * f] = lds0 + img * NB_IMG + 256 * r0 + 16 * (ch ^ nt_f(r0)) + 8 * sub;
            fad[2 * f + 1] = lds0 + img * NB_IMG + 256 * r1 + 16 * (ch ^ nt_f(r1)) + 8 * sub;
        }
    }
    nbf16x8 fa0[2], fb0[3], fa1[2], fb1[3];
#define GM3D_NB_FRAGS(FA, FB, OFF)                                                                           \
    _Pragma("unroll") for (int f = 0; f < 5; ++f) {                                                          \
        const nbf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_nbf16x4*)(fad[2 * f] + (OFF)));     \
        const nbf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_nbf16x4*)(fad[2 * f + 1] + (OFF))); \
        nbf16x8 v;                                                                                           \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }                    \
        if (f < 2) FA[f] = v; else FB[f - 2] = v;                                                            \
    }
#define GM3D_NB_MFMA(FA, FB, I0, I1)                                                                         \
    _Pragma("unroll") for (int i = I0; i < I1; ++i)                                                          \
        _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                        \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[i], FB[j], acc[i][j], 0, 0, 0);

    GM3D_NB_STAGE(0)
    if (nst > 1) GM3D_NB_STAGE(1)
    if (nst > 1) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");         // four pieces per wave and stage: stage 0 landed, stage 1 may be in flight
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (nst > 2) GM3D_NB_STAGE(2)
    GM3D_NB_FRAGS(fa0, fb0, 0)
    for (int st = 0; st < nst; ++st) {
        __builtin_amdgcn_sched_barrier(0);
        GM3D_NB_MFMA(fa0, fb0, 0, 1)
        __builtin_amdgcn_sched_barrier(0);
        GM3D_NB_FRAGS(fa1, fb1, 4096)
        __builtin_amdgcn_sched_barrier(0);
        GM3D_NB_MFMA(fa0, fb0, 1, 2)
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < nst) {
            // stage st + 1 landed (this wave's pieces; the barrier covers everybody's); at most stage st + 2's pieces are behind it.
            // lgkmcnt(0): this wave's reads of stage st are complete before the barrier that lets its buffer be refilled.
            if (st + 2 < nst) {
                asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (st + 3 < nst) GM3D_NB_STAGE(st + 3)
            const unsigned step = (st % NB_NBUF == NB_NBUF - 1) ? (unsigned)(-(NB_NBUF - 1) * NB_STAGE) : (unsigned)NB_STAGE;
#pragma unroll
            for (int f = 0; f < 10; ++f) fad[f] += step;
            __builtin_amdgcn_sched_barrier(0);
            GM3D_NB_MFMA(fa1, fb1, 0, 1)
            __builtin_amdgcn_sched_barrier(0);
            GM3D_NB_FRAGS(fa0, fb0, 0)
            __builtin_amdgcn_sched_barrier(0);
            GM3D_NB_MFMA(fa1, fb1, 1, 2)
            __builtin_amdgcn_sched_barrier(0);
        } else {
            GM3D_NB_MFMA(fa1, fb1, 0, 2)
        }
    }
#undef GM3D_NB_FRAGS
#undef GM3D_NB_MFMA
#undef GM3D_NB_STAGE
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int n = n0 + wn + 32 * i + (g & 3) + 8 * (g >> 2) + 4 * hh;
                Ob[(size_t)n * ldo + k0 + wk + 32 * j + r] = acc[i][j][g];
            }
}

}  // namespace gm3d

// measurement knob: 0 (default) = 128 x 128 tiles everywhere, 1 = 128 x 384 tiles where the shape is a multiple of them.
// Measured (tools/wgrad_tiles.py, kernel only, MI355X): the big tiles reach 530 TFLOP/s where the 128 x 128 kernel reaches 620-790 at
// the splits the step uses (three 48 KiB workgroups per CU keep more L2 -> LDS bytes in flight than one 96 KiB workgroup does; a
// 192 x 384 / 8-wave form without the software pipeline reached 800-820 at 8 splits, i.e. with twice the slab-sum traffic) -> off.
static int NT_BIG_TILES = 0;
extern "C" int gm3d_gemm_nt_set_big_tiles(int on) {
    NT_BIG_TILES = on ? 1 : 0;
    return GM3D_OK;
}
static bool nt_big(int N, int K) { return NT_BIG_TILES && N % 128 == 0 && K % 384 == 0; }

extern "C" int gm3d_gemm_nt_splits(int batch, int R, int N, int K) {
    // the smallest power-of-two row split (<= 64, gm3d_sum_few_rows' limit) that offers the chip >= 400 workgroups (tools/wgrad_split_sweep.py: the optimum sits at 400-600) while every
    // split keeps >= 512 rows (16 stages)
    if (batch < 1 || R < 1 || N < 1 || K < 1) return 1;
    if (nt_big(N, K)) {
        // one 512-thread workgroup per CU: the smallest split that offers >= 224 of them, every split >= 256 rows (8 stages)
        const long long tiles = (long long)batch * (N / 128) * (K / 384);
        int s = 1;
        while (tiles * s < 224 && s < 64 && R % (64 * s) == 0 && R / (2 * s) >= 256) s *= 2;
        return s;
    }
    const long long tiles = (long long)batch * ((N + 127) / 128) * ((K + 127) / 128);
    int s = 1;
    while (tiles * s < 400 && s < 64 && R % (64 * s) == 0 && R / (2 * s) >= 512) s *= 2;
    return s;
}

static int nt_launch(const void* dY, const void* X, float* out, int batch, int R, int N, int K, int ldy, int ldx, int ldo,
                     long long stride_y, long long stride_x, long long stride_o, int splits, long long stride_split, gm3d_stream_t stream,
                     int* counters, float* fin, long long stride_fin, int ldfin) {
    using namespace gm3d;
    if (!dY || !X || !out || batch < 0 || R < 1 || N < 1 || K < 1 || splits < 1) return GM3D_EINVAL;
    if (N % 8 || K % 8 || ldy % 8 || ldx % 8 || ldy < N || ldx < K || ldo < K) return GM3D_EUNSUPPORTED;   // ragged N / K: multiples of 8
    if (R % (NT_BR * splits)) return GM3D_EUNSUPPORTED;
    if (((size_t)dY | (size_t)X) & 15) return GM3D_EUNSUPPORTED;
    if (splits > 1 && stride_split < (long long)N * ldo) return GM3D_EINVAL;
    if (batch == 0) return GM3D_OK;
    if (!counters && nt_big(N, K)) {
        const long long total = (long long)batch * splits * (N / 128) * (K / 384);
        if (total > 0x7ffffff0LL) return GM3D_EUNSUPPORTED;
        const int grid = (int)((total + 7) / 8 * 8);
        const size_t lds = (size_t)NB_NBUF * NB_STAGE;
        static LdsAttr attr;
        if (!attr.ensure((const void*)gemm_nt384_bf16_kernel, lds)) return GM3D_ELAUNCH;
        hipLaunchKernelGGL(gemm_nt384_bf16_kernel, dim3(grid), dim3(512), lds, (hipStream_t)stream, (const bf16_t*)dY, (const bf16_t*)X, out,
                           R / splits, ldy, ldx, ldo, stride_y, stride_x, stride_o, stride_split, splits, N / 128, K / 384, (int)total);
        GM3D_CHECK_LAUNCH();
        return GM3D_OK;
    }
    const int tn_ = (N + 127) / 128, tk_ = (K + 127) / 128;
    const long long total = (long long)batch * splits * tn_ * tk_;
    if (total > 0x7ffffff0LL) return GM3D_EUNSUPPORTED;
    const int grid = (int)((total + 7) / 8 * 8);
    const size_t lds = (size_t)NT_NBUF * NT_STAGE;
    static LdsAttr attr;
    if (!attr.ensure((const void*)gemm_nt_bf16_kernel, lds)) return GM3D_ELAUNCH;
    hipLaunchKernelGGL(gemm_nt_bf16_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)dY, (const bf16_t*)X, out,
                       R / splits, ldy, ldx, ldo, stride_y, stride_x, stride_o, stride_split, splits, tn_, tk_, (int)total, N, K, counters, fin,
                       stride_fin, ldfin);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_gemm_nt_bf16(const void* dY, const void* X, float* out, int batch, int R, int N, int K, int ldy, int ldx, int ldo,
                                 long long stride_y, long long stride_x, long long stride_o, int splits, long long stride_split,
                                 gm3d_stream_t stream) {
    return nt_launch(dY, X, out, batch, R, N, K, ldy, ldx, ldo, stride_y, stride_x, stride_o, splits, stride_split, stream, nullptr, nullptr, 0,
                     0);
}

// ... with the sum over the row splits inside the launch: part (batch, splits, N, K) f32 scratch for the slabs, out (batch, N, ldo) the
// result (batch stride stride_o), counters: gm3d_gemm_nt_tiles(N, K) * batch ints that are ZERO on entry and zero again on exit (a
// slice no concurrently running launch uses).  Bit-identical to gm3d_gemm_nt_bf16 into part followed by gm3d_sum_few_rows.
extern "C" int gm3d_gemm_nt_bf16_sum(const void* dY, const void* X, float* part, float* out, int* counters, int batch, int R, int N, int K,
                                     int ldy, int ldx, int ldo, long long stride_y, long long stride_x, long long stride_o, int splits,
                                     gm3d_stream_t stream) {
    if (!part || !out || !counters || splits < 2 || ldo < K || ldo % 4 || ((size_t)out & 15) || ((size_t)part & 15) || K % 4) return GM3D_EINVAL;
    return nt_launch(dY, X, part, batch, R, N, K, ldy, ldx, K, stride_y, stride_x, (long long)splits * N * K, splits, (long long)N * K, stream,
                     counters, out, stride_o, ldo);
}

namespace gm3d {
// out[b] = sum over the row splits of part[b][s], for several problems in one launch (grid.y = problem, grid.z = batch): the slab sums of
// gm3d_gemm_nt_bf16_multi, in slab order like gm3d_sum_few_rows.
struct NtSumJob {
    const float* src;
    float* dst;
    long long src_bstride, dst_bstride;      // in floats
    int nrows, batch;
    long long ncols4;
};
struct NtSumMulti {
    int count;
    NtSumJob j[NT_MAXP];
};

__global__ __launch_bounds__(256) void nt_sum_multi_kernel(NtSumMulti m) {
    NtSumJob q = m.j[0];
#pragma unroll
    for (int j = 1; j < NT_MAXP; ++j)
        if (j < m.count && (int)blockIdx.y == j) q = m.j[j];
    if ((int)blockIdx.z >= q.batch) return;
    const float4* p = reinterpret_cast<const float4*>(q.src + (size_t)blockIdx.z * q.src_bstride);
    float4* o = reinterpret_cast<float4*>(q.dst + (size_t)blockIdx.z * q.dst_bstride);
    for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < (size_t)q.ncols4; c += (size_t)gridDim.x * 256) {
        float4 a = p[c];
#pragma unroll 8
        for (int r = 1; r < q.nrows; ++r) {
            const float4 b = p[(size_t)r * q.ncols4 + c];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        o[c] = a;
    }
}
}  // namespace gm3d

// `count` (<= 16) weight-gradient problems -- each what gm3d_gemm_nt_bf16 computes, 128 x 128 tiles -- in ONE launch, followed by ONE launch
// that adds the row-split slabs of every problem with splits > 1 (part[j]: (batch, splits, N, K) f32 scratch; out[j]: (batch, N, K)
// contiguous rows of K, batch stride stride_o[j]).  Problems with splits == 1 write out[j] directly.  Results are bit-identical to the
// separate launches (same tiles, same slab order).
static int NT_MULTI_TN_FASTEST = 0;
extern "C" int gm3d_gemm_nt_set_order(int tn_fastest) {
    NT_MULTI_TN_FASTEST = tn_fastest ? 1 : 0;
    return GM3D_OK;
}

extern "C" int gm3d_gemm_nt_bf16_multi(int count, const void* const* dY, const void* const* X, float* const* out, float* const* part,
                                       const int* batch, const int* R, const int* N, const int* K, const int* ldy, const int* ldx,
                                       const long long* stride_y, const long long* stride_x, const long long* stride_o, const int* splits,
                                       gm3d_stream_t stream) {
    using namespace gm3d;
    if (count < 1 || count > NT_MAXP || !dY || !X || !out || !part || !batch || !R || !N || !K || !ldy || !ldx || !stride_y || !stride_x ||
        !stride_o || !splits)
        return GM3D_EINVAL;
    NtMulti m;
    NtSumMulti sm;
    m.count = count;
    m.tn_fastest = NT_MULTI_TN_FASTEST;
    sm.count = 0;
    long long first = 0;
    int max_batch = 1;
    long long max_cols4 = 0;
    for (int j = 0; j < count; ++j) {
        if (!dY[j] || !X[j] || !out[j] || batch[j] < 1 || R[j] < 1 || N[j] < 1 || K[j] < 1 || splits[j] < 1) return GM3D_EINVAL;
        if (N[j] % 8 || K[j] % 8 || ldy[j] % 8 || ldx[j] % 8 || ldy[j] < N[j] || ldx[j] < K[j] || R[j] % (NT_BR * splits[j]) || K[j] % 4)
            return GM3D_EUNSUPPORTED;
        if ((((size_t)dY[j] | (size_t)X[j]) & 15) || ((size_t)out[j] & 15)) return GM3D_EUNSUPPORTED;
        if (splits[j] > 1 && (!part[j] || ((size_t)part[j] & 15) || splits[j] > 64)) return GM3D_EINVAL;
        if (stride_o[j] < (long long)N[j] * K[j]) return GM3D_EINVAL;
        NtProblem& q = m.p[j];
        const int tn_ = (N[j] + 127) / 128, tk_ = (K[j] + 127) / 128;
        const long long total = (long long)batch[j] * splits[j] * tn_ * tk_;
        if (first + total > 0x7ffffff0LL) return GM3D_EUNSUPPORTED;
        q.Y = (const bf16_t*)dY[j]; q.X = (const bf16_t*)X[j];
        q.sY = stride_y[j]; q.sX = stride_x[j];
        q.rows_split = R[j] / splits[j]; q.ldy = ldy[j]; q.ldx = ldx[j]; q.ldo = K[j];
        q.splits = splits[j]; q.tiles_n = tn_; q.tiles_k = tk_; q.total = (int)total; q.N = N[j]; q.K = K[j];
        q.first = (int)first; q.grid = (int)((total + 7) / 8 * 8);
        if (splits[j] > 1) {
            q.O = part[j]; q.sO = (long long)splits[j] * N[j] * K[j]; q.sOs = (long long)N[j] * K[j];
            NtSumJob& sj = sm.j[sm.count++];
            sj.src = part[j]; sj.dst = out[j]; sj.src_bstride = q.sO; sj.dst_bstride = stride_o[j];
            sj.nrows = splits[j]; sj.batch = batch[j]; sj.ncols4 = (long long)N[j] * K[j] / 4;
            if (batch[j] > max_batch) max_batch = batch[j];
            if (sj.ncols4 > max_cols4) max_cols4 = sj.ncols4;
        } else {
            q.O = out[j]; q.sO = stride_o[j]; q.sOs = 0;
        }
        first += q.grid;
    }
    for (int j = count; j < NT_MAXP; ++j) m.p[j] = m.p[0];
    for (int j = sm.count; j < NT_MAXP; ++j) sm.j[j] = sm.j[0];
    const size_t lds = (size_t)NT_NBUF * NT_STAGE;
    static LdsAttr attr;
    if (!attr.ensure((const void*)gemm_nt_multi_kernel, lds)) return GM3D_ELAUNCH;
    hipLaunchKernelGGL(gemm_nt_multi_kernel, dim3((unsigned)first), dim3(256), lds, (hipStream_t)stream, m);
    GM3D_CHECK_LAUNCH();
    if (sm.count > 0) {
        if (max_batch > 65535) return GM3D_EUNSUPPORTED;
        const int gx = (int)((max_cols4 + 255) / 256 < 256 ? (max_cols4 + 255) / 256 : 256);
        hipLaunchKernelGGL(nt_sum_multi_kernel, dim3(gx, sm.count, max_batch), dim3(256), 0, (hipStream_t)stream, sm);
        GM3D_CHECK_LAUNCH();
    }
    return GM3D_OK;
}

extern "C" int gm3d_gemm_nt_tiles(int N, int K) { return N < 1 || K < 1 ? 0 : ((N + 127) / 128) * ((K + 127) / 128); }
