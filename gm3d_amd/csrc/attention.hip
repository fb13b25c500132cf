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

// ===================================================================== LDS images (bf16 kernels)
// A head's (T,64) bf16 slice lives in LDS as a swizzled [rows][64] image with 128-byte rows and NO padding: the 16-byte chunk
// ch (0..7) of row `row` sits at byte 128*row + 16*(ch ^ f(row)), f = the bit-reversed (row >> 1) & 7.  One image serves both
// kinds of MFMA operand read conflict-free (bank rule of ds_read_b128 / ds_read_b64_tr_b16: 64 banks, MI355X_MICROARCH.md LDS):
//   row read   (ds_read_b128):        a 16-lane group reads one chunk of 16 different rows -> f spreads them over all 16 slots;
//   transposed (ds_read_b64_tr_b16):  a 32-lane half reads 4 consecutive rows x 32 columns -> (row & 1, f bit 2) picks a
//                                     different 64-byte quarter of the bank row for each of the four rows.
__device__ __forceinline__ int img_off(int row, int ch) {
    const int f = (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1);
    return row * 128 + ((ch ^ f) << 4);
}
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
// operand fragment read along a row: 8 consecutive d (chunk ch) of row `row`
__device__ __forceinline__ bf16x8 img_row(const unsigned char* img, int row, int ch) {
    return *reinterpret_cast<const bf16x8*>(img + img_off(row, ch));
}
// operand fragment read down the columns: this lane's column is colblk + (lane & 15) (+16 for the upper 16-lane group of each
// half, folded into colblk by the caller); it receives M[row0 + 0..3][col] -- the hardware transpose of a 4 x 16 block whose
// row q / columns 4p..4p+3 address is supplied by lane 4q + p of the 16-lane group.  EXEC must be all ones (it is: no caller
// sits inside a lane-divergent branch).
__device__ __forceinline__ bf16x4 img_tr(const unsigned char* img, int row0, int colblk, int lane) {
    const int l16 = lane & 15, q = l16 >> 2, p = l16 & 3;
    const int col = colblk + 4 * p;
    const int addr = img_off(row0 + q, col >> 3) + ((col >> 2) & 1) * 8;
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + addr));
}
// A[i = column][k] fragment for the accumulator-as-operand k order (permk): rows k0 + 16 s + 4 hh + {0..3, 8..11}
__device__ __forceinline__ bf16x8 img_tr8(const unsigned char* img, int k0, int s, int hh, int colblk, int lane) {
    const bf16x4 lo = img_tr(img, k0 + 16 * s + 4 * hh, colblk, lane);
    const bf16x4 hi = img_tr(img, k0 + 16 * s + 8 + 4 * hh, colblk, lane);
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
    return v;
}
// Transposed result tile X^T[d (registers)][row (lane)] (two 32-d halves) -> the image rows of this wave's 32 rows, then out
// to global memory as whole 128-byte rows in 16-byte pieces (8 lanes per row).  `img` rows row0..row0+31 must be owned by the
// calling wave; the caller has made sure nobody still reads them.
__device__ __forceinline__ void store_tile_t(unsigned char* img, int row0, const f32x16& x0, const f32x16& x1, int lane, bf16_t* g,
                                             size_t row_stride, int T) {
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x16& x = dt ? x1 : x0;
            bf16x4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = (bf16_t)x[4 * g4 + e];          // d = 32 dt + 8 g4 + 4 hh + e
            *reinterpret_cast<bf16x4*>(img + img_off(row0 + r, 4 * dt + g4) + 8 * hh) = pk;
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (lane >> 3) + 8 * i, ch = lane & 7;
        if (row0 + row < T)
            *reinterpret_cast<uint4*>(g + (size_t)(row0 + row) * row_stride + ch * 8) =
                *reinterpret_cast<const uint4*>(img + img_off(row0 + row, ch));
    }
}

// The attention proper on the three LDS images, by wave w = query tile w of NT.  Oi: image whose rows 32 w .. 32 w + 31 this wave
// may overwrite to stage its output rows (the Q image itself when nobody else reads those rows afterwards).
template <int NT>
__device__ __forceinline__ void attn_core_bf16(const unsigned char* Qi, const unsigned char* Ki, const unsigned char* Vi,
                                               unsigned char* Oi, int w, int lane, int T, float scale, bf16_t* outp, size_t os,
                                               float* lsep) {
    const int r = lane & 31, hh = lane >> 5;
    // S^T tiles: rows = keys (regs), cols = queries (lanes)
    f32x16 st[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) st[t] = zero16();
    const int qrow = 32 * w + r;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const bf16x8 bq = img_row(Qi, qrow, 2 * s + hh);
#pragma unroll
        for (int t = 0; t < NT; ++t) st[t] = MFMA_BF16(img_row(Ki, 32 * t + r, 2 * s + hh), bq, st[t]);
    }
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int key = 32 * t + crow(g, hh);
            st[t][g] = key < T ? st[t][g] * scale : -INFINITY;
            m = fmaxf(m, st[t][g]);
        }
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            st[t][g] = __expf(st[t][g] - m);
            sum += st[t][g];
        }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;

    // O^T = V^T . P^T: the probability accumulators are the B operand as they stand (k = key in permk order), V^T comes out of
    // the V image by transposed reads; the result has d in registers (4 consecutive per quad) and the query on the lane.
    f32x16 o0 = zero16(), o1 = zero16();
    const int cb = 16 * ((lane >> 4) & 1);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8 p = cvt8(st[t], s, inv);
            o0 = MFMA_BF16(img_tr8(Vi, 32 * t, s, hh, cb, lane), p, o0);
            o1 = MFMA_BF16(img_tr8(Vi, 32 * t, s, hh, 32 + cb, lane), p, o1);
        }
    store_tile_t(Oi, 32 * w, o0, o1, lane, outp, os, T);
    if (lsep && hh == 0 && qrow < T) lsep[qrow] = m + logf(sum);
}

// ===================================================================== forward, bf16
// NT = number of 32-row tiles = waves (T <= 32 NT).  LDS: Q, K, V images.  Every 16-byte chunk of the head's q|k|v slab is
// requested up front with coalesced loads (8 lanes per 128-byte row piece), 12 per thread.
template <int NT>
__global__ __launch_bounds__(64 * NT) void attn_fwd_bf16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                float* __restrict__ lse, int T, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ROWS = 32 * NT, IMG = ROWS * 128;
    unsigned char* Qi = smem;
    unsigned char* Ki = Qi + IMG;
    unsigned char* Vi = Ki + IMG;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const size_t rs = (size_t)3 * H * HD, os = (size_t)H * HD;
    const bf16_t* Qg = qkv + (size_t)b * T * rs + (size_t)h * HD;
    {
        uint4 v[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {                      // 3 tiles x ROWS x 8 chunks = 768 NT = 12 x (64 NT threads)
            const int c = tid + i * 64 * NT;
            const int tile = c / (ROWS * 8), rc = c - tile * (ROWS * 8), row = rc >> 3, ch = rc & 7;
            const int rr = row < T ? row : T - 1;           // clamped, not predicated: the loads all issue back to back
            v[i] = *reinterpret_cast<const uint4*>(Qg + (size_t)rr * rs + (size_t)tile * os + ch * 8);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int c = tid + i * 64 * NT;
            const int tile = c / (ROWS * 8), rc = c - tile * (ROWS * 8), row = rc >> 3, ch = rc & 7;
            *reinterpret_cast<uint4*>(smem + tile * IMG + img_off(row, ch)) = row < T ? v[i] : make_uint4(0, 0, 0, 0);
        }
    }
    __syncthreads();

    // this wave's Q rows are read by nobody else: the output tile is staged there and written as whole rows
    attn_core_bf16<NT>(Qi, Ki, Vi, Qi, w, lane, T, scale, out + (size_t)b * T * os + (size_t)h * HD, os,
                       lse ? lse + ((size_t)b * H + h) * T : nullptr);
}

// ===================================================================== qkv projection + attention in one launch (bf16, T <= 64)
// Beneath: timm Attention.forward lines qkv = self.qkv(x) ... x = (attn @ v) (in-tree twin Point-MAE_SA3D/models/Point_MAE.py:113-122)
// for one (cloud, head): the head's 192 rows of W_qkv (q | k | v slices, 64 x 384 each) times the cloud's 64 x 384 normalised
// tokens, then the attention of csrc's attn_core_bf16 on the products -- which never leave the CU: they are rounded to bf16 into
// the three LDS images exactly where attn_fwd_bf16_kernel would have loaded them from HBM.  The projection is the LDS-DMA ring of
// csrc/gemm_ring.hip on a 64 x 192 tile (stage = 64 token rows + 192 weight rows of 64 k = 32 KiB, 6 stages, LOOK in flight: 1, so that two workgroups share a CU);
// 4 waves: wave w owns token tile w >> 1 and columns 96 (w & 1) .. +95 of q|k|v.  The attention runs on waves 0 and 1 (query
// tiles) while, in the training variant, waves 2 and 3 write q|k|v out for the backward.  The workgroup order keeps a cloud's six
// heads on one XCD (its token tile is fetched from HBM once, then hit in that XCD's L2).
constexpr int QA_C = 384, QA_KT = QA_C / 64, QA_STAGE = (64 + 192) * 128;

__device__ __forceinline__ void qa_glds16(const void* gsrc, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
}

template <int LOOK, bool WRITE_QKV>
__global__ __launch_bounds__(256) void attn_qkv_fwd_bf16_kernel(const bf16_t* __restrict__ hin, const bf16_t* __restrict__ wqkv,
                                                                bf16_t* __restrict__ out, float* __restrict__ lse,
                                                                bf16_t* __restrict__ qkv_out, int T, int H, float scale, int total) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NBUF = LOOK + 1;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int per_xcd = gridDim.x >> 3;
    const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (logical >= total) return;
    const int b = logical / H, h = logical - b * H;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    // this lane's share of a stage: 1-KiB pieces p = w + 4 i (i < 8); p < 8: token rows 8p .. 8p+7, else weight rows 8(p-8) ..
    const int prow = lane >> 3, pslot = lane & 7;
    const bf16_t* src[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int p = w + 4 * i;
        if (p < 8) {
            const int row = 8 * p + prow;
            const int t = row < T ? row : T - 1;                          // rows past T: clamped here, zeroed in the images
            const int f = (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1);
            src[i] = hin + ((size_t)b * T + t) * QA_C + ((pslot ^ f) << 3);
        } else {
            const int row = 8 * (p - 8) + prow;                           // 0..191: q rows, k rows, v rows of this head
            const int f = (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1);
            src[i] = wqkv + ((size_t)(row >> 6) * H * HD + (size_t)h * HD + (row & 63)) * QA_C + ((pslot ^ f) << 3);
        }
    }
#define GM3D_QA_STAGE(ST)                                                                        \
    {                                                                                            \
        const unsigned base = lds0 + ((ST) % NBUF) * QA_STAGE;                                   \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) qa_glds16(src[i] + (ST) * 64, base + 1024 * (w + 4 * i)); \
    }
    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[j] = zero16();
    const int wm = (w >> 1) * 32, wn = (w & 1) * 96;
#pragma unroll
    for (int s = 0; s < LOOK; ++s) GM3D_QA_STAGE(s)
    // stage KT_: wait until it has landed (the younger stages in flight may stay out) AND until this wave's LDS reads of the stage
    // before have returned -- lgkmcnt(0): the barrier builtin is no memory fence for the compiler, which otherwise sinks the wait
    // for the last ds_read of stage KT_ - 1 below the barrier, while the refill issued by a faster wave right behind the barrier
    // goes to exactly that buffer (tools/kernel_stress.py: 5 of 3000 launches differed under a concurrent load) --, barrier, refill,
    // multiply
#define GM3D_QA_ITER(KT_)                                                                                                   \
    {                                                                                                                       \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(8 * ((QA_KT - 1 - (KT_)) < (LOOK - 1) ? (QA_KT - 1 - (KT_)) : (LOOK - 1))) : "memory"); \
        __builtin_amdgcn_s_barrier();                                                                                       \
        if ((KT_) + LOOK < QA_KT) GM3D_QA_STAGE((KT_) + LOOK)                                                               \
        const unsigned char* as = smem + ((KT_) % NBUF) * QA_STAGE;                                                         \
        const unsigned char* ws = as + 64 * 128;                                                                            \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                                     \
            const bf16x8 fa = img_row(as, wm + r, 2 * s + hh);                                                              \
            _Pragma("unroll") for (int j = 0; j < 3; ++j) acc[j] = MFMA_BF16(img_row(ws, wn + 32 * j + r, 2 * s + hh), fa, acc[j]); \
        }                                                                                                                   \
    }
    GM3D_QA_ITER(0) GM3D_QA_ITER(1) GM3D_QA_ITER(2) GM3D_QA_ITER(3) GM3D_QA_ITER(4) GM3D_QA_ITER(5)
#undef GM3D_QA_ITER
#undef GM3D_QA_STAGE
    // acc[j][g]: token wm + r, column wn + 32 j + crow(g, hh) of q|k|v.  The images alias ring buffer 0, whose last reader was
    // stage NBUF * floor(5 / NBUF) < 5 for every LOOK < 5: all waves are past it (the barrier of stage 5).
    static_assert((QA_KT - 1) % NBUF != 0, "the last stage must not sit in buffer 0");
    unsigned char* Qi = smem;
    unsigned char* Ki = smem + 8192;
    unsigned char* Vi = smem + 16384;
    unsigned char* Oi = smem + 24576;
    {
        const int token = wm + r;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int col = wn + 32 * j;
            unsigned char* img = smem + (col >> 6) * 8192;
            const int cbase = (col & 63) >> 3;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bf16x4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = token < T ? (bf16_t)acc[j][4 * q + e] : (bf16_t)0.0f;
                *reinterpret_cast<bf16x4*>(img + img_off(token, cbase + q) + 8 * hh) = pk;
            }
        }
    }
    __syncthreads();
    const size_t os = (size_t)H * HD;
    if (w < 2) {
        attn_core_bf16<2>(Qi, Ki, Vi, Oi, w, lane, T, scale, out + (size_t)b * T * os + (size_t)h * HD, os,
                          lse ? lse + ((size_t)b * H + h) * T : nullptr);
    } else if (WRITE_QKV) {
        bf16_t* g = qkv_out + (size_t)b * T * 3 * os + (size_t)h * HD;
        const int t2 = tid - 128;
#pragma unroll
        for (int i = 0; i < 12; ++i) {                                    // 3 images x 64 rows x 8 chunks = 12 x 128 threads
            const int c = t2 + 128 * i;
            const int tile = c >> 9, row = (c >> 3) & 63, ch = c & 7;
            if (row < T)
                *reinterpret_cast<uint4*>(g + (size_t)row * 3 * os + (size_t)tile * os + ch * 8) =
                    *reinterpret_cast<const uint4*>(smem + tile * 8192 + img_off(row, ch));
        }
    }
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
// LDS (dynamic): Q, K, V, dO images (ROWS x 128 B each) + lse[ROWS] + delta[ROWS].  2 NT waves: waves 0..NT-1 each own one key
// tile (dK, dV), waves NT..2NT-1 each own one query tile (dQ) -- the two phases of a (b,h) problem run side by side.  Every
// gradient tile is produced TRANSPOSED (d in registers, key / query on the lane: X^T = M^T . Y with M^T read from an image by
// transposed reads and the recomputed P / dS accumulators as the B operand), staged through its own image rows and written as
// whole 128-byte rows.
template <int NT>
__global__ __launch_bounds__(128 * NT) void attn_bwd_bf16_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                                const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                                bf16_t* __restrict__ dqkv, int T, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ROWS = 32 * NT, IMG = ROWS * 128;
    unsigned char* Qi = smem;
    unsigned char* Ki = Qi + IMG;
    unsigned char* Vi = Ki + IMG;
    unsigned char* Di = Vi + IMG;
    float* Ls = reinterpret_cast<float*>(Di + IMG);
    float* Del = Ls + ROWS;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const size_t rs = (size_t)3 * H * HD, os = (size_t)H * HD;
    const bf16_t* Qg = qkv + (size_t)b * T * rs + (size_t)h * HD;
    const bf16_t* Og = out + (size_t)b * T * os + (size_t)h * HD;
    const bf16_t* Dg = dout + (size_t)b * T * os + (size_t)h * HD;
    {
        uint4 v[6], d[2], o[2];
#pragma unroll
        for (int i = 0; i < 6; ++i) {                       // Q, K, V: 3 x ROWS x 8 chunks = 768 NT = 6 x (128 NT threads)
            const int c = tid + i * 128 * NT;
            const int tile = c / (ROWS * 8), rc = c - tile * (ROWS * 8), row = rc >> 3, ch = rc & 7;
            const int rr = row < T ? row : T - 1;
            v[i] = *reinterpret_cast<const uint4*>(Qg + (size_t)rr * rs + (size_t)tile * os + ch * 8);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {                       // dO and O: ROWS x 8 chunks = 2 x (128 NT threads)
            const int c = tid + i * 128 * NT, row = c >> 3, ch = c & 7;
            const int rr = row < T ? row : T - 1;
            d[i] = *reinterpret_cast<const uint4*>(Dg + (size_t)rr * os + ch * 8);
            o[i] = *reinterpret_cast<const uint4*>(Og + (size_t)rr * os + ch * 8);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int c = tid + i * 128 * NT;
            const int tile = c / (ROWS * 8), rc = c - tile * (ROWS * 8), row = rc >> 3, ch = rc & 7;
            *reinterpret_cast<uint4*>(smem + tile * IMG + img_off(row, ch)) = row < T ? v[i] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + i * 128 * NT, row = c >> 3, ch = c & 7;
            const uint4 dz = row < T ? d[i] : make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(Di + img_off(row, ch)) = dz;
            // delta[q] = sum_d dO[q][d] * O[q][d]: the 8 lanes that hold a row's chunks meet by lane exchange
            const bf16x8 d8 = *reinterpret_cast<const bf16x8*>(&dz), o8 = *reinterpret_cast<const bf16x8*>(&o[i]);
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += (float)o8[j] * (float)d8[j];
            acc += __shfl_xor(acc, 1);
            acc += __shfl_xor(acc, 2);
            acc += __shfl_xor(acc, 4);
            if (ch == 0) {
                Del[row] = acc;
                Ls[row] = row < T ? lse[((size_t)b * H + h) * T + row] : 0.f;
            }
        }
    }
    __syncthreads();

    bf16_t* dQg = dqkv + (size_t)b * T * rs + (size_t)h * HD;
    const int cb = 16 * ((lane >> 4) & 1);
    f32x16 x0 = zero16(), x1 = zero16(), y0 = zero16(), y1 = zero16();      // A: dV^T, dK^T;  B: dQ^T (y unused)
    // ---- phase A: this wave owns key tile kt.  Tiles X[q][key]: rows = q (regs), cols = key (lanes).
    if (w < NT) {
        const int kt = w;
        const int key = 32 * kt + r;
        for (int qt = 0; qt < NT; ++qt) {
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int ch = 2 * ks + hh;
                s = MFMA_BF16(img_row(Qi, 32 * qt + r, ch), img_row(Ki, key, ch), s);
                dp = MFMA_BF16(img_row(Di, 32 * qt + r, ch), img_row(Vi, key, ch), dp);
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int q = 32 * qt + crow(g, hh);
                const float p = (q < T && key < T) ? __expf(s[g] * scale - Ls[q]) : 0.f;
                s[g] = p;                       // P[q][key]
                dp[g] = p * (dp[g] - Del[q]);   // dS[q][key]
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {  // sum over q (X rows): M^T . X, M^T[i = d][k = q] by transposed reads
                const bf16x8 pa = cvt8(s, ks, 1.0f);
                const bf16x8 da = cvt8(dp, ks, scale);
                x0 = MFMA_BF16(img_tr8(Di, 32 * qt, ks, hh, cb, lane), pa, x0);            // dV^T[d][key]
                x1 = MFMA_BF16(img_tr8(Di, 32 * qt, ks, hh, 32 + cb, lane), pa, x1);
                y0 = MFMA_BF16(img_tr8(Qi, 32 * qt, ks, hh, cb, lane), da, y0);            // dK^T[d][key]
                y1 = MFMA_BF16(img_tr8(Qi, 32 * qt, ks, hh, 32 + cb, lane), da, y1);
            }
        }
    }
    // ---- phase B: this wave owns query tile qt.  Tiles X'[key][q]: rows = key (regs), cols = q (lanes).
    else {
        const int qt = w - NT;
        const int q = 32 * qt + r;
        const float lq = Ls[q], dq_ = Del[q];
        for (int kt = 0; kt < NT; ++kt) {
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int ch = 2 * ks + hh;
                s = MFMA_BF16(img_row(Ki, 32 * kt + r, ch), img_row(Qi, q, ch), s);
                dp = MFMA_BF16(img_row(Vi, 32 * kt + r, ch), img_row(Di, q, ch), dp);
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int key = 32 * kt + crow(g, hh);
                const float p = (q < T && key < T) ? __expf(s[g] * scale - lq) : 0.f;
                dp[g] = p * (dp[g] - dq_);  // dS^T[key][q]
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {  // sum over key: K^T . dS^T
                const bf16x8 da = cvt8(dp, ks, scale);
                x0 = MFMA_BF16(img_tr8(Ki, 32 * kt, ks, hh, cb, lane), da, x0);            // dQ^T[d][q]
                x1 = MFMA_BF16(img_tr8(Ki, 32 * kt, ks, hh, 32 + cb, lane), da, x1);
            }
        }
    }
    __syncthreads();            // every wave has finished reading the images: they become the staging area of the results
    if (w < NT) {
        store_tile_t(Vi, 32 * w, x0, x1, lane, dQg + 2 * os, rs, T);
        store_tile_t(Ki, 32 * w, y0, y1, lane, dQg + os, rs, T);
    } else {
        store_tile_t(Qi, 32 * (w - NT), x0, x1, lane, dQg, rs, T);
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
    const int nt = (T + 31) / 32;
    if (dtype == GM3D_BF16) {
        const size_t lds = (size_t)3 * 32 * nt * 128;
#define GM3D_AF(NT_) launch_attn(attn_fwd_bf16_kernel<NT_>, B * H, 64 * NT_, lds, st, (const bf16_t*)qkv, (bf16_t*)out, lse, T, H, scale)
        return nt == 1 ? GM3D_AF(1) : nt == 2 ? GM3D_AF(2) : nt == 3 ? GM3D_AF(3) : GM3D_AF(4);
#undef GM3D_AF
    }
    const int threads = 64 * nt;
    const int rows = T <= 64 ? 64 : 128;
    const size_t lds = sizeof(float) * 3 * rows * 64;
    return T <= 64 ? launch_attn(attn_fwd_f32_kernel<2>, B * H, threads, lds, st, (const float*)qkv, (float*)out, lse, T, H, scale)
                   : launch_attn(attn_fwd_f32_kernel<4>, B * H, threads, lds, st, (const float*)qkv, (float*)out, lse, T, H, scale);
}

extern "C" int gm3d_attention_qkv_fwd(const void* h, const void* wqkv, void* out, float* lse, void* qkv_out, int B, int T, int H,
                                      int C, float scale, int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!h || !wqkv || !out || B < 0 || T < 1 || H < 1) return GM3D_EINVAL;
    if (dtype != GM3D_BF16 || T > 64 || C != QA_C || H * HD != C) return GM3D_EUNSUPPORTED;
    if ((((size_t)h | (size_t)wqkv | (size_t)out | (size_t)qkv_out) & 15)) return GM3D_EUNSUPPORTED;       // 16-byte LDS-DMA / row stores
    if ((long long)B * H > 0x7ffffff0LL) return GM3D_EUNSUPPORTED;
    if (B == 0) return GM3D_OK;
    // LOOK = 1 (64 KiB of LDS, two workgroups per CU) measured best: 10.8 us at B = 64 against 13.7 / 14.2 for LOOK = 2 / 3
    constexpr int LOOK = 1;
    const size_t lds = (size_t)(LOOK + 1) * QA_STAGE;
    const int total = B * H, grid = (total + 7) / 8 * 8;
    hipStream_t st = (hipStream_t)stream;
    return qkv_out ? launch_attn(attn_qkv_fwd_bf16_kernel<LOOK, true>, grid, 256, lds, st, (const bf16_t*)h, (const bf16_t*)wqkv,
                                 (bf16_t*)out, lse, (bf16_t*)qkv_out, T, H, scale, total)
                   : launch_attn(attn_qkv_fwd_bf16_kernel<LOOK, false>, grid, 256, lds, st, (const bf16_t*)h, (const bf16_t*)wqkv,
                                 (bf16_t*)out, lse, (bf16_t*)nullptr, T, H, scale, total);
}

extern "C" int gm3d_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                                  int B, int T, int H, float scale, int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    int rc = attn_check(qkv, out, B, T, H, dtype);
    if (rc != GM3D_OK) return rc;
    if (!dout || !lse || !dqkv) return GM3D_EINVAL;
    if (B == 0) return GM3D_OK;
    hipStream_t st = (hipStream_t)stream;
    const int nt = (T + 31) / 32;
    if (dtype == GM3D_BF16) {
        const size_t lds = (size_t)4 * 32 * nt * 128 + 2 * 32 * nt * sizeof(float);
#define GM3D_AB(NT_)                                                                                                      \
    launch_attn(attn_bwd_bf16_kernel<NT_>, B * H, 128 * NT_, lds, st, (const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, \
                (bf16_t*)dqkv, T, H, scale)
        return nt == 1 ? GM3D_AB(1) : nt == 2 ? GM3D_AB(2) : nt == 3 ? GM3D_AB(3) : GM3D_AB(4);
#undef GM3D_AB
    }
    const int threads = 128 * nt;   // NT key-tile waves + NT query-tile waves
    const int rows = T <= 64 ? 64 : 128;
    const size_t lds = sizeof(float) * (4 * rows * 64 + 2 * rows);
    return T <= 64 ? launch_attn(attn_bwd_f32_kernel<2>, B * H, threads, lds, st, (const float*)qkv, (const float*)out,
                                 (const float*)dout, lse, (float*)dqkv, T, H, scale)
                   : launch_attn(attn_bwd_f32_kernel<4>, B * H, threads, lds, st, (const float*)qkv, (const float*)out,
                                 (const float*)dout, lse, (float*)dqkv, T, H, scale);
}
