// Brute-force k-NN and the fused KNN + neighbourhood gather + centre subtract
// ("grouping") for gfx950.
//
// Beneath: knn_cuda.KNN(k, transpose_mode=True) and the gather/subtract tail of
//   Group.forward (reference: Point-MAE_SA3D/models_mae_learn_loss.py:924,946-957).
// Algorithm contract: SURVEY.md Appendix B / oracle_knn -- squared distance
// ((dx*dx+dy*dy)+dz*dz) fp32 without FMA, k smallest ascending, equal distances keep
// the lower reference index first, returned distance = sqrt.
//
// Design (MI355X): the cloud is staged once per workgroup into LDS as SoA x[],y[],z[]
// (coalesced HBM read, conflict-free lane-consecutive LDS reads).  One wavefront owns
// one query at a time and keeps the running top-k as a wave-distributed sorted list:
// lane j holds the j-th smallest 64-bit key (distance bits << 32 | index; distances are
// non-negative so the unsigned order is the (distance, index) lexicographic order).
// A chunk of 64 candidates is tested against the current k-th key with one ballot;
// each surviving candidate is inserted with {ballot -> popcount -> DPP wave_shr:1}.
// Nothing but the final k results ever leaves the CU: no N x G distance matrix in HBM.
#include "common.hpp"
#include <stdlib.h>

namespace gm3d {

constexpr int KNN_BLOCK = 256;
// queries per workgroup (a multiple of the 4 waves): every workgroup stages the whole cloud into LDS, so more queries per
// workgroup amortise that, but a query is a long chain of dependent wave-wide steps and only many resident waves hide it:
// 4 per workgroup (one per wave) until the grid exceeds ~8 waves per SIMD, then 8 / 16.
static int knn_qpb(int B, int G) {
    const long long queries = (long long)B * G;
    return queries <= 8192 * 2 ? 4 : (queries <= 8192 * 8 ? 8 : 16);
}

template <bool GROUP>
__global__ __launch_bounds__(KNN_BLOCK) void knn_kernel(const float* __restrict__ ref,
                                                        const float* __restrict__ query, int N, int G, int k,
                                                        float* __restrict__ dist, int64_t* __restrict__ idx,
                                                        float* __restrict__ nb, float* __restrict__ nbo, int qpb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* ys = xs + N;
    float* zs = ys + N;
    const int b = blockIdx.y;
    const float* r = ref + (size_t)b * N * 3;
    for (int i = threadIdx.x; i < N * 3; i += KNN_BLOCK) {
        const float v = r[i];
        const int n = i / 3, d = i - n * 3;
        (d == 0 ? xs : (d == 1 ? ys : zs))[n] = v;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q0 = blockIdx.x * qpb;

    for (int qi = wave; qi < qpb; qi += KNN_BLOCK / 64) {
        const int g = q0 + qi;
        if (g >= G) break;  // wave-uniform
        const float* q = query + ((size_t)b * G + g) * 3;
        const float qx = q[0], qy = q[1], qz = q[2];

        unsigned e_hi = 0xFFFFFFFFu, e_lo = 0xFFFFFFFFu;  // lane j: j-th smallest key so far
        unsigned long long tau = ~0ull;                    // key of lane k-1 (wave-uniform)

        for (int base = 0; base < N; base += 64) {
            const int n = base + lane;
            unsigned c_hi = 0xFFFFFFFFu, c_lo = 0xFFFFFFFFu;
            if (n < N) {
                c_hi = __float_as_uint(sqdist3(xs[n], ys[n], zs[n], qx, qy, qz));
                c_lo = (unsigned)n;
            }
            const unsigned long long c = ((unsigned long long)c_hi << 32) | c_lo;
            unsigned long long pending = __ballot(c < tau);
            while (pending) {
                const int src = __builtin_ctzll(pending);
                pending &= pending - 1;
                const unsigned ch = (unsigned)__builtin_amdgcn_readlane((int)c_hi, src);
                const unsigned cl = (unsigned)__builtin_amdgcn_readlane((int)c_lo, src);
                const unsigned long long cc = ((unsigned long long)ch << 32) | cl;
                if (cc < tau) {  // tau may have dropped since the ballot
                    const unsigned long long e = ((unsigned long long)e_hi << 32) | e_lo;
                    const int pos = __popcll(__ballot(e < cc));  // sorted ascending -> prefix
                    const unsigned sh_hi = dpp_u32<0x138>(e_hi, e_hi);  // wave_shr:1
                    const unsigned sh_lo = dpp_u32<0x138>(e_lo, e_lo);
                    if (lane > pos) { e_hi = sh_hi; e_lo = sh_lo; }
                    else if (lane == pos) { e_hi = ch; e_lo = cl; }
                    const unsigned th = (unsigned)__builtin_amdgcn_readlane((int)e_hi, k - 1);
                    const unsigned tl = (unsigned)__builtin_amdgcn_readlane((int)e_lo, k - 1);
                    tau = ((unsigned long long)th << 32) | tl;
                }
            }
        }

        if (lane < k) {
            const size_t o = ((size_t)b * G + g) * k + lane;
            const int n = (int)e_lo;
            if (idx) idx[o] = (int64_t)n;
            if (!GROUP) {
                if (dist) dist[o] = sqrtf(__uint_as_float(e_hi));  // correctly rounded (hipcc default)
            } else {
                const float x = xs[n], y = ys[n], z = zs[n];
                nb[o * 3 + 0] = __fsub_rn(x, qx);
                nb[o * 3 + 1] = __fsub_rn(y, qy);
                nb[o * 3 + 2] = __fsub_rn(z, qz);
                if (nbo) { nbo[o * 3 + 0] = x; nbo[o * 3 + 1] = y; nbo[o * 3 + 2] = z; }
            }
        }
    }
}


// Selection form for N <= 64*NPL (the path's 1024- and 2048-point clouds): the insertion list above is a chain of ~4.5 k
// dependent wave-wide steps per query.  Here a lane keeps the squared distances of its NPL points (point j*64 + lane) in
// registers and the wave
//   1. finds T = the k-th smallest distance by bisection on the 31 value bits -- count(u < candidate) is NPL ballots +
//      scalar popcounts, no cross-lane traffic;
//   2. takes every point with u < T plus, of the points with u == T, the lowest-indexed ones up to k (the (distance, index)
//      order of the contract), compacted into k LDS slots by ballot prefix;
//   3. rank-sorts the k 64-bit keys (distance bits << 32 | index; unique) and writes them out ascending.
// ~2 k wave instructions per query instead of ~17 k; bit-identical results.  (The bisection's scalar popcounts share the CU's one
// scalar unit across its 32 resident waves: that, not the vector ALU, sets the pace.)
template <bool GROUP, int NPL>
__global__ __launch_bounds__(KNN_BLOCK) void knn_select_kernel(const float* __restrict__ ref, const float* __restrict__ query, int N,
                                                               int G, int k, float* __restrict__ dist, int64_t* __restrict__ idx,
                                                               float* __restrict__ nb, float* __restrict__ nbo, int qpb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* ys = xs + N;
    float* zs = ys + N;
    unsigned long long* slots = reinterpret_cast<unsigned long long*>(smem + ((3 * N + 1) & ~1));   // [4 waves][64]
    const int b = blockIdx.y;
    const float* r = ref + (size_t)b * N * 3;
    for (int i = threadIdx.x; i < N * 3; i += KNN_BLOCK) {
        const float v = r[i];
        const int n = i / 3, d = i - n * 3;
        (d == 0 ? xs : (d == 1 ? ys : zs))[n] = v;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    unsigned long long* slot = slots + wave * 64;
    const int q0 = blockIdx.x * qpb;

    for (int qi = wave; qi < qpb; qi += KNN_BLOCK / 64) {
        const int g = q0 + qi;
        if (g >= G) break;  // wave-uniform
        const float* q = query + ((size_t)b * G + g) * 3;
        const float qx = q[0], qy = q[1], qz = q[2];
        unsigned u[NPL];
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int n = j * 64 + lane;
            u[j] = n < N ? __float_as_uint(sqdist3(xs[n], ys[n], zs[n], qx, qy, qz)) : 0xFFFFFFFFu;
        }
        // 1. T = k-th smallest: the largest v with count(u < v) < k.  (Squared distances are >= 0 or NaN: as unsigned integers
        //    they order like the floats, NaNs last; padding lanes hold 0xFFFFFFFF and k <= N keeps T below it unless real NaNs.)
        unsigned T = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned cand = T | (1u << bit);
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < NPL; ++j) cnt += __popcll(__ballot(u[j] < cand));
            if (cnt < k) T = cand;
        }
        // 2. compaction into slot[0..k)
        int base = 0, ties = 0, need = 0;
        {
            int lt = 0;
#pragma unroll
            for (int j = 0; j < NPL; ++j) lt += __popcll(__ballot(u[j] < T));
            need = k - lt;      // how many of the u == T points are taken (>= 1)
        }
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const unsigned long long m_eq = __ballot(u[j] == T);
            const unsigned long long m_lt = __ballot(u[j] < T);
            if ((m_eq | m_lt) == 0) continue;   // wave-uniform
            const int rank = ties + __popcll(m_eq & lt_mask);
            const bool sel = (u[j] < T || (u[j] == T && rank < need)) && j * 64 + lane < N;   // never a padding lane
            const unsigned long long m_sel = __ballot(sel);
            if (sel) slot[base + __popcll(m_sel & lt_mask)] = ((unsigned long long)u[j] << 32) | (unsigned)(j * 64 + lane);
            base += __popcll(m_sel);
            ties += __popcll(m_eq);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0): the wave's own LDS writes have landed (same-wave ordering)
        __builtin_amdgcn_wave_barrier();
        // 3. rank sort of the k unique keys
        const unsigned long long key = lane < k ? slot[lane] : ~0ull;
        const unsigned khi = (unsigned)(key >> 32), klo = (unsigned)key;
        int rk = 0;
        for (int i = 0; i < k; ++i) {
            const unsigned oh = (unsigned)__builtin_amdgcn_readlane((int)khi, i);
            const unsigned ol = (unsigned)__builtin_amdgcn_readlane((int)klo, i);
            rk += (((unsigned long long)oh << 32) | ol) < key ? 1 : 0;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < k) slot[rk] = key;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (lane < k) {
            const unsigned long long e = slot[lane];
            const unsigned e_hi = (unsigned)(e >> 32);
            const int n = (int)(unsigned)e;
            const size_t o = ((size_t)b * G + g) * k + lane;
            if (idx) idx[o] = (int64_t)n;
            if (!GROUP) {
                if (dist) dist[o] = sqrtf(__uint_as_float(e_hi));
            } else {
                const float x = xs[n], y = ys[n], z = zs[n];
                nb[o * 3 + 0] = __fsub_rn(x, qx);
                nb[o * 3 + 1] = __fsub_rn(y, qy);
                nb[o * 3 + 2] = __fsub_rn(z, qz);
                if (nbo) { nbo[o * 3 + 0] = x; nbo[o * 3 + 1] = y; nbo[o * 3 + 2] = z; }
            }
        }
        __builtin_amdgcn_wave_barrier();        // the next query reuses the slots
    }
}

static int knn_check(const void* a, const void* c, int B, int N, int G, int k) {
    if (!a || !c || B < 0 || N < 1 || G < 1 || k < 1) return GM3D_EINVAL;
    if (k > N) return GM3D_EINVAL;
    if (k > 64 || N > 12288) return GM3D_EUNSUPPORTED;
    return GM3D_OK;
}

template <bool GROUP>
static int launch_knn(const float* ref, const float* query, int B, int N, int G, int k, float* dist,
                      int64_t* idx, float* nb, float* nbo, hipStream_t st) {
    const size_t lds = (size_t)N * 3 * sizeof(float);
    if (lds > 64 * 1024) {  // only N > 5461 (validation-size clouds); idempotent, not on the step path
        if (hipFuncSetAttribute((const void*)knn_kernel<GROUP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return GM3D_ELAUNCH;
    }
    const int qpb = knn_qpb(B, G);
    dim3 grid((G + qpb - 1) / qpb, B);
    const int use_select = 1;
    // registers hold the cloud's distances: selection by bisection (see knn_select_kernel).  Measured on MI355X: 2.0-2.5x the
    // insertion kernel at N = 1024, k = 32 (81 -> 32 us for 128 x 64 queries); at N = 2048, k = 16 (twice the registers to scan per
    // bisection step, half the insertions) the insertion kernel is 16 % faster, so longer clouds stay there unless forced (2).
    if ((use_select && N <= 1024) || (use_select == 2 && N <= 2048)) {
        const size_t lds2 = (((size_t)3 * N + 1) & ~(size_t)1) * sizeof(float) + 4 * 64 * sizeof(unsigned long long);
        if (N <= 1024)
            hipLaunchKernelGGL((knn_select_kernel<GROUP, 16>), grid, dim3(KNN_BLOCK), lds2, st, ref, query, N, G, k, dist, idx, nb, nbo, qpb);
        else
            hipLaunchKernelGGL((knn_select_kernel<GROUP, 32>), grid, dim3(KNN_BLOCK), lds2, st, ref, query, N, G, k, dist, idx, nb, nbo, qpb);
        GM3D_CHECK_LAUNCH();
        return GM3D_OK;
    }
    hipLaunchKernelGGL((knn_kernel<GROUP>), grid, dim3(KNN_BLOCK), lds, st, ref, query, N, G, k, dist, idx, nb, nbo, qpb);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

}  // namespace gm3d

extern "C" int gm3d_knn(const float* ref, const float* query, int B, int N, int G, int k, float* dist,
                        int64_t* idx, gm3d_stream_t stream) {
    using namespace gm3d;
    int rc = knn_check(ref, query, B, N, G, k);
    if (rc != GM3D_OK) return rc;
    if (!idx) return GM3D_EINVAL;
    if (B == 0) return GM3D_OK;
    return launch_knn<false>(ref, query, B, N, G, k, dist, idx, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int gm3d_knn_group(const float* xyz, const float* center, int B, int N, int G, int k, int64_t* idx,
                              float* neighborhood, float* neighborhood_org, gm3d_stream_t stream) {
    using namespace gm3d;
    int rc = knn_check(xyz, center, B, N, G, k);
    if (rc != GM3D_OK) return rc;
    if (!neighborhood) return GM3D_EINVAL;
    if (B == 0) return GM3D_OK;
    return launch_knn<true>(xyz, center, B, N, G, k, nullptr, idx, neighborhood, neighborhood_org,
                            (hipStream_t)stream);
}
