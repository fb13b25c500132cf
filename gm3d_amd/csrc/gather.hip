// Deterministic backward of a row gather with REPEATED indices:  y[b][j][:] = x[b][idx[b][j]][:]  =>  dx[b][s][:] = sum over the
// j with idx[b][j] == s of dy[b][j][:], summed in ascending j.
//
// Beneath: the hierarchical (Point-M2AE) model's token gathers -- a level's token embed reads the k members of every group from the
// previous level's tokens (a token belongs to several groups), and the decoder's token propagation reads the 3 nearest coarse
// tokens (Point-M2AE_SA3D/cfgs/config_Point_M2AE.yaml:57-99; gm3d_amd/point_m2ae.py).  PyTorch's gather backward is a
// scatter-add with colliding float atomics: the order of the additions, hence the last bits of every gradient upstream, changes
// from run to run (tools/m2ae_graph_diag.py: eager vs eager differs), and a captured step cannot be compared with an eager one.
//
// Two kernels.  gm3d_gather_inverse: per batch entry (one workgroup), the inverse lists of idx in CSR form -- counts and slot
// assignment by LDS atomics (any order), then every list sorted by j (lists are short: a source has a handful of readers) -- so
// the result does not depend on the order the atomics ran in.  gm3d_gather_rows_bwd: one half-wave... one thread per 8 channels of
// one source row, walking that row's list in order.
#include "common.hpp"

namespace gm3d {

// off (B, S + 1) int32, list (B, J) int32.  S <= 4096 sources, J <= 16384 references per batch entry.
__global__ __launch_bounds__(256) void gather_inverse_kernel(const long long* __restrict__ idx, int J, int S, int* __restrict__ off,
                                                             int* __restrict__ list) {
    extern __shared__ int gsm[];                 // cnt[S] | start[S + 1]
    int* cnt = gsm;
    int* start = gsm + S;
    const int b = blockIdx.x, tid = threadIdx.x;
    const long long* ib = idx + (size_t)b * J;
    int* lb = list + (size_t)b * J;
    for (int s = tid; s < S; s += 256) cnt[s] = 0;
    __syncthreads();
    for (int j = tid; j < J; j += 256) atomicAdd(&cnt[(int)ib[j]], 1);
    __syncthreads();
    if (tid == 0) {                              // S is a few hundred: a serial prefix sum is a microsecond
        int a = 0;
        for (int s = 0; s < S; ++s) { start[s] = a; a += cnt[s]; }
        start[S] = a;
    }
    __syncthreads();
    for (int s = tid; s <= S; s += 256) off[(size_t)b * (S + 1) + s] = start[s];
    for (int s = tid; s < S; s += 256) cnt[s] = 0;
    __syncthreads();
    for (int j = tid; j < J; j += 256) {
        const int s = (int)ib[j];
        lb[start[s] + atomicAdd(&cnt[s], 1)] = j;
    }
    __syncthreads();
    // order every list by j (insertion sort: lists hold a handful of entries) -> independent of the atomics' order
    for (int s = tid; s < S; s += 256) {
        int* l = lb + start[s];
        const int n = start[s + 1] - start[s];
        for (int i = 1; i < n; ++i) {
            const int v = l[i];
            int k = i - 1;
            while (k >= 0 && l[k] > v) { l[k + 1] = l[k]; --k; }
            l[k + 1] = v;
        }
    }
}

// dx (B, S, C) = sum over each source's list of dy (B, J, C) rows, in list order; T in / T out, fp32 accumulation
template <class T>
__global__ __launch_bounds__(256) void gather_rows_bwd_kernel(const T* __restrict__ dy, const int* __restrict__ off,
                                                              const int* __restrict__ list, T* __restrict__ dx, int J, int S, int C,
                                                              long long total) {
    const int cpr = C >> 3;                      // 8-channel chunks per row
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const long long row = t / cpr;           // b * S + s
        const int c = (int)(t - row * cpr) * 8;
        const int b = (int)(row / S), s = (int)(row - (long long)b * S);
        const int* ob = off + (size_t)b * (S + 1);
        const int lo = ob[s], hi = ob[s + 1];
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int i = lo; i < hi; ++i) {
            const int j = list[(size_t)b * J + i];
            float v[8];
            V8<T>::load(dy + ((size_t)b * J + j) * C + c, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
        V8<T>::store(dx + (size_t)row * C + c, acc);
    }
}

}  // namespace gm3d

extern "C" int gm3d_gather_inverse(const long long* idx, int B, int J, int S, int* off, int* list, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!idx || !off || !list || B < 0 || J < 1 || S < 1) return GM3D_EINVAL;
    if (S > 4096 || J > 16384) return GM3D_EUNSUPPORTED;
    if (B == 0) return GM3D_OK;
    hipLaunchKernelGGL(gather_inverse_kernel, dim3(B), dim3(256), (size_t)(2 * S + 1) * sizeof(int), (hipStream_t)stream, idx, J, S, off,
                       list);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_gather_rows_bwd(const void* dy, const int* off, const int* list, void* dx, int B, int J, int S, int C, int dtype,
                                    gm3d_stream_t stream) {
    using namespace gm3d;
    if (!dy || !off || !list || !dx || B < 0 || J < 1 || S < 1 || C < 8) return GM3D_EINVAL;
    if (C % 8) return GM3D_EUNSUPPORTED;
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    if (B == 0) return GM3D_OK;
    const long long total = (long long)B * S * (C / 8);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GM3D_BF16)
        hipLaunchKernelGGL(gather_rows_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)dy, off, list, (bf16_t*)dx, J, S, C, total);
    else
        hipLaunchKernelGGL(gather_rows_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)dy, off, list, (float*)dx, J, S, C, total);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

// ===================================================================== visible-first token order of a masked level
// The student's pass of the hierarchical encoder keeps only some tokens of a level visible (multi-scale masking:
// gm3d_amd/point_m2ae.py back_project; mask ratio 0.8 of Point-M2AE_SA3D/cfgs/config_Point_M2AE.yaml:57-99).  Block stacks work on
// the visible tokens moved to the front of every cloud (stable order), cut to a static bound Tc where one is known: the rows behind
// the visible count are padding that the attention mask blocks -- whole key / query tiles of it are skipped, and with Tc < T the
// GEMM / LayerNorm passes shrink as well.
//
// gm3d_partition_visible: one workgroup per cloud.  masked (B, T) bytes (non-zero = masked), Tc <= T:
//   perm_c (B, Tc) int32   token at compact slot j: the visible tokens in ascending token order, then masked ones (filler rows)
//   perm_v (B, Tc) int32   the same with -1 in the filler slots
//   inv_v  (B, T)  int32   compact slot of token t if it is visible (and fits below Tc), else -1
//   inv_m  (B, T)  int32   t if token t is masked, else -1   (the rows a merge takes from the un-encoded side)
//   vis_c  (B, Tc) bytes   1 for slots that hold a visible token
//   overflow (1) int32     set to 1 when some cloud has more than Tc visible tokens (never cleared here)
namespace gm3d {

__global__ __launch_bounds__(1024) void partition_visible_kernel(const unsigned char* __restrict__ masked, int T, int Tc,
                                                                 int* __restrict__ perm_c, int* __restrict__ perm_v,
                                                                 int* __restrict__ inv_v, int* __restrict__ inv_m,
                                                                 unsigned char* __restrict__ vis_c, int* __restrict__ overflow) {
    __shared__ int wsum[16];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const bool vis = t < T && masked[(size_t)b * T + t] == 0;
    const unsigned long long bal = __ballot(vis);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[w] = __popcll(bal);
    __syncthreads();
    int base = 0, cnt = 0;
    const int nw = (blockDim.x + 63) >> 6;
    for (int k = 0; k < nw; ++k) {
        const int c = wsum[k];
        if (k < w) base += c;
        cnt += c;
    }
    if (t >= T) return;
    const int nvis_before = base + before;
    const int slot = vis ? nvis_before : cnt + (t - nvis_before);
    if (slot < Tc) {
        perm_c[(size_t)b * Tc + slot] = t;
        perm_v[(size_t)b * Tc + slot] = vis ? t : -1;
        vis_c[(size_t)b * Tc + slot] = vis ? 1 : 0;
    }
    inv_v[(size_t)b * T + t] = (vis && slot < Tc) ? slot : -1;
    inv_m[(size_t)b * T + t] = vis ? -1 : t;
    if (t == 0 && cnt > Tc) *overflow = 1;
}

// out (B, T, C) rows picked per row:  idx[b][t] >= 0 -> a[b][idx[b][t]][:]  (a is (B, Ta, C)),  else alt[b][t][:] (alt (B, T, C)) or
// zeros when alt is NULL.  One thread per 16 bytes (C * sizeof(T) % 16 == 0) or per element otherwise.
template <class V>
__global__ __launch_bounds__(256) void select_rows_kernel(const V* __restrict__ a, const int* __restrict__ idx, const V* __restrict__ alt,
                                                          V* __restrict__ out, int Ta, int T, int cpr, long long total) {
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        const long long row = g / cpr;                // b * T + t
        const int c = (int)(g - row * cpr);
        const int b = (int)(row / T);
        const int i = idx[row];
        V v;
        if (i >= 0) v = a[((size_t)b * Ta + i) * cpr + c];
        else if (alt) v = alt[(size_t)row * cpr + c];
        else v = V{};
        out[(size_t)row * cpr + c] = v;
    }
}

}  // namespace gm3d

extern "C" int gm3d_partition_visible(const unsigned char* masked, int B, int T, int Tc, int* perm_c, int* perm_v, int* inv_v, int* inv_m,
                                      unsigned char* vis_c, int* overflow, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!masked || !perm_c || !perm_v || !inv_v || !inv_m || !vis_c || !overflow || B < 0 || T < 1 || Tc < 1 || Tc > T) return GM3D_EINVAL;
    if (T > 1024) return GM3D_EUNSUPPORTED;
    if (B == 0) return GM3D_OK;
    const int threads = (T + 63) / 64 * 64;
    hipLaunchKernelGGL(partition_visible_kernel, dim3(B), dim3(threads), 0, (hipStream_t)stream, masked, T, Tc, perm_c, perm_v, inv_v, inv_m,
                       vis_c, overflow);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_select_rows(const void* a, const int* idx, const void* alt, void* out, int B, int Ta, int T, int C, int elem_bytes,
                                gm3d_stream_t stream) {
    using namespace gm3d;
    if (!a || !idx || !out || B < 0 || Ta < 1 || T < 1 || C < 1 || (elem_bytes != 2 && elem_bytes != 4)) return GM3D_EINVAL;
    if (B == 0) return GM3D_OK;
    hipStream_t st = (hipStream_t)stream;
    const long long row_bytes = (long long)C * elem_bytes;
    const bool wide = row_bytes % 16 == 0 && ((uintptr_t)a % 16 == 0) && ((uintptr_t)out % 16 == 0) && (!alt || (uintptr_t)alt % 16 == 0);
    const int cpr = wide ? (int)(row_bytes / 16) : C;
    const long long total = (long long)B * T * cpr;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (wide)
        hipLaunchKernelGGL(select_rows_kernel<uint4>, dim3(grid), dim3(256), 0, st, (const uint4*)a, idx, (const uint4*)alt, (uint4*)out, Ta, T,
                           cpr, total);
    else if (elem_bytes == 4)
        hipLaunchKernelGGL(select_rows_kernel<unsigned>, dim3(grid), dim3(256), 0, st, (const unsigned*)a, idx, (const unsigned*)alt,
                           (unsigned*)out, Ta, T, cpr, total);
    else
        hipLaunchKernelGGL(select_rows_kernel<unsigned short>, dim3(grid), dim3(256), 0, st, (const unsigned short*)a, idx,
                           (const unsigned short*)alt, (unsigned short*)out, Ta, T, cpr, total);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

// ===================================================================== token propagation (3-NN interpolation) of the hierarchical decoder
// Up-block of Point-M2AE's H_Decoder (PointNet++ feature propagation; hyper-parameters Point-M2AE_SA3D/cfgs/config_Point_M2AE.yaml:88-99,
// gm3d_amd/point_m2ae.py TokenPropagation): every fine token takes the inverse-squared-distance weighted mean of its three nearest coarse
// tokens, concatenated behind its own features:
//   out[b][n][0:C1]      = fine[b][n][:]
//   out[b][n][C1:C1+C2]  = (w[b][n][0] * coarse[b][idx[b][n][0]] + w[..][1] * coarse[..idx 1]) + w[..][2] * coarse[..idx 2]   (fp32, one rounding)
// Backward of the interpolated half: gm3d_gather_inverse over idx (B, 3N) + gm3d_gather_rows_bwd_w (every coarse token sums its readers'
// weighted gradients in ascending reader order: deterministic).
namespace gm3d {

template <class T>
__global__ __launch_bounds__(256) void interp3_fwd_kernel(const T* __restrict__ coarse, const long long* __restrict__ idx,
                                                          const float* __restrict__ w, const T* __restrict__ fine, T* __restrict__ out,
                                                          int N, int S, int C1, int C2, long long total) {
    const int cpr1 = C1 >> 3, cpr = (C1 + C2) >> 3;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        const long long row = g / cpr;                    // b * N + n
        const int ch = (int)(g - row * cpr);
        float v[8];
        if (ch < cpr1) {
            V8<T>::load(fine + (size_t)row * C1 + ch * 8, v);
        } else {
            const int b = (int)(row / N), c = (ch - cpr1) * 8;
            const long long* ip = idx + (size_t)row * 3;
            const float* wp = w + (size_t)row * 3;
            float a0[8], a1[8], a2[8];
            V8<T>::load(coarse + ((size_t)b * S + (int)ip[0]) * C2 + c, a0);
            V8<T>::load(coarse + ((size_t)b * S + (int)ip[1]) * C2 + c, a1);
            V8<T>::load(coarse + ((size_t)b * S + (int)ip[2]) * C2 + c, a2);
            const float w0 = wp[0], w1 = wp[1], w2 = wp[2];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (w0 * a0[e] + w1 * a1[e]) + w2 * a2[e];
        }
        V8<T>::store(out + (size_t)row * (C1 + C2) + ch * 8, v);
    }
}

// dx (B, S, C) = sum over each source's list (ascending reference j) of w[b][j] * dy[b][j / rpr][col0 : col0 + C]; dy rows have pitch ldy
template <class T>
__global__ __launch_bounds__(256) void gather_rows_bwd_w_kernel(const T* __restrict__ dy, int ldy, int col0, int rpr,
                                                                const float* __restrict__ w, const int* __restrict__ off,
                                                                const int* __restrict__ list, T* __restrict__ dx, int J, int S, int C,
                                                                long long total) {
    const int cpr = C >> 3;
    const int rows_b = J / rpr;                           // dy rows per batch entry
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const long long row = t / cpr;                    // b * S + s
        const int c = (int)(t - row * cpr) * 8;
        const int b = (int)(row / S), s = (int)(row - (long long)b * S);
        const int* ob = off + (size_t)b * (S + 1);
        const int lo = ob[s], hi = ob[s + 1];
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int i = lo; i < hi; ++i) {
            const int j = list[(size_t)b * J + i];
            const float wj = w[(size_t)b * J + j];
            float v[8];
            V8<T>::load(dy + ((size_t)b * rows_b + j / rpr) * ldy + col0 + c, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += wj * v[e];
        }
        V8<T>::store(dx + (size_t)row * C + c, acc);
    }
}

// multi-scale masking, one level down: a group of level l-1 is visible iff at least one VISIBLE group of level l lists it as a member.
// masked_c (B, Gc) bytes (non-zero = masked) of the coarser level, member (B, Gc, k) int64 into the finer level's Gf groups
// -> masked_f (B, Gf) bytes.  One workgroup per cloud, flags in LDS.
__global__ __launch_bounds__(256) void back_project_kernel(const unsigned char* __restrict__ masked_c, const long long* __restrict__ member,
                                                           int Gc, int k, int Gf, unsigned char* __restrict__ masked_f) {
    extern __shared__ int seen[];
    const int b = blockIdx.x;
    for (int g = threadIdx.x; g < Gf; g += 256) seen[g] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < Gc * k; i += 256)
        if (masked_c[(size_t)b * Gc + i / k] == 0) seen[(int)member[(size_t)b * Gc * k + i]] = 1;      // benign race: every writer stores 1
    __syncthreads();
    for (int g = threadIdx.x; g < Gf; g += 256) masked_f[(size_t)b * Gf + g] = seen[g] ? 0 : 1;
}

}  // namespace gm3d

extern "C" int gm3d_interp3_fwd(const void* coarse, const long long* idx, const float* w, const void* fine, void* out, int B, int N, int S,
                                int C1, int C2, int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!coarse || !idx || !w || !out || B < 0 || N < 1 || S < 1 || C1 < 0 || C2 < 8 || (C1 > 0 && !fine)) return GM3D_EINVAL;
    if (C1 % 8 || C2 % 8) return GM3D_EUNSUPPORTED;
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    if (B == 0) return GM3D_OK;
    const long long total = (long long)B * N * ((C1 + C2) / 8);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GM3D_BF16)
        hipLaunchKernelGGL(interp3_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)coarse, idx, w, (const bf16_t*)fine,
                           (bf16_t*)out, N, S, C1, C2, total);
    else
        hipLaunchKernelGGL(interp3_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)coarse, idx, w, (const float*)fine,
                           (float*)out, N, S, C1, C2, total);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_gather_rows_bwd_w(const void* dy, int ldy, int col0, int refs_per_row, const float* w, const int* off, const int* list,
                                      void* dx, int B, int J, int S, int C, int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!dy || !w || !off || !list || !dx || B < 0 || J < 1 || S < 1 || C < 8 || refs_per_row < 1 || J % refs_per_row || col0 < 0 ||
        ldy < col0 + C)
        return GM3D_EINVAL;
    if (C % 8 || ldy % 8 || col0 % 8) return GM3D_EUNSUPPORTED;
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    if (B == 0) return GM3D_OK;
    const long long total = (long long)B * S * (C / 8);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GM3D_BF16)
        hipLaunchKernelGGL(gather_rows_bwd_w_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)dy, ldy, col0, refs_per_row, w, off,
                           list, (bf16_t*)dx, J, S, C, total);
    else
        hipLaunchKernelGGL(gather_rows_bwd_w_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)dy, ldy, col0, refs_per_row, w, off,
                           list, (float*)dx, J, S, C, total);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_back_project(const unsigned char* masked_c, const long long* member, int B, int Gc, int k, int Gf,
                                 unsigned char* masked_f, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!masked_c || !member || !masked_f || B < 0 || Gc < 1 || k < 1 || Gf < 1) return GM3D_EINVAL;
    if (Gf > 8192) return GM3D_EUNSUPPORTED;
    if (B == 0) return GM3D_OK;
    hipLaunchKernelGGL(back_project_kernel, dim3(B), dim3(256), (size_t)Gf * sizeof(int), (hipStream_t)stream, masked_c, member, Gc, k, Gf,
                       masked_f);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

// ===================================================================== row-wise choice by a per-token flag, gather by int64 lists
// out[b][t][:] = flag[b][t] ? alt[b][t][:] : a[b][t][:]   (flag (B,T) bytes; alt NULL: zeros; alt_bcast != 0: alt is ONE row (C) for every
// token -- the mask token).  The hierarchical model's "a masked token keeps / takes ..." sites (gm3d_amd/point_m2ae.py: the un-encoded
// embedding handed to the next level, the mask token of the decoder, the zeroed invisible rows) and their backward (flag inverted by
// `invert`): one launch instead of bitwise_not + where (+ zeros).
namespace gm3d {

template <class V>
__global__ __launch_bounds__(256) void where_rows_kernel(const unsigned char* __restrict__ flag, int invert, const V* __restrict__ a,
                                                         const V* __restrict__ alt, int alt_bcast, V* __restrict__ out, int cpr, long long total) {
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        const long long row = g / cpr;
        const int c = (int)(g - row * cpr);
        const bool f = (flag[row] != 0) != (invert != 0);
        V v;
        if (!f) v = a ? a[g] : V{};
        else if (alt) v = alt[alt_bcast ? (size_t)c : (size_t)g];
        else v = V{};
        out[g] = v;
    }
}

// out (B,J,C) = a[b][idx[b][j]][:] for int64 idx (B,J) that may repeat a source row (the member lists of the grouping)
template <class V>
__global__ __launch_bounds__(256) void take_rows_kernel(const V* __restrict__ a, const long long* __restrict__ idx, V* __restrict__ out, int S,
                                                        int J, int cpr, long long total) {
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        const long long row = g / cpr;                // b * J + j
        const int c = (int)(g - row * cpr);
        const int b = (int)(row / J);
        out[g] = a[((size_t)b * S + (int)idx[row]) * cpr + c];
    }
}

}  // namespace gm3d

extern "C" int gm3d_where_rows(const unsigned char* flag, int invert, const void* a, const void* alt, int alt_bcast, void* out, long long rows,
                               int C, int elem_bytes, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!flag || !out || rows < 0 || C < 1 || (elem_bytes != 2 && elem_bytes != 4) || (alt_bcast && !alt)) return GM3D_EINVAL;
    if (rows == 0) return GM3D_OK;
    hipStream_t st = (hipStream_t)stream;
    const long long row_bytes = (long long)C * elem_bytes;
    const bool wide = row_bytes % 16 == 0 && ((uintptr_t)out % 16 == 0) && (!a || (uintptr_t)a % 16 == 0) && (!alt || (uintptr_t)alt % 16 == 0);
    const int cpr = wide ? (int)(row_bytes / 16) : C;
    const long long total = rows * cpr;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (wide)
        hipLaunchKernelGGL(where_rows_kernel<uint4>, dim3(grid), dim3(256), 0, st, flag, invert, (const uint4*)a, (const uint4*)alt, alt_bcast,
                           (uint4*)out, cpr, total);
    else if (elem_bytes == 4)
        hipLaunchKernelGGL(where_rows_kernel<unsigned>, dim3(grid), dim3(256), 0, st, flag, invert, (const unsigned*)a, (const unsigned*)alt,
                           alt_bcast, (unsigned*)out, cpr, total);
    else
        hipLaunchKernelGGL(where_rows_kernel<unsigned short>, dim3(grid), dim3(256), 0, st, flag, invert, (const unsigned short*)a,
                           (const unsigned short*)alt, alt_bcast, (unsigned short*)out, cpr, total);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_take_rows(const void* a, const long long* idx, void* out, int B, int S, int J, int C, int elem_bytes,
                              gm3d_stream_t stream) {
    using namespace gm3d;
    if (!a || !idx || !out || B < 0 || S < 1 || J < 1 || C < 1 || (elem_bytes != 2 && elem_bytes != 4)) return GM3D_EINVAL;
    if (B == 0) return GM3D_OK;
    hipStream_t st = (hipStream_t)stream;
    const long long row_bytes = (long long)C * elem_bytes;
    const bool wide = row_bytes % 16 == 0 && ((uintptr_t)a % 16 == 0) && ((uintptr_t)out % 16 == 0);
    const int cpr = wide ? (int)(row_bytes / 16) : C;
    const long long total = (long long)B * J * cpr;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (wide)
        hipLaunchKernelGGL(take_rows_kernel<uint4>, dim3(grid), dim3(256), 0, st, (const uint4*)a, idx, (uint4*)out, S, J, cpr, total);
    else if (elem_bytes == 4)
        hipLaunchKernelGGL(take_rows_kernel<unsigned>, dim3(grid), dim3(256), 0, st, (const unsigned*)a, idx, (unsigned*)out, S, J, cpr, total);
    else
        hipLaunchKernelGGL(take_rows_kernel<unsigned short>, dim3(grid), dim3(256), 0, st, (const unsigned short*)a, idx, (unsigned short*)out,
                           S, J, cpr, total);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}
