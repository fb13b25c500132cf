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
