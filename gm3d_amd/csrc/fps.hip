// Farthest point sampling + fused centre gather, and the standalone gather_operation
// forward/backward, for gfx950.
//
// Beneath: pointnet2_utils.furthest_point_sample / gather_operation
//   (reference call sites Point-MAE_SA3D/models_mae_learn_loss.py:931-932,
//    utils/miscc.py:18-19).  Algorithm contract: SURVEY.md Appendix B / oracle_fps.
//
// Design (MI355X): one workgroup per cloud; the cloud and its running-min array live
// in registers for the whole launch (single HBM read, 12 B/point); every one of the
// npoint-1 dependent steps is  {packed-fp32 register scan -> DPP wave max/argmin ->
// one LDS key per wave -> one barrier -> 16-lane row reduction -> the winner's coordinates,
// 12 bytes re-read from the cloud in global memory (this workgroup loaded it a moment ago: an L2 hit)}.
// The kernel keeps NO long-lived LDS state: an earlier version held an LDS copy of the cloud for
// the winner's coordinates (2.6 us faster at 1024 -> 64) and gave a few clouds a wrong sample in
// 3.5 % of launches whenever ANOTHER PROCESS was running kernels on the same GPU -- as the two-rank
// tests do.  tools/fps_shared_gpu_probe.py isolated it: every variant that kept the LDS copy failed
// (DPP or shuffles, one wave or four, one barrier or two), the variant below did not in 111,898
// launches; with the load coming from a second stream of the SAME process nothing ever failed.
// Workgroups that live for tens of microseconds are the ones a process switch saves and restores.
#include "common.hpp"

namespace gm3d {

typedef float float2v __attribute__((ext_vector_type(2)));

// Wave-wide reductions by DPP (result uniform in every lane).  Rows that a step does not write keep `old` = v.
// The cross-lane move is evaluated ONCE per step into a temporary, outside any conditional: a DPP move under a
// divergent EXEC mask reads its neighbours as invalid lanes.
template <int CTRL, int RM> __device__ __forceinline__ float dpp_f32(float v) {
    return __uint_as_float(dpp_u32<CTRL, RM>(__float_as_uint(v), __float_as_uint(v)));
}
template <int CTRL, int RM> __device__ __forceinline__ int dpp_i32(int v) {
    return (int)dpp_u32<CTRL, RM>((unsigned)v, (unsigned)v);
}
__device__ __forceinline__ float wave_max_f32(float v) {
    { const float o = dpp_f32<0xB1, 0xF>(v); v = fmaxf(v, o); }
    { const float o = dpp_f32<0x4E, 0xF>(v); v = fmaxf(v, o); }
    { const float o = dpp_f32<0x141, 0xF>(v); v = fmaxf(v, o); }
    { const float o = dpp_f32<0x140, 0xF>(v); v = fmaxf(v, o); }
    { const float o = dpp_f32<0x142, 0xA>(v); v = fmaxf(v, o); }
    { const float o = dpp_f32<0x143, 0xC>(v); v = fmaxf(v, o); }
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_min_i32(int v) {
    { const int o = dpp_i32<0xB1, 0xF>(v); v = min(v, o); }
    { const int o = dpp_i32<0x4E, 0xF>(v); v = min(v, o); }
    { const int o = dpp_i32<0x141, 0xF>(v); v = min(v, o); }
    { const int o = dpp_i32<0x140, 0xF>(v); v = min(v, o); }
    { const int o = dpp_i32<0x142, 0xA>(v); v = min(v, o); }
    { const int o = dpp_i32<0x143, 0xC>(v); v = min(v, o); }
    return __builtin_amdgcn_readlane(v, 63);
}
// max of a 64-bit key over each 16-lane row (4 DPP steps), every lane of the row gets the result
__device__ __forceinline__ unsigned long long row_max_u64(unsigned long long k) {
#define GM3D_STEP(CTRL)                                                             \
    {                                                                               \
        unsigned lo = (unsigned)k, hi = (unsigned)(k >> 32);                        \
        unsigned olo = dpp_u32<CTRL, 0xF>(lo, lo), ohi = dpp_u32<CTRL, 0xF>(hi, hi);\
        unsigned long long o = ((unsigned long long)ohi << 32) | olo;               \
        k = o > k ? o : k;                                                          \
    }
    GM3D_STEP(0xB1) GM3D_STEP(0x4E) GM3D_STEP(0x141) GM3D_STEP(0x140)
#undef GM3D_STEP
    return k;
}

// key = (float bits of running-min distance) << 32 | ~index : the unsigned maximum is the largest distance, ties ->
// lowest index.  key 0 = "no candidate" (every point skipped), which decodes to index 0 like upstream's
// besti=0/best=-1 start.
//
// One step of the dependent chain, per wave (the launch is bound by ONE CU's VALU issue rate -- a cloud is one
// workgroup -- so the step is written for instruction count):
//   scan    PPT points as packed pairs: v_pk_add/v_pk_mul_f32 for ((dx*dx + dy*dy) + dz*dz) (separately rounded,
//           identical to the scalar contract), v_min + compare/select of (best, index) only -- the winner's
//           coordinates are NOT carried through the selects, they are re-read from the cloud (L2);
//           skipped points (|p|^2 <= 1e-3) carry tmin = -1 and never win, so the scan has no branches;
//   reduce  wave max of `best` (f32 DPP), then wave min of the index among the lanes that hold it;
//   publish one 64-bit key per wave, ONE barrier (double-buffered slots), lanes 0..NW-1 of every row re-read the
//           slots and finish with a 4-step row reduction.
template <int T, int PPT>
__global__ __launch_bounds__(T) void fps_kernel(const float* __restrict__ xyz, int N, int npoint,
                                                int32_t* __restrict__ idx_out,
                                                float* __restrict__ centers) {
    constexpr int NW = T / GM3D_WAVE;
    static_assert(NW <= 16 && PPT % 2 == 0, "one 16-lane row holds the per-wave keys; points are scanned in pairs");
    __shared__ unsigned long long slots[2][16];

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const float* p = xyz + (size_t)b * N * 3;
    int32_t* out = idx_out + (size_t)b * npoint;
    float* cen = centers ? centers + (size_t)b * npoint * 3 : nullptr;

    float2v px[PPT / 2], py[PPT / 2], pz[PPT / 2];
    float tmin[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int k = tid + i * T;
        const bool in = k < N;
        const float x = in ? p[(size_t)k * 3 + 0] : 0.f, y = in ? p[(size_t)k * 3 + 1] : 0.f, z = in ? p[(size_t)k * 3 + 2] : 0.f;
        px[i >> 1][i & 1] = x; py[i >> 1][i & 1] = y; pz[i >> 1][i & 1] = z;
        const float mag = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
        tmin[i] = (in && mag > 1e-3f) ? 1e10f : -1.0f;
    }
    if (tid < 32) slots[tid >> 4][tid & 15] = 0ull;
    __syncthreads();

    float ox = p[0], oy = p[1], oz = p[2];
    if (tid == 0) {
        out[0] = 0;
        if (cen) { cen[0] = ox; cen[1] = oy; cen[2] = oz; }
    }

    for (int j = 1; j < npoint; ++j) {
        float best = -1.0f;
        int bk = tid;
        const float2v o2x = {ox, ox}, o2y = {oy, oy}, o2z = {oz, oz};
#pragma unroll
        for (int h = 0; h < PPT / 2; ++h) {
            const float2v dx = px[h] - o2x, dy = py[h] - o2y, dz = pz[h] - o2z;
            const float2v d = (dx * dx + dy * dy) + dz * dz;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int i = 2 * h + e;
                const float t = __builtin_fminf(d[e], tmin[i]);
                tmin[i] = t;
                if (t > best) { best = t; bk = tid + i * T; }
            }
        }
        const float m = wave_max_f32(best);
        const int mi = wave_min_i32(best == m ? bk : 0x7fffffff);
        if (lane == 0)
            slots[j & 1][wave] = m < 0.f ? 0ull : (((unsigned long long)__float_as_uint(m) << 32) | (unsigned)(~(unsigned)mi));
        __syncthreads();
        unsigned long long fk = row_max_u64(slots[j & 1][lane & 15]);     // slots >= NW stay 0
        const unsigned flo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)fk);
        const int old = (__builtin_amdgcn_readfirstlane((int)(unsigned)(fk >> 32)) | (int)flo) != 0 ? (int)(~flo) : 0;
        ox = p[(size_t)old * 3 + 0]; oy = p[(size_t)old * 3 + 1]; oz = p[(size_t)old * 3 + 2];
        if (tid == 0) {
            out[j] = old;
            if (cen) { cen[(size_t)j * 3 + 0] = ox; cen[(size_t)j * 3 + 1] = oy; cen[(size_t)j * 3 + 2] = oz; }
        }
    }
}

__global__ void gather_points_kernel(const float* __restrict__ feat, const int32_t* __restrict__ idx,
                                     int C, int N, int M, float* __restrict__ out, size_t total) {
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(o % M);
        const size_t bc = o / M;
        const size_t b = bc / C;
        out[o] = feat[bc * N + idx[b * M + j]];
    }
}

// One thread per (b,c,n): sums grad_out over the j whose idx hits n, ascending j.
// O(N*M) compares per row but atomic-free and deterministic; M is 64..1200 on this path.
__global__ void gather_points_grad_kernel(const float* __restrict__ gout, const int32_t* __restrict__ idx,
                                          int C, int N, int M, float* __restrict__ gfeat) {
    extern __shared__ int32_t sidx[];
    const int b = blockIdx.y;
    for (int j = threadIdx.x; j < M; j += blockDim.x) sidx[j] = idx[(size_t)b * M + j];
    __syncthreads();
    const size_t rows = (size_t)C * N;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < rows; o += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(o % N);
        const size_t c = o / N;
        const float* g = gout + ((size_t)b * C + c) * M;
        float acc = 0.f;
        for (int j = 0; j < M; ++j)
            if (sidx[j] == n) acc += g[j];
        gfeat[(size_t)b * rows + o] = acc;
    }
}

template <int T, int PPT>
static int launch_fps(const float* xyz, int B, int N, int npoint, int32_t* idx, float* centers, hipStream_t st) {
    hipLaunchKernelGGL((fps_kernel<T, PPT>), dim3(B), dim3(T), 0, st, xyz, N, npoint, idx, centers);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

}  // namespace gm3d

extern "C" int gm3d_fps(const float* xyz, int B, int N, int npoint, int32_t* idx, float* centers,
                        gm3d_stream_t stream) {
    using namespace gm3d;
    if (!xyz || !idx || B < 0 || N < 1 || npoint < 1) return GM3D_EINVAL;
    if (N > 16384) return GM3D_EUNSUPPORTED;
    if (B == 0) return GM3D_OK;
    hipStream_t st = (hipStream_t)stream;
    if (N <= 256) return launch_fps<64, 4>(xyz, B, N, npoint, idx, centers, st);
    // thread / points-per-thread shapes measured on MI355X (tools/fps_bench.py): the step is bound by one CU's VALU issue
    // plus the per-wave reduction chain, so large clouds prefer fewer, fatter waves
    if (N <= 1024) return launch_fps<256, 4>(xyz, B, N, npoint, idx, centers, st);
    if (N <= 2048) return launch_fps<256, 8>(xyz, B, N, npoint, idx, centers, st);
    if (N <= 4096) return launch_fps<512, 8>(xyz, B, N, npoint, idx, centers, st);
    if (N <= 8192) return launch_fps<512, 16>(xyz, B, N, npoint, idx, centers, st);
    return launch_fps<1024, 16>(xyz, B, N, npoint, idx, centers, st);
}

extern "C" int gm3d_gather_points(const float* feat, const int32_t* idx, int B, int C, int N, int M,
                                  float* out, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!feat || !idx || !out || B < 0 || C < 1 || N < 1 || M < 1) return GM3D_EINVAL;
    if (B == 0) return GM3D_OK;
    const size_t total = (size_t)B * C * M;
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(gather_points_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, feat, idx, C, N, M, out, total);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_gather_points_grad(const float* grad_out, const int32_t* idx, int B, int C, int N, int M,
                                       float* grad_feat, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!grad_out || !idx || !grad_feat || B < 0 || C < 1 || N < 1 || M < 1) return GM3D_EINVAL;
    if (M > 16384) return GM3D_EUNSUPPORTED;
    if (B == 0) return GM3D_OK;
    const size_t rows = (size_t)C * N;
    const int gx = (int)((rows + 255) / 256 < 1024 ? (rows + 255) / 256 : 1024);
    hipLaunchKernelGGL(gather_points_grad_kernel, dim3(gx, B), dim3(256), (size_t)M * sizeof(int32_t),
                       (hipStream_t)stream, grad_out, idx, C, N, M, grad_feat);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}
