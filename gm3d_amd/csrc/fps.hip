// Farthest point sampling + fused centre gather, and the standalone gather_operation
// forward/backward, for gfx950.
//
// Beneath: pointnet2_utils.furthest_point_sample / gather_operation
//   (reference call sites Point-MAE_SA3D/models_mae_learn_loss.py:931-932,
//    utils/miscc.py:18-19).  Algorithm contract: SURVEY.md Appendix B / oracle_fps.
//
// Design (MI355X): one workgroup per cloud; the cloud and its running-min array live
// in registers for the whole launch (single HBM read, 12 B/point); every one of the
// npoint-1 dependent steps is  {register scan -> DPP wave max on a packed 64-bit key ->
// one LDS slot per wave -> one barrier}.  No global-memory round trip inside the loop.
#include "common.hpp"

namespace gm3d {

struct __attribute__((aligned(16))) FpsSlot {
    unsigned long long key;
    float x, y, z;
    float pad[3];
};

// key = (float bits of running-min distance) << 32 | ~index : the unsigned maximum is
// the largest distance, ties -> lowest index.  key 0 = "no candidate" (every point
// skipped), which decodes to index 0 like upstream's besti=0/best=-1 start.
template <int T, int PPT>
__global__ __launch_bounds__(T) void fps_kernel(const float* __restrict__ xyz, int N, int npoint,
                                                int32_t* __restrict__ idx_out,
                                                float* __restrict__ centers) {
    constexpr int NW = T / GM3D_WAVE;
    __shared__ FpsSlot slots[2][NW];

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const float* p = xyz + (size_t)b * N * 3;
    int32_t* out = idx_out + (size_t)b * npoint;
    float* cen = centers ? centers + (size_t)b * npoint * 3 : nullptr;

    float px[PPT], py[PPT], pz[PPT], tmin[PPT];
    bool ok[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int k = tid + i * T;
        const bool in = k < N;
        px[i] = in ? p[(size_t)k * 3 + 0] : 0.f;
        py[i] = in ? p[(size_t)k * 3 + 1] : 0.f;
        pz[i] = in ? p[(size_t)k * 3 + 2] : 0.f;
        const float mag = __fadd_rn(__fadd_rn(__fmul_rn(px[i], px[i]), __fmul_rn(py[i], py[i])),
                                    __fmul_rn(pz[i], pz[i]));
        ok[i] = in && (mag > 1e-3f);
        tmin[i] = 1e10f;
    }

    float ox = p[0], oy = p[1], oz = p[2];
    if (tid == 0) {
        out[0] = 0;
        if (cen) { cen[0] = ox; cen[1] = oy; cen[2] = oz; }
    }

    for (int j = 1; j < npoint; ++j) {
        float best = -1.0f, bx = px[0], by = py[0], bz = pz[0];
        int bk = tid;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            if (ok[i]) {
                const float d = sqdist3(px[i], py[i], pz[i], ox, oy, oz);
                const float t = d < tmin[i] ? d : tmin[i];
                tmin[i] = t;
                if (t > best) { best = t; bk = tid + i * T; bx = px[i]; by = py[i]; bz = pz[i]; }
            }
        }
        const unsigned long long key =
            best < 0.f ? 0ull : (((unsigned long long)__float_as_uint(best) << 32) | (unsigned)(~(unsigned)bk));
        const unsigned long long wkey = wave_max_u64(key);
        const bool owner = wkey != 0ull ? (key == wkey) : (lane == 0);
        FpsSlot* s = &slots[j & 1][wave];
        if (owner) {
            s->key = wkey;
            // all-skipped wave: lane 0 publishes its slot-0 point (thread 0 owns point 0)
            s->x = wkey != 0ull ? bx : px[0];
            s->y = wkey != 0ull ? by : py[0];
            s->z = wkey != 0ull ? bz : pz[0];
        }
        __syncthreads();
        unsigned long long fk = slots[j & 1][0].key;
        int fw = 0;
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const unsigned long long kw = slots[j & 1][w].key;
            if (kw > fk) { fk = kw; fw = w; }
        }
        ox = slots[j & 1][fw].x; oy = slots[j & 1][fw].y; oz = slots[j & 1][fw].z;
        const int old = fk != 0ull ? (int)(~(unsigned)fk) : 0;
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
    if (N <= 1024) return launch_fps<256, 4>(xyz, B, N, npoint, idx, centers, st);
    if (N <= 2048) return launch_fps<256, 8>(xyz, B, N, npoint, idx, centers, st);
    if (N <= 4096) return launch_fps<512, 8>(xyz, B, N, npoint, idx, centers, st);
    if (N <= 8192) return launch_fps<1024, 8>(xyz, B, N, npoint, idx, centers, st);
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
