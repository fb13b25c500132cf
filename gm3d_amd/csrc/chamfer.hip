// Chamfer nearest-neighbour squared-L2 distances (both directions) and their backward
// for gfx950.
//
// Beneath: extensions.chamfer_dist (ChamferDistanceL2/L1) -- reference call sites
//   Point-MAE_SA3D/models_mae_learn_loss.py:188,407 and models/Point_MAE.py:392-394,426.
// Algorithm contract: SURVEY.md Appendix B / oracle_chamfer_fwd/bwd (first minimum wins,
// d=((dx*dx+dy*dy)+dz*dz) fp32 without FMA, backward g=2(x1-x2) scattered to both clouds).
//
// Design (MI355X): GM3D calls this with P = B*M = 4992 "clouds" of 32 points
// (models_mae_learn_loss.py:397-398), so the hot shape gets its own kernel: ONE
// wavefront per patch, lanes 0-31 own the predicted points and lanes 32-63 the target
// points; each lane scans the other half through v_readlane broadcasts, so the 32x32
// pair tile never touches LDS or HBM.  Its backward is atomic-free and deterministic:
// every lane first forms its own-direction term, then collects the reverse-direction
// terms whose argmin points at it, in ascending order.
// Other shapes take an LDS-tiled general kernel (forward) and the upstream-style
// float-atomic scatter (backward).
#include "common.hpp"

namespace gm3d {

__device__ __forceinline__ float rl(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__global__ __launch_bounds__(256) void chamfer32_fwd_kernel(const float* __restrict__ x1,
                                                            const float* __restrict__ x2, int P,
                                                            float* __restrict__ d1, float* __restrict__ d2,
                                                            int32_t* __restrict__ i1, int32_t* __restrict__ i2) {
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;  // wave-uniform
    const bool first = lane < 32;
    const int l = lane & 31;
    const float* mp = (first ? x1 : x2) + ((size_t)p * 32 + l) * 3;
    const float mx = mp[0], my = mp[1], mz = mp[2];
    float best = 0.f;
    int bi = 0;
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        const float ox = first ? rl(mx, 32 + t) : rl(mx, t);
        const float oy = first ? rl(my, 32 + t) : rl(my, t);
        const float oz = first ? rl(mz, 32 + t) : rl(mz, t);
        const float d = sqdist3(mx, my, mz, ox, oy, oz);
        if (t == 0 || d < best) { best = d; bi = t; }
    }
    const size_t o = (size_t)p * 32 + l;
    if (first) { d1[o] = best; i1[o] = bi; }
    else { d2[o] = best; i2[o] = bi; }
}

__global__ __launch_bounds__(256) void chamfer32_bwd_kernel(const float* __restrict__ x1,
                                                            const float* __restrict__ x2,
                                                            const int32_t* __restrict__ i1,
                                                            const int32_t* __restrict__ i2,
                                                            const float* __restrict__ g1,
                                                            const float* __restrict__ g2, int P,
                                                            float* __restrict__ gx1, float* __restrict__ gx2) {
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    const bool first = lane < 32;
    const int l = lane & 31;
    const size_t o = (size_t)p * 32 + l;
    const float* mp = (first ? x1 : x2) + o * 3;
    const float mx = mp[0], my = mp[1], mz = mp[2];
    const int j = first ? i1[o] : i2[o];
    const float* gp = first ? g1 : g2;
    const float g = gp ? gp[o] : 0.f;
    // own-direction term: t = 2*(mine - other[j])*g
    const int src = first ? 32 + j : j;
    const float ox = __shfl(mx, src), oy = __shfl(my, src), oz = __shfl(mz, src);
    const float tx = __fmul_rn(__fmul_rn(2.0f, __fsub_rn(mx, ox)), g);
    const float ty = __fmul_rn(__fmul_rn(2.0f, __fsub_rn(my, oy)), g);
    const float tz = __fmul_rn(__fmul_rn(2.0f, __fsub_rn(mz, oz)), g);
    float ax = tx, ay = ty, az = tz;
    // reverse-direction terms: every point t of the other half whose argmin is me
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        const int sj = first ? __builtin_amdgcn_readlane(j, 32 + t) : __builtin_amdgcn_readlane(j, t);
        const float sx = first ? rl(tx, 32 + t) : rl(tx, t);
        const float sy = first ? rl(ty, 32 + t) : rl(ty, t);
        const float sz = first ? rl(tz, 32 + t) : rl(tz, t);
        if (sj == l) { ax = __fsub_rn(ax, sx); ay = __fsub_rn(ay, sy); az = __fsub_rn(az, sz); }
    }
    float* out = (first ? gx1 : gx2) + o * 3;
    out[0] = ax; out[1] = ay; out[2] = az;
}

// ------------------------------------------------------------------ the pre-training loss on the masked patches, fused
// forward_loss of the north-star model (models_mae_learn_loss.py:384-412): target = neighborhood[mask] -> (B*M,32,3), pred ->
// (B*M,32,3) cast to fp32, per-point Chamfer (d1 + d2), `matrix` = mean over the 32 points, Chamfer_mean = mean of everything.
// As separate ops that is a gather, a contiguous copy, a cast, the Chamfer kernel, an add and two means (and their five backward
// launches) around 15 us of work; here: ONE wave per masked patch reads its predicted points straight from the decoder head's
// output (any float type, batch-strided view) and its target patch through the id list, keeps the 32x32 pair tile in readlanes like
// chamfer32_fwd_kernel (identical distance arithmetic and argmin rule), and writes the patch mean; a one-block kernel finishes the
// scalar.  The backward writes d(mean)/d(pred) in pred's type from the saved argmins.
template <class T>
__global__ __launch_bounds__(256) void patch_loss_fwd_kernel(const T* __restrict__ pred, size_t pred_bstride,
                                                             const float* __restrict__ target, const long long* __restrict__ ids,
                                                             size_t ids_bstride, int B, int Tn, int M, float* __restrict__ matrix,
                                                             int32_t* __restrict__ i1, int32_t* __restrict__ i2) {
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= B * M) return;  // wave-uniform
    const int b = p / M, m = p - b * M;
    const bool first = lane < 32;
    const int l = lane & 31;
    float mx, my, mz;
    if (first) {
        const T* pp = pred + (size_t)b * pred_bstride + (size_t)m * 96 + l * 3;
        mx = (float)pp[0]; my = (float)pp[1]; mz = (float)pp[2];
    } else {
        const long long id = ids[(size_t)b * ids_bstride + m];
        const float* tp = target + (((size_t)b * Tn + (size_t)id) * 32 + l) * 3;
        mx = tp[0]; my = tp[1]; mz = tp[2];
    }
    float best = 0.f;
    int bi = 0;
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        const float ox = first ? rl(mx, 32 + t) : rl(mx, t);
        const float oy = first ? rl(my, 32 + t) : rl(my, t);
        const float oz = first ? rl(mz, 32 + t) : rl(mz, t);
        const float d = sqdist3(mx, my, mz, ox, oy, oz);
        if (t == 0 || d < best) { best = d; bi = t; }
    }
    const size_t o = (size_t)p * 32 + l;
    if (first) i1[o] = bi; else i2[o] = bi;
    float s = __fadd_rn(best, __shfl(best, lane ^ 32));          // d1[l] + d2[l] on both halves
#pragma unroll
    for (int w = 16; w > 0; w >>= 1) s = __fadd_rn(s, __shfl_xor(s, w));
    if (lane == 0) matrix[p] = s * (1.0f / 32.0f);
}

// mean of n values, one block, fixed order (thread-strided partial sums, then a tree)
__global__ __launch_bounds__(256) void mean_small_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += v[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] / (float)n;
}

template <class T>
__global__ __launch_bounds__(256) void patch_loss_bwd_kernel(const T* __restrict__ pred, size_t pred_bstride,
                                                             const float* __restrict__ target, const long long* __restrict__ ids,
                                                             size_t ids_bstride, const int32_t* __restrict__ i1,
                                                             const int32_t* __restrict__ i2, const float* __restrict__ gmean, int B,
                                                             int Tn, int M, T* __restrict__ dpred, int lead) {
    // lead > 0: dpred is the gradient of the FULL (B, lead + M, 96) prediction whose last M patches entered the loss -- the first
    // `lead` patches of every cloud get zeros (the backward of the [:, -M:] slice in the same launch)
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int LM = lead + M;
    if (q >= B * LM) return;
    const int b = q / LM, m = q - b * LM - lead;
    const bool first = lane < 32;
    const int l = lane & 31;
    if (m < 0) {
        if (first) { T* out = dpred + (size_t)q * 96 + l * 3; out[0] = (T)0.f; out[1] = (T)0.f; out[2] = (T)0.f; }
        return;
    }
    const int p = b * M + m;
    float mx, my, mz;
    if (first) {
        const T* pp = pred + (size_t)b * pred_bstride + (size_t)m * 96 + l * 3;
        mx = (float)pp[0]; my = (float)pp[1]; mz = (float)pp[2];
    } else {
        const long long id = ids[(size_t)b * ids_bstride + m];
        const float* tp = target + (((size_t)b * Tn + (size_t)id) * 32 + l) * 3;
        mx = tp[0]; my = tp[1]; mz = tp[2];
    }
    // d mean / d (d1 | d2)[point] = (g / (B M)) / 32: the two divisions autograd makes for mean() of mean(dim=-1)
    const float g = (gmean[0] / (float)(B * M)) / 32.0f;
    const size_t o = (size_t)p * 32 + l;
    const int j = first ? i1[o] : i2[o];
    const int src = first ? 32 + j : j;
    const float ox = __shfl(mx, src), oy = __shfl(my, src), oz = __shfl(mz, src);
    const float tx = __fmul_rn(__fmul_rn(2.0f, __fsub_rn(mx, ox)), g);
    const float ty = __fmul_rn(__fmul_rn(2.0f, __fsub_rn(my, oy)), g);
    const float tz = __fmul_rn(__fmul_rn(2.0f, __fsub_rn(mz, oz)), g);
    float ax = tx, ay = ty, az = tz;
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        const int sj = first ? __builtin_amdgcn_readlane(j, 32 + t) : __builtin_amdgcn_readlane(j, t);
        const float sx = first ? rl(tx, 32 + t) : rl(tx, t);
        const float sy = first ? rl(ty, 32 + t) : rl(ty, t);
        const float sz = first ? rl(tz, 32 + t) : rl(tz, t);
        if (sj == l) { ax = __fsub_rn(ax, sx); ay = __fsub_rn(ay, sy); az = __fsub_rn(az, sz); }
    }
    if (first) {
        T* out = dpred + (size_t)q * 96 + l * 3;
        out[0] = (T)ax; out[1] = (T)ay; out[2] = (T)az;
    }
}

// General shapes: thread per point of A, B streamed through LDS in tiles.
constexpr int CH_TILE = 1024;
__global__ __launch_bounds__(256) void chamfer_fwd_general_kernel(const float* __restrict__ A,
                                                                  const float* __restrict__ Bp, int n, int m,
                                                                  float* __restrict__ dist,
                                                                  int32_t* __restrict__ idx) {
    __shared__ float sb[CH_TILE * 3];
    const int p = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float* a = A + (size_t)p * n * 3;
    const float* b = Bp + (size_t)p * m * 3;
    const bool live = i < n;
    const float mx = live ? a[(size_t)i * 3 + 0] : 0.f;
    const float my = live ? a[(size_t)i * 3 + 1] : 0.f;
    const float mz = live ? a[(size_t)i * 3 + 2] : 0.f;
    float best = sqdist3(mx, my, mz, b[0], b[1], b[2]);  // j == 0 always seeds, as upstream
    int bi = 0;
    for (int base = 0; base < m; base += CH_TILE) {
        const int cnt = min(CH_TILE, m - base);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt * 3; t += 256) sb[t] = b[(size_t)base * 3 + t];
        __syncthreads();
        for (int t = 0; t < cnt; ++t) {
            const float d = sqdist3(mx, my, mz, sb[t * 3], sb[t * 3 + 1], sb[t * 3 + 2]);
            if (d < best) { best = d; bi = base + t; }
        }
    }
    if (live) { dist[(size_t)p * n + i] = best; idx[(size_t)p * n + i] = bi; }
}

// Zero fill by a kernel of our own, not hipMemsetAsync: a memset node captured into a hipGraph came back unexecuted / misdirected
// on replays once the process had made other copies in between (ROCm 7.2; the gradients of the 8-point patches of Point-M2AE
// turned into 1e34 garbage -- tools/m2ae_graph_diag.py found it), a kernel node replays like every other launch of the step.
__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0.f;
}

// gA[i] += t, gB[idx[i]] -= t with t = 2*(A[i]-B[idx[i]])*g[i]; outputs pre-zeroed.
__global__ void chamfer_bwd_general_kernel(const float* __restrict__ A, const float* __restrict__ Bp,
                                           const int32_t* __restrict__ idx, const float* __restrict__ g,
                                           int n, int m, float* __restrict__ gA, float* __restrict__ gB) {
    const int p = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j = idx[(size_t)p * n + i];
    const float gg = g[(size_t)p * n + i];
    const float* a = A + ((size_t)p * n + i) * 3;
    const float* b = Bp + ((size_t)p * m + j) * 3;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float t = __fmul_rn(__fmul_rn(2.0f, __fsub_rn(a[d], b[d])), gg);
        atomicAdd(gA + ((size_t)p * n + i) * 3 + d, t);
        atomicAdd(gB + ((size_t)p * m + j) * 3 + d, -t);
    }
}

// ---- small equal patches (n == m == 8 or 16: the fine patches of Point-M2AE) -----------------------------------------------------
// SEG lanes per patch (lane l holds point l of both clouds), 64 / SEG patches per wave, partners by width-SEG shuffles.  Forward: both
// directions in one launch, first minimum wins (j == 0 seeds, strict <, ascending j) like every other form.  Backward: no zero fill and
// no atomics -- each lane sums its own slot in the order of the sequential loops (direction 1 over ascending i, then direction 2 over
// ascending j): the general kernel's float atomics made the order a property of the hardware.
template <int SEG>
__global__ __launch_bounds__(256) void chamfer_small_fwd_kernel(const float* __restrict__ A, const float* __restrict__ Bp, int P,
                                                                float* __restrict__ d1, float* __restrict__ d2,
                                                                int32_t* __restrict__ i1, int32_t* __restrict__ i2) {
    const int g = blockIdx.x * (256 / SEG) + threadIdx.x / SEG, l = threadIdx.x % SEG;
    const bool live = g < P;
    const size_t o = ((size_t)(live ? g : 0) * SEG + l) * 3;
    const float ax = A[o], ay = A[o + 1], az = A[o + 2], bx = Bp[o], by = Bp[o + 1], bz = Bp[o + 2];
    float best1 = 0.f, best2 = 0.f;
    int k1 = 0, k2 = 0;
#pragma unroll
    for (int t = 0; t < SEG; ++t) {
        const float tbx = __shfl(bx, t, SEG), tby = __shfl(by, t, SEG), tbz = __shfl(bz, t, SEG);
        const float tax = __shfl(ax, t, SEG), tay = __shfl(ay, t, SEG), taz = __shfl(az, t, SEG);
        const float e1 = sqdist3(ax, ay, az, tbx, tby, tbz), e2 = sqdist3(bx, by, bz, tax, tay, taz);
        if (t == 0 || e1 < best1) { best1 = e1; k1 = t; }
        if (t == 0 || e2 < best2) { best2 = e2; k2 = t; }
    }
    if (live) {
        const size_t q = (size_t)g * SEG + l;
        d1[q] = best1; i1[q] = k1; d2[q] = best2; i2[q] = k2;
    }
}

template <int SEG>
__global__ __launch_bounds__(256) void chamfer_small_bwd_kernel(const float* __restrict__ A, const float* __restrict__ Bp,
                                                                const int32_t* __restrict__ i1, const int32_t* __restrict__ i2,
                                                                const float* __restrict__ g1, const float* __restrict__ g2, int P,
                                                                float* __restrict__ gA, float* __restrict__ gB) {
    const int g = blockIdx.x * (256 / SEG) + threadIdx.x / SEG, l = threadIdx.x % SEG;
    const bool live = g < P;
    const size_t q = (size_t)(live ? g : 0) * SEG + l, o = q * 3;
    const float a[3] = {A[o], A[o + 1], A[o + 2]}, b[3] = {Bp[o], Bp[o + 1], Bp[o + 2]};
    const int k1 = i1[q], k2 = i2[q];
    const float w1 = g1 ? g1[q] : 0.f, w2 = g2 ? g2[q] : 0.f;
    float t1[3], t2[3];       // this lane's term of direction 1 (point l of A against B[k1]) and of direction 2 (point l of B against A[k2])
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        t1[d] = __fmul_rn(__fmul_rn(2.0f, __fsub_rn(a[d], __shfl(b[d], k1, SEG))), w1);
        t2[d] = __fmul_rn(__fmul_rn(2.0f, __fsub_rn(b[d], __shfl(a[d], k2, SEG))), w2);
    }
    float ra[3] = {0.f, 0.f, 0.f}, rb[3] = {0.f, 0.f, 0.f};
    // direction 1, ascending i: gA[i] += t1_i ; gB[k1_i] -= t1_i
#pragma unroll
    for (int d = 0; d < 3; ++d) ra[d] = __fadd_rn(ra[d], t1[d]);
#pragma unroll
    for (int i = 0; i < SEG; ++i) {
        const bool hit = __shfl(k1, i, SEG) == l;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float v = __shfl(t1[d], i, SEG);
            if (hit) rb[d] = __fsub_rn(rb[d], v);
        }
    }
    // direction 2, ascending j: gB[j] += t2_j ; gA[k2_j] -= t2_j
#pragma unroll
    for (int d = 0; d < 3; ++d) rb[d] = __fadd_rn(rb[d], t2[d]);
#pragma unroll
    for (int j = 0; j < SEG; ++j) {
        const bool hit = __shfl(k2, j, SEG) == l;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float v = __shfl(t2[d], j, SEG);
            if (hit) ra[d] = __fsub_rn(ra[d], v);
        }
    }
    if (live) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { gA[o + d] = ra[d]; gB[o + d] = rb[d]; }
    }
}

}  // namespace gm3d

extern "C" int gm3d_chamfer_fwd(const float* xyz1, const float* xyz2, int P, int n, int m, float* dist1,
                                float* dist2, int32_t* idx1, int32_t* idx2, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!xyz1 || !xyz2 || !dist1 || !dist2 || !idx1 || !idx2 || P < 0 || n < 1 || m < 1) return GM3D_EINVAL;
    if (P == 0) return GM3D_OK;
    hipStream_t st = (hipStream_t)stream;
    if (n == 32 && m == 32) {
        hipLaunchKernelGGL(chamfer32_fwd_kernel, dim3((P + 3) / 4), dim3(256), 0, st, xyz1, xyz2, P, dist1, dist2,
                           idx1, idx2);
        GM3D_CHECK_LAUNCH();
        return GM3D_OK;
    }
    if (n == m && (n == 8 || n == 16)) {
        if (n == 8) hipLaunchKernelGGL(chamfer_small_fwd_kernel<8>, dim3((P + 31) / 32), dim3(256), 0, st, xyz1, xyz2, P, dist1, dist2, idx1, idx2);
        else hipLaunchKernelGGL(chamfer_small_fwd_kernel<16>, dim3((P + 15) / 16), dim3(256), 0, st, xyz1, xyz2, P, dist1, dist2, idx1, idx2);
        GM3D_CHECK_LAUNCH();
        return GM3D_OK;
    }
    if (P > 65535) return GM3D_EUNSUPPORTED;
    hipLaunchKernelGGL(chamfer_fwd_general_kernel, dim3((n + 255) / 256, P), dim3(256), 0, st, xyz1, xyz2, n, m,
                       dist1, idx1);
    GM3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(chamfer_fwd_general_kernel, dim3((m + 255) / 256, P), dim3(256), 0, st, xyz2, xyz1, m, n,
                       dist2, idx2);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_chamfer_bwd(const float* xyz1, const float* xyz2, const int32_t* idx1, const int32_t* idx2,
                                const float* grad_dist1, const float* grad_dist2, int P, int n, int m,
                                float* gxyz1, float* gxyz2, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!xyz1 || !xyz2 || !idx1 || !idx2 || !gxyz1 || !gxyz2 || P < 0 || n < 1 || m < 1) return GM3D_EINVAL;
    if (P == 0) return GM3D_OK;
    hipStream_t st = (hipStream_t)stream;
    if (n == 32 && m == 32) {
        hipLaunchKernelGGL(chamfer32_bwd_kernel, dim3((P + 3) / 4), dim3(256), 0, st, xyz1, xyz2, idx1, idx2,
                           grad_dist1, grad_dist2, P, gxyz1, gxyz2);
        GM3D_CHECK_LAUNCH();
        return GM3D_OK;
    }
    if (n == m && (n == 8 || n == 16)) {
        if (n == 8) hipLaunchKernelGGL(chamfer_small_bwd_kernel<8>, dim3((P + 31) / 32), dim3(256), 0, st, xyz1, xyz2, idx1, idx2, grad_dist1,
                                       grad_dist2, P, gxyz1, gxyz2);
        else hipLaunchKernelGGL(chamfer_small_bwd_kernel<16>, dim3((P + 15) / 16), dim3(256), 0, st, xyz1, xyz2, idx1, idx2, grad_dist1,
                                grad_dist2, P, gxyz1, gxyz2);
        GM3D_CHECK_LAUNCH();
        return GM3D_OK;
    }
    if (P > 65535) return GM3D_EUNSUPPORTED;
    const size_t na = (size_t)P * n * 3, nb = (size_t)P * m * 3;
    hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)((na + 255) / 256 < 2048 ? (na + 255) / 256 : 2048)), dim3(256), 0, st, gxyz1, na);
    hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)((nb + 255) / 256 < 2048 ? (nb + 255) / 256 : 2048)), dim3(256), 0, st, gxyz2, nb);
    GM3D_CHECK_LAUNCH();
    if (grad_dist1) {
        hipLaunchKernelGGL(chamfer_bwd_general_kernel, dim3((n + 255) / 256, P), dim3(256), 0, st, xyz1, xyz2, idx1,
                           grad_dist1, n, m, gxyz1, gxyz2);
        GM3D_CHECK_LAUNCH();
    }
    if (grad_dist2) {
        hipLaunchKernelGGL(chamfer_bwd_general_kernel, dim3((m + 255) / 256, P), dim3(256), 0, st, xyz2, xyz1, idx2,
                           grad_dist2, m, n, gxyz2, gxyz1);
        GM3D_CHECK_LAUNCH();
    }
    return GM3D_OK;
}

#define GM3D_CH_DISPATCH(dtype, CALL_BF16, CALL_F32) \
    do { if ((dtype) == GM3D_BF16) { CALL_BF16; } else { CALL_F32; } } while (0)

extern "C" int gm3d_patch_chamfer_loss_fwd(const void* pred, long long pred_bstride, const float* target, const long long* ids,
                                           long long ids_bstride, int B, int T, int M, float* matrix, int32_t* idx1, int32_t* idx2,
                                           float* mean_out, int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!pred || !target || !ids || !matrix || !idx1 || !idx2 || !mean_out || B < 1 || T < 1 || M < 1 || M > T) return GM3D_EINVAL;
    if (pred_bstride < (long long)M * 96 || ids_bstride < M) return GM3D_EINVAL;
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    if ((long long)B * M > 0x7ffffff0LL) return GM3D_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int P = B * M;
    GM3D_CH_DISPATCH(dtype,
                  hipLaunchKernelGGL(patch_loss_fwd_kernel<bf16_t>, dim3((P + 3) / 4), dim3(256), 0, st, (const bf16_t*)pred,
                                     (size_t)pred_bstride, target, ids, (size_t)ids_bstride, B, T, M, matrix, idx1, idx2),
                  hipLaunchKernelGGL(patch_loss_fwd_kernel<float>, dim3((P + 3) / 4), dim3(256), 0, st, (const float*)pred,
                                     (size_t)pred_bstride, target, ids, (size_t)ids_bstride, B, T, M, matrix, idx1, idx2));
    hipLaunchKernelGGL(mean_small_kernel, dim3(1), dim3(256), 0, st, matrix, P, mean_out);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_patch_chamfer_loss_bwd_full(const void* pred, long long pred_bstride, const float* target, const long long* ids,
                                                long long ids_bstride, const int32_t* idx1, const int32_t* idx2, const float* gmean, int B,
                                                int T, int M, int lead, void* dpred, int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!pred || !target || !ids || !idx1 || !idx2 || !gmean || !dpred || B < 1 || T < 1 || M < 1 || M > T || lead < 0) return GM3D_EINVAL;
    if (pred_bstride < (long long)M * 96 || ids_bstride < M) return GM3D_EINVAL;
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int P = B * (lead + M);
    GM3D_CH_DISPATCH(dtype,
                  hipLaunchKernelGGL(patch_loss_bwd_kernel<bf16_t>, dim3((P + 3) / 4), dim3(256), 0, st, (const bf16_t*)pred,
                                     (size_t)pred_bstride, target, ids, (size_t)ids_bstride, idx1, idx2, gmean, B, T, M, (bf16_t*)dpred, lead),
                  hipLaunchKernelGGL(patch_loss_bwd_kernel<float>, dim3((P + 3) / 4), dim3(256), 0, st, (const float*)pred,
                                     (size_t)pred_bstride, target, ids, (size_t)ids_bstride, idx1, idx2, gmean, B, T, M, (float*)dpred, lead));
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_patch_chamfer_loss_bwd(const void* pred, long long pred_bstride, const float* target, const long long* ids,
                                           long long ids_bstride, const int32_t* idx1, const int32_t* idx2, const float* gmean, int B,
                                           int T, int M, void* dpred, int dtype, gm3d_stream_t stream) {
    return gm3d_patch_chamfer_loss_bwd_full(pred, pred_bstride, target, ids, ids_bstride, idx1, idx2, gmean, B, T, M, 0, dpred, dtype, stream);
}
