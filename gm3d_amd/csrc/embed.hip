// Streaming kernels of the mini-PointNet token embed (forward and backward) for gfx950.
//
// Beneath: Encoder.forward, Point-MAE_SA3D/models_mae_learn_loss.py:868-899
//   first_conv  = Conv1d(3,128) -> BatchNorm1d(128) -> ReLU -> Conv1d(128,256)
//   global max over the k=32 points of a group, concat [global | local] (512)
//   second_conv = Conv1d(512,512) -> BatchNorm1d(512) -> ReLU -> Conv1d(512,384);  max over k
// on rows = B*G*k = 262,144 points per step (twice: EMA teacher in eval mode, student in train mode).
//
// The three wide 1x1 convolutions stay plain GEMMs (hipBLASLt); everything between them is here, each as ONE
// pass over a (group, k, C) activation with 16-byte accesses:
//   * layer 1 (K=3) + BatchNorm + ReLU are evaluated on the fly from xyz -- the (rows,128) pre-activation never
//     exists, and its batch statistics are analytic in the 3x3 input moments (moments3_kernel);
//   * concat([global, local]) @ W is local @ W_l + (global @ W_g)[group]: the per-group term `t` is broadcast
//     inside the BatchNorm kernels instead of being materialised over the rows;
//   * BatchNorm batch statistics, normalise+ReLU, both max-pools (with argmax for the backward scatter), the
//     BatchNorm backward reductions/apply and the per-group gradient sums are single passes with per-workgroup
//     partial column sums finished by gm3d_colsum_finish (deterministic, no atomics, and -- unlike PyTorch's
//     multi-block reduce_kernel on this stack -- safe under hipGraph replay).
#include "common.hpp"

namespace gm3d {


// Thread layout shared by the (G, K, C) kernels: a workgroup is SL row-slices x TPR threads per row, each
// thread owning 8 consecutive channels (16 B in bf16).  C in {128, 256, 384, 512}; blockDim = SL * TPR.
__device__ __forceinline__ int tpr_of(int C) { return C >> 3; }

// ------------------------------------------------------------------ input moments (layer-1 BN statistics)
// partial[block][0..2] = sum x_j, [3..8] = sum of xx, xy, xz, yy, yz, zz over this block's rows.
__global__ __launch_bounds__(256) void moments3_kernel(const float* __restrict__ x, int R, double* __restrict__ partial) {
    __shared__ double red[4][9];
    double s[9];   // fp64: the layer-1 weight gradient cancels against these moments (see pn_layer1_bwd_stats_kernel)
#pragma unroll
    for (int i = 0; i < 9; ++i) s[i] = 0.0;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < R; r += gridDim.x * 256) {
        const double a = x[(size_t)r * 3], b = x[(size_t)r * 3 + 1], c = x[(size_t)r * 3 + 2];
        s[0] += a; s[1] += b; s[2] += c;
        s[3] += a * a; s[4] += a * b; s[5] += a * c; s[6] += b * b; s[7] += b * c; s[8] += c * c;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        double v = s[i];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9) partial[(size_t)blockIdx.x * 9 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ------------------------------------------------------------------ layer 1: a1 = relu(x . wf^T + bf)
// wf (C1,3), bf (C1): conv1 with the BatchNorm affine folded in.  One thread = 8 channels of one row.
template <class T>
__global__ __launch_bounds__(256) void pn_layer1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wf,
                                                            const float* __restrict__ bf, T* __restrict__ a1, int R, int C1) {
    const int tpr = C1 >> 3;
    const size_t total = (size_t)R * tpr;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        const size_t r = t / tpr;
        const int c = (int)(t - r * tpr) * 8;
        const float a = x[r * 3], b = x[r * 3 + 1], d = x[r * 3 + 2];
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float h = a * wf[(c + i) * 3] + b * wf[(c + i) * 3 + 1] + d * wf[(c + i) * 3 + 2] + bf[c + i];
            v[i] = h > 0.f ? h : 0.f;
        }
        V8<T>::store(a1 + r * C1 + c, v);
    }
}

// ------------------------------------------------------------------ max over the K rows of each group (+argmax)
// One thread owns 8 channels of ONE group and walks all K rows (8 independent 16-byte loads in flight): no LDS, no
// barrier; a workgroup covers 256/(C/8) groups.  (The sliced layout the reductions below use needs a barrier and
// a 1-in-SL combine per group, which made this pass latency-bound at 2x its bandwidth time.)
template <class T>
__global__ __launch_bounds__(256) void group_max_fwd_kernel(const T* __restrict__ in, const float* __restrict__ bias,
                                                            T* __restrict__ out, uint8_t* __restrict__ arg, int G, int K, int C) {
    const int tpr = C >> 3, gpb = blockDim.x / tpr;
    const int lane = threadIdx.x % tpr, gl = threadIdx.x / tpr, c = lane * 8;
    for (int g = blockIdx.x * gpb + gl; g < G; g += gridDim.x * gpb) {
        const T* base = in + (size_t)g * K * C + c;
        float best[8];
        int bk[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { best[i] = -INFINITY; bk[i] = 0; }
        int k = 0;
        for (; k + 8 <= K; k += 8) {
            float v[8][8];
#pragma unroll
            for (int u = 0; u < 8; ++u) V8<T>::load(base + (size_t)(k + u) * C, v[u]);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (v[u][i] > best[i]) { best[i] = v[u][i]; bk[i] = k + u; }   // ascending k: first maximum wins
        }
        for (; k < K; ++k) {
            float v[8];
            V8<T>::load(base + (size_t)k * C, v);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (v[i] > best[i]) { best[i] = v[i]; bk[i] = k; }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (bias) best[i] += bias[c + i];
            arg[(size_t)g * C + c + i] = (uint8_t)bk[i];
        }
        V8<T>::store(out + (size_t)g * C + c, best);
    }
}

// din[g,k,c] = (k == arg[g,c]) ? dout[g,c] : 0      (dense scatter: the GEMMs that follow want a dense operand)
template <class T>
__global__ __launch_bounds__(256) void group_max_bwd_kernel(const T* __restrict__ dout, const uint8_t* __restrict__ arg, T* __restrict__ din,
                                     int G, int K, int C) {
    const int tpr = C >> 3, SL = blockDim.x / tpr;
    const int lane = threadIdx.x % tpr, sl = threadIdx.x / tpr, c = lane * 8;
    for (int g = blockIdx.x; g < G; g += gridDim.x) {
        float d[8];
        V8<T>::load(dout + (size_t)g * C + c, d);
        int a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = arg[(size_t)g * C + c + i];
        for (int k = sl; k < K; k += SL) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = a[i] == k ? d[i] : 0.f;
            V8<T>::store(din + ((size_t)g * K + k) * C + c, v);
        }
    }
}

// Per-workgroup column partials: every thread accumulates its 8 channels over the rows it visits; the SL
// slices of a workgroup meet in LDS once, at the end.  NQ quantities per channel.
template <int NQ>
__device__ __forceinline__ void write_partials(float (&acc)[NQ][8], float* sm, float* partial, int C, int c, int sl, int SL) {
    // sm: [SL][NQ*C]
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int i = 0; i < 8; ++i) sm[(size_t)sl * NQ * C + q * C + c + i] = acc[q][i];
    __syncthreads();
    for (int t = threadIdx.x; t < NQ * C; t += blockDim.x) {
        float s = 0.f;
        for (int s2 = 0; s2 < SL; ++s2) s += sm[(size_t)s2 * NQ * C + t];
        partial[(size_t)blockIdx.x * NQ * C + t] = s;
    }
}

// y = y0 + t[group]:  partial[block][0][c] = sum y, [1][c] = sum y^2
template <class T>
__global__ __launch_bounds__(256) void bn_bcast_stats_kernel(const T* __restrict__ y0, const T* __restrict__ t, int G, int K, int C,
                                      float* __restrict__ partial) {
    extern __shared__ float sm[];
    const int tpr = C >> 3, SL = blockDim.x / tpr;
    const int lane = threadIdx.x % tpr, sl = threadIdx.x / tpr, c = lane * 8;
    float acc[2][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[0][i] = acc[1][i] = 0.f;
    for (int g = blockIdx.x; g < G; g += gridDim.x) {
        float tv[8];
        V8<T>::load(t + (size_t)g * C + c, tv);
        for (int k = sl; k < K; k += SL) {
            float v[8];
            V8<T>::load(y0 + ((size_t)g * K + k) * C + c, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float y = v[i] + tv[i]; acc[0][i] += y; acc[1][i] += y * y; }
        }
    }
    write_partials<2>(acc, sm, partial, C, c, sl, SL);
}

// a2 = act((y0 + t[group]) * scale + shift), act(h) = h > 0 ? h : slope*h  (slope 0: ReLU, 0.2: the head's LeakyReLU)
template <class T>
__global__ __launch_bounds__(256) void bn_bcast_apply_relu_kernel(const T* __restrict__ y0, const T* __restrict__ t, const float* __restrict__ scale,
                                           const float* __restrict__ shift, T* __restrict__ a2, int G, int K, int C,
                                           float slope, const int* __restrict__ sel /*source group of output group g, or null*/) {
    const int tpr = C >> 3, SL = blockDim.x / tpr;
    const int lane = threadIdx.x % tpr, sl = threadIdx.x / tpr, c = lane * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = scale[c + i]; sh[i] = shift[c + i]; }
    for (int g = blockIdx.x; g < G; g += gridDim.x) {
        const int sg = sel ? sel[g] : g;
        float tv[8];
        V8<T>::load(t + (size_t)sg * C + c, tv);
        for (int k = sl; k < K; k += SL) {
            float v[8];
            V8<T>::load(y0 + ((size_t)sg * K + k) * C + c, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float h = (v[i] + tv[i]) * sc[i] + sh[i]; v[i] = h > 0.f ? h : slope * h; }
            V8<T>::store(a2 + ((size_t)g * K + k) * C + c, v);
        }
    }
}

// BatchNorm(+ReLU) backward, pass 1: g = da2 * [scale*y+shift > 0], yhat = (y - mean) * rstd
//   partial[block][0][c] = sum g, [1][c] = sum g * yhat
template <class T>
__global__ __launch_bounds__(256) void bn_bcast_bwd_stats_kernel(const T* __restrict__ da2, const T* __restrict__ y0, const T* __restrict__ t,
                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                          const float* __restrict__ mean, const float* __restrict__ rstd, int G, int K, int C,
                                          float* __restrict__ partial, float slope,
                                          const int* __restrict__ sel /*da2 holds G selected groups; y0/t group of da2 group g, or null*/) {
    extern __shared__ float sm[];
    const int tpr = C >> 3, SL = blockDim.x / tpr;
    const int lane = threadIdx.x % tpr, sl = threadIdx.x / tpr, c = lane * 8;
    float sc[8], sh[8], mu[8], rs[8], acc[2][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = scale[c + i]; sh[i] = shift[c + i]; mu[i] = mean[c + i]; rs[i] = rstd[c + i]; acc[0][i] = acc[1][i] = 0.f; }
    for (int g = blockIdx.x; g < G; g += gridDim.x) {
        const int sg = sel ? sel[g] : g;
        float tv[8];
        V8<T>::load(t + (size_t)sg * C + c, tv);
        for (int k = sl; k < K; k += SL) {
            float v[8], d[8];
            V8<T>::load(y0 + ((size_t)sg * K + k) * C + c, v);
            V8<T>::load(da2 + ((size_t)g * K + k) * C + c, d);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float y = v[i] + tv[i];
                const float gg = (y * sc[i] + sh[i] > 0.f) ? d[i] : slope * d[i];
                acc[0][i] += gg; acc[1][i] += gg * (y - mu[i]) * rs[i];
            }
        }
    }
    write_partials<2>(acc, sm, partial, C, c, sl, SL);
}

// pass 2: dy = scale * (g - s1/R - yhat * s2/R)   (scale = gamma*rstd)      -> T (G,K,C)
//         dt[g,c] = sum_k dy[g,k,c]                                           -> f32 (G,C)   (grad of the broadcast term)
// Thread = 8 channels of one group, all K rows (the per-group sum stays in registers: no LDS, no barrier).
template <class T>
__global__ __launch_bounds__(256) void bn_bcast_bwd_apply_kernel(const T* __restrict__ da2, const T* __restrict__ y0,
                                                                 const T* __restrict__ t, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, const float* __restrict__ s1,
                                                                 const float* __restrict__ s2, float inv_rows,
                                                                 T* __restrict__ dy, float* __restrict__ dt, int G, int K, int C,
                                                                 float slope,
                                                                 const int* __restrict__ inv /*group -> its group in da2, -1: da2 = 0; or null*/) {
    const int tpr = C >> 3, gpb = blockDim.x / tpr;
    const int lane = threadIdx.x % tpr, gl = threadIdx.x / tpr, c = lane * 8;
    float sc[8], sh[8], mu[8], rs[8], m1[8], m2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        sc[i] = scale[c + i]; sh[i] = shift[c + i]; mu[i] = mean[c + i]; rs[i] = rstd[c + i];
        m1[i] = s1[c + i] * inv_rows; m2[i] = s2[c + i] * inv_rows;
    }
    for (int g = blockIdx.x * gpb + gl; g < G; g += gridDim.x * gpb) {
        const int cg = inv ? inv[g] : g;
        float tv[8], gs[8];
        V8<T>::load(t + (size_t)g * C + c, tv);
#pragma unroll
        for (int i = 0; i < 8; ++i) gs[i] = 0.f;
        int k = 0;
        for (; k + 4 <= K; k += 4) {
            float v[4][8], d[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                V8<T>::load(y0 + ((size_t)g * K + k + u) * C + c, v[u]);
                if (cg >= 0) V8<T>::load(da2 + ((size_t)cg * K + k + u) * C + c, d[u]);
                else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) d[u][i] = 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float y = v[u][i] + tv[i];
                    const float gg = (y * sc[i] + sh[i] > 0.f) ? d[u][i] : slope * d[u][i];
                    const float r = sc[i] * (gg - m1[i] - (y - mu[i]) * rs[i] * m2[i]);
                    v[u][i] = r; gs[i] += r;
                }
                V8<T>::store(dy + ((size_t)g * K + k + u) * C + c, v[u]);
            }
        }
        for (; k < K; ++k) {
            float v[8], d[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const size_t o = ((size_t)g * K + k) * C + c;
            V8<T>::load(y0 + o, v);
            if (cg >= 0) V8<T>::load(da2 + ((size_t)cg * K + k) * C + c, d);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float y = v[i] + tv[i];
                const float gg = (y * sc[i] + sh[i] > 0.f) ? d[i] : slope * d[i];
                const float r = sc[i] * (gg - m1[i] - (y - mu[i]) * rs[i] * m2[i]);
                v[i] = r; gs[i] += r;
            }
            V8<T>::store(dy + o, v);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) dt[(size_t)g * C + c + i] = gs[i];
    }
}

// df[g, arg[g,c], c] += dfg[g,c] (the max-pool branch of the first stage), in place; partial[block][c] = column
// sums of the resulting df (gradient of the conv bias in front of it).
template <class T>
__global__ __launch_bounds__(256) void group_scatter_add_kernel(T* __restrict__ df, const T* __restrict__ dfg, const uint8_t* __restrict__ arg,
                                         int G, int K, int C, float* __restrict__ partial) {
    extern __shared__ float sm[];
    const int tpr = C >> 3, SL = blockDim.x / tpr;
    const int lane = threadIdx.x % tpr, sl = threadIdx.x / tpr, c = lane * 8;
    float acc[1][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[0][i] = 0.f;
    for (int g = blockIdx.x; g < G; g += gridDim.x) {
        float d[8];
        int a[8];
        V8<T>::load(dfg + (size_t)g * C + c, d);
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = arg[(size_t)g * C + c + i];
        for (int k = sl; k < K; k += SL) {
            float v[8];
            const size_t o = ((size_t)g * K + k) * C + c;
            V8<T>::load(df + o, v);
            bool hit = false;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (a[i] == k) { v[i] += d[i]; hit = true; }
                acc[0][i] += v[i];
            }
            if (hit) V8<T>::store(df + o, v);
        }
    }
    write_partials<1>(acc, sm, partial, C, c, sl, SL);
}

// Layer-1 backward reductions (conv K=3 + BatchNorm + ReLU), nothing of size (rows,128) is written:
//   g1 = da1 * [a1 > 0],  hhat = (x.w1 + b1 - mean) * rstd,  xc = x - xmean
//   partial[block][q][c]: q=0 sum g1, q=1 sum g1*hhat, q=2..4 sum g1*xc_j
// The weight gradient is the small difference of these large sums, so they are accumulated and finished in
// fp64 (full-rate on CDNA vector units; this kernel is bandwidth-bound anyway).
template <class T>
__global__ __launch_bounds__(256) void pn_layer1_bwd_stats_kernel(const T* __restrict__ da1, const T* __restrict__ a1, const float* __restrict__ x,
                                           const float* __restrict__ w1, const float* __restrict__ b1,
                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                           const float* __restrict__ xmean, int R, int C1, double* __restrict__ partial) {
    extern __shared__ double smd[];
    const int tpr = C1 >> 3, SL = blockDim.x / tpr;
    const int lane = threadIdx.x % tpr, sl = threadIdx.x / tpr, c = lane * 8;
    float wx[8], wy[8], wz[8], bb[8], rs[8];
    double acc[5][8];
    const float mx = xmean[0], my = xmean[1], mz = xmean[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        wx[i] = w1[(c + i) * 3]; wy[i] = w1[(c + i) * 3 + 1]; wz[i] = w1[(c + i) * 3 + 2];
        bb[i] = b1[c + i] - mean[c + i]; rs[i] = rstd[c + i];
#pragma unroll
        for (int q = 0; q < 5; ++q) acc[q][i] = 0.0;
    }
    for (size_t r = (size_t)blockIdx.x * SL + sl; r < (size_t)R; r += (size_t)gridDim.x * SL) {
        float d[8], a[8];
        V8<T>::load(da1 + r * C1 + c, d);
        V8<T>::load(a1 + r * C1 + c, a);
        const float px = x[r * 3], py = x[r * 3 + 1], pz = x[r * 3 + 2];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float gg = a[i] > 0.f ? d[i] : 0.f;
            const float hh = (px * wx[i] + py * wy[i] + pz * wz[i] + bb[i]) * rs[i];
            acc[0][i] += (double)gg; acc[1][i] += (double)gg * (double)hh;
            acc[2][i] += (double)gg * (double)(px - mx); acc[3][i] += (double)gg * (double)(py - my);
            acc[4][i] += (double)gg * (double)(pz - mz);
        }
    }
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int i = 0; i < 8; ++i) smd[(size_t)sl * 5 * C1 + q * C1 + c + i] = acc[q][i];
    __syncthreads();
    for (int t = threadIdx.x; t < 5 * C1; t += blockDim.x) {
        double s = 0.0;
        for (int s2 = 0; s2 < SL; ++s2) s += smd[(size_t)s2 * 5 * C1 + t];
        partial[(size_t)blockIdx.x * 5 * C1 + t] = s;
    }
}

// LIN3: the finish of gm3d_lin3_gelu_bwd's partial sums written where the layer's gradients live: columns [0, C) -> db (C) f32, columns
// [(1 + k) C, (2 + k) C) -> dW (C, 3) f32 column k (the same f64 sums, cast once: what .float() of the f64 result gave)
template <bool LIN3>
__global__ __launch_bounds__(256) void colsum_finish_f64_kernel(const double* __restrict__ partial, int nrows, int pitch,
                                                                int ncols, double* __restrict__ out, float* __restrict__ dW,
                                                                float* __restrict__ db, int C) {
    __shared__ double red[8][32];
    const int cx = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;      // four independent chains: the loads stream instead of serialising
    if (c < ncols) {
        int r = slice;
        for (; r + 24 < nrows; r += 32) {
            s0 += partial[(size_t)r * pitch + c];
            s1 += partial[(size_t)(r + 8) * pitch + c];
            s2 += partial[(size_t)(r + 16) * pitch + c];
            s3 += partial[(size_t)(r + 24) * pitch + c];
        }
        for (; r < nrows; r += 8) s0 += partial[(size_t)r * pitch + c];
    }
    red[slice][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slice == 0 && c < ncols) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][cx];
        if (LIN3) {
            const int which = c / C, cc = c - which * C;
            if (which == 0) db[cc] = (float)t;
            else dW[cc * 3 + which - 1] = (float)t;
        } else {
            out[c] = t;
        }
    }
}

// Generic column partial sums of a (R,C) matrix: partial[block][c]; with `roww` (R) the rows are weighted: sum_r roww[r] m[r][c]
// (the loss-prediction head's  a^T d  -- the gradient of its folded output vector -- as a deterministic two-stage sum instead of a
// library GEMV)
template <class T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ m, int R, int C, float* __restrict__ partial,
                                                             const float* __restrict__ roww) {
    extern __shared__ float sm[];
    const int tpr = C >> 3, SL = blockDim.x / tpr;
    const int lane = threadIdx.x % tpr, sl = threadIdx.x / tpr, c = lane * 8;
    float acc[1][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[0][i] = 0.f;
    for (size_t r = (size_t)blockIdx.x * SL + sl; r < (size_t)R; r += (size_t)gridDim.x * SL) {
        float v[8];
        V8<T>::load(m + r * C + c, v);
        const float wr = roww ? roww[r] : 1.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[0][i] += wr * v[i];
    }
    write_partials<1>(acc, sm, partial, C, c, sl, SL);
}


// ------------------------------------------------------------------ K=3 linear + GELU (pos_embed first layer)
__device__ __forceinline__ float gelu3(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu3_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

// out (R,C) = GELU(x (R,3) . w (C,3)^T + b): pos_embed[0..1] (models_mae_learn_loss.py:104-108) without a K=3 GEMM
template <class T>
__global__ __launch_bounds__(256) void lin3_gelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, T* __restrict__ out, int R, int C) {
    const int tpr = C >> 3;
    const size_t total = (size_t)R * tpr;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        const size_t r = t / tpr;
        const int c = (int)(t - r * tpr) * 8;
        const float px = x[r * 3], py = x[r * 3 + 1], pz = x[r * 3 + 2];
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = gelu3(px * w[(c + i) * 3] + py * w[(c + i) * 3 + 1] + pz * w[(c + i) * 3 + 2] + b[c + i]);
        V8<T>::store(out + r * C + c, v);
    }
}

// dpre = dout * GELU'(pre); partial[block][q][c] (fp64): q=0 sum dpre (bias grad), q=1..3 sum dpre * x_j (weight grad)
template <class T>
__global__ __launch_bounds__(256) void lin3_gelu_bwd_kernel(const T* __restrict__ dout, const float* __restrict__ x, const float* __restrict__ w,
                                     const float* __restrict__ b, int R, int C, double* __restrict__ partial) {
    extern __shared__ double smd[];
    const int tpr = C >> 3, SL = blockDim.x / tpr;
    const int lane = threadIdx.x % tpr, sl = threadIdx.x / tpr, c = lane * 8;
    float wx[8], wy[8], wz[8], bb[8];
    double acc[4][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        wx[i] = w[(c + i) * 3]; wy[i] = w[(c + i) * 3 + 1]; wz[i] = w[(c + i) * 3 + 2]; bb[i] = b[c + i];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q][i] = 0.0;
    }
    for (size_t r = (size_t)blockIdx.x * SL + sl; r < (size_t)R; r += (size_t)gridDim.x * SL) {
        float d[8];
        V8<T>::load(dout + r * C + c, d);
        const float px = x[r * 3], py = x[r * 3 + 1], pz = x[r * 3 + 2];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double g = (double)(d[i] * gelu3_grad(px * wx[i] + py * wy[i] + pz * wz[i] + bb[i]));
            acc[0][i] += g; acc[1][i] += g * px; acc[2][i] += g * py; acc[3][i] += g * pz;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 8; ++i) smd[(size_t)sl * 4 * C + q * C + c + i] = acc[q][i];
    __syncthreads();
    for (int t = threadIdx.x; t < 4 * C; t += blockDim.x) {
        double s = 0.0;
        for (int s2 = 0; s2 < SL; ++s2) s += smd[(size_t)s2 * 4 * C + t];
        partial[(size_t)blockIdx.x * 4 * C + t] = s;
    }
}

// ------------------------------------------------------------------ pairwise ranking loss (forward_learning_loss)
// models_mae_learn_loss.py:795-805: pos[i,j] = t_j > t_i, neg[i,j] = t_j < t_i, D = p_j - p_i,
//   loss = sum(-pos*log(sig(D)+1e-6) - neg*log(1-sig(D)+1e-6)) / sum(pos|neg).   One wave per sample, M <= 64.
// out[b][0] = sum of the pair terms, out[b][1] = number of ordered pairs; dp[b][k] = d(sum)/dp_k (unnormalised).
__global__ __launch_bounds__(64) void rank_loss_kernel(const float* __restrict__ p, int ldp, const float* __restrict__ t, int M,
                                                       float* __restrict__ out, float* __restrict__ dp) {
    const int b = blockIdx.x, j = threadIdx.x;
    const float pj = j < M ? p[(size_t)b * ldp + j] : 0.f, tj = j < M ? t[(size_t)b * M + j] : 0.f;
    float ls = 0.f, cnt = 0.f, g = 0.f;
    for (int i = 0; i < M; ++i) {
        const float pi = __shfl(pj, i), ti = __shfl(tj, i);
        if (j < M) {
            // pair (i, j): D = p_j - p_i
            const float sg = 1.0f / (1.0f + __expf(-(pj - pi)));
            const float ds = sg * (1.0f - sg);
            if (tj > ti) { ls -= __logf(sg + 1e-6f); cnt += 1.f; g -= ds / (sg + 1e-6f); }
            else if (tj < ti) { ls -= __logf(1.0f - sg + 1e-6f); cnt += 1.f; g += ds / (1.0f - sg + 1e-6f); }
            // pair (j, i): D' = p_i - p_j, contributes -dL/dD' to p_j
            const float s2 = 1.0f / (1.0f + __expf(-(pi - pj)));
            const float d2 = s2 * (1.0f - s2);
            if (ti > tj) g += d2 / (s2 + 1e-6f);
            else if (ti < tj) g -= d2 / (1.0f - s2 + 1e-6f);
        }
    }
    for (int o = 32; o > 0; o >>= 1) { ls += __shfl_xor(ls, o); cnt += __shfl_xor(cnt, o); }
    if (j == 0) { out[(size_t)b * 2] = ls; out[(size_t)b * 2 + 1] = cnt; }
    if (j < M) dp[(size_t)b * M + j] = g;
}

// tot[0] = sum_b out[b][0], tot[1] = sum_b out[b][1] (one workgroup, fixed order: per-thread strided partial sums, then a tree), and
// loss[0] = tot[0] / tot[1]: the .sum(dim=0) and the division of forward_learning_loss in one launch.
__global__ __launch_bounds__(256) void rank_loss_finish_kernel(const float* __restrict__ out, int B, float* __restrict__ tot,
                                                               float* __restrict__ loss) {
    __shared__ float red[2][256];
    float a = 0.f, c = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) { a += out[(size_t)b * 2]; c += out[(size_t)b * 2 + 1]; }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { tot[0] = red[0][0]; tot[1] = red[1][0]; loss[0] = red[0][0] / red[1][0]; }
}

// d loss / d loss_pred over the FULL (B,L) prediction whose last M columns entered the loss: dfull[b][L - M + j] = dp[b][j] * (g / tot[1]),
// zeros in front (the slice's backward, the division and the product of the autograd graph in one launch).
__global__ __launch_bounds__(256) void rank_loss_bwd_kernel(const float* __restrict__ dp, const float* __restrict__ g,
                                                            const float* __restrict__ tot, int B, int M, int L, float* __restrict__ dfull) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * L) return;
    const int b = i / L, l = i - b * L, j = l - (L - M);
    const float s = g[0] / tot[1];
    dfull[i] = j >= 0 ? dp[(size_t)b * M + j] * s : 0.f;
}

// ModelEma on the int64 counters (BatchNorm num_batches_tracked): e = (int64)(float(e) * decay + w * float(m)), w = 1 - decay, the arithmetic of
// e.copy_(e * decay + (1.0 - decay) * m) on int64 tensors (PyTorch computes in fp32 with the Python scalars as float, the converting copy truncates).
__global__ void ema_counters_kernel(long long* __restrict__ e, const long long* __restrict__ m, int n, float decay, float w) {
#pragma clang fp contract(off)      // two products and a sum, each rounded (PyTorch runs them as separate kernels): no fma
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float a = (float)e[i] * decay, b = w * (float)m[i];
        e[i] = (long long)(a + b);
    }
}

// DropPath factors floor(keep_s + u) / keep_s for S sites x B samples from one uniform draw u (S,B): the add_, floor_ and div_ of
// models_mae_learn_loss.drop_path_scales in one launch, the same IEEE operations.
__global__ __launch_bounds__(256) void drop_path_scales_kernel(const float* __restrict__ u, const float* __restrict__ keep, int S, int B,
                                                               float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S * B) return;
    const float k = keep[i / B];
    out[i] = __fdiv_rn(floorf(__fadd_rn(u[i], k)), k);
}

// dst (R,Np) = [src (R,N) | zeros]: a narrow matrix padded to a full tile width in one launch (instead of zeros + copy_).  8-element chunks.
template <class T>
__global__ __launch_bounds__(256) void pad_cols_kernel(const T* __restrict__ src, size_t lds, int R, int N, T* __restrict__ dst, int Np) {
    const int cpr = Np >> 3;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)R * cpr) return;
    const size_t r = i / cpr;
    const int c = (int)(i - r * cpr) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (c < N) V8<T>::load(src + r * lds + c, v);
    V8<T>::store(dst + r * (size_t)Np + c, v);
}

// ------------------------------------------------------------------ BatchNorm statistic finalisation (one tiny launch)
// Replaces ~16 scalar PyTorch launches per BatchNorm site: from the summed statistics (train) or the running buffers
// (eval) to the folded affine the streaming kernels consume, plus nn.BatchNorm1d's running-statistic update
// (momentum, unbiased variance, num_batches_tracked += 1).  fp64 inside: var = E[y^2] - mean^2 cancels.
__global__ void bn_finalize_kernel(const float* __restrict__ sums /*[2C] sum, sumsq; may be null in eval*/, double rows,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   long long* __restrict__ nbt, float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ mean_out, float* __restrict__ rstd_out, int C, int training) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && training && nbt) nbt[0] += 1;
    if (c >= C) return;
    double mean, var;
    if (training) {
        mean = (double)sums[c] / rows;
        var = (double)sums[C + c] / rows - mean * mean;
        if (var < 0.0) var = 0.0;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * var * (rows / (rows - 1.0)));
    } else {
        mean = running_mean[c]; var = running_var[c];
    }
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const double sc = (double)gamma[c] * rstd;
    scale[c] = (float)sc;
    shift[c] = (float)((double)beta[c] - mean * sc);
    mean_out[c] = (float)mean;
    rstd_out[c] = (float)rstd;
}

// Layer-1 variant: the statistics of h = x.w^T + b are analytic in the input moments mom9 = sums of
// (x,y,z, xx,xy,xz, yy,yz,zz) over `rows` rows.  Emits the folded conv weights wf = w*scale, bf = (b-mean)*scale+beta.
__global__ void pn1_finalize_kernel(const double* __restrict__ mom9, double rows, const float* __restrict__ w /*[C][3]*/,
                                    const float* __restrict__ b, const float* __restrict__ gamma, const float* __restrict__ beta,
                                    float eps, float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                    long long* __restrict__ nbt, float* __restrict__ wf, float* __restrict__ bf,
                                    float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                    double* __restrict__ mcov /*[12]: mean(3), cov(9) row-major; may be null*/,
                                    float* __restrict__ xmean /*[3]; may be null*/, int C, int training) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    double m[3] = {0, 0, 0}, cov[3][3] = {{0}};
    if (training) {
        for (int i = 0; i < 3; ++i) m[i] = mom9[i] / rows;
        const double S[3][3] = {{mom9[3], mom9[4], mom9[5]}, {mom9[4], mom9[6], mom9[7]}, {mom9[5], mom9[7], mom9[8]}};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) cov[i][j] = S[i][j] / rows - m[i] * m[j];
        if (c == 0) {
            if (nbt) nbt[0] += 1;
            if (xmean)
                for (int i = 0; i < 3; ++i) xmean[i] = (float)m[i];
            if (mcov) {
                for (int i = 0; i < 3; ++i) mcov[i] = m[i];
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) mcov[3 + 3 * i + j] = cov[i][j];
            }
        }
    }
    if (c >= C) return;
    const double wx = w[c * 3], wy = w[c * 3 + 1], wz = w[c * 3 + 2];
    double mean, var;
    if (training) {
        mean = wx * m[0] + wy * m[1] + wz * m[2] + (double)b[c];
        const double wv[3] = {wx, wy, wz};
        var = 0.0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) var += wv[i] * cov[i][j] * wv[j];
        if (var < 0.0) var = 0.0;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * var * (rows / (rows - 1.0)));
    } else {
        mean = running_mean[c]; var = running_var[c];
    }
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const double sc = (double)gamma[c] * rstd;
    wf[c * 3] = (float)(wx * sc); wf[c * 3 + 1] = (float)(wy * sc); wf[c * 3 + 2] = (float)(wz * sc);
    bf[c] = (float)(((double)b[c] - mean) * sc + (double)beta[c]);
    mean_out[c] = (float)mean;
    rstd_out[c] = (float)rstd;
}

// Layer-1 backward tail: q = [t1, t2, Ac_x, Ac_y, Ac_z] (5,C) fp64 sums from pn_layer1_bwd_stats ->
// dW1[c][j] = k_c * (Ac_j[c] - t2_c * rstd_c * (W cov)[c][j]), k = gamma*rstd; dgamma = t2; dbeta = t1.
__global__ void pn1_bwd_finalize_kernel(const double* __restrict__ q, const double* __restrict__ mcov,
                                        const float* __restrict__ w, const float* __restrict__ gamma,
                                        const float* __restrict__ rstd, float* __restrict__ dw, float* __restrict__ dgamma,
                                        float* __restrict__ dbeta, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double* cov = mcov + 3;
    const double t1 = q[c], t2 = q[C + c], rs = rstd[c], k = (double)gamma[c] * rs;
    const double wv[3] = {w[c * 3], w[c * 3 + 1], w[c * 3 + 2]};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double wc = wv[0] * cov[j] + wv[1] * cov[3 + j] + wv[2] * cov[6 + j];
        dw[c * 3 + j] = (float)(k * (q[(2 + j) * C + c] - t2 * rs * wc));
    }
    dgamma[c] = (float)t2;
    dbeta[c] = (float)t1;
}

// ------------------------------------------------------------------ teacher-guided mask + visible / masked id lists
// One wave per sample (L <= 64 tokens, lane = token).  generate_mask (P/models_mae_learn_loss.py:744-784): the
// `len_loss` tokens with the highest predicted loss are always masked; the rest are ranked by `noise` and the first
// `len_keep` stay visible.  Ranks are counted with v_readlane broadcasts ((value, index) order, so ties are
// well-defined); the id lists are the visible / masked token indices in ascending order (= boolean-mask indexing order,
// P/:298-299,649-650), written through ballot + popcount prefix positions.
__global__ __launch_bounds__(64) void mask_select_kernel(const float* __restrict__ loss_pred, const float* __restrict__ noise,
                                                         int L, int len_keep, int len_loss, float* __restrict__ mask,
                                                         long long* __restrict__ vis_ids, long long* __restrict__ mask_ids,
                                                         int vis_pitch, int mask_pitch, unsigned char* __restrict__ mask_b = nullptr) {
    const int b = blockIdx.x, l = threadIdx.x;
    const bool in = l < L;
    const float lp = in ? loss_pred[(size_t)b * L + l] : 0.f;
    float nz = in ? noise[(size_t)b * L + l] : 0.f;
    int rank = 0;
    for (int j = 0; j < L; ++j) {
        const float o = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lp), j));
        rank += (o < lp || (o == lp && j < l)) ? 1 : 0;
    }
    if (in && rank >= L - len_loss) nz = INFINITY;
    int rank2 = 0;
    for (int j = 0; j < L; ++j) {
        const float o = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nz), j));
        rank2 += (o < nz || (o == nz && j < l)) ? 1 : 0;
    }
    const bool keep = in && rank2 < len_keep;
    const unsigned long long kb = __ballot(keep), mb = __ballot(in && !keep);
    const unsigned long long below = l == 0 ? 0ull : (~0ull >> (64 - l));
    if (in) {
        mask[(size_t)b * L + l] = keep ? 0.f : 1.f;
        if (mask_b) mask_b[(size_t)b * L + l] = keep ? 0 : 1;
        if (keep) vis_ids[(size_t)b * vis_pitch + __popcll(kb & below)] = l;
        else mask_ids[(size_t)b * mask_pitch + __popcll(mb & below)] = l;
    }
}

// ------------------------------------------------------------------ token / positional-embedding assembly around the mask
// order (B,L) = [visible ids ascending | masked ids ascending] (a permutation of 0..L-1 per sample, gm3d_mask_select).
// Forward: x_vis[b,j] = tokens[b,order[b,j]] and pos_vis[b,j] = pos[b,order[b,j]] for j < V; pos_full[b,j] = pos[b,order[b,j]]
// for all j -- the boolean-mask gathers and the concat of P/models_mae_learn_loss.py:298-300,649-658 in one pass.
// Backward: because `order` is a permutation every source row is written exactly once (no atomics, no zero fill):
// dtokens[b,order[b,j]] = j < V ? dx_vis[b,j] : 0 ; dpos[b,order[b,j]] = dpos_full[b,j] + (j < V ? dpos_vis[b,j] : 0).
// ------------------------------------------------------------------ loss-predictor head tail (increase_dim_2[3] + mean(-1), :152-158,677)
// The last Conv1d(C -> nout) followed by the mean over its nout outputs is one C-vector: wv[c] = mean_o W1[o][c], bm = mean_o b1[o].
// head_fold: wv (f32 and T copies) + bm.  Workgroup = 16 columns x 16 row slices (C/16 workgroups: the matrix is small, the
// point is to spread its nout x C loads over many CUs); the last workgroup also averages the bias.
template <class T>
__global__ __launch_bounds__(256) void head_fold_kernel(const float* __restrict__ W1, const float* __restrict__ b1, int nout, int C,
                                                        float* __restrict__ wv, T* __restrict__ wv_t, float* __restrict__ bm) {
    __shared__ float red[256];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    if ((int)blockIdx.x * 16 < C) {
        const int c = blockIdx.x * 16 + cx;
        float s = 0.f;
        if (c < C) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;       // four independent chains: the loads of a thread overlap
            int o = ry;
            for (; o + 48 < nout; o += 64) {
                s0 += W1[(size_t)o * C + c];
                s1 += W1[(size_t)(o + 16) * C + c];
                s2 += W1[(size_t)(o + 32) * C + c];
                s3 += W1[(size_t)(o + 48) * C + c];
            }
            for (; o < nout; o += 16) s0 += W1[(size_t)o * C + c];
            s = (s0 + s1) + (s2 + s3);
        }
        red[threadIdx.x] = s;
        __syncthreads();
        if (ry == 0 && c < C) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[k * 16 + cx];
            t *= 1.0f / (float)nout;
            wv[c] = t;
            wv_t[c] = (T)t;
        }
    } else {                       // the extra workgroup: bm = mean(b1)
        float s = 0.f;
        for (int o = threadIdx.x; o < nout; o += blockDim.x) s += b1[o];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) bm[0] = red[0] / (float)nout;
    }
}

// out[r] = T-rounded(a[r,:] . wv) + bm : one wave per row (C % 8 == 0)
template <class T>
__global__ __launch_bounds__(256) void head_rowdot_kernel(const T* __restrict__ a, const T* __restrict__ wv, const float* __restrict__ bm,
                                                          int R, int C, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float b = bm[0];
    for (int r = blockIdx.x * 4 + w; r < R; r += gridDim.x * 4) {
        float s = 0.f;
        for (int c = lane * 8; c < C; c += 512) {
            float x[8], y[8];
            V8<T>::load(a + (size_t)r * C + c, x);
            V8<T>::load(wv + c, y);
#pragma unroll
            for (int i = 0; i < 8; ++i) s += x[i] * y[i];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) out[r] = (float)(T)s + b;
    }
}

// backward of the fold: dW1[o][c] = dwv[c] / nout (elementwise over nout*C), db1[o] = sum_r d[r] / nout (the extra, last workgroup)
__global__ __launch_bounds__(256) void head_fold_bwd_kernel(const float* __restrict__ dwv, const float* __restrict__ d, int R, int nout,
                                                            int C, int ew_blocks, float* __restrict__ dW1, float* __restrict__ db1) {
    if ((int)blockIdx.x < ew_blocks) {
        const size_t total = (size_t)nout * C;
        const float inv = 1.0f / (float)nout;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)ew_blocks * blockDim.x)
            dW1[i] = dwv[i % C] * inv;
        return;
    }
    __shared__ float red[256];
    float s = 0.f;
    for (int r = threadIdx.x; r < R; r += blockDim.x) s += d[r];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    const float g = red[0] / (float)nout;
    for (int o = threadIdx.x; o < nout; o += blockDim.x) db1[o] = g;
}

// da[r][c] = T(d[r] * wv[c])
template <class T>
__global__ __launch_bounds__(256) void head_outer_kernel(const float* __restrict__ d, const float* __restrict__ wv, int R, int C,
                                                         T* __restrict__ da) {
    const int tpr = C >> 3;
    const size_t total = (size_t)R * tpr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / tpr;
        const int c = (int)(i - r * tpr) * 8;
        const float dr = d[r];
        float v[8];
        V8<float>::load(wv + c, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] *= dr;
        V8<T>::store(da + r * C + c, v);
    }
}

// PointcloudScaleAndTranslate (datasets/data_transforms.py:20-35) on the device, in place: u (2,B,3) uniform draws ->
// scale = u0*span+lo, shift = (u1*2-1)*t, p = p*scale + shift, every product and sum rounded separately (the same
// values as the elementwise-op chain it replaces).
__global__ __launch_bounds__(256) void scale_translate_kernel(float* __restrict__ pc, const float* __restrict__ u, float lo, float span,
                                                              float t, int B, int N) {
#pragma clang fp contract(off)   // products and sums rounded separately, like the op chain (HIP's __fmul_rn is a contractable `*`)
    const long long total = (long long)B * N * 3;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / ((long long)N * 3)), j = (int)(i % 3);
        const float sc = u[b * 3 + j] * span + lo;
        const float sh = (u[(B + b) * 3 + j] * 2.0f - 1.0f) * t;
        pc[i] = pc[i] * sc + sh;
    }
}

// ids (B, V) int64 with row pitch `pitch` (the visible ids of every cloud) -> sel[b*V + v] = b*G + ids[b][v] (selected group list)
// and inv[b*G + g] = position of group g in that list, -1 when it is not selected.  One workgroup per cloud.
__global__ __launch_bounds__(256) void group_select_maps_kernel(const long long* __restrict__ ids, int pitch, int V, int G,
                                                                int* __restrict__ sel, int* __restrict__ inv) {
    const int b = blockIdx.x;
    for (int g = threadIdx.x; g < G; g += blockDim.x) inv[(size_t)b * G + g] = -1;
    __syncthreads();
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
        long long g = ids[(size_t)b * pitch + v];
        g = g < 0 ? 0 : (g >= G ? G - 1 : g);                 // ids are produced on the device: keep every later access in bounds
        sel[(size_t)b * V + v] = b * G + (int)g;
        inv[(size_t)b * G + g] = b * V + v;
    }
}

template <class T>
__global__ __launch_bounds__(64) void token_assemble_fwd_kernel(const T* __restrict__ tokens, const T* __restrict__ pos,
                                                                const long long* __restrict__ order, int L, int V, int C,
                                                                T* __restrict__ x_vis, T* __restrict__ pos_vis,
                                                                T* __restrict__ pos_full) {
    const int b = blockIdx.x / L, j = blockIdx.x - b * L;
    const int i = (int)order[(size_t)b * L + j];
    const size_t src = ((size_t)b * L + i) * C, dfull = ((size_t)b * L + j) * C, dvis = ((size_t)b * V + j) * C;
    for (int c = threadIdx.x * 8; c < C; c += 64 * 8) {
        float p[8];
        V8<T>::load(pos + src + c, p);
        V8<T>::store(pos_full + dfull + c, p);
        if (j < V) {
            V8<T>::store(pos_vis + dvis + c, p);
            if (tokens) {
                float t[8];
                V8<T>::load(tokens + src + c, t);
                V8<T>::store(x_vis + dvis + c, t);
            }
        }
    }
}

template <class T>
__global__ __launch_bounds__(64) void token_assemble_bwd_kernel(const T* __restrict__ dx_vis, const T* __restrict__ dpos_vis,
                                                                const T* __restrict__ dpos_full, const long long* __restrict__ order,
                                                                int L, int V, int C, T* __restrict__ dtokens, T* __restrict__ dpos) {
    const int b = blockIdx.x / L, j = blockIdx.x - b * L;
    const int i = (int)order[(size_t)b * L + j];
    const size_t dst = ((size_t)b * L + i) * C, sfull = ((size_t)b * L + j) * C, svis = ((size_t)b * V + j) * C;
    for (int c = threadIdx.x * 8; c < C; c += 64 * 8) {
        float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (dpos_full) V8<T>::load(dpos_full + sfull + c, g);
        if (j < V) {
            if (dpos_vis) {
                float v[8];
                V8<T>::load(dpos_vis + svis + c, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] += v[e];
            }
            if (dx_vis) V8<T>::load(dx_vis + svis + c, t);
        }
        V8<T>::store(dpos + dst + c, g);
        if (dtokens) V8<T>::store(dtokens + dst + c, t);
    }
}

static inline bool chan_ok(int C) { return C >= 8 && C % 8 == 0 && C <= 1024; }
static inline int threads_for(int C) { const int tpr = C / 8; int sl = 256 / tpr; if (sl < 1) sl = 1; return sl * tpr; }
static inline int slices_for(int C) { const int tpr = C / 8; int sl = 256 / tpr; return sl < 1 ? 1 : sl; }
static inline int group_grid(int G) { return G < 1024 ? G : 1024; }
static inline int row_grid(int R, int C) { const int sl = slices_for(C); int g = (R + sl - 1) / sl; return g < 1 ? 1 : (g > 1024 ? 1024 : g); }
static inline int moments_grid(int R) { int g = (R + 255) / 256; return g < 1 ? 1 : (g > 256 ? 256 : g); }

}  // namespace gm3d

#define GM3D_DISPATCH(dtype, CALL_BF16, CALL_F32) \
    do { if ((dtype) == GM3D_BF16) { CALL_BF16; } else { CALL_F32; } } while (0)

extern "C" int gm3d_embed_partial_rows(int kind, int n, int C) {
    using namespace gm3d;
    if (kind == 0) return moments_grid(n);        // moments3: n = rows
    if (kind == 1) return group_grid(n);          // (G,K,C) kernels: n = groups
    if (kind == 3) { const int sl = C > 256 ? 4 : 8; int g = (n + sl - 1) / sl; return g > 1024 ? 1024 : (g < 1 ? 1 : g); }   // gm3d_lin3_gelu_bwd (8 row slices per workgroup, 4 for C > 256)
    return row_grid(n, C);                        // (R,C) kernels: n = rows
}

extern "C" int gm3d_moments3(const float* x, int R, double* partial, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!x || !partial || R < 1) return GM3D_EINVAL;
    hipLaunchKernelGGL(moments3_kernel, dim3(moments_grid(R)), dim3(256), 0, (hipStream_t)stream, x, R, partial);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_pn_layer1_fwd(const float* x, const float* wf, const float* bf, void* a1, int R, int C1, int dtype,
                                  gm3d_stream_t stream) {
    using namespace gm3d;
    if (!x || !wf || !bf || !a1 || R < 1) return GM3D_EINVAL;
    if (!chan_ok(C1)) return GM3D_EUNSUPPORTED;
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    const size_t total = (size_t)R * (C1 / 8);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipStream_t st = (hipStream_t)stream;
    GM3D_DISPATCH(dtype,
                  hipLaunchKernelGGL(pn_layer1_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, x, wf, bf, (bf16_t*)a1, R, C1),
                  hipLaunchKernelGGL(pn_layer1_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, x, wf, bf, (float*)a1, R, C1));
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

static int gkc_check(const void* a, const void* b, int G, int K, int C, int dtype) {
    if (!a || !b || G < 1 || K < 1) return GM3D_EINVAL;
    if (!gm3d::chan_ok(C) || K > 255) return GM3D_EUNSUPPORTED;
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    return GM3D_OK;
}

extern "C" int gm3d_group_max_fwd(const void* in, const float* bias, void* out, uint8_t* arg, int G, int K, int C, int dtype,
                                  gm3d_stream_t stream) {
    using namespace gm3d;
    int rc = gkc_check(in, out, G, K, C, dtype);
    if (rc != GM3D_OK) return rc;
    if (!arg) return GM3D_EINVAL;
    const int gpb = 256 / (C / 8) < 1 ? 1 : 256 / (C / 8);
    int grid = (G + gpb - 1) / gpb; grid = grid > 8192 ? 8192 : grid;
    hipStream_t st = (hipStream_t)stream;
    GM3D_DISPATCH(dtype,
                  hipLaunchKernelGGL(group_max_fwd_kernel<bf16_t>, dim3(grid), dim3(gpb * (C / 8)), 0, st,
                                     (const bf16_t*)in, bias, (bf16_t*)out, arg, G, K, C),
                  hipLaunchKernelGGL(group_max_fwd_kernel<float>, dim3(grid), dim3(gpb * (C / 8)), 0, st,
                                     (const float*)in, bias, (float*)out, arg, G, K, C));
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_group_max_bwd(const void* dout, const uint8_t* arg, void* din, int G, int K, int C, int dtype,
                                  gm3d_stream_t stream) {
    using namespace gm3d;
    int rc = gkc_check(dout, din, G, K, C, dtype);
    if (rc != GM3D_OK) return rc;
    if (!arg) return GM3D_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    GM3D_DISPATCH(dtype,
                  hipLaunchKernelGGL(group_max_bwd_kernel<bf16_t>, dim3(group_grid(G)), dim3(threads_for(C)), 0, st,
                                     (const bf16_t*)dout, arg, (bf16_t*)din, G, K, C),
                  hipLaunchKernelGGL(group_max_bwd_kernel<float>, dim3(group_grid(G)), dim3(threads_for(C)), 0, st,
                                     (const float*)dout, arg, (float*)din, G, K, C));
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_bn_bcast_stats(const void* y0, const void* t, int G, int K, int C, float* partial, int dtype,
                                   gm3d_stream_t stream) {
    using namespace gm3d;
    int rc = gkc_check(y0, t, G, K, C, dtype);
    if (rc != GM3D_OK) return rc;
    if (!partial) return GM3D_EINVAL;
    const size_t lds = (size_t)slices_for(C) * 2 * C * 4;
    hipStream_t st = (hipStream_t)stream;
    GM3D_DISPATCH(dtype,
                  hipLaunchKernelGGL(bn_bcast_stats_kernel<bf16_t>, dim3(group_grid(G)), dim3(threads_for(C)), lds, st,
                                     (const bf16_t*)y0, (const bf16_t*)t, G, K, C, partial),
                  hipLaunchKernelGGL(bn_bcast_stats_kernel<float>, dim3(group_grid(G)), dim3(threads_for(C)), lds, st,
                                     (const float*)y0, (const float*)t, G, K, C, partial));
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_bn_bcast_apply_relu_sel(const void* y0, const void* t, const float* scale, const float* shift, void* a2,
                                            const int* sel, int G, int K, int C, float slope, int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    int rc = gkc_check(y0, t, G, K, C, dtype);
    if (rc != GM3D_OK) return rc;
    if (!scale || !shift || !a2) return GM3D_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    GM3D_DISPATCH(dtype,
                  hipLaunchKernelGGL(bn_bcast_apply_relu_kernel<bf16_t>, dim3(group_grid(G)), dim3(threads_for(C)), 0, st,
                                     (const bf16_t*)y0, (const bf16_t*)t, scale, shift, (bf16_t*)a2, G, K, C, slope, sel),
                  hipLaunchKernelGGL(bn_bcast_apply_relu_kernel<float>, dim3(group_grid(G)), dim3(threads_for(C)), 0, st,
                                     (const float*)y0, (const float*)t, scale, shift, (float*)a2, G, K, C, slope, sel));
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_bn_bcast_apply_relu(const void* y0, const void* t, const float* scale, const float* shift, void* a2,
                                        int G, int K, int C, float slope, int dtype, gm3d_stream_t stream) {
    return gm3d_bn_bcast_apply_relu_sel(y0, t, scale, shift, a2, nullptr, G, K, C, slope, dtype, stream);
}

extern "C" int gm3d_bn_bcast_bwd_stats_sel(const void* da2, const void* y0, const void* t, const float* scale,
                                           const float* shift, const float* mean, const float* rstd, const int* sel, int G, int K,
                                           int C, float* partial, float slope, int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    int rc = gkc_check(y0, t, G, K, C, dtype);
    if (rc != GM3D_OK) return rc;
    if (!da2 || !scale || !shift || !mean || !rstd || !partial) return GM3D_EINVAL;
    const size_t lds = (size_t)slices_for(C) * 2 * C * 4;
    hipStream_t st = (hipStream_t)stream;
    GM3D_DISPATCH(dtype,
                  hipLaunchKernelGGL(bn_bcast_bwd_stats_kernel<bf16_t>, dim3(group_grid(G)), dim3(threads_for(C)), lds, st,
                                     (const bf16_t*)da2, (const bf16_t*)y0, (const bf16_t*)t, scale, shift, mean, rstd, G, K, C, partial, slope, sel),
                  hipLaunchKernelGGL(bn_bcast_bwd_stats_kernel<float>, dim3(group_grid(G)), dim3(threads_for(C)), lds, st,
                                     (const float*)da2, (const float*)y0, (const float*)t, scale, shift, mean, rstd, G, K, C, partial, slope, sel));
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_bn_bcast_bwd_stats(const void* da2, const void* y0, const void* t, const float* scale,
                                       const float* shift, const float* mean, const float* rstd, int G, int K, int C,
                                       float* partial, float slope, int dtype, gm3d_stream_t stream) {
    return gm3d_bn_bcast_bwd_stats_sel(da2, y0, t, scale, shift, mean, rstd, nullptr, G, K, C, partial, slope, dtype, stream);
}

extern "C" int gm3d_bn_bcast_bwd_apply_sel(const void* da2, const void* y0, const void* t, const float* scale,
                                           const float* shift, const float* mean, const float* rstd, const float* s1,
                                           const float* s2, void* dy, float* dt, const int* inv, int G, int K, int C, float slope,
                                           int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    int rc = gkc_check(y0, t, G, K, C, dtype);
    if (rc != GM3D_OK) return rc;
    if (!da2 || !scale || !shift || !mean || !rstd || !s1 || !s2 || !dy || !dt) return GM3D_EINVAL;
    const float inv_rows = 1.0f / ((float)G * (float)K);
    const int gpb = 256 / (C / 8) < 1 ? 1 : 256 / (C / 8);
    int grid = (G + gpb - 1) / gpb; grid = grid > 8192 ? 8192 : grid;
    hipStream_t st = (hipStream_t)stream;
    GM3D_DISPATCH(dtype,
                  hipLaunchKernelGGL(bn_bcast_bwd_apply_kernel<bf16_t>, dim3(grid), dim3(gpb * (C / 8)), 0, st,
                                     (const bf16_t*)da2, (const bf16_t*)y0, (const bf16_t*)t, scale, shift, mean, rstd, s1, s2,
                                     inv_rows, (bf16_t*)dy, dt, G, K, C, slope, inv),
                  hipLaunchKernelGGL(bn_bcast_bwd_apply_kernel<float>, dim3(grid), dim3(gpb * (C / 8)), 0, st,
                                     (const float*)da2, (const float*)y0, (const float*)t, scale, shift, mean, rstd, s1, s2,
                                     inv_rows, (float*)dy, dt, G, K, C, slope, inv));
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_bn_bcast_bwd_apply(const void* da2, const void* y0, const void* t, const float* scale,
                                       const float* shift, const float* mean, const float* rstd, const float* s1,
                                       const float* s2, void* dy, float* dt, int G, int K, int C, float slope, int dtype,
                                       gm3d_stream_t stream) {
    return gm3d_bn_bcast_bwd_apply_sel(da2, y0, t, scale, shift, mean, rstd, s1, s2, dy, dt, nullptr, G, K, C, slope, dtype, stream);
}

extern "C" int gm3d_head_fold(const float* W1, const float* b1, int nout, int C, float* wv, void* wv_t, float* bm, int dtype,
                              gm3d_stream_t stream) {
    using namespace gm3d;
    if (!W1 || !b1 || !wv || !wv_t || !bm || nout < 1 || C < 1) return GM3D_EINVAL;
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (C + 15) / 16 + 1;
    GM3D_DISPATCH(dtype,
                  hipLaunchKernelGGL(head_fold_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, W1, b1, nout, C, wv, (bf16_t*)wv_t, bm),
                  hipLaunchKernelGGL(head_fold_kernel<float>, dim3(grid), dim3(256), 0, st, W1, b1, nout, C, wv, (float*)wv_t, bm));
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_head_rowdot(const void* a, const void* wv_t, const float* bm, int R, int C, float* out, int dtype,
                                gm3d_stream_t stream) {
    using namespace gm3d;
    if (!a || !wv_t || !bm || !out || R < 0) return GM3D_EINVAL;
    if (C < 8 || C % 8) return GM3D_EUNSUPPORTED;
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    if (R == 0) return GM3D_OK;
    hipStream_t st = (hipStream_t)stream;
    int grid = (R + 3) / 4; grid = grid > 4096 ? 4096 : grid;
    GM3D_DISPATCH(dtype,
                  hipLaunchKernelGGL(head_rowdot_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)wv_t, bm, R, C, out),
                  hipLaunchKernelGGL(head_rowdot_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)a, (const float*)wv_t, bm, R, C, out));
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_head_fold_bwd(const float* dwv, const float* d, int R, int nout, int C, float* dW1, float* db1,
                                  gm3d_stream_t stream) {
    using namespace gm3d;
    if (!dwv || !d || !dW1 || !db1 || R < 1 || nout < 1 || C < 1) return GM3D_EINVAL;
    long long ew = ((long long)nout * C + 255) / 256;
    const int ew_blocks = (int)(ew > 2048 ? 2048 : ew);
    hipLaunchKernelGGL(head_fold_bwd_kernel, dim3(ew_blocks + 1), dim3(256), 0, (hipStream_t)stream, dwv, d, R, nout, C, ew_blocks, dW1,
                       db1);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_head_outer(const float* d, const float* wv, int R, int C, void* da, int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!d || !wv || !da || R < 0) return GM3D_EINVAL;
    if (C < 8 || C % 8) return GM3D_EUNSUPPORTED;
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    if (R == 0) return GM3D_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t total = (size_t)R * (C / 8);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    GM3D_DISPATCH(dtype,
                  hipLaunchKernelGGL(head_outer_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, d, wv, R, C, (bf16_t*)da),
                  hipLaunchKernelGGL(head_outer_kernel<float>, dim3(grid), dim3(256), 0, st, d, wv, R, C, (float*)da));
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_scale_translate(float* pc, const float* u, float lo, float span, float t, int B, int N, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!pc || !u || B < 0 || N < 0) return GM3D_EINVAL;
    const long long total = (long long)B * N * 3;
    if (total == 0) return GM3D_OK;
    long long grid = (total + 255) / 256;
    hipLaunchKernelGGL(scale_translate_kernel, dim3((unsigned)(grid > 2048 ? 2048 : grid)), dim3(256), 0, (hipStream_t)stream, pc, u, lo, span,
                       t, B, N);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_group_select_maps(const long long* ids, int id_pitch, int B, int V, int G, int* sel, int* inv,
                                      gm3d_stream_t stream) {
    using namespace gm3d;
    if (!ids || !sel || !inv || B < 0 || V < 0 || G < 1 || V > G || id_pitch < V) return GM3D_EINVAL;
    if ((long long)B * G > 0x7fffffffLL) return GM3D_EUNSUPPORTED;
    if (B == 0) return GM3D_OK;
    hipLaunchKernelGGL(group_select_maps_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, ids, id_pitch, V, G, sel, inv);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_group_scatter_add(void* df, const void* dfg, const uint8_t* arg, int G, int K, int C, float* partial,
                                      int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    int rc = gkc_check(df, dfg, G, K, C, dtype);
    if (rc != GM3D_OK) return rc;
    if (!arg || !partial) return GM3D_EINVAL;
    const size_t lds = (size_t)slices_for(C) * C * 4;
    hipStream_t st = (hipStream_t)stream;
    GM3D_DISPATCH(dtype,
                  hipLaunchKernelGGL(group_scatter_add_kernel<bf16_t>, dim3(group_grid(G)), dim3(threads_for(C)), lds, st,
                                     (bf16_t*)df, (const bf16_t*)dfg, arg, G, K, C, partial),
                  hipLaunchKernelGGL(group_scatter_add_kernel<float>, dim3(group_grid(G)), dim3(threads_for(C)), lds, st,
                                     (float*)df, (const float*)dfg, arg, G, K, C, partial));
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_pn_layer1_bwd_stats(const void* da1, const void* a1, const float* x, const float* w1, const float* b1,
                                        const float* mean, const float* rstd, const float* xmean, int R, int C1,
                                        double* partial, int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!da1 || !a1 || !x || !w1 || !b1 || !mean || !rstd || !xmean || !partial || R < 1) return GM3D_EINVAL;
    if (!chan_ok(C1) || C1 > 256) return GM3D_EUNSUPPORTED;
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    const int tpr = C1 / 8, sl = 8;                       // 8 slices: fp64 LDS tile stays under 64 KB
    const size_t lds = (size_t)sl * 5 * C1 * sizeof(double);
    int grid = (R + sl - 1) / sl; grid = grid > 1024 ? 1024 : grid;
    hipStream_t st = (hipStream_t)stream;
    GM3D_DISPATCH(dtype,
                  hipLaunchKernelGGL(pn_layer1_bwd_stats_kernel<bf16_t>, dim3(grid), dim3(sl * tpr), lds, st,
                                     (const bf16_t*)da1, (const bf16_t*)a1, x, w1, b1, mean, rstd, xmean, R, C1, partial),
                  hipLaunchKernelGGL(pn_layer1_bwd_stats_kernel<float>, dim3(grid), dim3(sl * tpr), lds, st,
                                     (const float*)da1, (const float*)a1, x, w1, b1, mean, rstd, xmean, R, C1, partial));
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_colsum_finish_f64(const double* partial, int nrows, int pitch, int ncols, double* out,
                                      gm3d_stream_t stream) {
    using namespace gm3d;
    if (!partial || !out || nrows < 0 || ncols < 1 || pitch < ncols) return GM3D_EINVAL;
    hipLaunchKernelGGL(colsum_finish_f64_kernel<false>, dim3((ncols + 31) / 32), dim3(256), 0, (hipStream_t)stream, partial, nrows,
                       pitch, ncols, out, nullptr, nullptr, 1);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_lin3_finish(const double* partial, int nrows, int C, float* dW, float* db, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!partial || !dW || !db || nrows < 0 || C < 1) return GM3D_EINVAL;
    hipLaunchKernelGGL(colsum_finish_f64_kernel<true>, dim3((4 * C + 31) / 32), dim3(256), 0, (hipStream_t)stream, partial, nrows, 4 * C,
                       4 * C, nullptr, dW, db, C);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_colsum_partial_w(const void* m, const float* roww, int R, int C, float* partial, int dtype, gm3d_stream_t stream);
extern "C" int gm3d_colsum_partial(const void* m, int R, int C, float* partial, int dtype, gm3d_stream_t stream) {
    return gm3d_colsum_partial_w(m, nullptr, R, C, partial, dtype, stream);
}

extern "C" int gm3d_colsum_partial_w(const void* m, const float* roww, int R, int C, float* partial, int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!m || !partial || R < 1) return GM3D_EINVAL;
    if (C < 8 || C % 8 || C > 2048) return GM3D_EUNSUPPORTED;          // one thread per 8 columns, at most 256 threads per row slice
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    const size_t lds = (size_t)slices_for(C) * C * 4;
    hipStream_t st = (hipStream_t)stream;
    GM3D_DISPATCH(dtype,
                  hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, dim3(row_grid(R, C)), dim3(threads_for(C)), lds, st,
                                     (const bf16_t*)m, R, C, partial, roww),
                  hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3(row_grid(R, C)), dim3(threads_for(C)), lds, st,
                                     (const float*)m, R, C, partial, roww));
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_lin3_gelu_fwd(const float* x, const float* w, const float* b, void* out, int R, int C, int dtype,
                                  gm3d_stream_t stream) {
    using namespace gm3d;
    if (!x || !w || !b || !out || R < 1) return GM3D_EINVAL;
    if (!chan_ok(C)) return GM3D_EUNSUPPORTED;
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    const size_t total = (size_t)R * (C / 8);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipStream_t st = (hipStream_t)stream;
    GM3D_DISPATCH(dtype,
                  hipLaunchKernelGGL(lin3_gelu_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, x, w, b, (bf16_t*)out, R, C),
                  hipLaunchKernelGGL(lin3_gelu_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, x, w, b, (float*)out, R, C));
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_lin3_gelu_bwd(const void* dout, const float* x, const float* w, const float* b, int R, int C,
                                  double* partial, int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!dout || !x || !w || !b || !partial || R < 1) return GM3D_EINVAL;
    if (!chan_ok(C) || C > 512) return GM3D_EUNSUPPORTED;
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    const int tpr = C / 8, sl = C > 256 ? 4 : 8;       // C = 384: 192 threads, 48 KiB of fp64 partials
    const size_t lds = (size_t)sl * 4 * C * sizeof(double);
    int grid = (R + sl - 1) / sl; grid = grid > 1024 ? 1024 : grid;
    hipStream_t st = (hipStream_t)stream;
    GM3D_DISPATCH(dtype,
                  hipLaunchKernelGGL(lin3_gelu_bwd_kernel<bf16_t>, dim3(grid), dim3(sl * tpr), lds, st, (const bf16_t*)dout, x, w, b, R, C, partial),
                  hipLaunchKernelGGL(lin3_gelu_bwd_kernel<float>, dim3(grid), dim3(sl * tpr), lds, st, (const float*)dout, x, w, b, R, C, partial));
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_rank_loss(const float* pred, const float* target, int B, int M, float* out, float* dpred,
                              gm3d_stream_t stream) {
    using namespace gm3d;
    if (!pred || !target || !out || !dpred || B < 1 || M < 1) return GM3D_EINVAL;
    if (M > 64) return GM3D_EUNSUPPORTED;
    hipLaunchKernelGGL(rank_loss_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, pred, M, target, M, out, dpred);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_rank_loss_tail(const float* pred, int ldp, const float* target, int B, int M, float* out, float* dpred, float* tot,
                                   float* loss, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!pred || !target || !out || !dpred || !tot || !loss || B < 1 || M < 1 || ldp < M) return GM3D_EINVAL;
    if (M > 64) return GM3D_EUNSUPPORTED;
    hipLaunchKernelGGL(rank_loss_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, pred, ldp, target, M, out, dpred);
    GM3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(rank_loss_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)out, B, tot, loss);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_rank_loss_tail_bwd(const float* dpred, const float* g, const float* tot, int B, int M, int L, float* dfull,
                                       gm3d_stream_t stream) {
    using namespace gm3d;
    if (!dpred || !g || !tot || !dfull || B < 1 || M < 1 || L < M) return GM3D_EINVAL;
    hipLaunchKernelGGL(rank_loss_bwd_kernel, dim3((B * L + 255) / 256), dim3(256), 0, (hipStream_t)stream, dpred, g, tot, B, M, L, dfull);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_ema_counters(long long* e, const long long* m, int n, float decay, float w, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!e || !m || n < 0) return GM3D_EINVAL;
    if (n == 0) return GM3D_OK;
    hipLaunchKernelGGL(ema_counters_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, e, m, n, decay, w);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_drop_path_scales(const float* u, const float* keep, int S, int B, float* out, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!u || !keep || !out || S < 1 || B < 1) return GM3D_EINVAL;
    hipLaunchKernelGGL(drop_path_scales_kernel, dim3((S * B + 255) / 256), dim3(256), 0, (hipStream_t)stream, u, keep, S, B, out);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_pad_cols(const void* src, long long lds, int R, int N, void* dst, int Np, int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!src || !dst || R < 1 || N < 1 || Np < N || lds < N) return GM3D_EINVAL;
    if (N % 8 || Np % 8 || lds % 8) return GM3D_EUNSUPPORTED;
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    const size_t n = (size_t)R * (Np / 8);
    hipStream_t st = (hipStream_t)stream;
    GM3D_DISPATCH(dtype,
                  hipLaunchKernelGGL(pad_cols_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const bf16_t*)src, (size_t)lds, R,
                                     N, (bf16_t*)dst, Np),
                  hipLaunchKernelGGL(pad_cols_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)src, (size_t)lds, R, N,
                                     (float*)dst, Np));
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_bn_finalize(const float* sums, double rows, const float* gamma, const float* beta, float eps,
                                float momentum, float* running_mean, float* running_var, long long* nbt, float* scale,
                                float* shift, float* mean_out, float* rstd_out, int C, int training, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!gamma || !beta || !running_mean || !running_var || !scale || !shift || !mean_out || !rstd_out || C < 1) return GM3D_EINVAL;
    if (training && (!sums || rows < 2.0)) return GM3D_EINVAL;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, rows, gamma, beta, eps,
                       momentum, running_mean, running_var, nbt, scale, shift, mean_out, rstd_out, C, training);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_pn1_finalize(const double* mom9, double rows, const float* w, const float* b, const float* gamma,
                                 const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                 long long* nbt, float* wf, float* bf, float* mean_out, float* rstd_out, double* mcov,
                                 float* xmean, int C, int training, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!w || !b || !gamma || !beta || !running_mean || !running_var || !wf || !bf || !mean_out || !rstd_out || C < 1) return GM3D_EINVAL;
    if (training && (!mom9 || rows < 2.0)) return GM3D_EINVAL;
    hipLaunchKernelGGL(pn1_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, mom9, rows, w, b, gamma, beta,
                       eps, momentum, running_mean, running_var, nbt, wf, bf, mean_out, rstd_out, mcov, xmean, C, training);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_pn1_bwd_finalize(const double* q, const double* mcov, const float* w, const float* gamma, const float* rstd,
                                     float* dw, float* dgamma, float* dbeta, int C, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!q || !mcov || !w || !gamma || !rstd || !dw || !dgamma || !dbeta || C < 1) return GM3D_EINVAL;
    hipLaunchKernelGGL(pn1_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, q, mcov, w, gamma, rstd, dw,
                       dgamma, dbeta, C);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_mask_select_b(const float* loss_pred, const float* noise, int B, int L, int len_keep, int len_loss, float* mask,
                                  unsigned char* mask_bool, long long* vis_ids, long long* mask_ids, int id_pitch, gm3d_stream_t stream);

extern "C" int gm3d_mask_select(const float* loss_pred, const float* noise, int B, int L, int len_keep, int len_loss,
                                float* mask, long long* vis_ids, long long* mask_ids, int id_pitch, gm3d_stream_t stream) {
    return gm3d_mask_select_b(loss_pred, noise, B, L, len_keep, len_loss, mask, nullptr, vis_ids, mask_ids, id_pitch, stream);
}

extern "C" int gm3d_mask_select_b(const float* loss_pred, const float* noise, int B, int L, int len_keep, int len_loss, float* mask,
                                  unsigned char* mask_bool, long long* vis_ids, long long* mask_ids, int id_pitch, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!loss_pred || !noise || !mask || !vis_ids || !mask_ids || B < 0 || L < 1) return GM3D_EINVAL;
    if (len_keep < 0 || len_loss < 0 || len_keep + len_loss > L) return GM3D_EINVAL;
    if (L > 64) return GM3D_EUNSUPPORTED;
    if (B == 0) return GM3D_OK;
    if (id_pitch != 0 && id_pitch < L) return GM3D_EINVAL;
    hipLaunchKernelGGL(mask_select_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, loss_pred, noise, L, len_keep, len_loss,
                       mask, vis_ids, mask_ids, id_pitch ? id_pitch : len_keep, id_pitch ? id_pitch : L - len_keep, mask_bool);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_token_assemble_fwd(const void* tokens, const void* pos, const long long* order, int B, int L, int V, int C,
                                       void* x_vis, void* pos_vis, void* pos_full, int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!pos || !order || !pos_vis || !pos_full || B < 0 || L < 1 || V < 0 || V > L || C < 8) return GM3D_EINVAL;
    if ((tokens == nullptr) != (x_vis == nullptr)) return GM3D_EINVAL;      // both NULL: positions only (tokens already visible-only)
    if (C % 8) return GM3D_EUNSUPPORTED;
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    if (B == 0) return GM3D_OK;
    hipStream_t st = (hipStream_t)stream;
    GM3D_DISPATCH(dtype,
                  hipLaunchKernelGGL(token_assemble_fwd_kernel<bf16_t>, dim3(B * L), dim3(64), 0, st, (const bf16_t*)tokens,
                                     (const bf16_t*)pos, order, L, V, C, (bf16_t*)x_vis, (bf16_t*)pos_vis, (bf16_t*)pos_full),
                  hipLaunchKernelGGL(token_assemble_fwd_kernel<float>, dim3(B * L), dim3(64), 0, st, (const float*)tokens,
                                     (const float*)pos, order, L, V, C, (float*)x_vis, (float*)pos_vis, (float*)pos_full));
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_token_assemble_bwd(const void* dx_vis, const void* dpos_vis, const void* dpos_full, const long long* order, int B,
                                       int L, int V, int C, void* dtokens, void* dpos, int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!order || !dpos || B < 0 || L < 1 || V < 0 || V > L || C < 8) return GM3D_EINVAL;       // dtokens may be NULL
    if (C % 8) return GM3D_EUNSUPPORTED;
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    if (B == 0) return GM3D_OK;
    hipStream_t st = (hipStream_t)stream;
    GM3D_DISPATCH(dtype,
                  hipLaunchKernelGGL(token_assemble_bwd_kernel<bf16_t>, dim3(B * L), dim3(64), 0, st, (const bf16_t*)dx_vis,
                                     (const bf16_t*)dpos_vis, (const bf16_t*)dpos_full, order, L, V, C, (bf16_t*)dtokens, (bf16_t*)dpos),
                  hipLaunchKernelGGL(token_assemble_bwd_kernel<float>, dim3(B * L), dim3(64), 0, st, (const float*)dx_vis,
                                     (const float*)dpos_vis, (const float*)dpos_full, order, L, V, C, (float*)dtokens, (float*)dpos));
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}
