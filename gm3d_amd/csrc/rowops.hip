// Row-wise fused kernels around the transformer-block GEMMs (gfx950): residual + bias + DropPath +
// positional add + LayerNorm in one pass (forward and backward), and bias + exact-erf GELU (forward
// and backward), with the bias / gamma / beta gradient column sums folded into the backward passes.
//
// Beneath: timm-0.4.5 Block.forward -- in-tree twin Point-MAE_SA3D/models/Point_MAE.py:128-146 --
//   x = x + drop_path(attn(norm1(x)));  x = x + drop_path(mlp(norm2(x)))
// as driven by TransformerEncoder/Decoder.forward (models_mae_learn_loss.py:914-917,984-990:
//   x = block(x + pos) for every block, then a final LayerNorm), Mlp (Point_MAE.py:82-98: fc1, GELU, fc2).
//
// Why these exist (MI355X): the round-1 profile of the PyTorch op chain spends 12 of 28.8 ms per step in
// ~1700 tiny launches (dtype casts, adds, LayerNorm, GELU, bias-grad reductions).  Every one of them is a
// streaming pass over a (rows, 384|1536) tile; fusing them around the GEMMs leaves ONE HBM pass between
// two GEMMs.  The residual stream stays fp32; GEMM operands are `T` (bf16 in throughput mode, f32 in
// parity mode).  A 32-lane half-wave owns one row (384 = 32 lanes x 3 quads, 16-byte accesses): the row statistics
// are DPP reductions, no LDS, no barrier on the forward path.
#include "common.hpp"

namespace gm3d {

constexpr int LNC = 384;            // model width (trans_dim, models_mae_learn_loss.py:110)

__device__ __forceinline__ float wave_sum(float v) {
#define GM3D_SUM_STEP(CTRL, RM) \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, RM, 0xF, false));
    GM3D_SUM_STEP(0xB1, 0xF)   // quad_perm [1,0,3,2]
    GM3D_SUM_STEP(0x4E, 0xF)   // quad_perm [2,3,0,1]
    GM3D_SUM_STEP(0x141, 0xF)  // row_half_mirror
    GM3D_SUM_STEP(0x140, 0xF)  // row_mirror
    GM3D_SUM_STEP(0x142, 0xA)  // row_bcast15 -> rows 1,3
    GM3D_SUM_STEP(0x143, 0xC)  // row_bcast31 -> rows 2,3
#undef GM3D_SUM_STEP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// 4 consecutive elements of `T` as floats: one 16-byte access in f32, one 8-byte access in bf16.
template <class T> struct Quad;
template <> struct Quad<float> {
    static __device__ __forceinline__ void load(const float* p, float* v) {
        const float4 a = *reinterpret_cast<const float4*>(p); v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    }
    static __device__ __forceinline__ void store(float* p, const float* v) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <> struct Quad<bf16_t> {
    typedef __bf16 v4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ void load(const bf16_t* p, float* v) {
        const v4 a = *reinterpret_cast<const v4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (float)a[i];
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float* v) {
        v4 a;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = (bf16_t)v[i];
        *reinterpret_cast<v4*>(p) = a;
    }
};

// Sum over each 32-lane half of the wave, broadcast to the lanes of that half.
__device__ __forceinline__ float half_sum(float v, int lane) {
#define GM3D_SUM_STEP(CTRL, RM) \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, RM, 0xF, false));
    GM3D_SUM_STEP(0xB1, 0xF)
    GM3D_SUM_STEP(0x4E, 0xF)
    GM3D_SUM_STEP(0x141, 0xF)
    GM3D_SUM_STEP(0x140, 0xF)
    GM3D_SUM_STEP(0x142, 0xA)  // rows 1,3 += row 0,2: lane 31 = lower half, lane 63 = upper half
#undef GM3D_SUM_STEP
    const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
    const float hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    return lane < 32 ? lo : hi;
}

constexpr int LN_Q = 12;            // columns per lane: 3 quads at 4*(lane&31) + 128*i

// out_res = res + rowscale[r / rows_per_sample] * (y + bias) + add ;  h = LayerNorm(out_res) * gamma + beta
// res/out_res fp32; y, add, h are T; any of y / bias / rowscale / add / out_res may be null.
// A 32-lane half-wave owns one row: every lane moves 16-byte (f32) / 8-byte (bf16) pieces, the row statistics are
// 5-step DPP reductions, no LDS, no barrier.  Rows past R in the last wave are clamped for the loads and masked
// for the stores (the DPP steps need the whole wave).
template <class T>
__global__ __launch_bounds__(256) void residual_ln_fwd_kernel(const float* __restrict__ res, const T* __restrict__ y,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ rowscale, int rows_per_sample,
                                                              const T* __restrict__ add, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps,
                                                              float* __restrict__ out_res, T* __restrict__ h,
                                                              float* __restrict__ mean, float* __restrict__ rstd, int R) {
    const int lane = threadIdx.x & 63, hl = lane & 31, half = lane >> 5;
    const int wbase = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    const int stride = gridDim.x * 8;
    for (int rb = wbase; rb < R; rb += stride) {
        const bool valid = rb + half < R;
        const int r = valid ? rb + half : R - 1;
        const size_t base = (size_t)r * LNC;
        const float rs = rowscale ? rowscale[r / rows_per_sample] : 1.0f;
        float v[LN_Q];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = 4 * hl + 128 * i;
            float a[4] = {0.f, 0.f, 0.f, 0.f};
            if (res) Quad<float>::load(res + base + c, a);
            if (y) {
                float t[4];
                Quad<T>::load(y + base + c, t);
                if (bias) {
                    float bb[4];
                    Quad<float>::load(bias + c, bb);
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[j] += bb[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] += rs * t[j];
            }
            if (add) {
                float t[4];
                Quad<T>::load(add + base + c, t);
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] += t[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) v[4 * i + j] = a[j];
            if (out_res && valid) Quad<float>::store(out_res + base + c, a);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_Q; ++i) s += v[i];
        const float mu = half_sum(s, lane) * (1.0f / LNC);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_Q; ++i) { const float d = v[i] - mu; q += d * d; }
        const float rsd = rsqrtf(half_sum(q, lane) * (1.0f / LNC) + eps);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = 4 * hl + 128 * i;
            float g[4], b[4], o[4];
            Quad<float>::load(gamma + c, g);
            Quad<float>::load(beta + c, b);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (v[4 * i + j] - mu) * rsd * g[j] + b[j];
            if (valid) Quad<T>::store(h + base + c, o);
        }
        if (hl == 0 && valid) { mean[r] = mu; rstd[r] = rsd; }
    }
}

// ------------------------------------------------------------------ plain LayerNorm of any width C <= 512, C % 4 == 0
// nn.LayerNorm of the Point-M2AE levels (96 / 192 / 384 wide; Point-M2AE_SA3D/cfgs/config_Point_M2AE.yaml:57-99) and any other
// (rows, C) site: h = (x - mean) * rstd * gamma + beta, statistics in fp32 over the row (two passes over registers), x and h in T.
// Half a wave per row like residual_ln_fwd_kernel; lane hl owns the quads at columns 4 hl + 128 i (i < NQ) that lie below C.
template <class T, int NQ>
__global__ __launch_bounds__(256) void ln_plain_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, T* __restrict__ h,
                                                           float* __restrict__ mean, float* __restrict__ rstd, int R, int C,
                                                           const T* __restrict__ y, const float* __restrict__ ybias,
                                                           const float* __restrict__ rowscale, int rows_per_sample,
                                                           const T* __restrict__ z, T* __restrict__ s_out) {
    // optional residual form (the hierarchical encoder's blocks, any width): s = x + rowscale[sample] * (y + ybias) + z is formed
    // here, stored once in T (s_out) and normalised AS STORED (so the backward, which re-reads s, sees the same xhat)
    const int lane = threadIdx.x & 63, hl = lane & 31, half = lane >> 5;
    const int wbase = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    const int stride = gridDim.x * 8;
    const float invc = 1.0f / (float)C;
    for (int rb = wbase; rb < R; rb += stride) {
        const bool valid = rb + half < R;
        const int r = valid ? rb + half : R - 1;
        const size_t base = (size_t)r * C;
        float v[NQ][4];
        float s = 0.f;
        const float rsc = (y && rowscale) ? rowscale[r / rows_per_sample] : 1.0f;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int c = 4 * hl + 128 * i;
            if (c < C) {
                Quad<T>::load(x + base + c, v[i]);
                if (y) {
                    float t[4];
                    Quad<T>::load(y + base + c, t);
                    if (ybias) {
                        float bq[4];
                        Quad<float>::load(ybias + c, bq);
#pragma unroll
                        for (int j = 0; j < 4; ++j) t[j] += bq[j];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[i][j] += rsc * t[j];
                }
                if (z) {
                    float t[4];
                    Quad<T>::load(z + base + c, t);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[i][j] += t[j];
                }
                if (s_out) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[i][j] = (float)(T)v[i][j];
                    if (valid) Quad<T>::store(s_out + base + c, v[i]);
                }
            } else v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
        if (!gamma) continue;                    // sum only (the tail of a block stack: no LayerNorm follows)
        const float mu = half_sum(s, lane) * invc;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NQ; ++i)
            if (4 * hl + 128 * i < C) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mu; q += d * d; }
            }
        const float rsd = rsqrtf(half_sum(q, lane) * invc + eps);
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int c = 4 * hl + 128 * i;
            if (c < C) {
                float g[4], b[4], o[4];
                Quad<float>::load(gamma + c, g);
                Quad<float>::load(beta + c, b);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mu) * rsd * g[j] + b[j];
                if (valid) Quad<T>::store(h + base + c, o);
            }
        }
        if (hl == 0 && valid) { mean[r] = mu; rstd[r] = rsd; }
    }
}

// Backward: g = dh * gamma, dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)); partial[block][0][c] = sum_rows dh * xhat
// (dgamma), partial[block][1][c] = sum_rows dh (dbeta), finished by gm3d_colsum_finish over the blocks.
template <class T, int NQ>
__global__ __launch_bounds__(256) void ln_plain_bwd_kernel(const T* __restrict__ dh, const T* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, T* __restrict__ dx,
                                                           float* __restrict__ partial, int R, int C,
                                                           const T* __restrict__ gin, const float* __restrict__ rowscale,
                                                           int rows_per_sample, T* __restrict__ dy, int nsum,
                                                           T* __restrict__ acc, int acc_mode) {
    // optional residual form: dx = gin + LayerNorm-backward (the gradient of s in ln_plain_fwd_kernel's residual form: it goes to x
    // and z as it stands), dy = rowscale[sample] * dx (the gradient of y; written only when it differs from dx), and with
    // nsum == 3 a third column-sum block: sum_rows dy = the gradient of ybias (the bias of the Linear that produced y).
    // dh == nullptr: no LayerNorm behind the sum (the tail of a block stack) -- dx = gin, nothing of x / mean / rstd / gamma is read.
    // acc (optional): the running sum of dx over the sites that share an addend (the positional embedding, re-added in front of
    // every block): acc_mode 1 acc = dx, 2 acc += dx -- one rounding to T per site, like an accumulation of T tensors
    // (sized by the width class: with the former [8][3 * 512] = 48 KiB for every width a CU held 3 workgroups = 12 waves of this kernel,
    //  and the 96-wide pass over 65,536 rows ran at 2 TB/s)
    constexpr int RW = 128 * NQ;
    __shared__ float red[8][3 * RW];
    const int lane = threadIdx.x & 63, hl = lane & 31, half = lane >> 5, slot = (threadIdx.x >> 6) * 2 + half;
    const int wbase = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    const int stride = gridDim.x * 8;
    const float invc = 1.0f / (float)C;
    float sg[NQ][4], sb[NQ][4], sy[NQ][4], gm[NQ][4];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int c = 4 * hl + 128 * i;
        if (c < C && dh) Quad<float>::load(gamma + c, gm[i]);
        else gm[i][0] = gm[i][1] = gm[i][2] = gm[i][3] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) sg[i][j] = sb[i][j] = sy[i][j] = 0.f;
    }
    for (int rb = wbase; rb < R; rb += stride) {
        const bool valid = rb + half < R;
        const int r = valid ? rb + half : R - 1;
        const size_t base = (size_t)r * C;
        const float mu = dh ? mean[r] : 0.f, rs = dh ? rstd[r] : 0.f;
        float d[NQ][4], xh[NQ][4];
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int c = 4 * hl + 128 * i;
            if (c < C && dh) {
                Quad<T>::load(dh + base + c, d[i]);
                Quad<T>::load(x + base + c, xh[i]);
            } else {                                                    // (no LayerNorm behind the sum: dx = gin)
#pragma unroll
                for (int j = 0; j < 4; ++j) { d[i][j] = 0.f; xh[i][j] = mu; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xh[i][j] = (xh[i][j] - mu) * rs;
                const float g = d[i][j] * gm[i][j];
                a += g;
                b += g * xh[i][j];
            }
        }
        const float m1 = half_sum(a, lane) * invc, m2 = half_sum(b, lane) * invc;
        const float rsc = rowscale ? rowscale[r / rows_per_sample] : 1.0f;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int c = 4 * hl + 128 * i;
            if (c < C) {
                float o[4], gq[4] = {0.f, 0.f, 0.f, 0.f};
                if (gin) Quad<T>::load(gin + base + c, gq);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = dh ? gq[j] + rs * (d[i][j] * gm[i][j] - m1 - xh[i][j] * m2) : gq[j];
                    if (valid) { sg[i][j] += d[i][j] * xh[i][j]; sb[i][j] += d[i][j]; }
                }
                if (valid) Quad<T>::store(dx + base + c, o);
                if (acc_mode && valid) {
                    float aq[4] = {0.f, 0.f, 0.f, 0.f};
                    if (acc_mode == 2) Quad<T>::load(acc + base + c, aq);
#pragma unroll
                    for (int j = 0; j < 4; ++j) aq[j] += (float)(T)o[j];
                    Quad<T>::store(acc + base + c, aq);
                }
                if (nsum == 3) {
                    float oy[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        oy[j] = (float)(T)(rsc * (float)(T)o[j]);          // dy as the next kernel will read it
                        if (valid) sy[i][j] += oy[j];
                    }
                    if (valid && dy) Quad<T>::store(dy + base + c, oy);
                } else if (dy && valid) {
                    float oy[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) oy[j] = rsc * (float)(T)o[j];
                    Quad<T>::store(dy + base + c, oy);
                }
            }
        }
    }
    // the block's eight half-waves meet in LDS: [slot][0..C) = dgamma part, [slot][512..512+C) = dbeta part
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int c = 4 * hl + 128 * i;
        if (c < C) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { red[slot][c + j] = sg[i][j]; red[slot][RW + c + j] = sb[i][j]; red[slot][2 * RW + c + j] = sy[i][j]; }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < nsum * C; t += 256) {
        const int which = t / C, c = t - which * C;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += red[k][which * RW + c];
        partial[((size_t)blockIdx.x * nsum + which) * C + c] = acc;
    }
}

// Backward of residual_ln_fwd_kernel for one LayerNorm site.
//   xhat = (x - mean) * rstd,  g = dh * gamma
//   dx   = gin + rstd * (g - mean_c(g) - xhat * mean_c(g * xhat))          (grad wrt out_res)
//   dy   = rowscale * dx  (T, optional)      acc += dx (fp32, optional: positional-embedding grad)
//   partial[wg][0][c] = sum_rows dh * xhat (dgamma), [1] = sum_rows dh (dbeta), [2] = sum_rows dy (dbias)
template <class T>
__global__ __launch_bounds__(256) void residual_ln_bwd_kernel(const T* __restrict__ dh, const float* __restrict__ gin,
                                                              const float* __restrict__ x, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ rowscale, int rows_per_sample,
                                                              float* __restrict__ dx, T* __restrict__ dy,
                                                              float* __restrict__ acc, T* __restrict__ acc_out,
                                                              float* __restrict__ partial, int R) {
    __shared__ float red[3][8][LNC];
    const int lane = threadIdx.x & 63, hl = lane & 31, half = lane >> 5;
    const int slot = threadIdx.x >> 5;
    const int wbase = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    const int stride = gridDim.x * 8;
    float sg[LN_Q], sb[LN_Q], sy[LN_Q], gm[LN_Q];
#pragma unroll
    for (int i = 0; i < LN_Q; ++i) sg[i] = sb[i] = sy[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) Quad<float>::load(gamma + 4 * hl + 128 * i, gm + 4 * i);

    for (int rb = wbase; rb < R; rb += stride) {
        const bool valid = rb + half < R;
        const int r = valid ? rb + half : R - 1;
        const float keep = valid ? 1.0f : 0.0f;
        const size_t base = (size_t)r * LNC;
        const float mu = mean[r], rsd = rstd[r];
        const float rs = rowscale ? rowscale[r / rows_per_sample] : 1.0f;
        float d[LN_Q], xh[LN_Q];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = 4 * hl + 128 * i;
            float xv[4];
            Quad<T>::load(dh + base + c, d + 4 * i);
            Quad<float>::load(x + base + c, xv);
#pragma unroll
            for (int j = 0; j < 4; ++j) { xh[4 * i + j] = (xv[j] - mu) * rsd; d[4 * i + j] *= keep; }
        }
#pragma unroll
        for (int i = 0; i < LN_Q; ++i) {
            sg[i] += d[i] * xh[i];
            sb[i] += d[i];
            const float g = d[i] * gm[i];
            s1 += g; s2 += g * xh[i];
        }
        const float m1 = half_sum(s1, lane) * (1.0f / LNC), m2 = half_sum(s2, lane) * (1.0f / LNC);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = 4 * hl + 128 * i;
            float a[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = rsd * (d[4 * i + j] * gm[4 * i + j] - m1 - xh[4 * i + j] * m2);
            if (gin) {
                float g4[4];
                Quad<float>::load(gin + base + c, g4);
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] += g4[j];
            }
            if (valid) {
                Quad<float>::store(dx + base + c, a);
                if (acc) {
                    float p4[4];
                    Quad<float>::load(acc + base + c, p4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) p4[j] += a[j];
                    Quad<float>::store(acc + base + c, p4);
                    if (acc_out) Quad<T>::store(acc_out + base + c, p4);      // the finished sum in the GEMM-side type as well
                }
                if (dy) {
                    float y4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { y4[j] = rs * a[j]; sy[4 * i + j] += y4[j]; }
                    Quad<T>::store(dy + base + c, y4);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = 4 * hl + 128 * i;
        Quad<float>::store(&red[0][slot][c], sg + 4 * i);
        Quad<float>::store(&red[1][slot][c], sb + 4 * i);
        Quad<float>::store(&red[2][slot][c], sy + 4 * i);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 3 * LNC; t += 256) {
        const int k = t / LNC, c = t - k * LNC;
        partial[((size_t)blockIdx.x * 3 + k) * LNC + c] =
            ((red[k][0][c] + red[k][1][c]) + (red[k][2][c] + red[k][3][c])) +
            ((red[k][4][c] + red[k][5][c]) + (red[k][6][c] + red[k][7][c]));
    }
}

// out[c] (+)= sum over `nrows` partial rows (row pitch `pitch` floats).
// Block = 32 columns x 8 row slices: every thread streams nrows/8 independent loads (coalesced across the
// 32 columns), then the 8 slices meet in LDS.  (A one-thread-per-column loop is a ~1000-deep dependent
// load chain: 178 us per call in the first profile.)
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ partial, int nrows, int pitch,
                                                            int ncols, float* __restrict__ out, int accumulate) {
    __shared__ float red[8][32];
    const int cx = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < ncols) {
        int r = slice;
        // 16 loads in flight per thread: with 12..36 workgroups on the chip the pass is a chain of load latencies
        // (4 in flight: 11.8 us per call for 512..1024 partial rows)
        for (; r + 120 < nrows; r += 128) {
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = partial[(size_t)(r + 8 * k) * pitch + c];
#pragma unroll
            for (int k = 0; k < 16; k += 4) { s0 += v[k]; s1 += v[k + 1]; s2 += v[k + 2]; s3 += v[k + 3]; }
        }
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
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += red[k][cx];
        out[c] = accumulate ? out[c] + s : s;
    }
}

// g = GELU(f + bias), exact erf form (nn.GELU default).  C % 8 == 0.  blockDim.x = C/8: thread t owns columns
// 8t..8t+7 (bias in registers) of every row the block visits, two rows (2 x 16 B per lane) in flight.
template <class T>
__global__ void bias_gelu_fwd_kernel(const T* __restrict__ f, const float* __restrict__ bias, T* __restrict__ g, int R, int C) {
    const int c = threadIdx.x * 8;
    float bv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bv[i] = bias[c + i];
    int r = blockIdx.x;
    for (; r + (int)gridDim.x < R; r += 2 * gridDim.x) {
        const size_t o0 = (size_t)r * C + c, o1 = (size_t)(r + gridDim.x) * C + c;
        float v0[8], v1[8];
        V8<T>::load(f + o0, v0); V8<T>::load(f + o1, v1);
#pragma unroll
        for (int i = 0; i < 8; ++i) { v0[i] = gelu_f<T>(v0[i] + bv[i]); v1[i] = gelu_f<T>(v1[i] + bv[i]); }
        V8<T>::store(g + o0, v0); V8<T>::store(g + o1, v1);
    }
    for (; r < R; r += gridDim.x) {
        const size_t o0 = (size_t)r * C + c;
        float v0[8];
        V8<T>::load(f + o0, v0);
#pragma unroll
        for (int i = 0; i < 8; ++i) v0[i] = gelu_f<T>(v0[i] + bv[i]);
        V8<T>::store(g + o0, v0);
    }
}

// df = dg * GELU'(f + bias); partial[blockIdx][c] = sum over this block's rows of df (bias gradient).
// blockDim = (C/8, SL): thread (t, sl) owns columns 8t..8t+7 of the rows of slice sl (two rows in flight); the SL
// slices meet in LDS so the partial matrix stays `gridDim.x` rows while SL x more waves stream.
template <class T>
__global__ void bias_gelu_bwd_kernel(const T* __restrict__ dg, const T* __restrict__ f, const float* __restrict__ bias,
                                     T* __restrict__ df, float* __restrict__ partial, int R, int C) {
    extern __shared__ float gelu_red[];   // [SL][C]
    const int c = threadIdx.x * 8;
    const int SL = blockDim.y, sl = threadIdx.y;
    const int stride = gridDim.x * SL;
    float bv[8], s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { bv[i] = bias[c + i]; s[i] = 0.f; }
    int r = blockIdx.x * SL + sl;
    for (; r + stride < R; r += 2 * stride) {
        const size_t o0 = (size_t)r * C + c, o1 = (size_t)(r + stride) * C + c;
        float f0[8], g0[8], f1[8], g1[8];
        V8<T>::load(f + o0, f0); V8<T>::load(dg + o0, g0);
        V8<T>::load(f + o1, f1); V8<T>::load(dg + o1, g1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            g0[i] *= gelu_grad_f<T>(f0[i] + bv[i]); g1[i] *= gelu_grad_f<T>(f1[i] + bv[i]);
            s[i] += g0[i] + g1[i];
        }
        V8<T>::store(df + o0, g0); V8<T>::store(df + o1, g1);
    }
    for (; r < R; r += stride) {
        const size_t o0 = (size_t)r * C + c;
        float f0[8], g0[8];
        V8<T>::load(f + o0, f0); V8<T>::load(dg + o0, g0);
#pragma unroll
        for (int i = 0; i < 8; ++i) { g0[i] *= gelu_grad_f<T>(f0[i] + bv[i]); s[i] += g0[i]; }
        V8<T>::store(df + o0, g0);
    }
    V8<float>::store(gelu_red + (size_t)sl * C + c, s);
    __syncthreads();
    if (sl == 0) {
        for (int k = 1; k < SL; ++k) {
            float o[8];
            V8<float>::load(gelu_red + (size_t)k * C + c, o);
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] += o[i];
        }
        V8<float>::store(partial + (size_t)blockIdx.x * C + c, s);
    }
}

// Batched second stage: job j sums partial[j*job_stride + r*pitch + c] over r < nrows into out[j*out_stride + c].
__global__ __launch_bounds__(256) void colsum_finish_batched_kernel(const float* __restrict__ partial, size_t job_stride,
                                                                    int nrows, int pitch, int ncols,
                                                                    float* __restrict__ out, int out_stride) {
    __shared__ float red[8][32];
    const int cx = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    const float* p = partial + (size_t)blockIdx.y * job_stride;
    float s0 = 0.f, s1 = 0.f;
    if (c < ncols) {
        int r = slice;
        for (; r + 120 < nrows; r += 128) {          // 16 loads in flight (see colsum_finish_kernel)
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = p[(size_t)(r + 8 * k) * pitch + c];
#pragma unroll
            for (int k = 0; k < 16; k += 2) { s0 += v[k]; s1 += v[k + 1]; }
        }
        for (; r + 8 < nrows; r += 16) { s0 += p[(size_t)r * pitch + c]; s1 += p[(size_t)(r + 8) * pitch + c]; }
        for (; r < nrows; r += 8) s0 += p[(size_t)r * pitch + c];
    }
    red[slice][cx] = s0 + s1;
    __syncthreads();
    if (slice == 0 && c < ncols) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += red[k][cx];
        out[(size_t)blockIdx.y * out_stride + c] = s;
    }
}

// Few rows, very many columns (the S-way row-split weight gradients: S x (N*K) partial products per block): out[j][c] =
// sum_r partial[j][r][c], 16 bytes per lane per row, all S loads of a thread in flight.
__global__ __launch_bounds__(256) void sum_few_rows_kernel(const float* __restrict__ partial, int nrows, size_t ncols4,
                                                           float* __restrict__ out) {
    const size_t job = blockIdx.y;
    const float4* p = reinterpret_cast<const float4*>(partial) + job * nrows * ncols4;
    float4* o = reinterpret_cast<float4*>(out) + job * ncols4;
    for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < ncols4; c += (size_t)gridDim.x * 256) {
        float4 a = p[c];
#pragma unroll 8
        for (int r = 1; r < nrows; ++r) {
            const float4 b = p[(size_t)r * ncols4 + c];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        o[c] = a;
    }
}

// (long row streams -- the hierarchical encoder's 65,536-row level 0 -- get four times the workgroups; every workgroup writes one row of
//  the partial sums, which the stacks finish in one batched launch)
static int LN_GRID_CAP_MID = 512;      // measurement knob (gm3d_ln_set_grid_cap): the cap for 8192 <= R < 32768
static inline int ln_grid(int R) {
    int g = (R + 7) / 8;
    const int cap = R >= 32768 ? 2048 : (R >= 8192 ? LN_GRID_CAP_MID : 512);
    return g < 1 ? 1 : (g > cap ? cap : g);
}
static inline int ln_fwd_grid(int R) { int g = (R + 7) / 8; return g < 1 ? 1 : (g > 4096 ? 4096 : g); }   // no partial rows: one row per half-wave
// (long row streams -- the 65,536-row level 0 of the hierarchical encoder -- get four times the workgroups: at 512 a CU holds 6 waves of
//  this kernel and the pass runs at 2.7 TB/s; the partial rows grow with the grid and are finished in one batched launch per stack)
static inline int gelu_bwd_grid(int R) { return R < 512 ? R : (R >= 32768 ? 2048 : 512); }

}  // namespace gm3d

extern "C" int gm3d_ln_set_grid_cap(int cap) {
    if (cap < 64 || cap > 4096) return GM3D_EINVAL;
    gm3d::LN_GRID_CAP_MID = cap;
    return GM3D_OK;
}

extern "C" int gm3d_ln_partial_rows(int R) { return R < 1 ? 0 : gm3d::ln_grid(R); }
extern "C" int gm3d_gelu_partial_rows(int R) { return R < 1 ? 0 : gm3d::gelu_bwd_grid(R); }

extern "C" int gm3d_residual_ln_fwd(const float* res, const void* y, const float* bias, const float* rowscale,
                                    int rows_per_sample, const void* add, const float* gamma, const float* beta,
                                    float eps, float* out_res, void* h, float* mean, float* rstd, int R, int C,
                                    int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!gamma || !beta || !h || !mean || !rstd || R < 0 || (!res && !y && !add)) return GM3D_EINVAL;
    if (rowscale && rows_per_sample < 1) return GM3D_EINVAL;
    if (C != LNC) return GM3D_EUNSUPPORTED;
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    if (R == 0) return GM3D_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GM3D_BF16)
        hipLaunchKernelGGL(residual_ln_fwd_kernel<bf16_t>, dim3(ln_fwd_grid(R)), dim3(256), 0, st, res, (const bf16_t*)y, bias,
                           rowscale, rows_per_sample, (const bf16_t*)add, gamma, beta, eps, out_res, (bf16_t*)h, mean, rstd, R);
    else
        hipLaunchKernelGGL(residual_ln_fwd_kernel<float>, dim3(ln_fwd_grid(R)), dim3(256), 0, st, res, (const float*)y, bias,
                           rowscale, rows_per_sample, (const float*)add, gamma, beta, eps, out_res, (float*)h, mean, rstd, R);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_residual_ln_bwd(const void* dh, const float* gin, const float* x, const float* mean,
                                    const float* rstd, const float* gamma, const float* rowscale, int rows_per_sample,
                                    float* dx, void* dy, float* acc, void* acc_out, float* partial, int R, int C, int dtype,
                                    gm3d_stream_t stream) {
    using namespace gm3d;
    if (!dh || !x || !mean || !rstd || !gamma || !dx || !partial || R < 0) return GM3D_EINVAL;
    if (rowscale && rows_per_sample < 1) return GM3D_EINVAL;
    if (acc_out && !acc) return GM3D_EINVAL;
    if (C != LNC) return GM3D_EUNSUPPORTED;
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    if (R == 0) return GM3D_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GM3D_BF16)
        hipLaunchKernelGGL(residual_ln_bwd_kernel<bf16_t>, dim3(ln_grid(R)), dim3(256), 0, st, (const bf16_t*)dh, gin, x, mean,
                           rstd, gamma, rowscale, rows_per_sample, dx, (bf16_t*)dy, acc, (bf16_t*)acc_out, partial, R);
    else
        hipLaunchKernelGGL(residual_ln_bwd_kernel<float>, dim3(ln_grid(R)), dim3(256), 0, st, (const float*)dh, gin, x, mean,
                           rstd, gamma, rowscale, rows_per_sample, dx, (float*)dy, acc, (float*)acc_out, partial, R);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_ln_plain_partial_rows(int R) { return R < 1 ? 0 : gm3d::ln_grid(R); }

extern "C" int gm3d_add_ln_fwd(const void* x, const void* y, const float* ybias, const float* rowscale, int rows_per_sample, const void* z,
                               const float* gamma, const float* beta, float eps, void* s_out, void* h, float* mean, float* rstd, int R,
                               int C, int dtype, gm3d_stream_t stream);

extern "C" int gm3d_ln_plain_fwd(const void* x, const float* gamma, const float* beta, float eps, void* h, float* mean, float* rstd,
                                 int R, int C, int dtype, gm3d_stream_t stream) {
    return gm3d_add_ln_fwd(x, nullptr, nullptr, nullptr, 1, nullptr, gamma, beta, eps, nullptr, h, mean, rstd, R, C, dtype, stream);
}

extern "C" int gm3d_add_ln_fwd(const void* x, const void* y, const float* ybias, const float* rowscale, int rows_per_sample, const void* z,
                               const float* gamma, const float* beta, float eps, void* s_out, void* h, float* mean, float* rstd, int R,
                               int C, int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!x || R < 0) return GM3D_EINVAL;
    if (gamma ? (!beta || !h || !mean || !rstd) : (!s_out || !(y || z))) return GM3D_EINVAL;      // gamma NULL: the sum only
    if ((ybias || rowscale) && !y) return GM3D_EINVAL;
    if ((y || z) && !s_out) return GM3D_EINVAL;                    // the backward re-reads the sum
    if (rowscale && rows_per_sample < 1) return GM3D_EINVAL;
    if (C < 4 || C % 4 || C > 512) return GM3D_EUNSUPPORTED;
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    if (R == 0) return GM3D_OK;
    hipStream_t st = (hipStream_t)stream;
    const int nq = (C + 127) / 128;
#define GM3D_LNP_F(T_, NQ_) hipLaunchKernelGGL((ln_plain_fwd_kernel<T_, NQ_>), dim3(ln_fwd_grid(R)), dim3(256), 0, st, (const T_*)x, gamma, \
                                               beta, eps, (T_*)h, mean, rstd, R, C, (const T_*)y, ybias, rowscale, rows_per_sample,      \
                                               (const T_*)z, (T_*)s_out)
    if (dtype == GM3D_BF16) { if (nq == 1) GM3D_LNP_F(bf16_t, 1); else if (nq == 2) GM3D_LNP_F(bf16_t, 2); else if (nq == 3) GM3D_LNP_F(bf16_t, 3); else GM3D_LNP_F(bf16_t, 4); }
    else { if (nq == 1) GM3D_LNP_F(float, 1); else if (nq == 2) GM3D_LNP_F(float, 2); else if (nq == 3) GM3D_LNP_F(float, 3); else GM3D_LNP_F(float, 4); }
#undef GM3D_LNP_F
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_add_ln_bwd_acc(const void* dh, const void* gin, const void* x, const float* mean, const float* rstd, const float* gamma,
                                   const float* rowscale, int rows_per_sample, void* dx, void* dy, float* partial, int nsum, void* acc,
                                   int acc_mode, int R, int C, int dtype, gm3d_stream_t stream);
extern "C" int gm3d_add_ln_bwd(const void* dh, const void* gin, const void* x, const float* mean, const float* rstd, const float* gamma,
                               const float* rowscale, int rows_per_sample, void* dx, void* dy, float* partial, int nsum, int R, int C,
                               int dtype, gm3d_stream_t stream);

extern "C" int gm3d_ln_plain_bwd(const void* dh, const void* x, const float* mean, const float* rstd, const float* gamma, void* dx,
                                 float* partial, int R, int C, int dtype, gm3d_stream_t stream) {
    return gm3d_add_ln_bwd(dh, nullptr, x, mean, rstd, gamma, nullptr, 1, dx, nullptr, partial, 2, R, C, dtype, stream);
}

extern "C" int gm3d_add_ln_bwd(const void* dh, const void* gin, const void* x, const float* mean, const float* rstd, const float* gamma,
                               const float* rowscale, int rows_per_sample, void* dx, void* dy, float* partial, int nsum, int R, int C,
                               int dtype, gm3d_stream_t stream) {
    return gm3d_add_ln_bwd_acc(dh, gin, x, mean, rstd, gamma, rowscale, rows_per_sample, dx, dy, partial, nsum, nullptr, 0, R, C, dtype,
                               stream);
}

extern "C" int gm3d_add_ln_bwd_acc(const void* dh, const void* gin, const void* x, const float* mean, const float* rstd, const float* gamma,
                                   const float* rowscale, int rows_per_sample, void* dx, void* dy, float* partial, int nsum, void* acc,
                                   int acc_mode, int R, int C, int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!dx || !partial || R < 1 || (!dh && !gin)) return GM3D_EINVAL;
    if (dh && (!x || !mean || !rstd || !gamma)) return GM3D_EINVAL;           // dh == NULL: the sum-only backward reads none of them
    if (acc_mode < 0 || acc_mode > 2 || (acc_mode && !acc)) return GM3D_EINVAL;
    if ((nsum != 2 && nsum != 3) || (rowscale && (!dy || rows_per_sample < 1))) return GM3D_EINVAL;
    if (C < 4 || C % 4 || C > 512) return GM3D_EUNSUPPORTED;
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nq = (C + 127) / 128;
#define GM3D_LNP_B(T_, NQ_) hipLaunchKernelGGL((ln_plain_bwd_kernel<T_, NQ_>), dim3(ln_grid(R)), dim3(256), 0, st, (const T_*)dh, (const T_*)x, \
                                               mean, rstd, gamma, (T_*)dx, partial, R, C, (const T_*)gin, rowscale, rows_per_sample,     \
                                               (T_*)dy, nsum, (T_*)acc, acc_mode)
    if (dtype == GM3D_BF16) { if (nq == 1) GM3D_LNP_B(bf16_t, 1); else if (nq == 2) GM3D_LNP_B(bf16_t, 2); else if (nq == 3) GM3D_LNP_B(bf16_t, 3); else GM3D_LNP_B(bf16_t, 4); }
    else { if (nq == 1) GM3D_LNP_B(float, 1); else if (nq == 2) GM3D_LNP_B(float, 2); else if (nq == 3) GM3D_LNP_B(float, 3); else GM3D_LNP_B(float, 4); }
#undef GM3D_LNP_B
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_colsum_finish(const float* partial, int nrows, int pitch, int ncols, float* out, int accumulate,
                                  gm3d_stream_t stream) {
    using namespace gm3d;
    if (!partial || !out || nrows < 0 || ncols < 1 || pitch < ncols) return GM3D_EINVAL;
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((ncols + 31) / 32), dim3(256), 0, (hipStream_t)stream, partial, nrows,
                       pitch, ncols, out, accumulate);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_bias_gelu_fwd(const void* f, const float* bias, void* g, int R, int C, int dtype,
                                  gm3d_stream_t stream) {
    using namespace gm3d;
    if (!f || !bias || !g || R < 0 || C < 8) return GM3D_EINVAL;
    if (C % 8 || C / 8 > 1024) return GM3D_EUNSUPPORTED;
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    if (R == 0) return GM3D_OK;
    const int cap = R >= 32768 ? 8192 : 4096;          // one wave per workgroup at C = 384: 32 of them fit a CU
    int grid = (R + 1) / 2; grid = grid > cap ? cap : grid;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GM3D_BF16)
        hipLaunchKernelGGL(bias_gelu_fwd_kernel<bf16_t>, dim3(grid), dim3(C / 8), 0, st, (const bf16_t*)f, bias, (bf16_t*)g, R, C);
    else
        hipLaunchKernelGGL(bias_gelu_fwd_kernel<float>, dim3(grid), dim3(C / 8), 0, st, (const float*)f, bias, (float*)g, R, C);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_bias_gelu_bwd(const void* dg, const void* f, const float* bias, void* df, float* partial, int R,
                                  int C, int dtype, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!dg || !f || !bias || !df || !partial || R < 0 || C < 8) return GM3D_EINVAL;
    if (C % 8 || C / 8 > 1024) return GM3D_EUNSUPPORTED;
    if (dtype != GM3D_F32 && dtype != GM3D_BF16) return GM3D_EINVAL;
    if (R == 0) return GM3D_OK;
    hipStream_t st = (hipStream_t)stream;
    int SL = 1024 / (C / 8); SL = SL > 4 ? 4 : SL;
    const size_t lds = (size_t)SL * C * sizeof(float);
    if (dtype == GM3D_BF16)
        hipLaunchKernelGGL(bias_gelu_bwd_kernel<bf16_t>, dim3(gelu_bwd_grid(R)), dim3(C / 8, SL), lds, st, (const bf16_t*)dg,
                           (const bf16_t*)f, bias, (bf16_t*)df, partial, R, C);
    else
        hipLaunchKernelGGL(bias_gelu_bwd_kernel<float>, dim3(gelu_bwd_grid(R)), dim3(C / 8, SL), lds, st, (const float*)dg,
                           (const float*)f, bias, (float*)df, partial, R, C);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_colsum_finish_batched(const float* partial, int njobs, long long job_stride, int nrows, int pitch,
                                          int ncols, float* out, int out_stride, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!partial || !out || njobs < 1 || nrows < 0 || ncols < 1 || pitch < ncols || job_stride < 0 || out_stride < ncols)
        return GM3D_EINVAL;
    if (njobs > 65535) return GM3D_EUNSUPPORTED;
    hipLaunchKernelGGL(colsum_finish_batched_kernel, dim3((ncols + 31) / 32, njobs), dim3(256), 0, (hipStream_t)stream, partial,
                       (size_t)job_stride, nrows, pitch, ncols, out, out_stride);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_sum_few_rows(const float* partial, int njobs, int nrows, long long ncols, float* out, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!partial || !out || njobs < 1 || nrows < 1 || ncols < 4) return GM3D_EINVAL;
    if (ncols % 4 || njobs > 65535 || nrows > 64) return GM3D_EUNSUPPORTED;
    const size_t n4 = (size_t)ncols / 4;
    int gx = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(sum_few_rows_kernel, dim3(gx, njobs), dim3(256), 0, (hipStream_t)stream, partial, nrows, n4, out);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}
