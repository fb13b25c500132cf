// Flat-buffer optimizer step for gfx950: gradient-norm clipping + AdamW + EMA teacher + bf16 GEMM shadows in
// three launches over contiguous buffers.
//
// Beneath: NativeScalerWithGradNormCount.__call__ (Point-MAE_SA3D/util/misc.py:256-270: unscale -> clip_grad_norm_(5.0)
// -> optimizer.step), torch.optim.AdamW as configured by tools/builder.py:40-56 (decoupled weight decay, no decay for
// 1-D / bias / token parameters) and timm-0.4.5 ModelEma.update (ema = ema*decay + (1-decay)*param), called at
// engine_pretrain.py:197,208-212.
//
// Design (MI355X): the 36.8 M live parameters, their gradients, both Adam moments and the EMA teacher are each ONE
// contiguous fp32 buffer (147 MB; parameters that take weight decay first).  The PyTorch path was ~25 multi-tensor
// launches and ~9 passes over those buffers (norm, scale, AdamW, 2x EMA, 2x bf16 casts); here the update is one pass:
// read p,g,m,v,e, write p,m,v,e and the two bf16 shadows the next step's GEMMs consume.  Scalars that change between
// steps (lr, step count, 1-decay) live in device memory so a captured hipGraph replays correctly.
#include "common.hpp"

namespace gm3d {

// partial[block] = sum of g^2 over this block's grid-stride share (n % 4 == 0 guaranteed by the host layout)
__global__ __launch_bounds__(256) void flat_sumsq_kernel(const float* __restrict__ g, long long n, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// One block: total norm, clip coefficient (torch.nn.utils.clip_grad_norm_: max_norm / (norm + 1e-6), clamped to 1),
// step += 1 and the two bias corrections.  scal = {clip_coef, 1-beta1^t, 1-beta2^t, grad_norm}
__global__ __launch_bounds__(256) void optim_prep_kernel(const float* __restrict__ partial, int nb, float max_norm,
                                                         float* __restrict__ step, float beta1, float beta2,
                                                         float* __restrict__ scal) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) s += (double)partial[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
        const float t = step[0] + 1.0f;
        step[0] = t;
        float coef = max_norm > 0.f ? max_norm / (norm + 1e-6f) : 1.0f;
        scal[0] = coef < 1.0f ? coef : 1.0f;
        scal[1] = 1.0f - powf(beta1, t);
        scal[2] = 1.0f - powf(beta2, t);
        scal[3] = norm;
    }
}

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));

// torch.optim.AdamW (fused kernel's arithmetic): p *= 1 - lr*wd; m = lerp(m, g, 1-b1); v = b2*v + (1-b2)*g*g;
// p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps);  then e += w*(p - e);  shadows = bf16(p), bf16(e).
__global__ __launch_bounds__(256) void adamw_ema_flat_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                             float* __restrict__ m, float* __restrict__ v,
                                                             float* __restrict__ e, bf16_t* __restrict__ ps,
                                                             bf16_t* __restrict__ es, long long n, long long n_decay,
                                                             const float* __restrict__ lr_dev, float wd, float beta1,
                                                             float beta2, float eps, const float* __restrict__ ema_w_dev,
                                                             const float* __restrict__ scal,
                                                             const float* __restrict__ lr_scale /*per element, or null*/) {
    const float lr = lr_dev[0], coef = scal[0], bc1 = scal[1], rbc2 = rsqrtf(scal[2]);
    const float ew = ema_w_dev ? ema_w_dev[0] : 0.f;
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 P = reinterpret_cast<float4*>(p)[i];
        const float4 G = reinterpret_cast<const float4*>(g)[i];
        float4 M = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
        float ls[4] = {1.f, 1.f, 1.f, 1.f};                      // layer-wise lr decay (fine-tuning): lr_eff = lr * lr_scale[i]
        if (lr_scale) { const float4 S = reinterpret_cast<const float4*>(lr_scale)[i]; ls[0] = S.x; ls[1] = S.y; ls[2] = S.z; ls[3] = S.w; }
        const bool decayed = i * 4 < n_decay;                    // n_decay % 4 == 0
        float pp[4] = {P.x, P.y, P.z, P.w}, gg[4] = {G.x, G.y, G.z, G.w}, mm[4] = {M.x, M.y, M.z, M.w}, vv[4] = {V.x, V.y, V.z, V.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gr = gg[k] * coef, lre = lr * ls[k];
            pp[k] *= decayed ? 1.0f - lre * wd : 1.0f;
            mm[k] += (1.0f - beta1) * (gr - mm[k]);
            vv[k] = beta2 * vv[k] + (1.0f - beta2) * gr * gr;
            pp[k] -= (lre / bc1) * mm[k] / (sqrtf(vv[k]) * rbc2 + eps);
        }
        reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
        reinterpret_cast<float4*>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
        reinterpret_cast<float4*>(v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
        if (ps) { bf16x4_t q; for (int k = 0; k < 4; ++k) q[k] = (bf16_t)pp[k]; reinterpret_cast<bf16x4_t*>(ps)[i] = q; }
        if (e) {
            float4 E = reinterpret_cast<float4*>(e)[i];
            float ee[4] = {E.x, E.y, E.z, E.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) ee[k] += ew * (pp[k] - ee[k]);
            reinterpret_cast<float4*>(e)[i] = make_float4(ee[0], ee[1], ee[2], ee[3]);
            if (es) { bf16x4_t q; for (int k = 0; k < 4; ++k) q[k] = (bf16_t)ee[k]; reinterpret_cast<bf16x4_t*>(es)[i] = q; }
        }
    }
}

static inline int flat_grid(long long n) { long long g = (n / 4 + 255) / 256; return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g)); }

}  // namespace gm3d

extern "C" int gm3d_flat_partial_rows(long long n) { return n < 4 ? 0 : gm3d::flat_grid(n); }

extern "C" int gm3d_adamw_ema_flat_step_lrd(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* ema,
                                            void* shadow_p, void* shadow_e, long long n, long long n_decay,
                                            const float* lr_dev, const float* lr_scale, float weight_decay, float beta1,
                                            float beta2, float eps, const float* ema_w_dev, float max_norm, float* step_dev,
                                            float* partial, float* scal, gm3d_stream_t stream) {
    using namespace gm3d;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !lr_dev || !step_dev || !partial || !scal || n < 4) return GM3D_EINVAL;
    if ((n & 3) || (n_decay & 3) || n_decay < 0 || n_decay > n) return GM3D_EINVAL;
    if (ema && !ema_w_dev) return GM3D_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int grid = flat_grid(n);
    hipLaunchKernelGGL(flat_sumsq_kernel, dim3(grid), dim3(256), 0, st, grads, n, partial);
    GM3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(optim_prep_kernel, dim3(1), dim3(256), 0, st, partial, grid, max_norm, step_dev, beta1, beta2, scal);
    GM3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(adamw_ema_flat_kernel, dim3(grid), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, ema,
                       (bf16_t*)shadow_p, (bf16_t*)shadow_e, n, n_decay, lr_dev, weight_decay, beta1, beta2, eps, ema_w_dev, scal,
                       lr_scale);
    GM3D_CHECK_LAUNCH();
    return GM3D_OK;
}

extern "C" int gm3d_adamw_ema_flat_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* ema,
                                        void* shadow_p, void* shadow_e, long long n, long long n_decay,
                                        const float* lr_dev, float weight_decay, float beta1, float beta2, float eps,
                                        const float* ema_w_dev, float max_norm, float* step_dev, float* partial,
                                        float* scal, gm3d_stream_t stream) {
    return gm3d_adamw_ema_flat_step_lrd(params, grads, exp_avg, exp_avg_sq, ema, shadow_p, shadow_e, n, n_decay, lr_dev, nullptr,
                                        weight_decay, beta1, beta2, eps, ema_w_dev, max_norm, step_dev, partial, scal, stream);
}
