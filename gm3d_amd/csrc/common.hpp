// Shared device helpers for the gfx950 kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/gm3d.h"

#define GM3D_WAVE 64

#define GM3D_CHECK_LAUNCH()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return GM3D_ELAUNCH;           \
    } while (0)

namespace gm3d {

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: one static instance per launch site
// remembers, per device ordinal, how many bytes were granted (a process that launches on a second device, or two threads
// racing through the first launch, set it again instead of skipping it).  hipGetDevice is thread-local and capture-safe.
struct LdsAttr {
    std::atomic<int> granted[32];
    bool ensure(const void* fn, size_t lds) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        const bool slot = dev >= 0 && dev < 32;
        if (slot && granted[dev].load(std::memory_order_acquire) >= (int)lds) return true;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
        if (slot) granted[dev].store((int)lds, std::memory_order_release);
        return true;
    }
};

// Squared distance with the evaluation order fixed by the oracle contract
// ((dx*dx + dy*dy) + dz*dz, fp32, no FMA).  The translation unit is compiled with
// -ffp-contract=off; the __f*_rn intrinsics make the intent explicit as well.
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
    float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// ---- DPP cross-lane moves (gfx9 encodings) -------------------------------------------
// quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror 0x141, row_mirror 0x140,
// row_bcast15 0x142, row_bcast31 0x143, wave_shr1 0x138.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ unsigned dpp_u32(unsigned old, unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xF, false);
}

// Wave-wide max of a 64-bit unsigned key, result broadcast to every lane (as a uniform value).
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k) {
#define GM3D_STEP(CTRL, RM)                                                         \
    {                                                                               \
        unsigned lo = (unsigned)k, hi = (unsigned)(k >> 32);                        \
        unsigned olo = dpp_u32<CTRL, RM>(lo, lo), ohi = dpp_u32<CTRL, RM>(hi, hi);  \
        unsigned long long o = ((unsigned long long)ohi << 32) | olo;               \
        k = o > k ? o : k;                                                          \
    }
    GM3D_STEP(0xB1, 0xF)
    GM3D_STEP(0x4E, 0xF)
    GM3D_STEP(0x141, 0xF)
    GM3D_STEP(0x140, 0xF)
    GM3D_STEP(0x142, 0xA)
    GM3D_STEP(0x143, 0xC)
#undef GM3D_STEP
    unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)k, 63);
    unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(k >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

typedef __bf16 bf16_t;

// 8 consecutive elements of `T` as floats: one 16-byte access in bf16, two in f32.
template <class T> struct V8;
template <> struct V8<float> {
    static __device__ __forceinline__ void load(const float* p, float* v) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void store(float* p, const float* v) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
};
template <> struct V8<bf16_t> {
    typedef __bf16 v8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ void load(const bf16_t* p, float* v) {
        const v8 a = *reinterpret_cast<const v8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float* v) {
        v8 a;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (bf16_t)v[i];
        *reinterpret_cast<v8*>(p) = a;
    }
};


// erf for the bf16 path: Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below bf16's 4e-3 resolution) costs one
// rcp + one exp + 5 FMAs, about half of libm's erff; the GELU passes are VALU-bound on it (12.6 M elements per call).
// The f32 (parity) path keeps libm's erff.
template <class T> __device__ __forceinline__ float erf_t(float x);
template <> __device__ __forceinline__ float erf_t<float>(float x) { return erff(x); }
template <> __device__ __forceinline__ float erf_t<bf16_t>(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.0f - poly * __expf(-ax * ax);
    return copysignf(r, x);
}
template <class T> __device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_t<T>(x * 0.70710678118654752440f)); }
template <class T> __device__ __forceinline__ float gelu_grad_f(float x) {
    return 0.5f * (1.0f + erf_t<T>(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}


}  // namespace gm3d
