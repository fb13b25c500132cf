// Floor of a multi-workgroup FPS step: S workgroups of one cloud exchange one 64-bit {key, step-tag} word per step through L2
// (relaxed agent-scope store, every workgroup polls the S words of the step's parity until all carry the step's tag).  Nothing
// else is computed: the time per round is what the cross-CU hand-off alone costs a dependent chain.
//   hipcc --offload-arch=gfx950 -O3 tools/xchg_bench.hip -o tools/bin/xchg_bench && tools/bin/xchg_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned long long u64;
__global__ __launch_bounds__(256) void xchg(u64* slots, int S, int rounds, u64* out, unsigned* timeout) {
    const int cloud = blockIdx.x / S, s = blockIdx.x % S, tid = threadIdx.x;
    u64* my = slots + (size_t)cloud * 2 * S;
    __shared__ u64 win;
    u64 acc = 0;
    for (int j = 1; j <= rounds; ++j) {
        if (tid == 0) {
            const u64 key = ((u64)(unsigned)(j * 2654435761u + s) << 16) | (unsigned)(j & 0xffff);
            __hip_atomic_store(&my[(j & 1) * S + s], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid < 64) {
            u64 v = 0;
            unsigned spins = 0;
            bool ok;
            do {
                v = tid < S ? __hip_atomic_load(&my[(j & 1) * S + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (u64)(j & 0xffff);
                ok = (v & 0xffff) == (u64)(j & 0xffff);
                if (++spins > (1u << 22)) { if (tid == 0) *timeout = 1; break; }
            } while (!__all(ok));
            v >>= 16;
            for (int o = 32; o; o >>= 1) { u64 t = __shfl_xor(v, o); v = t > v ? t : v; }
            if (tid == 0) win = v;
        }
        __syncthreads();
        acc += win;
        __syncthreads();
    }
    if (tid == 0) out[blockIdx.x] = acc;
}
int main() {
    u64 *slots, *out; unsigned* tmo;
    hipMalloc(&slots, 1 << 20); hipMalloc(&out, 1 << 16); hipMalloc(&tmo, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int rounds = 1200;
    for (int B : {1, 32}) for (int S : {1, 2, 4, 8}) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(slots, 0, 1 << 20); hipMemset(tmo, 0, 4);
            hipEventRecord(e0);
            hipLaunchKernelGGL(xchg, dim3(B * S), dim3(256), 0, 0, slots, S, rounds, out, tmo);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        unsigned t; hipMemcpy(&t, tmo, 4, hipMemcpyDeviceToHost);
        printf("clouds %2d  workgroups/cloud %d : %.3f us per round%s\n", B, S, best * 1e3 / rounds, t ? "  (TIMEOUT)" : "");
    }
    return 0;
}
