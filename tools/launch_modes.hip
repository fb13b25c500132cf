// Per-kernel cost of a DEPENDENT chain on one stream: launched one by one from a tight host loop ("eager") vs captured as a hipGraph and
// replayed, for an empty kernel and for a ~10 us streaming kernel (each launch reads what the previous one wrote).  Also prints the host
// time the launch loop takes (can the host stay ahead of the GPU?).
//   hipcc -O2 --offload-arch=gfx950 tools/launch_modes.hip -o gpurun_out/launch_modes && gpurun_out/launch_modes
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void empty_kernel(float* p) { if (p == nullptr) p[0] = 1.f; }
__global__ void stream_kernel(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = a[i]; v.x += 1.f; b[i] = v; }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const int N = 600, REP = 20;
    hipStream_t s; CK(hipStreamCreate(&s));
    size_t n4 = (size_t)4 << 20;          // 64 MB buffers (16 B x 4 Mi): read 64 + write 64 MB per launch -> ~25 us at 5 TB/s; use a quarter
    n4 >>= 2;
    float4 *a, *b; CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&b, n4 * 16)); CK(hipMemset(a, 0, n4 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int kind = 0; kind < 2; ++kind) {
        auto launch = [&](int i) {
            if (kind == 0) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (float*)a);
            else hipLaunchKernelGGL(stream_kernel, dim3(1024), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, n4);
        };
        for (int i = 0; i < N; ++i) launch(i);
        CK(hipStreamSynchronize(s));
        // eager
        double host = 0; float ms = 0, tot = 0;
        for (int r = 0; r < REP; ++r) {
            CK(hipEventRecord(e0, s));
            double t0 = now_us();
            for (int i = 0; i < N; ++i) launch(i);
            host += now_us() - t0;
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
        }
        printf("%-8s eager  %7.2f us per kernel on the GPU, host loop %5.2f us per launch\n", kind ? "stream" : "empty", tot * 1e3 / REP / N, host / REP / N);
        // graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) launch(i);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        tot = 0; host = 0;
        for (int r = 0; r < REP; ++r) {
            CK(hipEventRecord(e0, s));
            double t0 = now_us();
            CK(hipGraphLaunch(ge, s));
            host += now_us() - t0;
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
        }
        printf("%-8s graph  %7.2f us per kernel on the GPU, hipGraphLaunch %5.2f us per node\n", kind ? "stream" : "empty", tot * 1e3 / REP / N, host / REP / N);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
