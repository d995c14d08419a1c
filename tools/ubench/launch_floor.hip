// Per-kernel cost of back-to-back dependent launches on one stream vs the same chain replayed from a hipGraph.
// Build: hipcc -O3 --offload-arch=gfx950 -o launch_floor launch_floor.hip ; run: ./launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>

__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f; }
__global__ void stream_kernel(const float4* __restrict__ a, float4* __restrict__ b, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) b[i] = a[i];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    float* p; float4 *a, *b;
    const int n = 1 << 19;     // 8 MB in, 8 MB out
    CK(hipMalloc(&p, 4096)); CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
    CK(hipMemset(p, 0, 4096)); CK(hipMemset(a, 0, n * 16));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int N = 2000;
    for (int mode = 0; mode < 2; ++mode) {
        auto body = [&]() { for (int i = 0; i < N; ++i) { if (mode == 0) tiny<<<1, 64, 0, s>>>(p); else stream_kernel<<<1024, 256, 0, s>>>(a, b, n); } };
        body(); CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        body(); CK(hipStreamSynchronize(s));
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("mode %d stream : %.2f us / kernel\n", mode, us / N);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        body();
        CK(hipStreamEndCapture(s, &g));
        auto ti = std::chrono::steady_clock::now();
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        double ins = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ti).count();
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        t0 = std::chrono::steady_clock::now();
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("mode %d graph  : %.2f us / kernel (instantiate %.0f us for %d nodes)\n", mode, us / N, ins, N);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
