// Micro-benchmark: cost of dependent kernel boundaries on this box, eager vs hipGraph replay.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void tiny(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.0f; }
int main() {
    int* d; float* f;
    CK(hipMalloc(&d, 4)); CK(hipMemset(d, 0, 4));
    const int n = 64 * 1536;
    CK(hipMalloc(&f, n * 4)); CK(hipMemset(f, 0, n * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int mode = 0; mode < 3; ++mode) {
        const int grid = mode == 0 ? 1 : (mode == 1 ? 256 : (n + 255) / 256);
        const int N = 2000;
        auto launch = [&](hipStream_t s) { if (mode == 2) touch<<<grid, 256, 0, s>>>(f, n); else tiny<<<grid, 64, 0, s>>>(d); };
        for (int i = 0; i < 100; ++i) launch(st);
        CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::high_resolution_clock::now();
        CK(hipEventRecord(a, st));
        for (int i = 0; i < N; ++i) launch(st);
        CK(hipEventRecord(b, st));
        CK(hipStreamSynchronize(st));
        auto t1 = std::chrono::high_resolution_clock::now();
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("mode %d grid %d eager: %.2f us/kernel (events), %.2f us/kernel (host wall)\n", mode, grid, 1000 * ms / N,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
        // graph of 200 nodes, replayed 10x
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 200; ++i) launch(st);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        t0 = std::chrono::high_resolution_clock::now();
        CK(hipEventRecord(a, st));
        for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(b, st));
        CK(hipStreamSynchronize(st));
        t1 = std::chrono::high_resolution_clock::now();
        CK(hipEventElapsedTime(&ms, a, b));
        printf("mode %d grid %d graph(200 nodes x10): %.2f us/kernel (events), %.2f us/kernel (host wall)\n", mode, grid,
               1000 * ms / 2000, std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
