// Micro-benchmark: aggregate kernel dispatch rate of N graph-replayed chains of dependent kernels (one chain per stream).
// Does the command processor dispatch chains on different queues in parallel, or is there one serial ~us-per-kernel budget?
//   bench_dispatch [spin_cycles] [wgs]      kernels spin for `spin_cycles` clocks on `wgs` workgroups of 256 threads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin_kernel(int* p, long long cycles) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (p && threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1;
}

int main(int argc, char** argv) {
    const long long cycles = argc > 1 ? atoll(argv[1]) : 0;
    const int wgs = argc > 2 ? atoi(argv[2]) : 1;
    const int K = 1000, maxn = 6, reps = 5;
    const bool use_null = argc > 3 && atoi(argv[3]) != 0;      // chain 0 runs on the null stream
    std::vector<hipStream_t> st(maxn);
    std::vector<hipGraphExec_t> ge(maxn);
    std::vector<int*> buf(maxn);
    for (int c = 0; c < maxn; ++c) {
        if (c == 0 && use_null) st[c] = nullptr; else CK(hipStreamCreateWithFlags(&st[c], hipStreamNonBlocking));
        CK(hipMalloc(&buf[c], 64)); CK(hipMemset(buf[c], 0, 64));
        hipGraph_t g;
        hipStream_t cs = st[c];
        if (!cs) CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));      // capture needs a real stream; the exec graph can launch anywhere
        CK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < K; ++k) spin_kernel<<<wgs, 256, 0, cs>>>(buf[c], cycles);
        CK(hipStreamEndCapture(cs, &g));
        CK(hipGraphInstantiate(&ge[c], g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
        CK(hipGraphLaunch(ge[c], st[c])); CK(hipStreamSynchronize(st[c]));
    }
    for (int n = 1; n <= maxn; ++n) {
        CK(hipDeviceSynchronize());
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a, st[0]));
        for (int c = 1; c < n; ++c) CK(hipStreamWaitEvent(st[c], a, 0));
        for (int r = 0; r < reps; ++r)
            for (int c = 0; c < n; ++c) CK(hipGraphLaunch(ge[c], st[c]));
        std::vector<hipEvent_t> done(n);
        for (int c = 1; c < n; ++c) { CK(hipEventCreate(&done[c])); CK(hipEventRecord(done[c], st[c])); CK(hipStreamWaitEvent(st[0], done[c], 0)); }
        CK(hipEventRecord(b, st[0])); CK(hipStreamSynchronize(st[0]));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("spin %lld cycles x %d WG(s), %d chain(s): %.2f us per kernel per chain, %.2f us per kernel aggregate (%.0f k kernels/s)\n", cycles, wgs, n,
               1000.0 * ms / (reps * K), 1000.0 * ms / (reps * K * n), reps * K * n / (ms * 1e-3) / 1e3);
    }
    return 0;
}
