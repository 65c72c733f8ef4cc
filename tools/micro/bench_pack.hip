#include "../../hqtransformer_amd/csrc/split_conv.hip"
#include "../../hqtransformer_amd/csrc/split_stream_conv.hip"
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const int B = 64;
    struct S { int hw, c; } shapes[] = {{128 * 128, 128}, {128 * 128, 256}, {64 * 64, 256}, {256 * 256, 128}, {32 * 32, 512}};
    float *x, *stats, *gm, *bt; half_t* y; int* flag;
    const size_t nmax = (size_t)B * 256 * 256 * 128;
    CK(hipMalloc(&x, nmax * 4)); CK(hipMalloc(&y, nmax * 4)); CK(hipMalloc(&stats, B * 32 * 2 * 4)); CK(hipMalloc(&gm, 512 * 4)); CK(hipMalloc(&bt, 512 * 4)); CK(hipMalloc(&flag, 256));
    CK(hipMemset(x, 0, nmax * 4)); CK(hipMemset(stats, 0, B * 32 * 2 * 4)); CK(hipMemset(gm, 0, 512 * 4)); CK(hipMemset(bt, 0, 512 * 4)); CK(hipMemset(flag, 0, 256));
    for (auto s : shapes) {
        CK(launch_split_pack(x, y, stats, gm, bt, B, s.hw, s.c, 32, 1, flag, st)); CK(hipStreamSynchronize(st));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a, st));
        for (int r = 0; r < 10; ++r) CK(launch_split_pack(x, y, stats, gm, bt, B, s.hw, s.c, 32, 1, flag, st));
        CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double bytes = (double)B * s.hw * s.c * 8;
        printf("in flight %d: hw %6d C %3d: %8.1f us  %.2f TB/s\n", SP_INFLIGHT, s.hw, s.c, 100.0 * ms, bytes / (ms / 10 * 1e-3) / 1e12);
    }
    return 0;
}
