// Micro-benchmark: what a plain streaming read reaches on this box (non-temporal 16-B loads, U in flight per lane),
// by workgroups per CU and bytes per launch -- the yardstick for the weight-streaming GEMMs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int U>
__global__ __launch_bounds__(512) void read_kernel(const u32x4* __restrict__ src, size_t n_vec, unsigned* out) {
    const size_t stride = (size_t)gridDim.x * 512;
    size_t i = (size_t)blockIdx.x * 512 + threadIdx.x;
    unsigned acc = 0;
    for (; i + (U - 1) * stride < n_vec; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int U>
static void run(const u32x4* buf, size_t total_bytes, size_t bytes_per_launch, int wgs, hipStream_t st, unsigned* out) {
    const int launches = (int)(total_bytes / bytes_per_launch);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a, st));
        for (int l = 0; l < launches; ++l)
            read_kernel<U><<<wgs, 512, 0, st>>>(buf + (size_t)l * (bytes_per_launch / 16), bytes_per_launch / 16, out);
        CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    }
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("  U=%2d  %4d WGs  %6.1f MB/launch x %3d : %7.2f us/launch  %5.2f TB/s\n", U, wgs, bytes_per_launch / 1e6, launches, 1000.f * ms / launches,
           (double)launches * bytes_per_launch / (ms * 1e-3) / 1e12);
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t total = (size_t)1200 << 20;
    u32x4* buf; unsigned* out;
    CK(hipMalloc(&buf, total)); CK(hipMemset(buf, 1, total)); CK(hipMalloc(&out, 4));
    for (size_t per : {(size_t)14 << 20, (size_t)56 << 20, (size_t)600 << 20}) {
        printf("bytes per launch %.0f MB\n", per / 1e6);
        for (int wgs : {144, 256, 512, 1024}) { run<4>(buf, total, per, wgs, st, out); run<12>(buf, total, per, wgs, st, out); }
    }
    return 0;
}
