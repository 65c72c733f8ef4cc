#include "fast_kernels.h"
#include "gemm_generic.h"

__global__ void f32_to_bf16_kernel(const float* src, bf16_t* dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = f32_to_bf16(src[i]);
}
hipError_t launch_f32_to_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t st) {
    const int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
    f32_to_bf16_kernel<<<grid, 256, 0, st>>>(src, dst, n);
    return hipGetLastError();
}

__global__ void repack_conv_kernel(const float* src, float* dst, int O, int I, int taps) {
    const size_t n = (size_t)O * I * taps;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ii = (int)(i % I);
        const int t = (int)((i / I) % taps);
        const size_t o = i / ((size_t)I * taps);
        dst[i] = src[(o * I + ii) * taps + t];
    }
}
hipError_t launch_repack_conv(const float* src, float* dst, int O, int I, int taps, hipStream_t st) {
    const size_t n = (size_t)O * I * taps;
    const int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
    repack_conv_kernel<<<grid, 256, 0, st>>>(src, dst, O, I, taps);
    return hipGetLastError();
}

bool stream_gemm_supported(int, int) { return false; }
hipError_t launch_pack_stream_weights(const float*, bf16_t*, int, int, hipStream_t) { return hipErrorNotSupported; }
bool stream_gemm_ok(const GemmArgs&, int, int) { return false; }
hipError_t launch_stream_gemm(const GemmArgs&, const bf16_t*, int, int, float*, size_t, hipStream_t) { return hipErrorNotSupported; }
bool mfma_gemm_ok(const GemmArgs&, int, int, int) { return false; }
hipError_t launch_mfma_gemm(const GemmArgs&, int, int, int, hipStream_t) { return hipErrorNotSupported; }
