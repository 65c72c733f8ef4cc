// Micro-benchmark: the LDS-DMA implicit-GEMM conv kernel on the HQ-VAE decoder's layer shapes (batch 64), with
// ablations that separate the epilogue, the DMA/address generation and the MFMA/LDS-read loop.
#include "../../hqtransformer_amd/csrc/mfma_gemm.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Shape { const char* name; int res, cin, cout, taps, up; };

template <int ABL>
static float run(const GemmArgs& g, hipStream_t st, int reps) {
    const dim3 grid((g.N + 127) / 128, (g.M + 127) / 128, 1);
    const size_t smem = 2 * 2 * 128 * 128;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_glds_kernel<bf16_t, 0, 2, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    conv_glds_kernel<bf16_t, 0, 2, ABL><<<grid, 256, smem, st>>>(g);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, st));
    for (int r = 0; r < reps; ++r) conv_glds_kernel<bf16_t, 0, 2, ABL><<<grid, 256, smem, st>>>(g);
    CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return 1000.f * ms / reps;
}

template <int ABL, int TY = 8>
static float run_halo(const GemmArgs& g, hipStream_t st, int reps) {
    const dim3 grid((g.N + 127) / 128, g.M / (TY * 16), 1);
    constexpr int HALO_LDS = halo_lds(TY);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<bf16_t, false, TY, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, HALO_LDS));
    conv3x3_halo_kernel<bf16_t, false, TY, ABL><<<grid, 256, HALO_LDS, st>>>(g);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, st));
    for (int r = 0; r < reps; ++r) conv3x3_halo_kernel<bf16_t, false, TY, ABL><<<grid, 256, HALO_LDS, st>>>(g);
    CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return 1000.f * ms / reps;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64;
    hipStream_t st; CK(hipStreamCreate(&st));
    const Shape shapes[] = {
        {"res16  512->512 3x3", 16, 512, 512, 9, 0}, {"res32  512->512 3x3", 32, 512, 512, 9, 0}, {"up64   512->512 3x3 (x2)", 64, 512, 512, 9, 1},
        {"res64  256->256 3x3", 64, 256, 256, 9, 0}, {"up128  256->256 3x3 (x2)", 128, 256, 256, 9, 1}, {"res128 128->128 3x3", 128, 128, 128, 9, 0},
        {"up256  128->128 3x3 (x2)", 256, 128, 128, 9, 1}, {"nin64  512->256 1x1", 64, 512, 256, 1, 0},
    };
    size_t amax = (size_t)B * 256 * 256 * 128, wmax = (size_t)512 * 9 * 512;
    bf16_t *A, *Wt, *C; float* bias; void* zero;
    CK(hipMalloc(&A, amax * 2)); CK(hipMalloc(&C, amax * 2)); CK(hipMalloc(&Wt, wmax * 2)); CK(hipMalloc(&bias, 512 * 4)); CK(hipMalloc(&zero, 256));
    CK(hipMemset(A, 0x3c, amax * 2)); CK(hipMemset(Wt, 0x3c, wmax * 2)); CK(hipMemset(bias, 0, 512 * 4)); CK(hipMemset(zero, 0, 256));
    printf("%-28s %9s | %8s %8s | %8s %8s %8s %8s\n", "layer (batch 64)", "GFLOP", "us", "TFLOP/s", "no-epi", "mfma-only", "dma-only", "no-addr");
    double tot_us = 0, tot_fl = 0, tot_halo = 0;
    for (const Shape& s : shapes) {
        GemmArgs g{};
        g.A = A; g.conv_taps = s.taps; g.H = s.res; g.W = s.res; g.Cin = s.cin; g.upsample = s.up;
        g.Bw = Wt; g.ldb = s.taps * s.cin; g.C = C; g.ldc = s.cout; g.M = B * s.res * s.res; g.N = s.cout; g.K = s.taps * s.cin; g.batch = 1;
        g.bias = bias; g.alpha = 1.f; g.store = STORE_ROWS; g.zero_page = zero; g.lda = s.cin;
        const double fl = 2.0 * g.M * g.N * g.K;
        const int reps = 5;
        const float t0 = run<0>(g, st, reps), t1 = run<1>(g, st, reps), t2 = run<2>(g, st, reps), t3 = run<3>(g, st, reps), t4 = run<4>(g, st, reps);
        printf("%-28s %9.1f | %8.1f %8.1f | %8.1f %8.1f %8.1f %8.1f\n", s.name, fl * 1e-9, t0, fl / t0 * 1e-6, t1, t2, t3, t4);
        if (s.taps == 9) {
            const float h0 = run_halo<0>(g, st, reps), h1 = run_halo<1>(g, st, reps), h2 = run_halo<2>(g, st, reps), h3 = run_halo<3>(g, st, reps), h5 = run_halo<5>(g, st, reps), h6 = run_halo<6>(g, st, reps), h7 = run_halo<7>(g, st, reps);
            printf("%-28s %9s | %8.1f %8.1f | %8.1f %8.1f %8.1f   no-global-store %8.1f no-staging %8.1f staging-only %8.1f\n", "   halo tile", "", h0, fl / h0 * 1e-6, h1, h2, h3, h5, h6, h7);
            const float w0 = run_halo<0, 16>(g, st, reps), w1 = run_halo<1, 16>(g, st, reps), w2 = run_halo<2, 16>(g, st, reps), w3 = run_halo<3, 16>(g, st, reps);
            printf("%-28s %9s | %8.1f %8.1f | %8.1f %8.1f %8.1f\n", "   halo tile 16x16", "", w0, fl / w0 * 1e-6, w1, w2, w3);
            tot_halo += std::min(h0, w0);
        } else tot_halo += t0;
        tot_us += t0; tot_fl += fl;
    }
    printf("sum: %.1f us, %.1f TFLOP/s; with the halo kernel on the 3x3 layers: %.1f us, %.1f TFLOP/s\n", tot_us, tot_fl / tot_us * 1e-6,
           tot_halo, tot_fl / tot_halo * 1e-6);
    return 0;
}
