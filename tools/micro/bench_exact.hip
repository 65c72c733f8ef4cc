// Micro-benchmark: exact_mfma_gemm_kernel (fp32 matrix instructions, EXACT / SPLIT nn.Linear up to 256 rows) on the AR loop's shapes at 64 and 256 rows, every row-tile count
// per wave (MT), cold weights (8 rotating copies: 300-600 MB, beyond the Infinity Cache).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/micro/bench_exact.hip -o tools/micro/bench_exact
#include "../../hqtransformer_amd/csrc/exact_gemm.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MT>
static float run(const GemmArgs& g0, float* const* w, int copies, hipStream_t st, int reps) {
    const int t16 = (g0.M + 15) / 16, TN = (g0.N + 15) / 16;
    if (t16 % MT) return -1.f;
    const int TM = t16 / MT;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    GemmArgs g = g0;
    for (int r = -2; r < reps; ++r) {
        if (r == 0) CK(hipEventRecord(a, st));
        g.Bw = w[(r + 2) % copies];
        exact_mfma_gemm_kernel<true, MT><<<TM * TN, 256, 0, st>>>(g, TM, TN);
    }
    CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return 1000.f * ms / reps;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const int D = 1536, copies = 8;
    struct Shape { const char* name; int N, K; } shapes[] = {{"qkv", 3 * D, D}, {"fc1", 4 * D, D}, {"fc2", D, 4 * D}, {"proj", D, D}};
    float *A, *C, *bias; float* w[copies];
    CK(hipMalloc(&A, (size_t)256 * 4 * D * 4)); CK(hipMalloc(&C, (size_t)256 * 4 * D * 4)); CK(hipMalloc(&bias, 4 * D * 4));
    CK(hipMemset(A, 0, (size_t)256 * 4 * D * 4)); CK(hipMemset(bias, 0, 4 * D * 4));
    for (int i = 0; i < copies; ++i) { CK(hipMalloc(&w[i], (size_t)4 * D * D * 4)); CK(hipMemset(w[i], 0, (size_t)4 * D * D * 4)); }
    for (int M : {64, 256}) {
        printf("rows %d: us per launch (weights GB/s) for MT = 1 / 2 / 4 row tiles per wave; fp32 matrix floor = rows x N x K x 2 / 157 TFLOP/s\n", M);
        for (auto& s : shapes) {
            GemmArgs g{};
            g.A = A; g.lda = s.K; g.ldb = s.K; g.C = C; g.ldc = s.N; g.M = M; g.N = s.N; g.K = s.K; g.batch = 1; g.bias = bias; g.alpha = 1.f; g.store = STORE_ROWS;
            g.k_quarters = 1; g.b_tile16 = 1;
            const float t1 = run<1>(g, w, copies, st, 40), t2 = run<2>(g, w, copies, st, 40), t4 = run<4>(g, w, copies, st, 40);
            const double wb = (double)s.N * s.K * 4;
            printf("%-5s N %5d K %5d: %6.1f (%4.0f)  %6.1f (%4.0f)  %6.1f (%4.0f) | floor %5.1f us, weights at 5 TB/s %5.1f us\n", s.name, s.N, s.K, t1, wb / t1 * 1e-3, t2, wb / t2 * 1e-3,
                   t4, wb / t4 * 1e-3, 2.0 * M * s.N * s.K / 157e6, wb / 5e6);
        }
    }
    return 0;
}
