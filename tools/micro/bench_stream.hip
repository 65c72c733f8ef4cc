// Micro-benchmark of stream_gemm_kernel variants on the AR loop's real shapes (cold weights: buffers rotate
// through > 256 MiB so neither L2 nor the Infinity Cache holds them between uses).
#include "../../hqtransformer_amd/csrc/fast_kernels.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Shape { const char* name; int N, K; };

template <int MBW, int NT, int NW, int U, int ABL = 0, bool PIPE = false>
static float run(const Shape& sh, int M, int S, const std::vector<bf16_t*>& wbufs, bf16_t* x, float* y, float* slabs, hipStream_t st) {
    const int MB = packed_mb(M);
    if (MB % MBW != 0 || sh.N % (32 * NT) != 0) return -1.f;
    const size_t smem = stream_gemm_lds(MBW, NT, NW);
    if (smem > 160 * 1024) return -1.f;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_gemm_kernel<MBW, NT, NW, U, float, ABL, PIPE>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    GemmArgs g{};
    g.A = x; g.a_packed_mb = MB; g.M = M; g.N = sh.N; g.K = sh.K; g.batch = 1; g.C = y; g.ldc = sh.N; g.alpha = 1.f; g.store = STORE_ROWS;
    const dim3 grid(sh.N / (32 * NT), MB / MBW, S);
    const int iters = (int)wbufs.size();
    hipGraph_t graph; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < iters; ++i)
        stream_gemm_kernel<MBW, NT, NW, U, float, ABL, PIPE><<<grid, NW * 64, smem, st>>>(g, reinterpret_cast<const u32x4*>(wbufs[i]), slabs);
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, st));
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(graph));
    return 1000.f * ms / (3 * iters);
}

template <int MBW, int NT, int NW, int U, bool PIPE = false>
static void stamps(const Shape& sh, int M, int S, bf16_t* w, bf16_t* x, float* y, float* slabs, hipStream_t st) {
    const int MB = packed_mb(M);
    const size_t smem = stream_gemm_lds(MBW, NT, NW);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_gemm_kernel<MBW, NT, NW, U, float, 9, PIPE>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    GemmArgs g{};
    g.A = x; g.a_packed_mb = MB; g.M = M; g.N = sh.N; g.K = sh.K; g.batch = 1; g.C = y; g.ldc = sh.N; g.alpha = 1.f; g.store = STORE_ROWS;
    const dim3 grid(sh.N / (32 * NT), MB / MBW, S);
    const int nwg = grid.x * grid.y * grid.z;
    for (int rep = 0; rep < 2; ++rep) {
        stream_gemm_kernel<MBW, NT, NW, U, float, 9, PIPE><<<grid, NW * 64, smem, st>>>(g, reinterpret_cast<const u32x4*>(w), slabs);
        CK(hipStreamSynchronize(st));
    }
    std::vector<long long> h((size_t)nwg * NW * 8);
    CK(hipMemcpy(h.data(), slabs, h.size() * 8, hipMemcpyDeviceToHost));
    long long t0 = h[0], t1 = h[5];
    double s[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < nwg * NW; ++i) {
        t0 = std::min(t0, h[i * 8]); t1 = std::max(t1, h[i * 8 + 5]);
        for (int k = 1; k <= 4; ++k) s[k] += (double)h[i * 8 + k];
    }
    long long last_start = t0;
    for (int i = 0; i < nwg * NW; ++i) last_start = std::max(last_start, h[i * 8]);
    printf("   stamps MBW=%d NT=%d NW=%d U=%d S=%d (%d WGs): kernel span %.2f us (first wave start -> last wave end), last wave starts at +%.2f us;\n"
           "      mean cycles from wave start: loads issued %.0f, MFMAs done %.0f, after barrier %.0f, end %.0f\n",
           MBW, NT, NW, U, S, nwg, (t1 - t0) / 100.0, (last_start - t0) / 100.0, s[1] / (nwg * NW), s[2] / (nwg * NW), s[3] / (nwg * NW), s[4] / (nwg * NW));
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const Shape shapes[] = {{"qkv", 4608, 1536}, {"proj", 1536, 1536}, {"fc1", 6144, 1536}, {"fc2", 1536, 6144}, {"head", 8192, 1536}};
    bf16_t* x; float *y, *slabs;
    CK(hipMalloc(&x, 2048 * 6144 * 2)); CK(hipMemset(x, 0x3c, 2048 * 6144 * 2));
    CK(hipMalloc(&y, 2048ull * 8192 * 4)); CK(hipMalloc(&slabs, 4ull * 2048 * 8192 * 4));
    for (const Shape& sh : shapes) {
        const size_t bytes = (size_t)sh.N * sh.K * 2;
        const int nbuf = (int)(((size_t)600 << 20) / bytes) + 1;
        std::vector<bf16_t*> w(nbuf);
        for (auto& p : w) { CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 0x3c, bytes)); }
        for (int M : {64, 256, 512, 1024, 2048}) {
            printf("== %s N=%d K=%d M=%d  (%.1f MB weights; HBM floor %.2f us @6.3TB/s)\n", sh.name, sh.N, sh.K, M, bytes / 1e6, bytes / 6.3e6);
#define V(MBW, NT, NW, U, S) { float t = run<MBW, NT, NW, U>(sh, M, S, w, x, y, slabs, st); if (t > 0) printf("   MBW=%d NT=%d NW=%d U=%2d S=%d : %7.2f us  (%.2f TB/s)\n", MBW, NT, NW, U, S, t, bytes / t / 1e6); }
            if (M == 64) {      // the same launches with ONE weight buffer (L2 / Infinity Cache warm): what a cross-kernel weight prefetch could buy at most
                std::vector<bf16_t*> same(w.size(), w[0]);
                const float c1 = run<2, 1, 8, 12>(sh, M, 1, w, x, y, slabs, st), h1 = run<2, 1, 8, 12>(sh, M, 1, same, x, y, slabs, st);
                const float c2 = run<1, 1, 8, 12>(sh, M, 1, w, x, y, slabs, st), h2 = run<1, 1, 8, 12>(sh, M, 1, same, x, y, slabs, st);
                const float c3 = run<2, 1, 8, 6>(sh, M, 1, w, x, y, slabs, st), h3 = run<2, 1, 8, 6>(sh, M, 1, same, x, y, slabs, st);
                printf("   cold -> warm weights: <2,1,8,12> %.2f -> %.2f us, <1,1,8,12> %.2f -> %.2f us, <2,1,8,6> %.2f -> %.2f us\n", c1, h1, c2, h2, c3, h3);
            }
            if (M == 64 && false) {} if (M == 64) { V(2, 1, 8, 12, 1) V(1, 1, 8, 12, 1) V(1, 1, 16, 12, 1) V(1, 1, 16, 6, 1) V(2, 1, 16, 6, 1) V(2, 1, 16, 4, 1) V(1, 1, 8, 6, 1) V(2, 1, 8, 4, 1) }
            if (M == 64 && false) { stamps<2, 1, 8, 12>(sh, M, 1, w[0], x, y, slabs, st); stamps<1, 1, 8, 12>(sh, M, 1, w[1], x, y, slabs, st); }
            if (M == 512 || M == 2048) {
                const float e0 = run<2, 2, 4, 4, 0, true>(sh, M, 1, w, x, y, slabs, st), e3 = run<2, 2, 4, 4, 3, true>(sh, M, 1, w, x, y, slabs, st);
                const float e6 = run<2, 2, 4, 4, 6, true>(sh, M, 1, w, x, y, slabs, st), e7 = run<2, 2, 4, 4, 7, true>(sh, M, 1, w, x, y, slabs, st);
                printf("   <2,2,4,4> pipelined: full %.2f us, without the epilogue (ABL 3) %.2f us, without the output stores %.2f, without the cross-wave sum reads %.2f\n", e0, e3, e6, e7);
            }
            if (M == 512 || M == 2048) { stamps<2, 2, 4, 4, true>(sh, M, 1, w[0], x, y, slabs, st); stamps<2, 2, 8, 6, false>(sh, M, 1, w[1], x, y, slabs, st); }
#define P(MBW, NT, NW, U, S) { float t = run<MBW, NT, NW, U, 0, true>(sh, M, S, w, x, y, slabs, st); if (t > 0) printf("   MBW=%d NT=%d NW=%d D=%2d S=%d pipelined : %7.2f us  (%.2f TB/s, %.0f TFLOP/s)\n", MBW, NT, NW, U, S, t, bytes / t / 1e6, 2.0 * M * sh.N * sh.K / t / 1e6); }
            if (M >= 256) { P(2, 2, 8, 4, 1) P(2, 2, 8, 3, 1) P(2, 2, 8, 6, 1) P(2, 2, 4, 4, 1) P(2, 2, 4, 6, 1) P(2, 3, 4, 4, 1) P(2, 3, 4, 6, 1) P(2, 1, 8, 6, 1) P(2, 1, 8, 4, 1) P(4, 2, 4, 3, 1) P(4, 2, 4, 4, 1) P(4, 1, 8, 4, 1) P(4, 1, 4, 6, 1) P(2, 3, 8, 4, 1) P(2, 2, 4, 4, 2) P(2, 2, 8, 4, 2) }
            if (M >= 512) { V(2, 2, 8, 6, 1) V(4, 2, 4, 3, 1) V(2, 1, 8, 6, 1) V(2, 1, 8, 12, 1) V(2, 2, 4, 6, 1) }
            else if (M == 256) { V(1, 1, 8, 12, 1) V(2, 1, 8, 12, 1) V(2, 1, 8, 4, 1) V(2, 1, 8, 6, 1) V(2, 1, 16, 4, 1) V(4, 1, 8, 3, 1) V(4, 1, 8, 4, 1) V(4, 1, 4, 4, 1) V(4, 1, 16, 2, 1) V(2, 2, 8, 4, 1) }
        }
        for (auto& p : w) CK(hipFree(p));
    }
    return 0;
}
