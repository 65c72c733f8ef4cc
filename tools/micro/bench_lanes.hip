// Micro-benchmark: aggregate throughput of N independent GEMM chains (the AR block's qkv -> proj -> fc1 -> fc2 at M = 64)
// replayed from hipGraphs on N streams, against one chain alone -- what limits several batches in flight?
#include "../../hqtransformer_amd/csrc/fast_kernels.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Chain {
    hipStream_t st; hipGraphExec_t ge;
    float *x, *parts, *colsum, *bias; bf16_t *xpk, *qkv, *hbuf;
};

int main(int argc, char** argv) {
    const int L = 12, M = argc > 1 ? atoi(argv[1]) : 64, D = 1536, reps = 20, maxn = argc > 3 ? atoi(argv[3]) : 4;
    const bool shared_w = argc > 2 ? atoi(argv[2]) != 0 : true;
    const int pk = packed_mb(M), Mpad = pk * 32;
    CK(stream_gemm_configure());
    auto mkw = [&](size_t n) { bf16_t* p; CK(hipMalloc(&p, n * 2)); CK(hipMemset(p, 0x3c, n * 2)); return p; };
    std::vector<std::vector<bf16_t*>> W(maxn);
    for (int c = 0; c < maxn; ++c) {
        if (c > 0 && shared_w) { W[c] = W[0]; continue; }
        for (int l = 0; l < L; ++l) { W[c].push_back(mkw((size_t)3 * D * D)); W[c].push_back(mkw((size_t)D * D)); W[c].push_back(mkw((size_t)4 * D * D)); W[c].push_back(mkw((size_t)4 * D * D)); }
    }
    std::vector<Chain> ch(maxn);
    for (int c = 0; c < maxn; ++c) {
        Chain& k = ch[c];
        CK(hipStreamCreateWithFlags(&k.st, hipStreamNonBlocking));
        CK(hipMalloc(&k.x, (size_t)Mpad * D * 4)); CK(hipMemset(k.x, 0, (size_t)Mpad * D * 4));
        CK(hipMalloc(&k.parts, (size_t)(D / 32) * Mpad * 8)); CK(hipMemset(k.parts, 0, (size_t)(D / 32) * Mpad * 8));
        CK(hipMalloc(&k.colsum, 6144 * 4)); CK(hipMemset(k.colsum, 0, 6144 * 4)); CK(hipMalloc(&k.bias, 6144 * 4)); CK(hipMemset(k.bias, 0, 6144 * 4));
        CK(hipMalloc(&k.xpk, (size_t)Mpad * D * 2)); CK(hipMemset(k.xpk, 0, (size_t)Mpad * D * 2));
        CK(hipMalloc(&k.qkv, (size_t)Mpad * 3 * D * 2)); CK(hipMalloc(&k.hbuf, (size_t)Mpad * 4 * D * 2));
        CK(hipMemset(k.qkv, 0, (size_t)Mpad * 3 * D * 2)); CK(hipMemset(k.hbuf, 0, (size_t)Mpad * 4 * D * 2));
        hipGraph_t graph;
        CK(hipStreamBeginCapture(k.st, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l < L; ++l) {
            GemmArgs g{};
            g.A = k.xpk; g.M = M; g.batch = 1; g.a_packed_mb = pk; g.N = 3 * D; g.K = D; g.alpha = 1.f; g.bias = k.bias;
            g.ln_parts = k.parts; g.ln_nparts = D / 32; g.ln_colsum = k.colsum; g.ln_eps = 1e-5f;
            g.C = k.qkv; g.ldc = 3 * D; g.store = STORE_PACKED; g.c_packed_mb = pk;
            CK(launch_stream_gemm(g, W[c][4 * l], DT_BF16, DT_BF16, 1, nullptr, k.st));
            g = GemmArgs{};
            g.A = k.qkv; g.M = M; g.batch = 1; g.a_packed_mb = pk; g.N = D; g.K = D; g.alpha = 1.f; g.bias = k.bias;
            g.C = k.x; g.ldc = D; g.store = STORE_RESID; g.resid_pk = k.xpk; g.resid_parts = k.parts; g.c_packed_mb = pk;
            CK(launch_stream_gemm(g, W[c][4 * l + 1], DT_BF16, DT_F32, 1, nullptr, k.st));
            g = GemmArgs{};
            g.A = k.xpk; g.M = M; g.batch = 1; g.a_packed_mb = pk; g.N = 4 * D; g.K = D; g.alpha = 1.f; g.bias = k.bias;
            g.ln_parts = k.parts; g.ln_nparts = D / 32; g.ln_colsum = k.colsum; g.ln_eps = 1e-5f;
            g.C = k.hbuf; g.ldc = 4 * D; g.store = STORE_PACKED; g.c_packed_mb = pk; g.act = ACT_GELU_ERF;
            CK(launch_stream_gemm(g, W[c][4 * l + 2], DT_BF16, DT_BF16, 1, nullptr, k.st));
            g = GemmArgs{};
            g.A = k.hbuf; g.M = M; g.batch = 1; g.a_packed_mb = pk; g.N = D; g.K = 4 * D; g.alpha = 1.f; g.bias = k.bias;
            g.C = k.x; g.ldc = D; g.store = STORE_RESID; g.resid_pk = k.xpk; g.resid_parts = k.parts; g.c_packed_mb = pk;
            CK(launch_stream_gemm(g, W[c][4 * l + 3], DT_BF16, DT_F32, 1, nullptr, k.st));
        }
        CK(hipStreamEndCapture(k.st, &graph));
        CK(hipGraphInstantiate(&k.ge, graph, nullptr, nullptr, 0));
        CK(hipGraphDestroy(graph));
        CK(hipGraphLaunch(k.ge, k.st)); CK(hipStreamSynchronize(k.st));
    }
    const double bytes_per_replay = (double)L * 12.0 * D * D * 2;
    for (int n = 1; n <= maxn; ++n) {
        CK(hipDeviceSynchronize());
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a, ch[0].st));
        for (int c = 1; c < n; ++c) CK(hipStreamWaitEvent(ch[c].st, a, 0));
        for (int r = 0; r < reps; ++r)
            for (int c = 0; c < n; ++c) CK(hipGraphLaunch(ch[c].ge, ch[c].st));
        std::vector<hipEvent_t> done(n);
        for (int c = 1; c < n; ++c) { CK(hipEventCreate(&done[c])); CK(hipEventRecord(done[c], ch[c].st)); CK(hipStreamWaitEvent(ch[0].st, done[c], 0)); }
        CK(hipEventRecord(b, ch[0].st)); CK(hipStreamSynchronize(ch[0].st));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double gemms = (double)n * reps * L * 4;
        printf("M=%d %s weights, %d chain(s): %.2f us per GEMM aggregate (%.2f us per GEMM per chain), %.2f TB/s of weight bytes\n", M,
               shared_w ? "shared" : "private", n, 1000.0 * ms / gemms, 1000.0 * ms / (reps * L * 4), n * reps * bytes_per_replay / (ms * 1e-3) / 1e12);
    }
    return 0;
}
