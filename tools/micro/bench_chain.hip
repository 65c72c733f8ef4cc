// Micro-benchmark: what a small kernel (LayerNorm, attention) costs alone, back to back, and in between
// weight-streaming GEMMs (cold caches), inside a hipGraph -- to separate kernel time from boundary effects.
#include "../../hqtransformer_amd/csrc/fast_kernels.hip"
#include "../../hqtransformer_amd/csrc/kernels.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename F>
static float graph_time(hipStream_t st, int reps, F body) {
    hipGraph_t graph; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    body();
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, st));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(graph));
    return 1000.f * ms / reps;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    CK(stream_gemm_configure());
    const int D = 1536, M = 64, nh = 24, T = 64;
    float *x, *gam, *bet; bf16_t *h, *q, *kc, *vc, *o; float* y;
    CK(hipMalloc(&x, 256 * D * 4)); CK(hipMemset(x, 0, 256 * D * 4));
    CK(hipMalloc(&gam, D * 4)); CK(hipMemset(gam, 0, D * 4)); CK(hipMalloc(&bet, D * 4)); CK(hipMemset(bet, 0, D * 4));
    CK(hipMalloc(&h, 256 * 6144 * 2)); CK(hipMalloc(&q, 256 * D * 2)); CK(hipMalloc(&o, 256 * D * 2));
    CK(hipMemset(q, 0, 256 * D * 2));
    CK(hipMalloc(&kc, (size_t)64 * 128 * D * 2)); CK(hipMalloc(&vc, (size_t)64 * 128 * D * 2));
    CK(hipMemset(kc, 0, (size_t)64 * 128 * D * 2)); CK(hipMemset(vc, 0, (size_t)64 * 128 * D * 2));
    CK(hipMalloc(&y, 256 * 8192 * 4));
    const int N = 4608, K = 1536, nbuf = 40;
    std::vector<bf16_t*> w(nbuf);
    for (auto& p : w) { CK(hipMalloc(&p, (size_t)N * K * 2)); CK(hipMemset(p, 0x3c, (size_t)N * K * 2)); }
    LNArgs ln{x, gam, bet, nullptr, h, M, D, 1, 0, 1e-5f, DT_BF16, 2, nullptr, 0, 0, nullptr, nullptr, 0, nullptr};
    AttnArgs at{q, kc, vc, o, 64, 1, nh, 64, 128, T - 1, nullptr, 1, DT_BF16, 2, nullptr};
    GemmArgs g{};
    g.A = h; g.a_packed_mb = 2; g.M = M; g.N = N; g.K = K; g.batch = 1; g.C = y; g.ldc = N; g.alpha = 1.f; g.store = STORE_ROWS;
    CK(sampler_configure(8192, false));
    const int n = 40;
    float t;
    t = graph_time(st, 5, [&] { for (int i = 0; i < n; ++i) CK(launch_layernorm(ln, st)); });
    printf("LayerNorm x%d back to back: %.2f us each\n", n, t / n);
    {
        LNArgs z = ln; z.M = 0;       // every wave exits at once: launch + kernarg cost of this geometry
        t = graph_time(st, 5, [&] { for (int i = 0; i < n; ++i) layernorm_kernel<bf16_t><<<16, 256, 0, st>>>(z); });
        printf("LayerNorm geometry (16 x 256 threads), immediate exit: %.2f us each\n", t / n);
        LNArgs one = ln; one.M = 4;
        t = graph_time(st, 5, [&] { for (int i = 0; i < n; ++i) CK(launch_layernorm(one, st)); });
        printf("LayerNorm with 4 rows (1 workgroup): %.2f us each\n", t / n);
        LNArgs f32 = ln; f32.out_dtype = DT_F32; f32.out_packed_mb = 0; f32.y = y;
        t = graph_time(st, 5, [&] { for (int i = 0; i < n; ++i) CK(launch_layernorm(f32, st)); });
        printf("LayerNorm fp32 row-major output: %.2f us each\n", t / n);
    }
    t = graph_time(st, 5, [&] { for (int i = 0; i < n; ++i) CK(launch_attention(at, st)); });
    printf("attention (B=64, 24 heads, %d keys) x%d back to back: %.2f us each\n", T, n, t / n);
    t = graph_time(st, 5, [&] { for (int i = 0; i < n; ++i) CK(launch_stream_gemm(g, w[i], DT_BF16, DT_F32, 1, nullptr, st)); });
    const float tg = t / n;
    printf("stream GEMM qkv (cold weights) x%d: %.2f us each\n", n, tg);
    t = graph_time(st, 5, [&] { for (int i = 0; i < n; ++i) { CK(launch_layernorm(ln, st)); CK(launch_stream_gemm(g, w[i], DT_BF16, DT_F32, 1, nullptr, st)); } });
    printf("LayerNorm + GEMM alternating: %.2f us per pair -> LayerNorm costs %.2f us in the chain\n", t / n, t / n - tg);
    t = graph_time(st, 5, [&] { for (int i = 0; i < n; ++i) { CK(launch_attention(at, st)); CK(launch_stream_gemm(g, w[i], DT_BF16, DT_F32, 1, nullptr, st)); } });
    printf("attention + GEMM alternating: %.2f us per pair -> attention costs %.2f us in the chain\n", t / n, t / n - tg);
    {   // the real pattern: split-K GEMM (fc2-like, S = 2, writes slabs) -> LayerNorm folding the slabs -> wide GEMM
        float* slabs; CK(hipMalloc(&slabs, 4ull * 64 * D * 4)); CK(hipMemset(slabs, 0, 4ull * 64 * D * 4));
        GemmArgs g2{};
        g2.A = h; g2.a_packed_mb = 2; g2.M = M; g2.N = D; g2.K = D; g2.batch = 1; g2.C = x; g2.ldc = D; g2.alpha = 1.f; g2.store = STORE_ROWS;
        LNArgs lf = ln; lf.slabs = slabs; lf.n_slabs = 2; lf.slab_rows = 64; lf.slab_bias = bet;
        float t2 = graph_time(st, 5, [&] { for (int i = 0; i < n; ++i) { CK(launch_stream_gemm(g2, w[i], DT_BF16, DT_F32, 2, slabs, st)); CK(launch_stream_gemm(g, w[(i + 7) % n], DT_BF16, DT_F32, 1, nullptr, st)); } });
        float t3 = graph_time(st, 5, [&] { for (int i = 0; i < n; ++i) { CK(launch_stream_gemm(g2, w[i], DT_BF16, DT_F32, 2, slabs, st)); CK(launch_layernorm(lf, st)); CK(launch_stream_gemm(g, w[(i + 7) % n], DT_BF16, DT_F32, 1, nullptr, st)); } });
        printf("proj(S=2) -> qkv: %.2f us per pair; proj(S=2) -> LN(fold 2 slabs) -> qkv: %.2f us -> LayerNorm costs %.2f us in the real chain\n", t2 / n, t3 / n, (t3 - t2) / n);
        LNArgs l256 = lf; l256.M = 256; l256.out_packed_mb = 8; l256.slab_rows = 256; l256.n_slabs = 0; l256.slabs = nullptr;
        t = graph_time(st, 5, [&] { for (int i = 0; i < n; ++i) CK(launch_layernorm(l256, st)); });
        printf("LayerNorm M=256 back to back: %.2f us each\n", t / n);
    }
    t = graph_time(st, 5, [&] { for (int i = 0; i < n; ++i) { CK(launch_layernorm(ln, st)); CK(launch_attention(at, st)); CK(launch_stream_gemm(g, w[i], DT_BF16, DT_F32, 1, nullptr, st)); } });
    printf("LN + attention + GEMM: %.2f us per triple\n", t / n);
    {   // in-kernel stamps of the attention kernel behind a GEMM (cold caches), by number of keys; device step state like the product
        long long* dbg; CK(hipMalloc(&dbg, (size_t)64 * nh * 8 * 8));
        int* tb; CK(hipMalloc(&tb, 4));
        for (int keys : {1, 8, 32, 64}) {
            const int tbv = keys - 1;
            CK(hipMemcpy(tb, &tbv, 4, hipMemcpyHostToDevice));
            AttnArgs ad = at; ad.t_base = 0; ad.t_base_dev = tb; ad.dbg = dbg;
            for (int rep = 0; rep < 3; ++rep) { CK(launch_stream_gemm(g, w[rep], DT_BF16, DT_F32, 1, nullptr, st)); CK(launch_attention(ad, st)); }
            CK(hipStreamSynchronize(st));
            std::vector<long long> hst((size_t)64 * nh * 8);
            CK(hipMemcpy(hst.data(), dbg, hst.size() * 8, hipMemcpyDeviceToHost));
            double s[5] = {0, 0, 0, 0, 0}; long long w0 = hst[0], w1 = hst[0];
            const int nw = 64 * nh;
            for (int i = 0; i < nw; ++i) { for (int k = 1; k <= 4; ++k) s[k] += (double)hst[(size_t)i * 8 + k]; w0 = std::min(w0, hst[(size_t)i * 8]); w1 = std::max(w1, hst[(size_t)i * 8]); }
            AttnArgs plain = at; plain.t_base = keys - 1;
            const float tt = graph_time(st, 5, [&] { for (int i = 0; i < n; ++i) { CK(launch_attention(plain, st)); CK(launch_stream_gemm(g, w[i], DT_BF16, DT_F32, 1, nullptr, st)); } }) / n - tg;
            printf("attention %2d keys: %.2f us in the chain; mean cycles from wave start: loads issued %.0f, scores done %.0f, softmax done %.0f, end %.0f; wave ends spread over %.2f us\n",
                   keys, tt, s[1] / nw, s[2] / nw, s[3] / nw, s[4] / nw, (w1 - w0) / 100.0);
        }
    }
    return 0;
}
