// Micro-benchmark + self-check of the LDS-tiled MFMA GEMM (hqtransformer_amd/csrc/tile_gemm.hip) at the shapes of the merged AR passes.
//   bench_tile [check]      check: compare every geometry / store mode against a naive fp32-accumulate reference first
#define HQT_TILE_STAMPS 1
#include "../../hqtransformer_amd/csrc/fast_kernels.hip"
#include "../../hqtransformer_amd/csrc/tile_gemm.hip"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <functional>
#include <climits>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void pack_rows_kernel(const bf16_t* rows, bf16_t* pk, int M, int K, int MB) {
    const size_t total = (size_t)M * K;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / K), k = (int)(i % K);
        pk[packed_off(m, k, MB)] = rows[i];
    }
}
__global__ void ref_gemm_kernel(const bf16_t* A, const float* W, float* C, int M, int N, int K) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), m = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (m >= M || n >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += bf16_to_f32(A[(size_t)m * K + k]) * bf16_to_f32(f32_to_bf16(W[(size_t)n * K + k]));
    C[(size_t)m * N + n] = s;
}
__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned s = (unsigned)i * 2654435761u + seed;
        s ^= s >> 15; s *= 2246822519u; s ^= s >> 13; s *= 3266489917u; s ^= s >> 16;
        p[i] = ((int)(s & 0xffff) - 32768) * (scale / 32768.f);
    }
}
__global__ void fill_bf16_kernel(bf16_t* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned s = (unsigned)i * 2654435761u + seed;
        s ^= s >> 15; s *= 2246822519u; s ^= s >> 13; s *= 3266489917u; s ^= s >> 16;
        p[i] = f32_to_bf16(((int)(s & 0xffff) - 32768) * (scale / 32768.f));
    }
}

static float time_us(hipStream_t st, int n_launch, const std::function<void(int)>& launch) {
    hipGraph_t graph; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n_launch; ++i) launch(i);
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, st));
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(graph));
    return 1000.f * ms / (3 * n_launch);
}

struct Shape { const char* name; int N, K; };
typedef TileGeom<2, 4, 4, 2, 4, 2> Tile256;       // 256 x 256, 8 waves of 128 x 64, 2 stages of 64 k: 128 KiB ring, one workgroup per CU
typedef TileGeom<4, 2, 2, 2, 4, 3> Tile256x128;   // 256 x 128, 8 waves of 64 x 64, 3 stages of 64 k: 144 KiB ring, one workgroup per CU
typedef TileGeom<2, 4, 4, 2, 2, 4> Tile256k3;     // 256 x 256, 4 stages of 32 k
typedef TileGeom<4, 2, 2, 2, 2, 4> Tile256x128k4; // 256 x 128, 4 stages of 32 k
typedef TileGeom<2, 2, 2, 2, 4, 2> Tile128k4;     // 128 x 128, 2 stages of 64 k
typedef TileGeom<2, 2, 2, 2, 2, 4> Tile128s3;     // 128 x 128, 4 stages of 32 k: two workgroups per CU
typedef TileGeom<2, 2, 2, 2, 4, 3, 4> Tile128PCk4;   // loader-wave geometry with 3 stages of 64 k (96 KiB ring, one workgroup per CU)
typedef TileGeom<2, 2, 2, 2, 2, 6, 4> Tile128PCs6;   // ... with 6 stages of 32 k (96 KiB ring)
typedef TileGeom<2, 2, 2, 2, 2, 4, 2> Tile128PC2;    // ... with two loader waves

struct Bufs {
    bf16_t *x_rows, *xpk, *cpk, *respk;
    float *y, *ref, *x32, *parts, *parts_out, *colsum, *bias, *slabs;
};

template <class G, int STORE, bool DLN, typename TC, int ACT = ACT_NONE>
static void launch_raw(GemmArgs g, const bf16_t* w, int S, float* slabs, hipStream_t st) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(tile_gemm_kernel<G, STORE, DLN, TC, ACT>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    CK((launch_tile_t<G, STORE, DLN, TC, ACT>(g, w, S, slabs, st)));
}

int main(int argc, char** argv) {
    const bool check = argc > 1 && !strcmp(argv[1], "check");
    hipStream_t st; CK(hipStreamCreate(&st));
    const int MMAX = 8192, NMAX = 8192, KMAX = 6144;
    Bufs b;
    CK(hipMalloc(&b.x_rows, (size_t)MMAX * KMAX * 2)); CK(hipMalloc(&b.xpk, (size_t)MMAX * KMAX * 2)); CK(hipMalloc(&b.cpk, (size_t)MMAX * NMAX * 2));
    CK(hipMalloc(&b.respk, (size_t)MMAX * 1536 * 2));
    CK(hipMalloc(&b.y, (size_t)MMAX * NMAX * 4)); CK(hipMalloc(&b.ref, (size_t)MMAX * NMAX * 4)); CK(hipMalloc(&b.x32, (size_t)MMAX * 1536 * 4));
    CK(hipMalloc(&b.parts, (size_t)64 * MMAX * 8)); CK(hipMalloc(&b.parts_out, (size_t)64 * MMAX * 8));
    CK(hipMalloc(&b.colsum, NMAX * 4)); CK(hipMalloc(&b.bias, NMAX * 4)); CK(hipMalloc(&b.slabs, (size_t)8 * MMAX * 1536 * 4));
    fill_bf16_kernel<<<1024, 256, 0, st>>>(b.x_rows, (size_t)MMAX * KMAX, 17u, 1.0f);
    fill_kernel<<<64, 256, 0, st>>>(b.bias, NMAX, 5u, 0.5f);
    fill_kernel<<<64, 256, 0, st>>>(b.colsum, NMAX, 6u, 0.5f);
    fill_kernel<<<1024, 256, 0, st>>>(b.parts, (size_t)64 * MMAX * 2, 7u, 1.0f);
    fill_kernel<<<1024, 256, 0, st>>>(b.x32, (size_t)MMAX * 1536, 8u, 1.0f);
    CK(tile_gemm_configure()); CK(stream_gemm_configure());
    const Shape shapes[] = {{"qkv", 4608, 1536}, {"proj", 1536, 1536}, {"fc1", 6144, 1536}, {"fc2", 1536, 6144}, {"head", 8192, 1536}};
    float* w32; CK(hipMalloc(&w32, (size_t)NMAX * KMAX * 4));
    long long* dbg; CK(hipMalloc(&dbg, (size_t)8192 * 64));
    unsigned* tile_ctr; CK(hipMalloc(&tile_ctr, TILE_CTR_MAX * 4)); CK(hipMemset(tile_ctr, 0, TILE_CTR_MAX * 4));

    auto make_args = [&](int M, const Shape& sh) {
        GemmArgs g{};
        g.A = b.xpk; g.a_packed_mb = packed_mb(M); g.M = M; g.N = sh.N; g.K = sh.K; g.batch = 1; g.alpha = 1.0f;
        g.C = b.y; g.ldc = sh.N; g.store = STORE_ROWS;
        return g;
    };

    if (check) {
        // every geometry against the naive reference (plain fp32 rows), then the store modes of the default geometry
        for (const Shape& sh : {shapes[1], shapes[3], shapes[0]}) {
            for (int M : {512, 1280}) {
                const int MB = packed_mb(M);
                fill_kernel<<<1024, 256, 0, st>>>(w32, (size_t)sh.N * sh.K, 99u, 0.05f);
                bf16_t* wpk; CK(hipMalloc(&wpk, (size_t)sh.N * sh.K * 2));
                CK(launch_pack_stream_weights(w32, wpk, sh.N, sh.K, st));
                CK(hipMemsetAsync(b.xpk, 0xff, (size_t)MB * 32 * sh.K * 2, st));
                pack_rows_kernel<<<1024, 256, 0, st>>>(b.x_rows, b.xpk, M, sh.K, MB);
                ref_gemm_kernel<<<dim3(sh.N / 64, (M + 3) / 4), 256, 0, st>>>(b.x_rows, w32, b.ref, M, sh.N, sh.K);
                std::vector<float> hr((size_t)M * sh.N), hy((size_t)M * sh.N);
                CK(hipMemcpyAsync(hr.data(), b.ref, hr.size() * 4, hipMemcpyDeviceToHost, st));
                auto cmp = [&](const char* what) {
                    CK(hipMemcpyAsync(hy.data(), b.y, hy.size() * 4, hipMemcpyDeviceToHost, st));
                    CK(hipStreamSynchronize(st));
                    double worst = 0, big = 0;
                    for (size_t i = 0; i < hr.size(); ++i) { worst = std::max(worst, (double)std::fabs(hr[i] - hy[i])); big = std::max(big, (double)std::fabs(hr[i])); }
                    printf("check %-5s M=%4d %-28s max|err| %.3e (max|ref| %.2f) %s\n", sh.name, M, what, worst, big, worst <= 2e-3 * big ? "ok" : "FAIL");
                };
                GemmArgs g = make_args(M, sh);
#define RUN(G, what) { CK(hipMemsetAsync(b.y, 0xff, (size_t)M * sh.N * 4, st)); launch_raw<G, TS_ROWS, false, float>(g, wpk, 1, nullptr, st); cmp(what); }
                RUN(Tile128PC, "128x128 k32 x4 + 4 loaders") RUN(Tile256, "256x256 k64 x2") RUN(Tile256k3, "256x256 k64 x2") RUN(Tile128, "128x128 k32 x3") RUN(Tile128k4, "128x128 k64 x2") RUN(Tile256x128, "256x128 k64 x3") RUN(Tile256x128k4, "256x128 k64 x3")
#undef RUN
                // split-K slabs + combine against ref + x32 (+ bias)
                if (sh.N == 1536) {
                    for (int S : {2, 4}) {
                        GemmArgs gr = g;
                        gr.C = b.y; gr.store = STORE_RESID; gr.resid_pk = b.respk; gr.resid_parts = b.parts_out; gr.c_packed_mb = MB; gr.bias = b.bias;
                        CK(hipMemcpyAsync(b.y, b.x32, (size_t)M * sh.N * 4, hipMemcpyDeviceToDevice, st));
                        launch_raw<Tile256, TS_SLAB, false, float>(gr, wpk, S, b.slabs, st);
                        CK(launch_resid_combine(gr, b.slabs, S, st));
                        std::vector<float> hx((size_t)M * sh.N), hb(sh.N);
                        CK(hipMemcpyAsync(hx.data(), b.x32, hx.size() * 4, hipMemcpyDeviceToHost, st));
                        CK(hipMemcpyAsync(hb.data(), b.bias, hb.size() * 4, hipMemcpyDeviceToHost, st));
                        CK(hipMemcpyAsync(hy.data(), b.y, hy.size() * 4, hipMemcpyDeviceToHost, st));
                        std::vector<float> hp((size_t)MB * 32 * 2);
                        CK(hipMemcpyAsync(hp.data(), b.parts_out, hp.size() * 4, hipMemcpyDeviceToHost, st));
                        CK(hipStreamSynchronize(st));
                        double worst = 0, worst_s = 0;
                        for (int m = 0; m < M; ++m) {
                            double rs = 0;
                            for (int n = 0; n < sh.N; ++n) {
                                const size_t i = (size_t)m * sh.N + n;
                                worst = std::max(worst, (double)std::fabs(hy[i] - (hx[i] + hr[i] + hb[n])));
                                rs += bf16_to_f32(f32_to_bf16(hy[i]));
                            }
                            worst_s = std::max(worst_s, std::fabs(rs - hp[2 * m]));
                        }
                        printf("check %-5s M=%4d split-K %d + combine           max|err| %.3e, row-sum err %.3e %s\n", sh.name, M, S, worst, worst_s, worst <= 0.05 && worst_s < 0.05 ? "ok" : "FAIL");
                    }
                    // split-K finished inside the launch (TS_FUSED): against ref + x32 + bias, row sums over the 12 column-tile parts, packed copy; twice (the counters clean themselves)
                    for (int rep = 0; rep < 4; ++rep) {
                        const int S = 4;
                        if ((sh.K / 16) % (2 * S * 4)) continue;
                        GemmArgs gr = g;
                        gr.C = b.y; gr.store = STORE_RESID; gr.resid_pk = b.respk; gr.resid_parts = b.parts_out; gr.c_packed_mb = MB; gr.bias = b.bias; gr.tile_ctr = tile_ctr;
                        CK(hipMemcpyAsync(b.y, b.x32, (size_t)M * sh.N * 4, hipMemcpyDeviceToDevice, st));
                        CK(hipMemsetAsync(b.slabs, 0xff, (size_t)S * MB * 32 * sh.N * 4, st));
                        if (rep & 1) launch_raw<Tile128, TS_FUSED, false, float>(gr, wpk, S, b.slabs, st);
                        else launch_raw<Tile128PC, TS_FUSED, false, float>(gr, wpk, S, b.slabs, st);
                        const int np = sh.N / 128;
                        std::vector<float> hx((size_t)M * sh.N), hb(sh.N), hp((size_t)np * MB * 32 * 2);
                        CK(hipMemcpyAsync(hx.data(), b.x32, hx.size() * 4, hipMemcpyDeviceToHost, st));
                        CK(hipMemcpyAsync(hb.data(), b.bias, hb.size() * 4, hipMemcpyDeviceToHost, st));
                        CK(hipMemcpyAsync(hy.data(), b.y, hy.size() * 4, hipMemcpyDeviceToHost, st));
                        CK(hipMemcpyAsync(hp.data(), b.parts_out, hp.size() * 4, hipMemcpyDeviceToHost, st));
                        std::vector<bf16_t> hpk((size_t)MB * 32 * sh.N);
                        CK(hipMemcpyAsync(hpk.data(), b.respk, hpk.size() * 2, hipMemcpyDeviceToHost, st));
                        std::vector<unsigned> hc(TILE_CTR_MAX);
                        CK(hipMemcpyAsync(hc.data(), tile_ctr, hc.size() * 4, hipMemcpyDeviceToHost, st));
                        CK(hipStreamSynchronize(st));
                        double worst = 0, worst_s = 0, worst_pk = 0;
                        unsigned dirty = 0;
                        for (unsigned v : hc) dirty |= v;
                        for (int m = 0; m < M; ++m) {
                            double rs = 0, ps = 0;
                            for (int n = 0; n < sh.N; ++n) {
                                const size_t i = (size_t)m * sh.N + n;
                                worst = std::max(worst, (double)std::fabs(hy[i] - (hx[i] + hr[i] + hb[n])));
                                rs += bf16_to_f32(f32_to_bf16(hy[i]));
                                worst_pk = std::max(worst_pk, (double)std::fabs(bf16_to_f32(hpk[packed_off(m, n, MB)]) - bf16_to_f32(f32_to_bf16(hy[i]))));
                            }
                            for (int p = 0; p < np; ++p) ps += hp[((size_t)p * MB * 32 + m) * 2];
                            worst_s = std::max(worst_s, std::fabs(rs - ps));
                        }
                        printf("check %-5s M=%4d split-K 4 FUSED %-10s          max|err| %.3e, row-sum err %.3e, packed copy err %.3e, counters %s %s\n", sh.name, M, (rep & 1) ? "128x128" : "128x128 PC", worst, worst_s, worst_pk,
                               dirty ? "DIRTY" : "clean", worst <= 0.05 && worst_s < 0.05 && worst_pk == 0 && !dirty ? "ok" : "FAIL");
                    }
                    // in-kernel residual epilogue
                    for (int geo = 0; geo < 2; ++geo) {
                        GemmArgs gr = g;
                        gr.C = b.y; gr.store = STORE_RESID; gr.resid_pk = b.respk; gr.resid_parts = b.parts_out; gr.c_packed_mb = MB; gr.bias = b.bias;
                        CK(hipMemcpyAsync(b.y, b.x32, (size_t)M * sh.N * 4, hipMemcpyDeviceToDevice, st));
                        if (geo == 0) launch_raw<Tile256, TS_RESID, false, float>(gr, wpk, 1, nullptr, st);
                        else launch_raw<Tile128, TS_RESID, false, float>(gr, wpk, 1, nullptr, st);
                        const int BN = geo == 0 ? 256 : 128, np = sh.N / BN;
                        std::vector<float> hx((size_t)M * sh.N), hb(sh.N), hp((size_t)np * MB * 32 * 2);
                        CK(hipMemcpyAsync(hx.data(), b.x32, hx.size() * 4, hipMemcpyDeviceToHost, st));
                        CK(hipMemcpyAsync(hb.data(), b.bias, hb.size() * 4, hipMemcpyDeviceToHost, st));
                        CK(hipMemcpyAsync(hy.data(), b.y, hy.size() * 4, hipMemcpyDeviceToHost, st));
                        CK(hipMemcpyAsync(hp.data(), b.parts_out, hp.size() * 4, hipMemcpyDeviceToHost, st));
                        std::vector<bf16_t> hpk((size_t)MB * 32 * sh.N);
                        CK(hipMemcpyAsync(hpk.data(), b.respk, hpk.size() * 2, hipMemcpyDeviceToHost, st));
                        CK(hipStreamSynchronize(st));
                        double worst = 0, worst_s = 0, worst_pk = 0;
                        for (int m = 0; m < M; ++m) {
                            double rs = 0, ps = 0;
                            for (int n = 0; n < sh.N; ++n) {
                                const size_t i = (size_t)m * sh.N + n;
                                worst = std::max(worst, (double)std::fabs(hy[i] - (hx[i] + hr[i] + hb[n])));
                                rs += bf16_to_f32(f32_to_bf16(hy[i]));
                                worst_pk = std::max(worst_pk, (double)std::fabs(bf16_to_f32(hpk[packed_off(m, n, MB)]) - bf16_to_f32(f32_to_bf16(hy[i]))));
                            }
                            for (int p = 0; p < np; ++p) ps += hp[((size_t)p * MB * 32 + m) * 2];
                            worst_s = std::max(worst_s, std::fabs(rs - ps));
                        }
                        printf("check %-5s M=%4d resid epilogue %s           max|err| %.3e, row-sum err %.3e, packed copy err %.3e %s\n", sh.name, M, geo ? "128" : "256", worst, worst_s, worst_pk,
                               worst <= 0.05 && worst_s < 0.05 && worst_pk == 0 ? "ok" : "FAIL");
                    }
                }
                CK(hipFree(wpk));
            }
        }
    }

    // ---- timing: weights cycle through > 256 MB (HBM-cold, as in the AR loop), activations stay hot
    for (const Shape& sh : shapes) {
        const size_t bytes = (size_t)sh.N * sh.K * 2;
        const int nbuf = (int)(((size_t)600 << 20) / bytes) + 1;
        std::vector<bf16_t*> w(nbuf);
        fill_kernel<<<1024, 256, 0, st>>>(w32, (size_t)sh.N * sh.K, 99u, 0.05f);
        for (auto& p : w) { CK(hipMalloc(&p, bytes)); CK(launch_pack_stream_weights(w32, p, sh.N, sh.K, st)); }
        for (int M : {640, 1280, 2560}) {
            const int MB = packed_mb(M);
            pack_rows_kernel<<<1024, 256, 0, st>>>(b.x_rows, b.xpk, M, sh.K, MB);
            const double gf = 2.0 * M * sh.N * sh.K * 1e-9;
            printf("== %s N=%d K=%d M=%d (%.1f GFLOP)\n", sh.name, sh.N, sh.K, M, gf);
            GemmArgs g = make_args(M, sh);
            auto report = [&](const char* what, float t) { printf("   %-44s %8.2f us  %6.0f TFLOP/s\n", what, t, gf / t * 1e3); };
#define T(G, what) report(what, time_us(st, nbuf, [&](int i) { launch_raw<G, TS_ROWS, false, float>(g, w[i], 1, nullptr, st); }));
            T(Tile256, "256x256 k64 x2, fp32 rows") T(Tile256k3, "256x256 k32 x4 , fp32 rows") T(Tile128, "128x128 k32 x3 (3/CU), fp32 rows") T(Tile128k4, "128x128 k64 x2, fp32 rows")
            T(Tile128PC, "128x128 k32 x4 + 4 loader waves, fp32 rows")
            T(Tile128PCk4, "128x128 k64 x3 + 4 loader waves, fp32 rows") T(Tile128PCs6, "128x128 k32 x6 + 4 loader waves, fp32 rows") T(Tile128PC2, "128x128 k32 x4 + 2 loader waves, fp32 rows")
            T(Tile256x128, "256x128 k64 x3, fp32 rows") T(Tile256x128k4, "256x128 k32 x4 , fp32 rows") T(Tile128s3, "128x128 k32 x3 (k32 x4: 2/CU), fp32 rows")
#undef T
            if (M == 640 || M == 1280 || M == 2560) {
                // in-kernel stamps of one launch: cycles to the first landed stage, main loop, epilogue; wall span of the grid
                auto stamps = [&](const char* what, int wgs, const std::function<void(GemmArgs&)>& run) {
                    GemmArgs gs = g;
                    gs.am_best = reinterpret_cast<unsigned long long*>(dbg);
                    CK(hipMemsetAsync(dbg, 0, (size_t)wgs * 64, st));
                    run(gs);
                    std::vector<long long> h((size_t)wgs * 8);
                    CK(hipMemcpyAsync(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost, st));
                    CK(hipStreamSynchronize(st));
                    std::vector<long long> pro, loop, epi, life, e1, e2, e3;
                    long long t0 = LLONG_MAX, t1 = 0;
                    for (int i = 0; i < wgs; ++i) {
                        pro.push_back(h[i * 8 + 1]); loop.push_back(h[i * 8 + 2] - h[i * 8 + 1]); epi.push_back(h[i * 8 + 3] - h[i * 8 + 2]);
                        life.push_back(h[i * 8 + 4] - h[i * 8]);
                        e1.push_back(h[i * 8 + 5] - h[i * 8 + 2]); e2.push_back(h[i * 8 + 6] - h[i * 8 + 5]); e3.push_back(h[i * 8 + 3] - h[i * 8 + 6]);
                        t0 = std::min(t0, h[i * 8]); t1 = std::max(t1, h[i * 8 + 4]);
                    }
                    auto med = [](std::vector<long long>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
                    printf("   stamps %-30s %4d wgs: prologue %6lld  loop %7lld  epilogue %6lld cycles (median); workgroup life %.2f us, grid span %.2f us\n",
                           what, wgs, med(pro), med(loop), med(epi), med(life) / 100.0, (t1 - t0) / 100.0);
                    printf("          epilogue split: sync %lld, values + stores issued %lld, store drain %lld\n", med(e1), med(e2), med(e3));
                };
                // the product's 8-wave 64 x 128 geometry (two workgroups per CU): what a 640-row pass runs
                stamps("64x128 8 waves k64 x3 rows", ((M + 63) / 64) * (sh.N / 128), [&](GemmArgs& gs) { launch_raw<Tile64W8, TS_ROWS, false, float>(gs, w[0], 1, nullptr, st); });
                stamps("64x128 8 waves k64 x3 packed bf16", ((M + 63) / 64) * (sh.N / 128), [&](GemmArgs& gs) { gs.C = b.cpk; gs.store = STORE_PACKED; gs.c_packed_mb = MB; launch_raw<Tile64W8, TS_PACKED, false, bf16_t>(gs, w[0], 1, nullptr, st); });
                stamps("256x256 k32 x4 rows", ((M + 255) / 256) * (sh.N / 256), [&](GemmArgs& gs) { launch_raw<Tile256, TS_ROWS, false, float>(gs, w[0], 1, nullptr, st); });
                stamps("256x128 k32 x4 rows", ((M + 255) / 256) * (sh.N / 128), [&](GemmArgs& gs) { launch_raw<Tile256x128, TS_ROWS, false, float>(gs, w[0], 1, nullptr, st); });
                stamps("128x128 k32 x3 rows", ((M + 127) / 128) * (sh.N / 128), [&](GemmArgs& gs) { launch_raw<Tile128, TS_ROWS, false, float>(gs, w[0], 1, nullptr, st); });
                stamps("128x128 k32 x3 rows (k32 x4: 2/CU)", ((M + 127) / 128) * (sh.N / 128), [&](GemmArgs& gs) { launch_raw<Tile128s3, TS_ROWS, false, float>(gs, w[0], 1, nullptr, st); });
                stamps("128x128 k32 x3 packed bf16", ((M + 127) / 128) * (sh.N / 128), [&](GemmArgs& gs) { gs.C = b.cpk; gs.store = STORE_PACKED; gs.c_packed_mb = MB; launch_raw<Tile128, TS_PACKED, false, bf16_t>(gs, w[0], 1, nullptr, st); });
                stamps("128x128 PC rows", ((M + 127) / 128) * (sh.N / 128), [&](GemmArgs& gs) { launch_raw<Tile128PC, TS_ROWS, false, float>(gs, w[0], 1, nullptr, st); });
                stamps("128x128 PC k64 x3 rows", ((M + 127) / 128) * (sh.N / 128), [&](GemmArgs& gs) { launch_raw<Tile128PCk4, TS_ROWS, false, float>(gs, w[0], 1, nullptr, st); });
                stamps("128x128 PC k32 x6 rows", ((M + 127) / 128) * (sh.N / 128), [&](GemmArgs& gs) { launch_raw<Tile128PCs6, TS_ROWS, false, float>(gs, w[0], 1, nullptr, st); });
                stamps("128x128 PC 2 loaders rows", ((M + 127) / 128) * (sh.N / 128), [&](GemmArgs& gs) { launch_raw<Tile128PC2, TS_ROWS, false, float>(gs, w[0], 1, nullptr, st); });
                stamps("128x128 PC packed bf16", ((M + 127) / 128) * (sh.N / 128), [&](GemmArgs& gs) { gs.C = b.cpk; gs.store = STORE_PACKED; gs.c_packed_mb = MB; launch_raw<Tile128PC, TS_PACKED, false, bf16_t>(gs, w[0], 1, nullptr, st); });
                stamps("128x128 k32 x3 packed bf16 (k32 x4: 2/CU)", ((M + 127) / 128) * (sh.N / 128), [&](GemmArgs& gs) { gs.C = b.cpk; gs.store = STORE_PACKED; gs.c_packed_mb = MB; launch_raw<Tile128s3, TS_PACKED, false, bf16_t>(gs, w[0], 1, nullptr, st); });
            }
            // ---- what lies BETWEEN two dependent launches: wall clock of the first workgroup start / last workgroup end of 8 launches queued back to back (one graph)
            if (M == 640 && (sh.N == 6144 || sh.N == 4608)) {
                const int NL = 8;
                auto gaps = [&](const char* what, int wgs, const std::function<void(GemmArgs&, int)>& run) {
                    CK(hipMemsetAsync(dbg, 0, (size_t)NL * wgs * 64, st));
                    hipGraph_t graph; hipGraphExec_t ge;
                    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                    for (int i = 0; i < NL; ++i) { GemmArgs gs = g; gs.am_best = reinterpret_cast<unsigned long long*>(dbg + (size_t)i * wgs * 8); run(gs, i); }
                    CK(hipStreamEndCapture(st, &graph)); CK(hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0));
                    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
                    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
                    std::vector<long long> h((size_t)NL * wgs * 8);
                    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
                    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(graph));
                    printf("   between launches %-34s", what);
                    double span_sum = 0, gap_sum = 0;
                    long long prev_end = 0;
                    for (int i = 0; i < NL; ++i) {
                        long long t0 = LLONG_MAX, t1 = 0;
                        for (int b2 = 0; b2 < wgs; ++b2) { t0 = std::min(t0, h[((size_t)i * wgs + b2) * 8]); t1 = std::max(t1, h[((size_t)i * wgs + b2) * 8 + 4]); }
                        if (i) { gap_sum += (t0 - prev_end) / 100.0; }
                        span_sum += (t1 - t0) / 100.0;
                        prev_end = t1;
                    }
                    printf(" grid span %.2f us, last end -> next first start %.2f us (mean of %d)\n", span_sum / NL, gap_sum / (NL - 1), NL - 1);
                };
                GemmArgs gq = g; gq.C = b.cpk; gq.store = STORE_PACKED; gq.c_packed_mb = MB;
                gaps("128x128 PC packed bf16", ((M + 127) / 128) * (sh.N / 128), [&](GemmArgs& gs, int i) { gs.C = b.cpk; gs.store = STORE_PACKED; gs.c_packed_mb = MB; launch_raw<Tile128PC, TS_PACKED, false, bf16_t>(gs, w[i], 1, nullptr, st); });
                gaps("128x128 PC fp32 rows", ((M + 127) / 128) * (sh.N / 128), [&](GemmArgs& gs, int i) { launch_raw<Tile128PC, TS_ROWS, false, float>(gs, w[i], 1, nullptr, st); });
                gaps("64x128 W8 packed bf16", ((M + 63) / 64) * (sh.N / 128), [&](GemmArgs& gs, int i) { gs.C = b.cpk; gs.store = STORE_PACKED; gs.c_packed_mb = MB; launch_raw<Tile64W8, TS_PACKED, false, bf16_t>(gs, w[i], 1, nullptr, st); });
            }
            // the store modes the AR loop uses, default geometries
            GemmArgs gd = g;
            gd.ln_parts = b.parts; gd.ln_nparts = 6; gd.ln_colsum = b.colsum; gd.ln_eps = 1e-5f; gd.bias = b.bias;
            if (sh.N != 1536) {
                GemmArgs gp = gd; gp.C = b.cpk; gp.store = STORE_PACKED; gp.c_packed_mb = MB; gp.act = ACT_GELU_ERF;
                report("256x256 DLN + GELU + packed bf16", time_us(st, nbuf, [&](int i) { launch_raw<Tile256, TS_PACKED, true, bf16_t, ACT_GELU_ERF>(gp, w[i], 1, nullptr, st); }));
                report("128x128 DLN + GELU + packed bf16", time_us(st, nbuf, [&](int i) { launch_raw<Tile128, TS_PACKED, true, bf16_t, ACT_GELU_ERF>(gp, w[i], 1, nullptr, st); }));
                report("128x128 PC DLN + GELU + packed bf16", time_us(st, nbuf, [&](int i) { launch_raw<Tile128PC, TS_PACKED, true, bf16_t, ACT_GELU_ERF>(gp, w[i], 1, nullptr, st); }));
                report("64x128 W8 DLN + GELU + packed bf16 (product@640)", time_us(st, nbuf, [&](int i) { launch_raw<Tile64W8, TS_PACKED, true, bf16_t, ACT_GELU_ERF>(gp, w[i], 1, nullptr, st); }));
                // the same launches with the weights cycling through 6 buffers only (<= 113 MB: they stay in the 256-MiB Infinity Cache) -- what a launch would cost if its weights were already there
                report("128x128 PC DLN + GELU + packed, weights in the Infinity Cache", time_us(st, nbuf, [&](int i) { launch_raw<Tile128PC, TS_PACKED, true, bf16_t, ACT_GELU_ERF>(gp, w[i % 6], 1, nullptr, st); }));
                report("128x128 DLN + GELU + packed, weights in the Infinity Cache", time_us(st, nbuf, [&](int i) { launch_raw<Tile128, TS_PACKED, true, bf16_t, ACT_GELU_ERF>(gp, w[i % 6], 1, nullptr, st); }));
                gp.ln_nparts = 1;
                report("256x256 DLN(1 part) + GELU + packed bf16", time_us(st, nbuf, [&](int i) { launch_raw<Tile256, TS_PACKED, true, bf16_t, ACT_GELU_ERF>(gp, w[i], 1, nullptr, st); }));
            } else {
                GemmArgs gr = g;
                gr.C = b.x32; gr.store = STORE_RESID; gr.resid_pk = b.respk; gr.resid_parts = b.parts_out; gr.c_packed_mb = MB; gr.bias = b.bias;
                report("256x256 residual epilogue", time_us(st, nbuf, [&](int i) { launch_raw<Tile256, TS_RESID, false, float>(gr, w[i], 1, nullptr, st); }));
                report("128x128 residual epilogue", time_us(st, nbuf, [&](int i) { launch_raw<Tile128, TS_RESID, false, float>(gr, w[i], 1, nullptr, st); }));
                for (int S : {2, 4, 8}) {
                    if ((sh.K / 16) % (2 * S)) continue;
                    char name[96];
                    snprintf(name, sizeof name, "256x256 split-K %d slabs + combine", S);
                    report(name, time_us(st, nbuf, [&](int i) { launch_raw<Tile256, TS_SLAB, false, float>(gr, w[i], S, b.slabs, st); CK(launch_resid_combine(gr, b.slabs, S, st)); }));
                    snprintf(name, sizeof name, "128x128 split-K %d slabs + combine", S);
                    report(name, time_us(st, nbuf, [&](int i) { launch_raw<Tile128, TS_SLAB, false, float>(gr, w[i], S, b.slabs, st); CK(launch_resid_combine(gr, b.slabs, S, st)); }));
                }
                if ((sh.K / 16) % 32 == 0) {
                    GemmArgs gf = gr; gf.tile_ctr = tile_ctr;
                    report("128x128 split-K 4 FUSED (last arriver)", time_us(st, nbuf, [&](int i) { launch_raw<Tile128, TS_FUSED, false, float>(gf, w[i], 4, b.slabs, st); }));
                    report("128x128 PC split-K 4 FUSED (last arriver)", time_us(st, nbuf, [&](int i) { launch_raw<Tile128PC, TS_FUSED, false, float>(gf, w[i], 4, b.slabs, st); }));
                    report("128x128 PC split-K 4 slabs + combine", time_us(st, nbuf, [&](int i) { launch_raw<Tile128PC, TS_SLAB, false, float>(gr, w[i], 4, b.slabs, st); CK(launch_resid_combine(gr, b.slabs, 4, st)); }));
                    report("64x64 residual epilogue (proj@640 today)", time_us(st, nbuf, [&](int i) { launch_raw<Tile64x64, TS_RESID, false, float>(gr, w[i], 1, nullptr, st); }));
                }
                report("combine alone (S = 4)", time_us(st, nbuf, [&](int i) { CK(launch_resid_combine(gr, b.slabs, 4, st)); }));
            }
            // the streaming kernel the AR loop runs today (PIPE variants, fused epilogue, fp32 rows)
            if (M <= 4096) {
                GemmArgs gs = g;
                report("stream_gemm<2,2,4,4,PIPE> fp32 rows (today)", time_us(st, nbuf, [&](int i) { CK((launch_stream_t<2, 2, 4, 4, float, true>(gs, w[i], 1, nullptr, st))); }));
            }
        }
        for (auto& p : w) CK(hipFree(p));
    }
    return 0;
}
