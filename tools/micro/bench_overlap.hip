// Micro-benchmark: the AR block's GEMM chain (qkv -> proj -> fc1 -> fc2, M = 64) replayed from a hipGraph
//   (a) on one stream (kernel boundaries between dependent launches), and
//   (b) on two alternating streams with the ChainSync hand-off (common.h): kernel i+1 streams its weights while
//       kernel i is still running and waits for i's arrival counter before it touches activations.
// Both variants start from the same state and must end with bit-identical residuals.
#include "../../hqtransformer_amd/csrc/fast_kernels.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill_bf16(bf16_t* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = f32_to_bf16(((h & 0xffff) / 32768.0f - 1.0f) * scale);
    }
}
__global__ void fill_f32(float* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((h & 0xffff) / 32768.0f - 1.0f) * scale;
    }
}
__global__ void pack_x(const float* x, bf16_t* xpk, float* parts, int M, int D, int MB) {   // one block per row
    const int m = blockIdx.x;
    float s = 0, q = 0;
    for (int k = threadIdx.x; k < D; k += blockDim.x) {
        bf16_t b = f32_to_bf16(m < M ? x[(size_t)m * D + k] : 0.f);
        xpk[packed_off(m, k, MB)] = b;
        float v = bf16_to_f32(b); s += v; q += v * v;
    }
    __shared__ float rs[256], rq[256];
    rs[threadIdx.x] = s; rq[threadIdx.x] = q; __syncthreads();
    if (threadIdx.x == 0) { float a = 0, b = 0; for (int i = 0; i < (int)blockDim.x; ++i) { a += rs[i]; b += rq[i]; } parts[2 * m] = a; parts[2 * m + 1] = b; }
}

struct Layer { bf16_t *wqkv, *wproj, *wfc1, *wfc2; };

int main(int argc, char** argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 12, M = argc > 2 ? atoi(argv[2]) : 64, D = 1536, reps = 10;
    const int pk = packed_mb(M), Mpad = pk * 32;
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    CK(stream_gemm_configure());
    std::vector<Layer> w(L);
    auto mk = [&](size_t n, unsigned seed) { bf16_t* p; CK(hipMalloc(&p, n * 2)); fill_bf16<<<1024, 256>>>(p, n, seed, 0.02f); return p; };
    for (int l = 0; l < L; ++l) { w[l].wqkv = mk((size_t)3 * D * D, 4 * l + 1); w[l].wproj = mk((size_t)D * D, 4 * l + 2); w[l].wfc1 = mk((size_t)4 * D * D, 4 * l + 3); w[l].wfc2 = mk((size_t)4 * D * D, 4 * l + 4); }
    float *x0, *x, *parts, *colsum, *bias; bf16_t *xpk, *qkv, *hbuf;
    CK(hipMalloc(&x0, (size_t)Mpad * D * 4)); CK(hipMalloc(&x, (size_t)Mpad * D * 4));
    CK(hipMalloc(&parts, (size_t)(D / 32) * Mpad * 2 * 4)); CK(hipMalloc(&colsum, 6144 * 4)); CK(hipMalloc(&bias, 6144 * 4));
    CK(hipMalloc(&xpk, (size_t)Mpad * D * 2)); CK(hipMalloc(&qkv, (size_t)Mpad * 3 * D * 2)); CK(hipMalloc(&hbuf, (size_t)Mpad * 4 * D * 2));
    fill_f32<<<256, 256>>>(x0, (size_t)Mpad * D, 99, 1.0f); fill_f32<<<24, 256>>>(colsum, 6144, 7, 0.1f); fill_f32<<<24, 256>>>(bias, 6144, 8, 0.1f);
    unsigned *ctr, *err; CK(hipMalloc(&ctr, 4096 * 4)); CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
    CK(hipDeviceSynchronize());

    auto wgs = [&](int N, int c_dt_unused) { (void)c_dt_unused; const int wgs2 = (N / 32) * (pk / 2); return (pk == 2 && wgs2 >= 128) ? (N / 32) * (pk / 2) : (N / 32) * pk; };
    // emits the chain; st(i) picks the stream of launch i; chained: use the counters
    auto emit = [&](bool chained, hipStream_t s0, hipStream_t s1) {
        int idx = 0; unsigned prev_wgs = 0;
        auto launch = [&](GemmArgs g, const bf16_t* wp, int c_dt) {
            if (chained) {
                g.chain.wait = idx ? ctr + (idx - 1) : nullptr; g.chain.target = prev_wgs; g.chain.signal = ctr + idx; g.chain.err = err;
            }
            CK(launch_stream_gemm(g, wp, DT_BF16, c_dt, 1, nullptr, (idx & 1) ? s1 : s0));
            prev_wgs = (unsigned)wgs(g.N, c_dt); ++idx;
        };
        int nparts = 1;
        for (int l = 0; l < L; ++l) {
            GemmArgs g{};
            g.A = xpk; g.M = M; g.batch = 1; g.a_packed_mb = pk; g.N = 3 * D; g.K = D; g.alpha = 1.f; g.bias = bias;
            g.ln_parts = parts; g.ln_nparts = nparts; g.ln_colsum = colsum; g.ln_eps = 1e-5f;
            g.C = qkv; g.ldc = 3 * D; g.store = STORE_PACKED; g.c_packed_mb = pk;
            launch(g, w[l].wqkv, DT_BF16);
            g = GemmArgs{};                                    // "proj": reads the first D columns of the packed qkv (stands in for attention output)
            g.A = qkv; g.M = M; g.batch = 1; g.a_packed_mb = pk; g.N = D; g.K = D; g.alpha = 1.f; g.bias = bias;
            g.C = x; g.ldc = D; g.store = STORE_RESID; g.resid_pk = xpk; g.resid_parts = parts; g.c_packed_mb = pk;
            launch(g, w[l].wproj, DT_F32);
            nparts = D / 32;
            g = GemmArgs{};
            g.A = xpk; g.M = M; g.batch = 1; g.a_packed_mb = pk; g.N = 4 * D; g.K = D; g.alpha = 1.f; g.bias = bias;
            g.ln_parts = parts; g.ln_nparts = nparts; g.ln_colsum = colsum; g.ln_eps = 1e-5f;
            g.C = hbuf; g.ldc = 4 * D; g.store = STORE_PACKED; g.c_packed_mb = pk; g.act = ACT_GELU_ERF;
            launch(g, w[l].wfc1, DT_BF16);
            g = GemmArgs{};
            g.A = hbuf; g.M = M; g.batch = 1; g.a_packed_mb = pk; g.N = D; g.K = 4 * D; g.alpha = 1.f; g.bias = bias;
            g.C = x; g.ldc = D; g.store = STORE_RESID; g.resid_pk = xpk; g.resid_parts = parts; g.c_packed_mb = pk;
            launch(g, w[l].wfc2, DT_F32);
        }
        return idx;
    };
    auto reset = [&](hipStream_t st) {
        CK(hipMemcpyAsync(x, x0, (size_t)Mpad * D * 4, hipMemcpyDeviceToDevice, st));
        pack_x<<<Mpad, 256, 0, st>>>(x, xpk, parts, M, D, pk);
    };
    std::vector<float> ref((size_t)M * D), got((size_t)M * D);
    float t_single = 0, t_chain = 0; int nk = 0;
    for (int mode = 0; mode < 6; ++mode) {
        // 0: graph, one stream; 1: graph, two streams + hand-off; 2: graph, one stream + hand-off (protocol cost alone);
        // 3: eager, one stream; 4: eager, two streams + hand-off; 5: eager, one stream + hand-off
        const bool graphed = mode < 3, chained = mode == 1 || mode == 2 || mode == 4 || mode == 5, two = mode == 1 || mode == 4;
        hipGraph_t graph = nullptr; hipGraphExec_t ge = nullptr;
        hipEvent_t fork, join; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
        auto body = [&] {
            if (chained) CK(hipMemsetAsync(ctr, 0, 4096 * 4, sa));
            if (two) { CK(hipEventRecord(fork, sa)); CK(hipStreamWaitEvent(sb, fork, 0)); }
            nk = emit(chained, sa, two ? sb : sa);
            if (two) { CK(hipEventRecord(join, sb)); CK(hipStreamWaitEvent(sa, join, 0)); }
        };
        if (graphed) {
            CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
            body();
            CK(hipStreamEndCapture(sa, &graph));
            CK(hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0));
        }
        auto run = [&] { if (graphed) CK(hipGraphLaunch(ge, sa)); else body(); };
        CK(hipMemset(err, 0, 4));
        reset(sa); run(); CK(hipStreamSynchronize(sa));
        CK(hipMemcpy(mode == 0 ? ref.data() : got.data(), x, (size_t)M * D * 4, hipMemcpyDeviceToHost));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        reset(sa);
        CK(hipEventRecord(a, sa));
        for (int r = 0; r < reps; ++r) run();
        CK(hipEventRecord(b, sa)); CK(hipStreamSynchronize(sa));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const float us = 1000.f * ms / reps / nk;
        unsigned e = 0; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        if (mode) for (size_t i = 0; i < ref.size(); ++i) bad += (ref[i] != got[i]) || !(got[i] == got[i]);
        static const char* names[] = {"graph, one stream", "graph, two streams + hand-off", "graph, one stream + hand-off",
                                      "eager, one stream", "eager, two streams + hand-off", "eager, one stream + hand-off"};
        printf("mode %d (%s): %d GEMMs, %.2f us per GEMM, %.1f us per layer; give-ups %u; mismatches vs mode 0: %zu (x[0]=%g)\n", mode,
               names[mode], nk, us, us * 4, e, bad, mode ? got[0] : ref[0]);
        if (mode == 0) t_single = us; if (mode == 4) t_chain = us;
        if (graphed) { CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(graph)); }
    }
    printf("speed-up of the chained form: %.3fx\n", t_single / t_chain);
    return 0;
}
