// Experiment: a 128 x 128 workgroup tile whose 4 waves (2 x 2, 64 x 64 each) walk K together and fetch their fragments straight into
// registers -- every A / W fragment is requested by two waves of the same CU at about the same time.  Does the L1 merge the pair
// (halving the L2 -> L1 traffic that bounds the 64-row-tile streaming GEMMs at 512+ rows), without an LDS stage?
#include "../../hqtransformer_amd/csrc/fast_kernels.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int D, int WM, int WN>          // ring depth (k-steps); waves along m / n (WM * WN = 4), wave tile 64 x 64
__global__ __launch_bounds__(256, 2) void panel_kernel(const u32x4* __restrict__ xpk, const u32x4* __restrict__ wpk, float* __restrict__ C, int MB, int N, int K) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave % WM, wn = wave / WM;
    const int KS = K >> 4;
    const int mb0 = blockIdx.y * (2 * WM) + 2 * wm, nt0 = blockIdx.x * (2 * WN) + 2 * wn;
    const u32x4* wp = wpk + ((size_t)nt0 * KS) * 64 + lane;
    const u32x4* xp = xpk + (size_t)mb0 * 64 + lane;
    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][m][r] = 0.0f;
    u32x4 wb[D][2], xb[D][2];
    auto fetch = [&](int u, int k) {
#pragma unroll
        for (int t = 0; t < 2; ++t) wb[u][t] = wp[((size_t)t * KS + k) * 64];
#pragma unroll
        for (int m = 0; m < 2; ++m) xb[u][m] = xp[((size_t)k * MB + m) * 64];
    };
    auto mul = [&](int u) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int m = 0; m < 2; ++m)
                acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wb[u][t]), __builtin_bit_cast(bf16x8, xb[u][m]), acc[t][m], 0, 0, 0);
    };
#pragma unroll
    for (int u = 0; u < D; ++u) fetch(u, u);
    __builtin_amdgcn_sched_barrier(0);
    int ks = 0;
    for (; ks + 2 * D <= KS; ks += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) { mul(u); fetch(u, ks + D + u); __builtin_amdgcn_sched_barrier(0); }
    }
#pragma unroll
    for (int u = 0; u < D; ++u) { mul(u); __builtin_amdgcn_sched_barrier(0); }
    // plain store: C[m][n], col = lane & 31 -> m, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -> n
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = (mb0 + m) * 32 + (lane & 31), col = (nt0 + t) * 32 + 8 * q + 4 * (lane >> 5);
                *reinterpret_cast<f32x4*>(C + (size_t)row * N + col) = f32x4{acc[t][m][4 * q], acc[t][m][4 * q + 1], acc[t][m][4 * q + 2], acc[t][m][4 * q + 3]};
            }
}

// Second experiment: the same 128 x 128 tile with BOTH operands staged through LDS (global_load_dwordx4 -> registers -> ds_write_b128,
// G k-steps per stage, two stages, two workgroups per CU): every fragment crosses the L1 once per workgroup instead of twice.
template <int G>
__global__ __launch_bounds__(256, 2) void panel2_kernel(const u32x4* __restrict__ xpk, const u32x4* __restrict__ wpk, float* __restrict__ C, int MB, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    u32x4* L = reinterpret_cast<u32x4*>(lds);                    // [stage][g][A0..A3 | W0..W3][64 lanes]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave & 1, wn = wave >> 1;
    const int KS = K >> 4, NS = KS / G;
    const int mb0 = blockIdx.y * 4, nt0 = blockIdx.x * 4;
    const u32x4* wp = wpk + ((size_t)(nt0 + wave) * KS) * 64 + lane;      // this wave stages W fragment `wave` ...
    const u32x4* xp = xpk + (size_t)(mb0 + wave) * 64 + lane;             // ... and A fragment `wave` of every k-step
    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][m][r] = 0.0f;
    u32x4 sa[G], sw[G];
    auto fetch = [&](int st) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int k = st * G + g;
            sa[g] = xp[(size_t)k * MB * 64];
            sw[g] = wp[(size_t)k * 64];
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            L[((buf * G + g) * 8 + wave) * 64 + lane] = sa[g];
            L[((buf * G + g) * 8 + 4 + wave) * 64 + lane] = sw[g];
        }
    };
    fetch(0);
    stash(0);
    __syncthreads();
    for (int st = 0; st < NS; ++st) {
        const int buf = st & 1;
        if (st + 1 < NS) fetch(st + 1);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const u32x4* S = L + (buf * G + g) * 8 * 64 + lane;
            const u32x4 a0 = S[(2 * wm) * 64], a1 = S[(2 * wm + 1) * 64], w0 = S[(4 + 2 * wn) * 64], w1 = S[(4 + 2 * wn + 1) * 64];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w0), __builtin_bit_cast(bf16x8, a0), acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w0), __builtin_bit_cast(bf16x8, a1), acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w1), __builtin_bit_cast(bf16x8, a0), acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w1), __builtin_bit_cast(bf16x8, a1), acc[1][1], 0, 0, 0);
        }
        if (st + 1 < NS) stash(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = (mb0 + 2 * wm + m) * 32 + (lane & 31), col = (nt0 + 2 * wn + t) * 32 + 8 * q + 4 * (lane >> 5);
                *reinterpret_cast<f32x4*>(C + (size_t)row * N + col) = f32x4{acc[t][m][4 * q], acc[t][m][4 * q + 1], acc[t][m][4 * q + 2], acc[t][m][4 * q + 3]};
            }
}

struct Shape { const char* name; int N, K; };
template <int G>
static float run2(const Shape& sh, int M, const std::vector<bf16_t*>& w, bf16_t* x, float* y, hipStream_t st) {
    const int MB = M / 32;
    if (MB % 4 || (sh.N / 32) % 4 || (sh.K / 16) % G) return -1.f;
    const dim3 grid(sh.N / 128, MB / 4);
    const size_t smem = (size_t)2 * G * 8 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(panel2_kernel<G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipGraph_t graph; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (size_t i = 0; i < w.size(); ++i)
        panel2_kernel<G><<<grid, 256, smem, st>>>(reinterpret_cast<const u32x4*>(x), reinterpret_cast<const u32x4*>(w[i]), y, MB, sh.N, sh.K);
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, st));
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(graph));
    return 1000.f * ms / (3 * w.size());
}

template <int D, int WM, int WN>
static float run(const Shape& sh, int M, const std::vector<bf16_t*>& w, bf16_t* x, float* y, hipStream_t st) {
    const int MB = M / 32;
    if (MB % (2 * WM) || (sh.N / 32) % (2 * WN) || (sh.K / 16) % D) return -1.f;
    const dim3 grid(sh.N / (64 * WN), MB / (2 * WM));
    hipGraph_t graph; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (size_t i = 0; i < w.size(); ++i)
        panel_kernel<D, WM, WN><<<grid, 256, 0, st>>>(reinterpret_cast<const u32x4*>(x), reinterpret_cast<const u32x4*>(w[i]), y, MB, sh.N, sh.K);
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, st));
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(graph));
    return 1000.f * ms / (3 * w.size());
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const Shape shapes[] = {{"qkv", 4608, 1536}, {"proj", 1536, 1536}, {"fc1", 6144, 1536}, {"fc2", 1536, 6144}, {"head", 8192, 1536}};
    bf16_t* x; float* y;
    const size_t xe = (size_t)4096 * 6144;
    CK(hipMalloc(&x, xe * 2)); CK(hipMalloc(&y, (size_t)4096 * 8192 * 4));
    {
        std::vector<unsigned short> h(xe);
        unsigned s = 1234567u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3c00u + ((s >> 16) & 0x03ffu) + ((s >> 3) & 0x8000u)); }
        CK(hipMemcpy(x, h.data(), xe * 2, hipMemcpyHostToDevice));
    }
    for (const Shape& sh : shapes) {
        const size_t bytes = (size_t)sh.N * sh.K * 2;
        const int nbuf = (int)(((size_t)600 << 20) / bytes) + 1;
        std::vector<bf16_t*> w(nbuf);
        for (auto& p : w) { CK(hipMalloc(&p, bytes)); CK(hipMemcpy(p, x, bytes, hipMemcpyDeviceToDevice)); }
        for (int M : {256, 512, 1024, 2048, 4096}) {
            const double gf = 2.0 * M * sh.N * sh.K * 1e-9;
            printf("== %s N=%d K=%d M=%d (%.1f GFLOP)\n", sh.name, sh.N, sh.K, M, gf);
#define P(D, WM, WN) { float t = run<D, WM, WN>(sh, M, w, x, y, st); if (t > 0) printf("   panel %dx%d waves, ring %d : %7.2f us  %6.0f TFLOP/s\n", WM, WN, D, t, gf / t * 1e3); }
            P(4, 2, 2) P(6, 2, 2) P(4, 1, 4) P(4, 4, 1) P(6, 1, 4)
#define Q(G) { float t = run2<G>(sh, M, w, x, y, st); if (t > 0) printf("   panel2 (both operands through LDS), %d k-steps per stage : %7.2f us  %6.0f TFLOP/s\n", G, t, gf / t * 1e3); }
            Q(2) Q(4) Q(8)
        }
        for (auto& p : w) CK(hipFree(p));
    }
    return 0;
}
