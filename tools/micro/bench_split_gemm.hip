// Micro-benchmark: the SPLIT 1x1-conv / plain GEMM kernel (fp16 hi/lo operands, 3 MFMAs per term) on the HQ-VAE decoder's shapes at
// batch 64: the attention blocks' q / k / v / proj (16384 x 512 x 512) and the two nin_shortcuts (262144 x 256 x 512, 1048576 x 128 x 256).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/micro/bench_split_gemm.hip -o tools/micro/bench_split_gemm
#define HQT_SPLIT_GEMM_STAMPS 1
#include "../../hqtransformer_amd/csrc/split_conv.hip"
#include "../../hqtransformer_amd/csrc/split_stream_conv.hip"
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Shape { const char* name; int M, N, K; };

static float timed(const GemmArgs& g, hipStream_t st, int reps) {
    CK(launch_split_gemm(g, st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, st));
    for (int r = 0; r < reps; ++r) CK(launch_split_gemm(g, st));
    CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return 1000.f * ms / reps;
}

// `bench_split_gemm ar [rows]`: the SPLIT AR loop's nn.Linear shapes (fp32 activation rows, split while staged) at one merged-pass row count, K-sliced as the engine would
static int ar_mode(int rows, bool packed = false) {      // packed: A as fp16 [row][hi K | lo K] planes (what a separate operand pass would leave) instead of fp32 rows
    hipStream_t st; CK(hipStreamCreate(&st));
    CK(split_kernels_configure());
    const int D = 1536;
    const Shape shapes[] = {{"qkv", rows, 3 * D, D}, {"fc1", rows, 4 * D, D}, {"fc2", rows, D, 4 * D}, {"proj", rows, D, D}};
    float *A, *C, *slabs, *bias; half_t *Wh, *Wl; void* zero;
    CK(hipMalloc(&A, (size_t)rows * 4 * D * 4 * 2)); CK(hipMalloc(&C, (size_t)rows * 4 * D * 4)); CK(hipMalloc(&slabs, (size_t)4 * rows * D * 4));
    CK(hipMalloc(&Wh, (size_t)4 * D * D * 2)); CK(hipMalloc(&Wl, (size_t)4 * D * D * 2)); CK(hipMalloc(&bias, 4 * D * 4)); CK(hipMalloc(&zero, 256));
    {
        std::vector<float> h((size_t)rows * 4 * D);
        unsigned x = 777u;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((int)(x >> 9) - (1 << 22)) * (1.0f / (1 << 22)); }
        CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        std::vector<unsigned short> w((size_t)4 * D * D);
        for (auto& v : w) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x2800u + ((x >> 16) & 0x07ffu) + ((x >> 8) & 0x8000u)); }
        CK(hipMemcpy(Wh, w.data(), w.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(Wl, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    }
    CK(hipMemset(bias, 0, 4 * D * 4)); CK(hipMemset(zero, 0, 256));
    printf("SPLIT AR GEMMs at %d rows (fp32 A): us per launch (incl. the combine of a K-sliced launch), TFLOP/s algorithmic, of the 2.5 PFLOP/s peak issued (x3)\n", rows);
    for (const Shape& s : shapes) {
        GemmArgs g{};
        g.A = A; g.a_f32 = packed ? 0 : 1; g.lda = packed ? 2 * s.K : s.K; g.Bw = Wh; g.Bw_lo = Wl; g.ldb = s.K; g.C = C; g.ldc = s.N; g.M = s.M; g.N = s.N; g.K = s.K; g.batch = 1;
        g.bias = bias; g.alpha = 1.f; g.store = STORE_ROWS; g.zero_page = zero;
        const int S = split_gemm_slices(g);
        if (S > 1) { g.k_slices = S; g.k_slabs = slabs; }
        const double fl = 2.0 * s.M * s.N * s.K;
        const float t = timed(g, st, 20);
        printf("%-5s N %5d K %5d slices %d: %7.1f us  %6.1f TFLOP/s  %.3f issued", s.name, s.N, s.K, S, t, fl / t * 1e-6, 3 * fl / t * 1e-6 / 2500.0);
        const int nwg = ((s.N + 127) / 128) * ((s.M + 127) / 128) * S;
        long long* dbg; CK(hipMalloc(&dbg, (size_t)nwg * 32)); CK(hipMemset(dbg, 0, (size_t)nwg * 32));
        GemmArgs gd = g; gd.am_best = reinterpret_cast<unsigned long long*>(dbg);
        CK(launch_split_gemm(gd, st)); CK(hipStreamSynchronize(st));
        std::vector<long long> h((size_t)nwg * 4); CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<long long> a(nwg), b(nwg), c(nwg);
        for (int i = 0; i < nwg; ++i) { a[i] = h[4 * i]; b[i] = h[4 * i + 1]; c[i] = h[4 * i + 2]; }
        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end()); std::sort(c.begin(), c.end());
        const int kt = s.K / S / 32;
        printf(" | %d workgroups x %d k-tiles: prologue %lld  loop %lld (%.0f per k-tile)  epilogue %lld cycles (median)\n", nwg, kt, a[nwg / 2], b[nwg / 2], (double)b[nwg / 2] / kt, c[nwg / 2]);
        CK(hipFree(dbg));
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "ar")) return ar_mode(argc > 2 ? atoi(argv[2]) : 640);
    if (argc > 1 && !strcmp(argv[1], "arp")) return ar_mode(argc > 2 ? atoi(argv[2]) : 640, true);
    const int B = argc > 1 ? atoi(argv[1]) : 64;
    hipStream_t st; CK(hipStreamCreate(&st));
    CK(split_kernels_configure());
    const Shape shapes[] = {
        {"attn q/k/v/proj 16^2 512->512", B * 256, 512, 512}, {"attn qkv fused 16^2 512->1536", B * 256, 1536, 512},
        {"nin_shortcut 64^2 512->256", B * 4096, 256, 512}, {"nin_shortcut 128^2 256->128", B * 16384, 128, 256},
    };
    const size_t amax = (size_t)B * 16384 * 256 * 2, cmax = (size_t)B * 16384 * 128, wmax = (size_t)1536 * 512;
    half_t *A, *Wh, *Wl; float *C, *bias; void* zero;
    CK(hipMalloc(&A, amax * 2)); CK(hipMalloc(&C, std::max(cmax, (size_t)B * 256 * 1536) * 4)); CK(hipMalloc(&Wh, wmax * 2)); CK(hipMalloc(&Wl, wmax * 2));
    CK(hipMalloc(&bias, 1536 * 4)); CK(hipMalloc(&zero, 256));
    {
        std::vector<unsigned short> hbuf(1 << 22);
        unsigned x = 12345u;
        for (auto& v : hbuf) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3800u + ((x >> 16) & 0x07ffu) + ((x >> 8) & 0x8000u)); }
        for (size_t off = 0; off < amax; off += hbuf.size()) CK(hipMemcpy(A + off, hbuf.data(), std::min(hbuf.size(), amax - off) * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(Wh, hbuf.data(), wmax * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(Wl, hbuf.data() + 77, wmax * 2, hipMemcpyHostToDevice));
    }
    CK(hipMemset(bias, 0, 1536 * 4)); CK(hipMemset(zero, 0, 256));
    printf("%-34s %8s | %8s %8s %8s | %8s %8s\n", "shape (batch 64)", "GF x3", "us", "TF iss.", "of peak", "fp32-A us", "of peak");
    for (const Shape& s : shapes) {
        GemmArgs g{};
        g.A = A; g.lda = 2 * s.K; g.Bw = Wh; g.Bw_lo = Wl; g.ldb = s.K; g.C = C; g.ldc = s.N; g.M = s.M; g.N = s.N; g.K = s.K; g.batch = 1;
        g.bias = bias; g.alpha = 1.f; g.store = STORE_ROWS; g.zero_page = zero;
        const double fl = 3 * 2.0 * s.M * s.N * s.K;
        const float t = timed(g, st, 5);
        GemmArgs gf = g; gf.a_f32 = 1; gf.lda = s.K;                       // A = the fp32 tensor, split while it is staged
        const float tf = timed(gf, st, 5);
        printf("%-34s %8.1f | %8.1f %8.1f %8.3f | %8.1f %8.3f\n", s.name, fl * 1e-9, t, fl / t * 1e-6, fl / t * 1e-6 / 2500.0, tf, fl / tf * 1e-6 / 2500.0);
        {   // in-kernel stamps of one launch (packed operand): median cycles of prologue / loop / epilogue, span of the grid in wall-clock ticks
            const int nwg = ((s.N + 127) / 128) * ((s.M + 127) / 128);
            long long* dbg; CK(hipMalloc(&dbg, (size_t)nwg * 32)); CK(hipMemset(dbg, 0, (size_t)nwg * 32));
            GemmArgs gd = g; gd.am_best = reinterpret_cast<unsigned long long*>(dbg);
            CK(launch_split_gemm(gd, st)); CK(hipStreamSynchronize(st));
            std::vector<long long> h((size_t)nwg * 4); CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
            std::vector<long long> a(nwg), b(nwg), c(nwg); long long w0 = h[3], w1 = h[3];
            for (int i = 0; i < nwg; ++i) { a[i] = h[4 * i]; b[i] = h[4 * i + 1]; c[i] = h[4 * i + 2]; w0 = std::min(w0, h[4 * i + 3]); w1 = std::max(w1, h[4 * i + 3]); }
            std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end()); std::sort(c.begin(), c.end());
            printf("      stamps (%d workgroups, %d k-tiles): prologue %lld  loop %lld (%.0f per k-tile)  epilogue %lld cycles (median); last - first workgroup end: %.2f us\n",
                   nwg, s.K / 32, a[nwg / 2], b[nwg / 2], (double)b[nwg / 2] / (s.K / 32), c[nwg / 2], (w1 - w0) / 100.0);
            CK(hipFree(dbg));
        }
    }
    return 0;
}
