// Micro-benchmark: the SPLIT 3x3 conv kernel (fp16 hi/lo operands, 3 MFMAs per term) on the HQ-VAE decoder's layer shapes
// (batch 64), with ablations that separate the epilogue, the DMA, the fragment reads and the bare MFMA stream.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/micro/bench_split.hip -o tools/micro/bench_split
#include "../../hqtransformer_amd/csrc/split_conv.hip"
#include "../../hqtransformer_amd/csrc/split_stream_conv.hip"
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Shape { const char* name; int res, cin, cout, up; };

template <int PC, int ABL>
static float run(const GemmArgs& g, hipStream_t st, int reps) {
    const dim3 grid((g.N + 127) / 128, g.M / 128, 1);
    auto* k = conv3x3_split_kernel<false, 128, PC, ABL>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, split_conv3_lds(128)));
    k<<<grid, PC ? 512 : 256, split_conv3_lds(128), st>>>(g);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, st));
    for (int r = 0; r < reps; ++r) k<<<grid, PC ? 512 : 256, split_conv3_lds(128), st>>>(g);
    CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return 1000.f * ms / reps;
}

template <int ABL>
static float run_ring16(const GemmArgs& g, hipStream_t st, int reps) {
    const dim3 grid((g.N + 127) / 128, g.M / 128, 1);
    void (*k)(GemmArgs) = conv3x3_split_ring16_kernel<ABL>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS));
    k<<<grid, 256, G_LDS, st>>>(g);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, st));
    for (int r = 0; r < reps; ++r) k<<<grid, 256, G_LDS, st>>>(g);
    CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return 1000.f * ms / reps;
}

template <int ABL>
static float run_up16(const GemmArgs& g, hipStream_t st, int reps) {
    const dim3 grid(4 * (g.N / 128), g.M / 4 / 128, 1);
    void (*k)(GemmArgs) = conv2x2_split_up16_kernel<ABL>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS));
    k<<<grid, 256, G_LDS, st>>>(g);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, st));
    for (int r = 0; r < reps; ++r) k<<<grid, 256, G_LDS, st>>>(g);
    CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return 1000.f * ms / reps;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64;
    hipStream_t st; CK(hipStreamCreate(&st));
    const Shape shapes[] = {
        {"res16  512->512", 16, 512, 512, 0}, {"up32   512->512 (x2)", 32, 512, 512, 1}, {"res32  512->512", 32, 512, 512, 0}, {"up64   512->512 (x2)", 64, 512, 512, 1},
        {"res64  256->256", 64, 256, 256, 0}, {"up128  256->256 (x2)", 128, 256, 256, 1}, {"res128 128->128", 128, 128, 128, 0},
        {"up256  128->128 (x2)", 256, 128, 128, 1},
    };
    const size_t amax = (size_t)B * 256 * 256 * 128, wmax = (size_t)512 * 9 * 512;
    half_t *A, *Wh, *Wl; float *C, *bias; void* zero;
    CK(hipMalloc(&A, amax * 4)); CK(hipMalloc(&C, amax * 4)); CK(hipMalloc(&Wh, wmax * 2)); CK(hipMalloc(&Wl, wmax * 2));
    CK(hipMalloc(&bias, 512 * 4)); CK(hipMalloc(&zero, 256));
    {   // random-ish operand bits (fp16 values in [0.5, 2) and small lo parts): MFMA power depends on the data
        std::vector<unsigned short> hbuf(1 << 22);
        unsigned x = 12345u;
        for (auto& v : hbuf) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3800u + ((x >> 16) & 0x07ffu) + ((x >> 8) & 0x8000u)); }
        for (size_t off = 0; off < amax * 2; off += hbuf.size()) CK(hipMemcpy(A + off, hbuf.data(), std::min(hbuf.size(), amax * 2 - off) * 2, hipMemcpyHostToDevice));
        for (size_t off = 0; off < wmax; off += hbuf.size()) {
            CK(hipMemcpy(Wh + off, hbuf.data(), std::min(hbuf.size(), wmax - off) * 2, hipMemcpyHostToDevice));
            CK(hipMemcpy(Wl + off, hbuf.data() + 77, std::min(hbuf.size() - 77, wmax - off) * 2, hipMemcpyHostToDevice));
        }
    }
    CK(hipMemset(bias, 0, 512 * 4)); CK(hipMemset(zero, 0, 256));
    half_t* Wfrag16; float* W32;
    CK(hipMalloc(&Wfrag16, split_frag_elems(512, 512) * 2)); CK(hipMalloc(&W32, wmax * 4));
    {
        std::vector<float> wf(wmax);
        unsigned x = 777u;
        for (auto& v : wf) { x = x * 1664525u + 1013904223u; v = ((int)(x >> 9) - (1 << 22)) * (1.0f / (1 << 24)); }
        CK(hipMemcpy(W32, wf.data(), wmax * 4, hipMemcpyHostToDevice));
    }
    if (argc > 2 && !strcmp(argv[2], "order")) {
        // workgroup order of the ring16 / up16 kernels: pixel tiles per panel (0 = n-tile fastest), and the kernels without their filter loads
        static half_t* Wup = nullptr;
        CK(hipMalloc(&Wup, split_up_elems(512, 512) * 2));
        const int panels[] = {0, 4, 8, 16, 32, 64, 128};
        printf("%-22s %-7s |", "layer (batch 64)", "kernel");
        for (int p : panels) printf(" p=%-5d", p);
        printf(" | no-filter-loads (p=0)\n");
        for (const Shape& s : shapes) {
            GemmArgs g{};
            g.A = A; g.conv_taps = 9; g.H = s.res; g.W = s.res; g.Cin = s.cin; g.upsample = s.up;
            g.Bw = Wh; g.Bw_lo = Wl; g.ldb = 9 * s.cin; g.C = C; g.ldc = s.cout; g.M = B * s.res * s.res; g.N = s.cout; g.K = 9 * s.cin; g.batch = 1;
            g.bias = bias; g.alpha = 1.f; g.store = STORE_ROWS; g.zero_page = zero;
            CK(launch_pack_split_frag16(W32, Wfrag16, s.cout, s.cin, st));
            g.Bw_frag16 = Wfrag16;
            printf("%-22s %-7s |", s.name, "ring16");
            for (int p : panels) { g.tile_panel = p; printf(" %7.1f", run_ring16<0>(g, st, 5)); }
            g.tile_panel = 0;
            printf(" | %7.1f\n", run_ring16<5>(g, st, 5));
            if (s.up) {
                CK(hipMemset(Wup, 0, split_up_elems(512, 512) * 2));
                CK(launch_pack_split_up16(W32, Wup, s.cout, s.cin, st));
                g.Bw_up16 = Wup;
                printf("%-22s %-7s |", "", "up16");
                for (int p : panels) { g.tile_panel = p; printf(" %7.1f", run_up16<0>(g, st, 5)); }
                g.tile_panel = 0;
                printf(" | %7.1f\n", run_up16<5>(g, st, 5));
            }
        }
        return 0;
    }
    printf("%-22s %8s | %7s %7s %7s | %7s %7s %7s %7s | %7s %7s\n", "layer (batch 64)", "GF x3", "pc1 us", "TF eq", "pc0 us", "no-epi", "no-dma", "dma", "mfma", "pc0nodma", "pc0 mfma");
    double tot = 0, totf = 0, tot_ring16 = 0, tot_up16 = 0;
    for (const Shape& s : shapes) {
        GemmArgs g{};
        g.A = A; g.conv_taps = 9; g.H = s.res; g.W = s.res; g.Cin = s.cin; g.upsample = s.up;
        g.Bw = Wh; g.Bw_lo = Wl; g.ldb = 9 * s.cin; g.C = C; g.ldc = s.cout; g.M = B * s.res * s.res; g.N = s.cout; g.K = 9 * s.cin; g.batch = 1;
        g.bias = bias; g.alpha = 1.f; g.store = STORE_ROWS; g.zero_page = zero;
        const double fl = 3 * 2.0 * g.M * g.N * g.K;
        const int reps = 3;
        const float t0 = run<1, 0>(g, st, reps), p0 = run<0, 0>(g, st, reps), t1 = run<1, 1>(g, st, reps), t2 = run<1, 2>(g, st, reps),
                    t3 = run<1, 3>(g, st, reps), t4 = run<1, 4>(g, st, reps), q2 = run<0, 2>(g, st, reps), q4 = run<0, 4>(g, st, reps);
        printf("%-22s %8.1f | %7.1f %7.1f %7.1f | %7.1f %7.1f %7.1f %7.1f | %7.1f %7.1f\n", s.name, fl * 1e-9, t0, fl / t0 * 1e-6, p0, t1, t2, t3, t4, q2, q4);
        {
            GemmArgs gs = g;
            CK(launch_pack_split_frag16(W32, Wfrag16, s.cout, s.cin, st));
            GemmArgs gh = gs; gh.Bw_frag16 = Wfrag16;
            const float h0 = run_ring16<0>(gh, st, reps), h1 = run_ring16<1>(gh, st, reps), h2 = run_ring16<2>(gh, st, reps), h4 = run_ring16<4>(gh, st, reps);
            printf("%-22s %8s | %7.1f %7.1f %7s | %7.1f %7.1f %7s %7.1f\n", "   ring16 (16x16x32 MFMA)", "", h0, fl / h0 * 1e-6, "", h1, h2, "", h4);
            tot_ring16 += h0;
            if (s.up) {       // the same layer as four 2x2 phase convolutions on the low-resolution image (4 / 9 of the MFMAs)
                static half_t* Wup = nullptr;
                if (!Wup) CK(hipMalloc(&Wup, split_up_elems(512, 512) * 2));
                CK(hipMemset(Wup, 0, split_up_elems(512, 512) * 2));
                CK(launch_pack_split_up16(W32, Wup, s.cout, s.cin, st));
                GemmArgs gu = gh; gu.Bw_up16 = Wup;
                const float u0 = run_up16<0>(gu, st, reps), u1 = run_up16<1>(gu, st, reps), u2 = run_up16<2>(gu, st, reps), u4 = run_up16<4>(gu, st, reps);
                printf("%-22s %8s | %7.1f %7.1f %7s | %7.1f %7.1f %7s %7.1f   (TF eq of the 9-tap work: %.1f)\n", "   up16 (4 phases x 2x2 taps)", "", u0, fl * 4 / 9 / u0 * 1e-6, "", u1, u2, "", u4, fl / u0 * 1e-6);
                tot_up16 += u0;
            } else tot_up16 += h0;
        }
        tot += t0; totf += fl;
    }
    printf("sum (one launch per shape): fallback kernel %.1f us, %.1f TFLOP/s MFMA-equivalent; ring16: %.1f us, %.1f; ring16 + up16: %.1f us   (the superseded generations -- wide, stream, ring -- and their numbers: git history up to round 3, profiles/r0[23]_micro_split_conv_variants.txt)\n", tot, totf / tot * 1e-6, tot_ring16, totf / tot_ring16 * 1e-6, tot_up16);
    return 0;
}
