// Test bench + micro-benchmark of the persistent AR chain (hqtransformer_amd/csrc/persist.hip).
//
//   bench_persist check [D heads layers V M t]   every phase of a small program against a CPU restatement of that phase computed from the
//                                                 GPU's own inputs to it (so a wrong phase is named, not smeared over a whole block)
//   bench_persist time  [D heads layers V M t]   launches of the whole program: us per launch, per phase, and a stamp breakdown
//                                                 (phase start -> barrier passed -> MFMA done -> epilogue -> stores drained -> signalled)
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/micro/bench_persist.hip -o tools/micro/bench_persist
#include "../../hqtransformer_amd/csrc/persist.hip"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct HostLin {
    int N, K;
    bool dln;
    std::vector<float> w, bias, gamma, beta;      // fp32 parameters
    std::vector<float> wb, colsum, bias_f;        // bf16-rounded (gamma o W) as floats, its column sums, folded bias
    float *d_w = nullptr, *d_gamma = nullptr, *d_bias = nullptr, *d_colsum = nullptr;
};
static std::mt19937 rng(1234);
static void make_lin(HostLin& l, int N, int K, bool dln, bool has_bias) {
    l.N = N; l.K = K; l.dln = dln;
    std::normal_distribution<float> nd(0.f, 0.02f), n1(0.f, 1.f);
    l.w.resize((size_t)N * K);
    for (auto& v : l.w) v = nd(rng);
    l.bias.assign(N, 0.f);
    if (has_bias) for (auto& v : l.bias) v = 0.1f * n1(rng);
    l.gamma.assign(K, 1.f); l.beta.assign(K, 0.f);
    if (dln) { for (auto& v : l.gamma) v = 1.f + 0.2f * n1(rng); for (auto& v : l.beta) v = 0.1f * n1(rng); }
    l.wb.resize((size_t)N * K); l.colsum.assign(N, 0.f); l.bias_f = l.bias;
    for (int n = 0; n < N; ++n) {
        double cs = 0, bb = 0;
        for (int k = 0; k < K; ++k) {
            const float f = bf16_to_f32(f32_to_bf16(l.w[(size_t)n * K + k] * l.gamma[k]));
            l.wb[(size_t)n * K + k] = f;
            cs += f; bb += (double)l.w[(size_t)n * K + k] * l.beta[k];
        }
        l.colsum[n] = (float)cs;
        if (dln) l.bias_f[n] = l.bias[n] + (float)bb;
    }
    CK(hipMalloc(&l.d_w, (size_t)N * K * 4)); CK(hipMemcpy(l.d_w, l.w.data(), (size_t)N * K * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&l.d_gamma, K * 4)); CK(hipMemcpy(l.d_gamma, l.gamma.data(), K * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&l.d_bias, N * 4)); CK(hipMemcpy(l.d_bias, l.bias_f.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&l.d_colsum, N * 4)); CK(hipMemcpy(l.d_colsum, l.colsum.data(), N * 4, hipMemcpyHostToDevice));
}

struct Bufs {       // one snapshot of every buffer the program touches (host copies)
    std::vector<bf16_t> xpk, q, abuf, mbuf, kc, vc;
    std::vector<float> x32, logits;
};

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "check";
    const int D = argc > 2 ? atoi(argv[2]) : 256, NH = argc > 3 ? atoi(argv[3]) : 4, L = argc > 4 ? atoi(argv[4]) : 2;
    const int V = argc > 5 ? atoi(argv[5]) : 512, M = argc > 6 ? atoi(argv[6]) : 64, T = argc > 7 ? atoi(argv[7]) : 5;
    const int nt = argc > 8 ? atoi(argv[8]) : 0;
    const int HS = D / NH, MB = packed_mb(M), Mpad = MB * 32, TMAX = 64 > T + 1 ? 64 : T + 1;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("# %s: D %d heads %d (hs %d) layers %d V %d rows %d cached keys %d | %s, %d CUs, nt %d\n", mode.c_str(), D, NH, HS, L, V, M, T, prop.name, ncu, nt);

    // ---- parameters and buffers
    std::vector<HostLin> qkv(L), proj(L), fc1(L), fc2(L);
    HostLin head;
    for (int l = 0; l < L; ++l) { make_lin(qkv[l], 3 * D, D, true, true); make_lin(proj[l], D, D, false, true); make_lin(fc1[l], 4 * D, D, true, true); make_lin(fc2[l], D, 4 * D, false, true); }
    if (V > 0) make_lin(head, V, D, true, false);
    std::normal_distribution<float> n1(0.f, 1.f);
    Bufs init;
    init.x32.resize((size_t)M * D);
    for (auto& v : init.x32) v = n1(rng);
    init.xpk.assign((size_t)Mpad * D, 0);
    for (int m = 0; m < M; ++m) for (int k = 0; k < D; ++k) init.xpk[packed_off(m, k, MB)] = f32_to_bf16(init.x32[(size_t)m * D + k]);
    init.q.assign((size_t)Mpad * D, 0x7fc0); init.abuf.assign((size_t)Mpad * D, 0x7fc0); init.mbuf.assign((size_t)Mpad * 4 * D, 0x7fc0);
    const size_t kvl = (size_t)M * TMAX * D;                       // per layer
    init.kc.assign(kvl * L, 0x7fc0); init.vc.assign(kvl * L, 0x7fc0);
    for (int l = 0; l < L; ++l) for (int b = 0; b < M; ++b) for (int j = 0; j < T; ++j) for (int d = 0; d < D; ++d) {
        init.kc[l * kvl + ((size_t)b * TMAX + j) * D + d] = f32_to_bf16(n1(rng));
        init.vc[l * kvl + ((size_t)b * TMAX + j) * D + d] = f32_to_bf16(n1(rng));
    }
    init.logits.assign((size_t)M * (V > 0 ? V : 1), NAN);
    bf16_t *d_xpk, *d_q, *d_abuf, *d_mbuf, *d_kc, *d_vc;
    float *d_x32, *d_logits;
    int* d_t;
    unsigned *d_counters, *d_err;
    float* d_slabs;
    CK(hipMalloc(&d_xpk, init.xpk.size() * 2)); CK(hipMalloc(&d_q, init.q.size() * 2)); CK(hipMalloc(&d_abuf, init.abuf.size() * 2));
    CK(hipMalloc(&d_mbuf, init.mbuf.size() * 2)); CK(hipMalloc(&d_kc, init.kc.size() * 2)); CK(hipMalloc(&d_vc, init.vc.size() * 2));
    CK(hipMalloc(&d_x32, init.x32.size() * 4)); CK(hipMalloc(&d_logits, init.logits.size() * 4));
    CK(hipMalloc(&d_t, 4)); CK(hipMemcpy(d_t, &T, 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_counters, PERSIST_COUNTER_BYTES)); CK(hipMalloc(&d_err, 256)); CK(hipMemset(d_err, 0, 256));
    CK(hipMalloc(&d_slabs, persist_slab_floats(256) * 4)); CK(hipMemset(d_slabs, 0xFF, persist_slab_floats(256) * 4));
    auto upload = [&](const Bufs& b) {
        CK(hipMemcpy(d_xpk, b.xpk.data(), b.xpk.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d_q, b.q.data(), b.q.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_abuf, b.abuf.data(), b.abuf.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d_mbuf, b.mbuf.data(), b.mbuf.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_kc, b.kc.data(), b.kc.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d_vc, b.vc.data(), b.vc.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_x32, b.x32.data(), b.x32.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_logits, b.logits.data(), b.logits.size() * 4, hipMemcpyHostToDevice));
    };
    auto download = [&](Bufs& b) {
        b = init;
        CK(hipMemcpy(b.xpk.data(), d_xpk, b.xpk.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.q.data(), d_q, b.q.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.abuf.data(), d_abuf, b.abuf.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.mbuf.data(), d_mbuf, b.mbuf.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.kc.data(), d_kc, b.kc.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.vc.data(), d_vc, b.vc.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.x32.data(), d_x32, b.x32.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.logits.data(), d_logits, b.logits.size() * 4, hipMemcpyDeviceToHost));
    };

    // ---- the program
    std::vector<PersistPhase> prog;
    std::vector<const HostLin*> lin_of;
    const int rot_qkv = 0;
    int k4n = 0;
    for (int l = 0; l < L; ++l) {
        PersistPhase ph{};
        ph.type = PP_QKV; ph.N = 3 * D; ph.K = D; ph.rot = rot_qkv; ph.dln = 1; ph.cache_T = TMAX; ph.A = d_xpk; ph.bias = qkv[l].d_bias; ph.colsum = qkv[l].d_colsum;
        ph.out = d_q; ph.kc = d_kc + l * kvl; ph.vc = d_vc + l * kvl;
        prog.push_back(ph); lin_of.push_back(&qkv[l]);
        ph = PersistPhase{}; ph.type = PP_ATTN; ph.cache_T = TMAX; ph.A = d_q; ph.out = d_abuf; ph.kc = d_kc + l * kvl; ph.vc = d_vc + l * kvl;
        prog.push_back(ph); lin_of.push_back(nullptr);
        ph = PersistPhase{}; ph.type = PP_RESID; ph.map = PP_MAP_QUAD; ph.N = D; ph.K = D; ph.A = d_abuf; ph.bias = proj[l].d_bias; ph.out = d_xpk;
        prog.push_back(ph); lin_of.push_back(&proj[l]);
        ph = PersistPhase{}; ph.type = PP_GELU; ph.N = 4 * D; ph.K = D; ph.dln = 1; ph.act = ACT_GELU_ERF; ph.A = d_xpk; ph.bias = fc1[l].d_bias; ph.colsum = fc1[l].d_colsum; ph.out = d_mbuf;
        prog.push_back(ph); lin_of.push_back(&fc1[l]);
        ph = PersistPhase{}; ph.type = PP_RESID_K4; ph.map = PP_MAP_K4; ph.k4_epoch = ++k4n; ph.N = D; ph.K = 4 * D; ph.A = d_mbuf; ph.bias = fc2[l].d_bias; ph.out = d_xpk;
        prog.push_back(ph); lin_of.push_back(&fc2[l]);
    }
    if (V > 0) {
        PersistPhase ph{};
        ph.type = PP_ROWS; ph.N = V; ph.K = D; ph.dln = 1; ph.A = d_xpk; ph.bias = head.d_bias; ph.colsum = head.d_colsum; ph.out = d_logits;
        prog.push_back(ph); lin_of.push_back(&head);
    }
    if (!persist_program_ok(prog, D, M, NH, ncu)) { printf("program not supported\n"); return 1; }
    std::vector<unsigned long long> cu_off, tile_off;
    const size_t stream_bytes = persist_layout(prog, ncu, cu_off, tile_off);
    char* d_stream;
    unsigned long long *d_cu_off, *d_tile_off;
    PersistPhase* d_prog;
    CK(hipMalloc(&d_stream, stream_bytes + 1024)); CK(hipMalloc(&d_cu_off, ncu * 8)); CK(hipMalloc(&d_tile_off, tile_off.size() * 8));
    CK(hipMalloc(&d_prog, prog.size() * sizeof(PersistPhase)));
    CK(hipMemcpy(d_cu_off, cu_off.data(), ncu * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_tile_off, tile_off.data(), tile_off.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_prog, prog.data(), prog.size() * sizeof(PersistPhase), hipMemcpyHostToDevice));
    for (size_t p = 0; p < prog.size(); ++p)
        if (lin_of[p]) CK(launch_persist_pack(lin_of[p]->d_w, lin_of[p]->dln ? lin_of[p]->d_gamma : nullptr, prog[p], ncu, d_stream, d_tile_off + p * ncu, 0));
    CK(hipDeviceSynchronize());
    CK(persist_configure());
    printf("# %zu phases, weight stream %.1f MB (%.1f KB per CU)\n", prog.size(), stream_bytes / 1e6, stream_bytes / 1e3 / ncu);

    PersistArgs a{};
    a.phases = d_prog; a.n_phases = (int)prog.size(); a.wstream = d_stream; a.cu_off = d_cu_off; a.counters = d_counters; a.err = d_err;
    a.x32 = d_x32; a.slabs = d_slabs; a.D = D; a.M = M; a.MB = MB; a.n_heads = NH; a.head_dim = HS; a.t_base = 0; a.t_base_dev = d_t; a.write_back = 1; a.nt_weights = nt;
    persist_default_fill(a);
    if (getenv("FILL")) { sscanf(getenv("FILL"), "%d,%d", &a.fill_s1, &a.fill_s3); printf("# loader budgets %d %d\n", a.fill_s1, a.fill_s3); }
    auto check_err = [&](const char* what) {
        unsigned e = 0;
        CK(hipMemcpy(&e, d_err, 4, hipMemcpyDeviceToHost));
        if (e) { printf("%s: the kernel gave up at the barrier in front of phase %u\n", what, e - 1); exit(2); }
    };

    if (mode == "check") {
        int bad_total = 0;
        Bufs prev = init, cur;
        const float scale = 1.0f / sqrtf((float)HS);
        for (int P = 1; P <= (int)prog.size(); ++P) {
            upload(init);
            a.n_phases = P;
            CK(launch_persist(a, ncu, 0));
            CK(hipDeviceSynchronize());
            check_err("check");
            download(cur);
            const PersistPhase& ph = prog[P - 1];
            const HostLin* lin = lin_of[P - 1];
            const int l = (P - 1) / 5;
            double max_err = 0;
            long bad = 0, cnt = 0;
            auto cmp = [&](float got, float want, float rel, float abs_) {
                const float d = fabsf(got - want);
                ++cnt;
                if (!(d <= abs_ + rel * fabsf(want))) { if (bad < 5) printf("    mismatch: got %g want %g\n", got, want); ++bad; }
                if (d > max_err) max_err = d;
            };
            // A operand of the phase as the GPU saw it
            auto A_of = [&](const std::vector<bf16_t>& pk, int K) {
                std::vector<float> A((size_t)M * K);
                for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) A[(size_t)m * K + k] = bf16_to_f32(pk[packed_off(m, k, MB)]);
                return A;
            };
            auto gemm_row = [&](const std::vector<float>& A, int m, int n) {   // the phase's value before its store, as the kernel defines it
                const int K = lin->K;
                double acc = 0, s = 0, q = 0;
                for (int k = 0; k < K; ++k) { const double x = A[(size_t)m * K + k]; acc += x * lin->wb[(size_t)n * K + k]; s += x; q += x * x; }
                double v = acc;
                if (lin->dln) { const double mean = s / K, var = std::max(q / K - mean * mean, 0.0); v = (acc - mean * lin->colsum[n]) / sqrt(var + 1e-5); }
                return (float)(v + lin->bias_f[n]);
            };
            const char* names[] = {"qkv", "kv1", "attention", "resid", "gelu", "rows", "resid_k4"};
            if (ph.type == PP_QKV) {
                const auto A = A_of(prev.xpk, D);
                for (int m = 0; m < M; ++m) for (int n = 0; n < 3 * D; ++n) {
                    const float want = bf16_to_f32(f32_to_bf16(gemm_row(A, m, n)));
                    const int part = n / D, nn = n % D;
                    const float got = part == 0 ? bf16_to_f32(cur.q[(size_t)m * D + nn])
                                                : bf16_to_f32((part == 1 ? cur.kc : cur.vc)[l * kvl + ((size_t)m * TMAX + T) * D + nn]);
                    cmp(got, want, 1.f / 64, 2e-3f);
                }
            } else if (ph.type == PP_ATTN) {
                for (int b = 0; b < M; ++b) for (int h = 0; h < NH; ++h) {
                    std::vector<float> sc(T + 1);
                    float mx = -INFINITY;
                    for (int j = 0; j <= T; ++j) {
                        float s = 0;
                        for (int d = 0; d < HS; ++d) s = fmaf(bf16_to_f32(prev.q[(size_t)b * D + h * HS + d]), bf16_to_f32(prev.kc[l * kvl + ((size_t)b * TMAX + j) * D + h * HS + d]) * scale, s);
                        sc[j] = s; mx = std::max(mx, s);
                    }
                    double sum = 0;
                    for (int j = 0; j <= T; ++j) sum += exp((double)sc[j] - mx);
                    for (int d = 0; d < HS; ++d) {
                        double o = 0;
                        for (int j = 0; j <= T; ++j) o += exp((double)sc[j] - mx) * bf16_to_f32(prev.vc[l * kvl + ((size_t)b * TMAX + j) * D + h * HS + d]);
                        cmp(bf16_to_f32(cur.abuf[packed_off(b, h * HS + d, MB)]), bf16_to_f32(f32_to_bf16((float)(o / sum))), 1.f / 64, 2e-3f);
                    }
                }
            } else if (ph.type == PP_RESID || ph.type == PP_RESID_K4) {
                const auto A = A_of(lin->K == D ? prev.abuf : prev.mbuf, lin->K);
                for (int m = 0; m < M; ++m) for (int n = 0; n < D; ++n) {
                    const float want = prev.x32[(size_t)m * D + n] + gemm_row(A, m, n);
                    cmp(cur.x32[(size_t)m * D + n], want, 1e-3f, 1e-3f);
                    cmp(bf16_to_f32(cur.xpk[packed_off(m, n, MB)]), bf16_to_f32(f32_to_bf16(cur.x32[(size_t)m * D + n])), 0.f, 0.f);
                }
            } else if (ph.type == PP_GELU) {
                const auto A = A_of(prev.xpk, D);
                for (int m = 0; m < M; ++m) for (int n = 0; n < 4 * D; ++n) {
                    const float v = gemm_row(A, m, n);
                    const float want = bf16_to_f32(f32_to_bf16(v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f))));
                    cmp(bf16_to_f32(cur.mbuf[packed_off(m, n, MB)]), want, 1.f / 64, 2e-3f);
                }
            } else if (ph.type == PP_ROWS) {
                const auto A = A_of(prev.xpk, D);
                for (int m = 0; m < M; ++m) for (int n = 0; n < V; ++n) cmp(cur.logits[(size_t)m * V + n], gemm_row(A, m, n), 1e-3f, 2e-3f);
            }
            printf("phase %3d %-9s N %5d K %5d : %ld values, max |err| %.3g, %ld out of tolerance%s\n", P - 1, names[ph.type], ph.N, ph.K, cnt, max_err, bad, bad ? "   <-- FAIL" : "");
            bad_total += bad != 0;
            prev = cur;
        }
        printf(bad_total ? "CHECK FAILED (%d phases)\n" : "CHECK OK\n", bad_total);
        return bad_total ? 3 : 0;
    }

    // ---- time
    upload(init);
    const int reps = 200;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) CK(launch_persist(a, ncu, 0));
    CK(hipDeviceSynchronize());
    check_err("warm-up");
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) CK(launch_persist(a, ncu, 0));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    check_err("timed");
    const double us = ms * 1e3 / reps;
    printf("whole program: %.1f us per launch (incl. the counter memset), %.2f us per phase, weight stream %.2f TB/s\n", us, us / prog.size(), stream_bytes / us / 1e6);
    // graph replay of the same launch
    {
        hipStream_t cs;
        CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 8; ++i) CK(launch_persist(a, ncu, cs));
        CK(hipStreamEndCapture(cs, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, cs)); CK(hipStreamSynchronize(cs));
        CK(hipEventRecord(e0, cs));
        for (int i = 0; i < reps / 8; ++i) CK(hipGraphLaunch(ge, cs));
        CK(hipEventRecord(e1, cs));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        check_err("graph");
        const double usg = ms * 1e3 / (reps / 8 * 8);
        printf("hipGraph of 8 launches: %.1f us per launch, %.2f us per phase, weight stream %.2f TB/s\n", usg, usg / prog.size(), stream_bytes / usg / 1e6);
    }
    // stamps
    long long* d_st;
    const size_t nst = (size_t)ncu * prog.size() * 8;
    CK(hipMalloc(&d_st, nst * 8)); CK(hipMemset(d_st, 0, nst * 8));
    a.stamps = d_st;
    CK(launch_persist(a, ncu, 0)); CK(launch_persist(a, ncu, 0));
    CK(hipDeviceSynchronize());
    std::vector<long long> st(nst);
    CK(hipMemcpy(st.data(), d_st, nst * 8, hipMemcpyDeviceToHost));
    // per phase: duration = last CU's signal of this phase - last CU's signal of the previous one (the barrier releases when the LAST CU has
    // signalled), the intervals of that last CU, and the median CU's
    const char* names[] = {"qkv", "kv1", "attention", "resid", "gelu", "rows", "resid_k4"};
    printf("# stamps of one launch, us.  dur = release-to-release; then the LAST CU's intervals | the median CU's:\n");
    printf("#   wait = phase start -> barrier passed (S1), x = -> partials in LDS, s3 = -> S3 passed, epi = epilogue (+ the quad reduce) -> stores issued, drain = write-through acks, s4 = -> S4 + signal\n");
    double sums[7][16] = {};
    int cnts[7] = {};
    long long prev_rel = 0;
    for (int c = 0; c < ncu; ++c) prev_rel = c == 0 ? st[0] : std::min(prev_rel, st[((size_t)c * prog.size()) * 8]);
    const long long t_begin = prev_rel;
    for (size_t p = 0; p < prog.size(); ++p) {
        std::vector<double> iv[6];
        long long rel = 0;
        int last = 0;
        for (int c = 0; c < ncu; ++c) {
            const long long* s_ = &st[((size_t)c * prog.size() + p) * 8];
            if (s_[6] > rel) { rel = s_[6]; last = c; }
            for (int i = 0; i < 6; ++i) iv[i].push_back((s_[i + 1] - s_[i]) / 100.0);
        }
        double med[6], lst[6];
        const long long* sl = &st[((size_t)last * prog.size() + p) * 8];
        for (int i = 0; i < 6; ++i) { std::sort(iv[i].begin(), iv[i].end()); med[i] = iv[i][iv[i].size() / 2]; lst[i] = (sl[i + 1] - sl[i]) / 100.0; }
        const double dur = (rel - prev_rel) / 100.0;
        if (p < 11 || p + 6 >= prog.size())
            printf("phase %3zu %-9s +%7.2f dur %6.2f last CU %3d: wait %5.2f x %5.2f s3 %5.2f epi %5.2f drain %5.2f s4 %5.2f | median: wait %5.2f x %5.2f s3 %5.2f epi %5.2f drain %5.2f s4 %5.2f\n",
                   p, names[prog[p].type], (prev_rel - t_begin) / 100.0, dur, last, lst[0], lst[1], lst[2], lst[3], lst[4], lst[5], med[0], med[1], med[2], med[3], med[4], med[5]);
        const int ty = prog[p].type;
        if (p > 0) {
            for (int i = 0; i < 6; ++i) { sums[ty][i] += lst[i]; sums[ty][8 + i] += med[i]; }
            sums[ty][6] += dur;
            cnts[ty]++;
        }
        prev_rel = rel;
    }
    for (int t = 0; t < 7; ++t)
        if (cnts[t]) printf("mean of %3d %-9s: dur %6.2f last CU: wait %5.2f x %5.2f s3 %5.2f epi %5.2f drain %5.2f s4 %5.2f | median: wait %5.2f x %5.2f s3 %5.2f epi %5.2f drain %5.2f s4 %5.2f\n", cnts[t], names[t],
                            sums[t][6] / cnts[t], sums[t][0] / cnts[t], sums[t][1] / cnts[t], sums[t][2] / cnts[t], sums[t][3] / cnts[t], sums[t][4] / cnts[t], sums[t][5] / cnts[t],
                            sums[t][8] / cnts[t], sums[t][9] / cnts[t], sums[t][10] / cnts[t], sums[t][11] / cnts[t], sums[t][12] / cnts[t], sums[t][13] / cnts[t]);
    return 0;
}
