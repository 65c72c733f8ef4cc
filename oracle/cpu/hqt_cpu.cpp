// CPU twin of hqt_sample / hqt_decode -- TEST INFRASTRUCTURE and the measured CPU baseline, NOT product code (include/hqt_cpu.h).
//
// fp32 restatement, in C++ / OpenMP, of what the reference computes on its CPU path (autocast off: every activation fp32).  Every
// function cites the reference lines it follows (paths relative to the reference root); third-party arithmetic is PyTorch's
// (nn.Linear / LayerNorm / GELU(erf) / softmax / topk / sort / cumsum / multinomial / Conv2d / GroupNorm / nearest interpolate /
// PixelShuffle / embedding), restated from the published definitions.  Pinned by the reference-generated fixtures G3 / G4 / G5
// (tests/test_cpu_twin.py) and cross-checked against the numpy oracle on other seeded shapes.
//
// Layout / parallelisation (what makes it a credible baseline on a many-core host, unlike the numpy oracle's 64-row BLAS calls):
//   * AR GEMMs (64 .. 256 rows): the N columns are cut into one contiguous slice per thread; the weights are copied at finalize by
//     the threads that will stream them (first touch = NUMA-local), every thread streams its slice once per call and keeps the
//     activation panel in its L2;
//   * decoder convolutions: NHWC, filters repacked tap-major [O][kh kw][I]; a task = 32 output pixels of one image row: im2col rows
//     [pixel][tap * Cin + c] in a per-thread buffer (nearest-x2 upsampling and the zero padding folded into the gather), then the
//     serial GEMM tile against the shared filters;
//   * GroupNorm statistics as per-chunk double partials of all 32 groups at once, combined in a fixed order.
#include "../../include/hqt_cpu.h"

#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

extern "C" void hqt_cpu_gemm_tile_avx2(const float*, long, const float*, long, float*, long, int, int, int);
extern "C" void hqt_cpu_gemm_tile_avx512(const float*, long, const float*, long, float*, long, int, int, int);
typedef void (*gemm_tile_fn)(const float*, long, const float*, long, float*, long, int, int, int);

static gemm_tile_fn pick_gemm(const char** isa) {
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx512f") && !getenv("HQT_CPU_NO_AVX512")) { *isa = "avx512"; return hqt_cpu_gemm_tile_avx512; }
    *isa = "avx2";
    return hqt_cpu_gemm_tile_avx2;
}
static const char* g_isa = nullptr;
static gemm_tile_fn g_gemm = pick_gemm(&g_isa);

static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define CHK(x) do { const int e_ = (x); if (e_ != HQT_OK) return e_; } while (0)

static float* alloc_f(size_t n) {
    void* p = nullptr;
    if (posix_memalign(&p, 64, std::max<size_t>(n, 16) * sizeof(float)) != 0) return nullptr;
    return static_cast<float*>(p);
}

struct Arr {
    std::vector<int64_t> shape;
    std::vector<float> d;
};
struct Lin {                 // y = x W^T + b, W [N][K] (K contiguous), placed by the threads that stream it
    float* w = nullptr;
    const float* b = nullptr;
    int N = 0, K = 0;
};
struct BlockW {
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    Lin qkv, proj, fc1, fc2;
    std::vector<float> qkv_bias;
};
struct ConvW {
    Lin l;                   // filters tap-major [O][taps * I]
    int taps = 1, cin = 0;
};

struct hqt_cpu_handle {
    hqt_config c{};
    int T = 1;
    bool finalized = false;
    double last_s = 0.0;
    std::map<std::string, Arr> w;
    std::vector<float*> owned;
    std::vector<BlockW> body, depth;
    Lin head_top, head_bot;
    std::map<std::string, ConvW> conv;
    ~hqt_cpu_handle() { for (float* p : owned) free(p); }
};

extern "C" const char* hqt_cpu_last_error(void) { return g_err.c_str(); }
extern "C" const char* hqt_cpu_isa(void) { return g_isa; }
extern "C" int hqt_cpu_threads(const hqt_cpu_handle* h) { return h ? h->T : -1; }
extern "C" double hqt_cpu_last_seconds(const hqt_cpu_handle* h) { return h ? h->last_s : -1.0; }

extern "C" int hqt_cpu_create(const hqt_config* cfg, int n_threads, hqt_cpu_handle** out) {
    if (!cfg || !out) return fail(HQT_ERR_INVALID, "null config / out");
    if (cfg->abi_version != HQT_ABI_VERSION) return fail(HQT_ERR_INVALID, "abi_version %d != %d", cfg->abi_version, HQT_ABI_VERSION);
    if (cfg->code_levels == 3) return fail(HQT_ERR_INVALID, "the CPU twin covers the two-level path (hqt_sample / hqt_decode) only");
    auto* h = new hqt_cpu_handle;
    h->c = *cfg;
    h->T = n_threads > 0 ? n_threads : omp_get_max_threads();
    *out = h;
    return HQT_OK;
}
extern "C" int hqt_cpu_destroy(hqt_cpu_handle* h) {
    delete h;
    return HQT_OK;
}
extern "C" int hqt_cpu_set_weight(hqt_cpu_handle* h, const char* name, const float* data, const int64_t* shape, int ndim) {
    if (!h || !name || !data || !shape || ndim < 1) return fail(HQT_ERR_INVALID, "null argument");
    if (h->finalized) return fail(HQT_ERR_STATE, "set_weight after finalize");
    Arr a;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { a.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    a.d.assign(data, data + n);
    h->w[name] = std::move(a);
    return HQT_OK;
}

// ---------------------------------------------------------------------------------------------- weights
static int get(hqt_cpu_handle* h, const std::string& name, std::vector<int64_t> shape, const float** out) {
    auto it = h->w.find(name);
    if (it == h->w.end()) return fail(HQT_ERR_MISSING_WEIGHT, "missing weight %s", name.c_str());
    if (it->second.shape != shape) return fail(HQT_ERR_SHAPE, "weight %s has the wrong shape", name.c_str());
    *out = it->second.d.data();
    return HQT_OK;
}
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// columns per thread slice of an N-split GEMM: a multiple of 4 (the micro-kernel's NR), at most T slices
static inline int slice_cols(int N, int T) { return std::max(4, cdiv(cdiv(N, T), 4) * 4); }

// copies W [N][K] so that the thread that will stream rows [n0, n1) is the one that first touches them
static int place(hqt_cpu_handle* h, Lin& l, const float* const* parts, const int* part_rows, int nparts, int K, const float* bias) {
    int N = 0;
    for (int i = 0; i < nparts; ++i) N += part_rows[i];
    l.N = N; l.K = K; l.b = bias;
    l.w = alloc_f((size_t)N * K);
    if (!l.w) return fail(HQT_ERR_INVALID, "out of memory");
    h->owned.push_back(l.w);
    const int nb = slice_cols(N, h->T), ns = cdiv(N, nb);
    // schedule(static, 1): slice t goes to thread t (as many threads as slices at most) -- in linear() below as well, so a slice is
    // streamed by the thread that first touched it
#pragma omp parallel for num_threads(h->T) schedule(static, 1)
    for (int t = 0; t < ns; ++t) {
        const int n0 = t * nb, n1 = std::min(N, n0 + nb);
        for (int n = n0; n < n1; ++n) {
            int p = 0, r = n;
            while (r >= part_rows[p]) { r -= part_rows[p]; ++p; }
            memcpy(l.w + (size_t)n * K, parts[p] + (size_t)r * K, (size_t)K * 4);
        }
    }
    return HQT_OK;
}
static int place1(hqt_cpu_handle* h, Lin& l, const float* w, int N, int K, const float* bias) { return place(h, l, &w, &N, 1, K, bias); }

static int load_block(hqt_cpu_handle* h, const std::string& p, BlockW& b) {
    const int64_t D = h->c.embed_dim;
    CHK(get(h, p + ".ln1.weight", {D}, &b.ln1_g)); CHK(get(h, p + ".ln1.bias", {D}, &b.ln1_b));
    CHK(get(h, p + ".ln2.weight", {D}, &b.ln2_g)); CHK(get(h, p + ".ln2.bias", {D}, &b.ln2_b));
    const float *wq, *wk, *wv, *bq, *bk, *bv, *w, *bb;
    CHK(get(h, p + ".attn.query.weight", {D, D}, &wq)); CHK(get(h, p + ".attn.query.bias", {D}, &bq));
    CHK(get(h, p + ".attn.key.weight", {D, D}, &wk)); CHK(get(h, p + ".attn.key.bias", {D}, &bk));
    CHK(get(h, p + ".attn.value.weight", {D, D}, &wv)); CHK(get(h, p + ".attn.value.bias", {D}, &bv));
    b.qkv_bias.resize(3 * D);
    memcpy(&b.qkv_bias[0], bq, D * 4); memcpy(&b.qkv_bias[D], bk, D * 4); memcpy(&b.qkv_bias[2 * D], bv, D * 4);
    const float* parts[3] = {wq, wk, wv};
    const int rows[3] = {(int)D, (int)D, (int)D};
    CHK(place(h, b.qkv, parts, rows, 3, (int)D, b.qkv_bias.data()));       // [query; key; value] rows: one GEMM, the three nn.Linear results
    CHK(get(h, p + ".attn.proj.weight", {D, D}, &w)); CHK(get(h, p + ".attn.proj.bias", {D}, &bb)); CHK(place1(h, b.proj, w, D, D, bb));
    CHK(get(h, p + ".mlp.0.weight", {4 * D, D}, &w)); CHK(get(h, p + ".mlp.0.bias", {4 * D}, &bb)); CHK(place1(h, b.fc1, w, 4 * D, D, bb));
    CHK(get(h, p + ".mlp.2.weight", {D, 4 * D}, &w)); CHK(get(h, p + ".mlp.2.bias", {D}, &bb)); CHK(place1(h, b.fc2, w, D, 4 * D, bb));
    return HQT_OK;
}

static int load_conv(hqt_cpu_handle* h, const std::string& name, int O, int I, int ks) {
    const float *w, *b;
    CHK(get(h, "stage1." + name + ".weight", {O, I, ks, ks}, &w));
    CHK(get(h, "stage1." + name + ".bias", {O}, &b));
    const int taps = ks * ks;
    std::vector<float> tm((size_t)O * taps * I);                         // [O][I][kh][kw] -> [O][tap][I]
    for (int o = 0; o < O; ++o)
        for (int i = 0; i < I; ++i)
            for (int t = 0; t < taps; ++t) tm[((size_t)o * taps + t) * I + i] = w[((size_t)o * I + i) * taps + t];
    ConvW c;
    c.taps = taps; c.cin = I;
    CHK(place1(h, c.l, tm.data(), O, taps * I, b));
    h->conv[name] = c;
    return HQT_OK;
}

struct DecStep { int kind; std::string name; int cin, cout, res; };      // 0 conv3, 1 res, 2 attn, 3 upconv
static std::vector<DecStep> decoder_plan(const hqt_config& c) {          // Decoder.__init__ / forward, stage1/modules/layers.py:300-410
    std::vector<DecStep> p;
    const int n = c.s1_n_mult;
    int res = c.s1_resolution >> (n - 1 + (c.s1_use_init_downsample ? 1 : 0));
    int bin = c.s1_ch * c.s1_ch_mult[n - 1];
    auto attn_at = [&](int r) { for (int i = 0; i < c.s1_n_attn_res; ++i) if (c.s1_attn_res[i] == r) return true; return false; };
    p.push_back({0, "decoder.conv_in", c.s1_z_channels, bin, res});
    if (c.s1_use_mid_block) {
        p.push_back({1, "decoder.mid.block_1", bin, bin, res});
        if (c.s1_use_attn) p.push_back({2, "decoder.mid.attn_1", bin, bin, res});
        p.push_back({1, "decoder.mid.block_2", bin, bin, res});
    }
    for (int lvl = n - 1; lvl >= 0; --lvl) {
        const int bout = c.s1_ch * c.s1_ch_mult[lvl];
        for (int blk = 0; blk <= c.s1_num_res_blocks; ++blk) {
            p.push_back({1, "decoder.up." + std::to_string(lvl) + ".block." + std::to_string(blk), bin, bout, res});
            bin = bout;
            if (attn_at(res) && c.s1_use_attn) p.push_back({2, "decoder.up." + std::to_string(lvl) + ".attn." + std::to_string(blk), bin, bin, res});
        }
        if (lvl != 0 || c.s1_use_init_downsample) {
            p.push_back({3, "decoder.up." + std::to_string(lvl) + ".upsample.conv", bin, bin, res});
            res *= 2;
        }
    }
    p.push_back({4, "decoder.conv_out", bin, c.s1_out_ch, res});
    return p;
}

extern "C" int hqt_cpu_finalize_weights(hqt_cpu_handle* h) {
    if (!h) return fail(HQT_ERR_INVALID, "null handle");
    if (h->finalized) return HQT_OK;
    const hqt_config& c = h->c;
    if (c.has_stage2) {
        const int64_t D = c.embed_dim, V = c.vocab_top;
        h->body.resize(c.n_layers);
        h->depth.resize(c.n_layers_depth);
        for (int i = 0; i < c.n_layers; ++i) CHK(load_block(h, "stage2.blocks." + std::to_string(i), h->body[i]));
        for (int i = 0; i < c.n_layers_depth; ++i) CHK(load_block(h, "stage2.depths." + std::to_string(i), h->depth[i]));
        const float* w;
        CHK(get(h, "stage2.head_top.weight", {V, D}, &w)); CHK(place1(h, h->head_top, w, V, D, nullptr));
        CHK(get(h, "stage2.head_bot.weight", {(int64_t)c.vocab_bot, D}, &w)); CHK(place1(h, h->head_bot, w, c.vocab_bot, D, nullptr));
    }
    if (c.has_stage1) {
        CHK(load_conv(h, "post_quant_conv_b", c.s1_z_channels, 2 * c.s1_embed_dim, 1));
        for (const DecStep& s : decoder_plan(c)) {
            if (s.kind == 0 || s.kind == 3 || s.kind == 4) CHK(load_conv(h, s.name, s.cout, s.cin, 3));
            else if (s.kind == 1) {
                CHK(load_conv(h, s.name + ".conv1", s.cout, s.cin, 3));
                CHK(load_conv(h, s.name + ".conv2", s.cout, s.cout, 3));
                if (s.cin != s.cout) CHK(load_conv(h, s.name + ".nin_shortcut", s.cout, s.cin, 1));
            } else {
                for (const char* q : {".q", ".k", ".v", ".proj_out"}) CHK(load_conv(h, s.name + q, s.cin, s.cin, 1));
            }
        }
    }
    h->finalized = true;
    return HQT_OK;
}

// ---------------------------------------------------------------------------------------------- primitives
static inline float gelu_erf(float x) { return x * 0.5f * (1.0f + erff(x / sqrtf(2.0f))); }      // stage2/layers.py:14-23 (nn.GELU)
static inline float gelu_sig(float x) { return x / (1.0f + expf(-1.702f * x)); }

// C[M][N] = A W^T (+bias) (act) (+resid), the N columns cut into one slice per thread (the slices `place` touched)
static void linear(const hqt_cpu_handle* h, const float* A, long lda, const Lin& l, float* C, long ldc, int M, int act, const float* resid) {
    const int N = l.N, K = l.K, nb = slice_cols(N, h->T), ns = cdiv(N, nb);
#pragma omp parallel num_threads(h->T)
    {
        std::vector<float> tmp((size_t)M * nb);                          // the slice's products; resid may alias C (x += ...)
#pragma omp for schedule(static, 1)
        for (int t = 0; t < ns; ++t) {
            const int n0 = t * nb, n1 = std::min(N, n0 + nb);
            g_gemm(A, lda, l.w + (size_t)n0 * K, K, tmp.data(), nb, M, n1 - n0, K);
            for (int m = 0; m < M; ++m) {
                float* row = C + (size_t)m * ldc;
                const float* pr = tmp.data() + (size_t)m * nb;
                for (int n = n0; n < n1; ++n) {
                    float v = pr[n - n0] + (l.b ? l.b[n] : 0.0f);
                    if (act == 1) v = gelu_erf(v);
                    else if (act == 2) v = gelu_sig(v);
                    if (resid) v = resid[(size_t)m * ldc + n] + v;
                    row[n] = v;
                }
            }
        }
    }
}

// nn.LayerNorm over the last axis, eps 1e-5, statistics in double (stage2/layers.py:302-303; hierarchical_ar.py:144,205,208)
static void layer_norm(const hqt_cpu_handle* h, const float* x, const float* g, const float* b, float* y, int M, int D) {
#pragma omp parallel for num_threads(h->T) schedule(static)
    for (int m = 0; m < M; ++m) {
        const float* r = x + (size_t)m * D;
        double s = 0.0;
        for (int i = 0; i < D; ++i) s += r[i];
        const double mu = s / D;
        double v = 0.0;
        for (int i = 0; i < D; ++i) { const double d = r[i] - mu; v += d * d; }
        const double sd = sqrt(v / D + 1e-5);
        float* o = y + (size_t)m * D;
        for (int i = 0; i < D; ++i) o[i] = (float)((r[i] - mu) / sd) * g[i] + b[i];
    }
}

// MultiHeadSelfAttention.forward with caching (stage2/layers.py:61-195): qkv [B*T][3D]; K / V rows are appended to the caches
// ([B][Tcap][D]) at t_past .. t_past + T; scores = q . (k / sqrt(hs)) (the scale is applied to K, :102); causal: query t sees keys
// <= t_past + t (:107-111,118-123); the depth head's masks reduce to "every key" on the rows a sampling sub-step evaluates (:125-152)
static void attention(const hqt_cpu_handle* h, const float* qkv, float* kc, float* vc, float* y, int B, int T, int Tcap, int t_past, bool causal) {
    const int D = h->c.embed_dim, nh = h->c.n_heads, hs = D / nh;
    const float scale = (float)(1.0 / sqrt((double)hs));
#pragma omp parallel for num_threads(h->T) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t) {
            const float* src = qkv + ((size_t)b * T + t) * 3 * D;
            memcpy(kc + ((size_t)b * Tcap + t_past + t) * D, src + D, (size_t)D * 4);
            memcpy(vc + ((size_t)b * Tcap + t_past + t) * D, src + 2 * D, (size_t)D * 4);
        }
#pragma omp parallel num_threads(h->T)
    {
        std::vector<float> sc(t_past + T);
#pragma omp for schedule(static) collapse(2)
        for (int b = 0; b < B; ++b)
            for (int hd = 0; hd < nh; ++hd)
                for (int t = 0; t < T; ++t) {
                    const float* q = qkv + ((size_t)b * T + t) * 3 * D + hd * hs;
                    const int nk = causal ? t_past + t + 1 : t_past + T;
                    float mx = -INFINITY;
                    for (int j = 0; j < nk; ++j) {
                        const float* k = kc + ((size_t)b * Tcap + j) * D + hd * hs;
                        float s = 0.0f;
                        for (int e = 0; e < hs; ++e) s += q[e] * (k[e] * scale);
                        sc[j] = s;
                        mx = std::max(mx, s);
                    }
                    float den = 0.0f;
                    for (int j = 0; j < nk; ++j) { sc[j] = expf(sc[j] - mx); den += sc[j]; }
                    float* o = y + ((size_t)b * T + t) * D + hd * hs;
                    for (int e = 0; e < hs; ++e) o[e] = 0.0f;
                    for (int j = 0; j < nk; ++j) {
                        const float p = sc[j] / den;
                        const float* v = vc + ((size_t)b * Tcap + j) * D + hd * hs;
                        for (int e = 0; e < hs; ++e) o[e] += p * v[e];
                    }
                }
    }
}

struct Work {                // activations of one hqt_cpu_sample call
    std::vector<float> x, hbuf, qkv, att, mlp, xd, logits;
};

// Block.sample / ParallelBlock.sample (stage2/layers.py:324-328,371-375) on M = B * T rows of x, in place
static void block(const hqt_cpu_handle* h, const BlockW& w, Work& k, float* x, int B, int T, float* kc, float* vc, int Tcap, int t_past, bool causal) {
    const int D = h->c.embed_dim, M = B * T;
    layer_norm(h, x, w.ln1_g, w.ln1_b, k.hbuf.data(), M, D);
    linear(h, k.hbuf.data(), D, w.qkv, k.qkv.data(), 3 * D, M, 0, nullptr);
    attention(h, k.qkv.data(), kc, vc, k.att.data(), B, T, Tcap, t_past, causal);
    linear(h, k.att.data(), D, w.proj, x, D, M, 0, x);
    layer_norm(h, x, w.ln2_g, w.ln2_b, k.hbuf.data(), M, D);
    linear(h, k.hbuf.data(), D, w.fc1, k.mlp.data(), 4 * D, M, h->c.gelu_approx ? 2 : 1, nullptr);
    linear(h, k.mlp.data(), 4 * D, w.fc2, x, D, M, 0, x);
}

// ---- sampler: logits / T -> top-k -> softmax -> top-p -> argmax(p / q)  (hqvae/utils/sampling.py:12-37; hierarchical_ar.py:762-785)
static inline uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
static void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// the Exp(1) variate libhqt's sampler draws for (row key, step * 5 + draw, vocabulary index): csrc/kernels.hip, sampler_kernel
static void philox_row(uint64_t seed, uint64_t grow, int counter, int V, float* q) {
    for (int i4 = 0; i4 * 4 < V; ++i4) {
        uint32_t r[4];
        philox4x32_10((uint32_t)i4, (uint32_t)counter, (uint32_t)grow, (uint32_t)(grow >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), r);
        for (int e = 0; e < 4 && i4 * 4 + e < V; ++e) q[i4 * 4 + e] = -logf(((float)(r[e] >> 9) + 0.5f) * (1.0f / 8388608.0f));
    }
}
static int64_t sample_row(const float* logits, const float* q, int V, float temperature, int top_k, float top_p, std::vector<float>& p, std::vector<int>& order) {
    p.resize(V);
    for (int i = 0; i < V; ++i) p[i] = logits[i] / temperature;
    if (top_k > 0 && top_k < V) {                                       // keep logits >= k-th largest, ties kept (sampling.py:12-19)
        std::vector<float> tmp(p);
        std::nth_element(tmp.begin(), tmp.begin() + (V - top_k), tmp.end());
        const float kth = tmp[V - top_k];
        for (int i = 0; i < V; ++i) if (p[i] < kth) p[i] = -INFINITY;
    }
    float mx = -INFINITY;
    for (int i = 0; i < V; ++i) mx = std::max(mx, p[i]);
    float den = 0.0f;
    for (int i = 0; i < V; ++i) { p[i] = expf(p[i] - mx); den += p[i]; }
    for (int i = 0; i < V; ++i) p[i] = p[i] / den;
    if (top_p > 0.0f) {                                                 // sampling.py:22-37; torch.cumsum accumulates an fp32 row in a double
        order.resize(V);
        for (int i = 0; i < V; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return p[a] > p[b]; });
        double cum = 0.0;
        bool remove_prev = false;
        float total = 0.0f;
        std::vector<char> rm(V, 0);
        for (int r = 0; r < V; ++r) {
            cum += (double)p[order[r]];
            const bool remove_here = (float)cum >= top_p;
            if (r > 0 && remove_prev) rm[order[r]] = 1;                 // shifted right by one: the first crossing token is kept
            remove_prev = remove_here;
        }
        for (int i = 0; i < V; ++i) { if (rm[i]) p[i] = 0.0f; total += p[i]; }
        for (int i = 0; i < V; ++i) p[i] = p[i] / total;
    }
    int best = 0;
    float bv = -1.0f;
    for (int i = 0; i < V; ++i) {
        const float r = p[i] / q[i];
        if (r > bv) { bv = r; best = i; }
    }
    return best;
}

extern "C" int hqt_cpu_sample(hqt_cpu_handle* h, int B, const int64_t* cond, const hqt_sample_opts* o, const float* noise,
                              const int64_t* force_top, const int64_t* force_bot, float* logits_out, int64_t* out_top, int64_t* out_bot) {
    if (!h || !o || !out_top || !out_bot) return fail(HQT_ERR_INVALID, "null argument");
    if (!h->finalized || !h->c.has_stage2) return fail(HQT_ERR_STATE, "hqt_cpu_sample needs finalized stage-2 weights");
    const hqt_config& c = h->c;
    const int D = c.embed_dim, V = c.vocab_top, n_steps = o->n_steps;
    if (B < 1 || n_steps < 1 || n_steps > c.ctx_len_img) return fail(HQT_ERR_INVALID, "bad batch / n_steps");
    if (c.cond_type != HQT_COND_NONE && !cond) return fail(HQT_ERR_INVALID, "cond is required");
    const auto t_start = std::chrono::steady_clock::now();
    const int ctx = c.cond_type == HQT_COND_TEXT ? c.ctx_len_txt : 0;
    const int T0 = ctx > 0 ? ctx : 1, Tcap = ctx + n_steps, Mmax = B * std::max(T0, 4);
    const float *tok_top, *tok_bot, *pos_top, *pos_emb = nullptr, *sos_depth, *tok_top_depth, *pos_depth, *lnf_g, *lnf_b, *lnt_g, *lnt_b, *lnb_g, *lnb_b;
    const float *sos = nullptr, *tok_txt = nullptr, *pos_txt = nullptr;
    const int64_t Dl = D;
    const int bot_dim = c.embedding_type == HQT_EMB_REDUCE ? D / 4 : D;
    CHK(get(h, "stage2.tok_emb_top.weight", {(int64_t)V, Dl}, &tok_top));
    CHK(get(h, "stage2.tok_emb_bot.weight", {(int64_t)c.vocab_bot, (int64_t)bot_dim}, &tok_bot));
    CHK(get(h, "stage2.pos_emb_top.weight", {(int64_t)c.ctx_len_img, Dl}, &pos_top));
    if (c.embedding_type == HQT_EMB_TRANSFORMER1) CHK(get(h, "stage2.pos_emb_emb.weight", {5, Dl}, &pos_emb));
    CHK(get(h, "stage2.sos_depth", {1, 1, Dl}, &sos_depth));
    CHK(get(h, "stage2.tok_emb_top_depth.weight", {(int64_t)V, Dl}, &tok_top_depth));
    CHK(get(h, "stage2.pos_emb_depth.weight", {5, Dl}, &pos_depth));
    CHK(get(h, "stage2.ln_f.weight", {Dl}, &lnf_g)); CHK(get(h, "stage2.ln_f.bias", {Dl}, &lnf_b));
    CHK(get(h, "stage2.ln_top.weight", {Dl}, &lnt_g)); CHK(get(h, "stage2.ln_top.bias", {Dl}, &lnt_b));
    CHK(get(h, "stage2.ln_bot.weight", {Dl}, &lnb_g)); CHK(get(h, "stage2.ln_bot.bias", {Dl}, &lnb_b));
    if (c.cond_type == HQT_COND_CLASS) CHK(get(h, "stage2.sos.weight", {(int64_t)c.n_classes, Dl}, &sos));
    else if (c.cond_type == HQT_COND_TEXT) {
        CHK(get(h, "stage2.tok_emb_txt.weight", {(int64_t)c.vocab_txt, Dl}, &tok_txt));
        CHK(get(h, "stage2.pos_emb_txt.weight", {(int64_t)c.ctx_len_txt, Dl}, &pos_txt));
    } else CHK(get(h, "stage2.sos", {1, 1, Dl}, &sos));

    Work k;
    k.x.resize((size_t)Mmax * D); k.hbuf.resize((size_t)Mmax * D); k.qkv.resize((size_t)Mmax * 3 * D); k.att.resize((size_t)Mmax * D);
    k.mlp.resize((size_t)Mmax * 4 * D); k.xd.resize((size_t)B * 4 * D); k.logits.resize((size_t)B * 4 * V);
    std::vector<float> kcache((size_t)c.n_layers * B * Tcap * D), vcache(kcache.size());
    std::vector<float> dk((size_t)c.n_layers_depth * B * 5 * D), dv(dk.size());
    std::vector<float> hs((size_t)B * D), qrow;
    std::vector<int64_t> cur_top(B), cur_bot((size_t)B * 4);

    auto draw = [&](const float* lg /*[rows = B][V] at stride ld*/, long ld, int cnt, int d, float temperature, int top_k, float top_p, int64_t* dst, int dst_stride) {
#pragma omp parallel num_threads(h->T)
        {
            std::vector<float> p, q(V);
            std::vector<int> order;
#pragma omp for schedule(static)
            for (int b = 0; b < B; ++b) {
                const float* qq;
                if (noise) qq = noise + (((size_t)cnt * 5 + d) * B + b) * V;
                else {
                    const uint64_t seed = o->row_seeds ? o->row_seeds[b] : o->seed;
                    const uint64_t grow = (uint64_t)(o->row_offsets ? o->row_offsets[b] : o->sample_offset + b);
                    philox_row(seed, grow, cnt * 5 + d, V, q.data());
                    qq = q.data();
                }
                dst[(size_t)b * dst_stride] = sample_row(lg + (size_t)b * ld, qq, V, temperature, top_k, top_p, p, order);
                if (logits_out) memcpy(logits_out + (((size_t)cnt * 5 + d) * B + b) * V, lg + (size_t)b * ld, (size_t)V * 4);
            }
        }
    };

    for (int cnt = 0; cnt < n_steps; ++cnt) {
        int T = 1;
        float* x = k.x.data();
        if (cnt == 0) {                                                  // sampling.py:183-192
            if (c.cond_type == HQT_COND_TEXT) {
                T = ctx;
                for (int b = 0; b < B; ++b)
                    for (int t = 0; t < ctx; ++t) {
                        const int64_t id = cond[(size_t)b * ctx + t];
                        if (id < 0 || id >= c.vocab_txt) return fail(HQT_ERR_INVALID, "text id out of range");
                        for (int i = 0; i < D; ++i) x[((size_t)b * ctx + t) * D + i] = tok_txt[(size_t)id * D + i] + pos_txt[(size_t)t * D + i];
                    }
            } else {
                for (int b = 0; b < B; ++b) {
                    int64_t id = 0;
                    if (c.cond_type == HQT_COND_CLASS) { id = cond[b]; if (id < 0 || id >= c.n_classes) return fail(HQT_ERR_INVALID, "class id out of range"); }
                    memcpy(x + (size_t)b * D, sos + (size_t)id * D, (size_t)D * 4);
                }
            }
        } else {                                                         // hierarchical_ar.py:505-544: input embedding from the 1 + 4 codes of cnt - 1
            const int pos = cnt - 1;
            for (int b = 0; b < B; ++b) {
                const int64_t ct = force_top ? force_top[(size_t)b * n_steps + pos] : cur_top[b];
                const int64_t* cb = force_bot ? force_bot + ((size_t)b * n_steps + pos) * 4 : &cur_bot[(size_t)b * 4];
                float* xr = x + (size_t)b * D;
                if (c.embedding_type == HQT_EMB_REDUCE) {                // :522-526: channel k * 4 + slot
                    for (int i = 0; i < D; ++i) xr[i] = (tok_top[(size_t)ct * D + i] + pos_top[(size_t)pos * D + i]) + tok_bot[(size_t)cb[i & 3] * bot_dim + (i >> 2)];
                } else {                                                 // :535-544: mean of the five embedded tokens
                    for (int i = 0; i < D; ++i) {
                        float s = (tok_top[(size_t)ct * D + i] + pos_top[(size_t)pos * D + i]) + pos_emb[i];
                        for (int sl = 0; sl < 4; ++sl) s += tok_bot[(size_t)cb[sl] * D + i] + pos_emb[(size_t)(sl + 1) * D + i];
                        xr[i] = s / 5.0f;
                    }
                }
            }
        }
        const int past = cnt == 0 ? 0 : (ctx > 0 ? ctx + cnt - 1 : cnt);   // keys already cached: the prompt (or the sos token) + the earlier positions
        for (int l = 0; l < c.n_layers; ++l)
            block(h, h->body[l], k, x, B, T, kcache.data() + (size_t)l * B * Tcap * D, vcache.data() + (size_t)l * B * Tcap * D, Tcap, past, true);
        // ln_f, then the predicting position (hierarchical_ar.py:561-563, 684-685: idx_pred - 1 = the last prompt token)
        if (T > 1) {
            std::vector<float> last((size_t)B * D);
            for (int b = 0; b < B; ++b) memcpy(&last[(size_t)b * D], x + ((size_t)b * T + T - 1) * D, (size_t)D * 4);
            layer_norm(h, last.data(), lnf_g, lnf_b, hs.data(), B, D);
        } else layer_norm(h, x, lnf_g, lnf_b, hs.data(), B, D);
        // depth sub-step 0: the top code (hierarchical_ar.py:682-695, 762-775)
        float* xd = k.xd.data();
        for (int b = 0; b < B; ++b)
            for (int i = 0; i < D; ++i) xd[(size_t)b * D + i] = hs[(size_t)b * D + i] + sos_depth[i];
        for (int l = 0; l < c.n_layers_depth; ++l)
            block(h, h->depth[l], k, xd, B, 1, dk.data() + (size_t)l * B * 5 * D, dv.data() + (size_t)l * B * 5 * D, 5, 0, false);
        layer_norm(h, xd, lnt_g, lnt_b, k.hbuf.data(), B, D);
        linear(h, k.hbuf.data(), D, h->head_top, k.logits.data(), V, B, 0, nullptr);
        draw(k.logits.data(), V, cnt, 0, o->temperature_top, o->top_k_top, o->top_p_top, cur_top.data(), 1);
        for (int b = 0; b < B; ++b) out_top[(size_t)b * n_steps + cnt] = cur_top[b];
        // depth sub-step 1: the four bottom codes in one pass (hierarchical_ar.py:696-719, 776-785)
        for (int b = 0; b < B; ++b) {
            const int64_t fed = force_top ? force_top[(size_t)b * n_steps + cnt] : cur_top[b];
            for (int s = 0; s < 4; ++s)
                for (int i = 0; i < D; ++i) xd[((size_t)b * 4 + s) * D + i] = tok_top_depth[(size_t)fed * D + i] + pos_depth[(size_t)s * D + i];
        }
        for (int l = 0; l < c.n_layers_depth; ++l)
            block(h, h->depth[l], k, xd, B, 4, dk.data() + (size_t)l * B * 5 * D, dv.data() + (size_t)l * B * 5 * D, 5, 1, false);
        layer_norm(h, xd, lnb_g, lnb_b, k.hbuf.data(), 4 * B, D);
        linear(h, k.hbuf.data(), D, h->head_bot, k.logits.data(), V, 4 * B, 0, nullptr);
        for (int s = 0; s < 4; ++s)                                       // slot order 0 .. 3 (:778)
            draw(k.logits.data() + (size_t)s * V, 4L * V, cnt, 1 + s, o->temperature_bot, o->top_k_bot, o->top_p_bot, cur_bot.data() + s, 4);
        for (int b = 0; b < B; ++b)
            for (int s = 0; s < 4; ++s) out_bot[((size_t)b * n_steps + cnt) * 4 + s] = cur_bot[(size_t)b * 4 + s];
    }
    h->last_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    return HQT_OK;
}

// ---------------------------------------------------------------------------------------------- stage 1
struct Img {                 // NHWC fp32
    int B, H, W, C;
    std::vector<float> d;
    Img(int b, int hh, int ww, int cc) : B(b), H(hh), W(ww), C(cc), d((size_t)b * hh * ww * cc) {}
};

// nn.Conv2d, stride 1, 'same' zero padding, kernel 1 or 3 (stage1/modules/layers.py:40-44,88-98); up: nearest x2 in front of the
// conv (F.interpolate, :50) folded into the gather (src = dst >> 1); resid: added to the result (the ResnetBlock / AttnBlock skip)
static void conv(const hqt_cpu_handle* h, const ConvW& cw, const Img& in, Img& out, bool up, const Img* resid) {
    const int Cin = in.C, O = cw.l.N, K = cw.l.K, Ho = out.H, Wo = out.W;
    const int PB = std::min(Wo, 32), xb = cdiv(Wo, PB);
    const long tasks = (long)in.B * Ho * xb;
#pragma omp parallel num_threads(h->T)
    {
        std::vector<float> col(cw.taps == 9 ? (size_t)PB * K : 0);
#pragma omp for schedule(dynamic, 4)
        for (long task = 0; task < tasks; ++task) {
            const int b = (int)(task / ((long)Ho * xb)), y = (int)((task / xb) % Ho), x0 = (int)(task % xb) * PB, np = std::min(PB, Wo - x0);
            float* dst = out.d.data() + (((size_t)b * Ho + y) * Wo + x0) * O;
            if (cw.taps == 1) {
                g_gemm(in.d.data() + (((size_t)b * in.H + y) * in.W + x0) * Cin, Cin, cw.l.w, K, dst, O, np, O, K);
            } else {
                for (int p = 0; p < np; ++p)
                    for (int dy = 0; dy < 3; ++dy)
                        for (int dx = 0; dx < 3; ++dx) {
                            const int yy = y + dy - 1, xx = x0 + p + dx - 1;
                            float* c = col.data() + (size_t)p * K + (dy * 3 + dx) * Cin;
                            if (yy < 0 || yy >= Ho || xx < 0 || xx >= Wo) memset(c, 0, (size_t)Cin * 4);
                            else memcpy(c, in.d.data() + (((size_t)b * in.H + (up ? yy >> 1 : yy)) * in.W + (up ? xx >> 1 : xx)) * Cin, (size_t)Cin * 4);
                        }
                g_gemm(col.data(), K, cw.l.w, K, dst, O, np, O, K);
            }
            const float* r = resid ? resid->d.data() + (((size_t)b * Ho + y) * Wo + x0) * O : nullptr;
            for (int p = 0; p < np; ++p)
                for (int o = 0; o < O; ++o) {
                    float v = dst[(size_t)p * O + o] + cw.l.b[o];
                    if (r) v = r[(size_t)p * O + o] + v;
                    dst[(size_t)p * O + o] = v;
                }
        }
    }
}

// GroupNorm(32, eps 1e-6, affine) (+ swish x * sigmoid(x)) -- stage1/modules/layers.py:12-21; statistics per (sample, group) in double
static void group_norm(const hqt_cpu_handle* h, const Img& in, Img& out, const float* g, const float* bta, bool swish) {
    const int G = 32, C = in.C, cg = C / G, HW = in.H * in.W;
    const int chunks = std::max(1, std::min(HW / 64, cdiv(4 * h->T, in.B)));
    std::vector<double> part((size_t)in.B * chunks * G * 2, 0.0), stat((size_t)in.B * G * 2);
    for (int pass = 0; pass < 2; ++pass) {                               // pass 0: means; pass 1: centred second moments
#pragma omp parallel for num_threads(h->T) schedule(static) collapse(2)
        for (int b = 0; b < in.B; ++b)
            for (int ch = 0; ch < chunks; ++ch) {
                const int p0 = (int)((long)HW * ch / chunks), p1 = (int)((long)HW * (ch + 1) / chunks);
                double acc[32] = {0};
                for (int p = p0; p < p1; ++p) {
                    const float* px = in.d.data() + ((size_t)b * HW + p) * C;
                    for (int gi = 0; gi < G; ++gi) {
                        const double mu = pass ? stat[((size_t)b * G + gi) * 2] : 0.0;
                        double s = 0.0;
                        for (int e = 0; e < cg; ++e) { const double d = px[gi * cg + e] - mu; s += pass ? d * d : d; }
                        acc[gi] += s;
                    }
                }
                for (int gi = 0; gi < G; ++gi) part[(((size_t)b * chunks + ch) * G + gi) * 2 + pass] = acc[gi];
            }
        for (int b = 0; b < in.B; ++b)
            for (int gi = 0; gi < G; ++gi) {
                double s = 0.0;
                for (int ch = 0; ch < chunks; ++ch) s += part[(((size_t)b * chunks + ch) * G + gi) * 2 + pass];
                stat[((size_t)b * G + gi) * 2 + pass] = s / ((double)HW * cg);
            }
    }
#pragma omp parallel for num_threads(h->T) schedule(static)
    for (long bp = 0; bp < (long)in.B * HW; ++bp) {
        const int b = (int)(bp / HW);
        const float* px = in.d.data() + (size_t)bp * C;
        float* o = out.d.data() + (size_t)bp * C;
        for (int ch = 0; ch < C; ++ch) {
            const double mu = stat[((size_t)b * G + ch / cg) * 2], sd = sqrt(stat[((size_t)b * G + ch / cg) * 2 + 1] + 1e-6);
            float v = (float)((px[ch] - mu) / sd) * g[ch] + bta[ch];
            if (swish) v = v / (1.0f + expf(-v));
            o[ch] = v;
        }
    }
}

static int gnw(hqt_cpu_handle* h, const std::string& name, int C, const float** g, const float** b) {
    CHK(get(h, "stage1." + name + ".weight", {(int64_t)C}, g));
    return get(h, "stage1." + name + ".bias", {(int64_t)C}, b);
}

// ResnetBlock.forward (stage1/modules/layers.py:115-133)
static int resblock(hqt_cpu_handle* h, const DecStep& s, Img& x) {
    const float *g, *b;
    Img t(x.B, x.H, x.W, s.cin), u(x.B, x.H, x.W, s.cout), v(x.B, x.H, x.W, s.cout);
    CHK(gnw(h, s.name + ".norm1", s.cin, &g, &b));
    group_norm(h, x, t, g, b, true);
    conv(h, h->conv[s.name + ".conv1"], t, u, false, nullptr);
    CHK(gnw(h, s.name + ".norm2", s.cout, &g, &b));
    group_norm(h, u, v, g, b, true);
    if (s.cin != s.cout) {
        Img sc(x.B, x.H, x.W, s.cout);
        conv(h, h->conv[s.name + ".nin_shortcut"], x, sc, false, nullptr);
        conv(h, h->conv[s.name + ".conv2"], v, u, false, &sc);
    } else conv(h, h->conv[s.name + ".conv2"], v, u, false, &x);
    x = std::move(u);
    return HQT_OK;
}

// AttnBlock.forward (stage1/modules/layers.py:163-186): single head, scale C^-0.5 applied AFTER q.k, softmax over the keys
static int attnblock(hqt_cpu_handle* h, const DecStep& s, Img& x) {
    const float *g, *b;
    const int C = s.cin, HW = x.H * x.W;
    Img t(x.B, x.H, x.W, C), q(x.B, x.H, x.W, C), kk(x.B, x.H, x.W, C), v(x.B, x.H, x.W, C), o(x.B, x.H, x.W, C), y(x.B, x.H, x.W, C);
    CHK(gnw(h, s.name + ".norm", C, &g, &b));
    group_norm(h, x, t, g, b, false);
    conv(h, h->conv[s.name + ".q"], t, q, false, nullptr);
    conv(h, h->conv[s.name + ".k"], t, kk, false, nullptr);
    conv(h, h->conv[s.name + ".v"], t, v, false, nullptr);
    const float scale = (float)pow((double)C, -0.5);
    const int QB = 32, qb = cdiv(HW, QB);
#pragma omp parallel num_threads(h->T)
    {
        std::vector<float> sc((size_t)QB * HW), vt;
#pragma omp for schedule(dynamic, 1)
        for (long task = 0; task < (long)x.B * qb; ++task) {
            const int bi = (int)(task / qb), q0 = (int)(task % qb) * QB, nq = std::min(QB, HW - q0);
            const float* qp = q.d.data() + ((size_t)bi * HW + q0) * C;
            const float* kp = kk.d.data() + (size_t)bi * HW * C;
            const float* vp = v.d.data() + (size_t)bi * HW * C;
            g_gemm(qp, C, kp, C, sc.data(), HW, nq, HW, C);              // scores[i][j] = q_i . k_j
            vt.resize((size_t)C * HW);
            for (int j = 0; j < HW; ++j)
                for (int ch = 0; ch < C; ++ch) vt[(size_t)ch * HW + j] = vp[(size_t)j * C + ch];
            for (int i = 0; i < nq; ++i) {
                float* r = sc.data() + (size_t)i * HW;
                float mx = -INFINITY;
                for (int j = 0; j < HW; ++j) { r[j] *= scale; mx = std::max(mx, r[j]); }
                float den = 0.0f;
                for (int j = 0; j < HW; ++j) { r[j] = expf(r[j] - mx); den += r[j]; }
                for (int j = 0; j < HW; ++j) r[j] = r[j] / den;
            }
            g_gemm(sc.data(), HW, vt.data(), HW, o.d.data() + ((size_t)bi * HW + q0) * C, C, nq, C, HW);   // o[i][c] = sum_j w[i][j] v[j][c]
        }
    }
    conv(h, h->conv[s.name + ".proj_out"], o, y, false, &x);
    x = std::move(y);
    return HQT_OK;
}

static int decode_impl(hqt_cpu_handle* h, int B, const int64_t* code_t, const int64_t* code_b, float* out, int clamp01, bool seq) {
    if (!h || !out || (!code_t && !code_b)) return fail(HQT_ERR_INVALID, "null argument");
    if (!h->finalized || !h->c.has_stage1) return fail(HQT_ERR_STATE, "hqt_cpu_decode needs finalized stage-1 weights");
    const auto t_start = std::chrono::steady_clock::now();
    const hqt_config& c = h->c;
    const int E = c.s1_embed_dim, nE = c.s1_n_embed;
    const int r = c.s1_resolution >> (c.s1_n_mult - 1 + (c.s1_use_init_downsample ? 1 : 0)), rt = r / 2;
    const float *et, *eb;
    CHK(get(h, "stage1.quantize_t.embedding", {(int64_t)nE, 4L * E}, &et));
    CHK(get(h, "stage1.quantize_b.embedding", {(int64_t)nE, (int64_t)E}, &eb));
    // codebook rows (quantizer.py:179-186), PixelShuffle(2) of the top level (out[c, 2h+i, 2w+j] = in[4c+2i+j, h, w]) and the concat
    // (generator.py:312-318); seq: the sampler's layouts, 'B (H W) -> B H W' / 'B (H W) (kh kw) -> B (H kh) (W kw)' (sampling_hqmodel.py:119-120)
    Img z(B, r, r, 2 * E);
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < r; ++y)
            for (int x = 0; x < r; ++x) {
                float* px = z.d.data() + (((size_t)b * r + y) * r + x) * 2 * E;
                const int hy = y >> 1, hx = x >> 1, sub = 2 * (y & 1) + (x & 1);
                if (code_t) {
                    const int64_t ct = code_t[((size_t)b * rt + hy) * rt + hx];
                    if (ct < 0 || ct >= nE) return fail(HQT_ERR_INVALID, "code out of range");
                    for (int ch = 0; ch < E; ++ch) px[ch] = et[(size_t)ct * 4 * E + 4 * ch + sub];
                } else memset(px, 0, (size_t)E * 4);
                if (code_b) {
                    const int64_t cb = seq ? code_b[(((size_t)b * rt + hy) * rt + hx) * 4 + sub] : code_b[((size_t)b * r + y) * r + x];
                    if (cb < 0 || cb >= nE) return fail(HQT_ERR_INVALID, "code out of range");
                    memcpy(px + E, eb + (size_t)cb * E, (size_t)E * 4);
                } else memset(px + E, 0, (size_t)E * 4);
            }
    Img x(B, r, r, c.s1_z_channels);
    conv(h, h->conv["post_quant_conv_b"], z, x, false, nullptr);
    for (const DecStep& s : decoder_plan(c)) {                            // Decoder.forward, stage1/modules/layers.py:385-410
        if (s.kind == 0) {
            Img y(B, x.H, x.W, s.cout);
            conv(h, h->conv[s.name], x, y, false, nullptr);
            x = std::move(y);
        } else if (s.kind == 1) CHK(resblock(h, s, x));
        else if (s.kind == 2) CHK(attnblock(h, s, x));
        else if (s.kind == 3) {
            Img y(B, 2 * x.H, 2 * x.W, s.cout);
            conv(h, h->conv[s.name], x, y, true, nullptr);
            x = std::move(y);
        } else {
            const float *g, *b;
            CHK(gnw(h, "decoder.norm_out", s.cin, &g, &b));
            Img t(B, x.H, x.W, s.cin), y(B, x.H, x.W, s.cout);
            group_norm(h, x, t, g, b, true);
            conv(h, h->conv[s.name], t, y, false, nullptr);
            const int HW = y.H * y.W, O = s.cout;
#pragma omp parallel for num_threads(h->T) schedule(static)
            for (long bp = 0; bp < (long)B * HW; ++bp)
                for (int o = 0; o < O; ++o) {
                    float v = y.d[(size_t)bp * O + o];
                    if (clamp01) v = std::min(std::max(0.5f * v + 0.5f, 0.0f), 1.0f);   // measure_throughput/__main__.py:113
                    out[((size_t)(bp / HW) * O + o) * HW + bp % HW] = v;
                }
        }
    }
    h->last_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    return HQT_OK;
}
extern "C" int hqt_cpu_decode(hqt_cpu_handle* h, int B, const int64_t* code_t, const int64_t* code_b, float* out, int clamp01) {
    return decode_impl(h, B, code_t, code_b, out, clamp01, false);
}
extern "C" int hqt_cpu_decode_seq(hqt_cpu_handle* h, int B, const int64_t* codes_top, const int64_t* codes_bot, float* out, int clamp01) {
    return decode_impl(h, B, codes_top, codes_bot, out, clamp01, true);
}
