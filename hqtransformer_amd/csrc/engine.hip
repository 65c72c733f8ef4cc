// libhqt.so host side: handle, weight registry, workspaces and the launch sequences behind the C ABI
// of include/hqt.h.  No CPU compute path exists here: every tensor operation is a HIP kernel.
#include "../../include/hqt.h"
#include "kernels.h"
#include "fast_kernels.h"
#include "split_kernels.h"
#include "persist.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return fail(HQT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                          __FILE__, __LINE__);                                              \
    } while (0)
#define CHK(expr)                  \
    do {                           \
        int r_ = (expr);           \
        if (r_ != HQT_OK) return r_; \
    } while (0)

// Entry points run on the handle's device and put the caller's current device back on return (C-ABI callers that drive several
// devices from one thread would otherwise find it silently changed).
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = prev == dev || hipSetDevice(dev) == hipSuccess;
        if (prev == dev) prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) hipSetDevice(prev); }
};
#define ON_DEVICE(h_)                                                                          \
    DeviceGuard dg_((h_)->device);                                                             \
    if (!dg_.ok) return fail(HQT_ERR_HIP, "hipSetDevice(%d) failed", (h_)->device)

struct Tensor {
    float* d = nullptr;
    std::vector<int64_t> shape;
    size_t n = 0;
};

struct Lin {            // one nn.Linear / conv filter bank in every layout the kernels read
    const float* w32 = nullptr;   // [N, K] fp32 (conv: tap-major [O][tap][I])
    const float* b32 = nullptr;   // [N] or null
    bf16_t* w16 = nullptr;        // [N, K] bf16 row-major (FAST generic + conv MFMA)
    half_t* w16h = nullptr;       // [N, K] fp16 hi / lo planes of w32 (SPLIT convolutions: split_kernels.h)
    half_t* w16l = nullptr;
    float* w32t = nullptr;        // fp32 copy in MFMA fragment order [N / 16][K / 32][chunk][lane][4] (EXACT AR loop at small row counts: exact_gemm.hip)
    half_t* wfrag16 = nullptr;    // 3x3 filters, hi + lo, packed in MFMA fragment order (16-channel blocks of v_mfma_f32_16x16x32_f16; split_stream_conv.hip)
    half_t* wup16 = nullptr;      // upsampling convs: the four 2x2 phase filters (pre-summed taps), same packing
    bf16_t* wpk = nullptr;        // MFMA-fragment-packed bf16 for the weight-streaming GEMM (FAST AR)
    bf16_t* wpk_ln = nullptr;     // same packing of gamma o W (deferred LayerNorm), with its column sums and folded bias
    float* colsum = nullptr;
    float* bias_ln = nullptr;
    int N = 0, K = 0;
};

struct BlockW {
    bool body = false;       // a block of the top GPT (its weights are read once per position), not of the depth head
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    Lin qkv, proj, fc1, fc2;
};

struct DecLayer {
    int kind;            // 0 conv3, 1 res, 2 attn, 3 upconv, 4 out; encoder: 5 Downsample, 6 conv_in, 7 out
    std::string name;
    int cin, cout, res;
    Lin conv1, conv2, nin, q, k, v, proj;      // conv3/upconv/out use conv1
    const float *n1_g = nullptr, *n1_b = nullptr, *n2_g = nullptr, *n2_b = nullptr;
};

// One program of the persistent AR chain (persist.h): the weight streams belong to the root handle, the phase table holds
// workspace pointers and is bound per handle (a clone binds its own).
struct PersistProg {
    std::vector<PersistPhase> phases;         // host copy; pointers filled by persist_bind
    char* stream = nullptr;
    unsigned long long* d_cu_off = nullptr;
    PersistPhase* d_phases = nullptr;
    bool ok = false;
};

struct TimingSlot {
    std::string name;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    size_t used = 0;
    int64_t launches = 0;
    double total_ms = 0.0;
};

struct hqt_handle {
    hqt_config cfg;
    int device = 0;
    bool finalized = false;
    std::map<std::string, Tensor> w;
    std::vector<void*> owned;                 // every device allocation, freed in destroy
    size_t workspace_bytes = 0;
    int64_t params[3] = {0, 0, 0};
    // ---- stage 2
    std::vector<BlockW> body, depth;
    Lin head_top, head_bot;
    int Tmax = 0;
    float *x = nullptr, *xd = nullptr, *logits = nullptr;
    void *hbuf = nullptr, *qbuf = nullptr, *abuf = nullptr, *mbuf = nullptr;    // fp32-sized, reused as bf16 in FAST
    void *kcache = nullptr, *vcache = nullptr, *dk = nullptr, *dv = nullptr;
    bf16_t *xpk = nullptr, *xdpk = nullptr;   // bf16 packed copies of the residual streams (FAST deferred LayerNorm)
    float *parts = nullptr, *partsd = nullptr; // their partial row statistics [D/32][Mpad][2]
    int nparts = 0, npartsd = 0;
    float* fold_tmp = nullptr;
    int resid_nparts = 0;                     // partial row statistics the last STORE_RESID GEMM left per row (run_linear)
    bool tile_gemm = true;                    // merged passes through the LDS-tiled MFMA kernels (HQT_NO_TILE_GEMM=1: streaming kernels at every row count)
    float* splitk = nullptr;                  // split-K partial slabs of the streaming GEMM
    size_t splitk_elems = 0;
    struct { const float* slabs; int S; int rows; const float* bias; } pend = {nullptr, 0, 0, nullptr};   // folded in by the next LayerNorm
    StepState* state = nullptr;
    RowKey* rows = nullptr;                   // [max_batch] Philox seed + global row of every batch row of the current call
    // per-row keys of merged steps travel through a ring of PINNED staging buffers (an asynchronous copy from pinned memory reads its
    // source when the stream gets there: a buffer is rewritten only after the copy that last read it has completed, hipEventSynchronize)
    static constexpr int ROWS_RING = 4;
    RowKey* rows_pinned[ROWS_RING] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t rows_ev[ROWS_RING] = {nullptr, nullptr, nullptr, nullptr};
    bool rows_busy[ROWS_RING] = {false, false, false, false};
    int rows_next = 0;
    int64_t *cond_buf = nullptr, *codes_top = nullptr, *codes_bot = nullptr;   // call-independent homes of cond / the drawn codes
    int64_t* codes_l2 = nullptr;              // third level: [B, max_steps, 16]
    Lin head_l2;                              // head_levels.2 (three-level models; head_top / head_bot hold levels 0 / 1)
    // ---- stage 1
    std::vector<DecLayer> dec;
    std::vector<DecLayer> enc;               // Encoder.forward in execution order (kinds 6 conv_in, 1, 2, 5 Downsample, 7 norm_out + conv_out)
    bool has_encoder = false;                 // the encoder tensors were set before finalize: hqt_encode is usable
    Lin post_quant, quant_conv;
    float* cb_norm[3] = {nullptr, nullptr, nullptr};   // |e_n|^2 of each codebook (distance GEMM epilogue)
    float *vq_h = nullptr, *vq_recon = nullptr, *vq_zz = nullptr, *vq_err = nullptr;
    void* vq_z = nullptr;
    unsigned long long* vq_best = nullptr;
    void* act[4] = {nullptr, nullptr, nullptr, nullptr};   // 3 rotating activation buffers + the normalised/activated copy (FAST)
    double* gn_partial = nullptr;
    float* gn_tiles = nullptr;                // per-tile output statistics of the last halo conv ([image][tile][32][2])
    struct { const void* tensor; int tiles; bool dbl; } gn_ready = {nullptr, 0, false};   // dbl: double partials (SPLIT conv)
    void* zero_page = nullptr;
    int* range_flag = nullptr;                // set by the SPLIT operand pass when an activation leaves the fp16 range (hqt_range_check)
    size_t act_elems = 0;
    int dec_chunk = 0;
    void *aq = nullptr, *ak = nullptr, *av = nullptr, *ao = nullptr, *as = nullptr, *quant = nullptr;
    float* gn = nullptr;                      // [chunk][32][2] x 2
    // ---- graph cache
    hipGraphExec_t graph_exec = nullptr;
    std::vector<uint64_t> graph_key;
    // ---- lanes: a clone shares every weight buffer of its parent (non-owning) and owns only its workspace
    hqt_handle* parent = nullptr;
    int n_clones = 0;
    int policy = 0;                           // HQT_POLICY_*: tile choice of the streaming GEMMs (part of the graph key)
    // ---- persistent AR chain (FAST precision, up to 64 rows, latency policy, root handle): body blocks / depth sub-step 0 + head_top
    PersistProg pbody, pfull;                 // body only (three code levels) / the whole position up to the top logits (persist_build)
    float* lnf_shift = nullptr;               // ln_f.bias + sos_depth (PP_LNF)
    int ncu = 0;
    unsigned* persist_counters = nullptr;     // workspace: barrier + quad counters (zeroed by every launch)
    unsigned* persist_err = nullptr;          // workspace: set by a launch that gave up on a barrier
    float* persist_slabs = nullptr;           // workspace: K-split partials
    bool persist_used = false;                // a persistent launch was queued since the last check of persist_err
    bool persist_enabled = true;              // HQT_SWITCH_PERSIST (default: on unless HQT_PERSIST=0 was in the environment at hqt_create)
    bool persist_tripped = false;             // a persistent launch of this handle gave up (hqt_range_check): the launch chain until HQT_SWITCH_PERSIST re-arms it
    bool single_key = true;                   // HQT_SWITCH_SINGLE_KEY (default: on unless HQT_NO_SINGLE_KEY was in the environment at hqt_create)
    bool split_kslices = true;                // HQT_SWITCH_SPLIT_KSLICES (default: on unless HQT_SPLIT_KSLICES_OFF was in the environment at hqt_create)
    bool capture_persist = false;             // run_persist ran since sample_run cleared it (is a persistent launch inside the graph being captured?)
    bool graph_has_persist = false;           // the cached graph holds persistent launches: every replay marks persist_used
    int layouts = 0;                          // HQT_LAYOUT_* bits hqt_finalize_weights builds for the AR loop's nn.Linear weights
    // ---- timing
    bool timing = false;
    std::vector<TimingSlot> slots;
    hipEvent_t chain_ev = nullptr;
    bool chain_valid = false;
    std::vector<hipEvent_t> all_events;
};

static int dev_alloc(hqt_handle* h, void** p, size_t bytes, bool workspace) {
    HIPCHK(hipMalloc(p, bytes ? bytes : 16));
    // test hook: HQT_POISON_WORKSPACE=1 fills every workspace buffer with 0xFF (NaN as fp32 / bf16, -1 as int64) so that
    // a read of something no kernel wrote shows up instead of passing on freshly zeroed memory
    const bool poison = getenv("HQT_POISON_WORKSPACE") != nullptr;       // read per allocation: tests switch it on around one engine
    if (poison && workspace && bytes) HIPCHK(hipMemset(*p, 0xFF, bytes));
    h->owned.push_back(*p);
    if (workspace) h->workspace_bytes += bytes;
    return HQT_OK;
}

// ------------------------------------------------------------------------------------------ timing
static int slot_id(hqt_handle* h, const char* name) {
    for (size_t i = 0; i < h->slots.size(); ++i)
        if (h->slots[i].name == name) return (int)i;
    h->slots.push_back(TimingSlot());
    h->slots.back().name = name;
    return (int)h->slots.size() - 1;
}
// Per-launch timers (bench.py's roofline numerator): consecutive launches share one event -- the end event of a
// launch is the start event of the next -- so a timed pass adds one hipEventRecord per kernel.
struct Timed {
    hqt_handle* h; int slot; hipStream_t st; bool on;
    Timed(hqt_handle* h_, const char* name, hipStream_t st_) : h(h_), slot(-1), st(st_), on(h_->timing) {
        if (!on) return;
        slot = slot_id(h, name);
        if (!h->chain_valid) {
            hipEventCreate(&h->chain_ev);
            hipEventRecord(h->chain_ev, st);
            h->chain_valid = true;
        }
    }
    // the launches so far belong to this timer's slot; what follows goes to `name` (a GEMM and its split-K combine)
    void next(const char* name) {
        if (!on) return;
        hipEvent_t e;
        hipEventCreate(&e);
        hipEventRecord(e, st);
        h->slots[slot].ev.push_back({h->chain_ev, e});
        h->slots[slot].used++;
        h->all_events.push_back(h->chain_ev);
        h->chain_ev = e;
        slot = slot_id(h, name);
    }
    ~Timed() {
        if (!on) return;
        hipEvent_t e;
        hipEventCreate(&e);
        hipEventRecord(e, st);
        h->slots[slot].ev.push_back({h->chain_ev, e});
        h->slots[slot].used++;
        h->all_events.push_back(h->chain_ev);
        h->chain_ev = e;
    }
};
// Which kernel variant served a launch, as zero-time slots of the timing report ("variant:<kernel>:<shape class>"): counted only
// while timing is on (un-graphed passes), so tests and bench.py can state what the timed schedule ran.
static void count_variant(hqt_handle* h, const char* fmt, ...) {
    if (!h->timing) return;
    char buf[96];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    h->slots[slot_id(h, buf)].launches++;
}
static void timing_collect(hqt_handle* h) {
    if (h->chain_valid) hipEventSynchronize(h->chain_ev);
    for (auto& s : h->slots) {
        for (auto& pr : s.ev) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { s.total_ms += ms; s.launches++; }
        }
        s.ev.clear();
        s.used = 0;
    }
    for (hipEvent_t e : h->all_events) hipEventDestroy(e);
    h->all_events.clear();
    if (h->chain_valid) { hipEventDestroy(h->chain_ev); h->chain_valid = false; }
}

// ------------------------------------------------------------------------------------------ plan
static void build_decoder_plan(hqt_handle* h) {
    const hqt_config& c = h->cfg;
    const int n = c.s1_n_mult;
    int block_in = c.s1_ch * c.s1_ch_mult[n - 1];
    int res = c.s1_resolution >> (c.s1_use_init_downsample ? n : n - 1);
    auto is_attn_res = [&](int r) {
        for (int i = 0; i < c.s1_n_attn_res; ++i) if (c.s1_attn_res[i] == r) return true;
        return false;
    };
    auto add = [&](int kind, const std::string& name, int cin, int cout, int r) {
        DecLayer l; l.kind = kind; l.name = name; l.cin = cin; l.cout = cout; l.res = r;
        h->dec.push_back(l);
    };
    add(0, "decoder.conv_in", c.s1_z_channels, block_in, res);
    if (c.s1_use_mid_block) {
        add(1, "decoder.mid.block_1", block_in, block_in, res);
        if (c.s1_use_attn) add(2, "decoder.mid.attn_1", block_in, block_in, res);
        add(1, "decoder.mid.block_2", block_in, block_in, res);
    }
    for (int lvl = n - 1; lvl >= 0; --lvl) {
        const int block_out = c.s1_ch * c.s1_ch_mult[lvl];
        for (int b = 0; b <= c.s1_num_res_blocks; ++b) {
            add(1, "decoder.up." + std::to_string(lvl) + ".block." + std::to_string(b), block_in, block_out, res);
            block_in = block_out;
            if (is_attn_res(res) && c.s1_use_attn)
                add(2, "decoder.up." + std::to_string(lvl) + ".attn." + std::to_string(b), block_in, block_in, res);
        }
        if (lvl != 0 || c.s1_use_init_downsample) {
            add(3, "decoder.up." + std::to_string(lvl) + ".upsample.conv", block_in, block_in, res);
            res *= 2;
        }
    }
    add(4, "decoder", block_in, c.s1_out_ch, res);
}

// Encoder.forward (stage1/modules/layers.py:270-297).  The reference tracks `curr_res` from `resolution` even when conv_in
// already halves the image (layers.py:221), so with use_init_downsample the attention test runs on twice the real size.
static void build_encoder_plan(hqt_handle* h) {
    const hqt_config& c = h->cfg;
    const int n = c.s1_n_mult;
    auto is_attn_res = [&](int r) {
        for (int i = 0; i < c.s1_n_attn_res; ++i) if (c.s1_attn_res[i] == r) return true;
        return false;
    };
    auto add = [&](int kind, const std::string& name, int cin, int cout, int r) {
        DecLayer l; l.kind = kind; l.name = name; l.cin = cin; l.cout = cout; l.res = r;
        h->enc.push_back(l);
    };
    add(6, "encoder.conv_in", 3, c.s1_ch, c.s1_resolution);
    int label = c.s1_resolution;
    int res = c.s1_use_init_downsample ? c.s1_resolution / 2 : c.s1_resolution;
    int block_in = c.s1_ch;
    for (int lvl = 0; lvl < n; ++lvl) {
        const int block_out = c.s1_ch * c.s1_ch_mult[lvl];
        for (int b = 0; b < c.s1_num_res_blocks; ++b) {
            add(1, "encoder.down." + std::to_string(lvl) + ".block." + std::to_string(b), block_in, block_out, res);
            block_in = block_out;
            if (is_attn_res(label) && c.s1_use_attn)
                add(2, "encoder.down." + std::to_string(lvl) + ".attn." + std::to_string(b), block_in, block_in, res);
        }
        if (lvl != n - 1) {
            add(5, "encoder.down." + std::to_string(lvl) + ".downsample.conv", block_in, block_in, res);
            res /= 2; label /= 2;
        }
    }
    if (c.s1_use_mid_block) {
        add(1, "encoder.mid.block_1", block_in, block_in, res);
        if (c.s1_use_attn) add(2, "encoder.mid.attn_1", block_in, block_in, res);
        add(1, "encoder.mid.block_2", block_in, block_in, res);
    }
    add(7, "encoder", block_in, c.s1_z_channels, res);
}

// ------------------------------------------------------------------------------------------ create
extern "C" int hqt_abi_version(void) { return HQT_ABI_VERSION; }
extern "C" const char* hqt_last_error(void) { return g_err.c_str(); }

static int alloc_workspace(hqt_handle* hp);

extern "C" int hqt_create(const hqt_config* cfg, int device, hqt_handle** out) {
    if (!cfg || !out) return fail(HQT_ERR_INVALID, "null argument");
    if (cfg->abi_version != HQT_ABI_VERSION) return fail(HQT_ERR_INVALID, "abi_version %d != %d", cfg->abi_version, HQT_ABI_VERSION);
    if (cfg->max_batch < 1) return fail(HQT_ERR_INVALID, "max_batch must be >= 1");
    DeviceGuard dg(device);
    if (!dg.ok) return fail(HQT_ERR_HIP, "hipSetDevice(%d) failed", device);
    std::unique_ptr<hqt_handle> h(new hqt_handle());
    // every environment switch of the per-call path is read HERE, once per handle (hqt_set_switch changes them afterwards)
    h->tile_gemm = getenv("HQT_NO_TILE_GEMM") == nullptr;
    { const char* e = getenv("HQT_PERSIST"); h->persist_enabled = !(e && atoi(e) == 0); }
    h->single_key = getenv("HQT_NO_SINGLE_KEY") == nullptr;
    h->split_kslices = getenv("HQT_SPLIT_KSLICES_OFF") == nullptr;
    h->cfg = *cfg;
    h->layouts = (cfg->ar_layouts & HQT_LAYOUT_ALL) ? (cfg->ar_layouts & HQT_LAYOUT_ALL) : HQT_LAYOUT_ALL;
    h->device = device;
    const hqt_config& c = h->cfg;
    if (c.has_stage2) {
        if (c.embed_dim % c.n_heads || c.embed_dim % 16) return fail(HQT_ERR_INVALID, "embed_dim must be a multiple of n_heads and 16");
        const int hs = c.embed_dim / c.n_heads;
        if (hs % 8 || hs > 256 || (hs & (hs - 1))) return fail(HQT_ERR_INVALID, "head_dim %d unsupported (power of two in [8, 256])", hs);
        if (c.vocab_top != c.vocab_bot) return fail(HQT_ERR_INVALID, "vocab_top != vocab_bot");
        if (c.depth_decoding < 0 || c.depth_decoding > HQT_DEPTH_TOP2MID2BOT || (c.depth_decoding && c.code_levels != 3))
            return fail(HQT_ERR_INVALID, "depth_decoding %d: 0 'parallel-add', 1 'parallel', 2 'parallel-reduce', 3 'top2mid2bot' (three code levels only)", c.depth_decoding);
        if (c.vocab_top > HQT_MAX_V || c.vocab_top % 4) return fail(HQT_ERR_INVALID, "vocab size %d unsupported", c.vocab_top);
        if (c.cond_type == HQT_COND_CLASS && c.n_classes < 1) return fail(HQT_ERR_INVALID, "n_classes");
        if (c.max_steps < 1 || c.max_steps > c.ctx_len_img) return fail(HQT_ERR_INVALID, "max_steps must be in [1, ctx_len_img]");
    }
    if (c.has_stage1) {
        if (c.s1_n_mult < 1 || c.s1_n_mult > 8) return fail(HQT_ERR_INVALID, "s1_n_mult");
        if (c.s1_ch % 32) return fail(HQT_ERR_INVALID, "GroupNorm(32) needs ch %% 32 == 0");
        if (c.s1_z_channels % 16 || (2 * c.s1_embed_dim) % 16) return fail(HQT_ERR_INVALID, "z_channels and 2*embed_dim must be multiples of 16");
        build_decoder_plan(h.get());
        build_encoder_plan(h.get());
    }
    CHK(alloc_workspace(h.get()));
    *out = h.release();
    return HQT_OK;
}

// Everything a lane owns: activations, KV caches, step state, decoder buffers (the weights live in `w` / the Lin structs).
// nearest-code search state of hqt_encode: allocated once the encoder tensors are known to be present (finalize), and for clones
static int alloc_encode_workspace(hqt_handle* h) {
    const hqt_config& c = h->cfg;
    const int r = h->dec.front().res;
    const size_t rows = (size_t)c.max_batch * r * r, elems = rows * c.s1_embed_dim;
    CHK(dev_alloc(h, (void**)&h->vq_h, elems * 4, true));
    CHK(dev_alloc(h, (void**)&h->vq_recon, elems * 4, true));
    CHK(dev_alloc(h, &h->vq_z, elems * 4, true));
    CHK(dev_alloc(h, (void**)&h->vq_zz, rows * 4, true));
    CHK(dev_alloc(h, (void**)&h->vq_best, rows * 8, true));
    CHK(dev_alloc(h, (void**)&h->vq_err, rows * 4 * 3, true));
    return HQT_OK;
}

static int alloc_workspace(hqt_handle* hp) {
    struct { hqt_handle* p; hqt_handle* get() const { return p; } hqt_handle* operator->() const { return p; } } h{hp};
    const hqt_config& c = h->cfg;
    const size_t B = (size_t)c.max_batch;
    // source of padded rows / taps for the LDS-DMA kernels (stage-2 GEMMs with more than 256 rows use them too: the text
    // prefill and the third code level)
    CHK(dev_alloc(h.get(), &h->zero_page, 256, true));
    HIPCHK(hipMemset(h->zero_page, 0, 256));
    CHK(dev_alloc(h.get(), (void**)&h->range_flag, 256, true));
    HIPCHK(hipMemset(h->range_flag, 0, 256));
    if (c.has_stage2) {
        const size_t D = c.embed_dim;
        const int Tp = c.cond_type == HQT_COND_TEXT ? c.ctx_len_txt : 1;    // rows of the widest body pass
        h->Tmax = (c.cond_type == HQT_COND_TEXT ? c.ctx_len_txt : 0) + c.max_steps;
        const int Tdepth = c.code_levels == 3 ? 16 : 4;      // rows per sample of the widest depth sub-step
        const int Kdepth = c.code_levels == 3 ? 21 : 5;      // keys of the depth cache
        const size_t rows = (B * (size_t)std::max(Tp, Tdepth) + 31) / 32 * 32;
        CHK(dev_alloc(h.get(), (void**)&h->x, rows * D * 4, true));
        CHK(dev_alloc(h.get(), (void**)&h->xd, B * Tdepth * D * 4, true));
        CHK(dev_alloc(h.get(), &h->hbuf, rows * D * 4, true));
        CHK(dev_alloc(h.get(), &h->qbuf, rows * D * 4, true));
        CHK(dev_alloc(h.get(), &h->abuf, rows * D * 4, true));
        CHK(dev_alloc(h.get(), &h->mbuf, rows * 4 * D * 4, true));
        CHK(dev_alloc(h.get(), (void**)&h->logits, B * Tdepth * (size_t)c.vocab_top * 4, true));
        const size_t kv = (size_t)c.n_layers * B * h->Tmax * D * 4;
        CHK(dev_alloc(h.get(), &h->kcache, kv, true));
        CHK(dev_alloc(h.get(), &h->vcache, kv, true));
        const size_t dkv = (size_t)c.n_layers_depth * B * Kdepth * D * 4;
        CHK(dev_alloc(h.get(), &h->dk, dkv, true));
        CHK(dev_alloc(h.get(), &h->dv, dkv, true));
        h->splitk_elems = (size_t)16 * rows * (size_t)std::max<size_t>(4 * D, (size_t)c.vocab_top);
        CHK(dev_alloc(h.get(), (void**)&h->splitk, h->splitk_elems * 4, true));
        CHK(dev_alloc(h.get(), (void**)&h->state, sizeof(StepState), true));
        CHK(dev_alloc(h.get(), (void**)&h->rows, B * sizeof(RowKey), true));
        CHK(dev_alloc(h.get(), (void**)&h->cond_buf, B * (size_t)std::max(1, c.cond_type == HQT_COND_TEXT ? c.ctx_len_txt : 1) * 8, true));
        CHK(dev_alloc(h.get(), (void**)&h->codes_top, B * (size_t)c.max_steps * 8, true));
        CHK(dev_alloc(h.get(), (void**)&h->codes_bot, B * (size_t)c.max_steps * 4 * 8, true));
        if (c.code_levels == 3) CHK(dev_alloc(h.get(), (void**)&h->codes_l2, B * (size_t)c.max_steps * 16 * 8, true));
        // packed_off() buffers are addressed with a row stride of 32 * packed_mb(M) (32 / 64 / ... / 4096 rows), which can
        // exceed round32(M): size them for the widest padded block any M <= PACKED_MAX_ROWS pass can use
        size_t rows_pk = 256;
        while (rows_pk < rows && rows_pk < (size_t)PACKED_MAX_ROWS) rows_pk *= 2;
        rows_pk = std::max(rows_pk, rows);
        CHK(dev_alloc(h.get(), (void**)&h->xpk, rows_pk * D * 2, true));
        CHK(dev_alloc(h.get(), (void**)&h->xdpk, rows_pk * D * 2, true));
        CHK(dev_alloc(h.get(), (void**)&h->parts, (D / 32 + 1) * rows_pk * 2 * 4, true));
        CHK(dev_alloc(h.get(), (void**)&h->partsd, (D / 32 + 1) * rows_pk * 2 * 4, true));
        if (!h->parent) {                        // the persistent chain runs on root handles only (persist_on)
            CHK(dev_alloc(h.get(), (void**)&h->persist_counters, PERSIST_COUNTER_BYTES, true));
            CHK(dev_alloc(h.get(), (void**)&h->persist_err, 256, true));
            HIPCHK(hipMemset(h->persist_err, 0, 256));
            CHK(dev_alloc(h.get(), (void**)&h->persist_slabs, persist_slab_floats(256) * 4, true));
        } else {
            h->persist_counters = nullptr; h->persist_err = nullptr; h->persist_slabs = nullptr;
        }
    }
    if (c.has_stage1) {
        std::vector<DecLayer> both(h->dec);                 // decoder and encoder share the activation / attention / statistics buffers
        both.insert(both.end(), h->enc.begin(), h->enc.end());
        size_t per_img = 0;
        for (auto& l : both) {
            const size_t in_e = (size_t)l.res * l.res * (l.kind == 6 ? 16 : l.cin);       // conv_in reads the image padded to <= 16 channels
            const size_t out_r = l.kind == 3 ? 2 * l.res : l.res;
            const size_t out_e = out_r * out_r * (size_t)(l.kind == 4 ? 0 : l.cout);
            per_img = std::max(per_img, std::max(in_e, out_e));
        }
        h->dec_chunk = std::max(1, std::min<int>(c.max_batch, getenv("HQT_DEC_CHUNK") ? atoi(getenv("HQT_DEC_CHUNK")) : 64));     // images per decode pass.  Decoder alone 64 -> 128 -> 256: 2047 -> 2090 -> 2102 images/s (SPLIT), but in the pipeline 128 measured 0.4 % slower (1275 vs 1280 images/s, same box)
        h->act_elems = per_img * h->dec_chunk;
        for (int i = 0; i < 3; ++i) CHK(dev_alloc(h.get(), &h->act[i], h->act_elems * 4, true));
        CHK(dev_alloc(h.get(), &h->act[3], h->act_elems * 4, true));     // bf16 copy (FAST) or fp16 hi / lo planes (SPLIT)
        {
            size_t pe = 0;
            for (auto& l : both) if (l.kind != 6) pe = std::max(pe, gn_stats_fast_partial_elems(h->dec_chunk, l.res * l.res, l.cin, 32));
            for (auto& l : both) if (l.kind == 1) pe = std::max(pe, gn_stats_fast_partial_elems(h->dec_chunk, l.res * l.res, l.cout, 32));
            CHK(dev_alloc(h.get(), (void**)&h->gn_partial, pe * sizeof(double), true));
            size_t te = 0;
            for (auto& l : both) { const int ro = l.kind == 3 ? 2 * l.res : l.res; te = std::max(te, (size_t)h->dec_chunk * (ro / 8 + 1) * (ro / 16 + 1) * 64); }
            CHK(dev_alloc(h.get(), (void**)&h->gn_tiles, te * sizeof(double), true));     // floats (FAST) or doubles (SPLIT)
        }
        const int r = h->dec.front().res;
        size_t attn_c = 0;
        for (auto& l : both) if (l.kind == 2) attn_c = std::max(attn_c, (size_t)l.cin * l.res * l.res);
        const size_t hw = (size_t)r * r;
        if (attn_c) {
            CHK(dev_alloc(h.get(), &h->aq, attn_c * h->dec_chunk * 4, true));
            CHK(dev_alloc(h.get(), &h->ak, attn_c * h->dec_chunk * 4, true));
            CHK(dev_alloc(h.get(), &h->av, attn_c * h->dec_chunk * 4, true));
            CHK(dev_alloc(h.get(), &h->ao, attn_c * h->dec_chunk * 4, true));
            size_t smax = 0;
            for (auto& l : both) if (l.kind == 2) smax = std::max(smax, (size_t)l.res * l.res * l.res * l.res);
            CHK(dev_alloc(h.get(), &h->as, smax * h->dec_chunk * 4, true));
        }
        CHK(dev_alloc(h.get(), &h->quant, hw * 2 * c.s1_embed_dim * h->dec_chunk * 4, true));
        CHK(dev_alloc(h.get(), (void**)&h->gn, (size_t)h->dec_chunk * 32 * 2 * 2 * 4, true));
        if (h->has_encoder) CHK(alloc_encode_workspace(h.get()));            // clones of a handle with an encoder
    }
    return HQT_OK;
}

// A second lane over the same weights: several batches in flight on one GPU, one lane + stream each (the AR loop is a
// chain of small latency-bound kernels that leaves most CUs idle; independent chains interleave).  The clone shares
// every weight buffer and derived layout of `src` (which must be finalized and outlive it) and owns a fresh workspace.
extern "C" int hqt_clone(hqt_handle* src, hqt_handle** out) {
    if (!src || !out) return fail(HQT_ERR_INVALID, "null argument");
    if (!src->finalized) return fail(HQT_ERR_STATE, "hqt_clone needs a finalized handle");
    ON_DEVICE(src);
    hqt_handle* root = src->parent ? src->parent : src;
    std::unique_ptr<hqt_handle> h(new hqt_handle(*root));       // weights map, Lin structs and decoder plan by value: same device pointers
    h->owned.clear();
    h->workspace_bytes = 0;
    h->parent = root;
    h->n_clones = 0;
    h->graph_exec = nullptr;
    h->graph_key.clear();
    h->timing = false;
    h->slots.clear();
    h->chain_ev = nullptr;
    h->chain_valid = false;
    h->all_events.clear();
    h->pend = {nullptr, 0, 0, nullptr};
    for (int i = 0; i < hqt_handle::ROWS_RING; ++i) { h->rows_pinned[i] = nullptr; h->rows_ev[i] = nullptr; h->rows_busy[i] = false; }
    h->rows_next = 0;
    h->nparts = h->npartsd = 0;
    h->pbody.d_phases = nullptr; h->pfull.d_phases = nullptr;      // phase tables hold workspace pointers: bound per handle (persist_bind)
    h->persist_used = false; h->persist_tripped = false; h->capture_persist = false; h->graph_has_persist = false;
    h->policy = HQT_POLICY_LATENCY;              // a lane's tile choice never depends on what the root ran when it was cloned
    const int rc = alloc_workspace(h.get());
    if (rc != HQT_OK) { for (void* p : h->owned) hipFree(p); return rc; }
    root->n_clones++;
    *out = h.release();
    return HQT_OK;
}

extern "C" int hqt_destroy(hqt_handle* h) {
    if (!h) return HQT_OK;
    if (h->n_clones > 0) return fail(HQT_ERR_STATE, "%d clone(s) of this handle are still alive", h->n_clones);
    DeviceGuard dg(h->device);
    hipDeviceSynchronize();
    if (h->graph_exec) hipGraphExecDestroy(h->graph_exec);
    timing_collect(h);
    for (void* p : h->owned) hipFree(p);             // a clone owns only its workspace
    for (int i = 0; i < hqt_handle::ROWS_RING; ++i) {
        if (h->rows_pinned[i]) hipHostFree(h->rows_pinned[i]);
        if (h->rows_ev[i]) hipEventDestroy(h->rows_ev[i]);
    }
    if (h->parent) h->parent->n_clones--;
    delete h;
    return HQT_OK;
}

// ------------------------------------------------------------------------------------------ weights
extern "C" int hqt_set_weight(hqt_handle* h, const char* name, const void* data, int dtype, const int64_t* shape, int ndim) {
    if (!h || !name || !data || !shape || ndim < 1 || ndim > 4) return fail(HQT_ERR_INVALID, "bad argument");
    if (dtype != HQT_DTYPE_F32) return fail(HQT_ERR_INVALID, "only fp32 weights are accepted");
    if (h->finalized) return fail(HQT_ERR_STATE, "weights already finalized");
    ON_DEVICE(h);
    Tensor t;
    t.n = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); t.n *= (size_t)shape[i]; }
    auto it = h->w.find(name);
    if (it != h->w.end()) {
        if (it->second.n != t.n) return fail(HQT_ERR_SHAPE, "%s: re-set with a different size", name);
        t.d = it->second.d;
    } else {
        CHK(dev_alloc(h, (void**)&t.d, t.n * 4, false));
        const int stage = strncmp(name, "stage1.", 7) == 0 ? 1 : 2;
        h->params[stage] += (int64_t)t.n;
    }
    HIPCHK(hipMemcpy(t.d, data, t.n * 4, hipMemcpyDefault));
    h->w[name] = t;
    return HQT_OK;
}

static int get_w(hqt_handle* h, const std::string& name, std::vector<int64_t> shape, const float** out) {
    auto it = h->w.find(name);
    if (it == h->w.end()) return fail(HQT_ERR_MISSING_WEIGHT, "missing weight '%s'", name.c_str());
    if (it->second.shape != shape) {
        std::string got, want;
        for (auto v : it->second.shape) got += std::to_string(v) + ",";
        for (auto v : shape) want += std::to_string(v) + ",";
        return fail(HQT_ERR_SHAPE, "weight '%s' has shape [%s] expected [%s]", name.c_str(), got.c_str(), want.c_str());
    }
    *out = it->second.d;
    return HQT_OK;
}

static int make_lin(hqt_handle* h, Lin& l, const float* w32, const float* b32, int N, int K, bool stream_pack, bool split_planes = false) {
    l.w32 = w32; l.b32 = b32; l.N = N; l.K = K;
    if (!stream_pack || (h->layouts & HQT_LAYOUT_FAST)) {
        CHK(dev_alloc(h, (void**)&l.w16, (size_t)N * K * 2, false));
        HIPCHK(launch_f32_to_bf16(w32, l.w16, (size_t)N * K, 0));
    }
    // the AR loop's nn.Linear weights (stream_pack) build only the layouts the handle was created for (hqt_config.ar_layouts)
    const bool ar = stream_pack;
    if (split_planes && (!ar || (h->layouts & HQT_LAYOUT_SPLIT))) {
        CHK(dev_alloc(h, (void**)&l.w16h, (size_t)N * K * 2, false));
        CHK(dev_alloc(h, (void**)&l.w16l, (size_t)N * K * 2, false));
        HIPCHK(launch_split_f32(w32, l.w16h, l.w16l, (size_t)N * K, 0));
    }
    if (stream_pack && (h->layouts & HQT_LAYOUT_EXACT) && N % 16 == 0 && K % 32 == 0) {          // the AR loop's linears: tile-contiguous fp32 for the EXACT small-row GEMM
        CHK(dev_alloc(h, (void**)&l.w32t, (size_t)N * K * 4, false));
        HIPCHK(launch_pack_exact_tiles(w32, l.w32t, N, K, 0));
    }
    if (stream_pack && (h->layouts & HQT_LAYOUT_FAST) && stream_gemm_supported(N, K)) {
        CHK(dev_alloc(h, (void**)&l.wpk, (size_t)N * K * 2, false));
        HIPCHK(launch_pack_stream_weights(w32, l.wpk, N, K, 0));
    }
    return HQT_OK;
}

// deferred LayerNorm: fold (gamma, beta) of the LayerNorm in front of `l` into a second packed copy of its weights
static int fold_ln(hqt_handle* h, Lin& l, const float* gamma, const float* beta) {
    if (!l.wpk) return HQT_OK;
    CHK(dev_alloc(h, (void**)&l.wpk_ln, (size_t)l.N * l.K * 2, false));
    CHK(dev_alloc(h, (void**)&l.colsum, (size_t)l.N * 4, false));
    CHK(dev_alloc(h, (void**)&l.bias_ln, (size_t)l.N * 4, false));
    HIPCHK(launch_fold_layernorm(l.w32, gamma, beta, l.b32, h->fold_tmp, l.wpk_ln, l.colsum, l.bias_ln, l.N, l.K, 0));
    return HQT_OK;
}

static int load_block(hqt_handle* h, const std::string& p, BlockW& b) {
    const int64_t D = h->cfg.embed_dim;
    CHK(get_w(h, p + ".ln1.weight", {D}, &b.ln1_g));
    CHK(get_w(h, p + ".ln1.bias", {D}, &b.ln1_b));
    CHK(get_w(h, p + ".ln2.weight", {D}, &b.ln2_g));
    CHK(get_w(h, p + ".ln2.bias", {D}, &b.ln2_b));
    const float *wq, *wk, *wv, *bq, *bk, *bv, *w, *bias;
    CHK(get_w(h, p + ".attn.query.weight", {D, D}, &wq));
    CHK(get_w(h, p + ".attn.key.weight", {D, D}, &wk));
    CHK(get_w(h, p + ".attn.value.weight", {D, D}, &wv));
    CHK(get_w(h, p + ".attn.query.bias", {D}, &bq));
    CHK(get_w(h, p + ".attn.key.bias", {D}, &bk));
    CHK(get_w(h, p + ".attn.value.bias", {D}, &bv));
    float *wqkv, *bqkv;                       // fused [query; key; value] rows
    CHK(dev_alloc(h, (void**)&wqkv, (size_t)3 * D * D * 4, false));
    CHK(dev_alloc(h, (void**)&bqkv, (size_t)3 * D * 4, false));
    HIPCHK(hipMemcpy(wqkv, wq, D * D * 4, hipMemcpyDeviceToDevice));
    HIPCHK(hipMemcpy(wqkv + D * D, wk, D * D * 4, hipMemcpyDeviceToDevice));
    HIPCHK(hipMemcpy(wqkv + 2 * D * D, wv, D * D * 4, hipMemcpyDeviceToDevice));
    HIPCHK(hipMemcpy(bqkv, bq, D * 4, hipMemcpyDeviceToDevice));
    HIPCHK(hipMemcpy(bqkv + D, bk, D * 4, hipMemcpyDeviceToDevice));
    HIPCHK(hipMemcpy(bqkv + 2 * D, bv, D * 4, hipMemcpyDeviceToDevice));
    CHK(make_lin(h, b.qkv, wqkv, bqkv, 3 * D, D, true, true));      // + fp16 hi / lo planes: the SPLIT AR loop
    CHK(get_w(h, p + ".attn.proj.weight", {D, D}, &w));
    CHK(get_w(h, p + ".attn.proj.bias", {D}, &bias));
    CHK(make_lin(h, b.proj, w, bias, D, D, true, true));      // + fp16 hi / lo planes: the SPLIT AR loop
    CHK(get_w(h, p + ".mlp.0.weight", {4 * D, D}, &w));
    CHK(get_w(h, p + ".mlp.0.bias", {4 * D}, &bias));
    CHK(make_lin(h, b.fc1, w, bias, 4 * D, D, true, true));      // + fp16 hi / lo planes: the SPLIT AR loop
    CHK(get_w(h, p + ".mlp.2.weight", {D, 4 * D}, &w));
    CHK(get_w(h, p + ".mlp.2.bias", {D}, &bias));
    CHK(make_lin(h, b.fc2, w, bias, D, 4 * D, true, true));      // + fp16 hi / lo planes: the SPLIT AR loop
    CHK(fold_ln(h, b.qkv, b.ln1_g, b.ln1_b));
    CHK(fold_ln(h, b.fc1, b.ln2_g, b.ln2_b));
    return HQT_OK;
}

static int load_conv(hqt_handle* h, const std::string& name, int O, int I, int ksz, Lin& l, bool upsampling = false) {
    const float *w, *b;
    CHK(get_w(h, "stage1." + name + ".weight", {O, I, ksz, ksz}, &w));
    CHK(get_w(h, "stage1." + name + ".bias", {O}, &b));
    const int taps = ksz * ksz;
    float* wt = const_cast<float*>(w);
    if (taps > 1) {                          // [O][I][kh][kw] -> tap-major [O][kh*kw][I] (k = tap * I + i)
        CHK(dev_alloc(h, (void**)&wt, (size_t)O * I * taps * 4, false));
        HIPCHK(launch_repack_conv(w, wt, O, I, taps, 0));
    }
    CHK(make_lin(h, l, wt, b, O, I * taps, false, true));
    if (taps == 9 && I % 32 == 0) {              // fragment-packed copies for the ring kernels (whole 128-channel tiles; conv_out: one 16-channel block)
        if (O % 128 == 0 || O <= 16) {
            CHK(dev_alloc(h, (void**)&l.wfrag16, split_frag_elems(O, I) * sizeof(half_t), false));
            HIPCHK(launch_pack_split_frag16(wt, l.wfrag16, O, I, 0));
        }
        if (upsampling && O % 128 == 0 && I % 64 == 0) {      // nearest x2 + 3x3 = four 2x2 phase convs on the low-resolution image
            CHK(dev_alloc(h, (void**)&l.wup16, split_up_elems(O, I) * sizeof(half_t), false));
            HIPCHK(hipMemset(l.wup16, 0, split_up_elems(O, I) * sizeof(half_t)));
            HIPCHK(launch_pack_split_up16(wt, l.wup16, O, I, 0));
        }
    }
    return HQT_OK;
}

// the image enters conv_in as NHWC with its 3 channels zero-padded: 4 for the 4x4 stride-2 filter (K = 64), 16 for the 3x3 one (K = 144)
static int conv_in_cpad(const hqt_config& c) { return c.s1_use_init_downsample ? 4 : 16; }

// Encoder + quant_conv_b + the codebooks as distance-GEMM operands (hqt_encode).  All-or-nothing: called when
// stage1.encoder.conv_in.weight was set; any other missing encoder tensor is an error.
static int load_encoder(hqt_handle* h) {
    const hqt_config& c = h->cfg;
    for (auto& l : h->enc) {
        if (l.kind == 6) {
            const int ks = c.s1_use_init_downsample ? 4 : 3, taps = ks * ks, cp = conv_in_cpad(c);
            const float *w, *b;
            CHK(get_w(h, "stage1.encoder.conv_in.weight", {l.cout, 3, ks, ks}, &w));
            CHK(get_w(h, "stage1.encoder.conv_in.bias", {l.cout}, &b));
            float* wt;
            CHK(dev_alloc(h, (void**)&wt, (size_t)l.cout * cp * taps * 4, false));
            HIPCHK(launch_repack_conv(w, wt, l.cout, 3, taps, 0, cp));
            CHK(make_lin(h, l.conv1, wt, b, l.cout, cp * taps, false));
        } else if (l.kind == 5) {
            CHK(load_conv(h, l.name, l.cout, l.cin, 3, l.conv1));
        } else if (l.kind == 1) {
            CHK(get_w(h, "stage1." + l.name + ".norm1.weight", {l.cin}, &l.n1_g));
            CHK(get_w(h, "stage1." + l.name + ".norm1.bias", {l.cin}, &l.n1_b));
            CHK(get_w(h, "stage1." + l.name + ".norm2.weight", {l.cout}, &l.n2_g));
            CHK(get_w(h, "stage1." + l.name + ".norm2.bias", {l.cout}, &l.n2_b));
            CHK(load_conv(h, l.name + ".conv1", l.cout, l.cin, 3, l.conv1));
            CHK(load_conv(h, l.name + ".conv2", l.cout, l.cout, 3, l.conv2));
            if (l.cin != l.cout) CHK(load_conv(h, l.name + ".nin_shortcut", l.cout, l.cin, 1, l.nin));
        } else if (l.kind == 2) {
            CHK(get_w(h, "stage1." + l.name + ".norm.weight", {l.cin}, &l.n1_g));
            CHK(get_w(h, "stage1." + l.name + ".norm.bias", {l.cin}, &l.n1_b));
            CHK(load_conv(h, l.name + ".q", l.cin, l.cin, 1, l.q));
            CHK(load_conv(h, l.name + ".k", l.cin, l.cin, 1, l.k));
            CHK(load_conv(h, l.name + ".v", l.cin, l.cin, 1, l.v));
            CHK(load_conv(h, l.name + ".proj_out", l.cin, l.cin, 1, l.proj));
        } else {
            CHK(get_w(h, "stage1.encoder.norm_out.weight", {l.cin}, &l.n1_g));
            CHK(get_w(h, "stage1.encoder.norm_out.bias", {l.cin}, &l.n1_b));
            CHK(load_conv(h, "encoder.conv_out", l.cout, l.cin, 3, l.conv1));
        }
    }
    const int E = c.s1_embed_dim, L = c.code_levels == 3 ? 3 : 2;
    CHK(load_conv(h, "quant_conv_b", E, c.s1_z_channels, 1, h->quant_conv));
    for (int l = 0; l < L; ++l) {
        const int dim = E << (2 * (L - 1 - l));
        const std::string name = L == 3 ? "stage1.quantizers." + std::to_string(l) + ".embedding"
                                        : (l == 0 ? "stage1.quantize_t.embedding" : "stage1.quantize_b.embedding");
        const float* e = h->w[name].d;
        CHK(dev_alloc(h, (void**)&h->cb_norm[l], (size_t)c.s1_n_embed * 4, false));
        HIPCHK(launch_row_sumsq(e, DT_F32, h->cb_norm[l], c.s1_n_embed, dim, 0));
    }
    h->has_encoder = true;
    return alloc_encode_workspace(h);
}

static std::string key2(const hqt_handle* h, const char* name);

static int finalize_impl(hqt_handle* h);
static int persist_build(hqt_handle* h);
extern "C" int hqt_finalize_weights(hqt_handle* h) {
    if (!h) return fail(HQT_ERR_INVALID, "null handle");
    if (h->finalized) return HQT_OK;
    DeviceGuard dg(h->device);
    if (!dg.ok) return fail(HQT_ERR_HIP, "hipSetDevice(%d) failed", h->device);
    // all or nothing: on failure every derived buffer allocated by this attempt is released and the derived state cleared, so a
    // retry (after the missing tensor was set) starts clean instead of re-allocating on top of half-built layouts
    const size_t owned_before = h->owned.size();
    const int rc = finalize_impl(h);
    if (h->fold_tmp) { hipFree(h->fold_tmp); h->fold_tmp = nullptr; }
    if (rc != HQT_OK) {
        hipDeviceSynchronize();
        for (size_t i = owned_before; i < h->owned.size(); ++i) hipFree(h->owned[i]);
        h->owned.resize(owned_before);
        h->body.clear(); h->depth.clear();
        h->pbody = PersistProg(); h->pfull = PersistProg();
        h->head_top = h->head_bot = h->head_l2 = h->post_quant = h->quant_conv = Lin();
        for (auto& l : h->dec) { l.conv1 = l.conv2 = l.nin = l.q = l.k = l.v = l.proj = Lin(); }
        for (auto& l : h->enc) { l.conv1 = l.conv2 = l.nin = l.q = l.k = l.v = l.proj = Lin(); }
        for (auto& p : h->cb_norm) p = nullptr;
        h->vq_h = h->vq_recon = h->vq_zz = h->vq_err = nullptr; h->vq_z = nullptr; h->vq_best = nullptr;
        h->has_encoder = false;
    }
    return rc;
}
static int finalize_impl(hqt_handle* h) {
    const hqt_config& c = h->cfg;
    if (c.has_stage2) {
        HIPCHK(hipMalloc((void**)&h->fold_tmp, (size_t)std::max(4 * c.embed_dim, c.vocab_top) * c.embed_dim * 4));
        h->body.resize(c.n_layers);
        h->depth.resize(c.n_layers_depth);
        for (int i = 0; i < c.n_layers; ++i) { CHK(load_block(h, "stage2.blocks." + std::to_string(i), h->body[i])); h->body[i].body = true; }
        for (int i = 0; i < c.n_layers_depth; ++i) CHK(load_block(h, "stage2.depths." + std::to_string(i), h->depth[i]));
        const float* w;
        const int64_t D = c.embed_dim;
        const bool l3 = c.code_levels == 3;
        CHK(get_w(h, key2(h, "head_top.weight"), {c.vocab_top, D}, &w));
        CHK(make_lin(h, h->head_top, w, nullptr, c.vocab_top, D, true, true));      // + fp16 hi / lo planes: the SPLIT AR loop
        CHK(get_w(h, key2(h, "head_bot.weight"), {c.vocab_bot, D}, &w));
        CHK(make_lin(h, h->head_bot, w, nullptr, c.vocab_bot, D, true, true));      // + fp16 hi / lo planes: the SPLIT AR loop
        if (l3) {
            CHK(get_w(h, "stage2.head_levels.2.weight", {c.vocab_top, D}, &w));
            CHK(make_lin(h, h->head_l2, w, nullptr, c.vocab_top, D, true, true));      // + fp16 hi / lo planes: the SPLIT AR loop
        }
        // presence/shape checks of the remaining tensors happen here so that sample() cannot fail late
        const float* t;
        CHK(get_w(h, "stage2.ln_f.weight", {D}, &t)); CHK(get_w(h, "stage2.ln_f.bias", {D}, &t));
        CHK(get_w(h, key2(h, "ln_top.weight"), {D}, &t)); CHK(get_w(h, key2(h, "ln_top.bias"), {D}, &t));
        CHK(get_w(h, key2(h, "ln_bot.weight"), {D}, &t)); CHK(get_w(h, key2(h, "ln_bot.bias"), {D}, &t));
        CHK(get_w(h, "stage2.sos_depth", {1, 1, D}, &t));
        CHK(get_w(h, key2(h, "tok_emb_top.weight"), {c.vocab_top, D}, &t));
        CHK(get_w(h, key2(h, "tok_emb_bot.weight"), {c.vocab_bot, c.embedding_type == HQT_EMB_REDUCE ? D / 4 : D}, &t));
        if (c.embedding_type == HQT_EMB_TRANSFORMER1) CHK(get_w(h, "stage2.pos_emb_emb.weight", {l3 ? 21 : 5, D}, &t));
        CHK(get_w(h, "stage2.pos_emb_top.weight", {c.ctx_len_img, D}, &t));
        const int dmul = (l3 && c.depth_decoding == HQT_DEPTH_PARALLEL_REDUCE) ? 4 : 1;      // 'reduce': [V, 4 D] depth tables (hqtransformer.py:108-116)
        const bool causal_head = l3 && c.depth_decoding == HQT_DEPTH_TOP2MID2BOT;            // its sub-step inputs come from tok_emb_levels (hqtransformer.py:718-725)
        if (!causal_head) CHK(get_w(h, key2(h, "tok_emb_top_depth.weight"), {c.vocab_top, dmul * D}, &t));
        CHK(get_w(h, key2(h, "pos_emb_depth.weight"), {causal_head ? 21 : (l3 ? 4 : 5), D}, &t));
        if (l3) {
            CHK(get_w(h, "stage2.tok_emb_levels.2.weight", {c.vocab_top, D}, &t));
            if (!causal_head) {
                CHK(get_w(h, "stage2.tok_emb_depth_levels.1.weight", {c.vocab_top, dmul * D}, &t));
                CHK(get_w(h, "stage2.pos_emb_depths.1.weight", {16, D}, &t));
            }
            CHK(get_w(h, "stage2.ln_levels.2.weight", {D}, &t)); CHK(get_w(h, "stage2.ln_levels.2.bias", {D}, &t));
        }
        {
            const float *g1, *b1;
            CHK(get_w(h, key2(h, "ln_top.weight"), {D}, &g1)); CHK(get_w(h, key2(h, "ln_top.bias"), {D}, &b1));
            CHK(fold_ln(h, h->head_top, g1, b1));
            CHK(get_w(h, key2(h, "ln_bot.weight"), {D}, &g1)); CHK(get_w(h, key2(h, "ln_bot.bias"), {D}, &b1));
            CHK(fold_ln(h, h->head_bot, g1, b1));
            if (l3) {
                CHK(get_w(h, "stage2.ln_levels.2.weight", {D}, &g1)); CHK(get_w(h, "stage2.ln_levels.2.bias", {D}, &b1));
                CHK(fold_ln(h, h->head_l2, g1, b1));
            }
            HIPCHK(hipDeviceSynchronize());          // fold_tmp is released by hqt_finalize_weights on every exit
        }
        CHK(persist_build(h));                       // weight streams of the persistent AR chain (FAST, up to 64 rows)
        if (c.cond_type == HQT_COND_CLASS) CHK(get_w(h, "stage2.sos.weight", {c.n_classes, D}, &t));
        else if (c.cond_type == HQT_COND_TEXT) {
            CHK(get_w(h, "stage2.tok_emb_txt.weight", {c.vocab_txt, D}, &t));
            CHK(get_w(h, "stage2.pos_emb_txt.weight", {c.ctx_len_txt, D}, &t));
        } else CHK(get_w(h, "stage2.sos", {1, 1, D}, &t));
    }
    if (c.has_stage1) {
        const int E = c.s1_embed_dim;
        const float* t;
        if (c.code_levels == 3) {                // HQVAEGenerator: additive pyramid, E channels into the 1x1 conv
            CHK(get_w(h, "stage1.quantizers.0.embedding", {c.s1_n_embed, 16 * E}, &t));
            CHK(get_w(h, "stage1.quantizers.1.embedding", {c.s1_n_embed, 4 * E}, &t));
            CHK(get_w(h, "stage1.quantizers.2.embedding", {c.s1_n_embed, E}, &t));
            CHK(load_conv(h, "post_quant_conv_b", c.s1_z_channels, E, 1, h->post_quant));
        } else {
            CHK(get_w(h, "stage1.quantize_t.embedding", {c.s1_n_embed, 4 * E}, &t));
            CHK(get_w(h, "stage1.quantize_b.embedding", {c.s1_n_embed, E}, &t));
            CHK(load_conv(h, "post_quant_conv_b", c.s1_z_channels, 2 * E, 1, h->post_quant));
        }
        for (auto& l : h->dec) {
            if (l.kind == 0 || l.kind == 3) CHK(load_conv(h, l.name, l.cout, l.cin, 3, l.conv1, l.kind == 3));
            else if (l.kind == 1) {
                CHK(get_w(h, "stage1." + l.name + ".norm1.weight", {l.cin}, &l.n1_g));
                CHK(get_w(h, "stage1." + l.name + ".norm1.bias", {l.cin}, &l.n1_b));
                CHK(get_w(h, "stage1." + l.name + ".norm2.weight", {l.cout}, &l.n2_g));
                CHK(get_w(h, "stage1." + l.name + ".norm2.bias", {l.cout}, &l.n2_b));
                CHK(load_conv(h, l.name + ".conv1", l.cout, l.cin, 3, l.conv1));
                CHK(load_conv(h, l.name + ".conv2", l.cout, l.cout, 3, l.conv2));
                if (l.cin != l.cout) CHK(load_conv(h, l.name + ".nin_shortcut", l.cout, l.cin, 1, l.nin));
            } else if (l.kind == 2) {
                CHK(get_w(h, "stage1." + l.name + ".norm.weight", {l.cin}, &l.n1_g));
                CHK(get_w(h, "stage1." + l.name + ".norm.bias", {l.cin}, &l.n1_b));
                CHK(load_conv(h, l.name + ".q", l.cin, l.cin, 1, l.q));
                CHK(load_conv(h, l.name + ".k", l.cin, l.cin, 1, l.k));
                CHK(load_conv(h, l.name + ".v", l.cin, l.cin, 1, l.v));
                CHK(load_conv(h, l.name + ".proj_out", l.cin, l.cin, 1, l.proj));
            } else {
                CHK(get_w(h, "stage1.decoder.norm_out.weight", {l.cin}, &l.n1_g));
                CHK(get_w(h, "stage1.decoder.norm_out.bias", {l.cin}, &l.n1_b));
                CHK(load_conv(h, "decoder.conv_out", l.cout, l.cin, 3, l.conv1));
            }
        }
    }
    if (c.has_stage1 && h->w.count("stage1.encoder.conv_in.weight")) CHK(load_encoder(h));
    HIPCHK(stream_gemm_configure());
    HIPCHK(tile_gemm_configure());
    HIPCHK(mfma_gemm_configure());
    HIPCHK(split_kernels_configure());
    HIPCHK(hipDeviceSynchronize());
    h->finalized = true;
    return HQT_OK;
}

// ------------------------------------------------------------------------------------------ GEMM dispatch
struct Mode {
    bool fast = false;
    bool split = false;     // stage 1: fp32 tensors, convolutions on the matrix cores with fp16 hi / lo operands (split_kernels.h)
    bool split_ar = false;  // stage 2 (hqt_sample / hqt_sample_l3): the EXACT launch sequence -- fp32 activations, LayerNorm, attention, sampler -- with
                            // every nn.Linear on the matrix cores: fp32 activation rows split into fp16 hi / lo while their tile is staged, the weights as
                            // fp16 hi / lo planes, three MFMAs per product term, fp32 accumulation (split_gemm_kernel<fp32 A>)
    int act_dt() const { return fast ? DT_BF16 : DT_F32; }
    size_t act_sz() const { return fast ? 2 : 4; }
};
static int mode_of(int precision, bool stage1, Mode* md) {
    if (precision == HQT_PRECISION_FAST) md->fast = true;
    else if (precision == HQT_PRECISION_SPLIT && stage1) md->split = true;
    else if (precision == HQT_PRECISION_SPLIT) md->split_ar = true;
    else if (precision != HQT_PRECISION_EXACT) return fail(HQT_ERR_INVALID, "unknown precision %d", precision);
    return HQT_OK;
}

// y = x W^T (+b)(act)(+resid): picks the MFMA kernels in FAST mode when the shape allows
static int run_linear(hqt_handle* h, const Mode& md, GemmArgs g, const Lin& l, int a_dt, int c_dt, hipStream_t st,
                      const char* tag, bool defer_residual = false) {
    g.N = l.N; g.K = l.K; g.ldb = l.K;
    g.bias = l.b32;
    g.zero_page = h->zero_page;
    g.tune = h->policy;
    if (g.alpha == 0.0f) g.alpha = 1.0f;
    if (g.lda == 0) g.lda = l.K;
    // tools/ar_pass_time.py --by-rows: one timing slot per (GEMM, row count) instead of per GEMM
    static const bool by_rows = getenv("HQT_TIMING_BY_ROWS") != nullptr;
    char slot_name[64];
    snprintf(slot_name, sizeof slot_name, by_rows ? "%s@%d" : "%s", tag, g.M);
    Timed t(h, slot_name, st);
    if (g.Bw_lo) {                              // SPLIT: the caller packed the operand planes (s1_operand) after checking the shape
        g.Bw = l.w16h; g.Bw_lo = l.w16l; g.Bw_frag16 = l.wfrag16; g.Bw_up16 = l.wup16;
        if (h->gn_ready.tensor == g.C) h->gn_ready.tensor = nullptr;
        if (g.conv_taps == 9) {
            if (g.store == STORE_ROWS && h->gn_tiles && conv_halo_stats_ok(g.N, 32)) {   // every such output is normalised next
                g.gn_part_out_d = reinterpret_cast<double*>(h->gn_tiles); g.gn_out_groups = 32;
                h->gn_ready.tensor = g.C; h->gn_ready.tiles = split_conv3_tiles_per_image(g); h->gn_ready.dbl = true;
            }
            HIPCHK(launch_split_conv3(g, st));
        } else {
            HIPCHK(launch_split_gemm(g, st));
        }
        return HQT_OK;
    }
    if (md.fast) {
        if (g.store == STORE_RESID) h->resid_nparts = g.N / 32;          // partial row statistics the producer leaves (streaming GEMM: one per 32 columns)
        // merged passes (512+ rows): the LDS-tiled MFMA kernels (tile_gemm.hip); same operands, same store modes
        if ((g.ln_parts ? l.wpk_ln : l.wpk) && h->tile_gemm && tile_gemm_ok(g, a_dt, c_dt)) {
            const TilePlan tp = tile_gemm_plan(g);
            if (tp.geom >= 0 && (size_t)tp.S * 32 * g.a_packed_mb * g.N <= h->splitk_elems) {
                if (g.ln_parts) g.bias = l.bias_ln;
                HIPCHK(launch_tile_gemm(g, g.ln_parts ? l.wpk_ln : l.wpk, a_dt, c_dt, tp, h->splitk, st));
                if (tp.S > 1) count_variant(h, "variant:tile_gemm_%dx%d_splitk%d:%s%s", tp.bm, tp.bn, tp.S, tag, by_rows ? slot_name + strlen(tag) : "");
                else count_variant(h, "variant:tile_gemm_%dx%d%s:%s%s", tp.bm, tp.bn, g.ln_parts ? "_dln" : "", tag, by_rows ? slot_name + strlen(tag) : "");
                if (tp.S > 1) {
                    char cn[64];
                    snprintf(cn, sizeof cn, by_rows ? "gemm_combine@%d" : "gemm_combine", g.M);
                    t.next(cn);
                    HIPCHK(launch_resid_combine(g, h->splitk, tp.S, st));
                }
                if (g.store == STORE_RESID) h->resid_nparts = tp.S > 1 ? 1 : g.N / tp.bn;
                return HQT_OK;
            }
        }
        if (g.ln_parts) {                       // deferred LayerNorm: gamma-folded weights, folded bias; stream kernel only
            if (!l.wpk_ln || !stream_gemm_ok(g, a_dt, c_dt)) return fail(HQT_ERR_STATE, "deferred-LayerNorm GEMM without folded weights (%s)", tag);
            g.bias = l.bias_ln;
            HIPCHK(launch_stream_gemm(g, l.wpk_ln, a_dt, c_dt, 1, nullptr, st));
            count_variant(h, "variant:stream_gemm_dln:%s", tag);
            return HQT_OK;
        }
        if (l.wpk && !g.conv_taps && stream_gemm_ok(g, a_dt, c_dt)) {
            int S = (defer_residual && g.store != STORE_RESID) ? stream_gemm_splitk(g) : 1;
            if ((size_t)S * 32 * g.a_packed_mb * g.N > h->splitk_elems) S = 1;
            HIPCHK(launch_stream_gemm(g, l.wpk, a_dt, c_dt, S, h->splitk, st));
            count_variant(h, "variant:stream_gemm:%s", tag);
            if (S > 1) { h->pend.slabs = h->splitk; h->pend.S = S; h->pend.rows = 32 * g.a_packed_mb; h->pend.bias = l.b32; }
            return HQT_OK;
        }
        g.Bw = l.w16;
        if (mfma_gemm_ok(g, a_dt, DT_BF16, c_dt)) {
            // a 3x3 halo conv also leaves per-tile GroupNorm statistics of its output (every such output is normalised next)
            if (g.conv_taps == 9 && g.store == STORE_ROWS && h->gn_tiles && conv_halo_ok(g, c_dt) && conv_halo_stats_ok(g.N, 32) &&
                !getenv("HQT_NO_FUSED_GN")) {
                g.gn_part_out = h->gn_tiles; g.gn_out_groups = 32;
                h->gn_ready.tensor = g.C; h->gn_ready.tiles = conv_halo_tiles_per_image(g); h->gn_ready.dbl = false;
            } else if (h->gn_ready.tensor == g.C) {
                h->gn_ready.tensor = nullptr;                           // the tensor is being overwritten by something else
            }
            HIPCHK(launch_mfma_gemm(g, a_dt, DT_BF16, c_dt, st));
            return HQT_OK;
        }
        if (h->gn_ready.tensor == g.C) h->gn_ready.tensor = nullptr;
        HIPCHK(launch_gemm_generic(g, a_dt, DT_BF16, c_dt, st));
        return HQT_OK;
    }
    // (up to 256 rows the fp32 matrix instructions of exact_gemm.hip are faster than three fp16 MFMAs on 128 x 128 tiles that are mostly padding:
    //  249 vs 483 ms of AR loop per batch-64 step -- SPLIT takes them there, and is bit-identical to EXACT on those launches)
    if (md.split_ar && l.w16h && l.w16l && !g.conv_taps && !g.a_packed_mb && g.M > 256) {
        GemmArgs sg = g;
        sg.a_f32 = 1; sg.Bw = l.w16h; sg.Bw_lo = l.w16l; sg.range_flag = h->range_flag;
        if (split_gemm_ok(sg)) {
            // few tiles (fc2 / proj at 640 rows: 60): K slices into the split-K workspace, summed in index order by the combine launch
            const int S = h->split_kslices ? split_gemm_slices(sg) : 1;
            if (S > 1 && (size_t)S * sg.M * sg.N <= h->splitk_elems) { sg.k_slices = S; sg.k_slabs = h->splitk; }
            HIPCHK(launch_split_gemm(sg, st));
            if (sg.k_slices > 1) count_variant(h, "variant:split_gemm_kslices%d:%s", sg.k_slices, tag);
            else count_variant(h, "variant:split_gemm:%s", tag);
            return HQT_OK;
        }
    }
    g.Bw = l.w32;
    g.k_quarters = strncmp(tag, "gemm_", 5) == 0;          // the AR loop's nn.Linear launches (gemm_qkv / proj / fc1 / fc2 / head): see GemmArgs.k_quarters
    if (h->gn_ready.tensor == g.C) h->gn_ready.tensor = nullptr;
    if (exact_mfma_ok(g)) {                      // plain fp32 nn.Linear (the AR loop): the fp32 matrix instructions, one tile per wave
        if (l.w32t && g.N % 16 == 0) { g.Bw = l.w32t; g.b_tile16 = 1; }
        HIPCHK(launch_exact_mfma_gemm(g, st));
        count_variant(h, "variant:exact_mfma:%s", tag);
        return HQT_OK;
    }
    HIPCHK(launch_gemm_generic(g, DT_F32, DT_F32, DT_F32, st));
    if (md.split_ar) count_variant(h, "variant:gemm_generic_f32:%s", tag);
    return HQT_OK;
}

// ------------------------------------------------------------------------------------------ stage 2
struct SampleCtx {
    int B;
    const int64_t* cond;
    hqt_sample_opts o;
    const float* noise;
    const int64_t *feed_top, *feed_bot;
    float* logits_out;
    int64_t *out_top, *out_bot;
    hipStream_t st;
    Mode md;
    // three-level calls (hqt_sample_l3): per-level sampler settings and the third level's code buffers
    int levels = 2;
    int top_k[3] = {0, 0, 0};
    float top_p[3] = {0.f, 0.f, 0.f}, temperature[3] = {1.f, 1.f, 1.f};
    const int64_t* feed_l2 = nullptr;
    int64_t* out_l2 = nullptr;
};

// State-dict key of a stage-2 tensor: the code below names tensors as iHQGPT does; the three-level HQTransformer keeps
// the same roles under indexed names (hqtransformer.py:24-205).
static std::string key2(const hqt_handle* h, const char* name) {
    std::string n(name);
    if (h->cfg.code_levels == 3) {
        static const char* const tr[][2] = {
            {"tok_emb_top.weight", "tok_emb_levels.0.weight"}, {"tok_emb_bot.weight", "tok_emb_levels.1.weight"},
            {"tok_emb_top_depth.weight", "tok_emb_depth_levels.0.weight"}, {"pos_emb_depth.weight", "pos_emb_depths.0.weight"},
            {"ln_top.weight", "ln_levels.0.weight"}, {"ln_top.bias", "ln_levels.0.bias"},
            {"ln_bot.weight", "ln_levels.1.weight"}, {"ln_bot.bias", "ln_levels.1.bias"},
            {"head_top.weight", "head_levels.0.weight"}, {"head_bot.weight", "head_levels.1.weight"}};
        for (auto& t : tr) if (n == t[0]) { n = t[1]; break; }
    }
    return "stage2." + n;
}
static const float* W(hqt_handle* h, const char* name) { return h->w[key2(h, name)].d; }


// ------------------------------------------------------------------------------------------ persistent AR chain (persist.h)
// FAST precision, up to 64 rows, decode steps (one token per sample), root handle under the latency policy: the twelve body blocks of a top
// position, ln_f + sos_depth, depth sub-step 0 (single-key blocks) and head_top run as ONE launch (three code levels: the body).  Everything else -- merged passes, lanes,
// EXACT / SPLIT, the text prefill, depth sub-step 1 (256 rows: MI355X_MICROARCH.md's verdict for 256-row blocks is "cut at every seam") --
// keeps the launch chain.  HQT_PERSIST=0 switches it off (A/B).
struct PackSrc { const float* w; const float* gamma; };

static void persist_block_shapes(hqt_handle* h, const BlockW& bw, bool single_key, int cache_T, int* k4, std::vector<PersistPhase>& out, std::vector<PackSrc>& src) {
    const int D = h->cfg.embed_dim;
    PersistPhase ph{};
    if (single_key) {        // depth sub-step 0: one query over one key -- the attention output IS the value row (run_block_dln); only [key; value] is computed
        ph.type = PP_KV1; ph.N = 2 * D; ph.K = D; ph.dln = 1; ph.cache_T = cache_T; ph.kv_row = 0;
        out.push_back(ph); src.push_back({bw.qkv.w32 + (size_t)D * D, bw.ln1_g});
    } else {
        ph.type = PP_QKV; ph.N = 3 * D; ph.K = D; ph.dln = 1; ph.cache_T = cache_T;
        out.push_back(ph); src.push_back({bw.qkv.w32, bw.ln1_g});
        ph = PersistPhase{}; ph.type = PP_ATTN; ph.cache_T = cache_T;
        out.push_back(ph); src.push_back({nullptr, nullptr});
    }
    ph = PersistPhase{}; ph.type = PP_RESID; ph.map = PP_MAP_QUAD; ph.N = D; ph.K = D;
    out.push_back(ph); src.push_back({bw.proj.w32, nullptr});
    ph = PersistPhase{}; ph.type = PP_GELU; ph.N = 4 * D; ph.K = D; ph.dln = 1; ph.act = h->cfg.gelu_approx ? ACT_GELU_SIGMOID : ACT_GELU_ERF;
    out.push_back(ph); src.push_back({bw.fc1.w32, bw.ln2_g});
    ph = PersistPhase{}; ph.type = PP_RESID_K4; ph.map = PP_MAP_K4; ph.N = D; ph.K = 4 * D; ph.k4_epoch = ++*k4;
    out.push_back(ph); src.push_back({bw.fc2.w32, nullptr});
}

static int persist_build_one(hqt_handle* h, PersistProg& pr, const std::vector<PackSrc>& src) {
    const hqt_config& c = h->cfg;
    pr.ok = false;
    if (!persist_program_ok(pr.phases, c.embed_dim, 64, c.n_heads, h->ncu)) return HQT_OK;
    std::vector<unsigned long long> cu_off, tile_off;
    const size_t bytes = persist_layout(pr.phases, h->ncu, cu_off, tile_off);
    CHK(dev_alloc(h, (void**)&pr.stream, bytes + 1024, false));
    CHK(dev_alloc(h, (void**)&pr.d_cu_off, cu_off.size() * 8, false));
    HIPCHK(hipMemcpy(pr.d_cu_off, cu_off.data(), cu_off.size() * 8, hipMemcpyHostToDevice));
    unsigned long long* d_tile = nullptr;
    HIPCHK(hipMalloc((void**)&d_tile, tile_off.size() * 8));
    hipError_t e = hipMemcpy(d_tile, tile_off.data(), tile_off.size() * 8, hipMemcpyHostToDevice);
    for (size_t p = 0; p < pr.phases.size() && e == hipSuccess; ++p)
        if (src[p].w) e = launch_persist_pack(src[p].w, src[p].gamma, pr.phases[p], h->ncu, pr.stream, d_tile + p * h->ncu, 0);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    hipFree(d_tile);
    HIPCHK(e);
    pr.ok = true;
    return HQT_OK;
}

// finalize (root): the weight streams of both programs
static int persist_build(hqt_handle* h) {
    const hqt_config& c = h->cfg;
    h->pbody = PersistProg(); h->pfull = PersistProg();
    if (!c.has_stage2 || !(h->layouts & HQT_LAYOUT_FAST) || getenv("HQT_PERSIST_BUILD_OFF")) return HQT_OK;
    HIPCHK(hipDeviceGetAttribute(&h->ncu, hipDeviceAttributeMultiprocessorCount, h->device));
    // one 576-thread workgroup with its whole LDS ring must fit a compute unit, or the grid barrier can never complete: no program then (launch chain)
    HIPCHK(persist_configure());
    if (persist_blocks_per_cu() < 1) return HQT_OK;
    for (auto& b : h->body) if (!b.qkv.bias_ln || !b.fc1.bias_ln) return HQT_OK;        // no deferred-LayerNorm layouts for these shapes: launch chain
    for (auto& b : h->depth) if (!b.qkv.bias_ln || !b.fc1.bias_ln) return HQT_OK;
    std::vector<PackSrc> src;
    int k4 = 0;
    if (c.code_levels != 3 && h->head_top.bias_ln) {
        // two levels: body -> ln_f + sos_depth (a phase on the residual stream itself) -> the four single-key blocks of depth sub-step 0 -> head_top
        PersistProg& pr = h->pfull;
        for (auto& b : h->body) persist_block_shapes(h, b, false, h->Tmax, &k4, pr.phases, src);
        PersistPhase ph{};
        ph.type = PP_LNF; ph.N = c.embed_dim; ph.K = c.embed_dim; ph.map = PP_MAP_QUAD; ph.dln = 1;
        pr.phases.push_back(ph); src.push_back({nullptr, nullptr});
        for (auto& b : h->depth) persist_block_shapes(h, b, true, 5, &k4, pr.phases, src);
        ph = PersistPhase{};
        ph.type = PP_ROWS; ph.N = c.vocab_top; ph.K = c.embed_dim; ph.dln = 1;
        pr.phases.push_back(ph); src.push_back({h->head_top.w32, h->w[key2(h, "ln_top.weight")].d});
        CHK(persist_build_one(h, pr, src));
        if (pr.ok) {
            const int D = c.embed_dim;
            std::vector<float> beta(D), sos(D);
            HIPCHK(hipMemcpy(beta.data(), h->w[key2(h, "ln_f.bias")].d, (size_t)D * 4, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(sos.data(), h->w[key2(h, "sos_depth")].d, (size_t)D * 4, hipMemcpyDeviceToHost));
            for (int i = 0; i < D; ++i) beta[i] += sos[i];
            CHK(dev_alloc(h, (void**)&h->lnf_shift, (size_t)D * 4, false));
            HIPCHK(hipMemcpy(h->lnf_shift, beta.data(), (size_t)D * 4, hipMemcpyHostToDevice));
        }
    }
    if (!h->pfull.ok) {                          // three code levels (or no folded head): the body alone
        src.clear(); k4 = 0;
        for (auto& b : h->body) persist_block_shapes(h, b, false, h->Tmax, &k4, h->pbody.phases, src);
        CHK(persist_build_one(h, h->pbody, src));
    }
    HIPCHK(persist_configure());
    return HQT_OK;
}

// per handle (a clone binds its own workspace): the phase tables with this handle's buffers.  Outside stream capture.
static int persist_bind(hqt_handle* h) {
    const hqt_config& c = h->cfg;
    if (h->parent) return HQT_OK;                // lanes never launch it (persist_on): no phase tables, no uploads
    const size_t D = c.embed_dim;
    auto upload = [&](PersistProg& pr) -> int {
        CHK(dev_alloc(h, (void**)&pr.d_phases, pr.phases.size() * sizeof(PersistPhase), true));
        HIPCHK(hipMemcpy(pr.d_phases, pr.phases.data(), pr.phases.size() * sizeof(PersistPhase), hipMemcpyHostToDevice));
        return HQT_OK;
    };
    auto bind_body = [&](PersistProg& pr) -> size_t {
        const size_t kv_layer = (size_t)c.max_batch * h->Tmax * D;              // bf16 elements
        size_t p = 0;
        for (int l = 0; l < c.n_layers; ++l) {
            const BlockW& bw = h->body[l];
            bf16_t* kc = reinterpret_cast<bf16_t*>(h->kcache) + l * kv_layer;
            bf16_t* vc = reinterpret_cast<bf16_t*>(h->vcache) + l * kv_layer;
            PersistPhase* ph = &pr.phases[p];
            ph[0].A = h->xpk; ph[0].bias = bw.qkv.bias_ln; ph[0].colsum = bw.qkv.colsum; ph[0].out = h->qbuf; ph[0].kc = kc; ph[0].vc = vc;
            ph[1].A = reinterpret_cast<bf16_t*>(h->qbuf); ph[1].out = h->abuf; ph[1].kc = kc; ph[1].vc = vc;
            ph[2].A = reinterpret_cast<bf16_t*>(h->abuf); ph[2].bias = bw.proj.b32; ph[2].out = h->xpk;
            ph[3].A = h->xpk; ph[3].bias = bw.fc1.bias_ln; ph[3].colsum = bw.fc1.colsum; ph[3].out = h->mbuf;
            ph[4].A = reinterpret_cast<bf16_t*>(h->mbuf); ph[4].bias = bw.fc2.b32; ph[4].out = h->xpk;
            p += 5;
        }
        return p;
    };
    if (h->pbody.ok && !h->pbody.d_phases) {
        bind_body(h->pbody);
        CHK(upload(h->pbody));
    }
    if (h->pfull.ok && !h->pfull.d_phases) {
        size_t p = bind_body(h->pfull);
        PersistPhase& lf = h->pfull.phases[p++];
        lf.A = h->xpk; lf.bias = h->lnf_shift; lf.colsum = W(h, "ln_f.weight"); lf.out = h->xdpk;
        const size_t dkv_layer = (size_t)c.max_batch * 5 * D;
        for (int l = 0; l < c.n_layers_depth; ++l) {
            const BlockW& bw = h->depth[l];
            PersistPhase* ph = &h->pfull.phases[p];
            ph[0].A = h->xdpk; ph[0].bias = bw.qkv.bias_ln + D; ph[0].colsum = bw.qkv.colsum + D;
            ph[0].kc = reinterpret_cast<bf16_t*>(h->dk) + l * dkv_layer; ph[0].vc = reinterpret_cast<bf16_t*>(h->dv) + l * dkv_layer;
            ph[0].vpk = reinterpret_cast<bf16_t*>(h->abuf);
            ph[1].A = reinterpret_cast<bf16_t*>(h->abuf); ph[1].bias = bw.proj.b32; ph[1].out = h->xdpk;
            ph[2].A = h->xdpk; ph[2].bias = bw.fc1.bias_ln; ph[2].colsum = bw.fc1.colsum; ph[2].out = h->mbuf;
            ph[3].A = reinterpret_cast<bf16_t*>(h->mbuf); ph[3].bias = bw.fc2.b32; ph[3].out = h->xdpk;
            p += 4;
        }
        PersistPhase& hd = h->pfull.phases[p];
        hd.A = h->xdpk; hd.bias = h->head_top.bias_ln; hd.colsum = h->head_top.colsum; hd.out = h->logits;
        CHK(upload(h->pfull));
    }
    return HQT_OK;
}

static bool persist_on(hqt_handle* h, const SampleCtx& c, const PersistProg& pr) {
    return h->persist_enabled && !h->persist_tripped && pr.ok && pr.d_phases && c.md.fast && c.B <= 64 && h->policy == HQT_POLICY_LATENCY && !h->parent;
}

static int run_persist(hqt_handle* h, const SampleCtx& c, const PersistProg& pr, float* x32, int write_back, const int* t_dev, const char* slot) {
    Timed t(h, slot, c.st);
    PersistArgs a{};
    a.phases = pr.d_phases; a.n_phases = (int)pr.phases.size(); a.wstream = pr.stream; a.cu_off = pr.d_cu_off;
    a.counters = h->persist_counters; a.err = h->persist_err; a.x32 = x32; a.slabs = h->persist_slabs;
    a.D = h->cfg.embed_dim; a.M = c.B; a.MB = packed_mb(c.B); a.n_heads = h->cfg.n_heads; a.head_dim = h->cfg.embed_dim / h->cfg.n_heads;
    a.t_base = 0; a.t_base_dev = t_dev; a.write_back = write_back;
    static const bool nt = !(getenv("HQT_PERSIST_NT") && atoi(getenv("HQT_PERSIST_NT")) == 0);
    a.nt_weights = nt ? 1 : 0;
    persist_default_fill(a);
    HIPCHK(launch_persist(a, h->ncu, c.st));
    h->persist_used = true;
    h->capture_persist = true;
    count_variant(h, "variant:%s", slot);
    return HQT_OK;
}

static int run_ln(hqt_handle* h, hipStream_t st, float* x, const float* g, const float* b, const float* add, void* y, int M,
                  int D, int in_rpg, int in_off, int out_dt, int out_pk) {
    Timed t(h, "layernorm", st);
    LNArgs ln{x, g, b, add, y, M, D, in_rpg, in_off, 1e-5f, out_dt, out_pk, h->pend.slabs, h->pend.S, h->pend.rows, h->pend.bias, nullptr, 0, nullptr};
    h->pend.slabs = nullptr; h->pend.S = 0;
    HIPCHK(launch_layernorm(ln, st));
    return HQT_OK;
}

// One transformer block over M = B*Tq rows (stage2/layers.py:324-328,371-375)
static int run_block(hqt_handle* h, const SampleCtx& c, const BlockW& bw, float* x, int Tq, void* kc, void* vc, int Tcache,
                     int t_base, const int* t_base_dev, int causal) {
    const int D = h->cfg.embed_dim, M = c.B * Tq;
    const int adt = c.md.act_dt();
    // FAST: GEMM A operands travel in the MFMA-fragment-packed layout when the streaming GEMM serves them
    const int pk = (c.md.fast && M <= PACKED_MAX_ROWS && bw.qkv.wpk && bw.proj.wpk && bw.fc1.wpk && bw.fc2.wpk) ? packed_mb(M) : 0;
    CHK(run_ln(h, c.st, x, bw.ln1_g, bw.ln1_b, nullptr, h->hbuf, M, D, 1, 0, adt, pk));
    GemmArgs g{};
    g.A = h->hbuf; g.M = M; g.batch = 1; g.a_packed_mb = pk;
    g.C = h->qbuf; g.C2 = kc; g.C3 = vc; g.ldc = D; g.qkv_D = D; g.store = STORE_QKV;
    g.rows_per_group = Tq; g.group_stride = Tcache; g.row_offset = t_base; g.row_offset_dev = t_base_dev;
    CHK(run_linear(h, c.md, g, bw.qkv, adt, adt, c.st, "gemm_qkv"));
    {
        Timed t(h, "attention", c.st);
        AttnArgs a{h->qbuf, kc, vc, h->abuf, c.B, Tq, h->cfg.n_heads, D / h->cfg.n_heads, Tcache, t_base, t_base_dev, causal, adt, pk, nullptr};
        HIPCHK(launch_attention(a, c.st));
    }
    g = GemmArgs{};
    g.A = h->abuf; g.M = M; g.batch = 1; g.a_packed_mb = pk; g.C = x; g.ldc = D; g.resid = x; g.store = STORE_ROWS;
    CHK(run_linear(h, c.md, g, bw.proj, adt, DT_F32, c.st, "gemm_proj", true));
    CHK(run_ln(h, c.st, x, bw.ln2_g, bw.ln2_b, nullptr, h->hbuf, M, D, 1, 0, adt, pk));
    g = GemmArgs{};
    g.A = h->hbuf; g.M = M; g.batch = 1; g.a_packed_mb = pk; g.C = h->mbuf; g.ldc = 4 * D;
    g.store = pk ? STORE_PACKED : STORE_ROWS; g.c_packed_mb = pk;
    g.act = h->cfg.gelu_approx ? ACT_GELU_SIGMOID : ACT_GELU_ERF;
    CHK(run_linear(h, c.md, g, bw.fc1, adt, adt, c.st, "gemm_fc1"));
    g = GemmArgs{};
    g.A = h->mbuf; g.M = M; g.batch = 1; g.a_packed_mb = pk; g.C = x; g.ldc = D; g.resid = x; g.store = STORE_ROWS;
    CHK(run_linear(h, c.md, g, bw.fc2, adt, DT_F32, c.st, "gemm_fc2", true));
    return HQT_OK;
}

// FAST block with deferred LayerNorm (5 launches instead of 7): the residual stream is kept as an fp32 master row plus
// a bf16 packed copy with partial row statistics; qkv / fc1 consume the copy with gamma-folded weights and normalise in
// their epilogue, proj / fc2 update all three in theirs.  (Round 4's ninth "prefetch" wave, which touched the NEXT launch's weights
// from inside this one, bought nothing -- profiles/r04_micro_weight_prefetch.txt -- and left with its switch in round 5.)
static bool dln_ok(hqt_handle* h, const SampleCtx& c, const BlockW& bw, int M) {
    static const bool off = getenv("HQT_NO_DLN") != nullptr;      // debugging / A-B switch: classic LayerNorm kernels
    if (off) return false;
    return c.md.fast && M <= PACKED_MAX_ROWS && bw.qkv.wpk_ln && bw.fc1.wpk_ln && bw.proj.wpk && bw.fc2.wpk && h->cfg.embed_dim % 32 == 0;
}
static int run_block_dln(hqt_handle* h, const SampleCtx& c, const BlockW& bw, float* x32, bf16_t* xpk, float* parts, int* nparts,
                         int Tq, void* kc, void* vc, int Tcache, int t_base, const int* t_base_dev, int causal) {
    const int D = h->cfg.embed_dim, M = c.B * Tq, pk = packed_mb(M);
    GemmArgs g{};
    g.A = xpk; g.M = M; g.batch = 1; g.a_packed_mb = pk;
    g.ln_parts = parts; g.ln_nparts = *nparts; g.ln_colsum = bw.qkv.colsum; g.ln_eps = 1e-5f;
    g.C = h->qbuf; g.C2 = kc; g.C3 = vc; g.ldc = D; g.qkv_D = D; g.store = STORE_QKV;
    g.rows_per_group = Tq; g.group_stride = Tcache; g.row_offset = t_base; g.row_offset_dev = t_base_dev;
    // One query over one key (depth sub-step 0: hierarchical_ar.py:690-702 with an empty cache): softmax of a single
    // score is exactly 1, so the attention output is the value row itself.  The GEMM then skips the query third of
    // the fused weight (rows [D, 3D) only), appends K/V to the cache for sub-step 1 and writes V straight into the
    // projection's operand; no attention launch.
    const bool single_key = Tq == 1 && t_base == 0 && !t_base_dev && h->single_key;
    if (single_key) {
        g.qkv_first = 1; g.qkv_v_pk = reinterpret_cast<bf16_t*>(h->abuf); g.c_packed_mb = pk;
        Lin kv = bw.qkv;
        kv.N = 2 * D;
        kv.wpk_ln = bw.qkv.wpk_ln + (size_t)D * bw.qkv.K;          // packed n-tiles are contiguous: skip D rows
        kv.bias_ln = bw.qkv.bias_ln + D;
        kv.colsum = bw.qkv.colsum + D;
        g.ln_colsum = kv.colsum;
        CHK(run_linear(h, c.md, g, kv, DT_BF16, DT_BF16, c.st, "gemm_qkv"));
    } else {
        CHK(run_linear(h, c.md, g, bw.qkv, DT_BF16, DT_BF16, c.st, "gemm_qkv"));
        Timed t(h, "attention", c.st);
        AttnArgs a{h->qbuf, kc, vc, h->abuf, c.B, Tq, h->cfg.n_heads, D / h->cfg.n_heads, Tcache, t_base, t_base_dev, causal, DT_BF16, pk, nullptr};
        HIPCHK(launch_attention(a, c.st));
    }
    g = GemmArgs{};
    g.A = h->abuf; g.M = M; g.batch = 1; g.a_packed_mb = pk;
    g.C = x32; g.ldc = D; g.store = STORE_RESID; g.resid_pk = xpk; g.resid_parts = parts; g.c_packed_mb = pk;
    CHK(run_linear(h, c.md, g, bw.proj, DT_BF16, DT_F32, c.st, "gemm_proj"));
    *nparts = h->resid_nparts;
    g = GemmArgs{};
    g.A = xpk; g.M = M; g.batch = 1; g.a_packed_mb = pk;
    g.ln_parts = parts; g.ln_nparts = *nparts; g.ln_colsum = bw.fc1.colsum; g.ln_eps = 1e-5f;
    g.C = h->mbuf; g.ldc = 4 * D; g.store = STORE_PACKED; g.c_packed_mb = pk;
    g.act = h->cfg.gelu_approx ? ACT_GELU_SIGMOID : ACT_GELU_ERF;
    CHK(run_linear(h, c.md, g, bw.fc1, DT_BF16, DT_BF16, c.st, "gemm_fc1"));
    g = GemmArgs{};
    g.A = h->mbuf; g.M = M; g.batch = 1; g.a_packed_mb = pk;
    g.C = x32; g.ldc = D; g.store = STORE_RESID; g.resid_pk = xpk; g.resid_parts = parts; g.c_packed_mb = pk;
    CHK(run_linear(h, c.md, g, bw.fc2, DT_BF16, DT_F32, c.st, "gemm_fc2"));
    *nparts = h->resid_nparts;
    return HQT_OK;
}

// Everything of one top position after the body input x is ready (hierarchical_ar.py:482-563,667-789)
static int run_position(hqt_handle* h, const SampleCtx& c, int Tq_body, int body_t_base, bool body_tbase_from_state) {
    const hqt_config& cf = h->cfg;
    const int D = cf.embed_dim, B = c.B, V = cf.vocab_top;
    const int adt = c.md.act_dt();
    const size_t esz = c.md.act_sz();
    const size_t kv_layer = (size_t)cf.max_batch * h->Tmax * D * esz;
    const int* tb_dev = body_tbase_from_state ? &h->state->t_base : nullptr;
    // deferred LayerNorm needs the packed copy + row statistics of x from the caller's embedding kernel: embed_step emits
    // them (decode steps, Tq = 1); the text prefill's embed_text does not, so the prefill pass takes the classic path
    const bool dln_body = Tq_body == 1 && dln_ok(h, c, h->body[0], B * Tq_body);
    // the twelve body blocks as ONE persistent launch (persist.h): decode steps of up to 64 samples
    const bool dln1 = dln_ok(h, c, h->depth[0], B) && h->head_top.wpk_ln;
    const bool dln4 = dln_ok(h, c, h->depth[0], 4 * B) && h->head_bot.wpk_ln;
    const bool body_persistable = dln_body && body_tbase_from_state && body_t_base == 0;
    // ... up to the top logits: body, ln_f + sos_depth, depth sub-step 0, head_top as ONE persistent launch
    const bool pfull = body_persistable && dln1 && h->single_key && persist_on(h, c, h->pfull);
    const bool pbody = !pfull && body_persistable && persist_on(h, c, h->pbody);
    if (pfull) CHK(run_persist(h, c, h->pfull, h->x, 0, tb_dev, "persist_position"));
    if (pbody) CHK(run_persist(h, c, h->pbody, h->x, 1, tb_dev, "persist_body"));
    for (int l = 0; l < (pbody || pfull ? 0 : cf.n_layers); ++l) {
        void* kc = (char*)h->kcache + l * kv_layer;
        void* vc = (char*)h->vcache + l * kv_layer;
        if (dln_body) CHK(run_block_dln(h, c, h->body[l], h->x, h->xpk, h->parts, &h->nparts, Tq_body, kc, vc, h->Tmax, body_t_base, tb_dev, 1));
        else CHK(run_block(h, c, h->body[l], h->x, Tq_body, kc, vc, h->Tmax, body_t_base, tb_dev, 1));
    }
    // ln_f on the last token of each sample, + sos_depth (hierarchical_ar.py:561,684-686) -> depth-head input
    if (!pfull) {
        Timed t(h, "layernorm", c.st);
        LNArgs ln{h->x, W(h, "ln_f.weight"), W(h, "ln_f.bias"), W(h, "sos_depth"), h->xd, B, D, Tq_body, Tq_body - 1, 1e-5f, DT_F32, 0,
                  h->pend.slabs, h->pend.S, h->pend.rows, h->pend.bias, dln1 ? h->xdpk : nullptr, dln1 ? packed_mb(B) : 0, h->partsd};
        h->pend.slabs = nullptr; h->pend.S = 0;
        HIPCHK(launch_layernorm(ln, c.st));
        h->npartsd = 1;
    }
    const size_t dkv_layer = (size_t)cf.max_batch * 5 * D * esz;
    const int pk1 = (c.md.fast && B <= PACKED_MAX_ROWS && h->head_top.wpk) ? packed_mb(B) : 0;
    const int pk4 = (c.md.fast && 4 * B <= PACKED_MAX_ROWS && h->head_bot.wpk) ? packed_mb(4 * B) : 0;
    // ---- depth sub-step 0: top code
    for (int l = 0; l < (pfull ? 0 : cf.n_layers_depth); ++l) {
        void* kc = (char*)h->dk + l * dkv_layer;
        void* vc = (char*)h->dv + l * dkv_layer;
        if (dln1) CHK(run_block_dln(h, c, h->depth[l], h->xd, h->xdpk, h->partsd, &h->npartsd, 1, kc, vc, 5, 0, nullptr, 0));
        else CHK(run_block(h, c, h->depth[l], h->xd, 1, kc, vc, 5, 0, nullptr, 0));
    }
    GemmArgs g{};
    if (dln1) {                                  // ln_top folded into head_top
        g.A = h->xdpk; g.a_packed_mb = pk1; g.ln_parts = h->partsd; g.ln_nparts = h->npartsd; g.ln_colsum = h->head_top.colsum; g.ln_eps = 1e-5f;
    } else {
        CHK(run_ln(h, c.st, h->xd, W(h, "ln_top.weight"), W(h, "ln_top.bias"), nullptr, h->hbuf, B, D, 1, 0, adt, pk1));
        g.A = h->hbuf; g.a_packed_mb = pk1;
    }
    g.M = B; g.batch = 1; g.C = h->logits; g.ldc = V; g.store = STORE_ROWS;
    if (!pfull) CHK(run_linear(h, c.md, g, h->head_top, adt, DT_F32, c.st, "gemm_head"));
    {
        Timed t(h, "sampler", c.st);
        SamplerArgs s{h->logits, B, V, 1, B, c.o.temperature_top, c.o.top_k_top, c.o.top_p_top, c.noise, 0,
                      h->state, h->rows, c.o.n_steps, c.out_top, c.logits_out};
        s.fast_math = c.md.fast ? 1 : 0;
        // the draw and the embedding lookup of the drawn code in one kernel: the sampler's workgroup of sample b also writes the
        // four input rows of depth sub-step 1 (HQT_NO_FUSED_EMBED=1: separate depth_embed_kernel, for A/B runs)
        static const bool fuse = !getenv("HQT_NO_FUSED_EMBED");
        if (fuse) {
            s.emb_tok = W(h, "tok_emb_top_depth.weight"); s.emb_pos = W(h, "pos_emb_depth.weight");
            s.emb_feed = c.feed_top != c.out_top ? c.feed_top : nullptr;
            s.emb_x = h->xd; s.emb_D = D;
            s.emb_xpk = dln4 ? h->xdpk : nullptr; s.emb_pk_mb = dln4 ? packed_mb(4 * B) : 0; s.emb_parts = h->partsd;
        }
        HIPCHK(launch_sampler(s, c.st));
        if (!fuse) {
            Timed t2(h, "embed", c.st);
            HIPCHK(launch_depth_embed(c.feed_top, c.o.n_steps, h->state, W(h, "tok_emb_top_depth.weight"), W(h, "pos_emb_depth.weight"),
                                      h->xd, B, D, dln4 ? h->xdpk : nullptr, dln4 ? packed_mb(4 * B) : 0, h->partsd, c.st, V));
        }
        h->npartsd = 1;
    }
    // ---- depth sub-step 1: four bottom codes in one pass
    for (int l = 0; l < cf.n_layers_depth; ++l) {
        void* kc = (char*)h->dk + l * dkv_layer;
        void* vc = (char*)h->dv + l * dkv_layer;
        if (dln4) CHK(run_block_dln(h, c, h->depth[l], h->xd, h->xdpk, h->partsd, &h->npartsd, 4, kc, vc, 5, 1, nullptr, 0));
        else CHK(run_block(h, c, h->depth[l], h->xd, 4, kc, vc, 5, 1, nullptr, 0));
    }
    g = GemmArgs{};
    if (dln4) {
        g.A = h->xdpk; g.a_packed_mb = pk4; g.ln_parts = h->partsd; g.ln_nparts = h->npartsd; g.ln_colsum = h->head_bot.colsum; g.ln_eps = 1e-5f;
    } else {
        CHK(run_ln(h, c.st, h->xd, W(h, "ln_bot.weight"), W(h, "ln_bot.bias"), nullptr, h->hbuf, 4 * B, D, 1, 0, adt, pk4));
        g.A = h->hbuf; g.a_packed_mb = pk4;
    }
    g.M = 4 * B; g.batch = 1; g.C = h->logits; g.ldc = V; g.store = STORE_ROWS;
    CHK(run_linear(h, c.md, g, h->head_bot, adt, DT_F32, c.st, "gemm_head"));
    {
        Timed t(h, "sampler", c.st);
        SamplerArgs s{h->logits, 4 * B, V, 4, B, c.o.temperature_bot, c.o.top_k_bot, c.o.top_p_bot, c.noise, 1,
                      h->state, h->rows, c.o.n_steps, c.out_bot, c.logits_out};
        s.fast_math = c.md.fast ? 1 : 0;
        HIPCHK(launch_sampler(s, c.st));
    }
    return HQT_OK;
}

// One top position of the three-level HQTransformer after the body input x is ready (hqtransformer.py:409-635): body
// blocks, ln_f, then three depth sub-steps over 1, 4 and 16 tokens.  Every sub-step's tokens see all earlier and current
// depth tokens (the 'parallel' mask of layers.py:154-178 restricted to the rows being evaluated is all-ones), so the
// attention kernel runs non-causally over t_base + Tq keys of a 21-row cache.
static int run_position_l3(hqt_handle* h, const SampleCtx& c, int Tq_body, int body_t_base, bool body_tbase_from_state) {
    const hqt_config& cf = h->cfg;
    const int D = cf.embed_dim, B = c.B, V = cf.vocab_top;
    const int adt = c.md.act_dt();
    const size_t esz = c.md.act_sz();
    const size_t kv_layer = (size_t)cf.max_batch * h->Tmax * D * esz;
    const int* tb_dev = body_tbase_from_state ? &h->state->t_base : nullptr;
    const bool dln_body = Tq_body == 1 && dln_ok(h, c, h->body[0], B * Tq_body);
    const bool pbody = dln_body && body_tbase_from_state && body_t_base == 0 && persist_on(h, c, h->pbody);      // the body is the two-level model's: one persistent launch
    if (pbody) CHK(run_persist(h, c, h->pbody, h->x, 1, tb_dev, "persist_body"));
    for (int l = 0; l < (pbody ? 0 : cf.n_layers); ++l) {
        void* kc = (char*)h->kcache + l * kv_layer;
        void* vc = (char*)h->vcache + l * kv_layer;
        if (dln_body) CHK(run_block_dln(h, c, h->body[l], h->x, h->xpk, h->parts, &h->nparts, Tq_body, kc, vc, h->Tmax, body_t_base, tb_dev, 1));
        else CHK(run_block(h, c, h->body[l], h->x, Tq_body, kc, vc, h->Tmax, body_t_base, tb_dev, 1));
    }
    const Lin* heads[3] = {&h->head_top, &h->head_bot, &h->head_l2};
    const char* ln_names[3][2] = {{"ln_top.weight", "ln_top.bias"}, {"ln_bot.weight", "ln_bot.bias"}, {"ln_levels.2.weight", "ln_levels.2.bias"}};
    const int dmul = cf.depth_decoding == HQT_DEPTH_PARALLEL_REDUCE ? 4 : 1;
    // 'top2mid2bot' (hqtransformer.py:700-800): 21 causal sub-steps of ONE token -- sub-step cnt >= 1 is fed the code drawn by cnt - 1
    // through tok_emb_levels[cnt == 1 ? 0 : (cnt < 5 ? 1 : 2)] (:718-723: the table follows the sub-step being computed, so the last middle
    // code is embedded with the level-2 table, as there) + pos_emb_depths.0[cnt - 1]; head of level (cnt == 0 ? 0 : cnt < 5 ? 1 : 2)
    const bool causal_head = cf.depth_decoding == HQT_DEPTH_TOP2MID2BOT;
    const int nsub = causal_head ? 21 : 3;
    int64_t* outs[3] = {c.out_top, c.out_bot, c.out_l2};
    const int64_t* feeds[3] = {c.feed_top, c.feed_bot, c.feed_l2};
    const size_t dkv_layer = (size_t)cf.max_batch * 21 * D * esz;
    for (int sub = 0; sub < nsub; ++sub) {
        const int lv = causal_head ? (sub == 0 ? 0 : (sub < 5 ? 1 : 2)) : sub;
        const int Tq = causal_head ? 1 : (lv == 0 ? 1 : (lv == 1 ? 4 : 16));
        const int tbase = causal_head ? sub : (lv == 0 ? 0 : (lv == 1 ? 1 : 5));
        const int draw0 = tbase;
        const int M = B * Tq;
        const bool dln = dln_ok(h, c, h->depth[0], M) && heads[lv]->wpk_ln;
        const int pk = (c.md.fast && M <= PACKED_MAX_ROWS && heads[lv]->wpk) ? packed_mb(M) : 0;
        if (sub == 0) {           // ln_f on the last token of each sample, + sos_depth -> depth input of level 0
            Timed t(h, "layernorm", c.st);
            LNArgs ln{h->x, W(h, "ln_f.weight"), W(h, "ln_f.bias"), W(h, "sos_depth"), h->xd, B, D, Tq_body, Tq_body - 1, 1e-5f, DT_F32, 0,
                      h->pend.slabs, h->pend.S, h->pend.rows, h->pend.bias, dln ? h->xdpk : nullptr, dln ? packed_mb(B) : 0, h->partsd};
            h->pend.slabs = nullptr; h->pend.S = 0;
            HIPCHK(launch_layernorm(ln, c.st));
        } else if (causal_head) { // the previous sub-step's code, embedded by the spatial tables, + its position in the 21-token sequence
            Timed t(h, "embed", c.st);
            const int prev = sub - 1, plv = prev == 0 ? 0 : (prev < 5 ? 1 : 2);
            const int tbl = sub == 1 ? 0 : (sub < 5 ? 1 : 2);
            const float* tok = tbl == 0 ? W(h, "tok_emb_top.weight") : (tbl == 1 ? W(h, "tok_emb_bot.weight") : h->w["stage2.tok_emb_levels.2.weight"].d);
            HIPCHK(launch_depth_embed_causal(feeds[plv], plv == 0 ? 1 : (plv == 1 ? 4 : 16), plv == 0 ? 0 : (plv == 1 ? prev - 1 : prev - 5), c.o.n_steps,
                                             h->state, tok, W(h, "pos_emb_depth.weight") + (size_t)prev * D, h->xd, B, D,
                                             dln ? h->xdpk : nullptr, dln ? packed_mb(M) : 0, h->partsd, c.st, V));
        } else if (lv == 1) {     // emb(top code) + positions 0..3
            Timed t(h, "embed", c.st);
            HIPCHK(launch_depth_embed(c.feed_top, c.o.n_steps, h->state, W(h, "tok_emb_top_depth.weight"), W(h, "pos_emb_depth.weight"),
                                      h->xd, B, D, dln ? h->xdpk : nullptr, dln ? packed_mb(M) : 0, h->partsd, c.st, V, dmul * D));
        } else {                  // parent's level-1 embedding + position i (+ emb(top code): 'add'), 16 tokens
            Timed t(h, "embed", c.st);
            HIPCHK(launch_depth_embed_l2(c.feed_top, c.feed_bot, c.o.n_steps, h->state,
                                         cf.depth_decoding == HQT_DEPTH_PARALLEL_ADD ? W(h, "tok_emb_top_depth.weight") : nullptr,
                                         h->w["stage2.tok_emb_depth_levels.1.weight"].d, h->w["stage2.pos_emb_depths.1.weight"].d,
                                         h->xd, B, D, dln ? h->xdpk : nullptr, dln ? packed_mb(M) : 0, h->partsd, c.st, V, dmul * D));
        }
        h->npartsd = 1;
        for (int l = 0; l < cf.n_layers_depth; ++l) {
            void* kc = (char*)h->dk + l * dkv_layer;
            void* vc = (char*)h->dv + l * dkv_layer;
            if (dln) CHK(run_block_dln(h, c, h->depth[l], h->xd, h->xdpk, h->partsd, &h->npartsd, Tq, kc, vc, 21, tbase, nullptr, 0));
            else CHK(run_block(h, c, h->depth[l], h->xd, Tq, kc, vc, 21, tbase, nullptr, 0));
        }
        GemmArgs g{};
        if (dln) {                // ln_levels[lv] folded into head_levels[lv]
            g.A = h->xdpk; g.a_packed_mb = pk; g.ln_parts = h->partsd; g.ln_nparts = h->npartsd; g.ln_colsum = heads[lv]->colsum; g.ln_eps = 1e-5f;
        } else {
            const float* lg = lv < 2 ? W(h, ln_names[lv][0]) : h->w["stage2.ln_levels.2.weight"].d;
            const float* lb = lv < 2 ? W(h, ln_names[lv][1]) : h->w["stage2.ln_levels.2.bias"].d;
            CHK(run_ln(h, c.st, h->xd, lg, lb, nullptr, h->hbuf, M, D, 1, 0, adt, pk));
            g.A = h->hbuf; g.a_packed_mb = pk;
        }
        g.M = M; g.batch = 1; g.C = h->logits; g.ldc = V; g.store = STORE_ROWS;
        CHK(run_linear(h, c.md, g, *heads[lv], adt, DT_F32, c.st, "gemm_head"));
        {
            Timed t(h, "sampler", c.st);
            SamplerArgs sa{h->logits, M, V, Tq, B, c.temperature[lv], c.top_k[lv], c.top_p[lv], c.noise, draw0,
                           h->state, h->rows, c.o.n_steps, outs[lv], c.logits_out, 21};
            if (causal_head && lv > 0) { sa.out_stride = lv == 1 ? 4 : 16; sa.out_slot = lv == 1 ? sub - 1 : sub - 5; }
            sa.fast_math = c.md.fast ? 1 : 0;
            HIPCHK(launch_sampler(sa, c.st));
        }
    }
    return HQT_OK;
}

// hqt_config.ar_layouts: a precision whose weight layout the handle was created without fails here, loudly (never a silent slower path)
static int layout_check(const hqt_handle* h, const Mode& md) {
    if (md.fast && !(h->layouts & HQT_LAYOUT_FAST)) return fail(HQT_ERR_STATE, "HQT_PRECISION_FAST on a handle created without HQT_LAYOUT_FAST (hqt_config.ar_layouts = %d)", h->cfg.ar_layouts);
    if (md.split_ar && !(h->layouts & HQT_LAYOUT_SPLIT)) return fail(HQT_ERR_STATE, "HQT_PRECISION_SPLIT on a handle created without HQT_LAYOUT_SPLIT (hqt_config.ar_layouts = %d)", h->cfg.ar_layouts);
    return HQT_OK;
}

static int run_decode_step(hqt_handle* h, const SampleCtx& c) {      // one KV-cached position, Tq = 1
    const hqt_config& cf = h->cfg;
    {
        Timed t(h, "embed", c.st);
        EmbedArgs e{c.B, cf.embed_dim, c.o.n_steps, cf.embedding_type, cf.cond_type, h->state, c.cond,
                    cf.cond_type == HQT_COND_CLASS ? W(h, "sos.weight") : (cf.cond_type == HQT_COND_NONE ? W(h, "sos") : nullptr),
                    W(h, "tok_emb_top.weight"), W(h, "tok_emb_bot.weight"), W(h, "pos_emb_top.weight"),
                    cf.embedding_type == HQT_EMB_TRANSFORMER1 ? W(h, "pos_emb_emb.weight") : nullptr,
                    c.feed_top, c.feed_bot, h->x, nullptr, 0, h->parts};
        if (dln_ok(h, c, h->body[0], c.B)) { e.xpk = h->xpk; e.pk_mb = packed_mb(c.B); h->nparts = 1; }
        if (c.levels == 3) { e.levels = 3; e.tok_l2 = h->w["stage2.tok_emb_levels.2.weight"].d; e.codes_l2 = c.feed_l2; }
        e.V = cf.vocab_top; e.n_classes = cf.n_classes;
        HIPCHK(launch_embed_step(e, c.st));
    }
    if (c.levels == 3) CHK(run_position_l3(h, c, 1, 0, true));
    else CHK(run_position(h, c, 1, 0, true));
    HIPCHK(launch_advance_step(h->state, 1, c.st));
    return HQT_OK;
}

static int sample_run(hqt_handle* h, const SampleCtx& c);

extern "C" int hqt_sample(hqt_handle* h, int B, const int64_t* cond, const hqt_sample_opts* opts, const float* noise,
                          const int64_t* force_top, const int64_t* force_bot, float* logits_out, int64_t* out_top,
                          int64_t* out_bot, void* stream) {
    if (!h || !opts || !out_top || !out_bot) return fail(HQT_ERR_INVALID, "null argument");
    if (!h->finalized) return fail(HQT_ERR_STATE, "hqt_finalize_weights has not run");
    const hqt_config& cf = h->cfg;
    if (!cf.has_stage2) return fail(HQT_ERR_STATE, "handle was created without stage 2");
    if (cf.code_levels == 3) return fail(HQT_ERR_STATE, "three-level model: use hqt_sample_l3");
    if (B < 1 || B > cf.max_batch) return fail(HQT_ERR_INVALID, "B=%d outside [1, max_batch=%d]", B, cf.max_batch);
    if (opts->n_steps < 1 || opts->n_steps > cf.max_steps) return fail(HQT_ERR_INVALID, "n_steps=%d outside [1, %d]", opts->n_steps, cf.max_steps);
    if (cf.cond_type != HQT_COND_NONE && !cond) return fail(HQT_ERR_INVALID, "cond is required for class/text conditioning");
    if (!(opts->temperature_top > 0.f) || !(opts->temperature_bot > 0.f)) return fail(HQT_ERR_INVALID, "temperatures must be > 0");
    if ((opts->top_p_top > 0.f || opts->top_p_bot > 0.f) && cf.vocab_top > 8192) return fail(HQT_ERR_INVALID, "top-p needs vocab <= 8192");
    ON_DEVICE(h);
    // The launch sequence reads cond and writes the drawn codes in buffers owned by the handle, and takes the Philox seed
    // and the global row offset from device memory: nothing that changes from call to call is baked into the captured
    // graph, so a steady stream of batches replays ONE graph (no re-capture, no exec destroyed under pending launches).
    SampleCtx c;
    c.B = B; c.cond = cond ? h->cond_buf : nullptr; c.o = *opts; c.noise = noise;
    c.feed_top = force_top ? force_top : h->codes_top;
    c.feed_bot = force_bot ? force_bot : h->codes_bot;
    c.logits_out = logits_out; c.out_top = h->codes_top; c.out_bot = h->codes_bot;
    c.st = (hipStream_t)stream;
    CHK(mode_of(opts->precision, false, &c.md));
    CHK(layout_check(h, c.md));
    if (cond) HIPCHK(hipMemcpyAsync(h->cond_buf, cond, (size_t)B * (cf.cond_type == HQT_COND_TEXT ? cf.ctx_len_txt : 1) * 8, hipMemcpyDefault, c.st));
    const int rc_run = sample_run(h, c);
    if (rc_run != HQT_OK) return rc_run;
    HIPCHK(hipMemcpyAsync(out_top, h->codes_top, (size_t)B * opts->n_steps * 8, hipMemcpyDeviceToDevice, c.st));
    HIPCHK(hipMemcpyAsync(out_bot, h->codes_bot, (size_t)B * opts->n_steps * 4 * 8, hipMemcpyDeviceToDevice, c.st));
    return HQT_OK;
}

extern "C" int hqt_sample_l3(hqt_handle* h, int B, const int64_t* cond, const hqt_sample_opts_l3* opts, const float* noise,
                             const int64_t* force0, const int64_t* force1, const int64_t* force2, float* logits_out,
                             int64_t* out0, int64_t* out1, int64_t* out2, void* stream) {
    if (!h || !opts || !out0 || !out1 || !out2) return fail(HQT_ERR_INVALID, "null argument");
    if (!h->finalized) return fail(HQT_ERR_STATE, "hqt_finalize_weights has not run");
    const hqt_config& cf = h->cfg;
    if (!cf.has_stage2 || cf.code_levels != 3) return fail(HQT_ERR_STATE, "handle does not hold a three-level stage 2");
    if (B < 1 || B > cf.max_batch) return fail(HQT_ERR_INVALID, "B=%d outside [1, max_batch=%d]", B, cf.max_batch);
    if (opts->n_steps < 1 || opts->n_steps > cf.max_steps) return fail(HQT_ERR_INVALID, "n_steps=%d outside [1, %d]", opts->n_steps, cf.max_steps);
    if (cf.cond_type != HQT_COND_NONE && !cond) return fail(HQT_ERR_INVALID, "cond is required for class/text conditioning");
    float pmax = 0.f;
    for (int i = 0; i < 3; ++i) {
        if (!(opts->temperature[i] > 0.f)) return fail(HQT_ERR_INVALID, "temperatures must be > 0");
        pmax = std::max(pmax, opts->top_p[i]);
    }
    if (pmax > 0.f && cf.vocab_top > 8192) return fail(HQT_ERR_INVALID, "top-p needs vocab <= 8192");
    ON_DEVICE(h);
    SampleCtx c;
    c.levels = 3;
    c.B = B; c.cond = cond ? h->cond_buf : nullptr; c.noise = noise;
    c.o = hqt_sample_opts{};                        // the shared loop reads n_steps / seed / offsets / graph flag from here
    c.o.precision = opts->precision; c.o.n_steps = opts->n_steps; c.o.seed = opts->seed; c.o.sample_offset = opts->sample_offset;
    c.o.use_graph = opts->use_graph; c.o.row_seeds = opts->row_seeds; c.o.row_offsets = opts->row_offsets;
    c.o.top_k_top = opts->top_k[0]; c.o.top_k_bot = opts->top_k[1]; c.o.top_p_top = opts->top_p[0]; c.o.top_p_bot = std::max(opts->top_p[1], opts->top_p[2]);
    c.o.temperature_top = opts->temperature[0]; c.o.temperature_bot = opts->temperature[1];
    for (int i = 0; i < 3; ++i) { c.top_k[i] = opts->top_k[i]; c.top_p[i] = opts->top_p[i]; c.temperature[i] = opts->temperature[i]; }
    c.feed_top = force0 ? force0 : h->codes_top;
    c.feed_bot = force1 ? force1 : h->codes_bot;
    c.feed_l2 = force2 ? force2 : h->codes_l2;
    c.logits_out = logits_out; c.out_top = h->codes_top; c.out_bot = h->codes_bot; c.out_l2 = h->codes_l2;
    c.st = (hipStream_t)stream;
    CHK(mode_of(opts->precision, false, &c.md));
    CHK(layout_check(h, c.md));
    if (cond) HIPCHK(hipMemcpyAsync(h->cond_buf, cond, (size_t)B * (cf.cond_type == HQT_COND_TEXT ? cf.ctx_len_txt : 1) * 8, hipMemcpyDefault, c.st));
    const int rc_run = sample_run(h, c);
    if (rc_run != HQT_OK) return rc_run;
    HIPCHK(hipMemcpyAsync(out0, h->codes_top, (size_t)B * opts->n_steps * 8, hipMemcpyDeviceToDevice, c.st));
    HIPCHK(hipMemcpyAsync(out1, h->codes_bot, (size_t)B * opts->n_steps * 4 * 8, hipMemcpyDeviceToDevice, c.st));
    HIPCHK(hipMemcpyAsync(out2, h->codes_l2, (size_t)B * opts->n_steps * 16 * 8, hipMemcpyDeviceToDevice, c.st));
    return HQT_OK;
}

// Persistent launches need every compute unit of the device: two of them in flight at once (two root handles sampling on two streams)
// would each hold part of the chip and spin until their time limit.  All persistent work of a process on one device is therefore
// ordered by an event chain: the stream about to receive persistent launches first waits for the event behind the previous such
// enqueue (of ANY handle), and records the next one behind its own.  The mutex is held across the enqueue (wait .. record): two host
// threads cannot both wait for the same predecessor.  Kernels of other streams that are not persistent (a decode on another lane)
// only delay a persistent launch -- they end by themselves.
struct PersistOrder {
    static std::mutex& mu() { static std::mutex m; return m; }
    static hipEvent_t& ev(int device) { static std::map<int, hipEvent_t> evs; return evs[device]; }
    std::unique_lock<std::mutex> lock;
    hipStream_t st;
    int device;
    bool active;
    PersistOrder(int device_, hipStream_t st_, bool active_) : st(st_), device(device_), active(active_) {
        if (!active) return;
        lock = std::unique_lock<std::mutex>(mu());
        hipEvent_t& e = ev(device);
        if (!e) { if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { e = nullptr; return; } }
        else hipStreamWaitEvent(st, e, 0);
    }
    ~PersistOrder() {
        if (!active) return;
        hipEvent_t e = ev(device);
        if (e) hipEventRecord(e, st);
    }
};

static int sample_run(hqt_handle* h, const SampleCtx& c) {
    const hqt_config& cf = h->cfg;
    const hqt_sample_opts* opts = &c.o;
    const int B = c.B;
    const int64_t* cond = c.cond;
    const float* noise = c.noise;
    float* logits_out = c.logits_out;
    HIPCHK(sampler_configure(cf.vocab_top, opts->top_p_top > 0.f || opts->top_p_bot > 0.f));
    CHK(persist_bind(h));
    HIPCHK(launch_set_step(h->state, 0, 0, c.st));
    if (opts->row_seeds || opts->row_offsets) {      // merged steps: per-row Philox keys (host arrays, staged through pinned-free pageable copies: B <= max_batch entries)
        if (!opts->row_seeds || !opts->row_offsets) return fail(HQT_ERR_INVALID, "row_seeds and row_offsets come together");
        const int slot = h->rows_next;
        h->rows_next = (slot + 1) % hqt_handle::ROWS_RING;
        if (!h->rows_pinned[slot]) {
            HIPCHK(hipHostMalloc((void**)&h->rows_pinned[slot], (size_t)h->cfg.max_batch * sizeof(RowKey), hipHostMallocDefault));
            HIPCHK(hipEventCreateWithFlags(&h->rows_ev[slot], hipEventDisableTiming));
        }
        if (h->rows_busy[slot]) HIPCHK(hipEventSynchronize(h->rows_ev[slot]));       // the copy that last read this buffer (4 calls ago) is done
        RowKey* rk = h->rows_pinned[slot];
        for (int b = 0; b < B; ++b) { rk[b].seed = opts->row_seeds[b]; rk[b].global_row = opts->row_offsets[b]; }
        HIPCHK(hipMemcpyAsync(h->rows, rk, (size_t)B * sizeof(RowKey), hipMemcpyHostToDevice, c.st));
        HIPCHK(hipEventRecord(h->rows_ev[slot], c.st));
        h->rows_busy[slot] = true;
    } else {
        HIPCHK(launch_set_rows(h->rows, B, opts->seed, opts->sample_offset, c.st));
    }
    int first = 0;
    if (cf.cond_type == HQT_COND_TEXT) {     // 64-token causal prefill (sampling.py:187-190, layers.py:107-111)
        const int T = cf.ctx_len_txt;
        HIPCHK(launch_embed_text(cond, W(h, "tok_emb_txt.weight"), W(h, "pos_emb_txt.weight"), h->x, B, T, cf.embed_dim, c.st, cf.vocab_txt));
        if (c.levels == 3) CHK(run_position_l3(h, c, T, 0, false));
        else CHK(run_position(h, c, T, 0, false));
        HIPCHK(launch_advance_step(h->state, T, c.st));
        first = 1;
    }
    const int remaining = opts->n_steps - first;
    if (remaining <= 0) return HQT_OK;
    if (opts->use_graph && !h->timing) {
        // positions per captured graph: the largest divisor of the remaining positions up to HQT_GRAPH_POSITIONS (default 16):
        // one graph launch then covers G positions (fewer host launches and graph-to-graph hand-overs on the device)
        static const int gmax = getenv("HQT_GRAPH_POSITIONS") ? std::max(1, atoi(getenv("HQT_GRAPH_POSITIONS"))) : 16;
        int G = 1;
        for (int d = std::min(gmax, remaining); d >= 1; --d) if (remaining % d == 0) { G = d; break; }
        std::vector<uint64_t> key = {(uint64_t)G, (uint64_t)B, (uint64_t)(cond != nullptr), (uint64_t)noise, (uint64_t)c.feed_top, (uint64_t)c.feed_bot,
                                     (uint64_t)logits_out, (uint64_t)opts->precision,
                                     (uint64_t)opts->n_steps, (uint64_t)opts->top_k_top, (uint64_t)opts->top_k_bot,
                                     (uint64_t)c.levels, (uint64_t)c.feed_l2, (uint64_t)c.top_k[2], (uint64_t)h->policy,
                                     (uint64_t)((h->persist_enabled && !h->persist_tripped ? 1 : 0) + (h->single_key ? 0 : 4) + (h->split_kslices ? 0 : 8))};
        { uint32_t f3[2]; memcpy(f3, &c.top_p[2], 4); memcpy(f3 + 1, &c.temperature[2], 4); key.push_back(f3[0]); key.push_back(f3[1]); }
        uint32_t f[4];
        memcpy(f, &opts->top_p_top, 4); memcpy(f + 1, &opts->top_p_bot, 4);
        memcpy(f + 2, &opts->temperature_top, 4); memcpy(f + 3, &opts->temperature_bot, 4);
        for (int i = 0; i < 4; ++i) key.push_back(f[i]);
        if (!h->graph_exec || key != h->graph_key) {
            if (h->graph_exec) {                         // rare (options or test-only buffers changed): drain before destroying
                HIPCHK(hipStreamSynchronize(c.st));
                hipGraphExecDestroy(h->graph_exec);
                h->graph_exec = nullptr;
            }
            hipStream_t cs;
            HIPCHK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
            SampleCtx cc = c;
            cc.st = cs;
            h->capture_persist = false;
            HIPCHK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
            int rc = HQT_OK;
            for (int gi = 0; gi < G && rc == HQT_OK; ++gi) rc = run_decode_step(h, cc);
            hipGraph_t graph = nullptr;
            hipError_t e = hipStreamEndCapture(cs, &graph);
            hipStreamDestroy(cs);
            if (rc != HQT_OK) { if (graph) hipGraphDestroy(graph); return rc; }
            HIPCHK(e);
            HIPCHK(hipGraphInstantiate(&h->graph_exec, graph, nullptr, nullptr, 0));
            hipGraphDestroy(graph);
            h->graph_key = key;
            h->graph_has_persist = h->capture_persist;
            h->persist_used = false;                     // capturing queued nothing
        }
        PersistOrder order(h->device, c.st, h->graph_has_persist);
        if (h->graph_has_persist) h->persist_used = true;   // every replay queues the persistent launches the graph holds (hqt_range_check reads their mark)
        for (int s = 0; s < remaining / G; ++s) HIPCHK(hipGraphLaunch(h->graph_exec, c.st));
        return HQT_OK;
    }
    const bool may_persist = c.md.fast && c.B <= 64 && (persist_on(h, c, h->pfull) || persist_on(h, c, h->pbody));
    PersistOrder order(h->device, c.st, may_persist);
    for (int s = 0; s < remaining; ++s) CHK(run_decode_step(h, c));
    return HQT_OK;
}

// ------------------------------------------------------------------------------------------ stage 1
static const float* W1(hqt_handle* h, const std::string& name) { return h->w["stage1." + name].d; }

static GemmArgs conv_args(const void* in, int nimg, int res_out, int cin, int taps, int upsample, void* out, int cout) {
    GemmArgs g{};
    g.A = in; g.conv_taps = taps; g.H = res_out; g.W = res_out; g.Cin = cin; g.upsample = upsample;
    g.M = nimg * res_out * res_out; g.batch = 1;
    g.C = out; g.ldc = cout; g.store = STORE_ROWS;
    return g;
}
static void with_gn(GemmArgs& g, const float* stats, const float* gamma, const float* beta, int swish) {
    g.gn_stats = stats; g.gn_gamma = gamma; g.gn_beta = beta; g.gn_groups = 32; g.gn_swish = swish;
}

// One chunk of images moving through the conv stack (decoder or encoder): the rotating activation buffers and the two
// GroupNorm statistics slots.
struct S1Ctx {
    hqt_handle* h;
    int n;
    Mode md;
    hipStream_t st;
    int adt;
    void *cur, *t1, *t2, *tn;
    float *gn1, *gn2;
    bool cur_planes = false;   // SPLIT: `cur` already holds fp16 hi / lo operand planes (the producing conv's epilogue emitted them: GemmArgs::out_split)
};

// SPLIT: can conv / GEMM `g` (A = an fp32 NHWC tensor) with filters `l` run on the matrix cores?  Shapes the split kernels do
// not take (strided taps, channel counts that are not multiples of 64, ...) fall back to the fp32 vector-ALU kernel.
static bool split_shape_ok(const hqt_handle* h, const GemmArgs& g, const Lin& l) {
    if (!l.w16h) return false;
    GemmArgs t = g;
    t.N = l.N; t.K = l.K; t.ldb = l.K; t.zero_page = h->zero_page; t.Bw_lo = l.w16l; t.Bw_frag16 = l.wfrag16; t.Bw_up16 = l.wup16;
    if (t.lda == 0) t.lda = l.K;
    return t.conv_taps == 9 ? split_conv3_ok(t) : split_gemm_ok(t);
}
// SPLIT operand pass: fp16 hi / lo planes of (GroupNorm + swish of) the fp32 tensor g->A into `tn`; the conv then reads `tn`.
static int s1_split_pack(S1Ctx& c, GemmArgs* g, const float* stats, const float* gamma, const float* beta, int swish) {
    hqt_handle* h = c.h;
    const int hw_in = g->conv_taps ? ((g->H << g->conv_stride2) >> g->upsample) * ((g->W << g->conv_stride2) >> g->upsample) : 0;
    const int C = g->conv_taps ? g->Cin : g->lda;
    const int rows = g->conv_taps ? c.n : 1, per = g->conv_taps ? hw_in : g->M;
    Timed t(h, "split_pack", c.st);
    HIPCHK(launch_split_pack(reinterpret_cast<const float*>(g->A), reinterpret_cast<half_t*>(c.tn), stats, gamma, beta, rows, per, C, 32, swish, h->range_flag, c.st));
    g->A = c.tn;
    g->Bw_lo = c.tn;                         // non-NULL marks the operand as split planes; run_linear substitutes the filter planes
    if (!g->conv_taps) g->lda = 2 * C;
    return HQT_OK;
}

// GroupNorm(+swish) in front of a conv.  EXACT: statistics pass, then the normalisation is applied inside
// the conv's operand loader.  FAST: statistics pass + one bandwidth-bound apply pass into `tn`, so the MFMA
// conv reads a plain bf16 tensor.  SPLIT: statistics (from the producing conv's epilogue when it left them), then the
// operand pass writes the normalised tensor as fp16 hi / lo planes into `tn`; shapes the split kernels do not take run
// the EXACT way.  Sets the tensor the conv must read / fills the loader's GN fields.
static int s1_norm(S1Ctx& c, const void* src, int C, int hw, float* stats, const float* gamma, const float* beta, int swish, GemmArgs* g,
                   const Lin* l = nullptr) {
    hqt_handle* h = c.h;
    hipStream_t st = c.st;
    if (c.md.fast) {
        if (h->gn_ready.tensor == src && !h->gn_ready.dbl) {     // the producing conv already reduced its tiles: only the fixed-order finalize is left
            Timed t(h, "gn_stats", st);
            HIPCHK(launch_gn_finalize_tiles(h->gn_tiles, stats, c.n, h->gn_ready.tiles, hw, C, 32, 1e-6f, st));
        } else {
            Timed t(h, "gn_stats", st);
            HIPCHK(launch_gn_stats_fast(src, stats, h->gn_partial, c.n, hw, C, 32, 1e-6f, st));
        }
        { Timed t(h, "gn_apply", st); HIPCHK(launch_gn_apply(src, c.tn, stats, gamma, beta, c.n, hw, C, 32, swish, st)); }
        g->A = c.tn;
    } else if (c.md.split && l && split_shape_ok(h, *g, *l)) {
        {
            Timed t(h, "gn_stats", st);
            if (h->gn_ready.tensor == src && h->gn_ready.dbl)
                HIPCHK(launch_gn_finalize_tiles_d(reinterpret_cast<const double*>(h->gn_tiles), stats, c.n, h->gn_ready.tiles, hw, C, 32, 1e-6f, st));
            else
                HIPCHK(launch_gn_stats_fast(src, stats, h->gn_partial, c.n, hw, C, 32, 1e-6f, st, DT_F32));
        }
        CHK(s1_split_pack(c, g, stats, gamma, beta, swish));
    } else {
        { Timed t(h, "gn_stats", st); HIPCHK(launch_gn_stats(src, c.adt, stats, c.n, hw, C, 32, 1e-6f, st)); }
        with_gn(*g, stats, gamma, beta, swish);
    }
    return HQT_OK;
}
// a conv without GroupNorm in front (conv_in, upsample conv, nin_shortcut, proj_out, the 1x1 convs around the quantiser): SPLIT
// needs the operand planes, the other modes read the tensor as it is
static int s1_plain(S1Ctx& c, GemmArgs* g, const Lin& l) {
    if (!c.md.split) return HQT_OK;
    if (g->conv_taps == 1) {                 // 1x1: the GEMM kernel splits the fp32 tensor while it stages the tile -- no operand pass
        GemmArgs t = *g;
        t.a_f32 = 1;
        if (split_shape_ok(c.h, t, l)) {
            g->a_f32 = 1; g->range_flag = c.h->range_flag;
            g->Bw_lo = g->A;                 // non-NULL marks a SPLIT launch; run_linear substitutes the filter planes
            return HQT_OK;
        }
    }
    if (split_shape_ok(c.h, *g, l)) CHK(s1_split_pack(c, g, nullptr, nullptr, nullptr, 0));
    return HQT_OK;
}

// SPLIT: a product of two fp32 ACTIVATION tensors (the attention block's q k^T and softmax v) on the matrix cores -- both operands are
// split while their tiles are staged (split_gemm_kernel<true, true>); shapes it does not take stay on the fp32 vector-ALU kernel
static bool s1_split_product(hqt_handle* h, const Mode& md, GemmArgs& sg) {
    if (!md.split) return false;
    GemmArgs t = sg;
    t.a_f32 = 1; t.b_f32 = 1; t.range_flag = h->range_flag;
    if (!split_gemm_ok(t)) return false;
    sg = t;
    return true;
}

// kinds 0 conv3, 1 ResnetBlock, 2 AttnBlock, 3 upsample conv, 5 Downsample conv (shared by Decoder.forward and Encoder.forward)
static int s1_layer(S1Ctx& c, const DecLayer& l, const DecLayer* next = nullptr) {
    hqt_handle* h = c.h;
    const Mode& md = c.md;
    hipStream_t st = c.st;
    const int adt = c.adt, n = c.n;
    void *&cur = c.cur, *&t1 = c.t1, *&t2 = c.t2;
    const int res = l.res, hw = res * res;
    if (l.kind == 0 || l.kind == 3) {
        const int ro = l.kind == 3 ? 2 * res : res;
        GemmArgs g = conv_args(cur, n, ro, l.cin, 9, l.kind == 3, t1, l.cout);
        if (c.cur_planes) { g.Bw_lo = g.A; c.cur_planes = false; }   // the planes are there (non-NULL Bw_lo marks a SPLIT launch; run_linear substitutes the filters)
        else CHK(s1_plain(c, &g, l.conv1));
        CHK(run_linear(h, md, g, l.conv1, adt, adt, st, "conv3x3"));
        std::swap(cur, t1);
    } else if (l.kind == 5) {               // Downsample (stage1/modules/layers.py:56-76): pad right / bottom by one, 3x3 stride 2
        GemmArgs g = conv_args(cur, n, res / 2, l.cin, 9, 0, t1, l.cout);
        g.conv_stride2 = 1; g.conv_nopad = 1;
        CHK(run_linear(h, md, g, l.conv1, adt, adt, st, "conv_down"));
        std::swap(cur, t1);
    } else if (l.kind == 1) {               // ResnetBlock (stage1/modules/layers.py:115-133)
        GemmArgs g = conv_args(cur, n, res, l.cin, 9, 0, t1, l.cout);
        CHK(s1_norm(c, cur, l.cin, hw, c.gn1, l.n1_g, l.n1_b, 1, &g, &l.conv1));
        CHK(run_linear(h, md, g, l.conv1, adt, adt, st, "conv3x3"));
        const void* shortcut = cur;
        void* outbuf = t2;
        if (l.cin != l.cout) {
            GemmArgs sc = conv_args(cur, n, res, l.cin, 1, 0, t2, l.cout);
            CHK(s1_plain(c, &sc, l.nin));
            CHK(run_linear(h, md, sc, l.nin, adt, adt, st, "conv1x1"));
            shortcut = t2;
            outbuf = cur;                   // x is dead once the shortcut is computed
        }
        g = conv_args(t1, n, res, l.cout, 9, 0, outbuf, l.cout);
        g.resid = shortcut;
        CHK(s1_norm(c, t1, l.cout, hw, c.gn2, l.n2_g, l.n2_b, 1, &g, &l.conv2));
        // SPLIT: when the block's only consumer is an upsampling conv (no GroupNorm in between) whose operand would be split by a separate pass, this conv's epilogue
        // writes the operand planes instead of the fp32 tensor (same bytes, one pass over the tensor less)
        bool planes = false;
        if (md.split && g.Bw_lo && next && next->kind == 3 && next->cin == l.cout) {
            GemmArgs up = conv_args(outbuf, n, 2 * res, next->cin, 9, 1, t1, next->cout);
            GemmArgs me = g;
            me.N = l.conv2.N; me.K = l.conv2.K; me.ldb = l.conv2.K; me.zero_page = h->zero_page; me.Bw = l.conv2.w16h; me.Bw_lo = l.conv2.w16l; me.Bw_frag16 = l.conv2.wfrag16;
            planes = split_shape_ok(h, up, next->conv1) && split_conv3_emits_planes(me);
        }
        if (planes) { g.out_split = 1; g.range_flag = h->range_flag; }
        CHK(run_linear(h, md, g, l.conv2, adt, adt, st, "conv3x3"));
        if (outbuf == t2) std::swap(cur, t2);
        c.cur_planes = planes;
        if (planes) count_variant(h, "variant:conv3x3_planes_out:conv3x3");
    } else if (l.kind == 2) {               // AttnBlock (stage1/modules/layers.py:163-186)
        const int C = l.cin;
        GemmArgs g = conv_args(cur, n, res, C, 1, 0, h->aq, C);
        CHK(s1_norm(c, cur, C, hw, c.gn1, l.n1_g, l.n1_b, 0, &g, &l.q));
        const GemmArgs normed = g;           // same normalised input for q, k, v
        CHK(run_linear(h, md, g, l.q, adt, adt, st, "conv1x1"));
        g = normed; g.C = h->ak;
        CHK(run_linear(h, md, g, l.k, adt, adt, st, "conv1x1"));
        g = normed; g.C = h->av;
        g.store = STORE_NCHW; g.rows_per_image = hw;                 // V^T per image: [C][hw]
        CHK(run_linear(h, md, g, l.v, adt, adt, st, "conv1x1"));
        {   // S[i, j] = q_i . k_j * C^-0.5
            Timed t(h, "attn_gemm", st);
            GemmArgs sg{};
            sg.A = h->aq; sg.lda = C; sg.a_batch_stride = (long long)hw * C;
            sg.Bw = h->ak; sg.ldb = C; sg.b_batch_stride = (long long)hw * C;
            sg.C = h->as; sg.ldc = hw; sg.c_batch_stride = (long long)hw * hw;
            sg.M = hw; sg.N = hw; sg.K = C; sg.batch = n; sg.alpha = 1.0f / sqrtf((float)C); sg.store = STORE_ROWS;
            sg.zero_page = h->zero_page;          // lets the LDS-DMA kernel take the batched product
            if (md.fast && mfma_gemm_ok(sg, adt, adt, adt)) HIPCHK(launch_mfma_gemm(sg, adt, adt, adt, st));
            else if (s1_split_product(h, md, sg)) HIPCHK(launch_split_gemm(sg, st));
            else HIPCHK(launch_gemm_generic(sg, adt, adt, adt, st));
        }
        { Timed t(h, "softmax", st); HIPCHK(launch_softmax_rows(h->as, adt, n * hw, hw, st)); }
        {   // o[i, c] = sum_j w[i, j] v[c, j]
            Timed t(h, "attn_gemm", st);
            GemmArgs sg{};
            sg.A = h->as; sg.lda = hw; sg.a_batch_stride = (long long)hw * hw;
            sg.Bw = h->av; sg.ldb = hw; sg.b_batch_stride = (long long)hw * C;
            sg.C = h->ao; sg.ldc = C; sg.c_batch_stride = (long long)hw * C;
            sg.M = hw; sg.N = C; sg.K = hw; sg.batch = n; sg.alpha = 1.0f; sg.store = STORE_ROWS;
            sg.zero_page = h->zero_page;
            if (md.fast && mfma_gemm_ok(sg, adt, adt, adt)) HIPCHK(launch_mfma_gemm(sg, adt, adt, adt, st));
            else if (s1_split_product(h, md, sg)) HIPCHK(launch_split_gemm(sg, st));
            else HIPCHK(launch_gemm_generic(sg, adt, adt, adt, st));
        }
        g = conv_args(h->ao, n, res, C, 1, 0, t1, C);
        g.resid = cur;
        CHK(s1_plain(c, &g, l.proj));
        CHK(run_linear(h, md, g, l.proj, adt, adt, st, "conv1x1"));
        std::swap(cur, t1);
    } else {
        return fail(HQT_ERR_INVALID, "s1_layer: kind %d", l.kind);
    }
    return HQT_OK;
}

static int decode_chunk(hqt_handle* h, int n, const int64_t* code_t, const int64_t* code_m, const int64_t* code_b, int seq_layout,
                        float* out, int clamp01, const Mode& md, hipStream_t st) {
    const hqt_config& cf = h->cfg;
    const int adt = md.act_dt();
    const int r = h->dec.front().res, E = cf.s1_embed_dim;
    h->gn_ready.tensor = nullptr;
    const bool l3 = cf.code_levels == 3;
    {
        Timed t(h, "quant_gather", st);
        if (l3) {
            QuantArgs3 q{code_t, code_m, code_b, seq_layout, W1(h, "quantizers.0.embedding"), W1(h, "quantizers.1.embedding"),
                         W1(h, "quantizers.2.embedding"), h->quant, n, r, E, adt, cf.s1_n_embed};
            HIPCHK(launch_quant_gather3(q, st));
        } else {
            QuantArgs q{code_t, code_b, seq_layout, W1(h, "quantize_t.embedding"), W1(h, "quantize_b.embedding"), h->quant, n, r, E, adt, cf.s1_n_embed};
            HIPCHK(launch_quant_gather(q, st));
        }
    }
    S1Ctx c{h, n, md, st, adt, h->act[0], h->act[1], h->act[2], h->act[3], h->gn, h->gn + (size_t)h->dec_chunk * 64};
    {
        GemmArgs g = conv_args(h->quant, n, r, l3 ? E : 2 * E, 1, 0, c.cur, cf.s1_z_channels);
        CHK(s1_plain(c, &g, h->post_quant));
        CHK(run_linear(h, md, g, h->post_quant, adt, adt, st, "conv1x1"));
    }
    for (size_t li = 0; li < h->dec.size(); ++li) {
        const DecLayer& l = h->dec[li];
        if (l.kind != 4) { CHK(s1_layer(c, l, li + 1 < h->dec.size() ? &h->dec[li + 1] : nullptr)); continue; }
        // norm_out -> swish -> conv_out, NCHW fp32 (+clamp)
        const int hw = l.res * l.res;
        GemmArgs g = conv_args(c.cur, n, l.res, l.cin, 9, 0, out, l.cout);
        g.store = STORE_NCHW; g.rows_per_image = hw; g.clamp01 = clamp01;
        if (md.split && l.conv1.w32) {       // SPLIT: the fp32 tensor is read once -- GroupNorm + swish + the three output channels in one fp32 kernel
            GemmArgs d = g;
            d.N = l.conv1.N; d.K = l.conv1.K; d.Bw = l.conv1.w32; d.bias = l.conv1.b32; d.alpha = 1.0f;
            with_gn(d, c.gn1, l.n1_g, l.n1_b, 1);
            if (conv_out_direct_ok(d)) {
                {
                    Timed t(h, "gn_stats", st);
                    if (h->gn_ready.tensor == c.cur && h->gn_ready.dbl)
                        HIPCHK(launch_gn_finalize_tiles_d(reinterpret_cast<const double*>(h->gn_tiles), c.gn1, n, h->gn_ready.tiles, hw, l.cin, 32, 1e-6f, st));
                    else
                        HIPCHK(launch_gn_stats_fast(c.cur, c.gn1, h->gn_partial, n, hw, l.cin, 32, 1e-6f, st, DT_F32));
                }
                Timed t(h, "conv_out", st);
                HIPCHK(launch_conv_out_direct(d, st));
                count_variant(h, "variant:conv_out_direct:conv_out");
                continue;
            }
        }
        CHK(s1_norm(c, c.cur, l.cin, hw, c.gn1, l.n1_g, l.n1_b, 1, &g, &l.conv1));
        CHK(run_linear(h, md, g, l.conv1, adt, DT_F32, st, "conv_out"));
    }
    return HQT_OK;
}

static int decode_impl(hqt_handle* h, int B, const int64_t* code_t, const int64_t* code_m, const int64_t* code_b, int seq_layout, float* out,
                       int clamp01, int precision, void* stream, int levels) {
    if (!h || !out) return fail(HQT_ERR_INVALID, "null argument");
    if (!h->finalized) return fail(HQT_ERR_STATE, "hqt_finalize_weights has not run");
    if (!h->cfg.has_stage1) return fail(HQT_ERR_STATE, "handle was created without stage 1");
    if ((h->cfg.code_levels == 3) != (levels == 3)) return fail(HQT_ERR_STATE, "stage 1 has %d code levels: use the matching decode entry point", h->cfg.code_levels == 3 ? 3 : 2);
    if (!code_t && !code_b && !code_m) return fail(HQT_ERR_INVALID, "every code grid is NULL");
    if (B < 1) return fail(HQT_ERR_INVALID, "B must be >= 1");
    ON_DEVICE(h);
    Mode md;
    CHK(mode_of(precision, true, &md));
    const int r = h->dec.front().res;
    const int rt = levels == 3 ? r / 4 : r / 2, rm = r / 2;
    const size_t out_per = (size_t)h->cfg.s1_out_ch * h->dec.back().res * h->dec.back().res;
    for (int b0 = 0; b0 < B; b0 += h->dec_chunk) {
        const int n = std::min(h->dec_chunk, B - b0);
        const int64_t* ct = code_t ? code_t + (size_t)b0 * rt * rt : nullptr;
        const int64_t* cm = code_m ? code_m + (size_t)b0 * rm * rm : nullptr;       // every layout holds rm*rm / r*r codes per image
        const int64_t* cb = code_b ? code_b + (size_t)b0 * r * r : nullptr;
        CHK(decode_chunk(h, n, ct, cm, cb, seq_layout, out + (size_t)b0 * out_per, clamp01, md, (hipStream_t)stream));
    }
    return HQT_OK;
}
extern "C" int hqt_decode(hqt_handle* h, int B, const int64_t* code_t, const int64_t* code_b, float* out, int clamp01,
                          int precision, void* stream) {
    return decode_impl(h, B, code_t, nullptr, code_b, 0, out, clamp01, precision, stream, 2);
}
extern "C" int hqt_decode_seq(hqt_handle* h, int B, const int64_t* codes_top, const int64_t* codes_bot, float* out,
                              int clamp01, int precision, void* stream) {
    return decode_impl(h, B, codes_top, nullptr, codes_bot, 1, out, clamp01, precision, stream, 2);
}
extern "C" int hqt_decode_l3(hqt_handle* h, int B, const int64_t* code_t, const int64_t* code_m, const int64_t* code_b, float* out,
                             int clamp01, int precision, void* stream) {
    return decode_impl(h, B, code_t, code_m, code_b, 0, out, clamp01, precision, stream, 3);
}
extern "C" int hqt_decode_seq_l3(hqt_handle* h, int B, const int64_t* codes0, const int64_t* codes1, const int64_t* codes2, float* out,
                                 int clamp01, int precision, void* stream) {
    return decode_impl(h, B, codes0, codes1, codes2, 1, out, clamp01, precision, stream, 3);
}

// ------------------------------------------------------------------------------------------ stage 1, encode side
// Encoder.forward + quant_conv_b for one chunk of images: h rows (fp32 NHWC [n, r, r, E]) into h_rows
static int encode_chunk(hqt_handle* h, int n, const float* pixels, float* h_rows, const Mode& md, hipStream_t st) {
    const hqt_config& cf = h->cfg;
    const int adt = md.act_dt();
    h->gn_ready.tensor = nullptr;
    S1Ctx c{h, n, md, st, adt, h->act[0], h->act[1], h->act[2], h->act[3], h->gn, h->gn + (size_t)h->dec_chunk * 64};
    for (auto& l : h->enc) {
        if (l.kind == 6) {                  // conv_in (layers.py:212-216): 3x3 stride 1, or 4x4 stride 2 with use_init_downsample
            const int cp = conv_in_cpad(cf), down = cf.s1_use_init_downsample ? 1 : 0;
            { Timed t(h, "image_layout", st); HIPCHK(launch_image_to_nhwc(pixels, c.t2, adt, n, l.res, cp, st)); }
            GemmArgs g = conv_args(c.t2, n, l.res >> down, cp, down ? 16 : 9, 0, c.t1, l.cout);
            g.conv_stride2 = down;
            CHK(run_linear(h, md, g, l.conv1, adt, adt, st, "conv_in"));
            std::swap(c.cur, c.t1);
        } else if (l.kind == 7) {           // norm_out -> swish -> conv_out (layers.py:289-292), then quant_conv_b (generator.py:299) in fp32 rows
            const int hw = l.res * l.res;
            GemmArgs g = conv_args(c.cur, n, l.res, l.cin, 9, 0, c.t1, l.cout);
            CHK(s1_norm(c, c.cur, l.cin, hw, c.gn1, l.n1_g, l.n1_b, 1, &g, &l.conv1));
            CHK(run_linear(h, md, g, l.conv1, adt, adt, st, "conv3x3"));
            GemmArgs q = conv_args(c.t1, n, l.res, l.cout, 1, 0, h_rows, cf.s1_embed_dim);
            CHK(s1_plain(c, &q, h->quant_conv));
            CHK(run_linear(h, md, q, h->quant_conv, adt, DT_F32, st, "conv1x1"));
        } else {
            CHK(s1_layer(c, l));
        }
    }
    return HQT_OK;
}

extern "C" int hqt_has_encoder(const hqt_handle* h) { return h ? (h->has_encoder ? 1 : 0) : -1; }

extern "C" int hqt_encode(hqt_handle* h, int B, const float* pixels, int precision, const hqt_encode_out* out, void* stream) {
    if (!h || !pixels || !out) return fail(HQT_ERR_INVALID, "null argument");
    if (!h->finalized) return fail(HQT_ERR_STATE, "hqt_finalize_weights has not run");
    if (!h->cfg.has_stage1) return fail(HQT_ERR_STATE, "handle was created without stage 1");
    if (!h->has_encoder) return fail(HQT_ERR_STATE, "the encoder tensors (stage1.encoder.*, stage1.quant_conv_b.*) were not set before hqt_finalize_weights");
    if (B < 1 || B > h->cfg.max_batch) return fail(HQT_ERR_INVALID, "B must be in [1, max_batch = %d]", h->cfg.max_batch);
    const hqt_config& cf = h->cfg;
    const int L = cf.code_levels == 3 ? 3 : 2;
    for (int l = 0; l < L; ++l) if (!out->codes[l]) return fail(HQT_ERR_INVALID, "codes[%d] is NULL", l);
    ON_DEVICE(h);
    hipStream_t st = (hipStream_t)stream;
    Mode md;
    CHK(mode_of(precision, true, &md));
    const int r = h->dec.front().res, E = cf.s1_embed_dim, R = cf.s1_resolution;
    for (int b0 = 0; b0 < B; b0 += h->dec_chunk) {
        const int n = std::min(h->dec_chunk, B - b0);
        CHK(encode_chunk(h, n, pixels + (size_t)b0 * 3 * R * R, h->vq_h + (size_t)b0 * r * r * E, md, st));
    }
    // residual quantisation, coarse -> fine, over the whole batch (generator.py:300-309 / 541-560)
    const size_t elems = (size_t)B * r * r * E;
    for (int l = 0; l < L; ++l) {
        const int k = L - 1 - l, rq = r >> k, dim = E << (2 * k), M = B * rq * rq;
        const std::string name = L == 3 ? "quantizers." + std::to_string(l) + ".embedding" : (l == 0 ? "quantize_t.embedding" : "quantize_b.embedding");
        VqArgs a{};
        a.h = h->vq_h; a.recon = l == 0 ? nullptr : h->vq_recon;
        a.B = B; a.r = r; a.E = E; a.k = k;
        a.z = h->vq_z; a.z_dtype = DT_F32; a.zz = h->vq_zz; a.resid_nchw = out->resid[l];
        a.best = h->vq_best; a.emb = W1(h, name); a.codes = out->codes[l]; a.quant_nchw = out->quant[l];
        a.err_rows = h->vq_err + (size_t)l * cf.max_batch * r * r;
        { Timed t(h, "vq_rows", st); HIPCHK(launch_vq_rows(a, st)); }
        HIPCHK(hipMemsetAsync(h->vq_best, 0xff, (size_t)M * 8, st));
        {
            Timed t(h, "vq_distance", st);
            GemmArgs g{};
            g.A = h->vq_z; g.lda = dim; g.M = M; g.N = cf.s1_n_embed; g.K = dim; g.batch = 1; g.ldb = dim; g.alpha = 1.0f;
            g.store = STORE_ARGMIN; g.am_rownorm = h->vq_zz; g.am_best = h->vq_best; g.zero_page = h->zero_page;
            // fp32 in both precisions: h leaves quant_conv_b in fp32 either way, the fp32 vector-ALU GEMM runs these two shapes at
            // ~70 TFLOP/s (1.8 ms of a 8 ms FAST encode at batch 64), and the codes then are the exact nearest ones of the
            // device's own feature map instead of a bf16 approximation of them
            g.Bw = a.emb; g.am_colnorm = h->cb_norm[l];
            HIPCHK(launch_gemm_generic(g, DT_F32, DT_F32, DT_F32, st));
        }
        if (l == 0) {                       // the running reconstruction starts at zero; level 0 reads h alone (generator.py:300-301)
            HIPCHK(hipMemsetAsync(h->vq_recon, 0, elems * 4, st));
            a.recon = h->vq_recon;
            // with recon == 0 the finish pass computes z = h - 0 and recon = q + 0: the same values as the reference's h_t / PS(quant_t)
        }
        { Timed t(h, "vq_finish", st); HIPCHK(launch_vq_finish(a, st)); }
        if (out->diff) HIPCHK(launch_vq_diff(a.err_rows, M, 0.25f / ((float)M * (float)dim), out->diff + l, st));
    }
    if (out->recon) HIPCHK(launch_nhwc_to_nchw_f32(h->vq_recon, out->recon, B, r * r, E, st));
    return HQT_OK;
}

extern "C" int hqt_set_policy(hqt_handle* h, int policy) {
    if (!h) return fail(HQT_ERR_INVALID, "null handle");
    if (policy != HQT_POLICY_LATENCY && policy != HQT_POLICY_THROUGHPUT) return fail(HQT_ERR_INVALID, "unknown policy %d", policy);
    h->policy = policy;
    return HQT_OK;
}

extern "C" int hqt_set_switch(hqt_handle* h, int which, int on) {
    if (!h) return fail(HQT_ERR_INVALID, "null handle");
    if (which == HQT_SWITCH_PERSIST) { h->persist_enabled = on != 0; if (on) h->persist_tripped = false; }
    else if (which == HQT_SWITCH_SINGLE_KEY) h->single_key = on != 0;
    else if (which == HQT_SWITCH_SPLIT_KSLICES) h->split_kslices = on != 0;
    else if (which == HQT_SWITCH_PERSIST_FAULT) {        // test hook: a device word next to the give-up mark (persist.h: err[1]); NOT part of the graph key -- a cached graph replays it
        if (!h->persist_err) return fail(HQT_ERR_STATE, "no persistent chain on this handle (a lane, or a handle without stage 2)");
        ON_DEVICE(h);
        const unsigned v = (unsigned)on;
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(h->persist_err + 1, &v, sizeof v, hipMemcpyHostToDevice));
    }
    else return fail(HQT_ERR_INVALID, "unknown switch %d", which);
    return HQT_OK;
}

// ------------------------------------------------------------------------------------------ range check of SPLIT calls
extern "C" int hqt_range_check(hqt_handle* h, void* stream) {
    if (!h) return fail(HQT_ERR_INVALID, "null");
    ON_DEVICE(h);
    int flag = 0;
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (h->persist_used && h->persist_err) {     // a persistent AR launch that gave up on a barrier (1 s: the GPU was shared with something that kept its workgroups out)
        unsigned pe = 0;
        HIPCHK(hipMemcpy(&pe, h->persist_err, sizeof pe, hipMemcpyDeviceToHost));
        h->persist_used = false;
        if (pe) {
            HIPCHK(hipMemset(h->persist_err, 0, sizeof pe));
            h->persist_tripped = true;           // part of the graph key: the next call captures the launch chain
            return fail(HQT_ERR_STATE, "a persistent AR launch on this handle gave up at the grid barrier in front of phase %u (its workgroups were not all resident within 1 s: the GPU is shared with a process that keeps compute units busy): "
                                       "the codes of that call are invalid; repeat it -- this handle now takes the launch chain (hqt_set_switch(h, HQT_SWITCH_PERSIST, 1) re-arms the persistent launch)", pe - 1);
        }
    }
    HIPCHK(hipMemcpy(&flag, h->range_flag, sizeof flag, hipMemcpyDeviceToHost));
    if (!flag) return HQT_OK;
    HIPCHK(hipMemset(h->range_flag, 0, sizeof flag));
    return fail(HQT_ERR_RANGE, "a SPLIT-precision call on this handle met an activation outside the fp16 range (NaN or |x| >= 65504): its output is "
                               "invalid; repeat the call with HQT_PRECISION_EXACT");
}

// ------------------------------------------------------------------------------------------ introspection
extern "C" int64_t hqt_param_count(const hqt_handle* h, int stage) { return (h && (stage == 1 || stage == 2)) ? h->params[stage] : -1; }
extern "C" int64_t hqt_workspace_bytes(const hqt_handle* h) { return h ? (int64_t)h->workspace_bytes : -1; }
extern "C" int hqt_timing_enable(hqt_handle* h, int on) { if (!h) return fail(HQT_ERR_INVALID, "null"); timing_collect(h); h->timing = on != 0; return HQT_OK; }
extern "C" int hqt_timing_reset(hqt_handle* h) {
    if (!h) return fail(HQT_ERR_INVALID, "null");
    timing_collect(h);
    for (auto& s : h->slots) { s.launches = 0; s.total_ms = 0.0; }
    return HQT_OK;
}
extern "C" int hqt_timing_slots(const hqt_handle* h) { return h ? (int)h->slots.size() : -1; }
extern "C" int hqt_timing_get(hqt_handle* h, int slot, char* name, int name_len, int64_t* launches, double* total_ms) {
    if (!h || slot < 0 || slot >= (int)h->slots.size()) return fail(HQT_ERR_INVALID, "bad slot");
    timing_collect(h);
    const TimingSlot& s = h->slots[slot];
    if (name && name_len > 0) { strncpy(name, s.name.c_str(), name_len - 1); name[name_len - 1] = 0; }
    if (launches) *launches = s.launches;
    if (total_ms) *total_ms = s.total_ms;
    return HQT_OK;
}
