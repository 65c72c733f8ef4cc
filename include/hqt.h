/*
 * hqt.h -- C ABI of libhqt.so, the MI355X (gfx950) HQ-Transformer sampling engine.
 *
 * The reference (kakaobrain/hqtransformer) has no FFI/plugin layer: its sampling path is ordinary
 * nn.Module methods.  The boundary is therefore the Python call surface its drivers use (SURVEY.md
 * §8b); hqtransformer_amd/ re-exposes that surface and binds the entry points below with ctypes.
 * Each entry point names the reference interface it replaces (paths relative to the reference root).
 *
 * Conventions: plain pointers and sizes only (no torch types).  Every tensor pointer is a DEVICE
 * pointer owned by the caller unless stated otherwise.  Calls enqueue work on `stream` (a
 * hipStream_t passed as void*; NULL = the null stream) and return without synchronising.  Every
 * function returns 0 on success or a negative hqt_status; the message is available from
 * hqt_last_error() (thread-local).  Nothing throws across the ABI.  A handle is bound to one device
 * and must be driven by one host thread at a time.
 */
#ifndef HQT_H
#define HQT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HQT_ABI_VERSION 7

typedef enum {
    HQT_OK = 0,
    HQT_ERR_INVALID = -1,      /* bad argument / unsupported configuration   */
    HQT_ERR_HIP = -2,          /* a HIP runtime call failed                  */
    HQT_ERR_STATE = -3,        /* call order (e.g. sample before finalize)   */
    HQT_ERR_UNKNOWN_WEIGHT = -4,
    HQT_ERR_SHAPE = -5,
    HQT_ERR_MISSING_WEIGHT = -6,
    HQT_ERR_RANGE = -7         /* SPLIT precision: an activation left the fp16 range (hqt_range_check) */
} hqt_status;

/* conditioning of the top GPT -- hqvae/models/stage2/hierarchical_ar.py:64-78 */
enum { HQT_COND_NONE = 0, HQT_COND_CLASS = 1, HQT_COND_TEXT = 2 };
/* input embedding -- hierarchical_ar.py:83-113 ('transformer1' has zero embedding blocks) */
enum { HQT_EMB_TRANSFORMER1 = 0, HQT_EMB_REDUCE = 1 };
/* arithmetic of a call.  EXACT = fp32 weights/activations/accumulation (what the reference computes
 * on its CPU path, where autocast is off); FAST = bf16 weights and MFMA, fp32 accumulation, fp32
 * softmax/normalisation/sampler (the counterpart of the reference's use_fp16=True autocast path).
 * SPLIT (stage-1 entry points: decode / encode) = fp32 tensors, fp32 GroupNorm / softmax / activations, and the
 * convolutions on the matrix cores with every fp32 operand carried as two fp16 values (hi, lo * 2^11): three
 * v_mfma_f32_16x16x32_f16 per product term, fp32 accumulation -- fp32-accurate (2^-22 per operand), what the reference's
 * fp32 decode (measure_throughput/__main__.py:108-113, outside autocast) computes up to summation order, at matrix-core
 * speed; layers whose shapes the split kernels do not take run the EXACT kernels.
 * SPLIT on the stage-2 entry points (hqt_sample / hqt_sample_l3, round 4): the EXACT launch sequence -- fp32 activations, LayerNorm,
 * attention, softmax and sampler exactly as EXACT -- with every nn.Linear (stage2/layers.py:73-85,190,313-315; the heads) on the matrix
 * cores: the fp32 activation rows are split into fp16 hi / lo while their tile is staged, the weights travel as fp16 hi / lo planes,
 * three v_mfma_f32_32x32x16_f16 per product term.  Code sequences are bit-identical to EXACT (and to the reference's CPU path) wherever
 * the draw is well-conditioned, logits within 2e-4; hqt_range_check applies as for stage 1. */
enum { HQT_PRECISION_EXACT = 0, HQT_PRECISION_FAST = 1, HQT_PRECISION_SPLIT = 2 };
enum { HQT_DTYPE_F32 = 0 };

typedef struct hqt_handle hqt_handle;

/* Model description.  Stage-2 fields restate the arguments of iHQGPT.__init__
 * (hierarchical_ar.py:24-33) as ImageGPT2 passes them (hqvae/models/__init__.py:123-137); stage-1
 * fields restate SimRQGAN2Generator / Decoder (hqvae/models/stage1/generator.py:179-259,
 * hqvae/models/stage1/modules/layers.py:300-383).  Set has_stage2 / has_stage1 to 0 to build half. */
typedef struct {
    int32_t abi_version;            /* HQT_ABI_VERSION */
    /* stage 2: 'hq-transformer/parallel', ratio_bot2top = 4 */
    int32_t has_stage2;
    int32_t embed_dim, n_layers, n_heads, n_layers_depth;
    int32_t vocab_top, vocab_bot, vocab_txt;
    int32_t ctx_len_img, ctx_len_txt, n_classes;
    int32_t cond_type;              /* HQT_COND_*  */
    int32_t embedding_type;         /* HQT_EMB_*   */
    int32_t gelu_approx;            /* hparams.gelu_use_approx */
    /* stage 1: 'simrqgan2', upsample = pixelshuffle(2), decoding_type = concat */
    int32_t has_stage1;
    int32_t s1_ch, s1_n_mult, s1_ch_mult[8];
    int32_t s1_num_res_blocks;
    int32_t s1_n_attn_res, s1_attn_res[4];
    int32_t s1_resolution, s1_z_channels, s1_embed_dim, s1_n_embed, s1_out_ch;
    int32_t s1_use_init_downsample, s1_use_mid_block, s1_use_attn;
    /* sizing */
    int32_t max_batch;              /* largest B of any later call; workspaces are sized once */
    int32_t max_steps;              /* largest number of top positions per call (<= ctx_len_img) */
    /* three code levels: stage 2 = HQTransformer 'multilevel-hq' / 'parallel-add'
     * (hqvae/models/stage2/hqtransformer.py:24-205, one vocabulary size for all levels = vocab_top), stage 1 =
     * HQVAEGenerator with code_levels = 3 (generator.py:451-515).  0 or 2: the two-level models above. */
    int32_t code_levels;
    /* three levels only: HQTransformer.decoding_type (hqtransformer.py:105-157: tables of the depth head; :526-551: what the
     * depth sub-steps are fed).  0 'parallel-add' (the released level-3 config), 1 'parallel' (level-2 tokens without the top
     * code's embedding), 2 'parallel-reduce' (tok_emb_depth_levels.{0,1} are [V, 4 D]: child position c of a code takes slice c).
     * 3 'top2mid2bot': a causal head of 21 sequential one-token sub-steps (hqtransformer.py:700-800; pos_emb_depths.0 is [21, D],
     * the sub-step inputs come from tok_emb_levels).  The other values of the reference ('tree', 'old-parallel',
     * 'parallel-add-reduce') cannot sample three levels there either. */
    int32_t depth_decoding;
    /* Which derived layouts of the AR loop's nn.Linear weights hqt_finalize_weights builds (bit mask of HQT_LAYOUT_*; 0 = all of them,
     * what ABI <= 6 always did: 9.6 GiB measured for the 2.1 GB ImageNet-12L model and its batch-64 workspace).  The fp32 tensors as received are always kept: they are
     * what EXACT computes from.  A replica that only ever samples in FAST precision passes HQT_LAYOUT_FAST (5.95 GiB measured); a hqt_sample /
     * hqt_sample_l3 call in a precision whose layout the handle was built without fails with HQT_ERR_STATE (EXACT without
     * HQT_LAYOUT_EXACT still runs, on the row-major fp32 weights: same results, slower below 257 rows).  Stage 1 is not affected. */
    int32_t ar_layouts;
} hqt_config;
#define HQT_LAYOUT_FAST 1   /* bf16 MFMA-fragment packing (+ the LayerNorm-folded copies, the persistent chain's per-CU stream) */
#define HQT_LAYOUT_EXACT 2  /* fragment-ordered fp32 copy (v_mfma_f32_16x16x4_f32 kernel of passes up to 256 rows) */
#define HQT_LAYOUT_SPLIT 4  /* fp16 hi / lo planes (SPLIT precision on the stage-2 entry points) */
#define HQT_LAYOUT_ALL 7
#define HQT_DEPTH_PARALLEL_ADD 0
#define HQT_DEPTH_PARALLEL 1
#define HQT_DEPTH_PARALLEL_REDUCE 2
#define HQT_DEPTH_TOP2MID2BOT 3

/* Sampling options = the keyword arguments of sampling_ihqgpt (hqvae/utils/sampling.py:164-177).
 * top_k <= 0 means None (no cut-off), top_p <= 0 means None. */
typedef struct {
    int32_t precision;              /* HQT_PRECISION_*  (use_fp16=True -> FAST) */
    int32_t n_steps;                /* max_seq_len: top positions to generate */
    int32_t top_k_top, top_k_bot;
    float top_p_top, top_p_bot;
    float temperature_top, temperature_bot;       /* softmax_temperature[0], [1] */
    uint64_t seed;                  /* Philox seed, used when noise == NULL */
    int64_t sample_offset;          /* global index of row 0 (sharded batches draw the noise of the
                                       global batch: Philox counters are keyed by global row) */
    int32_t use_graph;              /* 1: replay the per-position launch sequence from a hipGraph */
    /* Merged steps: several independent sampling_ihqgpt calls (each with its own seed and global offset) executed as ONE
     * batch, so that the weights are streamed once for all of them.  Optional HOST arrays of B entries: row b then draws
     * exactly what row (b - first row of its call) of a separate hqt_sample call with (row_seeds[b], sample_offset =
     * row_offsets[b] - that row index) would draw -- Philox counters are keyed by (seed, global row), nothing else.
     * NULL: every row uses `seed` and `sample_offset + b`. */
    const uint64_t* row_seeds;
    const int64_t* row_offsets;
} hqt_sample_opts;

/* hqt_create -- replaces ImageGPT2(config) construction (hqvae/models/__init__.py:92-174) for the
 * tensors of the sampling path; allocates all device workspaces (KV cache, activations). */
int hqt_create(const hqt_config* cfg, int device, hqt_handle** out);

/* hqt_set_weight -- replaces load_state_dict (sampling_hqmodel.py:77-79): `name` is the reference
 * state-dict key without the 'stage1.' / 'stage2.' prefix namespace collision ('stage2.' keys are
 * passed as e.g. "stage2.blocks.0.attn.key.weight", stage-1 keys as "stage1.decoder.conv_in.weight").
 * `data` may be a host or a device pointer to contiguous fp32 in the reference's own layout
 * ([out,in] Linear, [O,I,kh,kw] Conv2d, [n_embed,dim] codebooks); it is copied before return. */
int hqt_set_weight(hqt_handle* h, const char* name, const void* data, int dtype, const int64_t* shape, int ndim);

/* hqt_finalize_weights -- checks every tensor arrived, builds the device layouts the kernels read
 * (fused QKV, tap-major conv filters, bf16 MFMA-fragment packing for FAST). */
int hqt_finalize_weights(hqt_handle* h);

/* hqt_clone -- a further LANE over the weights of a finalized handle.  The reference's throughput harness
 * (measure_throughput/__main__.py:84-116) runs one batch at a time; its 64-row AR loop is a chain of small
 * latency-bound kernels that leaves most of an MI355X idle, so the harness here keeps several batches in flight,
 * one lane and one HIP stream each (independent chains interleave on the GPU).  The clone shares every weight
 * buffer of `src` and owns only its workspace (KV cache, activations, step state, graph cache); results of a lane
 * are bit-identical to the parent's.  Destroy clones before their parent (hqt_destroy(parent) fails otherwise). */
int hqt_clone(hqt_handle* src, hqt_handle** out);

/* hqt_set_policy -- kernel selection for the AR loop.  LATENCY (default): tile shapes that finish one batch soonest
 * (every GEMM spread over as many CUs as possible).  THROUGHPUT: shapes that cost the fewest CU-microseconds (64-row
 * weight tiles: half the activation re-reads per weight byte), for several lanes in flight (hqt_clone), where kernels
 * of different batches share the chip: ~7 % more images/s with 3 lanes, ~15 % slower alone.  Results are unchanged up
 * to fp32 summation order in FAST arithmetic (EXACT arithmetic does not use these kernels). */
enum { HQT_POLICY_LATENCY = 0, HQT_POLICY_THROUGHPUT = 1 };
int hqt_set_policy(hqt_handle* h, int policy);

/* hqt_set_switch -- two choices of launch sequence that do not change what is computed beyond fp32 summation order (no reference
 * counterpart; A/B runs and tests).  Both are part of the graph key: the next hqt_sample re-captures if one changed.
 *   HQT_SWITCH_PERSIST     1 (default): FAST hqt_sample of up to 64 rows on a root handle runs a top position as ONE persistent launch
 *                          (csrc/persist.h); 0: the launch chain.  The default is 0 when HQT_PERSIST=0 was in the environment at hqt_create
 *                          -- the environment is read there, once, never per call.  A handle whose persistent launch gave up
 *                          (hqt_range_check) switches itself to 0; setting 1 re-arms it.
 *   HQT_SWITCH_SINGLE_KEY  1 (default): depth sub-step 0 (one query over one key) skips the query third of the fused GEMM and the
 *                          attention launch, bit-identically; 0: the long way round (default 0 when HQT_NO_SINGLE_KEY was set at hqt_create). */
/*   HQT_SWITCH_SPLIT_KSLICES  1 (default): SPLIT-precision hqt_sample cuts K of the narrow GEMMs (proj / fc2) of passes up to 1024 rows into slices summed in index order
 *                          (deterministic; the fp32 summation order, hence the last bits of a logit, then depends on the row count of the pass); 0: one order at every
 *                          row count (default 0 when HQT_SPLIT_KSLICES_OFF was set at hqt_create).
 *   HQT_SWITCH_PERSIST_FAULT  test hook (0 = none, the default): on = c + 1 makes compute unit c withhold its first grid-barrier signal in every
 *                          later persistent launch, so the launch gives up after its time limit (tests/test_gpu_persist.py).  A device word, not
 *                          part of the graph key: a cached graph replays it.  Synchronises the device. */
enum { HQT_SWITCH_PERSIST = 0, HQT_SWITCH_SINGLE_KEY = 1, HQT_SWITCH_PERSIST_FAULT = 2, HQT_SWITCH_SPLIT_KSLICES = 3 };
int hqt_set_switch(hqt_handle* h, int which, int on);

/* hqt_sample -- replaces sampling_ihqgpt + iHQGPT.sampling_step (hqvae/utils/sampling.py:164-237,
 * hierarchical_ar.py:428-480, 482-563, 667-789) for a batch of B independent images.
 *   cond        int64 [B] class ids (HQT_COND_CLASS), int64 [B, ctx_len_txt] token ids
 *               (HQT_COND_TEXT), ignored/NULL (HQT_COND_NONE)
 *   noise       fp32 [n_steps, 5, B, V] Exp(1) variates q, draw order top, bot0..bot3; the draw is
 *               argmax(p / q) (== torch.multinomial(p, 1) for the q it draws).  NULL: Philox(seed).
 *   force_top   optional int64 [B, n_steps]: the top code fed back instead of the drawn one
 *               (given_top_code, hierarchical_ar.py:768-774); the draw is still written to out_top
 *   force_bot   optional int64 [B, n_steps, 4]: same for the four bottom codes (teacher forcing)
 *   logits_out  optional fp32 [n_steps, 5, B, V]: raw (pre-temperature) logits of every draw
 *   out_top     int64 [B, n_steps]      codes_top
 *   out_bot     int64 [B, n_steps, 4]   codes_bot, slot = 2*kh + kw */
int hqt_sample(hqt_handle* h, int B, const int64_t* cond, const hqt_sample_opts* opts, const float* noise,
               const int64_t* force_top, const int64_t* force_bot, float* logits_out,
               int64_t* out_top, int64_t* out_bot, void* stream);

/* Three-level counterparts.  hqt_sample_l3 replaces sampling_hqtransformer + HQTransformer.sampling_step
 * (hqvae/utils/sampling.py:240-307, hqtransformer.py:409-635): per top position 1 + 4 + 16 codes.
 *   noise / logits_out  fp32 [n_steps, 21, B, V], draw order level 0, level-1 slots 0..3, level-2 tokens 0..15 in
 *                       (H1 H2 W1 W2) raster order (= kh * 4 + kw of the 4x4 block)
 *   force0/1/2          optional teacher forcing, int64 [B, n_steps] / [B, n_steps, 4] / [B, n_steps, 16]
 *   out0/1/2            int64 [B, n_steps], [B, n_steps, 4], [B, n_steps, 16]
 * hqt_decode_l3 replaces HQVAEGenerator.decode_code([t, m, b]) (generator.py:577-599): code grids [B, r/4, r/4],
 * [B, r/2, r/2], [B, r, r] (any may be NULL = zero quant); hqt_decode_seq_l3 takes the sampler's own outputs and folds
 * the rearranges of sampling_hqmodel.py:150-153 into the lookup. */
typedef struct {
    int32_t precision, n_steps;
    int32_t top_k[3];
    float top_p[3];
    float temperature[3];
    uint64_t seed;
    int64_t sample_offset;
    int32_t use_graph;
    const uint64_t* row_seeds;      /* as in hqt_sample_opts */
    const int64_t* row_offsets;
} hqt_sample_opts_l3;
int hqt_sample_l3(hqt_handle* h, int B, const int64_t* cond, const hqt_sample_opts_l3* opts, const float* noise,
                  const int64_t* force0, const int64_t* force1, const int64_t* force2, float* logits_out,
                  int64_t* out0, int64_t* out1, int64_t* out2, void* stream);
int hqt_decode_l3(hqt_handle* h, int B, const int64_t* code_t, const int64_t* code_m, const int64_t* code_b, float* out_pixels,
                  int clamp01, int precision, void* stream);
int hqt_decode_seq_l3(hqt_handle* h, int B, const int64_t* codes0, const int64_t* codes1, const int64_t* codes2, float* out_pixels,
                      int clamp01, int precision, void* stream);

/* hqt_encode -- replaces SimRQGAN2Generator.encode / get_codes (generator.py:298-310, 369-370) and, on a three-level
 * handle, HQVAEGenerator.encode (generator.py:530-568): Encoder.forward (stage1/modules/layers.py:270-297), quant_conv_b,
 * then per level coarse -> fine: PixelUnshuffle of (h - reconstruction so far), nearest code
 * argmin(|z|^2 + |e|^2 - 2 z.e) (quantizer.py:91-103), reconstruction += PixelShuffle(z + (e - z)).
 * Needs the encoder tensors (stage1.encoder.*, stage1.quant_conv_b.*) to have been set before hqt_finalize_weights;
 * without them the call fails with HQT_ERR_STATE.
 *   pixels   fp32 [B, 3, R, R] NCHW (device)
 * Level index l runs coarse -> fine (two levels: 0 = top, 1 = bottom); r_l = r >> (levels - 1 - l),
 * dim_l = embed_dim * 4^(levels - 1 - l).  Every pointer of hqt_encode_out except codes[] may be NULL.
 *   codes[l]  int64 [B, r_l, r_l]                 (the reference returns them flattened)
 *   quant[l]  fp32 [B, dim_l, r_l, r_l]           the straight-through quantised tensor of level l (quant_t / quant_b)
 *   resid[l]  fp32 [B, dim_l, r_l, r_l]           the quantiser's input of level l (resid[1] of a two-level model is
 *                                                 the reference's code[2] = h_b)
 *   recon     fp32 [B, embed_dim, r, r]           sum of all levels in the bottom layout (HQVAEGenerator: recons[-1])
 *   diff      fp32 [levels]                       0.25 * mean((e - z)^2) per level (quantizer.py:130)
 * EXACT computes every convolution in fp32; FAST runs the convolutions in bf16 MFMA with fp32 accumulation.  The distance
 * GEMM and the argmin are fp32 in both, so the codes are always the exact nearest ones of the feature map the handle
 * computed (under FAST that map, and hence some codes, differ slightly from the fp32 one). */
typedef struct {
    int64_t* codes[3];
    float* quant[3];
    float* resid[3];
    float* recon;
    float* diff;
} hqt_encode_out;
int hqt_encode(hqt_handle* h, int B, const float* pixels, int precision, const hqt_encode_out* out, void* stream);
/* 1 when the handle holds the encoder tensors (hqt_encode is usable), else 0; -1 on a NULL handle */
int hqt_has_encoder(const hqt_handle* h);

/* hqt_decode -- replaces SimRQGAN2Generator.decode_code (generator.py:323-367: codebook lookup
 * quantizer.py:179-186, PixelShuffle, concat, post_quant_conv_b, Decoder.forward layers.py:385-410).
 *   code_t      int64 [B, r/2, r/2] or NULL (that level contributes a zero quant, generator.py:328-358)
 *   code_b      int64 [B, r, r] or NULL          (r = bottom grid = resolution / 2^n_levels)
 *   out_pixels  fp32 [B, out_ch, H, W] NCHW; clamp01 != 0 fuses clamp(0.5 x + 0.5, 0, 1)
 *               (measure_throughput/__main__.py:113) into the last kernel, else raw decoder output
 * hqt_decode_seq takes the sampler's own outputs (codes_top [B, (r/2)^2], codes_bot [B, (r/2)^2, 4])
 * and folds the two rearranges of sampling_hqmodel.py:119-120 into the lookup addressing. */
int hqt_decode(hqt_handle* h, int B, const int64_t* code_t, const int64_t* code_b, float* out_pixels,
               int clamp01, int precision, void* stream);
int hqt_decode_seq(hqt_handle* h, int B, const int64_t* codes_top, const int64_t* codes_bot, float* out_pixels,
                   int clamp01, int precision, void* stream);

/* hqt_range_check -- no reference counterpart (the reference decodes in fp32, generator.py:323-367, where nothing can leave the
 * range).  SPLIT-precision calls carry every fp32 activation as two fp16 values; an activation that is NaN or >= 65504 in magnitude
 * cannot travel that way.  The operand pass saturates it and sets a flag on the handle instead of failing the (asynchronous) call.
 * hqt_range_check synchronises `stream` (the one the calls were enqueued on), returns HQT_ERR_RANGE if any SPLIT call since the last
 * check met such a value (and clears the flag), HQT_OK otherwise.  The Python surface calls it wherever it hands pixels or codes of a
 * SPLIT call to the host side (decode_code / encode of lane 0, InflightSampler.drain).
 * Since round 5 it also reports the one asynchronous failure of a FAST call: hqt_sample of up to 64 samples on a root handle runs a top
 * position as ONE persistent launch that needs every compute unit of the device resident at once (hqtransformer_amd/csrc/persist.h); every
 * spin in it is bounded (1 s), and a launch that could not finish -- the GPU is shared with something that keeps compute units busy --
 * marks the handle instead of hanging: HQT_ERR_STATE here ("... gave up at the grid barrier in front of phase p"), the codes of that call
 * are invalid, later launches return at once until this call has cleared the mark -- and the handle then takes the launch chain
 * (hqt_set_switch(h, HQT_SWITCH_PERSIST, 1) re-arms the persistent launch).  Persistent launches of ALL handles of a process on one
 * device are ordered behind each other (an event chain inside the library), so two root handles sampling on two streams both finish.
 * hqt_set_switch(h, HQT_SWITCH_PERSIST, 0), or HQT_PERSIST=0 in the environment at hqt_create, selects the launch chain from the start. */
int hqt_range_check(hqt_handle* h, void* stream);

/* introspection */
int hqt_abi_version(void);
int64_t hqt_param_count(const hqt_handle* h, int stage);       /* stage 1 | 2: elements received */
int64_t hqt_workspace_bytes(const hqt_handle* h);
/* name of the kernel that dominates phase (0 = AR loop, 1 = decode) and its launch count / total
 * device time in ms since hqt_timing_reset, measured with HIP events on the caller's stream when
 * timing is enabled (bench.py's roofline numerator) */
int hqt_timing_enable(hqt_handle* h, int on);
int hqt_timing_reset(hqt_handle* h);
int hqt_timing_get(hqt_handle* h, int slot, char* name, int name_len, int64_t* launches, double* total_ms);
int hqt_timing_slots(const hqt_handle* h);

int hqt_destroy(hqt_handle* h);
const char* hqt_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* HQT_H */
