// Host-side launch wrappers of every kernel in libhqt.so.  All asynchronous on `st`.
#pragma once
#include "common.h"

enum { DT_F32 = 0, DT_BF16 = 1 };

// generic GEMM (vector ALUs).  ta/tb/tc are DT_*.
hipError_t launch_gemm_generic(const GemmArgs& g, int ta, int tb, int tc, hipStream_t st);
// EXACT nn.Linear of the AR loop on the fp32 matrix instructions, one T x T tile per wave (exact_gemm.hip)
bool exact_mfma_ok(const GemmArgs& g);
hipError_t launch_exact_mfma_gemm(const GemmArgs& g, hipStream_t st);
hipError_t launch_pack_exact_tiles(const float* w, float* out, int N, int K, hipStream_t st);   // fp32 [N][K] -> MFMA fragment order [N / 16][K / 32][chunk][lane][4]

struct EmbedArgs {
    int B, D, n_steps;
    int embedding;               // HQT_EMB_*
    int cond_type;               // HQT_COND_*
    const StepState* state;      // position = state->step
    const int64_t* cond;         // [B] class ids (class-cond)
    const float* sos;            // class: [n_classes, D]; uncond: [D]
    const float* tok_top;        // [V, D]
    const float* tok_bot;        // [V, D] or [V, D/4]
    const float* pos_top;        // [ctx_img, D]
    const float* pos_emb;        // [5, D] (transformer1)
    const int64_t* codes_top;    // [B, n_steps]    (drawn or teacher-forced)
    const int64_t* codes_bot;    // [B, n_steps, 4]
    float* x;                    // [B, D]
    bf16_t* xpk;                 // optional: bf16 copy in the packed_off() layout (pk_mb) + row statistics (FAST deferred LN)
    int pk_mb;
    float* parts;                // [1][32 pk_mb][2] (sum, sumsq) of the bf16 copy
    // three code levels (HQTransformer, hqtransformer.py:466-488): mean over 1 + 4 + 16 tokens; tok_top / tok_bot are then
    // tok_emb_levels.0 / .1, pos_emb has 21 rows
    int levels;                  // 0 or 2: two levels; 3: three
    const float* tok_l2;         // [V, D] tok_emb_levels.2
    const int64_t* codes_l2;     // [B, n_steps, 16]
    int V, n_classes;            // table sizes for the index clamp (0: unchecked)
};
hipError_t launch_embed_step(const EmbedArgs& a, hipStream_t st);

// text prefix: x[b, t, :] = tok_emb_txt[cond[b, t]] + pos_emb_txt[t]
hipError_t launch_embed_text(const int64_t* cond, const float* tok, const float* pos, float* x, int B, int T, int D,
                             hipStream_t st, int vocab = 0);

// depth sub-step 1 input: x[b*4+s, :] = tok_top_depth[top[b, step]] + pos_depth[s]
hipError_t launch_depth_embed(const int64_t* codes_top, int n_steps, const StepState* state, const float* tok,
                              const float* pos, float* x, int B, int D, bf16_t* xpk, int pk_mb, float* parts, hipStream_t st, int V = 0, int tok_ld = 0);

struct LNArgs {
    float* x;                    // [rows_in, D]; rewritten in place when split-K slabs are folded in
    const float* gamma;
    const float* beta;
    const float* add;            // optional [D] vector added after the affine (sos_depth)
    void* y;                     // [M, D] fp32 or bf16
    int M, D;
    int in_rows_per_group;       // input row = m * in_rows_per_group + in_row_offset  (pick one token per sample)
    int in_row_offset;
    float eps;
    int out_dtype;               // DT_*
    int out_packed_mb;           // > 0: write y in the packed_off() layout (bf16 only)
    // pending split-K residual of the previous GEMM: x[m] += slab_bias + sum_s slabs[s][m][:] before normalising
    const float* slabs;          // [n_slabs][slab_rows][D] fp32 or NULL
    int n_slabs;
    int slab_rows;
    const float* slab_bias;      // [D] or NULL
    // optional second output for the FAST deferred-LN path: bf16 packed copy of y (fp32 out) + its row statistics
    bf16_t* ypk;
    int ypk_mb;
    float* yparts;
};
hipError_t launch_layernorm(const LNArgs& a, hipStream_t st);

struct AttnArgs {
    const void* q;               // [B*Tq, D]
    const void* kcache;          // [B, Tmax, D]
    const void* vcache;
    void* out;                   // [B*Tq, D]
    int B, Tq, n_heads, head_dim, Tmax;
    int t_base;                  // keys already cached before this call's Tq tokens
    const int* t_base_dev;       // optional device int added to t_base
    int causal;                  // 1: query i sees keys [0, t_base + i]; 0: all t_base + Tq keys
    int dtype;                   // DT_* of q / cache / out
    int out_packed_mb;           // > 0: write out in the packed_off() layout
    long long* dbg;              // tools/micro only: per-wave clock stamps (NULL in the product)
};
hipError_t launch_attention(const AttnArgs& a, hipStream_t st);

struct SamplerArgs {
    const float* logits;         // [R, V], row r = b * slots + slot
    int R, V, slots, B;
    float temperature;
    int top_k;                   // <= 0: none
    float top_p;                 // <= 0: none
    const float* noise;          // [n_steps, 5, B, V] or NULL
    int draw0;                   // first draw index of slot 0 (0 for top, 1 for bottom; 5 for the third level)
    const StepState* state;      // step of the current call
    const RowKey* rows;          // [B] Philox seed and global row index of every batch row (merged steps: they differ per row)
    int n_steps;
    int64_t* out;                // top: [B, n_steps]; bottom: [B, n_steps, 4]
    float* logits_out;           // optional [n_steps, draws, B, V]
    int draws;                   // draws per top position: 5 (two levels; 0 means 5) or 21 (three levels) -- noise / logits_out stride
    // ---- fused embedding lookup of the code just drawn (top draw, slots == 1): the four input rows of depth sub-step 1,
    //      x[(b * 4 + s)] = emb_tok[code] + emb_pos[s] (hierarchical_ar.py:705-712), plus the packed bf16 copy and row statistics
    //      the deferred-LayerNorm GEMMs read.  emb_tok == NULL: no fusion (depth_embed_kernel does it).
    const float* emb_tok;        // [V, D]
    const float* emb_pos;        // [>= 4, D]
    const int64_t* emb_feed;     // code to embed instead of the drawn one (teacher forcing), same indexing as `out`; NULL: the drawn code
    float* emb_x;                // [4 B, D]
    int emb_D;
    bf16_t* emb_xpk;             // optional packed copy (packed_off layout with emb_pk_mb row blocks)
    int emb_pk_mb;
    float* emb_parts;            // [4 B][2]
    int out_stride, out_slot;    // out_stride > 0: the drawn code goes to out[(b * n_steps + step) * out_stride + out_slot] (slots == 1: one sub-step of the
                                 // 21-step causal head writes ONE slot of a 4- or 16-wide code group)
    int fast_math;               // FAST-precision calls: v_exp / v_log / v_rcp forms of exp, log and the divisions (1-2 ulp each; EXACT keeps the IEEE forms: its draws are the parity gate, compared bit for bit)
};
hipError_t launch_sampler(const SamplerArgs& a, hipStream_t st);
// depth sub-step 2 of the three-level model (hqtransformer.py:537-551): token i (raster (H1 H2 W1 W2)) =
// tok1[codes1[b, step, parent(i)]] + pos[i] (+ tok0[codes0[b, step]]: 'add', tok0 non-NULL), 16 rows per sample; tok1_ld = 4 D:
// the 'reduce' table, child (H2 W2) takes its D-slice of the parent's row
hipError_t launch_depth_embed_l2(const int64_t* codes0, const int64_t* codes1, int n_steps, const StepState* state, const float* tok0,
                                 const float* tok1, const float* pos, float* x, int B, int D, bf16_t* xpk, int pk_mb, float* parts,
                                 hipStream_t st, int V = 0, int tok1_ld = 0);
// 'top2mid2bot' head (hqtransformer.py:700-735): input of causal sub-step cnt >= 1 = tok[codes[(b * n_steps + step) * stride + slot]] + pos_row
hipError_t launch_depth_embed_causal(const int64_t* codes, int stride, int slot, int n_steps, const StepState* state, const float* tok,
                                     const float* pos_row, float* x, int B, int D, bf16_t* xpk, int pk_mb, float* parts, hipStream_t st, int V);
// raises the dynamic-LDS limit of the sampler for (V, top-p) outside any stream capture
hipError_t sampler_configure(int V, bool use_top_p);

hipError_t launch_advance_step(StepState* state, int d_tbase, hipStream_t st);
hipError_t launch_set_step(StepState* state, int step, int t_base, hipStream_t st);
// rows[b] = (seed, sample_offset + b) for b < B
hipError_t launch_set_rows(RowKey* rows, int B, uint64_t seed, int64_t sample_offset, hipStream_t st);

// int64 codes [B, n_steps(,4)] written by the sampler are final; this copies forced codes into the
// feed-back arrays when teacher forcing is on.
// ---- decoder
struct QuantArgs {
    const int64_t* code_t;       // grid [B, r/2, r/2] or seq [B, (r/2)^2]; NULL = zeros
    const int64_t* code_b;       // grid [B, r, r] or seq [B, (r/2)^2, 4]; NULL = zeros
    int seq_layout;              // 1: sampler layout (rearranges folded into the addressing)
    const float* emb_t;          // [n_embed, 4*E]
    const float* emb_b;          // [n_embed, E]
    void* quant;                 // NHWC [B, r, r, 2E]
    int B, r, E;
    int out_dtype;
    int n_embed;                 // codebook rows for the index clamp (0: unchecked)
};
hipError_t launch_quant_gather(const QuantArgs& a, hipStream_t st);
// Three-level additive pyramid (HQVAEGenerator.decode_code, generator.py:577-599): quant[b, Y, X, c] =
//   emb0[code_t[Y/4, X/4]][4 (4c + 2 (Y%2) + X%2) + 2 ((Y/2)%2) + (X/2)%2] + emb1[code_m[Y/2, X/2]][4c + 2 (Y%2) + X%2] + emb2[code_b[Y, X]][c]
// (two PixelShuffle(2) steps written out); NULL level = zeros; seq_layout = the sampler's [B, n], [B, n, 4], [B, n, 16].
struct QuantArgs3 {
    const int64_t *code_t, *code_m, *code_b;
    int seq_layout;
    const float *emb0, *emb1, *emb2;   // [n_embed, 16E], [n_embed, 4E], [n_embed, E]
    void* quant;                       // NHWC [B, r, r, E]
    int B, r, E;
    int out_dtype;
    int n_embed;
};
hipError_t launch_quant_gather3(const QuantArgs3& a, hipStream_t st);

// GroupNorm statistics over NHWC x: stats[b][g] = (mean, rstd)
hipError_t launch_gn_stats(const void* x, int dtype, float* stats, int B, int HW, int C, int groups, float eps,
                           hipStream_t st);

// in-place softmax over the last axis of fp32 [rows, n]
hipError_t launch_softmax_rows(void* x, int dtype, int rows, int n, hipStream_t st);

// FAST decoder: y = swish?(GroupNorm(x)) as its own bandwidth-bound pass (bf16 NHWC -> bf16 NHWC), so the
// convolution's operand loader stays a plain im2col gather.  stats: [B][groups][2] (mean, rstd).
hipError_t launch_gn_apply(const void* x, void* y, const float* stats, const float* gamma, const float* beta, int B, int HW,
                           int C, int groups, int swish, hipStream_t st);
// FAST decoder GroupNorm statistics: coalesced partial sums per (image, pixel chunk), then a finalize pass
hipError_t launch_gn_stats_fast(const void* x, float* stats, double* partial, int B, int HW, int C, int groups, float eps,
                                hipStream_t st, int dtype = DT_BF16);
size_t gn_stats_fast_partial_elems(int B, int HW, int C, int groups);
// statistics from the per-tile partials of a halo conv (fast_kernels.h: conv_halo_stats_ok)
hipError_t launch_gn_finalize_tiles(const float* partial, float* stats, int B, int tiles, int HW, int C, int groups, float eps, hipStream_t st);
// the same from the double partials a SPLIT conv leaves (split_kernels.h)
hipError_t launch_gn_finalize_tiles_d(const double* partial, float* stats, int B, int tiles, int HW, int C, int groups, float eps, hipStream_t st);

// ---- HQ-VAE encode side (generator.py:298-310, 530-568; quantizer.py:91-133)
// fp32 NCHW image [B, 3, R, R] -> NHWC [B, R, R, cpad] (channels >= 3 are zero) in the activation dtype
hipError_t launch_image_to_nhwc(const float* img, void* out, int out_dtype, int B, int R, int cpad, hipStream_t st);
// out[m] = sum_k rows[m][k]^2 (fp32, fixed order)
hipError_t launch_row_sumsq(const void* rows, int dtype, float* out, int M, int K, hipStream_t st);

// One residual-quantisation level.  Everything is kept in the bottom layout (NHWC fp32 [B, r, r, E]): h = quant_conv_b(encoder(x)),
// recon = the pixel-shuffled sum of the coarser levels' quants (NULL at level 0).  Level rows: m = (b, y, x) at resolution
// r >> k, column cq in [0, E * 4^k) <-> k nested pixel-unshuffles (cq = (...(c * 4 + 2 i1 + j1) * 4 + 2 i2 + j2)...).
struct VqArgs {
    const float* h;
    float* recon;
    int B, r, E, k;
    // rows pass
    void* z;                 // [M, dim] residual rows in `z_dtype` (the distance GEMM's A operand)
    int z_dtype;
    float* zz;               // [M] |z|^2 of the stored (rounded) rows
    float* resid_nchw;       // optional [B, dim, rq, rq] fp32: the quantiser's input (reference `resids` / code[2])
    // finish pass
    const unsigned long long* best;
    const float* emb;        // [n_embed, dim] fp32
    int64_t* codes;          // [B, rq, rq]
    float* quant_nchw;       // optional [B, dim, rq, rq]: z + (e - z)  (the straight-through value, quantizer.py:131)
    float* err_rows;         // [M] sum_c (e - z)^2
};
hipError_t launch_vq_rows(const VqArgs& a, hipStream_t st);
hipError_t launch_vq_finish(const VqArgs& a, hipStream_t st);
// diff[0] = scale * sum(err_rows[0..M)) in a fixed order (one workgroup)
hipError_t launch_vq_diff(const float* err_rows, int M, float scale, float* diff, hipStream_t st);
// fp32 NHWC [B, r, r, E] -> fp32 NCHW
hipError_t launch_nhwc_to_nchw_f32(const float* in, float* out, int B, int hw, int C, hipStream_t st);

