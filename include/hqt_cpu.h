/*
 * hqt_cpu.h -- the hqt_cpu_* twins of hqt_sample / hqt_decode: the SAME call surface computed on
 * the host cores in fp32 (C++ / OpenMP, own blocked GEMM; oracle/hqt_cpu.cpp -> oracle/_build/libhqt_cpu.so).
 *
 * This is TEST INFRASTRUCTURE and the measured CPU baseline, never a fallback: it is NOT part of libhqt.so, nothing under
 * hqtransformer_amd/ loads it, and the product keeps refusing CPU devices.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg bind it (oracle/hqt_cpu.py).  It restates what the reference computes on its CPU path, where autocast is off
 * and every activation is fp32: sampling_ihqgpt / iHQGPT.sampling_step (hqvae/utils/sampling.py:12-37,164-237;
 * hqvae/models/stage2/hierarchical_ar.py:428-563,667-789; stage2/layers.py:14-23,61-195,290-375) and
 * SimRQGAN2Generator.decode_code + Decoder.forward (stage1/generator.py:312-367; quantizer.py:179-186;
 * stage1/modules/layers.py:12-53,78-186,385-410), and is pinned by the reference-generated fixtures G3 / G4 / G5
 * (tests/test_cpu_twin.py).  All pointers are HOST pointers; calls are synchronous; status codes and structs are those of hqt.h.
 * Two code levels only (hqt_config.code_levels 0 / 2).
 */
#ifndef HQT_CPU_H
#define HQT_CPU_H

#include "hqt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hqt_cpu_handle hqt_cpu_handle;

/* n_threads <= 0: every hardware thread OpenMP reports.  The count is fixed per handle: the weights are first-touched by the
 * threads that will stream them (NUMA placement), in the static partition the GEMMs use. */
int hqt_cpu_create(const hqt_config* cfg, int n_threads, hqt_cpu_handle** out);
/* reference state-dict key with its 'stage1.' / 'stage2.' prefix, contiguous fp32 in the reference's own layout (copied) */
int hqt_cpu_set_weight(hqt_cpu_handle* h, const char* name, const float* data, const int64_t* shape, int ndim);
int hqt_cpu_finalize_weights(hqt_cpu_handle* h);
/* twin of hqt_sample (EXACT arithmetic only; opts->precision / use_graph are ignored): cond, noise [n_steps, 5, B, V] (NULL: the
 * Philox stream of libhqt's sampler, keyed by opts->seed / sample_offset / row_seeds / row_offsets), force_top / force_bot,
 * logits_out [n_steps, 5, B, V] (optional), out_top [B, n_steps], out_bot [B, n_steps, 4] */
int hqt_cpu_sample(hqt_cpu_handle* h, int B, const int64_t* cond, const hqt_sample_opts* opts, const float* noise,
                   const int64_t* force_top, const int64_t* force_bot, float* logits_out, int64_t* out_top, int64_t* out_bot);
/* twins of hqt_decode / hqt_decode_seq (fp32): code grids [B, r/2, r/2] / [B, r, r] (either may be NULL), or the sampler's
 * [B, (r/2)^2] / [B, (r/2)^2, 4]; out_pixels fp32 [B, out_ch, H, W] NCHW */
int hqt_cpu_decode(hqt_cpu_handle* h, int B, const int64_t* code_t, const int64_t* code_b, float* out_pixels, int clamp01);
int hqt_cpu_decode_seq(hqt_cpu_handle* h, int B, const int64_t* codes_top, const int64_t* codes_bot, float* out_pixels, int clamp01);
/* wall seconds the last hqt_cpu_sample / hqt_cpu_decode[_seq] call spent inside the library */
double hqt_cpu_last_seconds(const hqt_cpu_handle* h);
int hqt_cpu_threads(const hqt_cpu_handle* h);
/* "avx512" | "avx2": the GEMM micro-kernel picked for this host */
const char* hqt_cpu_isa(void);
int hqt_cpu_destroy(hqt_cpu_handle* h);
const char* hqt_cpu_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* HQT_CPU_H */
