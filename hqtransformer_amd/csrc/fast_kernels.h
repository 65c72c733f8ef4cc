// FAST-precision kernels: bf16 MFMA GEMMs (weight-streaming skinny GEMM for the AR loop, tiled
// implicit-GEMM for the decoder) plus the weight repack/convert passes run at finalize.
#pragma once
#include "common.h"
#include "kernels.h"

hipError_t launch_f32_to_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t st);
// [O][I][taps] -> [O][taps][I]
hipError_t launch_repack_conv(const float* src, float* dst, int O, int I, int taps, hipStream_t st, int Ipad = 0);

// ---- weight-streaming GEMM (AR loop, M = B or 4B rows): y[M,N] = x[M,K] W[N,K]^T, W pre-packed
bool stream_gemm_supported(int N, int K);
hipError_t launch_pack_stream_weights(const float* w32, bf16_t* packed, int N, int K, hipStream_t st);
// deferred LayerNorm: packed bf16 of gamma o W, its column sums and the folded bias b + W beta (see GemmArgs::ln_parts)
hipError_t launch_fold_layernorm(const float* w32, const float* gamma, const float* beta, const float* bias, float* wfold_tmp,
                                 bf16_t* packed, float* colsum, float* bias_out, int N, int K, hipStream_t st);
bool stream_gemm_ok(const GemmArgs& g, int a_dt, int c_dt);
hipError_t stream_gemm_configure();     // raise dynamic-LDS limits once (call outside stream capture)
// S = stream_gemm_splitk(g) (> 1 only when the caller can defer bias + residual to the next LayerNorm):
// slabs [S][32 * a_packed_mb][N] fp32 receive the partial sums and no epilogue runs.
int stream_gemm_splitk(const GemmArgs& g);
hipError_t launch_stream_gemm(const GemmArgs& g, const bf16_t* wpk, int a_dt, int c_dt, int S, float* slabs, hipStream_t st);

// ---- LDS-tiled MFMA GEMM of the merged AR passes (tile_gemm.hip): same operands and store modes as the streaming GEMM, 512+ rows.
// S > 1 (STORE_RESID only): fp32 partial slabs [S][32 * a_packed_mb][N], finished by launch_resid_combine (x += bias + sum of
// slabs, bf16 packed copy, whole-row statistics as ONE part).
struct TilePlan { int geom; int bm, bn; int S; };    // geom < 0: not taken; S: split-K factor
constexpr int TILE_CTR_MAX = 4096;          // output tiles of one TS_FUSED launch (GemmArgs.tile_ctr; tools/micro/bench_tile only: measured and lost, tile_gemm.hip)
bool tile_gemm_ok(const GemmArgs& g, int a_dt, int c_dt);
TilePlan tile_gemm_plan(const GemmArgs& g);
hipError_t launch_tile_gemm(const GemmArgs& g, const bf16_t* wpk, int a_dt, int c_dt, const TilePlan& p, float* slabs, hipStream_t st);
hipError_t launch_resid_combine(const GemmArgs& g, const float* slabs, int S, hipStream_t st);
hipError_t tile_gemm_configure();       // raise dynamic-LDS limits once (call outside stream capture)

// ---- tiled MFMA GEMM / implicit-GEMM conv (decoder, text prefill)
bool mfma_gemm_ok(const GemmArgs& g, int a_dt, int b_dt, int c_dt);
hipError_t launch_mfma_gemm(const GemmArgs& g, int a_dt, int b_dt, int c_dt, hipStream_t st);
// 3x3 halo-tile conv: will launch_mfma_gemm take it for g; can it also emit per-tile GroupNorm statistics of its output
// (GemmArgs::gn_part_out, [image][tile][group][2] floats); tiles per image
bool conv_halo_ok(const GemmArgs& g, int c_dt);
bool conv_halo_stats_ok(int N, int groups);
int conv_halo_tiles_per_image(const GemmArgs& g);
hipError_t mfma_gemm_configure();
