// SPLIT precision: fp32-accurate convolutions / GEMMs on the gfx950 matrix cores.
//
// Every fp32 operand x is carried as two fp16 values, hi = fp16(x) and lo = fp16((x - hi) * 2^11): hi + lo / 2^11
// reproduces x to 2^-22 relative (11 + 11 significand bits; the 2^11 scale keeps lo a NORMAL fp16 whenever hi is).
// Below |x| = 2^-14 (6.1e-5) hi is an fp16 subnormal: the conversions and the f16 MFMAs of gfx950 handle subnormals exactly (no
// flush), the split then carries fewer than 22 bits but its ABSOLUTE error stays below 2^-25 * 2^-11, far under the 1e-4 bar.  A product of two such operands is evaluated as
//     a w  ~=  a_hi w_hi  +  2^-11 (a_hi w_lo + a_lo w_hi)
// with three v_mfma_f32_32x32x16_f16 and two fp32 accumulators (main, cross); the dropped a_lo w_lo term is 2^-22
// relative.  a_hi w_hi is exact in fp32 (11 x 11 bits), so the result carries a few fp32 ulps of error per product --
// what an fp32 FMA chain in a different summation order also has -- at 3x the cost of one fp16/bf16 MFMA instead of the
// 16x of v_mfma_f32_32x32x2_f32.  Range: |x| must stay below 65504 (fp16); decoder activations are O(1..100).  The operand
// pass checks it: an activation outside the range is saturated and sets the handle's range flag (hqt_range_check, hqt.h).
#pragma once
#include "common.h"
#include "kernels.h"

typedef _Float16 half_t;

// fp32 [n] -> hi[n], lo[n] (finalize: conv filters)
hipError_t launch_split_f32(const float* src, half_t* hi, half_t* lo, size_t n, hipStream_t st);

// Operand pass of a SPLIT conv: y[b][p][0][c] = hi, y[b][p][1][c] = lo of  swish?(GroupNorm(x))  (stats == NULL: of x itself).
// x fp32 NHWC [B][HW][C]; y fp16 [B][HW][2][C]; stats [B][groups][2] (mean, rstd).  C % 8 == 0.
// range_flag (device int, may be NULL): ORed with 1 when an element is NaN or |element| >= 65504.
hipError_t launch_split_pack(const float* x, half_t* y, const float* stats, const float* gamma, const float* beta, int B, int HW,
                             int C, int groups, int swish, int* range_flag, hipStream_t st);

// SPLIT kernels read GemmArgs as the FAST ones do, with: A = split planes [pixel][2][Cin] (conv) or rows [m][2][K] (plain, lda = 2 K),
// Bw = hi filters [N][ldb], Bw_lo = lo filters, C / resid fp32.
bool split_conv3_ok(const GemmArgs& g);           // 3x3 stride-1 'same' conv (halo-tile kernel)
bool split_gemm_ok(const GemmArgs& g);            // 1x1 conv / plain GEMM (128 x 128 x 64 tiles)
hipError_t launch_split_conv3(const GemmArgs& g, hipStream_t st);
hipError_t launch_split_gemm(const GemmArgs& g, hipStream_t st);   // g.k_slices > 1 (at most split_gemm_slices(g), with g.k_slabs): K-sliced + the combine launch
int split_gemm_slices(const GemmArgs& g);         // 1 .. 4: slices worth taking for this shape (fp32-A rows out, few tiles)
int split_conv3_tiles_per_image(const GemmArgs& g);
hipError_t split_kernels_configure();             // raise the dynamic-LDS limits once (outside stream capture)

// ring kernels (split_stream_conv.hip): filters packed in MFMA fragment order (GemmArgs::Bw_frag16) and streamed straight into registers
size_t split_frag_elems(int N, int Cin);          // fp16 elements of the packed hi + lo fragments of an [N][9 Cin] filter bank (Cin % 32 == 0)
hipError_t launch_pack_split_frag16(const float* w_tapmajor, half_t* out, int N, int Cin, hipStream_t st);   // 16-channel blocks
// upsampling conv (nearest x2 + 3x3) as four 2x2 phase convolutions on the low-resolution image: pre-summed filters (GemmArgs::Bw_up16)
size_t split_up_elems(int N, int Cin);
hipError_t launch_pack_split_up16(const float* w_tapmajor, half_t* out, int N, int Cin, hipStream_t st);
bool split_stream_ok(const GemmArgs& g);          // g already passed split_conv3_ok
bool split_conv3_emits_planes(const GemmArgs& g); // g.out_split is honoured (conv3x3_split_ring16_kernel serves g)
int split_stream_tiles_per_image(const GemmArgs& g);
hipError_t launch_split_conv3_stream(const GemmArgs& g, hipStream_t st);
hipError_t split_stream_configure();
// norm_out -> swish -> conv_out (<= 4 channels, NCHW) in one fp32 kernel: g.A = the fp32 NHWC tensor, g.Bw = tap-major fp32 filters, g.gn_* set
bool conv_out_direct_ok(const GemmArgs& g);
hipError_t launch_conv_out_direct(const GemmArgs& g, hipStream_t st);
