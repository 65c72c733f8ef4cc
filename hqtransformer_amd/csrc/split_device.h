// SPLIT precision: the device-side hi / lo split shared by the operand pass, the GEMM / conv kernels that split while they stage a tile
// (split_conv.hip) and the conv epilogue that emits operand planes (split_stream_conv.hip).  See split_kernels.h for the arithmetic.
#pragma once
#include "split_kernels.h"

namespace {
constexpr float SPLIT_SCALE = 2048.0f, SPLIT_INV = 1.0f / 2048.0f;

__device__ __forceinline__ void split2(float x, half_t& hi, half_t& lo) {
    hi = (half_t)x;                                   // round to nearest even
    lo = (half_t)((x - (float)hi) * SPLIT_SCALE);     // the difference and the scaling are exact in fp32
}
// Activations: the same split with the fp16 range checked.  |x| >= 65504 (or NaN) cannot travel as fp16 planes: it is replaced by a
// finite saturated value (so that one bad element does not poison its whole receptive field with NaNs) and reported through `bad`,
// which the operand pass ORs into the handle's range flag -- hqt_range_check() turns it into HQT_ERR_RANGE.
__device__ __forceinline__ void split2_checked(float x, half_t& hi, half_t& lo, bool& bad) {
    if (!(fabsf(x) < 65504.0f)) { bad = true; x = x > 0.0f ? 65472.0f : (x < 0.0f ? -65472.0f : 0.0f); }
    split2(x, hi, lo);
}
// Eight fp32 values -> their hi / lo planes (16 B each), for the kernels that split an operand between its global load and its LDS write:
// the same roundings as split2 (bit-identical planes for values inside the range), on pairs -- v_med3 saturation, v_cvt_pk_f16_f32 and
// packed fp32 arithmetic: ~6 vector instructions per element instead of the ~13 of eight scalar split2_checked calls (which made the
// fp32-operand GEMM 15-40 % slower per launch than the packed-operand one).  Out of range: saturated to +-65472 (NaN: -65472) and flagged.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8_checked(const float (&x)[8], unsigned (&hi)[4], unsigned (&lo)[4], bool& bad) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        f32x2_t v = {x[2 * p], x[2 * p + 1]};
        bad |= !(fabsf(v.x) < 65504.0f) | !(fabsf(v.y) < 65504.0f);
        v.x = __builtin_amdgcn_fmed3f(v.x, -65472.0f, 65472.0f);
        v.y = __builtin_amdgcn_fmed3f(v.y, -65472.0f, 65472.0f);
        const half2_t h = __builtin_convertvector(v, half2_t);                       // round to nearest even
        const f32x2_t r = (v - __builtin_convertvector(h, f32x2_t)) * SPLIT_SCALE;   // exact in fp32
        const half2_t l = __builtin_convertvector(r, half2_t);
        hi[p] = __builtin_bit_cast(unsigned, h);
        lo[p] = __builtin_bit_cast(unsigned, l);
    }
}
}  // namespace
