// EXACT-precision nn.Linear of the AR loop on the fp32 matrix instructions (round 4).
//
//   y[M, N] = x[M, K] W[N, K]^T (+bias, act, +resid; fused [query; key; value] split with the KV-cache row remap)      stage2/layers.py:73-85,190,313-315
//
// EXACT is the arithmetic whose code sequences are compared bit for bit with the reference's CPU path: fp32 operands, one fp32 fused
// multiply-add per product, fp32 accumulation.  v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 compute exactly that -- per output a
// k-ordered chain of fmaf, one rounding per product, no wider internal accumulator (MI355X_MICROARCH.md, 'FP32-input MFMA') -- at the
// fp32 vector rate, but without the 64 x 64 LDS tile of gemm_tile_kernel (gemm_generic.h), which leaves 72 workgroups for a 64-row
// GEMM of the body (0.16 TB/s of weights: 15 ms per top position at batch 64).  Here a WAVE owns a T x T output tile (T = 16 or 32)
// and walks K alone: no LDS, no barrier, operands straight from L2 / HBM as 16-byte loads (lane (r, q) holds W[n0 + r][kb + 4 q ..]
// and x[m0 + r][kb + 4 q ..]), a register ring of DEPTH chunks in flight.
//
// Summation order.  Element e of a lane's float4 feeds MFMA step e, so inside a 16-wide chunk the k indices enter an output's chain in
// the order [0 4 8 12 | 1 5 9 13 | 2 6 10 14 | 3 7 11 15] (16x16x4: one instruction per group; 32x32x2: two, (0 4) then (8 12)), chunks
// ascending.  gemm_tile_kernel (the vector-ALU kernel larger row counts keep) walks its 16-wide chunks in that same order, and a
// k-ordered fmaf chain is what both compute, so an output's bits do not depend on the kernel the row count selects -- a step's EXACT
// draws AND logits stay independent of the pass it is merged into (tests/test_gpu_timed_schedule.py compares them bit for bit).
#include "gemm_generic.h"
#include "kernels.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <bool T32>
__global__ __launch_bounds__(256) void exact_mfma_gemm_kernel(GemmArgs g, int TM, int TN) {
    constexpr int T = T32 ? 32 : 16;
    constexpr int LPC = T32 ? 4 : 2;                     // float4 loads per operand and 32-k step (two 16-k chunks: a lane group reads whole 128-byte lines)
    constexpr int DEPTH = T32 ? 3 : 6;                   // 32-k steps in flight per wave (one wave per SIMD hides memory latency by depth, not by occupancy)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tile = (long long)blockIdx.x * 4 + wave;
    if (tile >= (long long)TM * TN) return;              // whole waves leave; the kernel has no barrier
    const int tm = (int)(tile % TM), tn = (int)(tile / TM);       // the waves of a workgroup: neighbouring row tiles of ONE column tile (W rows shared through the L1)
    const int r = lane & (T - 1), q = lane / T;
    const int m0 = tm * T, n0 = tn * T;
    const float* xrow = reinterpret_cast<const float*>(g.A) + (size_t)min(m0 + r, g.M - 1) * g.lda + 4 * q;
    const int NST = g.K >> 5;
    // W: row-major [N][K], or (b_tile16, 16 x 16 tiles) the tile-contiguous copy: a wave's 16 rows of a 32-k step are ONE 2-KiB run and its
    // whole stream one contiguous 2 NST KiB -- row-major, 4608 row streams advance 128 B at a time each and DRAM sees scattered lines
    // (0.7 TB/s of weights measured at 64 rows); the values and their order in the chain are the same either way
    const bool tiled = !T32 && g.b_tile16;
    const float* wrow = tiled ? reinterpret_cast<const float*>(g.Bw) + ((size_t)tn * NST * 16 + r) * 32 + 4 * q
                              : reinterpret_cast<const float*>(g.Bw) + (size_t)min(n0 + r, g.N - 1) * g.ldb + 4 * q;
    const int wstep = tiled ? 16 * 32 : 32;
    f32x4 wv[DEPTH][LPC], xv[DEPTH][LPC];
    // load t of a step: chunk t / (LPC / 2), half t % (LPC / 2): offsets 0, 16 (16x16) / 0, 8, 16, 24 (32x32) floats -- issued back to back per operand,
    // so the two halves of a row's 128-byte line are requested together
    auto fetch = [&](int c, int slot) {
        c = min(c, NST - 1);                             // past the end: a duplicate nobody multiplies (keeps the loads unconditional)
#pragma unroll
        for (int t = 0; t < LPC; ++t) wv[slot][t] = *reinterpret_cast<const f32x4*>(wrow + (size_t)c * wstep + (T32 ? 8 : 16) * t);
#pragma unroll
        for (int t = 0; t < LPC; ++t) xv[slot][t] = *reinterpret_cast<const f32x4*>(xrow + c * 32 + (T32 ? 8 : 16) * t);
    };
    typename std::conditional<T32, f32x16, f32x4>::type acc;
#pragma unroll
    for (int i = 0; i < (T32 ? 16 : 4); ++i) acc[i] = 0.0f;
    auto multiply = [&](int slot) {
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)                   // the two 16-k chunks of the step, ascending
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < LPC / 2; ++t) {
                    const int u = ch * (LPC / 2) + t;
                    if constexpr (T32) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[slot][u][e], xv[slot][u][e], acc, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[slot][u][e], xv[slot][u][e], acc, 0, 0, 0);
                }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) fetch(d, d);
    int c0 = 0;
    for (; c0 + DEPTH <= NST; c0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            multiply(d);
            fetch(c0 + d + DEPTH, d);
        }
    }
    // tail: NST % DEPTH steps, already in slots 0 .. (their loads were issued above)
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (c0 + d < NST) multiply(d);
    // D map: column = lane & (T - 1) -> row m of y; rows -> columns n of y (32x32: (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5); 16x16: 4 (lane >> 4) + reg)
    const int m = m0 + r;
    if (m >= g.M) return;
#pragma unroll
    for (int i = 0; i < (T32 ? 16 : 4); ++i) {
        const int n = n0 + (T32 ? (i & 3) + 8 * (i >> 2) + 4 * q : 4 * q + i);
        if (n < g.N) gemm_store<float>(g, 0, m, n, acc[i]);
    }
}

// plain fp32 row-major operands, K in whole chunks, the AR loop's store modes (gemm_store handles bias / act / resid / row remap / QKV split)
bool exact_mfma_small(const GemmArgs& g);
bool exact_mfma_ok(const GemmArgs& g) {
    static const bool off = getenv("HQT_NO_EXACT_MFMA") != nullptr;                // A/B switch: the 64 x 64 vector-ALU tile kernel for everything
    if (off || g.conv_taps || g.a_packed_mb || g.a_rows_per_group || g.batch > 1 || g.gn_stats) return false;
    if (g.K % 32 != 0 || g.K < 32 || g.lda % 4 != 0 || g.ldb % 4 != 0) return false;
    return (g.store == STORE_ROWS || g.store == STORE_QKV) && exact_mfma_small(g);
}

// Small row counts only: one 16 x 16 tile per wave puts 4 x the waves of a 32 x 32 tiling on the chip, which is what a 64-row GEMM
// needs (288 workgroups instead of 72: 960 -> 397 ms of AR loop per batch-64 step).  From ~1024 tiles of 32 x 32 the LDS-shared
// 64 x 64 tile of gemm_tile_kernel (the same summation order since round 4) moves half the operand bytes per FLOP and wins
// (a 32 x 32-per-wave variant of this kernel measured 178 vs 155 ms per 64 images at 640 rows, 154 vs 119 at 2048: dropped).
bool exact_mfma_small(const GemmArgs& g) { return (long long)((g.M + 31) / 32) * ((g.N + 31) / 32) < 1024; }

hipError_t launch_exact_mfma_gemm(const GemmArgs& g, hipStream_t st) {
    const int TM = (g.M + 15) / 16, TN = (g.N + 15) / 16;
    exact_mfma_gemm_kernel<false><<<(unsigned)(((long long)TM * TN + 3) / 4), 256, 0, st>>>(g, TM, TN);
    return hipGetLastError();
}

// fp32 [N][K] -> [N / 16][K / 32][16][32] (N % 16 == 0, K % 32 == 0)
__global__ void pack_exact_tiles_kernel(const float* __restrict__ w, float* __restrict__ out, int N, int K) {
    const size_t total = (size_t)N * K;
    const int NST = K >> 5;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int kk = (int)(i & 31), rr = (int)((i >> 5) & 15);
        const size_t st = i >> 9;
        const int c = (int)(st % NST), tn = (int)(st / NST);
        out[i] = w[(size_t)(tn * 16 + rr) * K + c * 32 + kk];
    }
}
hipError_t launch_pack_exact_tiles(const float* w, float* out, int N, int K, hipStream_t st) {
    if (N % 16 != 0 || K % 32 != 0) return hipErrorInvalidValue;
    pack_exact_tiles_kernel<<<(unsigned)std::min<size_t>(((size_t)N * K + 255) / 256, 8192), 256, 0, st>>>(w, out, N, K);
    return hipGetLastError();
}
