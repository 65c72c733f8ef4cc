// EXACT-precision nn.Linear of the AR loop on the fp32 matrix instructions (round 4).
//
//   y[M, N] = x[M, K] W[N, K]^T (+bias, act, +resid; fused [query; key; value] split with the KV-cache row remap)      stage2/layers.py:73-85,190,313-315
//
// EXACT is the arithmetic whose code sequences are compared bit for bit with the reference's CPU path: fp32 operands, one fp32 fused
// multiply-add per product, fp32 accumulation.  v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 compute exactly that -- per output a
// k-ordered chain of fmaf, one rounding per product, no wider internal accumulator (MI355X_MICROARCH.md, 'FP32-input MFMA') -- at the
// fp32 vector rate, but without the 64 x 64 LDS tile of gemm_tile_kernel (gemm_generic.h), which leaves 72 workgroups for a 64-row
// GEMM of the body (0.16 TB/s of weights: 15 ms per top position at batch 64).  Here a WORKGROUP owns a (16 MT) x 16 output tile and
// its four waves a quarter of K each: no barrier until the final sum, a register ring of DEPTH 32-k steps in flight per wave.
//
// Summation order (what makes the result independent of the kernel).  K is cut into steps of 32 and the steps into four contiguous
// quarters (boundaries (steps x s) / 4); partial s chains its quarter ascending; element e of a lane's float4 feeds matrix instruction
// e of a 16-wide chunk, so inside a chunk the k indices enter the chain as [0 4 8 12 | 1 5 9 13 | 2 6 10 14 | 3 7 11 15]; the output is
// (p0 + p1) + (p2 + p3).  gemm_tile_kernel<..., QUARTERS> -- the vector-ALU kernel the same nn.Linear takes above 256 rows -- computes
// the same four chains in the same order, and a k-ordered fmaf chain is what both instructions are, so an output's BITS do not depend
// on the kernel its row count selects: a step's EXACT draws and logits are independent of the pass it is merged into
// (tests/test_gpu_timed_schedule.py compares the logits of a 64-row call with the same rows of a 512 / 2048-row pass bit for bit).
#include "gemm_generic.h"
#include "kernels.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// One workgroup = one 16 MT x 16 output tile, its four waves a quarter of K each, v_mfma_f32_16x16x4_f32.  Operand maps (cdna_hip_programming.md §3): lane l = r + 16 q supplies
// A[i = r][k = q] and B[k = q][j = r]; with A = W (i = column n of y) and B = x (j = row m of y) the result D has m on the lanes and four consecutive
// n per lane.  Per 32-k step a lane needs W[n0 + r][kb + 16 ch + 4 q .. + 3] and x[m0 + r][the same k] (ch = 0, 1), element e feeding MFMA e of the chunk.
//   * W comes from the fragment-ordered fp32 copy made at finalize ([n / 16][k / 32][chunk][lane][4]): one fully coalesced 1-KiB load per chunk.
//     (Read from the row-major tensor the same lanes touch 16 rows x 64 B per instruction -- adjacent lanes 6 KiB apart -- and the 64-row GEMMs
//     of the body ran at 0.7 TB/s of weights.)
//   * x is row-major [M][K] (it is what the LayerNorm / attention / GELU kernels write): fetched as 8 rows x 128 B per instruction (adjacent
//     lanes adjacent), turned into the fragment order through a wave-private LDS patch -- no workgroup barrier, LDS operations of a wave execute in order.
template <bool TILED, int MT>
__global__ __launch_bounds__(256) void exact_mfma_gemm_kernel(GemmArgs g, int TM, int TN) {
    constexpr int DEPTH = MT == 1 ? 6 : 4;               // 32-k steps in flight per wave (few waves per SIMD: memory latency is hidden by depth)
    // MT: 16-row tiles per wave (TM counts wave rows of 16 MT rows): one W fragment feeds MT independent accumulator chains -- the fp32
    // instruction's 40-cycle dependent latency against its 32-cycle issue disappears, and the W stream is fetched by 1 / MT as many waves
    constexpr int PITCH = 36;                            // floats per staged row: the 16 rows of a fragment read start in 16 different 4-bank groups
    __shared__ __attribute__((aligned(16))) float patch_all[4][MT * 16 * PITCH];
    __shared__ __attribute__((aligned(16))) float red_all[3][MT][64][4];       // partial tiles of waves 1 .. 3
    // The FOUR WAVES of a workgroup share one output tile and split its K: wave w chains the w-th contiguous quarter of the 32-k steps, ascending;
    // the four partial tiles meet in LDS and leave as (p0 + p1) + (p2 + p3) -- the summation order gemm_tile_kernel uses too (gemm_generic.h).
    // Chains are a quarter as long (fc2 at 64 rows: 1536 dependent matrix instructions per wave before), and four times the waves fill the chip.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Workgroup b runs on XCD b % 8 and every XCD has its own L2: the TM workgroups that read the same W tile (same tn) must sit on ONE XCD, or the tile crosses
    // the fabric TM times (fc2 at 64 rows, TM = 4: 151 MB instead of 38).  AR loop of one batch-64 step in EXACT: 190.8 -> 183.7 ms (profiles/r05_exact_xcd_order.txt).  XCD x takes the contiguous id range
    // [x total / 8, (x + 1) total / 8); ids of one tn are adjacent.
    int id = (int)blockIdx.x;
    const int total = (int)gridDim.x;
    if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
    const int tm = id % TM, tn = id / TM;
    const int r = lane & 15, q = lane >> 4;
    const int m0 = tm * 16 * MT, n0 = tn * 16;
    const int NST = g.K >> 5;
    float* const patch = patch_all[wave];
    // x: load t of a step covers rows 8 t .. 8 t + 7 of the wave's 16 MT, lane l -> row 8 t + (l >> 3), floats 4 (l & 7) .. + 3 of the step's 32
    const float* xsrc[2 * MT];
#pragma unroll
    for (int t = 0; t < 2 * MT; ++t) xsrc[t] = reinterpret_cast<const float*>(g.A) + (size_t)min(m0 + 8 * t + (lane >> 3), g.M - 1) * g.lda + 4 * (lane & 7);
    float* const pw = patch + (lane >> 3) * PITCH + 4 * (lane & 7);          // + 8 t rows
    const float* const pr = patch + r * PITCH + 4 * q;                       // + 16 mt rows, + 16 ch floats
    const float* wsrc = TILED ? reinterpret_cast<const float*>(g.Bw) + (size_t)tn * NST * 512 + lane * 4
                              : reinterpret_cast<const float*>(g.Bw) + (size_t)min(n0 + r, g.N - 1) * g.ldb + 4 * q;
    f32x4 wv[DEPTH][2], xg[DEPTH][2 * MT];
    auto fetch = [&](int c, int slot) {
        c = min(c, NST - 1);                             // past the end: a duplicate nobody multiplies (keeps the loads unconditional)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) wv[slot][ch] = *reinterpret_cast<const f32x4*>(wsrc + (TILED ? ((size_t)c * 2 + ch) * 256 : (size_t)c * 32 + 16 * ch));
#pragma unroll
        for (int t = 0; t < 2 * MT; ++t) xg[slot][t] = *reinterpret_cast<const f32x4*>(xsrc[t] + c * 32);
    };
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    auto multiply = [&](int slot) {
        // rows -> fragments through the wave's patch (in-order LDS: the reads below see the writes above, the next step's writes follow these reads)
#pragma unroll
        for (int t = 0; t < 2 * MT; ++t) *reinterpret_cast<f32x4*>(pw + 8 * t * PITCH) = xg[slot][t];
        // other lanes read what this lane wrote: say so to the compiler (both compile to nothing; the DS unit executes a wave's accesses in order)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f32x4 xf[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) xf[mt][ch] = *reinterpret_cast<const f32x4*>(pr + 16 * mt * PITCH + 16 * ch);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // ... and the next step's stores stay behind these loads
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)                   // the two 16-k chunks of the step, ascending; per output ONE chain, whatever MT
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[slot][ch][e], xf[mt][ch][e], acc[mt], 0, 0, 0);
    };
    // this wave's steps: the contiguous quarter [lo, lo + NW) of the NST; slot d holds step lo + c0 + d
    const int lo = (NST * wave) >> 2, NW = ((NST * (wave + 1)) >> 2) - lo;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) fetch(lo + d, d);
    int c0 = 0;
    for (; c0 + DEPTH <= NW; c0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            multiply(d);
            fetch(lo + c0 + d + DEPTH, d);
        }
    }
    // tail: NW % DEPTH steps, already in slots 0 .. (their loads were issued above)
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (c0 + d < NW) multiply(d);
    // (p0 + p1) + (p2 + p3)
    if (wave > 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f32x4*>(red_all[wave - 1][mt][lane]) = acc[mt];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const f32x4 p1 = *reinterpret_cast<const f32x4*>(red_all[0][mt][lane]), p2 = *reinterpret_cast<const f32x4*>(red_all[1][mt][lane]),
                    p3 = *reinterpret_cast<const f32x4*>(red_all[2][mt][lane]);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[mt][i] = (acc[mt][i] + p1[i]) + (p2[i] + p3[i]);
    }
    // D map: column = lane & 15 -> row m of y; register i of lane group q -> column n0 + 4 q + i
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + 16 * mt + r;
        if (m >= g.M) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + 4 * q + i;
            if (n < g.N) gemm_store<float>(g, 0, m, n, acc[mt][i]);
        }
    }
}

// plain fp32 row-major operands, K in whole chunks, the AR loop's store modes (gemm_store handles bias / act / resid / row remap / QKV split)
bool exact_mfma_small(const GemmArgs& g);
bool exact_mfma_ok(const GemmArgs& g) {
    static const bool off = getenv("HQT_NO_EXACT_MFMA") != nullptr;                // A/B switch: the 64 x 64 vector-ALU tile kernel for everything
    if (off || !g.k_quarters || g.conv_taps || g.a_packed_mb || g.a_rows_per_group || g.batch > 1 || g.gn_stats) return false;
    if (g.K % 32 != 0 || g.K < 32 || g.lda % 4 != 0 || g.ldb % 4 != 0) return false;
    return (g.store == STORE_ROWS || g.store == STORE_QKV) && exact_mfma_small(g);
}

// Small row counts only (the 64-row passes of one batch-64 step at a time and their 256-row depth sub-step): a 64-row GEMM of the body is 72
// workgroups for the 64 x 64 LDS tile of gemm_tile_kernel and 144-1152 waves here (960 -> 263 ms of AR loop per batch-64 step with one tile per
// wave).  Above 256 rows gemm_tile_kernel (the same summation order since round 4) moves fewer operand bytes per FLOP and wins (a 32 x 32-per-wave
// variant of this kernel measured 178 vs 155 ms per 64 images at 640 rows, 154 vs 119 at 2048: dropped).
bool exact_mfma_small(const GemmArgs& g) { return g.M <= 256; }

hipError_t launch_exact_mfma_gemm(const GemmArgs& g, hipStream_t st) {
    // Row tiles per wave: a wave's time is ~MT x (K / 16) matrix instructions, the launch takes ceil(4 waves x tiles / 4096 wave slots) such
    // rounds (four per SIMD at these register counts).  Ties go to the larger MT (the W stream is fetched by fewer workgroups).
    const int t16 = (g.M + 15) / 16, TN = (g.N + 15) / 16;
    int MT = 1;
    long long best = -1;
    for (int mt : {1, 2, 4}) {
        if (t16 % mt) continue;
        const long long waves = 4LL * (t16 / mt) * TN, cost = ((waves + 4095) / 4096) * mt * (mt == 1 ? 5 : 4);
        if (best < 0 || cost <= best) { best = cost; MT = mt; }
    }
    // 64 rows, measured per shape (tools/micro/bench_exact, cold weights; us for MT = 1 / 2 / 4): qkv 18.8 / 20.2 / 27.4, fc1 26.4 / 22.0 / 28.8, fc2 26.4 / 24.1 / 43.1,
    // proj 9.2 / 9.8 / 16.9 -- the model above picks MT = 2 for qkv and MT = 1 for fc2: a long K (48 steps per wave at MT = 1) wants two chains per wave, 4608 waves still fit
    if (t16 == 4) MT = (g.K >= 4096 || 4LL * t16 * TN > 5120) ? 2 : 1;
    const int TM = t16 / MT;
    const unsigned grid = (unsigned)((long long)TM * TN);
#define HQT_EXACT_LAUNCH(TILED, MT_) exact_mfma_gemm_kernel<TILED, MT_><<<grid, 256, 0, st>>>(g, TM, TN)
    if (g.b_tile16) { if (MT == 4) HQT_EXACT_LAUNCH(true, 4); else if (MT == 2) HQT_EXACT_LAUNCH(true, 2); else HQT_EXACT_LAUNCH(true, 1); }
    else { if (MT == 4) HQT_EXACT_LAUNCH(false, 4); else if (MT == 2) HQT_EXACT_LAUNCH(false, 2); else HQT_EXACT_LAUNCH(false, 1); }
#undef HQT_EXACT_LAUNCH
    return hipGetLastError();
}

// fp32 [N][K] -> fragment order [N / 16][K / 32][chunk 2][lane 64][4]: lane = r + 16 q holds W[16 tn + r][32 c + 16 ch + 4 q .. + 3]   (N % 16 == 0, K % 32 == 0)
__global__ void pack_exact_tiles_kernel(const float* __restrict__ w, float* __restrict__ out, int N, int K) {
    const size_t total = (size_t)N * K;
    const int NST = K >> 5;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 3), lane = (int)((i >> 2) & 63), ch = (int)((i >> 8) & 1);
        const size_t st = i >> 9;
        const int c = (int)(st % NST), tn = (int)(st / NST);
        out[i] = w[(size_t)(tn * 16 + (lane & 15)) * K + c * 32 + 16 * ch + 4 * (lane >> 4) + e];
    }
}
hipError_t launch_pack_exact_tiles(const float* w, float* out, int N, int K, hipStream_t st) {
    if (N % 16 != 0 || K % 32 != 0) return hipErrorInvalidValue;
    pack_exact_tiles_kernel<<<(unsigned)std::min<size_t>(((size_t)N * K + 255) / 256, 8192), 256, 0, st>>>(w, out, N, K);
    return hipGetLastError();
}
