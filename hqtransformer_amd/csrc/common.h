// Shared declarations of libhqt.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

typedef uint16_t bf16_t;   // raw bfloat16 bits

__device__ __host__ inline float bf16_to_f32(bf16_t v) {
    union { uint32_t u; float f; } c;
    c.u = ((uint32_t)v) << 16;
    return c.f;
}
// round-to-nearest-even; NaN stays NaN.  On the device this is gfx950's v_cvt_pk_bf16_f32 (same rounding; the
// branchy software form cost more than the LDS traffic in the conv epilogues).
__device__ __host__ inline bf16_t f32_to_bf16(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(bf16_t, (__bf16)f);
#endif
    union { uint32_t u; float f; } c;
    c.f = f;
    if ((c.u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((c.u >> 16) | 0x40);
    return (bf16_t)((c.u + 0x7fffu + ((c.u >> 16) & 1u)) >> 16);
}

enum { ACT_NONE = 0, ACT_GELU_ERF = 1, ACT_GELU_SIGMOID = 2 };
enum { STORE_ROWS = 0, STORE_NCHW = 1, STORE_QKV = 2, STORE_PACKED = 3, STORE_RESID = 4, STORE_ARGMIN = 5 };

// filter edge of an implicit-GEMM convolution from its tap count: 1x1, 3x3, 4x4 (the encoder's strided conv_in)
__device__ __host__ inline int conv_ks(int taps) { return taps == 9 ? 3 : (taps == 16 ? 4 : 1); }
// total order on floats as unsigned integers (smaller float <-> smaller key); NaN sorts last
__device__ __host__ inline uint32_t float_order_key(float f) {
    union { float f; uint32_t u; } c;
    c.f = f;
    return (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);
}

// MFMA-fragment-packed activation layout of the AR loop's GEMM A operands (FAST precision):
// element (row m, column k) of an [Mpad = 32*MB, K] matrix lives at packed_off(m, k, MB), i.e.
// [k/16][m/32][lane = (k%16)/8*32 + m%32][k%8] -- one 1-KiB wave load is the B-operand fragment of
// v_mfma_f32_32x32x16_bf16 for (k-step, m-block).
__device__ __host__ inline long long packed_off(int m, int k, int MB) {
    return ((long long)(k >> 4) * MB + (m >> 5)) * 512 + ((((k >> 3) & 1) << 5) + (m & 31)) * 8 + (k & 7);
}
constexpr int PACKED_MAX_ROWS = 16384;     // most rows of a packed_off() activation (a merged pass: 4 depth tokens x up to 4096 samples)
__device__ __host__ inline int packed_mb(int M) {          // 32-row blocks of the packed layout for M rows: the next power of two (1 .. 512)
    if (M > PACKED_MAX_ROWS) return 0;
    int mb = 1;
    while (mb * 32 < M) mb <<= 1;
    return mb;
}

// One NT GEMM  C[b][m][n] = alpha * sum_k A[b][m][k] * B[b][n][k]  (+bias, act, +residual) with
//   * an optional implicit-GEMM A operand: 1x1 / 3x3 'same' convolution over an NHWC tensor, with
//     nearest-x2 upsampling folded into the addressing and GroupNorm(+swish) applied on load;
//   * an output row remap (KV-cache append), a per-image transposed store (NCHW / V^T), or a
//     three-way column split (fused QKV).
struct ChainSync {
    const unsigned* wait;    // NULL: no predecessor to wait for
    unsigned target;
    unsigned* signal;        // NULL: nobody waits for this kernel
    unsigned* err;
};

// Called by every thread of the workgroup.  Consumer side of the hand-off: one lane polls (relaxed, agent scope), fences
// once (acquire, agent scope: drops this CU's L1 and the XCD L2's stale lines), then the workgroup barrier.
#define HQT_CHAIN_SPIN_LIMIT (1u << 17)
__device__ inline unsigned chain_poll(const ChainSync& c) {      // issued before the independent loads; consumed by chain_wait
    return c.wait ? __hip_atomic_load(c.wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
}
__device__ inline void chain_wait(const ChainSync& c, unsigned polled) {
    if (!c.wait) return;
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (polled < c.target) {
            if (++spins > HQT_CHAIN_SPIN_LIMIT) { if (c.err) *c.err = 1u; break; }
            __builtin_amdgcn_s_sleep(1);
            polled = __hip_atomic_load(c.wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}
// Producer side: every wave drains its stores, workgroup barrier, one lane releases at agent scope and arrives.
__device__ inline void chain_signal(const ChainSync& c) {
    if (!c.signal) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(c.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

struct GemmArgs {
    // ---- A
    const void* A;           // plain: [M, lda]; conv: NHWC [batch_img, Hin, Win, Cin]
    int lda;
    long long a_batch_stride;  // elements, blockIdx.z batches (decoder attention)
    int a_rows_per_group;    // plain A row remap: a_row = (m / rpg) * a_group_stride + m % rpg + a_row_offset
    int a_group_stride;
    int a_row_offset;
    int a_packed_mb;         // > 0: A is in the packed_off() layout with this many 32-row blocks
    int conv_taps;           // 0 = plain, 1 = 1x1, 9 = 3x3, 16 = 4x4
    int conv_stride2;        // 1: stride 2 (the input is 2H x 2W): Downsample / the encoder's conv_in
    int conv_nopad;          // 1: no zero padding at the top / left (Downsample pads bottom / right only: layers.py:68-72)
    int H, W, Cin;           // output spatial size and input channels (conv)
    int upsample;            // 1: input is (H/2, W/2), nearest x2 before the conv
    const float* gn_stats;   // [batch_img][groups][2] mean, rstd  (NULL: no GroupNorm on load)
    const float* gn_gamma;
    const float* gn_beta;
    int gn_groups;
    int gn_swish;
    // ---- B (weights or second activation), [N, ldb]
    const void* Bw;
    int ldb;
    long long b_batch_stride;
    // ---- C
    void* C;
    int ldc;
    long long c_batch_stride;
    int M, N, K, batch;
    const float* bias;       // [N] or NULL
    const void* resid;       // same indexing as C (may alias C), or NULL
    int act;
    float alpha;
    int store;               // STORE_*
    int rows_per_group;      // STORE_ROWS: out_row = (m / rpg) * group_stride + m % rpg + row_offset
    int group_stride;
    int row_offset;
    const int* row_offset_dev;   // optional device int added to row_offset (graph replays)
    int rows_per_image;      // STORE_NCHW: C[(img * N + n) * rpi + m % rpi]
    int clamp01;             // STORE_NCHW epilogue clamp(0.5 x + 0.5, 0, 1)
    // STORE_QKV: columns [0,D) -> C, [D,2D) -> C2, [2D,3D) -> C3 ; C2/C3 use the row remap, C plain
    void* C2;
    void* C3;
    int qkv_D;
    int qkv_first;           // STORE_QKV: part index of column 0 (1: the GEMM computes only [key; value], N = 2 D)
    bf16_t* qkv_v_pk;        // STORE_QKV: optional second copy of the value part in the packed_off() layout (c_packed_mb):
                             //   with a single key the attention output IS the value row (depth sub-step 0)
    int c_packed_mb;         // STORE_PACKED (and the q part of STORE_QKV stays row-major)
    const void* zero_page;   // >= 16 zero bytes in device memory (source of padded im2col taps for LDS-DMA loads)
    // ---- deferred LayerNorm on the A operand (FAST AR loop).  A holds bf16(x), not LN(x); with W' = gamma o W,
    //      LN(x) W^T + b = rstd_m (x W'^T - mean_m colsum(W')) + (b + W beta), so the normalisation becomes an epilogue.
    //      Row statistics arrive as partial (sum, sum of squares) over column tiles: ln_parts[p][32 a_packed_mb][2].
    const float* ln_parts;
    int ln_nparts;
    const float* ln_colsum;  // [N] = sum_k bf16(W'[n][k])
    float ln_eps;
    // ---- STORE_RESID: C (fp32 [M, ldc]) += value + bias; the bf16 copy of the new row goes to resid_pk in the
    //      packed_off() layout (c_packed_mb) and the partial statistics of that copy to resid_parts as
    //      [n_tile][32 c_packed_mb][2].  (Dedicated fields: reusing C2/C3 here made hipcc drop the STORE_QKV stores
    //      through C3 in the same kernel.)
    bf16_t* resid_pk;
    float* resid_parts;
    unsigned* tile_ctr;      // split-K residual producers finished inside the launch (tile_gemm.hip: TS_FUSED): one arrival counter per output tile, zero between launches
    // ---- STORE_ARGMIN (nearest-code search, quantizer.py:99-103): d = (am_rownorm[m] + am_colnorm[n]) - 2 acc; the winner per row is
    //      kept as atomicMin over (float_order_key(d) << 32 | n) in am_best[m] (ties -> lowest n, as torch.argmin on equal values)
    const float* am_rownorm;
    const float* am_colnorm;
    unsigned long long* am_best;
    // ---- chained launches (AR loop on two alternating streams): consecutive kernels may be co-resident; a kernel issues
    //      the loads that do not depend on its predecessor (weights), then waits until chain_wait[0] >= chain_target
    //      (= the predecessor's workgroup count; every workgroup adds 1 to its chain_signal after its last store, behind
    //      an agent-scope release).  Spins are bounded: on give-up chain_err[0] is set and the results are garbage.
    ChainSync chain;
    // ---- GroupNorm statistics of the OUTPUT, fused into the 3x3 halo conv's epilogue: per-tile partial (sum, sum of
    //      squares) of the bf16-rounded outputs per group, gn_part_out[((img * tiles_per_img + tile) * gn_out_groups + group) * 2];
    //      a fixed-order finalize turns them into (mean, rstd).  NULL: not requested.
    float* gn_part_out;
    int gn_out_groups;
    int tune;                // 0: latency-oriented tile choice, 1: throughput-oriented (hqt_set_policy)
    // ---- SPLIT precision (split_kernels.h): A holds fp16 [row][hi K | lo K] planes, Bw the hi filters, Bw_lo the lo filters;
    //      C / resid are fp32.  gn_part_out_d: per-tile GroupNorm partials of the fp32 output as doubles (layout of gn_part_out).
    const void* Bw_lo;
    const void* Bw_frag16;   // optional: the same 3x3 filters packed in MFMA fragment order for v_mfma_f32_16x16x32_f16 (16-channel blocks; conv3x3_split_ring16_kernel)
    const void* Bw_up16;     // optional, upsampling convs: the four 2x2 phase filters (pre-summed taps), packed like Bw_frag16 (conv2x2_split_up16_kernel)
    double* gn_part_out_d;
    int a_f32;               // split_gemm_kernel only: A is the fp32 tensor itself ([row][K], lda floats); the hi / lo split happens while the tile is staged
    int b_f32;               // split_gemm_kernel only, with a_f32: Bw is an fp32 tensor too ([n][K], ldb floats; Bw_lo unused)
    int* range_flag;         // a_f32 / b_f32: where an element outside the fp16 range is reported (hqt_range_check)
    int out_split;           // conv3x3_split_ring16_kernel: store the output as fp16 hi / lo operand planes [pixel][2][N] (range-checked: range_flag) instead of the
                             // fp32 tensor -- the consumer is another SPLIT conv with nothing in between (resblock -> upsampling conv): no operand pass
    int k_quarters;          // fp32 nn.Linear of the AR loop (set by run_linear for its gemm_* launches): the four-quarter summation order shared by
                             //   exact_mfma_gemm_kernel (<= 256 rows) and gemm_tile_kernel<..., QUARTERS>; every other fp32 GEMM keeps one chain per output
    int b_tile16;            // exact_mfma_gemm_kernel only: Bw is the fragment-ordered fp32 copy [n / 16][k / 32][chunk][lane][4] (pack_exact_tiles_kernel)
    int k_slices;            // split_gemm_kernel<fp32 A> only (plain fp32 rows out, one batch): > 1 = blockIdx.z takes a slice of K and leaves its raw partial as
    float* k_slabs;          //   fp32 [slice][M][N] in k_slabs; split_rows_combine_kernel adds the slices in index order, the bias and the residual (launch_split_gemm)
    int tile_panel;          // workgroup-order experiments.  SPLIT 3x3 convs: pixel tiles per panel (0: the n-tile runs fastest; measured: no effect,
                             // profiles/r04_conv_tile_order.txt)
};

struct StepState {           // lives in device memory; lets one captured graph serve every position AND every call
    int step;                // current top position (0-based)
    int t_base;              // KV rows already in the body cache
};
struct RowKey {              // per batch row, device memory owned by the handle: the Philox key of the row's call and its global index
    unsigned long long seed;
    long long global_row;
};

#define HQT_MAX_V 16384

// Table indices that arrive from the caller (class / text ids, teacher-forced codes, code grids) are clamped into the table on
// the device: an out-of-range id can then never fault or read foreign memory.  The Python surface raises IndexError for them
// before the launch, as nn.Embedding / F.embedding do in the reference; n <= 0 disables the clamp.
__device__ __forceinline__ long long clamp_idx(long long v, int n) { return n > 0 ? (v < 0 ? 0 : (v >= n ? (long long)n - 1 : v)) : v; }
