// FAST-precision kernels (bf16 operands, fp32 accumulation on the gfx950 matrix cores).
#include "fast_kernels.h"
#include <cstdlib>
#include <type_traits>
#include "gemm_generic.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// ---------------------------------------------------------------------------------------------
// finalize-time repacks
// ---------------------------------------------------------------------------------------------
__global__ void f32_to_bf16_kernel(const float* src, bf16_t* dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = f32_to_bf16(src[i]);
}
hipError_t launch_f32_to_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t st) {
    const int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
    f32_to_bf16_kernel<<<grid, 256, 0, st>>>(src, dst, n);
    return hipGetLastError();
}

__global__ void repack_conv_kernel(const float* src, float* dst, int O, int I, int taps, int Ipad) {
    const size_t n = (size_t)O * Ipad * taps;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ii = (int)(i % Ipad);
        const int t = (int)((i / Ipad) % taps);
        const size_t o = i / ((size_t)Ipad * taps);
        dst[i] = ii < I ? src[(o * I + ii) * taps + t] : 0.0f;          // channels I..Ipad-1: zero filters for a zero-padded input
    }
}
hipError_t launch_repack_conv(const float* src, float* dst, int O, int I, int taps, hipStream_t st, int Ipad) {
    if (Ipad < I) Ipad = I;
    const size_t n = (size_t)O * Ipad * taps;
    const int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
    repack_conv_kernel<<<grid, 256, 0, st>>>(src, dst, O, I, taps, Ipad);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// K2/K4/K5: weight-streaming GEMM of the AR loop.
//
//   y[M, N] = x[M, K] W[N, K]^T,   M = B or 4B <= 256 rows, W streamed from HBM exactly once.
//
// Weights are packed at finalize into 1-KiB chunks, chunk (n-tile nt of 32 rows, k-step ks of 16) =
// the A-operand fragment of v_mfma_f32_32x32x16_bf16: lane l holds W[32 nt + (l & 31)][16 ks + 8 (l >> 5) + j],
// j = 0..7.  Chunks of one n-tile are contiguous along k, so a wave streams a contiguous run with
// one fully coalesced global_load_dwordx4 per chunk, straight into VGPRs (no LDS round trip: each
// weight byte is used by exactly one wave).  Activations arrive in the packed_off() layout (written
// by the producing LayerNorm / attention / GELU epilogue), so the B-operand fragments of the 32-row
// m-blocks are 1-KiB coalesced loads as well, served by L2.
//
// One workgroup = one n-tile x all of K; its NW waves split K into contiguous runs, accumulate
// [32 n x 32 MB m] in registers, reduce through LDS, and all threads run the fused epilogue
// (bias, GELU, residual, KV-cache append / QKV split, packed store for the next GEMM).
// ---------------------------------------------------------------------------------------------
__global__ void pack_stream_weights_kernel(const float* w, bf16_t* out, int N, int K) {
    const size_t total = (size_t)N * K;
    const int KS = K / 16;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        const int lane = (int)((i >> 3) & 63);
        const size_t chunk = i >> 9;
        const int ks = (int)(chunk % KS);
        const int nt = (int)(chunk / KS);
        const int n = nt * 32 + (lane & 31);
        const int k = ks * 16 + 8 * (lane >> 5) + j;
        out[i] = f32_to_bf16(w[(size_t)n * K + k]);
    }
}
bool stream_gemm_supported(int N, int K) { return N % 32 == 0 && K % 16 == 0; }
hipError_t launch_pack_stream_weights(const float* w32, bf16_t* packed, int N, int K, hipStream_t st) {
    const size_t n = (size_t)N * K;
    const int grid = (int)std::min<size_t>((n + 255) / 256, 8192);
    pack_stream_weights_kernel<<<grid, 256, 0, st>>>(w32, packed, N, K);
    return hipGetLastError();
}

// This thread's share of the partial LayerNorm row statistics (thread = pg * ROWS + row): loads issued together
// (clamped, never predicated), then summed in index order.
template <int ROWS, int PGROUPS>
__device__ inline void ln_partial_stats(const GemmArgs& g, float& ln_s, float& ln_q, int first_part = 0) {
    const int row = threadIdx.x % ROWS, pg = threadIdx.x / ROWS;
    const int Mpad = g.a_packed_mb * 32, mrow = blockIdx.y * ROWS + row;
    const float2* base = reinterpret_cast<const float2*>(g.ln_parts) + mrow;
    constexpr int NPL = 4;
    for (int p0 = first_part + pg; p0 < g.ln_nparts; p0 += NPL * PGROUPS) {
        float2 v[NPL];
#pragma unroll
        for (int i = 0; i < NPL; ++i) v[i] = base[(size_t)min(p0 + i * PGROUPS, g.ln_nparts - 1) * Mpad];
#pragma unroll
        for (int i = 0; i < NPL; ++i)
            if (p0 + i * PGROUPS < g.ln_nparts) { ln_s += v[i].x; ln_q += v[i].y; }
    }
}

// Tile of one workgroup: (32 NT) weight rows x (32 MBW) activation rows x K / gridDim.z; the NW waves
// split that K range into contiguous runs.  gridDim = (N / (32 NT), MB_total / MBW, S).  With S > 1
// (cross-workgroup split-K) the fp32 partial tile goes to slab z of `slabs` ([S][Mpad][N]) and the
// epilogue runs in the consumer (LayerNorm combine); with S == 1 the fused epilogue runs here.
// dynamic LDS of a variant: NW padded partial tiles (32 x 36 floats per 32 x 32 block) + row statistics
constexpr size_t stream_gemm_lds(int MBW, int NT, int NW) { return (size_t)NW * NT * MBW * 32 * 36 * 4 + (size_t)MBW * 256 + (size_t)NW * 512; }
// occupancy hint (waves per SIMD): 128 VGPRs let two 8-wave workgroups (or one 16-wave workgroup) share a CU
constexpr int stream_min_waves(int MBW, int NT, int NW, int U) {
    return (NW == 16 || (MBW * NT == 1 && NW == 8) || 16 * NT * MBW + 4 * U * (NT + MBW) + 40 <= 128) ? 4 : 1;
}
// PIPE: the wave's k-steps run as a software pipeline of depth U (k-step i + U is fetched into the registers k-step i just
// released) instead of load-a-run / wait / multiply-a-run: the variants of the merged passes (M = 256 .. 1024), whose
// per-wave K range is several runs long.  Needs cnt % U == 0 (the planner checks).
template <int MBW, int NT, int NW, int U, typename TC, int ABL = 0, bool PIPE = false>   // ABL: ablation switches of tools/micro/bench_stream
__global__ __launch_bounds__(NW * 64, stream_min_waves(MBW, NT, NW, U)) void stream_gemm_kernel(GemmArgs g, const u32x4* __restrict__ wpk, float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* red = reinterpret_cast<float*>(smem_raw);                  // [NW][NT][MBW][m32][RP]: row m holds its 32 n, padded to 36 floats
    constexpr int TILE = NT * MBW * 1024;                             // outputs per workgroup
    constexpr int RP = 36;                                            // row pitch: 8 lanes x ds_write_b128 land on 8 distinct 4-bank groups
    constexpr int TILE_P = NT * MBW * 32 * RP;
    long long stamp[6];
    if (ABL == 9) { stamp[0] = wall_clock64(); stamp[1] = clock64(); }
    // deferred LayerNorm: this thread's share of the partial row statistics (issued first, consumed in the epilogue)
    float* lnstat = red + (size_t)NW * TILE_P;                        // [MBW*32][2] (sum, sumsq), then (mean, rstd)
    float ln_s = 0.0f, ln_q = 0.0f;
    constexpr int ROWS = MBW * 32, PGROUPS = NW * 64 / ROWS;          // threads per row
    // Variants with registers to spare (MBW >= 2: one workgroup per CU anyway) fetch their share of the partial row
    // statistics together with the first run's operands and hold it across the MFMA loop; the others fetch it after
    // the loop, when the operand registers are dead.  Unconditional loads (a valid dummy row when there is no LayerNorm).
    constexpr bool EARLY_STATS = MBW >= 2 && (U >= 12 || PIPE);
    constexpr int NPE = PIPE ? 12 : 6;                 // pipelined variants hold every partial of D <= 1536 (48 parts over 4 thread groups)
    float2 sv[NPE];
    bool sv_loaded = false;
    const float2* sbase = reinterpret_cast<const float2*>(g.ln_parts ? (const void*)g.ln_parts : g.A) + (blockIdx.y * ROWS + threadIdx.x % ROWS);
    const int snp = g.ln_parts ? g.ln_nparts : 1;
    // KV-cache row of this step (device-side step state): fetched first, consumed in the epilogue
    const int qkv_row_dev = (g.store == STORE_QKV && g.row_offset_dev) ? *g.row_offset_dev : 0;
    const int ntile0 = blockIdx.x * NT;
    const int mb0 = blockIdx.y * MBW;
    const int MB = g.a_packed_mb;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int KS = g.K >> 4;
    const int S = gridDim.z;
    const int kz0 = (int)(((long long)KS * blockIdx.z) / S), kz1 = (int)(((long long)KS * (blockIdx.z + 1)) / S);
    const int per = (kz1 - kz0 + NW - 1) / NW;
    const int ks0 = kz0 + wave * per;
    const int cnt = max(0, min(kz1, ks0 + per) - ks0);
    const u32x4* wp = wpk + ((size_t)ntile0 * KS + ks0) * 64 + lane;          // + t * KS * 64 per n-tile
    const u32x4* xp = reinterpret_cast<const u32x4*>(g.A) + ((size_t)ks0 * MB + mb0) * 64 + lane;

    f32x16 acc[NT][MBW];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][mb][r] = 0.0f;

    // Loads are issued unconditionally (indices clamped, never predicated): a load inside a branch
    // gets its own s_waitcnt and the whole run degenerates into serial round trips.
    u32x4 wbuf[U][NT];
    u32x4 xbuf[U][MBW];
    int ks = 0;
    bool first = true;
    const unsigned polled = chain_poll(g.chain);
    // ---- vector epilogue (the AR loop's store modes): a thread finishes 4 consecutive columns of one row.  Its bias, column sums and
    //      residual row are fetched under the end of the K loop (pipelined variants) or under the reduction, never as dependent loads after it.
    const bool vec_epi = S == 1 && (ABL == 0 || ABL == 9 || ABL == 6 || ABL == 7) && g.N % 4 == 0 && g.ldc % 4 == 0 &&
                         (g.store == STORE_RESID || g.store == STORE_PACKED || (g.store == STORE_QKV && g.qkv_D % (32 * NT) == 0) ||
                          (g.store == STORE_ROWS && g.rows_per_group == 0 && !g.resid && g.batch <= 1));
    constexpr int VGROUPS = TILE / 4;                                 // 4-column groups per workgroup
    constexpr int VPT = (VGROUPS + NW * 64 - 1) / (NW * 64);          // groups per thread (1 for the instantiated variants)
    f32x4 e_bias[VPT], e_cs[VPT], e_res[VPT];
    bool epi_prefetched = false;
    auto prefetch_epilogue = [&]() {
        epi_prefetched = true;
        if (!vec_epi) return;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const int o4 = min((int)(threadIdx.x + j * NW * 64), VGROUPS - 1) * 4;
            const int n = o4 & 31, ml = (o4 >> 5) & 31, blk = o4 >> 10;
            const int mb = blk % MBW, t = blk / MBW;
            const int m = min((mb0 + mb) * 32 + ml, g.M - 1);
            const int ncol = (ntile0 + t) * 32 + n;
            const float* zsrc = reinterpret_cast<const float*>(g.A);          // any valid 16 bytes: unconditional loads
            e_bias[j] = *reinterpret_cast<const f32x4*>(g.bias ? g.bias + ncol : zsrc);
            e_cs[j] = *reinterpret_cast<const f32x4*>(g.ln_parts ? g.ln_colsum + ncol : zsrc);
            e_res[j] = *reinterpret_cast<const f32x4*>(g.store == STORE_RESID ? reinterpret_cast<const float*>(g.C) + (size_t)m * g.ldc + ncol : zsrc);
        }
    };
    // The weights are fetched with ORDINARY loads: the depth blocks run twice per position and steps in flight on other lanes re-read
    // what this one just pulled through the 256 MB Infinity Cache (the non-temporal hint measured 1173 / 1141 vs 1183 images/s with
    // 3 lanes, body blocks only / every weight; one lane alone 76.4 / 76.9 vs 75.9 ms).  WNT = true is the non-temporal form of the
    // same loop, kept as a template argument for tools/micro/bench_stream.
    auto run_main = [&](auto nt_tag) {
    constexpr bool WNT = decltype(nt_tag)::value;
    for (; ks + U <= cnt; ks += U) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const u32x4* wsrc = wp + ((size_t)t * KS + (ABL == 2 ? 0 : ks + u)) * 64;
                wbuf[u][t] = WNT ? __builtin_nontemporal_load(wsrc) : *wsrc;
            }
        if (first) {                                  // weights are in flight; everything below depends on the predecessor
            __builtin_amdgcn_sched_barrier(0);
            chain_wait(g.chain, polled);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mb = 0; mb < MBW; ++mb) xbuf[u][mb] = xp[((size_t)(ABL == 1 ? 0 : ks + u) * MB + mb) * 64];
        if (EARLY_STATS && first) {
            sv_loaded = true;
#pragma unroll
            for (int i = 0; i < NPE; ++i) sv[i] = sbase[(size_t)min((int)(threadIdx.x / ROWS) + i * PGROUPS, snp - 1) * (g.a_packed_mb * 32)];
        }
        first = false;
        // keep every load of the run in flight before the first MFMA: without this fence the
        // scheduler re-rolls the block into 2-4 loads per wait to save VGPRs (12+ serial round trips)
        __builtin_amdgcn_sched_barrier(0);
        if (ABL == 9) { stamp[2] = clock64(); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bf16x8 wf = __builtin_bit_cast(bf16x8, wbuf[u][t]);
#pragma unroll
                for (int mb = 0; mb < MBW; ++mb)
                    acc[t][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, __builtin_bit_cast(bf16x8, xbuf[u][mb]), acc[t][mb], 0, 0, 0);
            }
    }
    };
    auto run_pipe = [&](auto nt_tag) {
    constexpr bool WNT = decltype(nt_tag)::value;
    auto fetch_w = [&](int u, int k) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const u32x4* wsrc = wp + ((size_t)t * KS + k) * 64;
            wbuf[u][t] = WNT ? __builtin_nontemporal_load(wsrc) : *wsrc;
        }
    };
    auto fetch_x = [&](int u, int k) {
#pragma unroll
        for (int mb = 0; mb < MBW; ++mb) xbuf[u][mb] = xp[((size_t)k * MB + mb) * 64];
    };
    auto multiply = [&](int u) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bf16x8 wf = __builtin_bit_cast(bf16x8, wbuf[u][t]);
#pragma unroll
            for (int mb = 0; mb < MBW; ++mb)
                acc[t][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, __builtin_bit_cast(bf16x8, xbuf[u][mb]), acc[t][mb], 0, 0, 0);
        }
    };
    if (cnt < U) return;                              // (planner: cnt % U == 0; anything else falls to the ragged tail below)
#pragma unroll
    for (int u = 0; u < U; ++u) fetch_w(u, u);
    __builtin_amdgcn_sched_barrier(0);
    chain_wait(g.chain, polled);                      // weights are in flight; the activations depend on the predecessor
#pragma unroll
    for (int u = 0; u < U; ++u) fetch_x(u, u);
    if (EARLY_STATS) {                                // the partial row statistics travel with the first operands (consumed after the loop)
        sv_loaded = true;
#pragma unroll
        for (int i = 0; i < NPE; ++i) sv[i] = sbase[(size_t)min((int)(threadIdx.x / ROWS) + i * PGROUPS, snp - 1) * (g.a_packed_mb * 32)];
    }
    first = false;
    __builtin_amdgcn_sched_barrier(0);
    for (; ks + 2 * U <= cnt; ks += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            multiply(u);
            fetch_w(u, ks + U + u);
            fetch_x(u, ks + U + u);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    prefetch_epilogue();                              // bias / column sums / residual rows: in flight under the last U k-steps and the reduction
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) { multiply(u); __builtin_amdgcn_sched_barrier(0); }
    ks += U;
    };
    if (PIPE) run_pipe(std::false_type{}); else run_main(std::false_type{});
    if (first) {                                      // K too short for a full run: nothing was prefetched
        chain_wait(g.chain, polled);
    }
    if (ks < cnt) {                                   // ragged tail (tiny K only)
        const int rem = cnt - ks;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = ks + min(u, rem - 1);
#pragma unroll
            for (int t = 0; t < NT; ++t) wbuf[u][t] = wp[((size_t)t * KS + kk) * 64];
#pragma unroll
            for (int mb = 0; mb < MBW; ++mb) xbuf[u][mb] = xp[((size_t)kk * MB + mb) * 64];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (u < rem) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bf16x8 wf = __builtin_bit_cast(bf16x8, wbuf[u][t]);
#pragma unroll
                    for (int mb = 0; mb < MBW; ++mb)
                        acc[t][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, __builtin_bit_cast(bf16x8, xbuf[u][mb]), acc[t][mb], 0, 0, 0);
                }
            }
    }
    if (ABL == 9) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); stamp[3] = clock64(); __builtin_amdgcn_sched_barrier(0); }
    // deferred LayerNorm: partial row statistics, loaded while the operand registers are dead and the accumulators drain
    if (g.ln_parts) {
        if (EARLY_STATS && sv_loaded) {
#pragma unroll
            for (int i = 0; i < NPE; ++i)
                if ((int)(threadIdx.x / ROWS) + i * PGROUPS < g.ln_nparts) { ln_s += sv[i].x; ln_q += sv[i].y; }
            ln_partial_stats<ROWS, PGROUPS>(g, ln_s, ln_q, NPE * PGROUPS);
        } else {
            ln_partial_stats<ROWS, PGROUPS>(g, ln_s, ln_q);
        }
    }
    if (!epi_prefetched) prefetch_epilogue();
    // ---- cross-wave reduction through LDS; C/D map: col = lane & 31 -> m, row = (r&3) + 8 (r>>2) + 4 (lane>>5) -> n
    {
        float* my = red + (size_t)wave * TILE_P;
        const int h = lane >> 5, c = lane & 31;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 v = {acc[t][mb][4 * gq], acc[t][mb][4 * gq + 1], acc[t][mb][4 * gq + 2], acc[t][mb][4 * gq + 3]};
                    *reinterpret_cast<f32x4*>(my + ((t * MBW + mb) * 32 + c) * RP + 8 * gq + 4 * h) = v;
                }
    }
    if (g.ln_parts) {                                   // fixed-order (deterministic) reduction of the partial statistics
        float* lnpart = lnstat + 2 * ROWS;              // [PGROUPS][ROWS][2]
        lnpart[2 * threadIdx.x] = ln_s;                 // thread id = pg * ROWS + row
        lnpart[2 * threadIdx.x + 1] = ln_q;
    }
    __syncthreads();
    if (g.ln_parts) {
        const float* lnpart = lnstat + 2 * ROWS;
        if (threadIdx.x < ROWS) {
            float a = 0.0f, q = 0.0f;
            for (int pg = 0; pg < PGROUPS; ++pg) { a += lnpart[2 * (pg * ROWS + threadIdx.x)]; q += lnpart[2 * (pg * ROWS + threadIdx.x) + 1]; }
            const float mean = a / (float)g.K;
            const float var = fmaxf(q / (float)g.K - mean * mean, 0.0f);
            lnstat[2 * threadIdx.x] = mean;
            lnstat[2 * threadIdx.x + 1] = 1.0f / sqrtf(var + g.ln_eps);
        }
        __syncthreads();
    }
    if (ABL == 9) stamp[4] = clock64();
    if (ABL == 3) {
        if (threadIdx.x == 0) reinterpret_cast<float*>(g.C)[blockIdx.x] = red[0] + red[TILE_P];
        return;
    }
    // STORE_QKV destination of this workgroup's 32 columns
    const bool qkv_uniform = g.store == STORE_QKV && g.qkv_D % (32 * NT) == 0;      // the workgroup's columns lie in one part
    const int qkv_part_local = g.store == STORE_QKV ? (ntile0 * 32) / max(g.qkv_D, 1) : 0;
    const int qkv_part = qkv_part_local + g.qkv_first;
    TC* const qkv_base = reinterpret_cast<TC*>(qkv_part == 0 ? g.C : (qkv_part == 1 ? g.C2 : g.C3));
    bf16_t* const qkv_vcopy = (g.store == STORE_QKV && qkv_part == 2) ? g.qkv_v_pk : nullptr;
    if (vec_epi) {
        f32x4 part[VPT];                                  // this workgroup's sum over its K range, 4 columns of one row per group
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const int o4 = min((int)(threadIdx.x + j * NW * 64), VGROUPS - 1) * 4;
            const float* rp = red + ((o4 >> 10) * 32 + ((o4 >> 5) & 31)) * RP + (o4 & 31);
            f32x4 sv = *reinterpret_cast<const f32x4*>(rp);
#pragma unroll
            for (int w = 1; w < (ABL == 7 ? 1 : NW); ++w) sv += *reinterpret_cast<const f32x4*>(rp + (size_t)w * TILE_P);   // ABL 7 (bench_stream): no cross-wave sum
            part[j] = sv;
        }
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const int vo = threadIdx.x + j * NW * 64;
            const bool live = vo < VGROUPS;
            const int o4 = min(vo, VGROUPS - 1) * 4;
            const int n = o4 & 31, ml = (o4 >> 5) & 31, blk = o4 >> 10;
            const int mb = blk % MBW, t = blk / MBW;
            const int m = (mb0 + mb) * 32 + ml;
            const int ncol = (ntile0 + t) * 32 + n;
            const f32x4 sv = part[j];
            float v[4] = {sv[0], sv[1], sv[2], sv[3]};
            if (g.ln_parts) {
                const float mean = lnstat[2 * (mb * 32 + ml)], rstd = lnstat[2 * (mb * 32 + ml) + 1];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = rstd * (v[e] - mean * e_cs[j][e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] * g.alpha + (g.bias ? e_bias[j][e] : 0.0f);
            const bool row_ok = live && m < g.M;
            if (g.store == STORE_RESID) {
                // residual producer: fp32 master row, bf16 packed copy for the next GEMM, partial row statistics of that copy
                float rs = 0.0f, rq = 0.0f;
                if (row_ok) {
                    f32x4 x4;
                    bf16_t hb[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        x4[e] = e_res[j][e] + v[e];
                        hb[e] = f32_to_bf16(x4[e]);
                        const float r = bf16_to_f32(hb[e]);
                        rs += r; rq += r * r;
                    }
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + (size_t)m * g.ldc + ncol) = x4;
                    uint2 pk;
                    pk.x = (unsigned)hb[0] | ((unsigned)hb[1] << 16);
                    pk.y = (unsigned)hb[2] | ((unsigned)hb[3] << 16);
                    *reinterpret_cast<uint2*>(g.resid_pk + packed_off(m, ncol, g.c_packed_mb)) = pk;
                }
#pragma unroll
                for (int off = 4; off > 0; off >>= 1) { rs += __shfl_xor(rs, off, 64); rq += __shfl_xor(rq, off, 64); }   // 8 lanes share row m
                if (live && n == 0 && m < g.c_packed_mb * 32) {
                    float* pp = g.resid_parts + ((size_t)(ntile0 + t) * (g.c_packed_mb * 32) + m) * 2;
                    pp[0] = rs; pp[1] = rq;
                }
                continue;
            }
            if (!row_ok) continue;
            if (g.store == STORE_QKV) {
                const int nn = ncol - qkv_part_local * g.qkv_D;
                long long row = m;
                if (qkv_part > 0) row = (m / g.rows_per_group) * g.group_stride + m % g.rows_per_group + g.row_offset + qkv_row_dev;
                st4<TC>(qkv_base + row * g.ldc + nn, v);
                if (qkv_vcopy) st4<bf16_t>(qkv_vcopy + packed_off(m, nn, g.c_packed_mb), v);
                continue;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], g.act);
            if (g.store == STORE_PACKED) st4<TC>(reinterpret_cast<TC*>(g.C) + packed_off(m, ncol, g.c_packed_mb), v);
            else if (ABL != 6 || v[0] == 12345.678f) st4<TC>(reinterpret_cast<TC*>(g.C) + (long long)m * g.ldc + ncol, v);      // ABL 6 (bench_stream): no output stores
        }
    } else
    for (int o = threadIdx.x; o < TILE; o += NW * 64) {
        const int n = o & 31, ml = (o >> 5) & 31, blk = o >> 10;
        const int mb = blk % MBW, t = blk / MBW;
        const int m = (mb0 + mb) * 32 + ml;
        const int op = (blk * 32 + ml) * RP + n;
        float s = red[op];
#pragma unroll
        for (int w = 1; w < NW; ++w) s += red[(size_t)w * TILE_P + op];
        const int ncol = (ntile0 + t) * 32 + n;
        if (g.ln_parts) s = lnstat[2 * (mb * 32 + ml) + 1] * (s - lnstat[2 * (mb * 32 + ml)] * g.ln_colsum[ncol]);
        if (g.store == STORE_RESID && S == 1) {
            // residual producer: fp32 master row, bf16 packed copy for the next GEMM, partial row statistics of that copy
            float r = 0.0f;
            if (m < g.M) {
                float* xp = reinterpret_cast<float*>(g.C) + (size_t)m * g.ldc + ncol;
                const float v = *xp + s * g.alpha + (g.bias ? g.bias[ncol] : 0.0f);
                *xp = v;
                const bf16_t hb = f32_to_bf16(v);
                g.resid_pk[packed_off(m, ncol, g.c_packed_mb)] = hb;
                r = bf16_to_f32(hb);
            }
            float rs = r, rq = r * r;                               // the 32 lanes of a half-wave share row m
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) { rs += __shfl_xor(rs, off, 64); rq += __shfl_xor(rq, off, 64); }
            if (n == 0 && m < g.c_packed_mb * 32) {
                float* pp = g.resid_parts + ((size_t)(ntile0 + t) * (g.c_packed_mb * 32) + m) * 2;
                pp[0] = rs; pp[1] = rq;
            }
            continue;
        }
        if (m >= g.M) continue;
        if (S > 1 && ABL != 9) { slabs[((size_t)blockIdx.z * (MB * 32) + m) * g.N + ncol] = s; continue; }
        if (g.store == STORE_QKV && qkv_uniform) {
            // fused [query; key; value]: D % 32 == 0, so the part is uniform per workgroup; destination picked once
            // (a per-element select between C / C2 / C3 inside gemm_store was miscompiled by hipcc -O3 whenever a
            // second bf16 store followed it: stores silently went missing)
            const float v = s * g.alpha + (g.bias ? g.bias[ncol] : 0.0f);
            const int nn = ncol - qkv_part_local * g.qkv_D;
            long long row = m;
            if (qkv_part > 0) {
                row = (m / g.rows_per_group) * g.group_stride + m % g.rows_per_group + g.row_offset + qkv_row_dev;
            }
            st1<TC>(qkv_base + row * g.ldc + nn, v);
            if (qkv_vcopy) qkv_vcopy[packed_off(m, nn, g.c_packed_mb)] = f32_to_bf16(v);
            continue;
        }
        gemm_store<TC>(g, 0, m, ncol, s);
    }
    chain_signal(g.chain);
    if (ABL == 9) {
        stamp[5] = clock64();
        if (lane == 0) {
            long long* dbg = reinterpret_cast<long long*>(slabs) + ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * NW + wave) * 8;
            dbg[0] = stamp[0]; dbg[1] = stamp[2] - stamp[1]; dbg[2] = stamp[3] - stamp[1]; dbg[3] = stamp[4] - stamp[1]; dbg[4] = stamp[5] - stamp[1];
            dbg[5] = wall_clock64();
        }
    }
}

// ---- deferred-LayerNorm weight folding (finalize): W' = gamma o W, colsum[n] = sum_k bf16(W'[n][k]), b' = b + W beta
__global__ void fold_gamma_kernel(const float* w, const float* gamma, float* out, int N, int K) {
    const size_t total = (size_t)N * K;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        out[i] = w[i] * gamma[i % K];
}
__global__ __launch_bounds__(256) void fold_rows_kernel(const float* wfold, const float* w, const float* beta, const float* bias,
                                                        float* colsum, float* bias_out, int K) {
    __shared__ float red[8];
    const int n = blockIdx.x;
    float cs = 0.0f, bb = 0.0f;
    for (int k = threadIdx.x; k < K; k += 256) {
        cs += bf16_to_f32(f32_to_bf16(wfold[(size_t)n * K + k]));
        bb += w[(size_t)n * K + k] * beta[k];
    }
    for (int off = 32; off > 0; off >>= 1) { cs += __shfl_xor(cs, off, 64); bb += __shfl_xor(bb, off, 64); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = cs; red[4 + (threadIdx.x >> 6)] = bb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        colsum[n] = (red[0] + red[1]) + (red[2] + red[3]);
        bias_out[n] = (bias ? bias[n] : 0.0f) + ((red[4] + red[5]) + (red[6] + red[7]));
    }
}
hipError_t launch_fold_layernorm(const float* w32, const float* gamma, const float* beta, const float* bias, float* wfold_tmp,
                                 bf16_t* packed, float* colsum, float* bias_out, int N, int K, hipStream_t st) {
    const size_t n = (size_t)N * K;
    fold_gamma_kernel<<<(int)std::min<size_t>((n + 255) / 256, 8192), 256, 0, st>>>(w32, gamma, wfold_tmp, N, K);
    fold_rows_kernel<<<N, 256, 0, st>>>(wfold_tmp, w32, beta, bias, colsum, bias_out, K);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_pack_stream_weights(wfold_tmp, packed, N, K, st);
}

bool stream_gemm_ok(const GemmArgs& g, int a_dt, int c_dt) {
    (void)c_dt;
    return g.a_packed_mb > 0 && a_dt == DT_BF16 && !g.conv_taps && g.M <= PACKED_MAX_ROWS && g.batch <= 1 && g.N % 32 == 0 &&
           g.K % 16 == 0 && g.a_packed_mb == packed_mb(g.M);
}

template <int MBW, int NT, int NW, int U, typename TC, bool PIPE = false>
static hipError_t launch_stream_t(const GemmArgs& g, const bf16_t* wpk, int S, float* slabs, hipStream_t st) {
    const size_t smem = stream_gemm_lds(MBW, NT, NW);
    const dim3 grid(g.N / (32 * NT), g.a_packed_mb / MBW, S);
    stream_gemm_kernel<MBW, NT, NW, U, TC, 0, PIPE><<<grid, NW * 64, smem, st>>>(g, reinterpret_cast<const u32x4*>(wpk), slabs);
    return hipGetLastError();
}
// configuration table (tools/micro/bench_stream on MI355X): per-CU L2->L1 bandwidth (~40 GB/s) bounds these
// kernels, so the decomposition maximises the number of busy CUs; narrow-N GEMMs split K across
// workgroups (S > 1) and leave the fp32 partial slabs to the next LayerNorm.
#define STREAM_CASES(X, TC) \
    X(1, 1, 8, 12, TC) X(2, 1, 8, 12, TC) X(2, 1, 8, 6, TC) \
    X(2, 2, 4, 6, TC) X(2, 2, 8, 6, TC) X(2, 1, 4, 12, TC) X(1, 1, 4, 12, TC) X(1, 2, 4, 8, TC) X(4, 1, 4, 4, TC) X(4, 2, 4, 3, TC) X(2, 1, 4, 6, TC)
// pipelined variants (512 .. 1024 activation rows: depth sub-step 1 of a merged pass, large merges)
#define STREAM_PIPE_CASES(X, TC) X(2, 2, 4, 4, TC) X(2, 3, 4, 4, TC)
int stream_gemm_splitk(const GemmArgs& g) {
    const int KS = g.K / 16;
    const int wgs = (g.N / 32) * g.a_packed_mb;            // with MBW = 1
    if (g.a_packed_mb > 2 || wgs >= 128) return 1;
    for (int S = 4; S >= 2; S >>= 1)
        if (wgs * S <= 256 && KS % (S * 8) == 0) return S;
    return 1;
}
template <typename TC>
static hipError_t launch_stream_c(const GemmArgs& g, const bf16_t* wpk, int S, float* slabs, hipStream_t st) {
    if (S == 1 && g.tune == 1) {
        // throughput policy (several batches in flight): fewest CU-microseconds per GEMM.  Re-swept with cacheable weight loads
        // (round-1 sweep, 3 lanes): 64-row tiles over 6-step runs for the 64-row GEMMs, 64-row WEIGHT tiles for the
        // narrow 256-row ones (proj / fc2 of depth sub-step 1): 1180 -> 1220 images/s; alone these choices cost 4-5 ms of AR.
        if (g.a_packed_mb == 2 && g.N >= 3072) return launch_stream_t<2, 1, 8, 6, TC>(g, wpk, 1, nullptr, st);
        if (g.a_packed_mb == 2) return launch_stream_t<2, 1, 8, 12, TC>(g, wpk, 1, nullptr, st);   // (16-wave variants <1,1,16,6> etc.: no difference beyond the +-2 % run-to-run noise)
        if (g.a_packed_mb == 8 && g.N >= 3072 && (g.N / 32) % 2 == 0) return launch_stream_t<4, 2, 4, 3, TC>(g, wpk, 1, nullptr, st);
        if (g.a_packed_mb == 8 && (g.N / 32) % 2 == 0) return launch_stream_t<2, 2, 8, 6, TC>(g, wpk, 1, nullptr, st);
    }
    if (S == 1 && g.a_packed_mb >= 16) {
        // M = 512 .. 1024: bound by the L2 -> L1 traffic of the 64-row tiles (~20 TB/s over the chip), so the widest weight tile whose
        // partial tiles still fit the LDS twice; tools/micro/bench_stream, M = 1024: qkv 29.7, proj 14.1, fc1 35.4, fc2 33.3 us
        // (round 1 took these shapes through the generic LDS-DMA GEMM at 48 .. 86 us plus separate LayerNorm launches)
        if (g.a_packed_mb >= 32 && g.K > 2 * g.N && (g.N / 32) % 3 == 0) return launch_stream_t<2, 3, 4, 4, TC, true>(g, wpk, 1, nullptr, st);   // fc2 from 1024 rows (at 512: 33 vs 26 us)
        if ((g.N / 32) % 2 == 0) return launch_stream_t<2, 2, 4, 4, TC, true>(g, wpk, 1, nullptr, st);
    }
    const int wgs2 = (g.N / 32) * (g.a_packed_mb / 2);
    if (S == 1 && g.a_packed_mb == 2 && wgs2 >= 128) {
        return launch_stream_t<2, 1, 8, 12, TC>(g, wpk, 1, nullptr, st);
    }
    if (S == 1 && g.a_packed_mb >= 4) {                 // M = 128..256 (depth sub-step 1): 64-row activation tiles halve the weight re-reads
        if (g.N >= 3072) return launch_stream_t<2, 1, 8, 6, TC>(g, wpk, 1, nullptr, st);      // qkv / fc1 / heads: 18-20 us vs 20-25 us
        return launch_stream_t<2, 1, 8, 12, TC>(g, wpk, 1, nullptr, st);                        // proj / fc2: 8.7 / 20 us vs 9.3 / 25 us
    }
    return launch_stream_t<1, 1, 8, 12, TC>(g, wpk, S, slabs, st);
}
hipError_t launch_stream_gemm(const GemmArgs& g, const bf16_t* wpk, int a_dt, int c_dt, int S, float* slabs, hipStream_t st) {
    (void)a_dt;
    return c_dt == DT_BF16 ? launch_stream_c<bf16_t>(g, wpk, S, slabs, st) : launch_stream_c<float>(g, wpk, S, slabs, st);
}
// every instantiation raises its dynamic-LDS limit once, outside stream capture
hipError_t stream_gemm_configure() {
    hipError_t e;
#define CFG(MBW, NT, NW, U, TC)                                                                                    \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(stream_gemm_kernel<MBW, NT, NW, U, TC>),                 \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)stream_gemm_lds(MBW, NT, NW));      \
    if (e != hipSuccess) return e;
    STREAM_CASES(CFG, bf16_t)
    STREAM_CASES(CFG, float)
#undef CFG
#define CFGP(MBW, NT, NW, U, TC)                                                                                   \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(stream_gemm_kernel<MBW, NT, NW, U, TC, 0, true>),        \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)stream_gemm_lds(MBW, NT, NW));      \
    if (e != hipSuccess) return e;
    STREAM_PIPE_CASES(CFGP, bf16_t)
    STREAM_PIPE_CASES(CFGP, float)
#undef CFGP
    return hipSuccess;
}
