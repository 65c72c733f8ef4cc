// Tiled MFMA GEMM of the merged AR passes (FAST precision, 512 .. 16384 activation rows).
//
//   y[M, N] = x[M, K] W[N, K]^T   with the AR loop's fused epilogues (stage2/layers.py:73-85,190,313-315: the nn.Linear calls of
//   a Block; hierarchical_ar.py:561-563,746-760: the heads).
//
// At 64 rows a GEMM of the AR loop is one pass over its weights (stream_gemm_kernel, fast_kernels.hip).  Merged passes
// (InflightSampler(merge=k): 64 k rows in the body, 256 k in depth sub-step 1) sit above the bf16 ridge of the part
// (~312 FLOP/B): they are MFMA-bound, and what bounds a register-direct 64 x 64 tile there is the L2 -> L1 fill of its
// operands.  This kernel shares both operands across the waves of a workgroup through LDS:
//
//   * one workgroup = a BM x BN output tile (256 x 256 with 8 waves, 128 x 128 with 4), wave tile (32 MBW) x (32 NT);
//   * both operands are ALREADY stored as 1-KiB MFMA fragments (activations: packed_off(), common.h; weights:
//     pack_stream_weights_kernel) -- a fragment is one lane-linear LDS-DMA (global_load_lds_dwordx4, no VGPR staging, no
//     swizzle) and one conflict-free ds_read_b128 per wave;
//   * a ring of NSTAGE LDS stages of KU k-steps (16 k each), NSTAGE - 1 stages of DMA in flight across a raw s_barrier with
//     counted s_waitcnt vmcnt (one barrier per stage);
//   * XCD-aware tile order: an XCD's consecutive workgroups walk the row tiles of one weight column block, so a weight byte
//     is fetched from HBM by one XCD;
//   * epilogues straight from the accumulators (the weights are the MFMA A operand: a lane owns 4 consecutive output
//     columns of one row per register quad): deferred LayerNorm (row statistics combined in a prologue), bias, GELU, the
//     fused [query; key; value] split with the KV-cache append, the packed store for the next GEMM, the residual update with
//     its bf16 copy and partial row statistics, or raw fp32 split-K slabs that resid_combine_kernel finishes.
#include "fast_kernels.h"
#include "gemm_generic.h"
#include <algorithm>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

enum { TS_QKV = 0, TS_PACKED = 1, TS_RESID = 2, TS_ROWS = 3, TS_SLAB = 4,
       TS_FUSED = 5 };    // split-K residual producer finished INSIDE the launch (round 6): every slice stores its fp32 partial tile write-through, takes a ticket on the
                          // tile's counter, and the workgroup that draws the last one sums the S partials in slice order and runs the TS_RESID epilogue on the sum --
                          // no resid_combine_kernel launch, no second pass over the slabs from a cold start; nobody waits (MI355X_MICROARCH.md: valid forms, counter row)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#ifndef HQT_TILE_STAGGER
#define HQT_TILE_STAGGER 1
#endif

// NLOAD_ > 0 (round 6): NLOAD_ further waves do nothing but the LDS-DMA of the ring (loader waves), the WGM x WGN waves only read fragments and
// multiply: a `buffer_load ... lds` piece costs its wave 60 .. 180 ISSUE cycles (the texture path takes 64 B/clk), and a wave issues in order -- in the
// all-consumer form every stage opens with CPW such pieces in front of the matrix instructions of the wave that issues them.
template <int WGM_, int WGN_, int MBW_, int NT_, int KU_, int NSTAGE_, int NLOAD_ = 0>
struct TileGeom {
    static constexpr int WGM = WGM_, WGN = WGN_, MBW = MBW_, NT = NT_, KU = KU_, NSTAGE = NSTAGE_, NLOAD = NLOAD_;
    static constexpr int NC = WGM * WGN;                       // consumer waves (wave tile (32 MBW) x (32 NT))
    static constexpr int NW = NC + NLOAD;                      // waves per workgroup
    static constexpr int NDMA = NLOAD ? NLOAD : NC;            // waves that issue the ring's DMA
    static constexpr int BM = 32 * MBW * WGM, BN = 32 * NT * WGN;
    static constexpr int ACH = BM / 32, WCH = BN / 32;         // 1-KiB fragments per k-step
    static constexpr int CPS = ACH + WCH;
    static constexpr int CH_STAGE = KU * CPS;
    static constexpr int CPW = CH_STAGE / NDMA;                // LDS-DMA instructions per issuing wave per stage
    static constexpr int STAGE_BYTES = CH_STAGE * 1024;
    static constexpr int RING_BYTES = NSTAGE * STAGE_BYTES;
    static constexpr int GR = NC * 64 / BM;                    // threads per row in the statistics prologue (consumer waves)
    // after the ring: (mean, rstd) per row, (bias, colsum) per column, prologue scratch
    static constexpr int AUX_BYTES = BM * 8 + BN * 8 + GR * BM * 8;
    static constexpr int LDS_BYTES = RING_BYTES + AUX_BYTES;
    static constexpr int WG_PER_CU = NW == 4 ? (160 * 1024 / LDS_BYTES < 4 ? 160 * 1024 / LDS_BYTES : 4) : (LDS_BYTES <= 80 * 1024 ? 2 : 1);   // by LDS; 4-wave workgroups up to 4 per CU
    static_assert(CH_STAGE % NDMA == 0, "every issuing wave issues the same number of DMA pieces per stage");
    static_assert(NC * 64 % BM == 0 && GR >= 1, "statistics prologue: whole thread groups per row");
    static_assert(WGN * BM * 8 <= RING_BYTES, "residual epilogue: per-wave partial statistics fit the dead ring");
};

// GELU (erf form, stage2/layers.py: nn.GELU()) for bf16 outputs: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the
// bf16 rounding of the result) -- one v_rcp, one v_exp and 8 FMAs instead of erff's ~40 instructions; 128 of them per lane made
// the epilogue of a 256 x 256 tile cost a third of its main loop.
__device__ __forceinline__ float gelu_erf_fast(float v) {
    const float x = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float e = 1.0f - poly * __expf(-x * x);                 // erf(|v| / sqrt 2)
    return 0.5f * v + 0.5f * fabsf(v) * e;
}
// The activation is a TEMPLATE parameter of the kernel (round 6): as a run-time field of GemmArgs hipcc kept three scalar branches PER ELEMENT in the
// epilogue (293 branches in the 128 x 128 kernel; in-kernel stamps: 17 k of a workgroup's 50 k cycles went into "issuing" 64 values per lane, with or without an
// activation selected -- taken branches and the instruction fetches behind them, not arithmetic).
template <int ACT> __device__ __forceinline__ float tile_act(float v) {
    if (ACT == ACT_GELU_ERF) return gelu_erf_fast(v);
    if (ACT == ACT_GELU_SIGMOID) return v / (1.0f + __expf(-1.702f * v));
    return v;
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Workgroup id -> (split-K slice, row tile, column tile).  Workgroups b and b + 8 share an XCD (MI355X_MICROARCH.md), and an XCD runs ~32 of
// them at a time; what its L2 has to pull over the fabric is one K-long fragment stream per DISTINCT row tile and column tile among its
// workgroups (tools/micro/bench_fill: a CU fills at ~135 GB/s from a hot L2 whatever the path, at ~27 GB/s when every CU pulls from beyond it).
// Each XCD therefore gets a contiguous run of a grouped order (bands of GM row tiles, walked column by column): a compact block of tiles, not a
// 32 x 1 column (which made every XCD pull the whole activation panel every k-step: fabric-bound at 4096+ rows).
//   * GM: up to 12 row tiles make ONE band (an XCD then owns whole columns: the weights cross the fabric once, the small activation panel
//     eight times; bands of 8 left 2 of 10 row tiles at 640 rows to the last XCDs, which pulled two thirds of the weight matrix each);
//     more row tiles are cut into equal bands of at most 8.
//   * split-K: with S | 8 slices, XCD x works on slice x % S only (S = 8: every XCD streams ITS eighth of K of both operands, once).
__device__ __forceinline__ void tile_of(int b, int total, int S, int TM, int TN, int& z, int& tm, int& tn) {
    int t;
    if (S > 1 && (8 % S) == 0) {
        const int xcd = b & 7, G = 8 / S, xg = xcd / S;
        z = xcd - xg * S;
        const int q = total / G, r = total - q * G;
        t = (xg < r ? xg * (q + 1) : r * (q + 1) + (xg - r) * q) + (b >> 3);
    } else {
        z = b / total;
        const int id = b - z * total;
        const int q = total >> 3, r = total & 7, xcd = id & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int nb = TM <= 12 ? 1 : (TM + 7) >> 3;
    const int GM = (TM + nb - 1) / nb;
    const int per_group = GM * TN;
    const int group = t / per_group, in_group = t - group * per_group;
    const int first = group * GM, rows = min(TM - first, GM);
    tn = in_group / rows;
    tm = first + in_group - tn * rows;
}

template <class G, int STORE, bool DLN, typename TC, int ACT = ACT_NONE>
__global__ __launch_bounds__(G::NW * 64, (G::NW / 4) * G::WG_PER_CU) void tile_gemm_kernel(GemmArgs g, const char* __restrict__ wpk,
                                                                                              float* __restrict__ slabs, int TM, int TN) {
#if defined(__HIP_DEVICE_COMPILE__)     // (the buffer-resource builtins have no host-side declaration: the host pass sees an empty body)
    constexpr int WGM = G::WGM, MBW = G::MBW, NT = G::NT, KU = G::KU, NSTAGE = G::NSTAGE, NW = G::NW, NC = G::NC, NLOAD = G::NLOAD, NDMA = G::NDMA;
    constexpr int BM = G::BM, BN = G::BN, ACH = G::ACH, CPS = G::CPS, CPW = G::CPW, STAGE = G::STAGE_BYTES;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    float* const meanrstd = reinterpret_cast<float*>(lds + G::RING_BYTES);        // [BM][2]
    float* const colb = meanrstd + 2 * BM;                                        // [BN] bias, [BN] column sums
    float* const scratch = colb + 2 * BN;                                         // [GR][BM][2]
#ifdef HQT_TILE_STAMPS
    long long stamp[8];
    stamp[5] = stamp[6] = stamp[7] = 0;
    stamp[0] = wall_clock64(); stamp[1] = clock64();
#endif
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wm = wave % WGM, wn = wave / WGM;
    const bool loader = NLOAD > 0 && wave >= NC;                                  // wave-uniform
    const int dw = NLOAD > 0 ? max(wave - NC, 0) : wave;                          // index among the DMA-issuing waves
    const int total = TM * TN;
    const int S = gridDim.x / total;
    int z, tile_m, tile_n;                                                        // z: split-K slice (TS_SLAB)
    tile_of(blockIdx.x, total, S, TM, TN, z, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int MB = g.a_packed_mb, KS = g.K >> 4, NTILES = g.N >> 5;
    const int ks_lo = (int)(((long long)KS * z) / S), ks_hi = (int)(((long long)KS * (z + 1)) / S);
    const int KT = (ks_hi - ks_lo) / KU;                                          // stages of this workgroup (launcher: divisible)

    // ---- LDS-DMA pieces of this wave.  The fragments of a stage, A first ([ku][row block]) then W ([ku][column block]), are dealt
    // round-robin to the waves: piece i of wave w is fragment f = i NW + w.  buffer_load ... lds (MUBUF) rather than
    // global_load_lds: a FLAT-encoded LDS-DMA in flight makes hipcc wait lgkmcnt(0) in front of every MFMA group instead of
    // counting the fragment reads (SIInsertWaitcnts treats it as a pending flat access to both memories), and the per-piece
    // address is one SGPR offset against a wave-uniform descriptor instead of a 64-bit vector add.
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), 0, 0xFFFFFFFF, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wpk), 0, 0xFFFFFFFF, 0x00020000);
    constexpr int NA = KU * ACH;                                  // A fragments per stage
    static_assert(NA % NDMA == 0, "pieces 0 .. NA / NDMA - 1 of every issuing wave are A fragments, the rest W fragments");
    unsigned off[CPW], dstoff[CPW];
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int f = i * NDMA + dw;
        if (i < NA / NDMA) {
            const int ku = f / ACH, r = f - ku * ACH;
            off[i] = (unsigned)(((size_t)(ks_lo + ku) * MB + min(tile_m * ACH + r, MB - 1)) * 1024);
            dstoff[i] = (ku * CPS + r) * 1024;
        } else {
            const int fw = f - NA, ku = fw / G::WCH, r = fw - ku * G::WCH;
            off[i] = (unsigned)(((size_t)min(tile_n * G::WCH + r, NTILES - 1) * KS + ks_lo + ku) * 1024);
            dstoff[i] = (ku * CPS + ACH + r) * 1024;
        }
    }
    const unsigned stepA = (unsigned)KU * MB * 1024, stepW = KU * 1024;
    const int lane16 = lane * 16;
    auto issue = [&](int slot) {
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            char* dst = lds + slot * STAGE + dstoff[i];                              // wave-uniform; the hardware adds 16 * lane
            if (i < NA / NDMA) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)dst, 16, lane16, off[i], 0, 0);
                off[i] += stepA;
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)dst, 16, lane16, off[i], 0, 0);
                off[i] += stepW;
            }
        }
    };

    f32x16 acc[NT][MBW];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MBW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.0f;

    // ---- prologue.  The small operands of the epilogue (bias, column sums, partial row statistics) are requested FIRST, then every
    // slot of the ring; vector-memory operations complete in issue order, so the small values are in registers while the ring is
    // still filling (counted waits), and nothing drains the DMA: the first stage is computed as soon as IT has landed.  (Requested
    // after the ring, their first use drained all of it: 128 KiB per CU from every CU at once, ~9 k cycles before the first MFMA.)
    float bias_r = 0.0f, cs_r = 0.0f;
    const int pc = min((int)threadIdx.x, BN - 1);
    if (STORE != TS_SLAB) {
        const float* zsrc = reinterpret_cast<const float*>(g.A);          // any valid address: unconditional loads
        const int n = min(n0 + pc, g.N - 1);
        bias_r = *(g.bias ? g.bias + n : zsrc);
        cs_r = *(DLN ? g.ln_colsum + n : zsrc);
    }
    constexpr int NPL = 6;                                        // one round covers 12 partials (a 128-column-tile producer at D = 1536)
    float2 pv[NPL];
    const int pr = threadIdx.x % BM, pgi = threadIdx.x / BM;
    if (DLN && !loader) {
        const float2* base = reinterpret_cast<const float2*>(g.ln_parts) + min(m0 + pr, MB * 32 - 1);
#pragma unroll
        for (int i = 0; i < NPL; ++i) pv[i] = base[(size_t)min(pgi + i * G::GR, g.ln_nparts - 1) * (MB * 32)];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (NLOAD == 0 || loader) {                                  // (wave-uniform; with loader waves the consumers issue no DMA at all)
#pragma unroll
        for (int s = 0; s < NSTAGE; ++s) issue(s);               // unconditional (the launcher guarantees KT >= NSTAGE): behind a branch hipcc
    }
    __builtin_amdgcn_sched_barrier(0);                           // would count none of them as younger than the small loads and drain the ring
    if (STORE != TS_SLAB && (int)threadIdx.x < BN) {
        colb[pc] = g.bias ? bias_r : 0.0f;
        colb[BN + pc] = cs_r;
    }
    if (DLN) {
        float sm = 0.0f, sq = 0.0f;
#pragma unroll
        for (int i = 0; i < NPL; ++i)
            if (pgi + i * G::GR < g.ln_nparts) { sm += pv[i].x; sq += pv[i].y; }
        if (g.ln_nparts > NPL * G::GR) {                          // more parts than one round holds (48 partials of a streaming-GEMM producer): ordinary loop
            const float2* base = reinterpret_cast<const float2*>(g.ln_parts) + min(m0 + pr, MB * 32 - 1);
            for (int p0 = pgi + NPL * G::GR; p0 < g.ln_nparts; p0 += G::GR) { const float2 v = base[(size_t)p0 * (MB * 32)]; sm += v.x; sq += v.y; }
        }
        if (!loader) {
            scratch[2 * threadIdx.x] = sm;
            scratch[2 * threadIdx.x + 1] = sq;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // raw: __syncthreads() would drain the DMA in flight
        __builtin_amdgcn_sched_barrier(0);
        if (threadIdx.x < BM) {
            float a = 0.0f, b = 0.0f;
#pragma unroll
            for (int k = 0; k < G::GR; ++k) { a += scratch[2 * (k * BM + threadIdx.x)]; b += scratch[2 * (k * BM + threadIdx.x) + 1]; }
            const float mean = a / (float)g.K;
            const float var = fmaxf(b / (float)g.K - mean * mean, 0.0f);
            meanrstd[2 * threadIdx.x] = mean;
            meanrstd[2 * threadIdx.x + 1] = 1.0f / sqrtf(var + g.ln_eps);
        }
    }
    const int qkv_row_dev = (STORE == TS_QKV && g.row_offset_dev) ? *g.row_offset_dev : 0;

    // ---- main loop.  The wave walks the k-steps with two fragment register sets: while the MFMAs of k-step s run, the
    // fragments of k-step s + 1 are read.  The ONE barrier of a stage sits in front of its last k-step: by then every
    // fragment read of the stage has been issued, so behind the barrier (lgkmcnt(0) first) its slot is refilled with stage
    // kt + NSTAGE, and the first fragments of stage kt + 1 -- whose DMA that barrier also makes visible (counted vmcnt:
    // the NSTAGE - 2 younger stages stay in flight) -- are read under the last k-step's MFMAs.
    bf16x8 af[2][MBW], wf[2][NT];
    auto read_frags = [&](auto set_tag, int slot, int ku) {
        constexpr int SET = decltype(set_tag)::value;
        const char* st = lds + slot * STAGE + ku * (CPS * 1024) + lane * 16;
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[SET][j] = *reinterpret_cast<const bf16x8*>(st + (ACH + wn * NT + j) * 1024);
#pragma unroll
        for (int i = 0; i < MBW; ++i) af[SET][i] = *reinterpret_cast<const bf16x8*>(st + (wm * MBW + i) * 1024);
    };
    auto multiply = [&](auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < MBW; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[SET][j], af[SET][i], acc[j][i], 0, 0, 0);   // D rows = n, cols = m
    };
    auto wait_stage = [&](int younger) {                          // all but the `younger` most recent stages of this wave's DMA have landed
        constexpr int MAXY = (63 / CPW) < (NSTAGE - 1) ? (63 / CPW) : (NSTAGE - 1);        // vmcnt is a 6-bit immediate
        if (MAXY >= 5 && younger >= 5) wait_vmcnt<MAXY >= 5 ? CPW * 5 : 0>();
        else if (MAXY >= 4 && younger >= 4) wait_vmcnt<MAXY >= 4 ? CPW * 4 : 0>();
        else if (MAXY >= 3 && younger >= 3) wait_vmcnt<MAXY >= 3 ? CPW * 3 : 0>();
        else if (MAXY >= 2 && younger >= 2) wait_vmcnt<MAXY >= 2 ? CPW * 2 : 0>();
        else if (MAXY >= 1 && younger >= 1) wait_vmcnt<CPW>();
        else wait_vmcnt<0>();
    };
    const bool early = NW < 8 || wave < NW / 2 || HQT_TILE_STAGGER == 0;
    if (NLOAD > 0 && loader) {
        // ---- loader waves: one barrier per stage, in step with the consumers' -- in front of barrier kt this wave's pieces of stage kt + 1 have
        // landed (counted vmcnt: the NSTAGE - 2 younger stages stay in flight), behind it the consumers are done with stage kt: its slot is refilled
        wait_stage(NSTAGE - 1);
        __builtin_amdgcn_s_barrier();
        int slot = 0;
        for (int kt = 0; kt < KT - 1; ++kt) {
            wait_stage(min(NSTAGE - 2, KT - 2 - kt));
            __builtin_amdgcn_s_barrier();
            if (kt + NSTAGE < KT) issue(slot);
            slot = slot + 1 == NSTAGE ? 0 : slot + 1;
        }
        return;                                                   // (a terminated wave no longer counts for the workgroup's barriers: the epilogue's are the consumers')
    }
    if (NLOAD == 0) wait_stage(NSTAGE - 1);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // no scalar load may be pending when the loop is entered: lgkmcnt also counts them, they return out of order, and with one
    // possibly in flight hipcc waits lgkmcnt(0) in front of every MFMA group instead of counting the fragment reads
#ifdef HQT_TILE_STAMPS
    stamp[2] = clock64();
#endif
    __builtin_amdgcn_s_waitcnt(0xC07F);
    read_frags(std::integral_constant<int, 0>{}, 0, 0);
    int rd = 0;
    // one stage; P = fragment set that holds k-step 0 of this stage; LAST: no successor (no barrier, no refill, no reads ahead).
    // The last stage is peeled rather than tested for inside the loop: a conditional read block would make hipcc merge the
    // two paths' pending-read counts and wait for half of the reads it has just issued.
    auto stage = [&](auto par_tag, auto last_tag, int kt) {
        constexpr int P = decltype(par_tag)::value;
        constexpr bool LAST = decltype(last_tag)::value;
#pragma unroll
        for (int ku = 0; ku < KU; ++ku) {
            const bool cur1 = ((P + ku) & 1) != 0;
            if (ku < KU - 1) {
                if (cur1) { read_frags(std::integral_constant<int, 0>{}, rd, ku + 1); __builtin_amdgcn_sched_barrier(0); multiply(std::integral_constant<int, 1>{}); }
                else { read_frags(std::integral_constant<int, 1>{}, rd, ku + 1); __builtin_amdgcn_sched_barrier(0); multiply(std::integral_constant<int, 0>{}); }
                __builtin_amdgcn_sched_barrier(0);
            } else {
                const int nxt = rd + 1 == NSTAGE ? 0 : rd + 1;
                if (!LAST) {
                    if (NLOAD == 0) wait_stage(min(NSTAGE - 2, KT - 2 - kt));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's reads of stage kt are in registers
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    // An LDS-DMA piece costs its wave 60 .. 180 issue cycles.  Waves w and w + 4 share a SIMD: the lower half refills
                    // the slot right behind the barrier, the upper half after this k-step's MFMAs, so that one of the two always
                    // has matrix work in the pipe (all eight issuing together left it idle ~400 cycles per stage).
                    if (NLOAD == 0 && early && kt + NSTAGE < KT) issue(rd);
                    if (cur1) read_frags(std::integral_constant<int, 0>{}, nxt, 0);
                    else read_frags(std::integral_constant<int, 1>{}, nxt, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (cur1) multiply(std::integral_constant<int, 1>{});
                else multiply(std::integral_constant<int, 0>{});
                __builtin_amdgcn_sched_barrier(0);
                if (NLOAD == 0 && !LAST && !early && kt + NSTAGE < KT) issue(rd);
                __builtin_amdgcn_sched_barrier(0);
                rd = nxt;
            }
        }
    };
    constexpr std::integral_constant<int, 0> P0{};
    constexpr std::integral_constant<int, 1> P1{};
    constexpr std::false_type MORE{};
    constexpr std::true_type FINAL{};
    if (KU % 2 == 0) {
        for (int kt = 0; kt < KT - 1; ++kt) stage(P0, MORE, kt);
        stage(P0, FINAL, KT - 1);
    } else {
        int kt = 0;
        for (; kt + 2 < KT; kt += 2) { stage(P0, MORE, kt); stage(P1, MORE, kt + 1); }
        if (kt + 2 == KT) { stage(P0, MORE, kt); stage(P1, FINAL, kt + 1); }
        else stage(P0, FINAL, kt);
    }

#ifdef HQT_TILE_STAMPS
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    stamp[3] = clock64();
    __builtin_amdgcn_sched_barrier(0);
#endif
    // ---- epilogue.  Register r of block (j, i): column n0 + 32 (wn NT + j) + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), row
    // m0 + 32 (wm MBW + i) + (lane & 31): a lane owns 4 consecutive columns of one row per register quad.  That is the shape of the
    // packed_off() layout (8-byte pieces, 512 contiguous bytes per wave store), so packed outputs leave straight from the
    // registers.  Row-major outputs would leave as 32 rows x 32 B per store instruction (measured: 15 .. 45 us per 256 x 256
    // fp32 tile); they are transposed through the wave's private patch of the (dead) ring instead, one 32-row block at a time,
    // and leave as whole 128 / 256-byte row segments, 16 B per lane.
    const int c = lane & 31, h = lane >> 5;
    const int Mc = g.c_packed_mb * 32;
    constexpr int WCOLS = 32 * NT, PITCH = WCOLS + 4;                  // floats; +4: the 8 lanes of a ds_write_b128 group land on 8 distinct 4-bank sets
    float* const stg = reinterpret_cast<float*>(lds) + wave * (32 * PITCH);
    static_assert(NC * 32 * PITCH * 4 <= G::RING_BYTES, "staging patches fit the ring");
    float mean[MBW], rstd[MBW];
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
        mean[i] = DLN ? meanrstd[2 * ((wm * MBW + i) * 32 + c)] : 0.0f;
        rstd[i] = DLN ? meanrstd[2 * ((wm * MBW + i) * 32 + c) + 1] : 1.0f;
    }
    // STORE_QKV: the launcher guarantees qkv_D % BN == 0, so the part (query / key / value) is uniform per workgroup and
    // the destination is resolved once (a per-element select between C / C2 / C3 was miscompiled by hipcc -O3, fast_kernels.hip)
    const int qkv_part_local = STORE == TS_QKV ? n0 / max(g.qkv_D, 1) : 0;
    const int qkv_part = qkv_part_local + g.qkv_first;
    bf16_t* const qkv_base = reinterpret_cast<bf16_t*>(qkv_part == 0 ? g.C : (qkv_part == 1 ? g.C2 : g.C3));
    bf16_t* const qkv_vcopy = (STORE == TS_QKV && qkv_part == 2) ? g.qkv_v_pk : nullptr;
    const auto rs_qkv = __builtin_amdgcn_make_buffer_rsrc(qkv_base, 0, 0xFFFFFFFF, 0x00020000);
    float* const slab_out = STORE == TS_SLAB ? slabs + (size_t)z * (MB * 32) * g.N : nullptr;
    const auto rs_slab = __builtin_amdgcn_make_buffer_rsrc(STORE == TS_FUSED ? slabs : nullptr, 0, 0xFFFFFFFF, 0x00020000);
    const unsigned slab_stride = (unsigned)((size_t)(MB * 32) * g.N * 4);          // bytes between two slices' slabs (the launcher keeps S slabs under 4 GiB)
    float rs[MBW], rq[MBW];
#pragma unroll
    for (int i = 0; i < MBW; ++i) rs[i] = rq[i] = 0.0f;
    const int wcol0 = n0 + wn * WCOLS;                                 // first column of this wave (the launcher guarantees N % BN == 0)
    if (STORE != TS_PACKED) __syncthreads();                           // every wave is past its last fragment read: the ring is dead
    static_assert(STORE != TS_FUSED || (!DLN && G::NLOAD >= 0), "TS_FUSED: plain residual producers");
#ifdef HQT_TILE_STAMPS
    __builtin_amdgcn_sched_barrier(0);
    stamp[5] = clock64();
    __builtin_amdgcn_sched_barrier(0);
#endif
    auto run_epilogue = [&](auto mode_c, auto from_slabs_c) {
    constexpr int MODE = decltype(mode_c)::value;                  // the store mode of this pass over the tile (TS_FUSED: TS_SLAB, then TS_RESID in the last arriver)
    constexpr bool FROM_SLABS = decltype(from_slabs_c)::value;
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
        const int mrow0 = m0 + (wm * MBW + i) * 32;
        // ---- accumulators -> values (deferred LayerNorm, bias, activation); packed stores leave from here
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                if (FROM_SLABS) continue;                              // (second phase of TS_FUSED: the values are the sum of the slabs, fetched on the row-major side)
                const int cl = j * 32 + 8 * q4 + 4 * h;                // column inside the wave's WCOLS
                float v[4];
                if (MODE == TS_SLAB) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[j][i][4 * q4 + e];
                } else {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(colb + wn * WCOLS + cl);
                    const f32x4 cs4 = *reinterpret_cast<const f32x4*>(colb + BN + wn * WCOLS + cl);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = acc[j][i][4 * q4 + e];
                        if (DLN) t = rstd[i] * (t - mean[i] * cs4[e]);
                        v[e] = t * g.alpha + b4[e];
                    }
                }
                if (MODE == TS_PACKED) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = tile_act<ACT>(v[e]);
                    if (mrow0 + c < g.M) st4<bf16_t>(reinterpret_cast<bf16_t*>(g.C) + packed_off(mrow0 + c, wcol0 + cl, g.c_packed_mb), v);
                    continue;
                }
                if (MODE == TS_ROWS) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = tile_act<ACT>(v[e]);
                }
                if (qkv_vcopy && mrow0 + c < g.M) st4<bf16_t>(qkv_vcopy + packed_off(mrow0 + c, wcol0 + cl - qkv_part_local * g.qkv_D, g.c_packed_mb), v);
                *reinterpret_cast<f32x4*>(stg + c * PITCH + cl) = f32x4{v[0], v[1], v[2], v[3]};
            }
        if (MODE == TS_PACKED) continue;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");         // the patch is exchanged between the lanes of this wave only: LDS
        __builtin_amdgcn_wave_barrier();                               // operations of a wave execute in order, the compiler must keep that order
        // ---- row-major side: a lane holds 16 bytes of one output row
        if (MODE == TS_SLAB || MODE == TS_RESID || (MODE == TS_ROWS && sizeof(TC) == 4)) {
            constexpr int LPR = WCOLS / 4, RPP = 64 / LPR, NP = 32 / RPP;         // lanes per row, rows per pass, passes
            const int cg = (lane % LPR) * 4, r0 = lane / LPR;
            float* const out = MODE == TS_SLAB ? slab_out : reinterpret_cast<float*>(g.C);
            const int ldo = MODE == TS_SLAB ? g.N : g.ldc;
            f32x4 x0[NP];
            if (MODE == TS_RESID) {                                   // the residual rows, all passes in flight together
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const int m = min(mrow0 + p * RPP + r0, g.M - 1);
                    x0[p] = *reinterpret_cast<const f32x4*>(out + (size_t)m * ldo + wcol0 + cg);
                }
            }
            if (FROM_SLABS) {
                // TS_FUSED, last arriver: value = bias + sum over the S slices' partial tiles in slice order (what resid_combine_kernel computed: x += sum + bias).
                // Write-through stores on the producers' side, sc1 loads here (every load of the handed-off bytes), behind the ticket: no fence.
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(colb + wn * WCOLS + cg);
                constexpr int HP = NP / 2;                             // two rounds of NP / 2 passes: S x NP / 2 16-byte loads in flight per lane
#pragma unroll
                for (int hp = 0; hp < 2; ++hp) {
                    u32x4 pz[8][HP];
#pragma unroll
                    for (int p = 0; p < HP; ++p) {
                        const int m = min(mrow0 + (hp * HP + p) * RPP + r0, g.M - 1);
                        const unsigned boff = (unsigned)(((size_t)m * g.N + wcol0 + cg) * 4);
#pragma unroll
                        for (int zz = 0; zz < 8; ++zz)
                            if (zz < S) pz[zz][p] = __builtin_amdgcn_raw_buffer_load_b128(rs_slab, boff + (unsigned)zz * slab_stride, 0, 16);
                    }
#pragma unroll
                    for (int p = 0; p < HP; ++p) {
                        const int r = (hp * HP + p) * RPP + r0, m = mrow0 + r;
                        f32x4 sum = __builtin_bit_cast(f32x4, pz[0][p]);
#pragma unroll
                        for (int zz = 1; zz < 8; ++zz)
                            if (zz < S) sum += __builtin_bit_cast(f32x4, pz[zz][p]);     // fixed slice order
                        f32x4 v = x0[hp * HP + p];
                        v += sum + b4;
                        *reinterpret_cast<f32x4*>(stg + r * PITCH + cg) = v;
                        if (m < g.M) *reinterpret_cast<f32x4*>(out + (size_t)m * ldo + wcol0 + cg) = v;
                    }
                }
            } else {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int r = p * RPP + r0, m = mrow0 + r;
                f32x4 v = *reinterpret_cast<const f32x4*>(stg + r * PITCH + cg);
                if (MODE == TS_RESID) {
                    v += x0[p];
                    *reinterpret_cast<f32x4*>(stg + r * PITCH + cg) = v;            // the new residual row goes back for the packed copy below
                }
                if (MODE == TS_SLAB && STORE == TS_FUSED) {            // write-through: the last arriver of this tile reads these bytes inside this launch
                    if (m < g.M) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_slab, (unsigned)(((size_t)m * g.N + wcol0 + cg) * 4) + (unsigned)z * slab_stride, 0, 16);
                } else if (m < g.M) *reinterpret_cast<f32x4*>(out + (size_t)m * ldo + wcol0 + cg) = v;
            }
            }
        } else {                                                       // bf16 rows: fused [query; key; value] or plain
            constexpr int LPR = WCOLS / 8, RPP = 64 / LPR, NP = 32 / RPP;
            const int cg = (lane % LPR) * 8, r0 = lane / LPR;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int r = p * RPP + r0, m = mrow0 + r;
                const f32x4 a = *reinterpret_cast<const f32x4*>(stg + r * PITCH + cg), b = *reinterpret_cast<const f32x4*>(stg + r * PITCH + cg + 4);
                uint4 pk;
                pk.x = (unsigned)f32_to_bf16(a[0]) | ((unsigned)f32_to_bf16(a[1]) << 16);
                pk.y = (unsigned)f32_to_bf16(a[2]) | ((unsigned)f32_to_bf16(a[3]) << 16);
                pk.z = (unsigned)f32_to_bf16(b[0]) | ((unsigned)f32_to_bf16(b[1]) << 16);
                pk.w = (unsigned)f32_to_bf16(b[2]) | ((unsigned)f32_to_bf16(b[3]) << 16);
                if (m >= g.M) continue;
                if (MODE == TS_QKV) {
                    long long row = m;
                    if (qkv_part > 0) row = (long long)(m / g.rows_per_group) * g.group_stride + m % g.rows_per_group + g.row_offset + qkv_row_dev;
                    const long long eoff = row * g.ldc + (wcol0 + cg - qkv_part_local * g.qkv_D);      // elements from the (wave-uniform) base of this part
                    // WRITE-THROUGH (sc1): a launch whose output stays dirty in L2 ends with the write-back of all of it at once; the attention kernel that reads
                    // these rows runs on other CUs anyway (round 6 A/B, same box, AR pass at 640 rows: qkv 22.6 -> 22.1 ms per pass; fp32 rows, slabs and residual
                    // rows gained nothing and keep plain stores: profiles/r06_tile_write_through_ab.txt)
                    if (eoff < (1ll << 30)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pk), rs_qkv, (unsigned)(eoff * 2), 0, 16);
                    else *reinterpret_cast<uint4*>(qkv_base + eoff) = pk;
                } else {
                    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(g.C) + (long long)m * g.ldc + wcol0 + cg) = pk;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (MODE == TS_RESID) {
            // back on the accumulator side: bf16 packed copy of the new residual rows for the next GEMM + their partial statistics
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int cl = j * 32 + 8 * q4 + 4 * h;
                    const f32x4 x4 = *reinterpret_cast<const f32x4*>(stg + c * PITCH + cl);
                    if (mrow0 + c < g.M) {
                        bf16_t hb[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            hb[e] = f32_to_bf16(x4[e]);
                            const float r = bf16_to_f32(hb[e]);
                            rs[i] += r; rq[i] += r * r;
                        }
                        uint2 pk;
                        pk.x = (unsigned)hb[0] | ((unsigned)hb[1] << 16);
                        pk.y = (unsigned)hb[2] | ((unsigned)hb[3] << 16);
                        *reinterpret_cast<uint2*>(g.resid_pk + packed_off(mrow0 + c, wcol0 + cl, g.c_packed_mb)) = pk;
                    }
                }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    };
    if (STORE != TS_FUSED) run_epilogue(std::integral_constant<int, STORE>{}, std::false_type{});
    else {
        run_epilogue(std::integral_constant<int, TS_SLAB>{}, std::false_type{});
        // ---- ticket: every storing wave drains its write-through stores, then ONE lane adds to the tile's counter; the value the add returns tells who was last
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* const ctr = g.tile_ctr + (tile_m * TN + tile_n);
        if (threadIdx.x == 0) {
            const unsigned t = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t + 1u == (unsigned)S) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // all S have arrived: clean for the next launch
            reinterpret_cast<volatile unsigned*>(scratch)[0] = t;
        }
        __syncthreads();
        const unsigned ticket = reinterpret_cast<volatile unsigned*>(scratch)[0];
        if (ticket + 1u != (unsigned)S) return;
        __syncthreads();                                               // (the patches are rewritten below)
        run_epilogue(std::integral_constant<int, TS_RESID>{}, std::true_type{});
    }
#ifdef HQT_TILE_STAMPS
    __builtin_amdgcn_sched_barrier(0);
    stamp[6] = clock64();
    __builtin_amdgcn_sched_barrier(0);
#endif
    if (STORE == TS_RESID || STORE == TS_FUSED) {
        // row statistics of this tile's BN columns: lanes l and l ^ 32 hold the two halves of a row's columns within a wave,
        // the WGN waves of a row block meet in LDS (behind the staging patches); fixed order -> deterministic
        float* red = reinterpret_cast<float*>(lds) + NC * 32 * PITCH;  // [WGN][BM][2]
        static_assert((NC * 32 * PITCH + G::WGN * BM * 2) * 4 <= G::RING_BYTES, "statistics scratch fits behind the staging patches");
#pragma unroll
        for (int i = 0; i < MBW; ++i) {
            const float a = rs[i] + __shfl_xor(rs[i], 32, 64), b = rq[i] + __shfl_xor(rq[i], 32, 64);
            if (h == 0) {
                red[2 * (wn * BM + (wm * MBW + i) * 32 + c)] = a;
                red[2 * (wn * BM + (wm * MBW + i) * 32 + c) + 1] = b;
            }
        }
        __syncthreads();
        for (int r = threadIdx.x; r < BM; r += NC * 64) {
            float a = 0.0f, b = 0.0f;
#pragma unroll
            for (int w = 0; w < G::WGN; ++w) { a += red[2 * (w * BM + r)]; b += red[2 * (w * BM + r) + 1]; }
            if (m0 + r < Mc) {
                float* pp = g.resid_parts + ((size_t)tile_n * Mc + m0 + r) * 2;
                pp[0] = a; pp[1] = b;
            }
        }
    }
#ifdef HQT_TILE_STAMPS
    if (g.am_best && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long* dbg = reinterpret_cast<long long*>(g.am_best) + (size_t)blockIdx.x * 8;
        dbg[0] = stamp[0]; dbg[1] = stamp[2] - stamp[1]; dbg[2] = stamp[3] - stamp[1]; dbg[3] = clock64() - stamp[1]; dbg[4] = wall_clock64();
        dbg[5] = stamp[5] - stamp[1]; dbg[6] = stamp[6] - stamp[1];
    }
#endif
#endif
}

// ---- split-K finish of a residual producer: x[m] += bias + sum_z slab[z][m]; bf16 packed copy; whole-row statistics (one part).
// One workgroup per row, one 4-column group per thread and round; the S slab reads of a group are issued together (S is a
// template parameter: a runtime loop made them S dependent round trips).
// NT threads: the launcher picks N / 4 rounded up to whole waves (384 at D = 1536: every thread exactly one 4-column group, all S + 1 loads of the row in flight
// at once; with 256 threads half of them ran a second, dependent round), at most 1024.
template <int S, int NT>
__global__ __launch_bounds__(NT) void resid_combine_kernel(float* __restrict__ x, const float* __restrict__ slabs, const float* __restrict__ bias,
                                                           bf16_t* __restrict__ xpk, float* __restrict__ parts, int M, int N, int slab_rows, int pk_mb) {
    __shared__ float red[2 * (NT / 64)];
    const int m = blockIdx.x;
    float rs = 0.0f, rq = 0.0f;
    for (int n4 = threadIdx.x * 4; n4 < N; n4 += NT * 4) {
        f32x4 p[S];
#pragma unroll
        for (int zz = 0; zz < S; ++zz) p[zz] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(slabs + ((size_t)zz * slab_rows + m) * N + n4));
        f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)m * N + n4);
        const f32x4 b = bias ? *reinterpret_cast<const f32x4*>(bias + n4) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 sum = p[0];
#pragma unroll
        for (int zz = 1; zz < S; ++zz) sum += p[zz];                                   // fixed z order
        v += sum + b;
        *reinterpret_cast<f32x4*>(x + (size_t)m * N + n4) = v;
        bf16_t hb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hb[e] = f32_to_bf16(v[e]);
            const float r = bf16_to_f32(hb[e]);
            rs += r; rq += r * r;
        }
        uint2 pk;
        pk.x = (unsigned)hb[0] | ((unsigned)hb[1] << 16);
        pk.y = (unsigned)hb[2] | ((unsigned)hb[3] << 16);
        *reinterpret_cast<uint2*>(xpk + packed_off(m, n4, pk_mb)) = pk;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { rs += __shfl_xor(rs, off, 64); rq += __shfl_xor(rq, off, 64); }
    constexpr int NWV = NT / 64;
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = rs; red[NWV + (threadIdx.x >> 6)] = rq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.0f, b = 0.0f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { a += red[w]; b += red[NWV + w]; }      // fixed order
        parts[2 * m] = a;
        parts[2 * m + 1] = b;
    }
}

// ---------------------------------------------------------------------------------------------------------------- host side
// 128 x 128, 4 waves of 64 x 64, 3 stages of 32 k: 48 KiB ring, THREE workgroups per CU.  What the shapes of the AR loop want
// (tools/micro/bench_tile, in-kernel stamps): a workgroup spends about half of its life outside the main loop (ring fill,
// epilogue stores at ~10 B/clk/CU, slot turnover), so throughput comes from co-resident workgroups covering each other --
// with three per CU the matrix pipe is busy 93 % of a main loop.  256 x 256 (one per CU, 71 % busy loop, 20 us of exposed
// prologue + epilogue per 32 us loop) and 256 x 128 tiles measured 5-30 % slower in the dependent chain of the AR loop
// at 512 .. 5120 rows (profiles/r03_micro_tile_gemm.txt); they live on in tools/micro/bench_tile.hip.
typedef TileGeom<2, 2, 2, 2, 2, 3> Tile128;
// 64 x 128, 4 waves of 32 x 64, 36 KiB ring, FOUR workgroups per CU: the 512-row passes.  A 512 x 6144 output is only 192 tiles of
// 128 x 128 (qkv: 144) -- fewer workgroups than the chip has CUs, each walking K alone on its CU at the latency-bound pace of a single
// ring; halving the row tile doubles the workgroups for 1.5x the LDS-DMA bytes per FLOP, which these shapes have to spare.
typedef TileGeom<2, 2, 1, 2, 2, 3> Tile64;
// 64 x 128 with EIGHT waves (32 x 32 each), stages of 64 k (24 KiB), two workgroups per CU: launches of about one workgroup per CU
// (640-row passes: 360 tiles).  In-kernel stamps put a lone 4-wave workgroup's stage at ~650 cycles against 128-256 of MFMA work whatever
// the ring depth (6 stages in flight instead of 2 measured within 1 %: profiles/r04_tile_variants.txt) -- what paces it is the ISSUE of its
// LDS-DMA pieces, ~150 cycles per 1-KiB piece and wave; eight waves issue the same pieces twice as fast.
typedef TileGeom<2, 4, 1, 1, 4, 3> Tile64W8;
// 128 x 128 with FOUR loader waves beside the four multiplying waves (round 6): 4 stages of 32 k (64 KiB ring, two workgroups per CU)
typedef TileGeom<2, 2, 2, 2, 2, 4, 4> Tile128PC;
typedef TileGeom<2, 2, 1, 1, 4, 3> Tile64x64;     // 64 x 64, 4 waves of 32 x 32, stages of 64 k: 48 KiB ring, three workgroups per CU (proj at 640 rows: tile_gemm_plan)

template <class G, int STORE, bool DLN, typename TC, int ACT = ACT_NONE>
static hipError_t launch_tile_t(const GemmArgs& g, const bf16_t* wpk, int S, float* slabs, hipStream_t st) {
    const int TM = (g.M + G::BM - 1) / G::BM, TN = (g.N + G::BN - 1) / G::BN;
    tile_gemm_kernel<G, STORE, DLN, TC, ACT><<<TM * TN * S, G::NW * 64, G::LDS_BYTES, st>>>(g, reinterpret_cast<const char*>(wpk), slabs, TM, TN);
    return hipGetLastError();
}
// STORE_PACKED (the only store mode the AR loop pairs with an activation: mlp.0 + GELU): one instantiation per activation
template <class G, bool DLN>
static hipError_t launch_tile_packed(const GemmArgs& g, const bf16_t* wpk, hipStream_t st) {
    switch (g.act) {
    case ACT_NONE: return launch_tile_t<G, TS_PACKED, DLN, bf16_t, ACT_NONE>(g, wpk, 1, nullptr, st);
    case ACT_GELU_ERF: return launch_tile_t<G, TS_PACKED, DLN, bf16_t, ACT_GELU_ERF>(g, wpk, 1, nullptr, st);
    case ACT_GELU_SIGMOID: return launch_tile_t<G, TS_PACKED, DLN, bf16_t, ACT_GELU_SIGMOID>(g, wpk, 1, nullptr, st);
    }
    return hipErrorInvalidValue;
}

template <class G>
static hipError_t launch_tile_g(const GemmArgs& g, const bf16_t* wpk, int c_dt, int S, float* slabs, hipStream_t st) {
    const bool dln = g.ln_parts != nullptr;
    if (S > 1) return launch_tile_t<G, TS_SLAB, false, float>(g, wpk, S, slabs, st);
    switch (g.store) {
    case STORE_QKV: return dln ? launch_tile_t<G, TS_QKV, true, bf16_t>(g, wpk, 1, nullptr, st) : launch_tile_t<G, TS_QKV, false, bf16_t>(g, wpk, 1, nullptr, st);
    case STORE_PACKED: return dln ? launch_tile_packed<G, true>(g, wpk, st) : launch_tile_packed<G, false>(g, wpk, st);
    case STORE_RESID: return launch_tile_t<G, TS_RESID, false, float>(g, wpk, 1, nullptr, st);
    case STORE_ROWS:
        if (c_dt == DT_F32) return dln ? launch_tile_t<G, TS_ROWS, true, float>(g, wpk, 1, nullptr, st) : launch_tile_t<G, TS_ROWS, false, float>(g, wpk, 1, nullptr, st);
        return dln ? launch_tile_t<G, TS_ROWS, true, bf16_t>(g, wpk, 1, nullptr, st) : launch_tile_t<G, TS_ROWS, false, bf16_t>(g, wpk, 1, nullptr, st);
    }
    return hipErrorInvalidValue;
}

// Shapes the tile kernels take: fragment-packed bf16 operands, 512+ rows, k-steps divisible into stages, the AR loop's store
// modes with 16-byte row segments; STORE_QKV parts aligned to the column tile.
bool tile_gemm_ok(const GemmArgs& g, int a_dt, int c_dt) {
    if (!(g.a_packed_mb >= 16 && a_dt == DT_BF16 && !g.conv_taps && g.batch <= 1 && g.M >= 512)) return false;
    if (g.N % 128 != 0 || g.K % 32 != 0 || g.K < 256 || g.ldc % 8 != 0 || g.resid) return false;
    if (g.act != ACT_NONE && g.store != STORE_PACKED) return false;
    if (g.store == STORE_QKV) return c_dt == DT_BF16 && g.qkv_D % 128 == 0 && g.rows_per_group > 0;
    if (g.store == STORE_PACKED) return c_dt == DT_BF16 && g.c_packed_mb > 0;
    // (proj, K = D: one short K loop over 48 column-tile rows.  Below 640 rows the streaming kernel's 64-row tiles win (23.5 vs 25.2 us at 512 rows); from 640
    //  rows the 8-wave 64 x 128 tiles do (27.8 -> 24.9 us at 640, 30.0 -> 26.8 at 768, 37.5 -> 29.3 at 1024))
    if (g.store == STORE_RESID) return c_dt == DT_F32 && g.c_packed_mb > 0 && !g.ln_parts && g.resid_pk && g.resid_parts && g.N == g.ldc && (g.K >= 3072 || g.M >= 640);
    if (g.store == STORE_ROWS) return g.rows_per_group == 0 && g.act == ACT_NONE;      // (the activation is a template parameter, instantiated for STORE_PACKED only)
    return false;
}

// Split-K factor: only the residual producers with a long K (fc2: K = 4 D) whose tiles would leave most of the 768 resident
// slots empty; the slices leave fp32 slabs that resid_combine_kernel finishes (one extra launch: not worth it at K = D).
TilePlan tile_gemm_plan(const GemmArgs& g) {
    constexpr int max_s = 8;
    TilePlan p{0, Tile128::BM, Tile128::BN, 1};
    const int KS = g.K / 16, tiles = ((g.M + p.bm - 1) / p.bm) * (g.N / p.bn);
    // the narrow residual producer (proj: N = K = D) at the driver's 640 rows: 240 tiles of 64 x 64 (4 waves, three workgroups per CU) instead of 120 of 64 x 128 --
    // every CU gets one, half the matrix work per workgroup: 18.9 -> 16.5 ms per pass (1024 rows: 384 such tiles lose to 192 of 64 x 128, 21.4 vs 19.4)
    if (g.store == STORE_RESID && g.K < 3072 && ((g.M + 63) / 64) * (g.N / 64) <= 256 && KS % Tile64x64::KU == 0 && KS / Tile64x64::KU >= 2 * Tile64x64::NSTAGE)
        return TilePlan{3, 64, 64, 1};
    // one round of 128 x 128 tiles that fills most of the chip (640 rows: qkv 180, fc1 240 tiles): the loader-wave geometry -- in-kernel stamps at 640 rows
    // (profiles/r06_micro_tile_gemm.txt): main loop 18.4 k cycles against 28.9 k for the all-consumer 128 x 128 tile, workgroup life 13.5-14 us against the
    // 8-wave 64 x 128 tiles' grid span of 16.5-17 us (360 / 480 workgroups: two rounds on part of the chip)
    if (g.store != STORE_RESID && tiles >= 160 && tiles <= 256 && KS % Tile128PC::KU == 0 && KS / Tile128PC::KU >= 2 * Tile128PC::NSTAGE)
        return TilePlan{4, Tile128PC::BM, Tile128PC::BN, 1};
    if (g.store != STORE_RESID && tiles < 256 && KS / Tile64::KU >= Tile64::NSTAGE) {
        const int t64 = ((g.M + Tile64::BM - 1) / Tile64::BM) * (g.N / Tile64::BN);
        if (t64 <= 512 && KS % Tile64W8::KU == 0 && KS / Tile64W8::KU >= 2 * Tile64W8::NSTAGE) return TilePlan{2, Tile64::BM, Tile64::BN, 1};
        return TilePlan{1, Tile64::BM, Tile64::BN, 1};
    }
    if (g.store == STORE_RESID && g.K < 3072 && tiles < 256 && KS % Tile64W8::KU == 0 && KS / Tile64W8::KU >= 2 * Tile64W8::NSTAGE)
        return TilePlan{2, Tile64::BM, Tile64::BN, 1};
    // the wide-K residual producer when its 128 x 128 tiles are ONE round over most of the chip (fc2 at 2560 rows: 240 tiles): the loader-wave geometry without split-K --
    // grid span 46 us against 60-69 for 480 tiles of 64 x 128 (profiles/r06_micro_tile_gemm.txt: main loop 73 k cycles against 115 k for the all-consumer 128 x 128 tile)
    if (g.store == STORE_RESID && g.K >= 3072 && tiles >= 160 && tiles <= 256 && KS % Tile128PC::KU == 0 && KS / Tile128PC::KU >= 2 * Tile128PC::NSTAGE)
        return TilePlan{4, Tile128PC::BM, Tile128PC::BN, 1};
    if (g.store == STORE_RESID && g.K >= 3072 && tiles < 256) {
        // The wide-K residual producer (fc2).  Measured per (geometry, S) at the row counts the schedules produce (profiles/r04_fc2_plans.txt; GEMM + combine,
        // ms per pass): what counts is whole ROUNDS of workgroups (128 x 128: one per CU, 8-wave 64 x 128: two) and the slab traffic of the combine.
        //   * 8-wave 64 x 128 tiles without split-K when they fill the chip evenly (at most one or close to two per CU): 1024 rows 53.4 -> 47.5, 2560 rows 20.2 -> 18.4;
        //   * S = 4 when 4 x tiles is one round or a whole number of rounds: 512 rows 36.4 -> 34.7, 640 rows 38.0 -> 35.9, 2048 rows 80.9 -> 74.0 (was S = 8 / 8 / 3);
        //   * else the largest S that keeps the grid under 2.25 rounds.
        const int t64 = ((g.M + Tile64::BM - 1) / Tile64::BM) * (g.N / Tile64::BN);
        if (((t64 >= 176 && t64 <= 208) || (t64 >= 448 && t64 <= 512)) && KS % Tile64W8::KU == 0 && KS / Tile64W8::KU >= 2 * Tile64W8::NSTAGE)
            return TilePlan{2, Tile64::BM, Tile64::BN, 1};
        if ((tiles * 4 <= 256 || tiles % 64 == 0) && KS % (Tile128::KU * 4) == 0 && KS / 4 / Tile128::KU >= 2 * Tile128::NSTAGE) {
            p.S = 4;
            // one round (640 rows: 60 tiles x 4 slices = 240 workgroups): the loader-wave geometry (fc2 at 640 rows, GEMM + combine: 29.2 -> 26.6 us).
            // Finishing the slices INSIDE the launch (TS_FUSED: write-through slabs, the tile's last arriver sums them) was built and measured in round 6 and LOSES:
            // 47.1 us (59.1 on this geometry) against 26.6 -- 256 KB of dependent sc1 reads by one workgroup per tile behind 64 KB of write-through stores per workgroup
            // (profiles/r06_micro_tile_fused.txt); the product never selects it, tools/micro/bench_tile still times and checks it.
            if (tiles * 4 <= 256 && KS / 4 / Tile128PC::KU >= 2 * Tile128PC::NSTAGE) p.geom = 4;
            return p;
        }
        for (int S : {8, 6, 4, 3, 2})
            if (S <= max_s && tiles * S <= 576 && KS % (Tile128::KU * S) == 0 && KS / S / Tile128::KU >= 2 * Tile128::NSTAGE) { p.S = S; break; }
    }
    return p;
}

hipError_t launch_tile_gemm(const GemmArgs& g, const bf16_t* wpk, int a_dt, int c_dt, const TilePlan& p, float* slabs, hipStream_t st) {
    (void)a_dt;
    const GemmArgs& gg = g;
    if (p.geom == 0) return launch_tile_g<Tile128>(gg, wpk, c_dt, p.S, slabs, st);
    if (p.geom == 1) return launch_tile_g<Tile64>(gg, wpk, c_dt, p.S, slabs, st);
    if (p.geom == 2) return launch_tile_g<Tile64W8>(gg, wpk, c_dt, p.S, slabs, st);
    if (p.geom == 3) return launch_tile_g<Tile64x64>(gg, wpk, c_dt, p.S, slabs, st);
    if (p.geom == 4) return launch_tile_g<Tile128PC>(gg, wpk, c_dt, p.S, slabs, st);
    return hipErrorInvalidValue;
}

hipError_t launch_resid_combine(const GemmArgs& g, const float* slabs, int S, hipStream_t st) {
    float* x = reinterpret_cast<float*>(g.C);
    const int rows = g.a_packed_mb * 32;
    switch (S) {
#define COMBINE(S_) case S_:                                                                                                                          \
        if (g.N == 1536) resid_combine_kernel<S_, 384><<<g.M, 384, 0, st>>>(x, slabs, g.bias, g.resid_pk, g.resid_parts, g.M, g.N, rows, g.c_packed_mb);     \
        else resid_combine_kernel<S_, 256><<<g.M, 256, 0, st>>>(x, slabs, g.bias, g.resid_pk, g.resid_parts, g.M, g.N, rows, g.c_packed_mb);                 \
        break;
    COMBINE(2) COMBINE(3) COMBINE(4) COMBINE(6) COMBINE(8)
#undef COMBINE
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// every instantiation raises its dynamic-LDS limit once, outside stream capture
template <class G, int STORE, bool DLN, typename TC, int ACT = ACT_NONE>
static hipError_t configure_one() {
    void (*kernel)(GemmArgs, const char*, float*, int, int) = tile_gemm_kernel<G, STORE, DLN, TC, ACT>;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
}
template <class G>
static hipError_t configure_g() {
    hipError_t e;
#define CFG(STORE, DLN, TC)                      \
    e = configure_one<G, STORE, DLN, TC>();      \
    if (e != hipSuccess) return e;
    CFG(TS_QKV, true, bf16_t) CFG(TS_QKV, false, bf16_t) CFG(TS_PACKED, true, bf16_t) CFG(TS_PACKED, false, bf16_t)
    CFG(TS_RESID, false, float) CFG(TS_ROWS, true, float) CFG(TS_ROWS, false, float) CFG(TS_ROWS, true, bf16_t) CFG(TS_ROWS, false, bf16_t)
    CFG(TS_SLAB, false, float)
#undef CFG
    for (hipError_t e2 : {configure_one<G, TS_PACKED, true, bf16_t, ACT_GELU_ERF>(), configure_one<G, TS_PACKED, false, bf16_t, ACT_GELU_ERF>(),
                          configure_one<G, TS_PACKED, true, bf16_t, ACT_GELU_SIGMOID>(), configure_one<G, TS_PACKED, false, bf16_t, ACT_GELU_SIGMOID>()})
        if (e2 != hipSuccess) return e2;
    return hipSuccess;
}
hipError_t tile_gemm_configure() {
    hipError_t e = configure_g<Tile128>();
    if (e == hipSuccess) e = configure_g<Tile64>();
    if (e == hipSuccess) e = configure_g<Tile64W8>();
    if (e == hipSuccess) e = configure_g<Tile64x64>();
    if (e == hipSuccess) e = configure_g<Tile128PC>();
    return e;
}
