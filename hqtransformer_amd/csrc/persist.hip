// Persistent AR chain of one top position (see persist.h): one 9-wave workgroup per CU, 8 consumer waves + 1 loader wave.
//
// Per phase every CU owns `nc` = 8 .. 32 output columns of the phase's nn.Linear over all (up to 64) rows:
//   loader wave    streams the CU's weight tiles, phase after phase, into a 118-KiB LDS ring with 1-KiB LDS-DMA pieces; it runs
//                  ahead by the whole ring, so a phase finds its tile landed when the grid barrier releases it;
//   consumer wave  (kg, mh): k-group kg of 4 (a quarter of the phase's K range) x row block mh of 2 (32 rows): A fragments straight
//                  from L2 into registers with sc1 loads (the bytes were written by other CUs in this launch), W fragments from
//                  the ring, v_mfma_f32_32x32x16_bf16; LayerNorm row statistics from the fragments passing through -- the row sum
//                  as a ones column of the weight tile (lane 31 of the A operand reads a constant), the sum of squares by
//                  v_dot2c_f32_bf16;
//   epilogue       the four k-group partials meet in LDS; thread (row m, 4 columns) finishes half a 16-byte granule of the output
//                  layout (deferred LayerNorm, bias, GELU, residual, QKV split + cache append) and stores it write-through.
// Tile of (phase, CU) in its stream / in the ring: [k16-step][half of the step's 16 k][nc columns][8 bf16] -- the A operand of the
// MFMA for lane (n = lane % 32 clamped to nc - 1, half = lane / 32) is one ds_read_b128, conflict-free.
// Workgroup barriers per phase: S1 (grid barrier passed, tile landed) and S3 (partials in LDS, tile consumed); the loader executes the
// same two.  "Every wave has drained its stores" (before the grid signal, and before the quad counter of the K-split phase) is a counter
// in LDS whose last arriver signals -- no s_barrier, so the loader streams through the epilogue and the grid wait at its own pace.
#include "persist.h"
#include "gemm_generic.h"
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

namespace {
constexpr int RING_UNITS = PERSIST_RING_UNITS;
constexpr int RING_BYTES = RING_UNITS * 1024;
constexpr int RED_KG = 64 * 36;                          // floats of one k-group's partial tile: row m holds its columns, pitch 36
constexpr int RED_OFF = RING_BYTES;
constexpr int RED_BYTES = 4 * RED_KG * 4;
constexpr int STAT_OFF = RED_OFF + RED_BYTES;            // [4 k-groups][64 rows][sum, sumsq]
constexpr int XS_OFF = STAT_OFF + 4 * 64 * 2 * 4;        // fp32 master of the residual stream: this CU's 8 columns x 64 rows
constexpr int CB_OFF = XS_OFF + 64 * 8 * 4;              // 2 x ([32] bias, [32] column sums): phase p takes buffer p & 1 (filled while slower waves may still finish phase p - 1)
constexpr int ONES_OFF = CB_OFF + 2 * 64 * 4;            // 8 x bf16(1.0): what lane 31 of the weight operand reads in a deferred-LayerNorm phase
constexpr int FLAG_OFF = ONES_OFF + 16;                  // [0] give-up flag, [1] waves that drained their stores (S4), [2] ... their K-split partials (S3b)
constexpr int LDS_BYTES = FLAG_OFF + 16;
constexpr long long SPIN_LIMIT = 100000000ll;            // 1 s of the 100 MHz wall clock: a barrier that waits longer gives up
constexpr int NR = 4;                                    // k16-steps of one A round of a consumer wave
constexpr int WB = 4;                                    // k16-steps whose W fragments are read from the ring together
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
}  // namespace

#if defined(__HIP_DEVICE_COMPILE__)
namespace {
__device__ __forceinline__ void cbar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void raw_bar() { asm volatile("s_barrier" ::: "memory"); }

// A buffer descriptor must sit in SGPRs: hipcc wraps a descriptor it cannot PROVE wave-uniform (the phase table is read with vector loads,
// because the kernel also stores to global memory) in a waterfall loop per load.  readfirstlane makes the pointer provably uniform.
__device__ __forceinline__ int rfl(int x) { return __builtin_amdgcn_readfirstlane(x); }
template <typename T> __device__ __forceinline__ T* rflp(T* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(rflp(const_cast<void*>(p)), 0, 0xFFFFFFFF, 0x00020000);
}
// wave-uniform copy of one phase descriptor
__device__ __forceinline__ PersistPhase phase_of(const PersistPhase* t) {
    PersistPhase ph;
    ph.type = rfl(t->type); ph.N = rfl(t->N); ph.K = rfl(t->K); ph.rot = rfl(t->rot); ph.map = rfl(t->map); ph.dln = rfl(t->dln); ph.act = rfl(t->act);
    ph.cache_T = rfl(t->cache_T); ph.kv_row = rfl(t->kv_row); ph.k4_epoch = rfl(t->k4_epoch);
    ph.A = rflp(t->A); ph.bias = rflp(t->bias); ph.colsum = rflp(t->colsum); ph.out = rflp(t->out); ph.kc = rflp(t->kc); ph.vc = rflp(t->vc); ph.vpk = rflp(t->vpk);
    return ph;
}

// wait until at most n (rounded down to a multiple of 4, at most 60) of this wave's vector-memory operations are outstanding
__device__ __forceinline__ void wait_vm_le(int n) {
    switch (n < 0 ? 0 : (n > 60 ? 15 : n >> 2)) {
#define HQT_W(i) case i: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * i) : "memory"); break;
        HQT_W(0) HQT_W(1) HQT_W(2) HQT_W(3) HQT_W(4) HQT_W(5) HQT_W(6) HQT_W(7) HQT_W(8) HQT_W(9) HQT_W(10) HQT_W(11) HQT_W(12) HQT_W(13) HQT_W(14)
#undef HQT_W
    default: asm volatile("s_waitcnt vmcnt(60)" ::: "memory"); break;
    }
}

template <int AUX>
__device__ __forceinline__ void dma_unit(__amdgpu_buffer_rsrc_t rs, char* dst, int lane16, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, lane16, soff, 0, AUX);
}

// ---- the loader wave: mirrors the workgroup barriers of every phase and streams between them.  A phase's tile is issued completely
// before its S1 (the ring holds the largest tile); beyond that the wave issues in small budgets so that it reaches every barrier before
// the consumers do (a piece costs 60-180 cycles of issue, and the 64th outstanding one blocks the wave until an older one has landed).
__device__ void persist_loader(const PersistArgs& a, char* lds, int cu, int ncu, int lane) {
    const auto rs = rsrc_of(a.wstream + a.cu_off[cu]);
    // KiB of every phase's tile, lane l holding phases l and 64 + l: all descriptor reads in flight together (one after the other -- 60
    // dependent round trips -- they kept the consumers of phase 0 waiting 33 us per launch: profiles/r05_micro_persist.txt)
    int units_lo = 0, units_hi = 0;
    if (lane < a.n_phases) units_lo = persist_tile_units(a.phases[lane], cu, ncu);
    if (lane + 64 < a.n_phases) units_hi = persist_tile_units(a.phases[lane + 64], cu, ncu);
    int total = units_lo + units_hi;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) total += __shfl_xor(total, o);
    total = rfl(total);
    auto units_of = [&](int p) { return p < 64 ? __builtin_amdgcn_readlane(units_lo, p) : __builtin_amdgcn_readlane(units_hi, p - 64); };
    int issued = 0, consumed = 0, end = 0, slot = 0;
    const int lane16 = lane * 16;
    auto issue_one = [&]() {
        char* dst = lds + rfl(slot) * 1024;
        const int soff = rfl(issued) * 1024;
        if (a.nt_weights) dma_unit<2>(rs, dst, lane16, soff); else dma_unit<0>(rs, dst, lane16, soff);
        ++issued;
        slot = slot + 1 == RING_UNITS ? 0 : slot + 1;
    };
    auto fill = [&](int budget) {
        while (budget-- > 0 && issued < total && issued - consumed < RING_UNITS) issue_one();
    };
    fill(RING_UNITS);
    volatile int* flag = reinterpret_cast<volatile int*>(lds + FLAG_OFF);
    // (Measured and removed, profiles/r05_micro_persist_kv_touch.txt: touching the cached K / V rows of a CU's attention unit a phase ahead -- one dword per row, by the
    // consumer waves in front of the grid wait or by this wave behind S1 of the QKV phase.  The attention phase gains 3 us at 63 cached keys, but the 25 MB of misses
    // sit in front of wave 0's poll (in-order returns) or compete with the operand loads of the QKV phase: the body costs 15-28 us MORE with either form.)
    for (int p = 0; p < a.n_phases; ++p) {
        end += units_of(p);
        while (issued < end) issue_one();                // (space is there: a tile never exceeds the ring, and everything before it is consumed)
        wait_vm_le(issued - end);                        // everything up to the end of phase p's tile has landed
        raw_bar();                                       // S1
        if (*flag) return;
        fill(a.fill_s1);                                 // (pieces in flight slow the consumers' operand loads down: default 0)
        raw_bar();                                       // S3: the consumers are done with tile p
        consumed = end;
        fill(a.fill_s3);                                 // epilogue + grid wait of the consumers: the loader's window
    }
}

__device__ __forceinline__ bool spin_until(const unsigned* c, unsigned target, bool active) {
    const long long t0 = wall_clock64();
    for (;;) {
        const unsigned v = active ? __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
        if (__all(v >= target)) return true;
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > SPIN_LIMIT) return false;
    }
}
// one-hop grid barrier: CU c adds to the 8 replicas of shard c % 8, and polls replica c % 8 of all 8 shards
__device__ __forceinline__ bool grid_wait(const PersistArgs& a, int epoch, int cu, int ncu, int lane) {
    return spin_until(a.counters + ((lane & 7) * 8 + (cu & 7)) * 32, (unsigned)epoch * (unsigned)((ncu - (lane & 7) + 7) >> 3), lane < 8);
}

// GELU (erf form) for bf16 outputs: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 rounding of the result), as the
// tile GEMMs of the merged passes use it (tile_gemm.hip: gelu_erf_fast) -- one v_rcp, one v_exp and 8 FMAs instead of erff's ~40 instructions
__device__ __forceinline__ float persist_act(float v, int act) {
    if (act == ACT_GELU_ERF) {
        const float x = fabsf(v) * 0.70710678118654752440f;
        const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
        const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
        const float e = 1.0f - poly * __expf(-x * x);
        return 0.5f * v + 0.5f * fabsf(v) * e;
    }
    if (act == ACT_GELU_SIGMOID) return v / (1.0f + __expf(-1.702f * v));
    return v;
}
__device__ __forceinline__ u32x2 pack4(const float (&v)[4]) {
    u32x2 pk;
    pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
    pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
    return pk;
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
    uint4 pk;
    pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
    pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
    pk.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
    pk.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
    return pk;
}
__device__ __forceinline__ void unpack8(const u32x4 t, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { const unsigned w = t[i]; f[2 * i] = __builtin_bit_cast(float, w << 16); f[2 * i + 1] = __builtin_bit_cast(float, w & 0xffff0000u); }
}
__device__ __forceinline__ void st8_sc1(__amdgpu_buffer_rsrc_t rs, long long byte_off, u32x2 v) { __builtin_amdgcn_raw_buffer_store_b64(v, rs, (int)byte_off, 0, 16); }
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t rs, long long byte_off, u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)byte_off, 0, 16); }
__device__ __forceinline__ u32x4 ld16_sc1(__amdgpu_buffer_rsrc_t rs, long long byte_off) { return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 16); }
__device__ __forceinline__ u32x4 ld16(__amdgpu_buffer_rsrc_t rs, long long byte_off) { return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, 0); }
}  // namespace
#endif

__global__ __launch_bounds__(576, 1) void persist_kernel(PersistArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int wave = rfl(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int cu = blockIdx.x, ncu = gridDim.x;
    volatile int* flag = reinterpret_cast<volatile int*>(lds + FLAG_OFF);
    // a launch that gave up leaves its mark until the host has read it (hqt_range_check): every later launch returns at once instead of
    // waiting out its own time limit (a GPU shared with another process that keeps some CUs busy would otherwise cost a second per launch)
    if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    const unsigned fault = a.err[1];                     // test hook (hqt_set_switch(HQT_SWITCH_PERSIST_FAULT, c + 1)): a device word, so a cached graph replays it
    if (wave == 8) {
        persist_loader(a, lds, cu, ncu, lane);
        return;
    }
    if (threadIdx.x < 4) const_cast<int*>(flag)[threadIdx.x] = 0;
    if (threadIdx.x < 4) reinterpret_cast<unsigned*>(lds + ONES_OFF)[threadIdx.x] = 0x3F803F80u;
    float* const red = reinterpret_cast<float*>(lds + RED_OFF);
    float* const stat = reinterpret_cast<float*>(lds + STAT_OFF);
    float* const xs = reinterpret_cast<float*>(lds + XS_OFF);
    unsigned* const drained = reinterpret_cast<unsigned*>(lds + FLAG_OFF) + 1;
    // called by every consumer wave once its stores are out (s_waitcnt vmcnt(0) first); true in the wave that arrives last
    auto last_wave = [&](unsigned* cnt, unsigned target) {
        unsigned old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return (unsigned)rfl((int)old) + 1u == target;
    };
    const int kg = wave >> 1, mh = wave & 1;
    const int M = a.M, MB = a.MB, D = a.D;
    const int t_cur = a.t_base + (a.t_base_dev ? *a.t_base_dev : 0);
    // this CU's columns of the residual stream: the fp32 master rows stay in LDS for the whole launch
    int rg0, rng, rkq;
    persist_cols(D, 0, PP_MAP_QUAD, cu, ncu, &rg0, &rng, &rkq);
    if (rng && threadIdx.x < 64 && (int)threadIdx.x < M) {
        const float* src = a.x32 + (size_t)threadIdx.x * D + rg0 * 8;
        *reinterpret_cast<f32x4*>(xs + threadIdx.x * 8) = *reinterpret_cast<const f32x4*>(src);
        *reinterpret_cast<f32x4*>(xs + threadIdx.x * 8 + 4) = *reinterpret_cast<const f32x4*>(src + 4);
    }
    int unit0 = 0;                                       // ring slot of the running tile's first unit
    auto stamp = [&](int p, int i) {
        if (a.stamps && threadIdx.x == 0) a.stamps[((size_t)cu * a.n_phases + p) * 8 + i] = wall_clock64();
    };
    auto give_up = [&](int p) {
        if (lane == 0) { __hip_atomic_store(a.err, (unsigned)(p + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *flag = 1; }
    };
    // attention geometry (PP_ATTN): hs / 8 adjacent lanes cover a key row with 16-byte vectors (attention_kernel's scheme)
    const int hs = a.head_dim, chunks = hs >> 3, rpp = 64 / chunks;
    const int nb8 = (M + 7) >> 3, nunits = a.n_heads * nb8;

    const int lane_k = lane;
    for (int p = 0; p < a.n_phases; ++p) {
        // Per-lane values derived from the lane id are loop-invariant: hipcc hoists them ALL out of the phase loop, runs out of registers and
        // spills them to scratch -- reloaded (a memory round trip each) at the head of every phase.  An opaque copy per phase keeps them local.
        int lane = lane_k;
        asm volatile("" : "+v"(lane));
        const int tid = wave * 64 + lane;
        const PersistPhase ph = phase_of(a.phases + p);
        stamp(p, 0);
        const bool gemm = persist_is_gemm(ph.type);
        const int ac = lane % chunks, aslot = lane / chunks;
        int g0 = 0, ng = 0, kq = 0;
        if (gemm) persist_cols(ph.N, ph.rot, ph.map, cu, ncu, &g0, &ng, &kq);
        const int nc = ng * 8, n0 = g0 * 8;
        float* const cb = reinterpret_cast<float*>(lds + CB_OFF) + (p & 1) * 64;
        if (gemm && tid < 32) {
            const int t = tid;
            cb[t] = (ph.bias && t < nc) ? ph.bias[n0 + t] : 0.0f;
            cb[32 + t] = (ph.colsum && t < nc) ? ph.colsum[n0 + t] : 0.0f;
        }
        if (p > 0 && wave == 0 && !grid_wait(a, p, cu, ncu, lane)) give_up(p);
        cbar();                                          // S1: the previous phase is complete on every CU; this phase's tile has landed
        if (*flag) return;
        stamp(p, 1);
        if (gemm) {
            const bool k4 = ph.type == PP_RESID_K4;
            const bool active = ng > 0 && mh < MB;
            const bool dln = ph.dln != 0;
            const bool lnf = ph.type == PP_LNF;          // no product: row statistics of x, then x itself is normalised (its fp32 master is in this CU's LDS)
            const bool ones_col = dln && nc < 32 && !lnf;        // the row sums ride in column 31 of the MFMA tile
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            float ssum = 0.0f, ssq = 0.0f;
            const int Kp = k4 ? ph.K >> 2 : ph.K;        // K range of this CU
            if (active) {
                const int KS4 = Kp >> 6;                 // k16-steps of one k-group
                const int ksA = kq * (Kp >> 4) + kg * KS4;   // first k16-step of this wave in the A operand
                const int ksW = kg * KS4;                //                               ... in the tile
                const auto rsA = rsrc_of(ph.A);
                const int tb = unit0 * 1024;
                const int wstep = nc * 32;
                const bool ones_lane = ones_col && (lane & 31) == 31;
                const int lane_w = (lane >> 5) * (nc * 16) + min(lane & 31, nc - 1) * 16;
                const bf16x2 ones = __builtin_bit_cast(bf16x2, 0x3F803F80u);
                u32x4 b0[NR], b1[NR], b2[NR];
                auto issue = [&](u32x4(&b)[NR], int r) {
#pragma unroll
                    for (int u = 0; u < NR; ++u) {
                        const int ks = ksA + min(r * NR + u, KS4 - 1);
                        b[u] = __builtin_amdgcn_raw_buffer_load_b128(rsA, lane * 16, (ks * MB + mh) * 1024, 16);      // (wave-uniform part as the scalar offset)
                    }
                };
                // One round = NR k16-steps as straight-line code (a branch per step makes every step its own basic block: ds_read -> wait -> MFMA ->
                // dot products in series, ~170 cycles per step instead of ~50).  All W fragments of the round are read first.  MODE: 0 no statistics,
                // 1 sum of squares (the sum rides in the ones column), 2 both by dot products.  TAIL: steps beyond the k-group's range multiply by
                // zero rows (their loads were clamped to a valid address).  MODE 3 (PP_LNF): statistics only, no weights, no matrix instructions.
                auto compute = [&](u32x4(&b)[NR], int r, auto mode_tag, auto tail_tag) {
                    constexpr int MODE = decltype(mode_tag)::value;
                    constexpr bool TAIL = decltype(tail_tag)::value;
#pragma unroll
                    for (int h = 0; h < NR; h += WB) {     // W fragments in batches of WB steps: all of a batch's LDS reads are in flight before its first MFMA
                        bf16x8 wf[WB];
#pragma unroll
                        for (int u = 0; u < WB; ++u) {
                            if (MODE == 3) continue;
                            int off = tb + (ksW + min(r * NR + h + u, KS4 - 1)) * wstep + lane_w;
                            if (off >= RING_BYTES) off -= RING_BYTES;
                            if (MODE == 1 && ones_lane) off = ONES_OFF;
                            wf[u] = *reinterpret_cast<const bf16x8*>(lds + off);
                        }
#pragma unroll
                        for (int u = 0; u < WB; ++u) {
                            u32x4 bu = b[h + u];
                            if (TAIL && r * NR + h + u >= KS4) bu = u32x4{0u, 0u, 0u, 0u};
                            if (MODE != 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u], __builtin_bit_cast(bf16x8, bu), acc, 0, 0, 0);
                            if (MODE != 0) {
                                // (NOT __builtin_bit_cast(bf16x2, b[u][j]): clang reads an ext-vector ELEMENT lvalue under bit_cast at the vector's
                                //  base -- all four became dword 0, and the load shrank to one dword)
                                const bf16x8 t = __builtin_bit_cast(bf16x8, bu);
                                const bf16x2 v0 = __builtin_shufflevector(t, t, 0, 1), v1 = __builtin_shufflevector(t, t, 2, 3);
                                const bf16x2 v2 = __builtin_shufflevector(t, t, 4, 5), v3 = __builtin_shufflevector(t, t, 6, 7);
                                ssq = __builtin_amdgcn_fdot2_f32_bf16(v0, v0, ssq, false);
                                ssq = __builtin_amdgcn_fdot2_f32_bf16(v1, v1, ssq, false);
                                ssq = __builtin_amdgcn_fdot2_f32_bf16(v2, v2, ssq, false);
                                ssq = __builtin_amdgcn_fdot2_f32_bf16(v3, v3, ssq, false);
                                if (MODE >= 2) {
                                    ssum = __builtin_amdgcn_fdot2_f32_bf16(v0, ones, ssum, false);
                                    ssum = __builtin_amdgcn_fdot2_f32_bf16(v1, ones, ssum, false);
                                    ssum = __builtin_amdgcn_fdot2_f32_bf16(v2, ones, ssum, false);
                                    ssum = __builtin_amdgcn_fdot2_f32_bf16(v3, ones, ssum, false);
                                }
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                const int nr = (KS4 + NR - 1) / NR;
                // three rounds of A fragments in flight (3 x NR x 1 KiB per wave, 96 KiB per CU)
                auto run = [&](auto mode_tag, auto tail_tag) {
                    issue(b0, 0);
                    if (nr > 1) issue(b1, 1);
                    if (nr > 2) issue(b2, 2);
                    __builtin_amdgcn_sched_barrier(0);
                    for (int r = 0; r < nr; r += 3) {
                        compute(b0, r, mode_tag, tail_tag);
                        if (r + 3 < nr) issue(b0, r + 3);
                        __builtin_amdgcn_sched_barrier(0);
                        if (r + 1 < nr) {
                            compute(b1, r + 1, mode_tag, tail_tag);
                            if (r + 4 < nr) issue(b1, r + 4);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (r + 2 < nr) {
                            compute(b2, r + 2, mode_tag, tail_tag);
                            if (r + 5 < nr) issue(b2, r + 5);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                };
                const int mode = lnf ? 3 : (!dln ? 0 : (ones_col ? 1 : 2));
                if (KS4 % NR == 0) {
                    if (mode == 0) run(std::integral_constant<int, 0>{}, std::false_type{});
                    else if (mode == 1) run(std::integral_constant<int, 1>{}, std::false_type{});
                    else if (mode == 2) run(std::integral_constant<int, 2>{}, std::false_type{});
                    else run(std::integral_constant<int, 3>{}, std::false_type{});
                } else {                                   // small or odd K (the benchmark shapes never come here): one generic body
                    if (mode == 1) run(std::integral_constant<int, 1>{}, std::true_type{});
                    else if (mode == 3) run(std::integral_constant<int, 3>{}, std::true_type{});
                    else run(std::integral_constant<int, 2>{}, std::true_type{});
                }
                if (dln) {
                    ssum += __shfl_xor(ssum, 32, 64);
                    ssq += __shfl_xor(ssq, 32, 64);
                    if (lane < 32) { stat[(kg * 64 + mh * 32 + lane) * 2] = ssum; stat[(kg * 64 + mh * 32 + lane) * 2 + 1] = ssq; }
                }
                // C/D map of the 32 x 32 MFMA: column (lane & 31) = row m of the output, register r = column n = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
                float* my = red + kg * RED_KG + (mh * 32 + (lane & 31)) * 36 + 4 * (lane >> 5);
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 v = {acc[4 * gq], acc[4 * gq + 1], acc[4 * gq + 2], acc[4 * gq + 3]};
                    *reinterpret_cast<f32x4*>(my + 8 * gq) = v;
                }
            }
            stamp(p, 2);
            cbar();                                      // S3: partial tiles and row sums are in LDS; the ring slots of this tile are free
            stamp(p, 3);
            // thread (row m, column group g, half h4): 4 columns.  The two halves of a granule sit in adjacent lanes (wave = (g, row block),
            // lane = 2 (m % 32) + h4): bf16 outputs are exchanged with one lane swap and leave as ONE 16-byte store per granule (8-byte
            // write-through stores into lines that other CUs also write cost 3-5x the time: stamps, profiles/r05_micro_persist.txt)
            const int g = wave >> 1, h4 = lane & 1, m = (wave & 1) * 32 + (lane >> 1);
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            const bool mine = g < ng && m < M;
            if (mine) {
                const float* rp = red + m * 36 + g * 8 + h4 * 4;
                f32x4 sv = *reinterpret_cast<const f32x4*>(rp);
#pragma unroll
                for (int k = 1; k < 4; ++k) sv += *reinterpret_cast<const f32x4*>(rp + k * RED_KG);
                v[0] = sv[0]; v[1] = sv[1]; v[2] = sv[2]; v[3] = sv[3];
            }
            if (k4) {
                // ---- K-split: the partial goes to the quad's slab; three of the four CUs then finish 8 columns each
                const int q = cu >> 2;
                const auto rsS = rsrc_of(a.slabs);
                if (mine) st16_sc1(rsS, ((((long long)(q * 3 + g) * 4 + kq) * 64 + m) * 8 + h4 * 4) * 4, __builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]}));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const bool fin = kq < ng;                // this CU finishes column group g0 + kq (wave 0: one thread per row)
                unsigned* qc = a.counters + (64 + q) * 32;
                if (last_wave(drained + 1, 8u * (unsigned)ph.k4_epoch) && lane == 0)      // this CU's partial is out: the quad's counter
                    __hip_atomic_fetch_add(qc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (fin && wave == 0 && !spin_until(qc, 4u * (unsigned)ph.k4_epoch, lane == 0)) give_up(p);
                if (fin && wave == 0 && !*flag && tid < M) {   // thread = row: the group's 8 columns, partials summed in quarter order
                    const int mm = tid;
                    u32x4 pv[8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const long long o = ((((long long)(q * 3 + kq) * 4 + k) * 64 + mm) * 8) * 4;
                        pv[2 * k] = ld16_sc1(rsS, o);
                        pv[2 * k + 1] = ld16_sc1(rsS, o + 16);
                    }
                    f32x4 lo = __builtin_bit_cast(f32x4, pv[0]), hi = __builtin_bit_cast(f32x4, pv[1]);
#pragma unroll
                    for (int k = 1; k < 4; ++k) { lo += __builtin_bit_cast(f32x4, pv[2 * k]); hi += __builtin_bit_cast(f32x4, pv[2 * k + 1]); }
                    float* xr = xs + mm * 8;
                    float o[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] = xr[e] + (lo[e] + cb[kq * 8 + e]); o[4 + e] = xr[4 + e] + (hi[e] + cb[kq * 8 + 4 + e]); }
#pragma unroll
                    for (int e = 0; e < 8; ++e) xr[e] = o[e];
                    st16_sc1(rsrc_of(ph.out), packed_off(mm, (g0 + kq) * 8, MB) * 2, __builtin_bit_cast(u32x4, pack8(o)));
                }
            } else {
                if (mine && lnf) {                       // the residual stream itself: x <- (x - mean) rstd gamma + (beta + shift), fp32 master in place
                    float s = 0.0f, qq = 0.0f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { s += stat[(k * 64 + m) * 2]; qq += stat[(k * 64 + m) * 2 + 1]; }
                    const float mean = s / (float)ph.K;
                    const float rstd = 1.0f / sqrtf(fmaxf(qq / (float)ph.K - mean * mean, 0.0f) + 1e-5f);
                    float* xr = xs + m * 8 + h4 * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = (xr[e] - mean) * rstd * cb[32 + g * 8 + h4 * 4 + e] + cb[g * 8 + h4 * 4 + e]; xr[e] = v[e]; }
                } else if (mine) {
                    if (dln) {
                        float s = 0.0f, qq = 0.0f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) { s += ones_col ? red[k * RED_KG + m * 36 + 31] : stat[(k * 64 + m) * 2]; qq += stat[(k * 64 + m) * 2 + 1]; }
                        const float mean = s / (float)ph.K;
                        const float var = fmaxf(qq / (float)ph.K - mean * mean, 0.0f);
                        const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = rstd * (v[e] - mean * cb[32 + g * 8 + h4 * 4 + e]);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += cb[g * 8 + h4 * 4 + e];
                    if (ph.type == PP_RESID) {
                        float* xr = xs + m * 8 + h4 * 4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] += xr[e]; xr[e] = v[e]; }
                    } else if (ph.type == PP_GELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = persist_act(v[e], ph.act);
                    }
                }
                const int ncol = n0 + g * 8;             // first column of the granule
                if (ph.type == PP_ROWS) {                // fp32 rows (logits): each lane's 4 columns are 16 bytes
                    if (mine) st16_sc1(rsrc_of(ph.out), ((long long)m * ph.N + ncol + h4 * 4) * 4, __builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]}));
                } else {
                    const u32x2 pk = pack4(v);
                    const unsigned ox = __shfl_xor(pk.x, 1, 64), oy = __shfl_xor(pk.y, 1, 64);
                    if (mine && h4 == 0) {
                        const u32x4 gr = {pk.x, pk.y, ox, oy};
                        if (ph.type == PP_QKV) {
                            const int part = ncol / D, nn = ncol - part * D;
                            if (part == 0) st16_sc1(rsrc_of(ph.out), ((long long)m * D + nn) * 2, gr);
                            else st16_sc1(rsrc_of(part == 1 ? ph.kc : ph.vc), (((long long)m * ph.cache_T + t_cur) * D + nn) * 2, gr);
                        } else if (ph.type == PP_KV1) {
                            const int part = ncol / D, nn = ncol - part * D;          // 0: key, 1: value
                            st16_sc1(rsrc_of(part == 0 ? ph.kc : ph.vc), (((long long)m * ph.cache_T + ph.kv_row) * D + nn) * 2, gr);
                            if (part == 1) st16_sc1(rsrc_of(ph.vpk), packed_off(m, nn, MB) * 2, gr);
                        } else {                         // PP_RESID / PP_GELU: the packed operand of the next GEMM
                            st16_sc1(rsrc_of(ph.out), packed_off(m, ncol, MB) * 2, gr);
                        }
                    }
                }
            }
            if (!lnf) unit0 = (unit0 + ng * (Kp >> 6)) % RING_UNITS;
        } else {
            // ---- PP_ATTN: one query per (sample, head) over the cached keys + this step's (layers.py:93-102: scale on K, fp32 softmax).
            // A unit = (head h, 8 consecutive samples): wave w serves sample 8 bb + w, so the unit's output is hs / 8 whole 128-byte
            // lines of the packed layout.
            const int c = ac, slot = aslot;
            const int nkeys = t_cur + 1;
            const auto rsQ = rsrc_of(ph.A), rsK = rsrc_of(ph.kc), rsV = rsrc_of(ph.vc);
            char* const stage = lds + RED_OFF;
            const float scale = 1.0f / sqrtf((float)hs);
            int it = 0;
            for (int u = cu; u < nunits; u += ncu, ++it) {
                const int h = u / nb8, bb = u - h * nb8;
                const int b = bb * 8 + wave;
                float acc[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
                if (b < M) {
                    const long long kvbase = ((long long)b * ph.cache_T) * D + h * hs + c * 8;
                    float qv[8];
                    const u32x4 qraw = ld16_sc1(rsQ, ((long long)b * D + h * hs + c * 8) * 2);
                    float run_max = -INFINITY, run_sum = 0.0f;
                    // NP passes of rpp keys from j0 on, their rows in kbuf / vbuf
                    auto group = [&](int j0, auto np_tag, u32x4* kbuf, u32x4* vbuf) {
                        constexpr int NP = decltype(np_tag)::value;
                        __builtin_amdgcn_sched_barrier(0);
                        if (j0 == 0) unpack8(qraw, qv);
                        float sc[NP];
                        float gmax = -INFINITY;
#pragma unroll
                        for (int q = 0; q < NP; ++q) {
                            float kv[8];
                            unpack8(kbuf[q], kv);
                            float s = 0.0f;
#pragma unroll
                            for (int i = 0; i < 8; ++i) s = fmaf(qv[i], kv[i] * scale, s);
                            for (int off = chunks >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
                            sc[q] = (j0 + q * rpp + slot < nkeys) ? s : -INFINITY;
                            gmax = fmaxf(gmax, sc[q]);
                        }
                        for (int off = chunks; off < 64; off <<= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, off, 64));
                        const float new_max = fmaxf(run_max, gmax);
                        const float rescale = __expf(run_max - new_max);      // (FAST precision: v_exp_f32 forms)
                        float gsum = 0.0f;
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc[i] *= rescale;
#pragma unroll
                        for (int q = 0; q < NP; ++q) {
                            const float e = __expf(sc[q] - new_max);
                            gsum += e;
                            float vv[8];
                            unpack8(vbuf[q], vv);
#pragma unroll
                            for (int i = 0; i < 8; ++i) acc[i] = fmaf(e, vv[i], acc[i]);
                        }
                        for (int off = chunks; off < 64; off <<= 1) gsum += __shfl_xor(gsum, off, 64);
                        run_sum = run_sum * rescale + gsum;
                        run_max = new_max;
                    };
                    const int npass = (nkeys + rpp - 1) / rpp;
                    // a group whose rows are fetched here (write-through rows of this launch included: sc1)
                    auto fetched = [&](int j0, auto np_tag) {
                        constexpr int NP = decltype(np_tag)::value;
                        u32x4 kbuf[NP], vbuf[NP];
#pragma unroll
                        for (int q = 0; q < NP; ++q) {
                            const int j = min(j0 + q * rpp + slot, nkeys - 1);
                            kbuf[q] = ld16_sc1(rsK, (kvbase + (long long)j * D) * 2);
                            vbuf[q] = ld16_sc1(rsV, (kvbase + (long long)j * D) * 2);
                        }
                        group(j0, np_tag, kbuf, vbuf);
                    };
                    int p0 = 0;
                    for (; p0 < npass; p0 += 8) {
                        const int j0 = p0 * rpp;
                        switch (min(8, npass - p0)) {
                        case 1: fetched(j0, std::integral_constant<int, 1>{}); break;
                        case 2: fetched(j0, std::integral_constant<int, 2>{}); break;
                        case 3: fetched(j0, std::integral_constant<int, 3>{}); break;
                        case 4: fetched(j0, std::integral_constant<int, 4>{}); break;
                        case 5: fetched(j0, std::integral_constant<int, 5>{}); break;
                        case 6: fetched(j0, std::integral_constant<int, 6>{}); break;
                        case 7: fetched(j0, std::integral_constant<int, 7>{}); break;
                        default: fetched(j0, std::integral_constant<int, 8>{}); break;
                        }
                    }
                    for (int off = chunks; off < 64; off <<= 1)
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc[i] += __shfl_xor(acc[i], off, 64);
                    const float inv = 1.0f / run_sum;
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] *= inv;
                }
                if (slot == 0) *reinterpret_cast<uint4*>(stage + (size_t)it * (hs * 16) + (c * 8 + wave) * 16) = pack8(acc);
            }
            stamp(p, 2);
            cbar();                                      // S3
            stamp(p, 3);
            {
                const auto rsO = rsrc_of(ph.out);
                const int nit = cu < nunits ? (nunits - cu + ncu - 1) / ncu : 0;
                for (int i2 = wave; i2 < nit; i2 += 8) {
                    const int u = cu + i2 * ncu;
                    const int h = u / nb8, bb = u - h * nb8;
                    if (lane < chunks * 8) {
                        const int cc = lane >> 3, w = lane & 7, m = bb * 8 + w;
                        if (m < M) st16_sc1(rsO, packed_off(m, h * hs + cc * 8, MB) * 2, *reinterpret_cast<const u32x4*>(stage + (size_t)i2 * (hs * 16) + lane * 16));
                    }
                }
            }
        }
        stamp(p, 4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores ...
        stamp(p, 5);
        if (last_wave(drained, 8u * (unsigned)(p + 1)) && p + 1 < a.n_phases && lane < 8 && !(fault == (unsigned)cu + 1u && p == 0))   // ... and the last one to have done so signals
            __hip_atomic_fetch_add(a.counters + ((cu & 7) * 8 + lane) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stamp(p, 6);
    }
    if (a.write_back && rng && threadIdx.x < 64 && (int)threadIdx.x < M) {
        float* dst = a.x32 + (size_t)threadIdx.x * D + rg0 * 8;
        *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(xs + threadIdx.x * 8);
        *reinterpret_cast<f32x4*>(dst + 4) = *reinterpret_cast<const f32x4*>(xs + threadIdx.x * 8 + 4);
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
bool persist_program_ok(const std::vector<PersistPhase>& phases, int D, int M, int n_heads, int ncu) {
    if (ncu < 8 || ncu % 4 != 0 || ncu > 256 || D % 8 != 0 || (D / 8 + 2) / 3 > ncu / 4 || phases.empty() || phases.size() > 128 || M < 1 || M > 64) return false;
    if (n_heads < 1 || D % n_heads != 0) return false;
    const int hs = D / n_heads;
    if (hs % 8 != 0 || hs > 512 || ((hs / 8) & (hs / 8 - 1)) != 0) return false;
    int k4 = 0;
    for (const PersistPhase& ph : phases) {
        if (ph.type == PP_ATTN) {
            const int nunits = n_heads * ((M + 7) / 8);
            if ((size_t)((nunits + ncu - 1) / ncu) * hs * 16 > (size_t)RED_BYTES) return false;
            continue;
        }
        if (ph.N % 8 != 0 || ph.K % 64 != 0 || ph.N <= 0 || ph.K <= 0) return false;
        if ((ph.type == PP_RESID || ph.type == PP_LNF) && (ph.N != D || ph.map != PP_MAP_QUAD)) return false;
        if (ph.type == PP_LNF && (ph.K != D || !ph.dln)) return false;
        if (ph.type == PP_RESID_K4 && (ph.N != D || ph.map != PP_MAP_K4 || ph.K % 256 != 0 || ph.k4_epoch != ++k4)) return false;
        if (ph.type != PP_RESID && ph.type != PP_RESID_K4 && ph.type != PP_LNF && ph.map != PP_MAP_EVEN) return false;
        if ((ph.type == PP_QKV || ph.type == PP_KV1) && (ph.N % D != 0)) return false;
        for (int c = 0; c < ncu; ++c) {
            int g0, ng, kq;
            persist_cols(ph.N, ph.rot, ph.map, c, ncu, &g0, &ng, &kq);
            if (ng * 8 > PERSIST_MAX_NC || persist_tile_units(ph, c, ncu) > PERSIST_RING_UNITS) return false;
        }
    }
    return true;
}

size_t persist_layout(const std::vector<PersistPhase>& phases, int ncu, std::vector<unsigned long long>& cu_off, std::vector<unsigned long long>& tile_off) {
    cu_off.assign(ncu, 0);
    tile_off.assign(phases.size() * (size_t)ncu, 0);
    unsigned long long run = 0;
    for (int c = 0; c < ncu; ++c) {
        cu_off[c] = run;
        for (size_t p = 0; p < phases.size(); ++p) {
            tile_off[p * ncu + c] = run;
            run += (unsigned long long)persist_tile_units(phases[p], c, ncu) * 1024ull;
        }
    }
    return (size_t)run;
}

__global__ void persist_pack_kernel(const float* __restrict__ w, const float* __restrict__ gamma, int N, int K, int rot, int map, int ncu,
                                    char* __restrict__ stream, const unsigned long long* __restrict__ tile_off) {
    const int K8 = K >> 3, G = N >> 3;
    const size_t total = (size_t)N * K8;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / K8);
        int kk = (int)(i % K8);                            // 8-column block of k
        const int gi = n >> 3;
        int cu, j, ng;
        if (map == PP_MAP_QUAD) {
            cu = 4 * (gi / 3) + gi % 3; j = 0; ng = 1;
        } else if (map == PP_MAP_K4) {
            const int q = gi / 3, K8q = K8 >> 2, kq = kk / K8q;
            cu = 4 * q + kq; j = gi - 3 * q; ng = min(3, G - 3 * q);
            kk -= kq * K8q;
        } else {
            int c;
            if (G >= ncu) {
                const int base = G / ncu, extra = G % ncu;
                if (gi < extra * (base + 1)) { c = gi / (base + 1); j = gi - c * (base + 1); ng = base + 1; }
                else { const int r = gi - extra * (base + 1); c = extra + r / base; j = r - (r / base) * base; ng = base; }
            } else { c = gi; j = 0; ng = 1; }
            cu = (c + rot) % ncu;
        }
        const int nc = ng * 8, nl = j * 8 + (n & 7);
        const int ks = kk >> 1, half = kk & 1;
        const int k0 = (int)(i % K8) * 8;
        const float* src = w + (size_t)n * K + k0;
        unsigned pk[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ga = gamma ? gamma[k0 + 2 * e] : 1.0f, gb = gamma ? gamma[k0 + 2 * e + 1] : 1.0f;
            pk[e] = (unsigned)f32_to_bf16(src[2 * e] * ga) | ((unsigned)f32_to_bf16(src[2 * e + 1] * gb) << 16);
        }
        uint4* dst = reinterpret_cast<uint4*>(stream + tile_off[cu] + ((size_t)(ks * 2 + half) * nc + nl) * 16);
        *dst = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    }
}
hipError_t launch_persist_pack(const float* w, const float* gamma, const PersistPhase& ph, int ncu, char* stream, const unsigned long long* d_tile_off, hipStream_t st) {
    const size_t n = (size_t)ph.N * (ph.K / 8);
    persist_pack_kernel<<<(int)std::min<size_t>((n + 255) / 256, 8192), 256, 0, st>>>(w, gamma, ph.N, ph.K, ph.rot, ph.map, ncu, stream, d_tile_off);
    return hipGetLastError();
}

hipError_t persist_configure() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(persist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
}
int persist_blocks_per_cu() {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(persist_kernel), 576, LDS_BYTES) != hipSuccess) return 0;
    return n;
}
// zeroes the barrier / quad counters in front of every launch.  A KERNEL, not hipMemsetAsync: inside a captured graph replayed on a stream other than the
// null stream the memset node did not order the way a kernel node does (round 6, tools/diag_two_handles.py: a 16-position graph of the persistent position
// replayed on a side stream drew different codes than on the null stream -- 19 % of them -- with nothing else running; one position per graph, eager launches
// and the launch chain were fine).
__global__ __launch_bounds__(256) void persist_zero_kernel(unsigned* counters) {
    for (int i = threadIdx.x; i < PERSIST_COUNTER_BYTES / 4; i += 256) counters[i] = 0u;
}
hipError_t launch_persist(const PersistArgs& a, int ncu, hipStream_t st) {
    persist_zero_kernel<<<1, 256, 0, st>>>(a.counters);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    persist_kernel<<<ncu, 576, LDS_BYTES, st>>>(a);
    return hipGetLastError();
}
