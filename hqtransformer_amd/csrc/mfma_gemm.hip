// K7/K9: LDS-tiled bf16 MFMA GEMM / implicit-GEMM convolution (FAST precision).
//
//   C[b][m][n] = alpha * sum_k A[b][m][k] * B[b][n][k]   (+bias, act, +residual), fp32 accumulation.
//
// A is either a plain row-major bf16 matrix or the im2col view of an NHWC bf16 tensor (1x1 / 3x3
// 'same' conv, nearest-x2 upsample folded into the addressing, GroupNorm+swish applied in registers
// between the global load and the LDS write), B is a K-contiguous bf16 matrix (filters repacked
// tap-major at finalize, or a second activation for the decoder attention).
//
// BM x BN output tile per 256-thread workgroup (4 waves as 2 x 2, each wave (BM/2) x (BN/2) in 32x32x16
// MFMA tiles), BK = 32.  Both operands are staged global -> registers -> LDS with the next k-tile's
// global loads issued before the current tile's MFMAs and written to the other LDS buffer after them
// (one barrier per k-tile).  LDS rows are padded to 80 B so that the 16-B fragment reads of 16
// different rows fall on 16 different 16-B bank slots (conflict-free ds_read_b128).
#include "fast_kernels.h"
#include "gemm_generic.h"
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {
constexpr int BK = 32;
constexpr int LDS_ROW = 40;            // bf16 elements per LDS row (32 + 8 pad = 80 B)

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    return (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
}

struct RowCtx {                        // one A row handled by this thread's loader slot
    bool ok;
    int img, y, x;
    const bf16_t* row;
};

__device__ __forceinline__ RowCtx make_row(const GemmArgs& g, const bf16_t* base, int m) {
    RowCtx r;
    r.ok = m < g.M;
    r.img = r.y = r.x = 0;
    r.row = base;
    if (!r.ok) return r;
    if (g.conv_taps) {
        const int hw = g.H * g.W;
        r.img = m / hw;
        const int rem = m - r.img * hw;
        r.y = rem / g.W;
        r.x = rem - r.y * g.W;
    } else {
        int ar = m;
        if (g.a_rows_per_group > 0) ar = (m / g.a_rows_per_group) * g.a_group_stride + m % g.a_rows_per_group + g.a_row_offset;
        r.row = base + (long long)ar * g.lda;
    }
    return r;
}

// raw 16-byte load of 8 consecutive channels (c..c+7) of filter tap `tap` for one A row; the k-tile
// lies inside a single tap, so (tap, c) are workgroup-uniform up to the per-thread 8-channel offset
__device__ __forceinline__ u32x4 load_a_raw(const GemmArgs& g, const bf16_t* base, const RowCtx& r, int k, int tap, int c, bool& inb) {
    u32x4 z = {0u, 0u, 0u, 0u};
    inb = false;
    if (!r.ok) return z;
    if (!g.conv_taps) { inb = true; return *reinterpret_cast<const u32x4*>(r.row + k); }
    int iy = r.y, ix = r.x;
    if (g.conv_taps == 9) { const int t3 = tap / 3; iy += t3 - 1; ix += tap - 3 * t3 - 1; }
    if ((unsigned)iy >= (unsigned)g.H || (unsigned)ix >= (unsigned)g.W) return z;
    inb = true;
    const int Win = g.W >> g.upsample;
    const long long pix = ((long long)r.img * (g.H >> g.upsample) + (iy >> g.upsample)) * Win + (ix >> g.upsample);
    return *reinterpret_cast<const u32x4*>(base + pix * g.Cin + c);
}

// GroupNorm (+swish) on 8 packed bf16 values of channels c..c+7 of image img
__device__ __forceinline__ u32x4 gn_apply(const GemmArgs& g, u32x4 v, int img, int c) {
    const int cpg = g.Cin / g.gn_groups;
    float f[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = bf16_to_f32((bf16_t)(v[i] & 0xffffu));
        f[2 * i + 1] = bf16_to_f32((bf16_t)(v[i] >> 16));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float* st = g.gn_stats + ((long long)img * g.gn_groups + (c + i) / cpg) * 2;
        float t = (f[i] - st[0]) * st[1] * g.gn_gamma[c + i] + g.gn_beta[c + i];
        if (g.gn_swish) t = t / (1.0f + __expf(-t));
        f[i] = t;
    }
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = pack2(f[2 * i], f[2 * i + 1]);
    return o;
}
}  // namespace

// Workgroups are dealt round-robin over the 8 XCDs (ids b and b + 8 share one).  Remap the linear id so
// that every XCD owns a contiguous run of tiles in n-fastest order: the ~64 workgroups resident on one XCD
// then cover a few m-tiles x all n-tiles and march through K together, so each A and B k-slice is fetched
// into that XCD's L2 once and reused by all of them (speed only; any placement is correct).
__device__ __forceinline__ void xcd_tile(int& tile_m, int& tile_n) {
    const int nx = gridDim.x, total = gridDim.x * gridDim.y;
    int id = blockIdx.x + nx * blockIdx.y;
    if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
    tile_m = id / nx;
    tile_n = id - tile_m * nx;
}

template <int BM, int BN, typename TC>
__global__ __launch_bounds__(256) void mfma_gemm_kernel(GemmArgs g) {
    constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 32, NT = WN / 32;
    constexpr int A_SLOTS = BM * 4 / 256, B_SLOTS = BN * 4 / 256;     // 16-byte chunks per thread per k-tile
    __shared__ __attribute__((aligned(16))) bf16_t As[2][BM * LDS_ROW];
    __shared__ __attribute__((aligned(16))) bf16_t Bs[2][BN * LDS_ROW];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bz = blockIdx.z;
    int tile_m, tile_n;
    xcd_tile(tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const bf16_t* Abase = reinterpret_cast<const bf16_t*>(g.A) + (long long)bz * g.a_batch_stride;
    const bf16_t* Bbase = reinterpret_cast<const bf16_t*>(g.Bw) + (long long)bz * g.b_batch_stride;

    RowCtx arow[A_SLOTS];
    int a_lrow[A_SLOTS], a_kc[A_SLOTS];
#pragma unroll
    for (int s = 0; s < A_SLOTS; ++s) {
        const int c = tid + s * 256;
        a_lrow[s] = c >> 2;
        a_kc[s] = (c & 3) * 8;
        arow[s] = make_row(g, Abase, m0 + a_lrow[s]);
    }
    const bf16_t* brow[B_SLOTS];
    bool b_ok[B_SLOTS];
    int b_lrow[B_SLOTS], b_kc[B_SLOTS];
#pragma unroll
    for (int s = 0; s < B_SLOTS; ++s) {
        const int c = tid + s * 256;
        b_lrow[s] = c >> 2;
        b_kc[s] = (c & 3) * 8;
        b_ok[s] = n0 + b_lrow[s] < g.N;
        brow[s] = Bbase + (long long)(b_ok[s] ? n0 + b_lrow[s] : 0) * g.ldb;
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    u32x4 areg[A_SLOTS], breg[B_SLOTS];
    bool a_inb[A_SLOTS];
    const int cpt = g.conv_taps ? g.Cin / BK : 1;          // k-tiles per filter tap
    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        const int tap = g.conv_taps ? kt / cpt : 0;         // workgroup-uniform (scalar) arithmetic
        const int c0 = k0 - tap * g.Cin;
#pragma unroll
        for (int s = 0; s < A_SLOTS; ++s) areg[s] = load_a_raw(g, Abase, arow[s], k0 + a_kc[s], tap, c0 + a_kc[s], a_inb[s]);
#pragma unroll
        for (int s = 0; s < B_SLOTS; ++s) {
            const u32x4 z = {0u, 0u, 0u, 0u};
            breg[s] = b_ok[s] ? *reinterpret_cast<const u32x4*>(brow[s] + k0 + b_kc[s]) : z;
        }
    };
    auto write_tile = [&](int kt, int buf) {
        const int k0 = kt * BK;
#pragma unroll
        for (int s = 0; s < A_SLOTS; ++s) {
            u32x4 v = areg[s];
            if (g.gn_stats && a_inb[s]) {
                const int k = k0 + a_kc[s];
                v = gn_apply(g, v, arow[s].img, k - (k / g.Cin) * g.Cin);
            }
            *reinterpret_cast<u32x4*>(&As[buf][a_lrow[s] * LDS_ROW + a_kc[s]]) = v;
        }
#pragma unroll
        for (int s = 0; s < B_SLOTS; ++s) *reinterpret_cast<u32x4*>(&Bs[buf][b_lrow[s] * LDS_ROW + b_kc[s]]) = breg[s];
    };

    const int KT = g.K / BK;
    load_tile(0);
    write_tile(0, 0);
    __syncthreads();
    const int fr = lane & 31, fh = lane >> 5;
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KT) load_tile(kt + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[MT], bfr[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(&As[buf][(wm * WM + i * 32 + fr) * LDS_ROW + ks * 16 + fh * 8]);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                bfr[j] = *reinterpret_cast<const bf16x8*>(&Bs[buf][(wn * WN + j * 32 + fr) * LDS_ROW + ks * 16 + fh * 8]);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < KT) write_tile(kt + 1, buf ^ 1);
        __syncthreads();
    }
    // epilogue: D col = lane & 31 -> n, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -> m
    const bool plain = g.store == STORE_ROWS && g.rows_per_group == 0;
    TC* Cb = reinterpret_cast<TC*>(g.C) + (long long)bz * g.c_batch_stride;
    const TC* Rb = g.resid ? reinterpret_cast<const TC*>(g.resid) + (long long)bz * g.c_batch_stride : nullptr;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + wn * WN + j * 32 + fr;
            if (n >= g.N) continue;
            const float bn = g.bias ? g.bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (m >= g.M) continue;
                if (plain) {
                    const long long idx = (long long)m * g.ldc + n;
                    float v = apply_act(acc[i][j][r] * g.alpha + bn, g.act);
                    if (Rb) v += ld1<TC>(Rb + idx);
                    st1<TC>(Cb + idx, v);
                } else {
                    gemm_store<TC>(g, bz, m, n, acc[i][j][r]);
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------
// 128 x 128 x 64 tile, operands staged by LDS-DMA (global_load_lds_dwordx4: no VGPR staging), two
// LDS buffers, the next k-tile's DMA in flight under the current tile's 16 MFMAs per wave.
// LDS image per operand: [128 rows][8 chunks of 16 B]; chunk c of row r lives in slot c ^ ((r >> 1) & 7).
// LDS-DMA writes lane-linearly (base + 16 lane), so the swizzle is applied to the SOURCE address: the
// lane that fills (row, slot) fetches chunk slot ^ ((row >> 1) & 7) of that row; readers apply the same
// XOR.  With 128-B rows two rows share a 256-B bank line and 16 rows of one ds_read_b128 group land on 16
// distinct 16-B slots.  Zero padding of the 3x3 taps comes from a zero page (DMA cannot predicate).
// ---------------------------------------------------------------------------------------------
// NSTAGE = 2: two LDS buffers, plain __syncthreads() (hipcc drains the DMA in front of it), 2 workgroups per CU.
// NSTAGE = 4: four-slot LDS ring (128 KiB, 1 workgroup per CU) with two k-tiles of DMA in flight across the barrier:
//   counted s_waitcnt vmcnt(8 * tiles_in_flight) + raw s_barrier per k-tile; the slot of tile kt-1 is refilled right
//   after the barrier that every wave passes only once it has finished reading that tile.
// BM_ = 64: 64 x 128 tiles (4 waves side by side, 64 x 32 each) for GEMMs whose 128 x 128 grid would leave most CUs idle
// (the 1024-row narrow GEMMs of the third code level: 96 -> 192 workgroups); two LDS buffers only.
template <typename TC, int MODE, int NSTAGE, int ABL = 0, int BM_ = 128>   // MODE 0: plain rows, 1: NCHW, 2: any store mode (element-wise), 3: fused QKV, 4 columns at a time; ABL: tools/micro/bench_conv
__global__ __launch_bounds__(256, NSTAGE == 2 ? 2 : 1) void conv_glds_kernel(GemmArgs g) {
    constexpr bool NCHW = MODE == 1;
    static_assert(BM_ == 128 || (BM_ == 64 && NSTAGE == 2), "64-row tiles run with two LDS buffers");
    constexpr int BM = BM_, BN = 128, BKG = 64, ROWB = 128;            // ROWB: bytes per LDS row
    constexpr int WNW = BM == 128 ? 64 : 32, FJ = WNW / 32, APW = BM / 32;   // columns per wave, 32-column fragments per wave, A DMA pieces per wave
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];     // [NSTAGE][A|B][BM * ROWB]
    auto LDS = [&](int stage, int op) -> char* { return lds_raw + (size_t)stage * ((BM + BN) * ROWB) + (op ? BM * ROWB : 0); };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = BM == 128 ? wave >> 1 : 0, wn = BM == 128 ? wave & 1 : wave;
    const int bz = blockIdx.z;
    int tile_m, tile_n;
    xcd_tile(tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const bf16_t* Abase = reinterpret_cast<const bf16_t*>(g.A) + (long long)bz * g.a_batch_stride;
    const bf16_t* Bbase = reinterpret_cast<const bf16_t*>(g.Bw) + (long long)bz * g.b_batch_stride;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g.zero_page);

    // DMA instruction i (0..3) of this wave fills rows (wave * 4 + i) * 8 .. + 7; lane -> (row, slot)
    RowCtx arow[APW];
    const bf16_t* brow[4];
    int chunk[4], achunk[APW];                                        // source chunk of this lane (elements: * 8)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        chunk[i] = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
        brow[i] = (n0 + row < g.N) ? Bbase + (long long)(n0 + row) * g.ldb : nullptr;
    }
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const int row = (wave * APW + i) * 8 + (lane >> 3);
        achunk[i] = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
        arow[i] = make_row(g, Abase, m0 + row);
    }
    const int cpt = g.conv_taps ? g.Cin / BKG : 1;
    const int Hv = g.H << g.conv_stride2, Wv = g.W << g.conv_stride2;      // input size as the filter sees it (Downsample: stride 2)
    const int Hin = Hv >> g.upsample, Win = Wv >> g.upsample;
    auto issue = [&](int kt, int buf) {
        const int k0 = kt * BKG;
        const int tap = g.conv_taps ? kt / cpt : 0;
        const int c0 = k0 - tap * g.Cin;
        int dy = 0, dx = 0;
        if (g.conv_taps == 9) { const int t3 = tap / 3; dy = t3 - 1 + g.conv_nopad; dx = tap - 3 * t3 - 1 + g.conv_nopad; }
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            const bf16_t* src = zero;
            if (arow[i].ok) {
                if (!g.conv_taps) src = arow[i].row + k0 + achunk[i];
                else {
                    const int iy = (arow[i].y << g.conv_stride2) + dy, ix = (arow[i].x << g.conv_stride2) + dx;
                    if ((unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv)
                        src = Abase + (((long long)arow[i].img * Hin + (iy >> g.upsample)) * Win + (ix >> g.upsample)) * g.Cin + c0 + achunk[i];
                }
            }
            char* dst = LDS(buf, 0) + (wave * APW + i) * 8 * ROWB;          // wave-uniform base; HW adds 16 * lane
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16_t* src = brow[i] ? brow[i] + k0 + chunk[i] : zero;
            char* dst = LDS(buf, 1) + (wave * 4 + i) * 8 * ROWB;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    f32x16 acc[2][FJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int KT = g.K / BKG;
    const int fr = lane & 31, fh = lane >> 5;
    auto compute = [&](int buf) {
        const char* Ab = LDS(buf, 0);
        const char* Bb = LDS(buf, 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int c = ks * 2 + fh;
            bf16x8 af[2], bfr[FJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wm * 64 + i * 32 + fr;
                af[i] = *reinterpret_cast<const bf16x8*>(Ab + r * ROWB + ((c ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                const int r = wn * WNW + j * 32 + fr;
                bfr[j] = *reinterpret_cast<const bf16x8*>(Bb + r * ROWB + ((c ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // D rows = n, cols = m
        }
    };
    if constexpr (NSTAGE == 2) {
        issue(0, 0);
        __syncthreads();                               // hipcc drains the DMA (vmcnt(0)) in front of the barrier
        for (int kt = 0; kt < KT; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < KT && ABL != 2) issue(ABL == 4 ? 1 : kt + 1, buf ^ 1);
            if (ABL != 3 || kt == 0) compute(buf);
            __syncthreads();
        }
    } else {
        for (int s0 = 0; s0 < NSTAGE - 1 && s0 < KT; ++s0) issue(s0, s0);
        for (int kt = 0; kt < KT; ++kt) {
            // tiles kt+1 .. kt+NSTAGE-2 may stay in flight; each tile is 8 DMA instructions per wave
            const int inflight = min(NSTAGE - 2, KT - 1 - kt);
            if (inflight >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (inflight == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (kt + NSTAGE - 1 < KT) issue(kt + NSTAGE - 1, (kt + NSTAGE - 1) % NSTAGE);
            compute(kt % NSTAGE);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's fragment reads are done before it re-arrives
        }
    }
    if (ABL == 1) {                                     // no epilogue: keep the accumulators alive, one store per wave
        float sacc = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < FJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
        if (sacc == 12345.678f) reinterpret_cast<float*>(g.C)[0] = sacc;
        return;
    }
    // epilogue.  The B operand (weights / second activation) is the MFMA A operand, so D col = lane & 31 -> row m of
    // fragment i and D row = (r & 3) + 8 (r >> 2) + 4 fh -> column n: a lane owns 4 consecutive columns of one row per
    // register quad and stores them with one 8-byte (bf16) / 16-byte (fp32) access; NCHW stores run along the lanes.
    TC* Cb = reinterpret_cast<TC*>(g.C) + (long long)bz * g.c_batch_stride;
    const TC* Rb = g.resid ? reinterpret_cast<const TC*>(g.resid) + (long long)bz * g.c_batch_stride : nullptr;
    const bool vec4 = (g.N & 3) == 0 && (g.ldc & 3) == 0;
    const int qkv_dev = (MODE == 3 && g.row_offset_dev) ? *g.row_offset_dev : 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 64 + i * 32 + fr;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < FJ; ++j)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int n4 = n0 + wn * WNW + j * 32 + 8 * q4 + 4 * fh;
                if (n4 >= g.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q4 + e] * g.alpha + ((g.bias && n4 + e < g.N) ? g.bias[n4 + e] : 0.0f);
                if (NCHW) {                       // fp32 / bf16 NCHW (+clamp): conv_out, V^T of the decoder attention
                    const int img = m / g.rows_per_image, pix = m - img * g.rows_per_image;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (n4 + e >= g.N) continue;
                        float x = v[e];
                        if (g.clamp01) x = fminf(fmaxf(0.5f * x + 0.5f, 0.0f), 1.0f);
                        st1<TC>(Cb + ((long long)img * g.N + n4 + e) * g.rows_per_image + pix, x);
                    }
                } else if (MODE == 3) {           // dispatcher guarantees: STORE_QKV, N, ldc, qkv_D multiples of 4
                    // fused [query; key; value] with the KV-cache row remap: the 4 columns lie in one part
                    const int part = n4 / g.qkv_D, nn = n4 - part * g.qkv_D, which = part + g.qkv_first;
                    long long row = m;
                    if (which > 0) row = (m / g.rows_per_group) * g.group_stride + m % g.rows_per_group + g.row_offset + qkv_dev;
                    TC* base = reinterpret_cast<TC*>(which == 0 ? g.C : (which == 1 ? g.C2 : g.C3));
                    st4<TC>(base + row * g.ldc + nn, v);
                } else if (MODE == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n4 + e < g.N) gemm_store<TC>(g, bz, m, n4 + e, acc[i][j][4 * q4 + e]);
                } else {
                    const long long idx = (long long)m * g.ldc + n4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], g.act);
                    if (vec4) {
                        if (Rb) {
                            float rv[4];
                            ld4<TC>(Rb + idx, rv);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += rv[e];
                        }
                        st4<TC>(Cb + idx, v);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n4 + e < g.N) { float x = v[e]; if (Rb) x += ld1<TC>(Rb + idx + e); st1<TC>(Cb + idx + e, x); }
                    }
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// 3x3 'same' convolution with the input patch resident in LDS ("halo tile").
// One workgroup = an 8 x 16 pixel tile of one image x 128 output channels.  Per 64-channel chunk the (8+2) x (16+2)
// input patch (180 rows of 128 B) is DMA'd into LDS ONCE and serves all nine taps: tap (dy, dx) of output pixel
// (py, px) is patch row (py + dy) * 18 + px + dx, so the im2col view never exists, not even as addresses.  Against
// the generic implicit GEMM above this cuts the LDS-DMA pieces per k-tile from 8 to 4.6 per wave (the measured
// bound of that kernel: tools/micro/bench_conv) and the address arithmetic to one add per chunk.  Zero padding and
// the nearest x2 upsample are folded into the patch rows' source addresses (zero page for rows outside the image).
// K order is chunk-major (c, tap); the filter matrix stays tap-major [O][tap][I], k0 = tap * Cin + 64 c.
// The weights are the MFMA A operand here (D rows = n, D cols = pixels): a lane then owns 4 consecutive output
// channels of one pixel per register quad, and the epilogue stages the fp32 tile through LDS (the dead patch/filter
// buffers) to write full 256-B NHWC rows with 16-B stores per lane instead of 64 two-byte stores.
// LDS: patch 2 x 184 rows + filters 2 x 128 rows, 128 B each = 78 KiB -> two workgroups per CU.
// ---------------------------------------------------------------------------------------------
constexpr int HALO_TX = 16, HALO_PITCH = HALO_TX + 2;
constexpr int halo_rows(int TY) { return (TY + 2) * HALO_PITCH; }                        // 180 | 324
constexpr int halo_pieces(int TY) { return (halo_rows(TY) + 7) / 8; }                   // 23 | 41 DMA pieces of 8 rows
constexpr int halo_patch_bytes(int TY) { return halo_pieces(TY) * 1024; }
constexpr int halo_npatch(int TY) { return TY == 8 ? 2 : 1; }                           // 16-row tiles keep ONE patch buffer (LDS)
constexpr int HALO_B_BYTES = 128 * 128;
constexpr int halo_lds(int TY, int BN = 128) { return halo_npatch(TY) * halo_patch_bytes(TY) + 2 * BN * 128; }   // 79872 | 74752: two workgroups per CU
constexpr int HALO_CPITCH = 128 * 4 + 16;                                              // fp32 staging row (bytes)
static_assert(128 * HALO_CPITCH <= halo_lds(8) && 128 * HALO_CPITCH <= halo_lds(16), "epilogue staging must fit in the operand buffers");

// TY = 8: 8 x 16 pixel tile (128 pixels, each wave 64 pixels x 64 channels), both operand kinds double-buffered.
// TY = 16: 16 x 16 pixel tile (256 pixels, each wave 128 x 64: twice the MFMAs per barrier and per DMA'd filter
//          byte, 0.75 instead of 1 fragment read per MFMA); the 324-row patch is single-buffered and re-filled
//          between channel chunks (the other workgroup of the CU covers that gap); the epilogue stages two halves.
// BN = 128 output channels per workgroup (waves 2 x 2), or BN = 32 for conv_out (3 channels: 4 waves along the pixels, one
// 32-channel fragment each -- 4x fewer MFMAs than padding 3 channels to 128; NCHW epilogue only).
template <typename TC, bool NCHW, int TY, int ABL = 0, int BN = 128>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(GemmArgs g) {
    static_assert(BN == 128 || (BN == 32 && NCHW), "the 32-channel variant exists for the NCHW conv_out store only");
    constexpr int WMW = BN == 128 ? 2 : 4, FJ = BN / (BN == 128 ? 64 : 32), B_BYTES = BN * 128;
    constexpr int ROWS = halo_rows(TY), PIECES = halo_pieces(TY), NPATCH = halo_npatch(TY), PATCH_BYTES = halo_patch_bytes(TY);
    constexpr int FI = (TY / 2) / WMW;                              // 32-pixel fragments per wave
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    auto PATCH = [&](int s) -> char* { return lds_raw + (size_t)s * PATCH_BYTES; };
    auto BT = [&](int s) -> char* { return lds_raw + NPATCH * PATCH_BYTES + (size_t)s * B_BYTES; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = BN == 128 ? wave >> 1 : wave, wn = BN == 128 ? wave & 1 : 0;
    const int fr = lane & 31, fh = lane >> 5;
    int tile_m, tile_n;
    xcd_tile(tile_m, tile_n);
    const int n0 = tile_n * BN;
    const int tiles_x = g.W / HALO_TX, tiles_y = g.H / TY;
    const int img = tile_m / (tiles_x * tiles_y);
    const int trem = tile_m - img * (tiles_x * tiles_y);
    const int ty0 = (trem / tiles_x) * TY, tx0 = (trem % tiles_x) * HALO_TX;
    const int Hin = g.H >> g.upsample, Win = g.W >> g.upsample;
    const bf16_t* Abase = reinterpret_cast<const bf16_t*>(g.A);
    const bf16_t* Bbase = reinterpret_cast<const bf16_t*>(g.Bw);
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g.zero_page);

    // ---- DMA sources.  Patch piece p = wave + 4 t fills patch rows 8 p .. 8 p + 7; lane -> (row q, slot lane & 7),
    //      source chunk = slot ^ ((q >> 1) & 7) (the same involution the fragment reads apply).  The addresses are
    //      rebuilt per channel chunk (once per nine k-tiles) instead of being held in registers.
    constexpr int PPW = (PIECES + 3) / 4;                           // pieces per wave
    constexpr int BPW = BN / 32;                                    // filter DMA pieces per wave (4 | 1)
    const bf16_t* brow[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const int row = (wave * BPW + i) * 8 + (lane >> 3);
        const int ch = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
        brow[i] = (n0 + row < g.N) ? Bbase + (long long)(n0 + row) * g.ldb + ch : nullptr;
    }
    auto issue_patch = [&](int c, int s) {
#pragma unroll
        for (int t = 0; t < PPW; ++t) {
            if (wave + 4 * t < PIECES) {                            // wave-uniform
                const int q = (wave + 4 * t) * 8 + (lane >> 3);
                const int qy = (q * 3641) >> 16, qx = q - qy * HALO_PITCH;          // q / 18 for q < 328
                const int iy = ty0 + qy - 1, ix = tx0 + qx - 1;
                const int ch = ((lane & 7) ^ ((q >> 1) & 7)) * 8;
                const bf16_t* src = (q < ROWS && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W)
                                        ? Abase + (((long long)img * Hin + (iy >> g.upsample)) * Win + (ix >> g.upsample)) * g.Cin + ch + c * 64
                                        : zero;
                char* dst = PATCH(s) + (wave + 4 * t) * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        }
    };
    auto issue_b = [&](int k0, int s) {
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            const bf16_t* src = brow[i] ? brow[i] + k0 : zero;
            char* dst = BT(s) + (wave * BPW + i) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    f32x16 acc[FI][FJ];                                             // [pixel block i][channel block j]
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // pixel of this lane in fragment i: tile row (wm TY/2 + 2 i + fr / 16, fr % 16); patch row of tap (0, 0)
    int qbase[FI];
#pragma unroll
    for (int i = 0; i < FI; ++i) qbase[i] = (wm * (2 * FI) + i * 2 + (fr >> 4)) * HALO_PITCH + (fr & 15);
    int brd[FJ];                                                    // filter fragment rows
#pragma unroll
    for (int j = 0; j < FJ; ++j) brd[j] = wn * 64 + j * 32 + fr;

    auto compute = [&](int ps, int bs, int tapoff) {
        const char* Pb = PATCH(ps);
        const char* Bb = BT(bs);
        int qa[FI], sw[FI];
#pragma unroll
        for (int i = 0; i < FI; ++i) { qa[i] = (qbase[i] + tapoff) * 128; sw[i] = ((qbase[i] + tapoff) >> 1) & 7; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int c = ks * 2 + fh;
            bf16x8 af[FI], bfr[FJ];
#pragma unroll
            for (int i = 0; i < FI; ++i) af[i] = *reinterpret_cast<const bf16x8*>(Pb + qa[i] + ((c ^ sw[i]) << 4));
#pragma unroll
            for (int j = 0; j < FJ; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(Bb + brd[j] * 128 + ((c ^ ((brd[j] >> 1) & 7)) << 4));
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    };

    float bv[8];                                        // epilogue bias of this thread's 8 channels, fetched under the main loop
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (g.bias && n0 + (tid & 15) * 8 + e < g.N) ? g.bias[n0 + (tid & 15) * 8 + e] : 0.0f;
    const int NC = g.Cin / 64, KT = NC * 9;
    issue_patch(0, 0);
    issue_b(0, 0);
    __syncthreads();                                   // hipcc drains the DMA (vmcnt(0)) in front of the barrier
    int kt = 0;
#pragma unroll 1
    for (int c = 0; c < NC; ++c) {
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap, ++kt) {               // rolled: unrolled nine-fold the LDS addresses alone cost > 100 VGPRs
            if (kt + 1 < KT && ABL != 2) {
                const int tn = tap == 8 ? 0 : tap + 1, cn = tap == 8 ? c + 1 : c;
                issue_b(tn * g.Cin + cn * 64, (kt + 1) & 1);
                if (NPATCH == 2 && tap == 0 && c + 1 < NC) issue_patch(c + 1, (c + 1) & 1);
            }
            const int t3 = (tap * 11) >> 5;            // tap / 3 for tap < 9
            if (ABL != 3 || kt == 0) compute(NPATCH == 2 ? (c & 1) : 0, kt & 1, t3 * HALO_PITCH + (tap - 3 * t3));
            __syncthreads();
        }
        if (NPATCH == 1 && c + 1 < NC && ABL != 2) {   // every wave is past its last read of the patch: refill it in place
            issue_patch(c + 1, 0);
            __syncthreads();
        }
    }
    if (ABL == 1) {
        float sacc = 0.0f;
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int j = 0; j < FJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
        if (sacc == 12345.678f) reinterpret_cast<float*>(g.C)[0] = sacc;
        return;
    }
    // ---- epilogue.  D map: col = lane & 31 -> pixel fr of block i; row = (r & 3) + 8 (r >> 2) + 4 fh -> channel
    const long long pix0 = ((long long)img * g.H + ty0) * g.W + tx0;         // NHWC row of tile pixel (0, 0)
    if (NCHW) {                                        // conv_out: fp32 NCHW (+clamp); lanes = consecutive pixels of a row
        TC* Cb = reinterpret_cast<TC*>(g.C);
        const long long hw = (long long)g.H * g.W;
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            const int py = wm * (2 * FI) + i * 2 + (fr >> 4);
            const long long pix = (long long)(ty0 + py) * g.W + tx0 + (fr & 15);
#pragma unroll
            for (int j = 0; j < FJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int n = n0 + wn * 64 + j * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                    if (n >= g.N) continue;
                    float v = acc[i][j][e] * g.alpha + (g.bias ? g.bias[n] : 0.0f);
                    if (g.clamp01) v = fminf(fmaxf(0.5f * v + 0.5f, 0.0f), 1.0f);
                    st1<TC>(Cb + ((long long)img * g.N + n) * hw + pix, v);
                }
        }
        return;
    }
    if constexpr (BN == 128) {
    // Staged store, 128 pixels (8 tile rows = the pixels of one wave row wm) at a time: fp32 tile through the dead operand
    // buffers, then full 256-B NHWC rows with 16-B stores per lane.
    char* stage = lds_raw;                              // [128 pixels][HALO_CPITCH] fp32; every operand read is behind the last barrier
    TC* Cb = reinterpret_cast<TC*>(g.C);
    const TC* Rb = reinterpret_cast<const TC*>(g.resid);
    const int c8 = (tid & 15) * 8;                      // 8 consecutive channels per thread, 16 threads per pixel row
    const int nn = n0 + c8;
    float gs[8], gq[8];                                 // fused GroupNorm statistics of this thread's channels
#pragma unroll
    for (int e = 0; e < 8; ++e) { gs[e] = 0.0f; gq[e] = 0.0f; }
    constexpr int HALVES = TY / 8;
#pragma unroll
    for (int half = 0; half < HALVES; ++half) {
        if (half > 0) __syncthreads();                  // the previous half has been read back
        if (ABL != 6 && (HALVES == 1 || wm == half)) {
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                const int r = (HALVES == 1 ? wm * 64 : 0) + i * 32 + fr;           // pixel within the staged 128
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int nl = wn * 64 + j * 32 + 8 * q4 + 4 * fh;
                        const f32x4 v = {acc[i][j][4 * q4], acc[i][j][4 * q4 + 1], acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]};
                        *reinterpret_cast<f32x4*>(stage + r * HALO_CPITCH + nl * 4) = v;
                    }
            }
        }
        __syncthreads();
        if (ABL == 7) continue;
        if (nn < g.N) {                                 // halo_ok(): N % 8 == 0, so a thread's 8 channels are all in or all out
            long long moff[8];
            u32x4 rr[8];
#pragma unroll
            for (int pass = 0; pass < 8; ++pass) {      // residual rows first: eight 16-B loads in flight, not one per pass
                const int r = pass * 16 + (tid >> 4);
                moff[pass] = (pix0 + (long long)(half * 8 + (r >> 4)) * g.W + (r & 15)) * g.ldc + nn;
                if (Rb) rr[pass] = *reinterpret_cast<const u32x4*>(Rb + moff[pass]);
            }
#pragma unroll
            for (int pass = 0; pass < 8; ++pass) {
                const int r = pass * 16 + (tid >> 4);
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + r * HALO_CPITCH + c8 * 4);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + r * HALO_CPITCH + c8 * 4 + 16);
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * g.alpha + bv[e];      // halo_ok(): no activation on this path
                if (Rb) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[2 * e] += bf16_to_f32((bf16_t)(rr[pass][e] & 0xffffu)); v[2 * e + 1] += bf16_to_f32((bf16_t)(rr[pass][e] >> 16)); }
                }
                const u32x4 o = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
                if (ABL != 5 || o[0] == 0x12345678u) *reinterpret_cast<u32x4*>(Cb + moff[pass]) = o;
                if (g.gn_part_out) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float rl = bf16_to_f32((bf16_t)(o[e] & 0xffffu)), rh = bf16_to_f32((bf16_t)(o[e] >> 16));
                        gs[2 * e] += rl; gq[2 * e] += rl * rl; gs[2 * e + 1] += rh; gq[2 * e + 1] += rh * rh;
                    }
                }
            }
        }
    }
    if (g.gn_part_out && ABL != 7) {                    // uniform branch (kernel argument): barriers are safe here
        __syncthreads();                                // every staged value has been read
        float* redw = reinterpret_cast<float*>(lds_raw);                    // [16 pixel rows][128 channels][2]; zeros from idle threads
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            redw[(((tid >> 4) * 128) + c8 + e) * 2] = gs[e];
            redw[(((tid >> 4) * 128) + c8 + e) * 2 + 1] = gq[e];
        }
        __syncthreads();
        const float* red = reinterpret_cast<const float*>(lds_raw);
        if (tid < 128) {                                // one channel per thread, then its group (cpg consecutive channels = lanes)
            float sa = 0.0f, sq = 0.0f;
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) { sa += red[((rg * 128) + tid) * 2]; sq += red[((rg * 128) + tid) * 2 + 1]; }
            const int cpg = g.N / g.gn_out_groups;
            for (int off = cpg >> 1; off > 0; off >>= 1) { sa += __shfl_xor(sa, off, 64); sq += __shfl_xor(sq, off, 64); }
            const int ch = n0 + tid;
            if (ch < g.N && (tid & (cpg - 1)) == 0) {
                float* pp = g.gn_part_out + (((long long)img * (tiles_x * tiles_y) + trem) * g.gn_out_groups + ch / cpg) * 2;
                pp[0] = sa; pp[1] = sq;
            }
        }
    }
    }
}

// fused output statistics need whole groups inside a wave: channels per group a power of two <= 64
bool conv_halo_stats_ok(int N, int groups) {
    if (groups <= 0 || N % groups != 0) return false;
    const int cpg = N / groups;
    return cpg <= 64 && (cpg & (cpg - 1)) == 0 && 128 % cpg == 0;
}
// 16-row tiles when the grid still fills the chip twice over (two workgroups per CU x 256 CUs), else 8-row tiles
int conv_halo_tile_rows(const GemmArgs& g) {
    const char* fe = getenv("HQT_HALO_TY");                                          // tuning / test hook (read per launch): 8 or 16
    const int forced = fe ? atoi(fe) : 0;
    if (g.H % 16 != 0) return 8;
    if (forced == 8 || forced == 16) return forced;
    const long long tiles16 = (long long)(g.M / 256) * ((g.N + 127) / 128);
    return tiles16 >= 1024 ? 16 : 8;
}
int conv_halo_tiles_per_image(const GemmArgs& g) { return (g.H / conv_halo_tile_rows(g)) * (g.W / HALO_TX); }

// shapes the halo kernel takes: bf16 3x3 conv, whole 8 x 16 pixel tiles, 64-channel chunks, NHWC bf16 rows that
// can be stored 16 B at a time (or the fp32 NCHW store of conv_out)
static bool halo_ok(const GemmArgs& g, int c_dt) {
    if (g.conv_taps != 9 || g.conv_stride2 || g.conv_nopad || !g.zero_page || g.gn_stats || g.a_packed_mb || (g.batch > 1)) return false;
    if (g.Cin % 64 != 0 || g.ldb % 8 != 0 || g.K != 9 * g.Cin) return false;
    if (g.H % 8 != 0 || g.W % HALO_TX != 0 || g.M % (g.H * g.W) != 0) return false;
    if (g.act != ACT_NONE) return false;               // swish lives in the GroupNorm pass; keeps the epilogue code small
    if (g.store == STORE_NCHW) return !g.resid;
    return g.store == STORE_ROWS && g.rows_per_group == 0 && c_dt == DT_BF16 && g.ldc % 8 == 0 && g.N % 8 == 0;
}

static bool glds_ok(const GemmArgs& g) {
    if (!g.zero_page || g.gn_stats || g.a_packed_mb) return false;
    if (!((g.store == STORE_ROWS && g.rows_per_group == 0) || (g.store == STORE_NCHW && !g.resid && g.act == ACT_NONE) ||
          (g.store == STORE_QKV && !g.conv_taps))) return false;
    if (g.K % 64 != 0 || g.ldb % 8 != 0) return false;
    if (g.conv_taps) return g.Cin % 64 == 0 && (g.conv_taps == 1 || g.conv_taps == 9) && !(g.conv_stride2 && g.upsample);
    return g.lda % 8 == 0;
}

static bool big_tile_shape(const GemmArgs& g);

bool mfma_gemm_ok(const GemmArgs& g, int a_dt, int b_dt, int c_dt) {
    (void)c_dt;
    if (a_dt != DT_BF16 || b_dt != DT_BF16 || g.a_packed_mb) return false;
    if (g.store == STORE_ARGMIN) return false;
    if (g.conv_stride2 || g.conv_nopad || g.conv_taps == 16) return big_tile_shape(g) && glds_ok(g);   // only the LDS-DMA kernel knows strided taps
    if (g.K % BK != 0) return false;
    if (g.conv_taps && g.Cin % BK != 0) return false;
    if (!g.conv_taps && (g.lda % 8 != 0)) return false;
    if (g.ldb % 8 != 0) return false;
    return true;
}

template <int BM, int BN>
static hipError_t launch_t(const GemmArgs& g, int c_dt, hipStream_t st) {
    const dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.batch > 0 ? g.batch : 1);
    if (c_dt == DT_BF16) mfma_gemm_kernel<BM, BN, bf16_t><<<grid, 256, 0, st>>>(g);
    else mfma_gemm_kernel<BM, BN, float><<<grid, 256, 0, st>>>(g);
    return hipGetLastError();
}

static bool big_tile_shape(const GemmArgs& g) {
    const long long tiles128 = (long long)((g.M + 127) / 128) * ((g.N + 127) / 128) * (g.batch > 0 ? g.batch : 1);
    static const bool force128 = getenv("HQT_FORCE_TILE128") != nullptr;          // test hook: exercise the big-tile kernels on tiny shapes
    const bool narrow = g.N < 32 && g.M >= 4096 && glds_ok(g);       // conv_out (N = 3): one zero-padded 128-wide n-tile
    return narrow || (g.N >= 128 && g.M >= 128 && (tiles128 >= 96 || force128));      // 96: the 1024-row x 1536-column GEMMs of the third code level
}
// true when launch_mfma_gemm will run the halo-tile kernel for g (the engine asks before requesting fused statistics)
bool conv_halo_ok(const GemmArgs& g, int c_dt) {
    const bool no_halo = getenv("HQT_NO_HALO") != nullptr;                    // A/B switch (read per launch): generic implicit GEMM for the 3x3 convs too
    return big_tile_shape(g) && !no_halo && halo_ok(g, c_dt);
}

hipError_t launch_mfma_gemm(const GemmArgs& g, int a_dt, int b_dt, int c_dt, hipStream_t st) {
    (void)a_dt; (void)b_dt;
    if (big_tile_shape(g)) {
        if (conv_halo_ok(g, c_dt)) {
            const int ty = conv_halo_tile_rows(g);
            const dim3 grid((g.N + 127) / 128, g.M / (ty * HALO_TX), 1);
#define LAUNCH_HALO(TC, NCHW_)                                                                              \
            if (ty == 16) conv3x3_halo_kernel<TC, NCHW_, 16><<<grid, 256, halo_lds(16), st>>>(g);           \
            else conv3x3_halo_kernel<TC, NCHW_, 8><<<grid, 256, halo_lds(8), st>>>(g);
            const bool no_narrow = getenv("HQT_NO_NARROW_OUT") != nullptr;   // A/B switch (read per launch)
            if (g.store == STORE_NCHW && g.N <= 32 && c_dt == DT_F32 && !no_narrow) {   // conv_out: 32-channel tiles instead of 3 padded to 128
                if (ty == 16) conv3x3_halo_kernel<float, true, 16, 0, 32><<<grid, 256, halo_lds(16, 32), st>>>(g);
                else conv3x3_halo_kernel<float, true, 8, 0, 32><<<grid, 256, halo_lds(8, 32), st>>>(g);
            } else if (g.store == STORE_NCHW) {
                if (c_dt == DT_BF16) { LAUNCH_HALO(bf16_t, true) } else { LAUNCH_HALO(float, true) }
            } else {
                LAUNCH_HALO(bf16_t, false)
            }
#undef LAUNCH_HALO
            return hipGetLastError();
        }
        if (glds_ok(g)) {
            const dim3 grid((g.N + 127) / 128, (g.M + 127) / 128, g.batch > 0 ? g.batch : 1);
            const size_t smem = (size_t)2 * 2 * 128 * 128;        // two LDS buffers, two workgroups per CU (a 4-slot ring at one workgroup per CU measured 1.5x slower)
            // few 128 x 128 tiles and rows to spare (third code level: [1024, 1536] GEMMs = 96 tiles): 64-row tiles fill the chip
            const long long t128 = (long long)grid.x * grid.y * grid.z;
            // (latency policy only: with several steps in flight the 128 x 128 tiles cost fewer CU-microseconds -- level-3 config,
            // 3 lanes: 594 vs 576 images/s; one lane: AR 158.7 vs 169.1 ms)
            if (t128 < 192 && g.M >= 256 && !g.conv_taps && g.store == STORE_ROWS && g.tune == 0 && !getenv("HQT_NO_TILE64")) {
                const dim3 grid64((g.N + 127) / 128, (g.M + 63) / 64, grid.z);
                const size_t smem64 = (size_t)2 * (64 + 128) * 128;
                if (g.rows_per_group == 0) {
                    if (c_dt == DT_BF16) conv_glds_kernel<bf16_t, 0, 2, 0, 64><<<grid64, 256, smem64, st>>>(g);
                    else conv_glds_kernel<float, 0, 2, 0, 64><<<grid64, 256, smem64, st>>>(g);
                } else {
                    if (c_dt == DT_BF16) conv_glds_kernel<bf16_t, 2, 2, 0, 64><<<grid64, 256, smem64, st>>>(g);
                    else conv_glds_kernel<float, 2, 2, 0, 64><<<grid64, 256, smem64, st>>>(g);
                }
                return hipGetLastError();
            }
#define LAUNCH_GLDS(TC, NCHW_)                                                                      \
            conv_glds_kernel<TC, NCHW_, 2><<<grid, 256, smem, st>>>(g);
            if (g.store == STORE_NCHW) {
                if (c_dt == DT_BF16) { LAUNCH_GLDS(bf16_t, 1) } else { LAUNCH_GLDS(float, 1) }
            } else if (g.store == STORE_ROWS && g.rows_per_group == 0) {
                if (c_dt == DT_BF16) { LAUNCH_GLDS(bf16_t, 0) } else { LAUNCH_GLDS(float, 0) }
            } else if (g.store == STORE_QKV && g.N % 4 == 0 && g.ldc % 4 == 0 && g.qkv_D % 4 == 0 && g.rows_per_group > 0) {
                if (c_dt == DT_BF16) { LAUNCH_GLDS(bf16_t, 3) } else { LAUNCH_GLDS(float, 3) }
            } else {
                if (c_dt == DT_BF16) { LAUNCH_GLDS(bf16_t, 2) } else { LAUNCH_GLDS(float, 2) }
            }
#undef LAUNCH_GLDS
            return hipGetLastError();
        }
        return launch_t<128, 128>(g, c_dt, st);
    }
    return launch_t<64, 64>(g, c_dt, st);
}

// raise the dynamic-LDS limits of the LDS-DMA conv kernels once (call outside stream capture)
hipError_t mfma_gemm_configure() {
    hipError_t e;
#define CFG(TC, NCHW_, NS)                                                                                          \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_glds_kernel<TC, NCHW_, NS>),                        \
                            hipFuncAttributeMaxDynamicSharedMemorySize, NS * 2 * 128 * 128);                        \
    if (e != hipSuccess) return e;
    CFG(bf16_t, 0, 2) CFG(float, 0, 2) CFG(bf16_t, 1, 2) CFG(float, 1, 2)
    CFG(bf16_t, 2, 2) CFG(float, 2, 2) CFG(bf16_t, 3, 2) CFG(float, 3, 2)
#undef CFG
#define CFGH(TC, NCHW_)                                                                                             \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<TC, NCHW_, 8>),                          \
                            hipFuncAttributeMaxDynamicSharedMemorySize, halo_lds(8));                                   \
    if (e != hipSuccess) return e;                                                                                      \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<TC, NCHW_, 16>),                         \
                            hipFuncAttributeMaxDynamicSharedMemorySize, halo_lds(16));                                  \
    if (e != hipSuccess) return e;
    CFGH(bf16_t, false) CFGH(bf16_t, true) CFGH(float, true)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<float, true, 8, 0, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, halo_lds(8, 32));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<float, true, 16, 0, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, halo_lds(16, 32));
    if (e != hipSuccess) return e;
#undef CFGH
    return hipSuccess;
}
