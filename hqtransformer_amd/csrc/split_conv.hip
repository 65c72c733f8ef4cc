// SPLIT precision kernels (see split_kernels.h): fp32-accurate 3x3 convolutions and GEMMs on the gfx950 matrix cores,
// fp16 hi/lo operand planes, three v_mfma_f32_32x32x16_f16 per fragment pair, fp32 accumulation.
#include "split_kernels.h"
#include "gemm_generic.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#include "split_device.h"

namespace {
__device__ __forceinline__ unsigned pack_h2(half_t a, half_t b) {
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

// same XCD-aware tile order as the FAST conv kernels (mfma_gemm.hip): every XCD owns a contiguous run of tiles, n fastest
__device__ __forceinline__ void xcd_tile(int& tile_m, int& tile_n) {
    const int nx = gridDim.x, total = gridDim.x * gridDim.y;
    int id = blockIdx.x + nx * blockIdx.y;
    if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
    tile_m = id / nx;
    tile_n = id - tile_m * nx;
}
}  // namespace

// ---------------------------------------------------------------------------------------------
// finalize: filters -> hi / lo planes
// ---------------------------------------------------------------------------------------------
__global__ void split_f32_kernel(const float* __restrict__ src, half_t* __restrict__ hi, half_t* __restrict__ lo, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) split2(src[i], hi[i], lo[i]);
}
hipError_t launch_split_f32(const float* src, half_t* hi, half_t* lo, size_t n, hipStream_t st) {
    split_f32_kernel<<<(int)std::min<size_t>((n + 255) / 256, 4096), 256, 0, st>>>(src, hi, lo, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// operand pass: fp32 NHWC -> [pixel][hi C | lo C] fp16, GroupNorm (+swish) applied on the way (layers.py:17-21,115-133).
// Same arithmetic as the EXACT path's operand loader: (x - mean) * rstd * gamma + beta, t / (1 + expf(-t)).
// A thread keeps the parameters of its 8 channels in registers and streams pixels, four loads in flight.
// ---------------------------------------------------------------------------------------------
static inline int sp_chunk_pix(int HW) { int c = HW / 32; return c < 64 ? 64 : (c > 1024 ? 1024 : c); }
static inline bool sp_fixed_ok(int C) { return C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0; }

#ifndef SP_INFLIGHT
#define SP_INFLIGHT 8
#endif
template <bool GN>
__global__ __launch_bounds__(256) void split_pack_kernel(const float* __restrict__ x, half_t* __restrict__ y, const float* __restrict__ stats,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, int HW, int C, int groups,
                                                         int swish, int nchunk, int chunk_pix, int* __restrict__ range_flag) {
    bool bad = false;
    const int b = blockIdx.x / nchunk, ch = blockIdx.x % nchunk;
    const int p0 = ch * chunk_pix, p1 = min(HW, p0 + chunk_pix);
    const int vpp = C / 8, rpi = 256 / vpp;
    const int cv = threadIdx.x % vpp, pl = threadIdx.x / vpp;
    float mu[8], rs[8], gm[8], bt[8];
    if (GN) {
        const int cpg = C / groups;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = cv * 8 + i;
            const float* st = stats + ((long long)b * groups + c / cpg) * 2;
            mu[i] = st[0]; rs[i] = st[1]; gm[i] = gamma[c]; bt[i] = beta[c];
        }
    }
    const float* xb = x + ((long long)b * HW) * C + cv * 8;
    half_t* yb = y + ((long long)b * HW) * 2 * C + cv * 8;
    auto emit = [&](const float4& a0, const float4& a1, int pix) {
        float f[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (GN) {
                float t = (f[i] - mu[i]) * rs[i] * gm[i] + bt[i];
                if (swish) t = t * __builtin_amdgcn_rcpf(1.0f + __expf(-t));     // v_exp_f32 / v_rcp_f32 (1 ulp each): the pass is close to VALU-bound with IEEE division
                f[i] = t;
            }
        }
        unsigned hi[4], lo[4];
        split8_checked(f, hi, lo, bad);
        const u32x4 vh = {hi[0], hi[1], hi[2], hi[3]};
        const u32x4 vl = {lo[0], lo[1], lo[2], lo[3]};
        half_t* dst = yb + (long long)pix * 2 * C;
        *reinterpret_cast<u32x4*>(dst) = vh;
        *reinterpret_cast<u32x4*>(dst + C) = vl;
    };
    int p = p0 + pl;
    constexpr int UF = SP_INFLIGHT;                   // pixels in flight per thread (32 bytes each): 2 / 4 / 8 / 16 -> 5.0 / 5.0 / 5.4 / 5.0 TB/s (tools/micro/bench_pack)
    for (; p + (UF - 1) * rpi < p1; p += UF * rpi) {
        float4 r0[UF], r1[UF];
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            const float* s = xb + (long long)(p + u * rpi) * C;
            r0[u] = *reinterpret_cast<const float4*>(s);
            r1[u] = *reinterpret_cast<const float4*>(s + 4);
        }
#pragma unroll
        for (int u = 0; u < UF; ++u) emit(r0[u], r1[u], p + u * rpi);
    }
    for (; p < p1; p += rpi) {
        const float* s = xb + (long long)p * C;
        emit(*reinterpret_cast<const float4*>(s), *reinterpret_cast<const float4*>(s + 4), p);
    }
    if (bad && range_flag) atomicOr(range_flag, 1);
}
// channel counts the fixed thread <-> channel mapping does not cover
__global__ __launch_bounds__(256) void split_pack_generic_kernel(const float* __restrict__ x, half_t* __restrict__ y, const float* __restrict__ stats,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta, long long total,
                                                                 int HW, int C, int groups, int swish, int* __restrict__ range_flag) {
    const int cpg = stats ? C / groups : 1;
    bool bad = false;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long pix = i / C;
        const int c = (int)(i - pix * C);
        float t = x[i];
        if (stats) {
            const float* st = stats + ((pix / HW) * groups + c / cpg) * 2;
            t = (t - st[0]) * st[1] * gamma[c] + beta[c];
            if (swish) t = t / (1.0f + expf(-t));
        }
        half_t hi, lo;
        split2_checked(t, hi, lo, bad);
        y[pix * 2 * C + c] = hi;
        y[pix * 2 * C + C + c] = lo;
    }
    if (bad && range_flag) atomicOr(range_flag, 1);
}
hipError_t launch_split_pack(const float* x, half_t* y, const float* stats, const float* gamma, const float* beta, int B, int HW,
                             int C, int groups, int swish, int* range_flag, hipStream_t st) {
    if (sp_fixed_ok(C)) {
        const int cp = sp_chunk_pix(HW), nchunk = (HW + cp - 1) / cp;
        if (stats) split_pack_kernel<true><<<B * nchunk, 256, 0, st>>>(x, y, stats, gamma, beta, HW, C, groups, swish, nchunk, cp, range_flag);
        else split_pack_kernel<false><<<B * nchunk, 256, 0, st>>>(x, y, nullptr, nullptr, nullptr, HW, C, groups, 0, nchunk, cp, range_flag);
        return hipGetLastError();
    }
    const long long total = (long long)B * HW * C;
    split_pack_generic_kernel<<<(int)std::min<long long>((total + 255) / 256, 256 * 16), 256, 0, st>>>(x, y, stats, gamma, beta, total, HW, C, groups, swish, range_flag);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// 3x3 'same' convolution, input patch resident in LDS (the SPLIT counterpart of conv3x3_halo_kernel, mfma_gemm.hip).
// One workgroup = an 8 x 16 pixel tile of one image x BN output channels.  Per 64-channel chunk the (8+2) x (16+2) input
// patch -- both planes -- is DMA'd into LDS once and serves all nine taps; the filters of one (tap, chunk) k-tile, both
// planes, are DMA'd per k-tile.  Both operand kinds are double-buffered: 2 x 2 x 23 KiB of patch + 2 x 2 x 16 KiB of
// filters = 156 KiB, one workgroup (4 waves, one per SIMD) per CU.  Per k-tile a wave issues 48 MFMAs (16 fragment pairs
// x 3) against 32 ds_read_b128 and ~9 DMA pieces: three times the matrix work of the bf16 kernel per byte staged.
// Weights are the MFMA A operand (D rows = channels, D cols = pixels): a lane owns 4 consecutive output channels of one
// pixel per register quad; the epilogue stages the fp32 tile through the dead operand buffers and writes whole NHWC
// rows (two 16-B stores per lane), adding bias and the fp32 residual and reducing the GroupNorm statistics of the
// output in a fixed order (double partials per tile and group).
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int S_TY = 8, S_TX = 16, S_PITCH = S_TX + 2;
constexpr int S_ROWS = (S_TY + 2) * S_PITCH;                    // 180 patch rows (pixels) of 128 B per plane
constexpr int S_PIECES = (S_ROWS + 7) / 8;                      // 23 DMA pieces of 8 rows
constexpr int S_PATCH_BYTES = S_PIECES * 1024;                  // per plane
constexpr int S_CPITCH = 128 * 4 + 16;                          // fp32 staging row (bytes)
constexpr int split_conv3_lds(int BN) { return 4 * S_PATCH_BYTES + 4 * BN * 128; }
static_assert(128 * S_CPITCH <= split_conv3_lds(32), "epilogue staging must fit in the operand buffers");
static_assert(split_conv3_lds(128) <= 160 * 1024, "LDS budget");
}  // namespace

// PC = 1: producer / consumer waves.  The workgroup has 8 waves, two per SIMD: waves 0-3 (consumers) only read fragments and
// issue MFMAs, waves 4-7 (producers) only issue the LDS-DMA of the next k-tile.  With one 4-wave workgroup per CU the ~37 DMA
// pieces per k-tile (address arithmetic + ~60-100 issue cycles each) sat in front of every wave's MFMAs with no second wave
// on the SIMD to cover them; as a separate wave they run beside the matrix pipe.
// ABL: ablation switches of tools/micro/bench_split (0 in the product): 1 no epilogue, 2 no DMA inside the loop, 3 DMA only (MFMAs of the
// first k-tile only), 4 MFMAs on whatever the registers hold (no fragment reads)
template <bool NCHW, int BN, int PC = 0, int ABL = 0>
__global__ __launch_bounds__(PC ? 512 : 256, 1) void conv3x3_split_kernel(GemmArgs g) {
    static_assert(BN == 128 || (BN == 32 && NCHW), "the 32-channel variant exists for the NCHW conv_out store only");
    constexpr int WMW = BN == 128 ? 2 : 4, FJ = BN == 128 ? 2 : 1, B_BYTES = BN * 128;
    constexpr int FI = (S_TY / 2) / WMW;                            // 32-pixel fragments per wave (2 | 1)
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    auto PATCH = [&](int s, int plane) -> char* { return lds_raw + (size_t)(s * 2 + plane) * S_PATCH_BYTES; };
    auto BT = [&](int s, int plane) -> char* { return lds_raw + 4 * S_PATCH_BYTES + (size_t)(s * 2 + plane) * B_BYTES; };
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;       // role-local ids (PC: the same for both roles)
    const bool producer = PC && threadIdx.x >= 256, consumer = !PC || threadIdx.x < 256;
    const int wm = BN == 128 ? wave >> 1 : wave, wn = BN == 128 ? wave & 1 : 0;
    const int fr = lane & 31, fh = lane >> 5;
    int tile_m, tile_n;
    xcd_tile(tile_m, tile_n);
    const int n0 = tile_n * BN;
    const int tiles_x = g.W / S_TX, tiles_y = g.H / S_TY;
    const int img = tile_m / (tiles_x * tiles_y);
    const int trem = tile_m - img * (tiles_x * tiles_y);
    const int ty0 = (trem / tiles_x) * S_TY, tx0 = (trem % tiles_x) * S_TX;
    const int Hin = g.H >> g.upsample, Win = g.W >> g.upsample;
    const half_t* Abase = reinterpret_cast<const half_t*>(g.A);                 // [pixel][hi Cin | lo Cin]
    const half_t* Bhi = reinterpret_cast<const half_t*>(g.Bw);
    const half_t* Blo = reinterpret_cast<const half_t*>(g.Bw_lo);
    const half_t* zero = reinterpret_cast<const half_t*>(g.zero_page);

    constexpr int PPW = (S_PIECES + 3) / 4;                         // patch pieces per wave and plane
    constexpr int BPW = BN / 32;                                    // filter pieces per wave and plane (4 | 1)
    long long boff[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const int row = (wave * BPW + i) * 8 + (lane >> 3);
        const int ch = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
        boff[i] = (n0 + row < g.N) ? (long long)(n0 + row) * g.ldb + ch : -1;
    }
    auto issue_patch = [&](int c, int s) {
#pragma unroll
        for (int t = 0; t < PPW; ++t) {
            if (wave + 4 * t < S_PIECES) {                          // wave-uniform
                const int q = (wave + 4 * t) * 8 + (lane >> 3);
                const int qy = (q * 3641) >> 16, qx = q - qy * S_PITCH;             // q / 18 for q < 328
                const int iy = ty0 + qy - 1, ix = tx0 + qx - 1;
                const int ch = ((lane & 7) ^ ((q >> 1) & 7)) * 8;
                const bool in = q < S_ROWS && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
                const half_t* src = Abase + (((long long)img * Hin + (iy >> g.upsample)) * Win + (ix >> g.upsample)) * (2 * g.Cin) + ch + c * 64;
#pragma unroll
                for (int plane = 0; plane < 2; ++plane) {
                    const half_t* sp = in ? src + plane * g.Cin : zero;
                    char* dst = PATCH(s, plane) + (wave + 4 * t) * 1024;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp,
                                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                }
            }
        }
    };
    auto issue_b = [&](int k0, int s) {
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            const half_t* sh = boff[i] >= 0 ? Bhi + boff[i] + k0 : zero;
            const half_t* sl = boff[i] >= 0 ? Blo + boff[i] + k0 : zero;
            char* dh = BT(s, 0) + (wave * BPW + i) * 1024;
            char* dl = BT(s, 1) + (wave * BPW + i) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sh, (__attribute__((address_space(3))) void*)dh, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sl, (__attribute__((address_space(3))) void*)dl, 16, 0, 0);
        }
    };

    f32x16 accm[FI][FJ], accx[FI][FJ];                              // main (hi x hi) and cross (hi x lo + lo x hi, scaled 2^11)
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accm[i][j][r] = 0.0f; accx[i][j][r] = 0.0f; }

    int qbase[FI];                                                  // patch row of tap (0, 0) for this lane's pixel of fragment i
#pragma unroll
    for (int i = 0; i < FI; ++i) qbase[i] = (wm * (2 * FI) + i * 2 + (fr >> 4)) * S_PITCH + (fr & 15);
    int brd[FJ];
#pragma unroll
    for (int j = 0; j < FJ; ++j) brd[j] = wn * 64 + j * 32 + fr;

    // Fragment reads run one k-step ahead of the MFMAs that consume them (two register sets): with one matrix wave per
    // SIMD nothing else covers the LDS latency.  hipcc left to itself sinks every read next to its use and waits
    // lgkmcnt(0) in front of each group of MFMAs (measured: 41 % of the matrix peak); the reads are therefore asm
    // statements hipcc cannot move, each set is retired by ONE counted wait that names its registers (so no MFMA is
    // scheduled above it), and the MFMAs stay builtins.
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_raw;
    unsigned wa0[FJ];                                               // byte offset of this lane's filter fragment (k-step 0) inside a filter plane
#pragma unroll
    for (int j = 0; j < FJ; ++j) wa0[j] = brd[j] * 128 + ((fh ^ ((brd[j] >> 1) & 7)) << 4);
    auto compute = [&](int ps, int bs, int tapoff) {
        const unsigned pbase = lds_base + ps * 2 * S_PATCH_BYTES, wbase = lds_base + 4 * S_PATCH_BYTES + bs * 2 * B_BYTES;
        unsigned pa0[FI];
#pragma unroll
        for (int i = 0; i < FI; ++i) pa0[i] = (qbase[i] + tapoff) * 128 + ((fh ^ (((qbase[i] + tapoff) >> 1) & 7)) << 4);
        half8 ah[2][FI], al[2][FI], wh[2][FJ], wl[2][FJ];
        auto issue = [&](int ks, int set) {                         // chunk (2 ks + fh) ^ swizzle = chunk(0) ^ (ks << 1): bits 5..6 of the offset
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                const unsigned a = wbase + (wa0[j] ^ (ks << 5));
                asm volatile("ds_read_b128 %0, %1" : "=v"(wh[set][j]) : "v"(a));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wl[set][j]) : "v"(a), "n"(B_BYTES));
            }
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                const unsigned a = pbase + (pa0[i] ^ (ks << 5));
                asm volatile("ds_read_b128 %0, %1" : "=v"(ah[set][i]) : "v"(a));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(al[set][i]) : "v"(a), "n"(S_PATCH_BYTES));
            }
        };
        // wait until all but the P youngest reads have landed (P = 0 | one k-step's reads); naming the registers orders the MFMAs behind it
#define HQT_RETIRE(set, P)                                                                                                          \
        do {                                                                                                                       \
            if constexpr (FI == 2 && FJ == 2)                                                                                      \
                asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(wh[set][0]), "+v"(wl[set][0]), "+v"(wh[set][FJ - 1]), "+v"(wl[set][FJ - 1]), \
                             "+v"(ah[set][0]), "+v"(al[set][0]), "+v"(ah[set][FI - 1]), "+v"(al[set][FI - 1]) : "n"(P));           \
            else                                                                                                                   \
                asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(wh[set][0]), "+v"(wl[set][0]), "+v"(ah[set][0]), "+v"(al[set][0]) : "n"(P)); \
        } while (0)
        constexpr int NRD = 2 * (FI + FJ);                          // reads per k-step
        if (ABL != 4) issue(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1;
            if (ABL == 4) {                                         // registers as they are: keep them opaque so nothing folds away
#pragma unroll
                for (int j = 0; j < FJ; ++j) asm volatile("" : "+v"(wh[cur][j]), "+v"(wl[cur][j]));
#pragma unroll
                for (int i = 0; i < FI; ++i) asm volatile("" : "+v"(ah[cur][i]), "+v"(al[cur][i]));
            } else if (ks + 1 < 4) { issue(ks + 1, cur ^ 1); HQT_RETIRE(cur, NRD); }
            else HQT_RETIRE(cur, 0);
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j) {
                    accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[cur][j], ah[cur][i], accm[i][j], 0, 0, 0);
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[cur][j], al[cur][i], accx[i][j], 0, 0, 0);
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[cur][j], ah[cur][i], accx[i][j], 0, 0, 0);
                }
        }
#undef HQT_RETIRE
    };

    float bv[8];                                        // epilogue bias of this thread's 8 channels, fetched under the main loop
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (BN == 128 && g.bias && n0 + (tid & 15) * 8 + e < g.N) ? g.bias[n0 + (tid & 15) * 8 + e] : 0.0f;
    const int NC = g.Cin / 64, KT = NC * 9;
    if (PC && consumer) __builtin_amdgcn_s_setprio(2);      // the matrix stream wins issue arbitration against its SIMD's producer wave
    if (!PC || producer) {
        issue_patch(0, 0);
        issue_b(0, 0);
    }
    __syncthreads();                                   // hipcc drains the DMA (vmcnt(0)) in front of the barrier
    int kt = 0;
#pragma unroll 1
    for (int c = 0; c < NC; ++c) {
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap, ++kt) {
            if ((!PC || producer) && kt + 1 < KT && ABL != 2) {
                const int tn = tap == 8 ? 0 : tap + 1, cn = tap == 8 ? c + 1 : c;
                issue_b(tn * g.Cin + cn * 64, (kt + 1) & 1);
                if (tap == 0 && c + 1 < NC) issue_patch(c + 1, (c + 1) & 1);
            }
            if (consumer && (ABL != 3 || kt == 0)) {
                const int t3 = (tap * 11) >> 5;        // tap / 3 for tap < 9
                compute(c & 1, kt & 1, t3 * S_PITCH + (tap - 3 * t3));
            }
            __syncthreads();
        }
    }
    if (PC && consumer) __builtin_amdgcn_s_setprio(0);
    if (ABL == 1) {                                    // no epilogue: keep the accumulators alive, one store per wave at most
        float sacc = 0.0f;
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int j = 0; j < FJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += accm[i][j][r] + accx[i][j][r];
        if (sacc == 12345.678f) reinterpret_cast<float*>(g.C)[0] = sacc;
        return;
    }
    // ---- epilogue.  D map: col = lane & 31 -> pixel fr of block i; row = (r & 3) + 8 (r >> 2) + 4 fh -> channel
    if (NCHW) {                                        // conv_out: fp32 NCHW (+clamp); lanes = consecutive pixels of a row
        if (!consumer) return;
        float* Cb = reinterpret_cast<float*>(g.C);
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
                    float v = (accm[i][j][e] + accx[i][j][e] * SPLIT_INV) * g.alpha + (g.bias ? g.bias[n] : 0.0f);
                    if (g.clamp01) v = fminf(fmaxf(0.5f * v + 0.5f, 0.0f), 1.0f);
                    Cb[((long long)img * g.N + n) * hw + pix] = v;
                }
        }
        return;
    }
    if constexpr (BN == 128) {
        const long long pix0 = ((long long)img * g.H + ty0) * g.W + tx0;         // NHWC row of tile pixel (0, 0)
        char* stage = lds_raw;                              // [128 pixels][S_CPITCH] fp32; every operand read is behind the last barrier
        float* Cb = reinterpret_cast<float*>(g.C);
        const float* Rb = reinterpret_cast<const float*>(g.resid);
        const int c8 = (tid & 15) * 8;                      // 8 consecutive channels per thread, 16 threads per pixel row
        const int nn = n0 + c8;
        float gs[8], gq[8];                                 // GroupNorm statistics of this thread's channels
#pragma unroll
        for (int e = 0; e < 8; ++e) { gs[e] = 0.0f; gq[e] = 0.0f; }
        if (consumer) {
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            const int r = wm * 64 + i * 32 + fr;            // pixel within the staged 128
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int nl = wn * 64 + j * 32 + 8 * q4 + 4 * fh;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = accm[i][j][4 * q4 + e] + accx[i][j][4 * q4 + e] * SPLIT_INV;
                    *reinterpret_cast<f32x4*>(stage + r * S_CPITCH + nl * 4) = v;
                }
        }
        }
        __syncthreads();
        if (consumer && nn < g.N) {                                 // split_conv3_ok(): N % 8 == 0, so a thread's 8 channels are all in or all out
            long long moff[8];
            f32x4 r0[8], r1[8];
#pragma unroll
            for (int pass = 0; pass < 8; ++pass) {      // residual rows first: sixteen 16-B loads in flight
                const int r = pass * 16 + (tid >> 4);
                moff[pass] = (pix0 + (long long)(r >> 4) * g.W + (r & 15)) * g.ldc + nn;
                if (Rb) { r0[pass] = *reinterpret_cast<const f32x4*>(Rb + moff[pass]); r1[pass] = *reinterpret_cast<const f32x4*>(Rb + moff[pass] + 4); }
            }
#pragma unroll
            for (int pass = 0; pass < 8; ++pass) {
                const int r = pass * 16 + (tid >> 4);
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + r * S_CPITCH + c8 * 4);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + r * S_CPITCH + c8 * 4 + 16);
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * g.alpha + bv[e];
                if (Rb) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += r0[pass][e]; v[4 + e] += r1[pass][e]; }
                }
                const f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                *reinterpret_cast<f32x4*>(Cb + moff[pass]) = o0;
                *reinterpret_cast<f32x4*>(Cb + moff[pass] + 4) = o1;
                if (g.gn_part_out_d) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { gs[e] += v[e]; gq[e] += v[e] * v[e]; }
                }
            }
        }
        if (g.gn_part_out_d) {                              // uniform branch (kernel argument): barriers are safe here
            __syncthreads();                                // every staged value has been read
            float* redw = reinterpret_cast<float*>(lds_raw);                    // [16 pixel rows][128 channels][2]; zeros from idle threads
            if (consumer) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                redw[(((tid >> 4) * 128) + c8 + e) * 2] = gs[e];
                redw[(((tid >> 4) * 128) + c8 + e) * 2 + 1] = gq[e];
            }
            }
            __syncthreads();
            const float* red = reinterpret_cast<const float*>(lds_raw);
            if (consumer && tid < 128) {                                // one channel per thread, then its group (cpg consecutive channels = lanes)
                double sa = 0.0, sq = 0.0;
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) { sa += (double)red[((rg * 128) + tid) * 2]; sq += (double)red[((rg * 128) + tid) * 2 + 1]; }
                const int cpg = g.N / g.gn_out_groups;
                for (int off = cpg >> 1; off > 0; off >>= 1) { sa += __shfl_xor(sa, off, 64); sq += __shfl_xor(sq, off, 64); }
                const int ch = n0 + tid;
                if (ch < g.N && (tid & (cpg - 1)) == 0) {
                    double* pp = g.gn_part_out_d + (((long long)img * (tiles_x * tiles_y) + trem) * g.gn_out_groups + ch / cpg) * 2;
                    pp[0] = sa; pp[1] = sq;
                }
            }
        }
    }
}

// 3x3 convs the ring kernels (split_stream_conv.hip) do not take -- channel counts that are not multiples of 128, conv_out with more
// than 16 channels -- run conv3x3_split_kernel above when its tile fits; anything else is left to the fp32 vector-ALU kernel.
bool split_conv3_ok(const GemmArgs& g) {
    if (g.conv_taps != 9 || g.conv_stride2 || g.conv_nopad || !g.zero_page || g.gn_stats || g.a_packed_mb || g.batch > 1) return false;
    if (!g.Bw_lo || g.ldb % 8 != 0 || g.K != 9 * g.Cin || g.M % (g.H * g.W) != 0) return false;
    if (g.Cin % 64 != 0 || g.H % S_TY != 0 || g.W % S_TX != 0) return false;
    if (g.act != ACT_NONE) return false;
    if (g.store == STORE_NCHW) return !g.resid && g.N <= 32;
    return g.store == STORE_ROWS && g.rows_per_group == 0 && g.ldc % 8 == 0 && g.N % 8 == 0;
}
int split_conv3_tiles_per_image(const GemmArgs& g) { return split_stream_ok(g) ? split_stream_tiles_per_image(g) : (g.H / S_TY) * (g.W / S_TX); }
hipError_t launch_split_conv3(const GemmArgs& g, hipStream_t st) {
    if (split_stream_ok(g)) return launch_split_conv3_stream(g, st);
    if (g.store == STORE_NCHW) conv3x3_split_kernel<true, 32, 1><<<dim3(1, g.M / (S_TY * S_TX), 1), 512, split_conv3_lds(32), st>>>(g);
    else conv3x3_split_kernel<false, 128, 1><<<dim3((g.N + 127) / 128, g.M / (S_TY * S_TX), 1), 512, split_conv3_lds(128), st>>>(g);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// 1x1 convolutions / plain GEMMs: C[m][n] = alpha sum_k A[m][k] W[n][k] (+bias) (+resid), 128 x 128 x 32 tiles, both
// operands (two planes each) staged through registers into two LDS stages (64 KiB, two workgroups per CU), 4 waves x (64 x 64).
// A rows are [m][hi K | lo K] (lda elements per row, lo at +a_lo_off) or, AF32, the fp32 tensor itself ([m][K], lda floats):
// the hi / lo split (range-checked like the operand pass) then happens between the load and the LDS write, and the separate
// 8-byte-per-element operand pass in front of a nin_shortcut / proj_out disappears.  BF32: the same for the B operand (the decoder
// attention's q k^T and softmax v products: both operands are activations).  Same source-side bank swizzle as conv_glds_kernel.
// The loads run TWO k-tiles ahead of the MFMAs (two register slots; round 2 had one: a k-tile is 768 matrix cycles, a third of
// a memory round trip, and the 16 k-tiles of a 512-channel 1x1 conv each waited for theirs: 23 % of the matrix peak).  Loads are
// unconditional (clamped k-tile) so that the loop body is one basic block and the compiler's counted waits stay exact.
// ---------------------------------------------------------------------------------------------
// (Round 5, measured and removed: four register slots -- twice the bytes in flight, one workgroup per CU, accumulators in AGPRs -- for the AR loop's launches of at most one
// tile per CU: 480 vs 472 ms per 640-row pass, profiles/r05_split_kslices.txt.  A k-tile of such a launch is not waiting for bytes: with one workgroup on the CU nothing
// overlaps its ds_read -> MFMA -> stash -> barrier chain.)
template <bool AF32, bool BF32>
__global__ __launch_bounds__(256, 2) void split_gemm_kernel(GemmArgs g, int a_lo_off) {
    constexpr int BM = 128, BN = 128, BKG = 32, ROWB = 64, OPB = 128 * ROWB;        // 32-wide k-tiles: 64 KiB of LDS, two workgroups per CU
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];                   // [2 stages][A hi, A lo, B hi, B lo][128 rows of 64 B]
    auto LDS = [&](int stage, int op) -> char* { return lds_raw + (size_t)(stage * 4 + op) * OPB; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // K slices (AF32 rows of the SPLIT AR loop whose 128 x 128 tiles would fill a quarter of the chip: fc2 / proj at 640 rows are 60 tiles):
    // blockIdx.z is the slice, the partial goes to its fp32 slab as it is
    const int S = (AF32 && !BF32 && g.k_slices > 1) ? g.k_slices : 1;
    const int ks = S > 1 ? (int)blockIdx.z : 0, bz = S > 1 ? 0 : (int)blockIdx.z;
    const int kbase = ks * (g.K / S);
    int tile_m, tile_n;
    xcd_tile(tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const half_t* Abase = reinterpret_cast<const half_t*>(g.A) + (long long)bz * g.a_batch_stride;
    const float* Abase32 = reinterpret_cast<const float*>(g.A) + (long long)bz * g.a_batch_stride;
    const half_t* Bhi = reinterpret_cast<const half_t*>(g.Bw) + (long long)bz * g.b_batch_stride;
    const half_t* Blo = reinterpret_cast<const half_t*>(g.Bw_lo) + (long long)bz * g.b_batch_stride;
    const float* Bbase32 = reinterpret_cast<const float*>(g.Bw) + (long long)bz * g.b_batch_stride;     // BF32: B is an fp32 tensor too ([n][K], ldb floats)
    const half_t* zero = reinterpret_cast<const half_t*>(g.zero_page);
    // a 1-KiB piece = 16 rows x 4 chunks of 16 B; chunk c of row r lives at position c ^ ((r >> 2) & 3): the 16 rows a ds_read_b128
    // lane group covers (lanes {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}) then fall on 16 different 16-B slots of the 256-B LDS row
    long long aoff[2], boff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 16 + (lane >> 2);
        const int ch = ((lane & 3) ^ ((row >> 2) & 3)) * 8;
        aoff[i] = (m0 + row < g.M) ? (long long)(m0 + row) * g.lda + ch : -1;
        boff[i] = (n0 + row < g.N) ? (long long)(n0 + row) * g.ldb + ch : -1;
    }
    const int KT = g.K / S / BKG;
    u32x4 stg[2][2][4];                                 // [slot][piece][A hi | A lo | W hi | W lo]; AF32: [first | second four floats | W hi | W lo]
    auto fetch = [&](int kt, u32x4 (&sl)[2][4]) {
        const int k0 = kbase + min(kt, KT - 1) * BKG;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (AF32) {
                const float* sa = aoff[i] >= 0 ? Abase32 + aoff[i] + k0 : reinterpret_cast<const float*>(zero);
                sl[i][0] = *reinterpret_cast<const u32x4*>(sa);
                sl[i][1] = *reinterpret_cast<const u32x4*>(sa + 4);
            } else {
                sl[i][0] = *reinterpret_cast<const u32x4*>(aoff[i] >= 0 ? Abase + aoff[i] + k0 : zero);
                sl[i][1] = *reinterpret_cast<const u32x4*>(aoff[i] >= 0 ? Abase + aoff[i] + a_lo_off + k0 : zero);
            }
            if (BF32) {
                const float* sb = boff[i] >= 0 ? Bbase32 + boff[i] + k0 : reinterpret_cast<const float*>(zero);
                sl[i][2] = *reinterpret_cast<const u32x4*>(sb);
                sl[i][3] = *reinterpret_cast<const u32x4*>(sb + 4);
            } else {
                sl[i][2] = *reinterpret_cast<const u32x4*>(boff[i] >= 0 ? Bhi + boff[i] + k0 : zero);
                sl[i][3] = *reinterpret_cast<const u32x4*>(boff[i] >= 0 ? Blo + boff[i] + k0 : zero);
            }
        }
    };
    bool bad = false;
    auto stash = [&](int buf, const u32x4 (&sl)[2][4]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            char* at = lds_raw + (size_t)buf * 4 * OPB + (wave * 2 + i) * 16 * ROWB + lane * 16;
            if (AF32) {
                float xs[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned bits = sl[i][e >> 2][e & 3];      // (hipcc 7.2: __builtin_bit_cast straight on a vector ELEMENT reads element 0)
                    xs[e] = __builtin_bit_cast(float, bits);
                }
                unsigned hi[4], lo[4];
                split8_checked(xs, hi, lo, bad);
                *reinterpret_cast<u32x4*>(at) = u32x4{hi[0], hi[1], hi[2], hi[3]};
                *reinterpret_cast<u32x4*>(at + OPB) = u32x4{lo[0], lo[1], lo[2], lo[3]};
            } else {
                *reinterpret_cast<u32x4*>(at) = sl[i][0];
                *reinterpret_cast<u32x4*>(at + OPB) = sl[i][1];
            }
            if (BF32) {
                float xs[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned bits = sl[i][2 + (e >> 2)][e & 3];
                    xs[e] = __builtin_bit_cast(float, bits);
                }
                unsigned hi[4], lo[4];
                split8_checked(xs, hi, lo, bad);
                *reinterpret_cast<u32x4*>(at + 2 * OPB) = u32x4{hi[0], hi[1], hi[2], hi[3]};
                *reinterpret_cast<u32x4*>(at + 3 * OPB) = u32x4{lo[0], lo[1], lo[2], lo[3]};
            } else {
                *reinterpret_cast<u32x4*>(at + 2 * OPB) = sl[i][2];
                *reinterpret_cast<u32x4*>(at + 3 * OPB) = sl[i][3];
            }
        }
    };
    // fp32 A rows (the SPLIT AR loop, the decoder's nin_shortcut / q / k / v / proj_out): the hi / lo split of the NEXT k-tile's rows is ~130 vector instructions per lane.
    // convert() is pure register work and shares a scheduling region with compute(): hipcc issues it in the shadow of the matrix instructions, only the LDS writes stay
    // behind them (2048-row AR GEMMs 2800 -> 2520 cycles per k-tile with two workgroups per CU, the decoder's fp32-A 1x1 convs 399 -> 376 us; a launch of ONE workgroup
    // per CU stays at 2140 cycles per k-tile -- its four waves run in lockstep between barriers -- whatever the order of the tiles, the prefetch depth or the tile height:
    // profiles/r05_micro_split_gemm_ar.txt).
    u32x4 cva[2][2];                                    // [piece][hi | lo] of the slot on its way to LDS
    auto convert = [&](const u32x4 (&sl)[2][4]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float xs[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned bits = sl[i][e >> 2][e & 3];
                xs[e] = __builtin_bit_cast(float, bits);
            }
            unsigned hi[4], lo[4];
            split8_checked(xs, hi, lo, bad);
            cva[i][0] = u32x4{hi[0], hi[1], hi[2], hi[3]};
            cva[i][1] = u32x4{lo[0], lo[1], lo[2], lo[3]};
        }
    };
    auto write = [&](int buf, const u32x4 (&sl)[2][4]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            char* at = lds_raw + (size_t)buf * 4 * OPB + (wave * 2 + i) * 16 * ROWB + lane * 16;
            *reinterpret_cast<u32x4*>(at) = cva[i][0];
            *reinterpret_cast<u32x4*>(at + OPB) = cva[i][1];
            *reinterpret_cast<u32x4*>(at + 2 * OPB) = sl[i][2];
            *reinterpret_cast<u32x4*>(at + 3 * OPB) = sl[i][3];
        }
    };
    f32x16 accm[2][2], accx[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accm[i][j][r] = 0.0f; accx[i][j][r] = 0.0f; }
    const int fr = lane & 31, fh = lane >> 5;
    auto compute = [&](int buf) {
        const char *Ah = LDS(buf, 0), *Al = LDS(buf, 1), *Wh = LDS(buf, 2), *Wl = LDS(buf, 3);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int c = ks * 2 + fh;
            half8 ah[2], al[2], wh[2], wl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wm * 64 + i * 32 + fr;
                const int o = r * ROWB + ((c ^ ((r >> 2) & 3)) << 4);
                ah[i] = *reinterpret_cast<const half8*>(Ah + o);
                al[i] = *reinterpret_cast<const half8*>(Al + o);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = wn * 64 + j * 32 + fr;
                const int o = r * ROWB + ((c ^ ((r >> 2) & 3)) << 4);
                wh[j] = *reinterpret_cast<const half8*>(Wh + o);
                wl[j] = *reinterpret_cast<const half8*>(Wl + o);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {                               // D rows = n, cols = m
                    accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], ah[i], accm[i][j], 0, 0, 0);
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], al[i], accx[i][j], 0, 0, 0);
                    accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[j], ah[i], accx[i][j], 0, 0, 0);
                }
        }
    };
#ifdef HQT_SPLIT_GEMM_STAMPS                            // tools/micro only: cycles of the prologue / main loop / epilogue of every workgroup's wave 0
    long long st_[4];
    st_[0] = clock64();
#endif
    fetch(0, stg[0]);
    fetch(1, stg[1]);
    stash(0, stg[0]);
    __syncthreads();
#ifdef HQT_SPLIT_GEMM_STAMPS
    st_[1] = clock64();
#endif
    if constexpr (AF32 && !BF32) {
        for (int kt = 0; kt < KT; kt += 2) {
            fetch(kt + 2, stg[0]);
            __builtin_amdgcn_sched_barrier(0);
            compute(0);
            convert(stg[1]);                            // k-tile kt + 1, in the shadow of the MFMAs of k-tile kt
            __builtin_amdgcn_sched_barrier(0);
            write(1, stg[1]);
            __syncthreads();
            fetch(kt + 3, stg[1]);
            __builtin_amdgcn_sched_barrier(0);
            compute(1);
            convert(stg[0]);
            __builtin_amdgcn_sched_barrier(0);
            write(0, stg[0]);
            __syncthreads();
        }
    } else
    for (int kt = 0; kt < KT; kt += 2) {                // KT is even (K % 64 == 0)
        fetch(kt + 2, stg[0]);                          // slot 0 went to LDS one k-tile ago; in flight under TWO k-tiles of MFMAs
        __builtin_amdgcn_sched_barrier(0);              // (without the fences hipcc sinks the loads below the LDS writes of the other slot)
        compute(0);
        __builtin_amdgcn_sched_barrier(0);
        stash(1, stg[1]);                               // k-tile kt + 1: its readers passed the barrier below one k-tile ago
        __syncthreads();
        fetch(kt + 3, stg[1]);
        __builtin_amdgcn_sched_barrier(0);
        compute(1);
        __builtin_amdgcn_sched_barrier(0);
        stash(0, stg[0]);                               // k-tile kt + 2 (past the end: a clamped duplicate nobody reads)
        __syncthreads();
    }
    if ((AF32 || BF32) && bad && g.range_flag) atomicOr(g.range_flag, 1);
#ifdef HQT_SPLIT_GEMM_STAMPS
    st_[2] = clock64();
#endif
    // epilogue: a lane owns 4 consecutive columns of one row per register quad.  The bias of the tile's 128 columns goes through the
    // (now dead) LDS stages: as one scalar load in front of every store, each a dependent L2 round trip, the epilogue took 32 k cycles
    // of a workgroup's 80 k (tools/micro/bench_split_gemm, in-kernel stamps); 14-25 k since.
    float* Cb = S > 1 ? g.k_slabs + (size_t)ks * g.M * g.N : reinterpret_cast<float*>(g.C) + (long long)bz * g.c_batch_stride;
    const float* Rb = (g.resid && S == 1) ? reinterpret_cast<const float*>(g.resid) + (long long)bz * g.c_batch_stride : nullptr;
    const int ldc = S > 1 ? g.N : g.ldc, act = S > 1 ? ACT_NONE : g.act;       // (bias, activation and residual of a sliced launch: the combine)
    // STORE_QKV (SPLIT AR loop: the fused [query; key; value] GEMM of a block, stage2/layers.py:73-85): a wave's 64-column block lies in ONE
    // part (the launcher guarantees qkv_D % 64 == 0), so the destination -- q rows as they are, K / V rows through the cache's row remap --
    // is resolved once per wave, never per element (a per-element select between C / C2 / C3 was miscompiled by hipcc -O3, fast_kernels.hip)
    const bool qkv = g.store == STORE_QKV;
    const bool plain = ((g.store == STORE_ROWS && g.rows_per_group == 0) || qkv) && (g.N & 3) == 0 && (ldc & 3) == 0;
    const int qkv_row_dev = (qkv && g.row_offset_dev) ? *g.row_offset_dev : 0;
    float* lbias = reinterpret_cast<float*>(lds_raw);
    if (tid < BN) lbias[tid] = (g.bias && S == 1 && n0 + tid < g.N) ? g.bias[n0 + tid] : 0.0f;     // every wave passed the loop's last barrier: the stages are free
    __syncthreads();
    if (plain) {
        // Row-major stores straight from the accumulators leave as 32 rows x 32 B per instruction (a million 32-byte write requests per
        // launch of the 512-channel convs).  As in tile_gemm.hip every wave transposes its own 64 x 64 block through a PRIVATE patch of the
        // dead stages, 32 rows at a time -- no workgroup barrier -- and stores whole 256-byte row segments, 16 B per lane; the residual is
        // read the same way, all passes in flight together.  (A workgroup-wide staging of 64 x 128 halves with barriers measured slower
        // than the direct stores: 46.6 vs 38.5 us.)
        constexpr int PITCH = 64 + 4;                     // floats; +4: the 8 lanes of a ds_write_b128 group land on 8 distinct 4-bank sets
        float* const stg = reinterpret_cast<float*>(lds_raw) + BN + wave * (32 * PITCH);
        static_assert((BN + 4 * 32 * PITCH) * 4 <= 2 * 4 * OPB, "bias + staging patches fit the operand stages");
        const int cg = (lane & 15) * 4, r0 = lane >> 4;
        const int ncol0 = n0 + wn * 64;
        const int part_local = qkv ? ncol0 / max(g.qkv_D, 1) : 0, part = part_local + (qkv ? g.qkv_first : 0);
        float* const Cw = qkv ? reinterpret_cast<float*>(part == 0 ? g.C : (part == 1 ? g.C2 : g.C3)) : Cb;
        const int ccol0 = ncol0 - part_local * (qkv ? g.qkv_D : 0);          // column inside the destination
        const bool remap = qkv && part > 0;
        dispatch_act(act, [&](auto act_c) {                           // (the activation as a constant: gemm_generic.h)
        constexpr int ACT = decltype(act_c)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mrow0 = m0 + wm * 64 + i * 32;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int cl = j * 32 + 8 * q4 + 4 * fh;
                    const f32x4 bq = *reinterpret_cast<const f32x4*>(lbias + wn * 64 + cl);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = apply_act_c<ACT>((accm[i][j][4 * q4 + e] + accx[i][j][4 * q4 + e] * SPLIT_INV) * g.alpha + bq[e]);
                    *reinterpret_cast<f32x4*>(stg + fr * PITCH + cl) = v;
                }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // the patch is exchanged between the lanes of this wave only
            __builtin_amdgcn_wave_barrier();
            f32x4 x0[8];
            const bool col_ok = ncol0 + cg < g.N;                      // N % 4 == 0
            if (Rb) {
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const int m = min(mrow0 + p * 4 + r0, g.M - 1);
                    x0[p] = *reinterpret_cast<const f32x4*>(Rb + (long long)m * ldc + min(ncol0 + cg, g.N - 4));
                }
            }
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int r = p * 4 + r0, m = mrow0 + r;
                f32x4 v = *reinterpret_cast<const f32x4*>(stg + r * PITCH + cg);
                if (Rb) v += x0[p];
                long long orow = m;
                if (remap) orow = (long long)(m / g.rows_per_group) * g.group_stride + m % g.rows_per_group + g.row_offset + qkv_row_dev;
                if (m < g.M && col_ok) *reinterpret_cast<f32x4*>(Cw + orow * ldc + ccol0 + cg) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();                           // the patch is rewritten by the next row block
        }
        });
    } else {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 64 + i * 32 + fr;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int nl = wn * 64 + j * 32 + 8 * q4 + 4 * fh, n4 = n0 + nl;
                if (n4 >= g.N) continue;
                const f32x4 bq = *reinterpret_cast<const f32x4*>(lbias + nl);
                float s[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] = accm[i][j][4 * q4 + e] + accx[i][j][4 * q4 + e] * SPLIT_INV;
                if (g.store == STORE_NCHW) {                // per-image transposed store (V^T of the decoder attention: layers.py:180-183): 32 lanes = 32 consecutive pixels
                    const int img = m / g.rows_per_image, pix = m - img * g.rows_per_image;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n4 + e < g.N) {
                            float x = s[e] * g.alpha + bq[e];
                            if (g.clamp01) x = fminf(fmaxf(0.5f * x + 0.5f, 0.0f), 1.0f);
                            Cb[((long long)img * g.N + n4 + e) * g.rows_per_image + pix] = x;
                        }
                } else {                                      // ragged N / ldc: element-wise rows
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n4 + e < g.N) {
                            const long long idx = (long long)m * ldc + n4 + e;
                            float x = apply_act(s[e] * g.alpha + bq[e], act);
                            if (Rb) x += Rb[idx];
                            Cb[idx] = x;
                        }
                }
            }
    }
    }
#ifdef HQT_SPLIT_GEMM_STAMPS
    if (g.am_best && tid == 0) {
        st_[3] = clock64();
        long long* d = reinterpret_cast<long long*>(g.am_best) + (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 4;
        d[0] = st_[1] - st_[0]; d[1] = st_[2] - st_[1]; d[2] = st_[3] - st_[2]; d[3] = wall_clock64();
    }
#endif
}

// finishes a K-sliced launch: C[m][n] = (sum over the slices in index order) + bias[n] (+ resid[m][n]); 4 columns per thread
__global__ __launch_bounds__(256) void split_rows_combine_kernel(float* C, int ldc, const float* resid, const float* __restrict__ slabs,
                                                                 const float* __restrict__ bias, int M, int N, int S, int act) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x, n4 = N >> 2;
    if (i >= (long long)M * n4) return;
    const int m = (int)(i / n4), n = (int)(i - (long long)m * n4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(slabs + (size_t)m * N + n);
    for (int z = 1; z < S; ++z) v += *reinterpret_cast<const f32x4*>(slabs + ((size_t)z * M + m) * N + n);
    if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
    if (act != ACT_NONE)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], act);
    if (resid) v += *reinterpret_cast<const f32x4*>(resid + (long long)m * ldc + n);
    *reinterpret_cast<f32x4*>(C + (long long)m * ldc + n) = v;
}

// K slices of a launch whose tiles alone would leave most of the chip idle (up to 4, while two workgroups per slice-tile still fit 256 CUs once)
int split_gemm_slices(const GemmArgs& g) {
    if (!g.a_f32 || g.b_f32 || g.store != STORE_ROWS || g.rows_per_group != 0 || g.batch > 1 || g.conv_taps) return 1;
    if (g.N % 4 != 0 || g.ldc % 4 != 0 || g.alpha != 1.0f) return 1;
    const int tiles = ((g.N + 127) / 128) * ((g.M + 127) / 128);
    // measured on the AR loop's shapes (profiles/r05_split_kslices.txt): slices pay while the launch stays within one workgroup per CU, and
    // -- fc2, K = 4 D -- up to two per CU as long as a slice keeps 1536 of K; beyond that the slab traffic and the shorter loops cost more
    int S = 1;
    while (S < 4 && g.K % (S * 2 * 64) == 0 && (tiles * S * 2 <= 256 || (tiles * S * 2 <= 512 && g.K / (S * 2) >= 1536))) S *= 2;
    return S;
}

bool split_gemm_ok(const GemmArgs& g) {
    if (!g.zero_page || (!g.Bw_lo && !g.b_f32) || g.gn_stats || g.a_packed_mb || g.a_rows_per_group) return false;
    if (g.conv_taps > 1 || g.conv_stride2 || g.upsample) return false;
    if (g.K % 64 != 0 || g.ldb % 8 != 0) return false;
    if (g.a_f32 && (g.conv_taps ? g.Cin : g.lda) % 4 != 0) return false;
    if (g.b_f32 && (!g.a_f32 || g.ldb % 4 != 0)) return false;
    if (g.store == STORE_QKV)       // the SPLIT AR loop's fused [query; key; value] GEMM: fp32 rows, parts aligned to a wave's 64 columns
        return g.a_f32 && !g.b_f32 && g.qkv_D > 0 && g.qkv_D % 64 == 0 && g.N % 64 == 0 && g.ldc % 4 == 0 && g.rows_per_group > 0 && !g.resid && !g.qkv_v_pk && g.batch <= 1;
    if (g.store != STORE_ROWS && g.store != STORE_NCHW) return false;
    if (g.store == STORE_ROWS && g.rows_per_group != 0) return false;
    if (g.store == STORE_NCHW && (g.resid || g.act != ACT_NONE)) return false;
    return true;
}
hipError_t launch_split_gemm(const GemmArgs& g0, hipStream_t st) {
    GemmArgs g = g0;
    if (g.conv_taps == 1) { g.lda = g.a_f32 ? g.Cin : 2 * g.Cin; g.conv_taps = 0; }   // a 1x1 conv over [pixel][hi C | lo C] (or [pixel][C] fp32) is a plain GEMM
    const int a_lo_off = g.lda / 2;
    if (g.k_slices > 1 && (!g.k_slabs || g.k_slices > split_gemm_slices(g))) g.k_slices = 1;
    const int S = g.k_slices > 1 ? g.k_slices : 1;
    const dim3 grid((g.N + 127) / 128, (g.M + 127) / 128, S > 1 ? S : (g.batch > 0 ? g.batch : 1));
    if (g.a_f32 && g.b_f32) split_gemm_kernel<true, true><<<grid, 256, 2 * 4 * 128 * 64, st>>>(g, a_lo_off);
    else if (g.a_f32) split_gemm_kernel<true, false><<<grid, 256, 2 * 4 * 128 * 64, st>>>(g, a_lo_off);
    else split_gemm_kernel<false, false><<<grid, 256, 2 * 4 * 128 * 64, st>>>(g, a_lo_off);
    if (S > 1) {
        const long long n = (long long)g.M * (g.N / 4);
        split_rows_combine_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(reinterpret_cast<float*>(g.C), g.ldc, reinterpret_cast<const float*>(g.resid), g.k_slabs, g.bias, g.M, g.N, S, g.act);
    }
    return hipGetLastError();
}

hipError_t split_kernels_configure() {
    hipError_t e = split_stream_configure();
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_split_kernel<false, 128, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, split_conv3_lds(128));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_split_kernel<true, 32, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, split_conv3_lds(32));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(split_gemm_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4 * 128 * 64);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(split_gemm_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4 * 128 * 64);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(split_gemm_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4 * 128 * 64);
}
