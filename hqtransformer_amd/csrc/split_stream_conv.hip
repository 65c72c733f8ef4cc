// SPLIT precision, third generation of the 3x3 convolution ("stream"): one wave per SIMD, everything software-pipelined
// inside ONE instruction stream.
//
// What the two earlier kernels (split_conv.hip) taught, by in-kernel stamps and ablations (tools/micro/bench_split):
//   * the two waves of a SIMD do not overlap vector / LDS / memory issue of one with matrix issue of the other: the ~100
//     non-MFMA instructions a wave needs per k-tile (fragment reads, staging, address arithmetic, waits, the barrier) simply
//     add to the 1536 matrix cycles of a step -- 62-68 % matrix-pipe occupancy at best;
//   * a single wave, on the other hand, issues up to ~5 independent instructions in the shadow of each of its own 32-cycle
//     MFMAs for free;
//   * filters staged through LDS cost a ring, a ds_write pass, 8 fragment reads per k-tile and wave, and a barrier per k-tile.
// Hence:
//   * 4 waves per workgroup, 8 x 16 pixel tile x 128 channels, each wave 64 pixels x 64 channels (128 accumulator registers:
//     256 -- a 16 x 16 tile -- would need the accumulators in AGPRs, and hipcc then shuffles them through v_accvgpr moves and
//     spills), two workgroups per CU: 24 MFMAs per k-tile and wave against 8 fragment reads (the patch), 8 global loads (the
//     filters) and <= 1 LDS-DMA piece, every wave self-contained between two chunk barriers;
//   * the FILTERS never touch LDS: they are packed at finalize in MFMA A-operand fragment order (1-KiB chunks, [n-tile of 32]
//     [chunk][tap][k-step][hi | lo]), and every wave streams the fragments of its own 64 channels straight into registers
//     with coalesced global_load_dwordx4, one k-tile ahead (two waves share each fragment: L1 / L2 hits);
//   * the PATCH ((8+2) x (16+2) pixels x 32 channels, both planes, 24 KiB) is double-buffered in LDS and filled by LDS-DMA,
//     one piece per wave at each of the first six taps of the previous chunk: ONE workgroup barrier per chunk of nine k-tiles;
//   * fragment reads run one k-step ahead; reads, loads, DMA pieces and waits are placed between the MFMAs in program
//     order (asm statements keep their order; the MFMAs are pinned between them by their register dependencies).
// Registers: 128 accumulators (main + cross) + 32 patch fragments (two sets) + 64 filter fragments (two k-tiles) + addresses < 256.
#include "split_kernels.h"
#include "gemm_generic.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

namespace {
constexpr float R_INV = 1.0f / 2048.0f;
constexpr int R_TY = 8, R_TX = 16, R_PITCH = R_TX + 2;
constexpr int R_ROWS = (R_TY + 2) * R_PITCH;                    // 180 patch rows (pixels) of 64 B per plane
constexpr int R_PIECES = (R_ROWS + 15) / 16;                    // 12 DMA pieces of 16 rows per plane
constexpr int R_PLANE = R_PIECES * 1024;
constexpr int R_CPITCH = 128 * 4 + 16;                          // fp32 staging row of the epilogue (bytes)
constexpr int R_LDS = 4 * R_PLANE;                              // two buffers x two planes = 48 KiB
static_assert(64 * R_CPITCH <= R_LDS, "epilogue staging (64 pixels at a time) must fit in the patch buffers");

__device__ __forceinline__ void r_xcd_tile(int& tile_m, int& tile_n) {
    const int nx = gridDim.x, total = gridDim.x * gridDim.y;
    int id = blockIdx.x + nx * blockIdx.y;
    if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
    tile_m = id / nx;
    tile_n = id - tile_m * nx;
}
}  // namespace

// ---------------------------------------------------------------------------------------------
// finalize: tap-major fp32 filters [N][9 Cin] -> fragment-packed fp16 hi / lo.  Chunk index of (n-tile t, chunk c, tap, k-step ks,
// plane p) = (((t NC + c) 9 + tap) 2 + ks) 2 + p; inside a chunk lane l holds W[32 t + (l & 31)][tap Cin + 32 c + 16 ks + 8 (l >> 5) + j].
// Rows beyond N are zero.
// ---------------------------------------------------------------------------------------------
__global__ void pack_split_frag_kernel(const float* __restrict__ w, half_t* __restrict__ out, int N, int Cin, size_t total) {
    const int NC = Cin / 32;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        size_t ch = i >> 9;
        const int plane = (int)(ch & 1); ch >>= 1;
        const int ks = (int)(ch & 1); ch >>= 1;
        const int tap = (int)(ch % 9); ch /= 9;
        const int c = (int)(ch % NC);
        const int t = (int)(ch / NC);
        const int n = t * 32 + (lane & 31), k = tap * Cin + c * 32 + ks * 16 + 8 * (lane >> 5) + j;
        const float x = n < N ? w[(size_t)n * 9 * Cin + k] : 0.0f;
        const half_t hi = (half_t)x;
        out[i] = plane ? (half_t)((x - (float)hi) * 2048.0f) : hi;
    }
}
// (+ three k-tiles of padding: the kernels prefetch the filters of up to five k-steps past the end of the last n-tile's stream, and drop them)
constexpr size_t R_FRAG_PAD = 3 * 4 * 512;
size_t split_frag_elems(int N, int Cin) { return (size_t)((N + 31) / 32) * (Cin / 32) * 9 * 4 * 512 + R_FRAG_PAD; }
hipError_t launch_pack_split_frag(const float* w_tapmajor, half_t* out, int N, int Cin, hipStream_t st) {
    const size_t total = split_frag_elems(N, Cin) - R_FRAG_PAD;
    pack_split_frag_kernel<<<(int)std::min<size_t>((total + 255) / 256, 8192), 256, 0, st>>>(w_tapmajor, out, N, Cin, total);
    return hipGetLastError();
}

// ABL: ablation switches of tools/micro/bench_split (0 in the product): 1 no epilogue, 2 no patch DMA, 4 no fragment reads / loads
template <bool NCHW, int BN, int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_split_stream_kernel(GemmArgs g) {
    static_assert(BN == 128 || (BN == 32 && NCHW), "the 32-channel variant exists for the NCHW conv_out store only");
    constexpr int FI = BN == 128 ? 2 : 1, FJ = BN == 128 ? 2 : 1;   // BN = 32: 4 waves along the pixels, 32 pixels x 32 channels each
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_raw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = BN == 128 ? wave >> 1 : wave, wn = BN == 128 ? wave & 1 : 0;
    const int fr = lane & 31, fh = lane >> 5;
    int tile_m, tile_n;
    r_xcd_tile(tile_m, tile_n);
    const int n0 = tile_n * BN;
    const int tiles_x = g.W / R_TX, tiles_y = g.H / R_TY;
    const int img = tile_m / (tiles_x * tiles_y);
    const int trem = tile_m - img * (tiles_x * tiles_y);
    const int ty0 = (trem / tiles_x) * R_TY, tx0 = (trem % tiles_x) * R_TX;
    const int Hin = g.H >> g.upsample, Win = g.W >> g.upsample;
    const half_t* Abase = reinterpret_cast<const half_t*>(g.A);                 // [pixel][hi Cin | lo Cin]
    const half_t* zero = reinterpret_cast<const half_t*>(g.zero_page);
    const int NC = g.Cin / 32, KT = NC * 9;

    // ---- patch DMA: 24 pieces of 16 rows x 64 B per chunk, piece id = wave + 4 u (u < 6): one piece per wave at each of taps 0..5.
    //      lane -> (row = 16 piece + lane / 4, slot = lane & 3), source chunk = slot ^ ((row >> 2) & 3).
    constexpr int PPW = 2 * R_PIECES / 4;
    static_assert(PPW == 6, "one piece per wave and tap, taps 0..5");
    const int npp = PPW;
    // (the source offsets are recomputed per piece -- ~20 VALU instructions in the shadow of the MFMAs -- instead of being held in six
    //  registers: at 256 registers they spilled, and a scratch reload in front of a DMA piece costs a vmcnt(0) that drains the filter loads)
    auto patch_src = [&](int c, int u) -> const half_t* {
        const int id = wave + 4 * u, plane = id / R_PIECES, piece = id - plane * R_PIECES;
        int lq = lane >> 2;
        asm volatile("" : "+v"(lq));                    // opaque: keeps hipcc from hoisting the six address computations out of the loop (and spilling them)
        const int q = piece * 16 + lq;
        const int qy = (q * 3641) >> 16, qx = q - qy * R_PITCH;                 // q / 18 for q < 192
        const int iy = ty0 + qy - 1, ix = tx0 + qx - 1;
        const int ch = ((lane & 3) ^ ((q >> 2) & 3)) * 8;
        const bool in = q < R_ROWS && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
        return in ? Abase + (((long long)img * Hin + (iy >> g.upsample)) * Win + (ix >> g.upsample)) * (2 * g.Cin) + ch + plane * g.Cin + c * 32 : zero;
    };
    auto issue_patch_piece = [&](int c, int s, int u) {             // prologue only: LDS-DMA (nothing competes with it there)
        const int id = wave + 4 * u, plane = id / R_PIECES, piece = id - plane * R_PIECES;
        char* dst = lds_raw + (size_t)(s * 2 + plane) * R_PLANE + piece * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)patch_src(c, u),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    // In the main loop the patch pieces travel through a register instead: an LDS-DMA piece cost the issuing wave 150+ cycles there
    // (2753 vs 2236 us on the 512 -> 512 upsampling conv with / without them), a global_load_dwordx4 + ds_write_b128 pair does not block.
    u32x4 pst;                                                      // one piece in flight: loaded at tap t, written to LDS at tap t + 1
    auto load_patch_piece = [&](int c, int u) {
        const half_t* sp = patch_src(c, u);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pst) : "v"(sp));
    };
    auto store_patch_piece = [&](int s, int u, int younger) {       // `younger` loads were issued after the piece's (static): vmcnt retires in order
        const int id = wave + 4 * u, plane = id / R_PIECES, piece = id - plane * R_PIECES;
        const unsigned a = lds_base + (s * 2 + plane) * R_PLANE + piece * 1024 + lane * 16;
        if (younger == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(pst));
        else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(pst) : "n"(2 * (BN == 128 ? 2 : 1)));
        asm volatile("ds_write_b128 %0, %1" :: "v"(a), "v"(pst) : "memory");
    };

    // ---- filter fragments: two streams (j) of 4 KiB per k-tile, [ks][plane] chunks of 1 KiB
    const half_t* bfrag[FJ];
#pragma unroll
    for (int j = 0; j < FJ; ++j) bfrag[j] = reinterpret_cast<const half_t*>(g.Bw_frag) + (size_t)((n0 + wn * 64 + j * 32) / 32) * KT * 2048 + lane * 8;

    f32x16 accm[FI][FJ], accx[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accm[i][j][r] = 0.0f; accx[i][j][r] = 0.0f; }

    int qbase[FI];                                                  // patch row of tap (0, 0) for this lane's pixel of fragment i
#pragma unroll
    for (int i = 0; i < FI; ++i) qbase[i] = (wm * (2 * FI) + i * 2 + (fr >> 4)) * R_PITCH + (fr & 15);

    half8 ah[2][FI], al[2][FI];                                     // patch fragments, set = k-step parity
    half8 wh[2][2][FJ], wl[2][2][FJ];                               // filter fragments [k-tile parity][ks][j]
    auto read_a = [&](int ps, int tapoff, int ks, int set, int i) {
        int q = qbase[i] + tapoff;
        asm volatile("" : "+v"(q));                     // opaque: the 36 fragment addresses of an unrolled chunk are computed where they are used, not hoisted (and spilled)
        const unsigned a = lds_base + ps * 2 * R_PLANE + ((q * 64 + ((fh ^ ((q >> 2) & 3)) << 4)) ^ (ks << 5));
        asm volatile("ds_read_b128 %0, %1" : "=v"(ah[set][i]) : "v"(a));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(al[set][i]) : "v"(a), "n"(R_PLANE));
    };
    // Filter fragments are asm loads INSIDE a chunk (issued in k-tile t, retired by a counted wait in k-tile t + 1: straight-line code)
    // and ORDINARY loads across a chunk boundary.  An asm load whose value crosses a branch is unsafe: hipcc copies the destination
    // registers at the block boundary -- BEFORE the data has landed -- and the load later writes registers that hold something else by
    // then (seen: a DMA source address -> memory fault).  Ordinary loads are counted by hipcc itself; its waits then are conservative
    // about the asm loads it cannot see (vmcnt retires in issue order), never too weak.
    auto load_b = [&](int kt, int par, int ks, int j, bool plain) {
        const half_t* p = bfrag[j] + (size_t)kt * 2048 + ks * 1024;
        if (plain) {
            wh[par][ks][j] = *reinterpret_cast<const half8*>(p);
            wl[par][ks][j] = *reinterpret_cast<const half8*>(p + 512);
        } else {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wh[par][ks][j]) : "v"(p));
            asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(wl[par][ks][j]) : "v"(p));
        }
    };
#define HQT_WAIT_B(par, ks, P)                                                                                                 \
    do {                                                                                                                       \
        if constexpr (FJ == 2)                                                                                                 \
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(wh[par][ks][0]), "+v"(wl[par][ks][0]), "+v"(wh[par][ks][1]), "+v"(wl[par][ks][1]) : "n"(P)); \
        else                                                                                                                   \
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(wh[par][ks][0]), "+v"(wl[par][ks][0]) : "n"(P));                          \
    } while (0)
#define HQT_WAIT_A(set, P)                                                                                                     \
    do {                                                                                                                       \
        if constexpr (FI == 2)                                                                                                 \
            asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(ah[set][0]), "+v"(al[set][0]), "+v"(ah[set][1]), "+v"(al[set][1]) : "n"(P)); \
        else                                                                                                                   \
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(ah[set][0]), "+v"(al[set][0]) : "n"(P));                               \
    } while (0)
    constexpr int NBL = 2 * FJ;                                     // filter loads per k-step (hi + lo per j)

    // one k-step: FI x FJ x 3 MFMAs with `between(slot)` called after each (i, j) group -- the slots carry the prefetches
    auto mfma_step = [&](int aset, int par, int ks, auto&& between) {
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[par][ks][j], ah[aset][i], accm[i][j], 0, 0, 0);
                accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[par][ks][j], al[aset][i], accx[i][j], 0, 0, 0);
                accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[par][ks][j], ah[aset][i], accx[i][j], 0, 0, 0);
                between(i * FJ + j);
            }
    };

    // ---- prologue: the first patch, the filters of k-tile 0, the patch fragments of (k-tile 0, k-step 0)
#pragma unroll
    for (int u = 0; u < PPW; ++u)
        if (u < npp) issue_patch_piece(0, 0, u);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < FJ; ++j) load_b(0, 0, ks, j, true);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NBL) : "memory");        // the DMA pieces are older than the filter loads (vmcnt is in issue order)

    // ---- main loop: one iteration = TWO 32-channel chunks = 18 k-tiles, fully unrolled STRAIGHT-LINE code with static tap offsets, patch
    //      buffers and filter register sets (k-tile parity; 18 is even, so ONE instantiation serves every iteration).  No branch inside:
    //      an asm fragment load and the wait that retires it never have a block boundary between them (see load_b); the only values that
    //      cross the loop's back edge are the accumulators and the filters of the next iteration's first k-tile (ordinary loads).
    //      Issue order inside a k-tile: k-step 0: [A(kt, 1) reads] [one patch DMA piece, taps 0..5] [filters (kt+1, 0), last];
    //      k-step 1: [A(kt+1, 0) reads] [filters (kt+1, 1)].  The patch of chunk c + 1 is fetched during chunk c (of chunk NC - 1 again
    //      during the last one, into the buffer nobody reads any more); the filters of k-tile KT, one past the end, are loaded and
    //      dropped (split_frag_elems pads the buffer).
    constexpr int NS = FI * FJ;
#pragma unroll 1
    for (int c0 = 0; c0 < NC; c0 += 2) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int c = c0 + cc, kt0 = c * 9, cn = min(c + 1, NC - 1);
            // the DMA pieces of this chunk's patch were issued by tap 5 of the previous chunk (the prologue for chunk 0) and are older than
            // loads that have been waited for since; the barrier makes all four waves' pieces visible and closes the previous chunk's reads
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < FI; ++i) read_a(cc, 0, 0, 0, i);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int PAR = (cc + tap) & 1;                     // static: kt = 9 c + tap, c = c0 + cc, c0 even
                const bool last = cc == 1 && tap == 8;              // the filters loaded here are consumed behind the back edge
                const int tapoff = (tap / 3) * R_PITCH + tap % 3, ntapoff = ((tap + 1) / 3) * R_PITCH + (tap + 1) % 3;
                // k-step 0 needs A set 0 and the filters (PAR, k-step 0): all but the NBL youngest loads (k-step 1 of this k-tile) have landed
                if (ABL != 4) { HQT_WAIT_A(0, 0); HQT_WAIT_B(PAR, 0, NBL); }
                mfma_step(0, PAR, 0, [&](int slot) {
                    if (ABL == 4) return;
                    if (slot % FJ == 0) read_a(cc, tapoff, 1, 1, slot / FJ);                         // A of (this k-tile, k-step 1)
                    if (slot == (FJ > 1 ? 1 : 0) && ABL != 2) {                                        // next chunk's patch: piece `tap` goes out, piece `tap - 1` comes in
                        if (tap >= 1 && tap <= PPW) store_patch_piece(cc ^ 1, tap - 1, 1);          // behind it only the filters of (this k-tile, k-step 1) are in flight
                        if (tap < PPW) load_patch_piece(cn, tap);
                    }
                    if (slot == NS - 1) {                                                            // filters of the next k-tile, k-step 0
                        if (last) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < FJ; ++j) load_b(kt0 + tap + 1, PAR ^ 1, 0, j, last);
                        if (last) __builtin_amdgcn_sched_barrier(0);
                    }
                });
                // k-step 1 needs A set 1 and the filters (PAR, k-step 1): older than the NBL loads issued during k-step 0 (the DMA piece is older still)
                if (ABL != 4) { HQT_WAIT_A(1, 0); HQT_WAIT_B(PAR, 1, NBL); }
                mfma_step(1, PAR, 1, [&](int slot) {
                    if (ABL == 4) return;
                    if (slot % FJ == 0 && tap < 8) read_a(cc, ntapoff, 0, 0, slot / FJ);             // A of (next k-tile, k-step 0): same patch buffer
                    if (slot == NS - 1) {                                                            // filters of the next k-tile, k-step 1
                        if (last) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < FJ; ++j) load_b(kt0 + tap + 1, PAR ^ 1, 1, j, last);
                        if (last) __builtin_amdgcn_sched_barrier(0);
                    }
                });
            }
        }
    }
#undef HQT_WAIT_A
#undef HQT_WAIT_B
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // the patch buffers become the epilogue's staging area
    __builtin_amdgcn_sched_barrier(0);
    if (ABL == 1) {
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
                    float v = (accm[i][j][e] + accx[i][j][e] * R_INV) * g.alpha + (g.bias ? g.bias[n] : 0.0f);
                    if (g.clamp01) v = fminf(fmaxf(0.5f * v + 0.5f, 0.0f), 1.0f);
                    Cb[((long long)img * g.N + n) * hw + pix] = v;
                }
        }
        return;
    }
    if constexpr (BN == 128) {
        // Staged store, 64 pixels (the 4 tile rows of wave row wm = half) at a time: fp32 tile through the dead patch buffers,
        // then whole NHWC rows, two 16-B stores per lane; 256 threads cover 16 pixels x 128 channels per pass.
        const long long pix0 = ((long long)img * g.H + ty0) * g.W + tx0;
        char* stage = lds_raw;
        float* Cb = reinterpret_cast<float*>(g.C);
        const float* Rb = reinterpret_cast<const float*>(g.resid);
        const int c8 = (tid & 15) * 8, nn = n0 + c8;
        float bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = (g.bias && nn + e < g.N) ? g.bias[nn + e] : 0.0f;
        float gs[8], gq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { gs[e] = 0.0f; gq[e] = 0.0f; }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half > 0) __syncthreads();              // the previous half has been read back
            if (wm == half) {
#pragma unroll
                for (int i = 0; i < FI; ++i) {
                    const int r = i * 32 + fr;          // pixel within the staged 64
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            const int nl = wn * 64 + j * 32 + 8 * q4 + 4 * fh;
                            f32x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = accm[i][j][4 * q4 + e] + accx[i][j][4 * q4 + e] * R_INV;
                            *reinterpret_cast<f32x4*>(stage + r * R_CPITCH + nl * 4) = v;
                        }
                }
            }
            __syncthreads();
            if (nn < g.N) {                             // N % 8 == 0
                long long moff[4];
                f32x4 r0[4], r1[4];
#pragma unroll
                for (int p4 = 0; p4 < 4; ++p4) {        // the residual rows of the four passes are fetched together
                    const int r = p4 * 16 + (tid >> 4);
                    moff[p4] = (pix0 + (long long)(half * 4 + (r >> 4)) * g.W + (r & 15)) * g.ldc + nn;
                    if (Rb) { r0[p4] = *reinterpret_cast<const f32x4*>(Rb + moff[p4]); r1[p4] = *reinterpret_cast<const f32x4*>(Rb + moff[p4] + 4); }
                }
#pragma unroll
                for (int p4 = 0; p4 < 4; ++p4) {
                    const int r = p4 * 16 + (tid >> 4);
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + r * R_CPITCH + c8 * 4);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + r * R_CPITCH + c8 * 4 + 16);
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] * g.alpha + bv[e];
                    if (Rb) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] += r0[p4][e]; v[4 + e] += r1[p4][e]; }
                    }
                    const f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                    *reinterpret_cast<f32x4*>(Cb + moff[p4]) = o0;
                    *reinterpret_cast<f32x4*>(Cb + moff[p4] + 4) = o1;
                    if (g.gn_part_out_d) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { gs[e] += v[e]; gq[e] += v[e] * v[e]; }
                    }
                }
            }
        }
        if (g.gn_part_out_d) {                              // uniform branch (kernel argument): barriers are safe here
            __syncthreads();
            float* redw = reinterpret_cast<float*>(lds_raw);                    // [16 pixel rows][128 channels][2]; zeros from idle threads
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                redw[(((tid >> 4) * 128) + c8 + e) * 2] = gs[e];
                redw[(((tid >> 4) * 128) + c8 + e) * 2 + 1] = gq[e];
            }
            __syncthreads();
            const float* red = reinterpret_cast<const float*>(lds_raw);
            if (tid < 128) {
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


// ---------------------------------------------------------------------------------------------
// Fourth generation ("ring"): the same tile, patch and packed filters, but each wave owns 128 pixels x 32 channels (4 x 1 fragments
// instead of 2 x 2).  What that buys, measured against the kernel above (512 -> 512 upsampling conv, tools/micro/bench_split):
//   * vmcnt retires in issue order, so every counted wait for a filter fragment also waits for every older load -- the patch pieces
//     included.  With filters one k-tile (0.6 us of this SIMD's matrix time) ahead, an L2 hit (~0.6 us) barely made it and a patch piece
//     from HBM (1-2 us) stalled the wave at the next filter wait: ~550 us of fragment stalls + ~550 us of patch stalls on a 1740 us
//     matrix stream.
//   * One channel fragment per wave halves the filter registers per k-step (8 instead of 16), so a ring of SIX k-steps fits where two
//     k-tiles did: filters are fetched five k-steps (2.5 k-tiles) ahead, a patch piece has three k-tiles to arrive before anything
//     waits for it, and each filter fragment is fetched by one wave instead of two.
//   * The patch fragments (LDS, ~150 cycles) are single-buffered instead: fragment i of the next k-step is read right after the three
//     MFMAs that consume fragment i of this one were issued, nine MFMAs (288 cycles) before its first use.
// Registers: 128 accumulators + 32 patch fragments + 48 filter fragments (ring of 6) + 12 patch pieces in flight + addresses.
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int G_RING = 6, G_AHEAD = G_RING - 1, G_STEPS = 36;         // ring slots (k-steps), prefetch distance, k-steps per loop body (two chunks)
static_assert(G_STEPS % G_RING == 0, "static ring slots");
// Patch rows are 64 B of data on an 80-B pitch: 8 consecutive pixels then start in 8 different 16-B bank groups (5 q mod 8), so the
// fragment reads need no XOR swizzle -- and without one a fragment address is  base(fragment) + constant(buffer, tap, k-step, plane),
// i.e. an immediate offset: ZERO vector instructions per read (the swizzled layout cost ~7 each, 28+ per k-step, and vector
// instructions are not hidden behind this wave's or its SIMD neighbour's MFMAs).
constexpr int G_PITCH = 80, G_PLANE = 16 * R_PIECES * G_PITCH, G_LDS = 4 * G_PLANE;       // 15 KiB per plane, 60 KiB per workgroup
static_assert(64 * R_CPITCH <= G_LDS, "epilogue staging (64 pixels at a time) must fit in the patch buffers");
static_assert(3 * G_PLANE + 38 * G_PITCH + 32 < 65536, "ds_read immediate offsets");
// patch piece loaded during body step u (k-step 1 of taps 0..5)?
constexpr bool g_piece_at(int u) { u = ((u % G_STEPS) + G_STEPS) % G_STEPS; return (u & 1) && ((u % 18) >> 1) < 6; }
// loads issued after the filters of body step s (fetched during step s - G_AHEAD, first hook) and before step s begins
constexpr int g_younger(int s) {
    int n = 2 * (G_AHEAD - 1);
    for (int u = s - G_AHEAD; u < s; ++u) n += g_piece_at(u) ? 1 : 0;
    return n;
}
}  // namespace

// ABL: ablation switches of tools/micro/bench_split (0 in the product): 1 no epilogue, 2 no patch pieces, 4 MFMAs only, 5 patch fragment
// reads only (no filter loads, no pieces), 6 filter loads only (no fragment reads, no pieces)
template <int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_split_ring_kernel(GemmArgs g) {
    constexpr int FI = 4;
    constexpr bool A_ONLY = ABL == 5 || ABL == 7 || ABL == 8;       // 7: + no chunk barrier, 8: + hi-plane reads only (timing experiments)
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_raw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    int tile_m, tile_n;
    r_xcd_tile(tile_m, tile_n);
    const int n0 = tile_n * 128;
    const int tiles_x = g.W / R_TX, tiles_y = g.H / R_TY;
    const int img = tile_m / (tiles_x * tiles_y);
    const int trem = tile_m - img * (tiles_x * tiles_y);
    const int ty0 = (trem / tiles_x) * R_TY, tx0 = (trem % tiles_x) * R_TX;
    const int Hin = g.H >> g.upsample, Win = g.W >> g.upsample;
    const half_t* Abase = reinterpret_cast<const half_t*>(g.A);                 // [pixel][hi Cin | lo Cin]
    const int NC = g.Cin / 32;

    // ---- patch pieces: 24 pieces of 16 rows x 64 B per chunk (12 per plane), piece id = wave + 4 u (u < 6): plane u / 3, piece
    //      wave + 4 (u % 3); lane -> (row = 16 piece + lane / 4, 16-B slot = lane & 3).  Fetched with buffer loads: a 32-bit byte offset
    //      into THIS image's planes (<= 2 GiB) + the chunk's 64 B as the scalar offset, and the hardware's range check returns zeros for
    //      the padding ring (offset 2^31 >= num_records): no address arithmetic in the loop, no select.
    constexpr int PPW = 2 * R_PIECES / 4;
    static_assert(PPW == 6, "one piece per wave at taps 0..5");
    typedef int rsrc_t __attribute__((ext_vector_type(4)));
    rsrc_t img_rsrc;
    {
        const unsigned long long ib = (unsigned long long)(size_t)(Abase + (long long)img * Hin * Win * (2 * g.Cin));
        img_rsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)ib);
        img_rsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(ib >> 32) & 0xffff);      // stride 0
        img_rsrc[2] = __builtin_amdgcn_readfirstlane(Hin * Win * 2 * g.Cin * 2);               // bytes
        img_rsrc[3] = 0x00020000;                                                              // raw buffer, 32-bit data format (gfx9)
    }
    unsigned poff[PPW];                                             // source byte offset of piece u at chunk 0
#pragma unroll
    for (int u = 0; u < PPW; ++u) {
        const int plane = u / 3, piece = wave + 4 * (u % 3);
        const int q = piece * 16 + (lane >> 2);
        const int qy = (q * 3641) >> 16, qx = q - qy * R_PITCH;                 // q / 18 for q < 192
        const int iy = ty0 + qy - 1, ix = tx0 + qx - 1;
        const bool in = (q < R_ROWS) & ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
        const unsigned off = (unsigned)((((iy >> g.upsample) * Win + (ix >> g.upsample)) * (2 * g.Cin) + (lane & 3) * 8 + plane * g.Cin) * 2);
        poff[u] = in ? off : 0x80000000u;
    }
    u32x4 pst[3];                                                   // pieces in flight: loaded at tap t (k-step 1), written to LDS at tap t + 3
    unsigned piece_base = lds_base + wave * (16 * G_PITCH) + (lane >> 2) * G_PITCH + (lane & 3) * 16;     // + (buffer, u) constant
    auto load_piece = [&](int c, int u, int r) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(pst[r]) : "v"(poff[u]), "s"(img_rsrc), "s"(c * 64));
    };
#define HQT_STORE_PIECE(buf, u, r)                                                                                             \
    asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(piece_base), "v"(pst[r]),                                            \
                 "n"(((buf) * 2 + (u) / 3) * G_PLANE + 4 * ((u) % 3) * 16 * G_PITCH) : "memory")

    // this wave's filter stream: a scalar base (+ 2 KiB per k-step) and one lane offset
    const char* bfrag = reinterpret_cast<const char*>(reinterpret_cast<const half_t*>(g.Bw_frag) + (size_t)(n0 / 32 + wave) * ((size_t)NC * 9 * 2048));
    unsigned lane16 = lane * 16;

    f32x16 accm[FI], accx[FI];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accm[i][r] = 0.0f; accx[i][r] = 0.0f; }
    unsigned abase[FI];                                             // LDS address of this lane's pixel of fragment i (tile rows 2 i, 2 i + 1) at tap (0, 0)
#pragma unroll
    for (int i = 0; i < FI; ++i) abase[i] = lds_base + ((i * 2 + (fr >> 4)) * R_PITCH + (fr & 15)) * G_PITCH + fh * 16;

    half8 ah[FI], al[FI];                                           // patch fragments of the current k-step (refilled one by one)
    half8 wh[G_RING], wl[G_RING];                                   // filter fragments, slot = k-step % 6
#define HQT_READ_A(ps, tapoff, ks, i)                                                                                          \
    do {                                                                                                                       \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ah[i]) : "v"(abase[i]), "n"((ps) * 2 * G_PLANE + (tapoff) * G_PITCH + (ks) * 32));            \
        if (ABL != 8) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(al[i]) : "v"(abase[i]), "n"((ps) * 2 * G_PLANE + (tapoff) * G_PITCH + (ks) * 32 + G_PLANE));  \
    } while (0)
    // Every filter load is an asm load, also the ones consumed behind the loop's back edge: the body is ONE basic block and the ring
    // registers are loop-carried in place (tools/micro/audit_ring.py checks both on the generated ISA -- an in-flight register that
    // hipcc copied or reused would be garbage; tests/test_gpu_split.py would see it too).
    auto load_b = [&](long long S, int slot) {
        const char* p = bfrag + S * 2048;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(wh[slot]) : "v"(lane16), "s"(p));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(wl[slot]) : "v"(lane16), "s"(p));
    };

    // ---- prologue: the first patch (two rounds of three pieces through the piece registers), the filters of k-steps 0 .. 4; everything
    //      has landed before the loop starts, so the counted waits of the first body (which assume the steady-state issue pattern) can
    //      only be too strict, never too weak
#pragma unroll
    for (int u = 0; u < 3; ++u) load_piece(0, u, u);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pst[0]), "+v"(pst[1]), "+v"(pst[2]));
    HQT_STORE_PIECE(0, 0, 0); HQT_STORE_PIECE(0, 1, 1); HQT_STORE_PIECE(0, 2, 2);
#pragma unroll
    for (int u = 3; u < 6; ++u) load_piece(0, u, u - 3);
#pragma unroll
    for (int s = 0; s < G_AHEAD; ++s) load_b(s, s);
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(pst[0]), "+v"(pst[1]), "+v"(pst[2]) : "n"(2 * G_AHEAD));
    HQT_STORE_PIECE(0, 3, 0); HQT_STORE_PIECE(0, 4, 1); HQT_STORE_PIECE(0, 5, 2);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    // ---- main loop: one iteration = two chunks = 36 k-steps of straight-line code with static taps, buffers, ring slots and wait counts
#pragma unroll 1
    for (int c0 = 0; c0 < NC; c0 += 2) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int c = c0 + cc, cn = min(c + 1, NC - 1);
            if (ABL != 7) __builtin_amdgcn_s_barrier(); // every wave's pieces of this chunk's patch are in LDS; the previous chunk's reads are done
            __builtin_amdgcn_sched_barrier(0);
            if (ABL != 4 && ABL != 6) {
                HQT_READ_A(cc, 0, 0, 0); HQT_READ_A(cc, 0, 0, 1); HQT_READ_A(cc, 0, 0, 2); HQT_READ_A(cc, 0, 0, 3);
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int s = cc * 18 + tap * 2 + ks, slot = s % G_RING, nslot = (s + G_AHEAD) % G_RING;
                    const long long S = (long long)c0 * 18 + s;
                    const bool refill = !(tap == 8 && ks == 1);         // the next chunk's first fragments are read after its barrier
                    const int ntap = ks ? tap + 1 : tap;
                    const int ntapoff = (ntap / 3) * R_PITCH + ntap % 3;
                    if (ABL != 4 && !A_ONLY) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(wh[slot]), "+v"(wl[slot]) : "n"(g_younger(s)));
                    // Fragments go in pairs: [m0 m1 x0 x1 x0' x1'] then [m2 m3 x2 x3 x2' x3'] (m = hi.hi, x = hi.lo, x' = lo.hi into the same cross
                    // accumulator): no MFMA directly follows the one whose result it accumulates onto, so ONE wave keeps the matrix pipe busy
                    // and a neighbour delayed by its memory instructions does not open bubbles.
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        const int i0 = 2 * pr, i1 = 2 * pr + 1;
                        // reads in issue order: (s, 0) (s, 1) [mid of step s - 1], (s, 2) (s, 3) [end of step s - 1], (s + 1, 0) (s + 1, 1) [mid of s]
                        if (ABL != 4 && ABL != 6) {
                            if (pr == 0 || refill) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ah[i0]), "+v"(al[i0]), "+v"(ah[i1]), "+v"(al[i1]));
                            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[i0]), "+v"(al[i0]), "+v"(ah[i1]), "+v"(al[i1]));
                        }
                        accm[i0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[slot], ah[i0], accm[i0], 0, 0, 0);
                        accm[i1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[slot], ah[i1], accm[i1], 0, 0, 0);
                        accx[i0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[slot], al[i0], accx[i0], 0, 0, 0);
                        accx[i1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[slot], al[i1], accx[i1], 0, 0, 0);
                        accx[i0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[slot], ah[i0], accx[i0], 0, 0, 0);
                        accx[i1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[slot], ah[i1], accx[i1], 0, 0, 0);
                        if (ABL == 4) continue;
                        if (refill && ABL != 6) {
                            if (pr == 0) { HQT_READ_A(cc, ntapoff, ks ^ 1, 0); HQT_READ_A(cc, ntapoff, ks ^ 1, 1); }
                            else { HQT_READ_A(cc, ntapoff, ks ^ 1, 2); HQT_READ_A(cc, ntapoff, ks ^ 1, 3); }
                        }
                        if (pr == 0 && !A_ONLY) load_b(S + G_AHEAD, nslot);         // the slot k-step s - 1 released takes the filters of k-step s + 5
                        if (pr == 1 && ks == 1 && ABL != 2 && !A_ONLY && ABL != 6) {
                            // next chunk's patch: the piece of tap - 3 is older than the filters just waited for (it has landed) and goes to
                            // LDS; the piece of this tap goes out into the register it frees
                            if (tap >= 3 && tap - 3 < PPW) HQT_STORE_PIECE(cc ^ 1, tap - 3, tap % 3);
                            if (tap < PPW) load_piece(cn, tap, tap % 3);
                        }
                    }
                }
        }
    }
#undef HQT_READ_A
#undef HQT_STORE_PIECE
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // the patch buffers become the epilogue's staging area
    __builtin_amdgcn_sched_barrier(0);
    if (ABL == 1) {
        float sacc = 0.0f;
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc += accm[i][r] + accx[i][r];
        if (sacc == 12345.678f) reinterpret_cast<float*>(g.C)[0] = sacc;
        return;
    }
    // ---- epilogue.  D map: col = lane & 31 -> pixel fr of fragment i; row = (r & 3) + 8 (r >> 2) + 4 fh -> channel within the wave's 32.
    //      Staged store, 64 pixels (fragments 2 half, 2 half + 1) at a time: fp32 tile through the dead patch buffers, then whole NHWC
    //      rows, two 16-B stores per lane; 256 threads cover 16 pixels x 128 channels per pass.
    const long long pix0 = ((long long)img * g.H + ty0) * g.W + tx0;
    char* stage = lds_raw;
    float* Cb = reinterpret_cast<float*>(g.C);
    const float* Rb = reinterpret_cast<const float*>(g.resid);
    const int c8 = (tid & 15) * 8, nn = n0 + c8;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (g.bias && nn + e < g.N) ? g.bias[nn + e] : 0.0f;
    float gs[8], gq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gs[e] = 0.0f; gq[e] = 0.0f; }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half > 0) __syncthreads();                  // the previous half has been read back
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = half * 2 + ii, r = ii * 32 + fr;      // pixel within the staged 64
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int nl = wave * 32 + 8 * q4 + 4 * fh;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = accm[i][4 * q4 + e] + accx[i][4 * q4 + e] * R_INV;
                *reinterpret_cast<f32x4*>(stage + r * R_CPITCH + nl * 4) = v;
            }
        }
        __syncthreads();
        if (nn < g.N) {                                 // N % 8 == 0
            long long moff[4];
            f32x4 r0[4], r1[4];
#pragma unroll
            for (int p4 = 0; p4 < 4; ++p4) {            // the residual rows of the four passes are fetched together
                const int r = p4 * 16 + (tid >> 4);
                moff[p4] = (pix0 + (long long)(half * 4 + (r >> 4)) * g.W + (r & 15)) * g.ldc + nn;
                if (Rb) { r0[p4] = *reinterpret_cast<const f32x4*>(Rb + moff[p4]); r1[p4] = *reinterpret_cast<const f32x4*>(Rb + moff[p4] + 4); }
            }
#pragma unroll
            for (int p4 = 0; p4 < 4; ++p4) {
                const int r = p4 * 16 + (tid >> 4);
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + r * R_CPITCH + c8 * 4);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + r * R_CPITCH + c8 * 4 + 16);
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * g.alpha + bv[e];
                if (Rb) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += r0[p4][e]; v[4 + e] += r1[p4][e]; }
                }
                const f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                *reinterpret_cast<f32x4*>(Cb + moff[p4]) = o0;
                *reinterpret_cast<f32x4*>(Cb + moff[p4] + 4) = o1;
                if (g.gn_part_out_d) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { gs[e] += v[e]; gq[e] += v[e] * v[e]; }
                }
            }
        }
    }
    if (g.gn_part_out_d) {                              // uniform branch (kernel argument): barriers are safe here
        __syncthreads();
        float* redw = reinterpret_cast<float*>(lds_raw);                    // [16 pixel rows][128 channels][2]; zeros from idle threads
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            redw[(((tid >> 4) * 128) + c8 + e) * 2] = gs[e];
            redw[(((tid >> 4) * 128) + c8 + e) * 2 + 1] = gq[e];
        }
        __syncthreads();
        const float* red = reinterpret_cast<const float*>(lds_raw);
        if (tid < 128) {
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


// ---------------------------------------------------------------------------------------------
// Fifth generation ("ring16"): the ring kernel on v_mfma_f32_16x16x32_f16.  The ring kernel keeps the matrix pipe busy 85 % of the
// time (profiles/r02_pmc_mfma_util_*.txt) -- but under that load the chip holds its clock near 1.5 GHz instead of 2.4
// (GRBM_GUI_ACTIVE / 8 / duration), so what is left is the energy per MFMA, not the issue stream; MI355X_MICROARCH.md ('DVFS give-back',
// item 7) measures the 16x16x32 shape ~1.12-1.15x faster by wall than 32x32x16 at equal cycles per FLOP with operands read from LDS.
// Same tile, patch image, piece schedule and epilogue staging; per wave 8 pixel blocks (the 8 tile rows of 16 pixels) x 2 channel blocks
// of 16; one MFMA covers a whole 32-channel chunk of a tap, so a "step" is a tap: 48 MFMAs of 16 cycles against 16 fragment reads,
// 4 filter loads and <= 1 patch piece.  Filters are packed per 16-channel block ([32-channel group][chunk][tap][block][hi | lo],
// 1 KiB each: lane l holds W[16 block + (l & 15)][tap Cin + 32 c + 8 (l >> 4) + j]), ring of 3 taps (48 registers).
// ---------------------------------------------------------------------------------------------
__global__ void pack_split_frag16_kernel(const float* __restrict__ w, half_t* __restrict__ out, int N, int Cin, size_t total) {
    const int NC = Cin / 32;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        size_t ch = i >> 9;
        const int plane = (int)(ch & 1); ch >>= 1;
        const int blk = (int)(ch & 1); ch >>= 1;
        const int tap = (int)(ch % 9); ch /= 9;
        const int c = (int)(ch % NC);
        const int t = (int)(ch / NC);
        const int n = t * 32 + blk * 16 + (lane & 15), k = tap * Cin + c * 32 + 8 * (lane >> 4) + j;
        const float x = n < N ? w[(size_t)n * 9 * Cin + k] : 0.0f;
        const half_t hi = (half_t)x;
        out[i] = plane ? (half_t)((x - (float)hi) * 2048.0f) : hi;
    }
}
hipError_t launch_pack_split_frag16(const float* w_tapmajor, half_t* out, int N, int Cin, hipStream_t st) {
    const size_t total = split_frag_elems(N, Cin) - R_FRAG_PAD;
    pack_split_frag16_kernel<<<(int)std::min<size_t>((total + 255) / 256, 8192), 256, 0, st>>>(w_tapmajor, out, N, Cin, total);
    return hipGetLastError();
}

namespace {
constexpr int H_RING = 3, H_AHEAD = H_RING - 1, H_STEPS = 18;         // ring slots (taps), prefetch distance, taps per loop body (two chunks)
static_assert(H_STEPS % H_RING == 0, "static ring slots");
constexpr bool h_piece_at(int u) { u = ((u % H_STEPS) + H_STEPS) % H_STEPS; return (u % 9) < 6; }
// loads issued after the filters of body step s (fetched during step s - 2, first hook) and before step s begins
constexpr int h_younger(int s) {
    int n = 4 * (H_AHEAD - 1);
    for (int u = s - H_AHEAD; u < s; ++u) n += h_piece_at(u) ? 1 : 0;
    return n;
}
}  // namespace

template <int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_split_ring16_kernel(GemmArgs g) {
    constexpr int NI = 8;                                           // pixel blocks = tile rows
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_raw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fx = lane & 15, fk = lane >> 4;
    int tile_m, tile_n;
    r_xcd_tile(tile_m, tile_n);
    const int n0 = tile_n * 128;
    const int tiles_x = g.W / R_TX, tiles_y = g.H / R_TY;
    const int img = tile_m / (tiles_x * tiles_y);
    const int trem = tile_m - img * (tiles_x * tiles_y);
    const int ty0 = (trem / tiles_x) * R_TY, tx0 = (trem % tiles_x) * R_TX;
    const int Hin = g.H >> g.upsample, Win = g.W >> g.upsample;
    const half_t* Abase = reinterpret_cast<const half_t*>(g.A);                 // [pixel][hi Cin | lo Cin]
    const int NC = g.Cin / 32;

    // ---- patch pieces: exactly as in the ring kernel
    constexpr int PPW = 2 * R_PIECES / 4;
    static_assert(PPW == 6, "one piece per wave at taps 0..5");
    typedef int rsrc_t __attribute__((ext_vector_type(4)));
    rsrc_t img_rsrc;
    {
        const unsigned long long ib = (unsigned long long)(size_t)(Abase + (long long)img * Hin * Win * (2 * g.Cin));
        img_rsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)ib);
        img_rsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(ib >> 32) & 0xffff);      // stride 0
        img_rsrc[2] = __builtin_amdgcn_readfirstlane(Hin * Win * 2 * g.Cin * 2);               // bytes
        img_rsrc[3] = 0x00020000;                                                              // raw buffer, 32-bit data format (gfx9)
    }
    unsigned poff[3];                                               // hi-plane source offset of pieces wave + 4 (u % 3); the lo plane is + Cin halves (scalar offset)
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int plane = 0, piece = wave + 4 * u;
        const int q = piece * 16 + (lane >> 2);
        const int qy = (q * 3641) >> 16, qx = q - qy * R_PITCH;                 // q / 18 for q < 192
        const int iy = ty0 + qy - 1, ix = tx0 + qx - 1;
        const bool in = (q < R_ROWS) & ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
        const unsigned off = (unsigned)((((iy >> g.upsample) * Win + (ix >> g.upsample)) * (2 * g.Cin) + (lane & 3) * 8 + plane * g.Cin) * 2);
        poff[u] = in ? off : 0x80000000u;
    }
    u32x4 pst[2];                                                   // pieces in flight: loaded at tap t, written to LDS at tap t + 2
    unsigned piece_base = lds_base + wave * (16 * G_PITCH) + (lane >> 2) * G_PITCH + (lane & 3) * 16;
    auto load_piece = [&](int c, int u, int r) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(pst[r]) : "v"(poff[u % 3]), "s"(img_rsrc), "s"(c * 64 + (u / 3) * g.Cin * 2));
    };
#define HQT_STORE_PIECE(buf, u, r)                                                                                             \
    asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(piece_base), "v"(pst[r]),                                            \
                 "n"(((buf) * 2 + (u) / 3) * G_PLANE + 4 * ((u) % 3) * 16 * G_PITCH) : "memory")

    // this wave's filter stream: 4 KiB per tap ([block 0 hi][block 0 lo][block 1 hi][block 1 lo])
    const char* bfrag = reinterpret_cast<const char*>(reinterpret_cast<const half_t*>(g.Bw_frag16) + (size_t)(n0 / 32 + wave) * ((size_t)NC * 9 * 2048));
    unsigned lane16 = lane * 16;

    f32x4 accm[NI][2], accx[NI][2];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { accm[i][j][r] = 0.0f; accx[i][j][r] = 0.0f; }
    // fragment of pixel block i at tap (dy, dx): patch row (i + dy) * 18 + fx + dx, 16-B group fk -- one base + immediates
    const unsigned abase = lds_base + fx * G_PITCH + fk * 16;
    half8 ah[4], al[4];                                             // pixel blocks in flight: slot = block % 4
    half8 wh[H_RING][2], wl[H_RING][2];                             // filter fragments [tap % 3][channel block]
#define HQT_READ_A16(ps, tapoff, i)                                                                                            \
    do {                                                                                                                       \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ah[(i) % 4]) : "v"(abase), "n"((ps) * 2 * G_PLANE + ((i) * R_PITCH + (tapoff)) * G_PITCH));            \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(al[(i) % 4]) : "v"(abase), "n"((ps) * 2 * G_PLANE + ((i) * R_PITCH + (tapoff)) * G_PITCH + G_PLANE));  \
    } while (0)
    auto load_b = [&](long long S, int slot) {
        const char* p = bfrag + S * 4096;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(wh[slot][0]) : "v"(lane16), "s"(p));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(wl[slot][0]) : "v"(lane16), "s"(p));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(wh[slot][1]) : "v"(lane16), "s"(p));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=v"(wl[slot][1]) : "v"(lane16), "s"(p));
    };

    // ---- prologue (as the ring kernel): first patch through the piece registers, filters of taps 0 and 1; everything lands first
#pragma unroll
    for (int rnd = 0; rnd < 3; ++rnd) {
        load_piece(0, 2 * rnd, 0); load_piece(0, 2 * rnd + 1, 1);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pst[0]), "+v"(pst[1]));
        if (rnd == 0) { HQT_STORE_PIECE(0, 0, 0); HQT_STORE_PIECE(0, 1, 1); }
        else if (rnd == 1) { HQT_STORE_PIECE(0, 2, 0); HQT_STORE_PIECE(0, 3, 1); }
        else { HQT_STORE_PIECE(0, 4, 0); HQT_STORE_PIECE(0, 5, 1); }
    }
#pragma unroll
    for (int s = 0; s < H_AHEAD; ++s) load_b(s, s);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    // ---- main loop: one iteration = two chunks = 18 taps of straight-line code
#pragma unroll 1
    for (int c0 = 0; c0 < NC; c0 += 2) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int c = c0 + cc, cn = min(c + 1, NC - 1);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (ABL != 4) { HQT_READ_A16(cc, 0, 0); HQT_READ_A16(cc, 0, 1); HQT_READ_A16(cc, 0, 2); HQT_READ_A16(cc, 0, 3); }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int s = cc * 9 + tap, slot = s % H_RING, nslot = (s + H_AHEAD) % H_RING;
                const long long S = (long long)c0 * 9 + s;
                const int tapoff = (tap / 3) * R_PITCH + tap % 3, ntapoff = ((tap + 1) / 3) * R_PITCH + (tap + 1) % 3;
                if (ABL != 4) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(wh[slot][0]), "+v"(wl[slot][0]), "+v"(wh[slot][1]), "+v"(wl[slot][1]) : "n"(h_younger(s)));
                // pixel blocks in pairs [m m m m x x x x x' x' x' x']: no MFMA directly follows the one it accumulates onto
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    const int i0 = 2 * pr, i1 = 2 * pr + 1;
                    // reads in issue order: blocks (2 pr, 2 pr + 1) of this tap were issued two pairs ago; behind them: the next pair (4 reads)
                    if (ABL != 4) {
                        const bool more = !(tap == 8 && pr == 3);       // the last pair of a chunk has nothing behind it
                        if (more) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ah[i0 % 4]), "+v"(al[i0 % 4]), "+v"(ah[i1 % 4]), "+v"(al[i1 % 4]));
                        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[i0 % 4]), "+v"(al[i0 % 4]), "+v"(ah[i1 % 4]), "+v"(al[i1 % 4]));
                    }
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            accm[i0 + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[slot][j], ah[(i0 + ii) % 4], accm[i0 + ii][j], 0, 0, 0);
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            accx[i0 + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[slot][j], al[(i0 + ii) % 4], accx[i0 + ii][j], 0, 0, 0);
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            accx[i0 + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[slot][j], ah[(i0 + ii) % 4], accx[i0 + ii][j], 0, 0, 0);
                    if (ABL == 4) continue;
                    // refill the two slots: blocks (i0 + 4, i1 + 4) of this tap, or blocks (i0 - 4, i1 - 4) of the next tap
                    if (pr < 2) {
                        if (pr == 0) { HQT_READ_A16(cc, tapoff, 4); HQT_READ_A16(cc, tapoff, 5); }
                        else { HQT_READ_A16(cc, tapoff, 6); HQT_READ_A16(cc, tapoff, 7); }
                    } else if (tap < 8) {
                        if (pr == 2) { HQT_READ_A16(cc, ntapoff, 0); HQT_READ_A16(cc, ntapoff, 1); }
                        else { HQT_READ_A16(cc, ntapoff, 2); HQT_READ_A16(cc, ntapoff, 3); }
                    }
                    if (pr == 0) load_b(S + H_AHEAD, nslot);            // the slot tap s - 1 released takes the filters of tap s + 2
                    if (pr == 1 && ABL != 2) {
                        // the piece of tap - 2: behind it were issued the filters of tap + 1 (4), the piece of tap - 1 and the filters of tap + 2 (4)
                        if (tap >= 2 && tap - 2 < PPW) {
                            if (tap - 1 < PPW) asm volatile("s_waitcnt vmcnt(9)" : "+v"(pst[tap % 2]));
                            else asm volatile("s_waitcnt vmcnt(8)" : "+v"(pst[tap % 2]));
                            HQT_STORE_PIECE(cc ^ 1, tap - 2, tap % 2);
                        }
                        if (tap < PPW) load_piece(cn, tap, tap % 2);
                    }
                }
            }
        }
    }
#undef HQT_READ_A16
#undef HQT_STORE_PIECE
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // the patch buffers become the epilogue's staging area
    __builtin_amdgcn_sched_barrier(0);
    if (ABL == 1) {
        float sacc = 0.0f;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) sacc += accm[i][j][r] + accx[i][j][r];
        if (sacc == 12345.678f) reinterpret_cast<float*>(g.C)[0] = sacc;
        return;
    }
    // ---- epilogue.  D map: col = lane & 15 -> pixel fx of tile row i; row = 4 (lane >> 4) + r -> channel 16 j + 4 fk + r of the wave's 32.
    //      Staged store, 64 pixels (tile rows 4 half .. 4 half + 3) at a time, then as the ring kernel.
    const long long pix0 = ((long long)img * g.H + ty0) * g.W + tx0;
    char* stage = lds_raw;
    float* Cb = reinterpret_cast<float*>(g.C);
    const float* Rb = reinterpret_cast<const float*>(g.resid);
    const int c8 = (tid & 15) * 8, nn = n0 + c8;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (g.bias && nn + e < g.N) ? g.bias[nn + e] : 0.0f;
    float gs[8], gq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gs[e] = 0.0f; gq[e] = 0.0f; }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half > 0) __syncthreads();                  // the previous half has been read back
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = half * 4 + ii, r = ii * 16 + fx;      // pixel within the staged 64
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nl = wave * 32 + j * 16 + 4 * fk;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = accm[i][j][e] + accx[i][j][e] * R_INV;
                *reinterpret_cast<f32x4*>(stage + r * R_CPITCH + nl * 4) = v;
            }
        }
        __syncthreads();
        if (nn < g.N) {                                 // N % 8 == 0
            long long moff[4];
            f32x4 r0[4], r1[4];
#pragma unroll
            for (int p4 = 0; p4 < 4; ++p4) {            // the residual rows of the four passes are fetched together
                const int r = p4 * 16 + (tid >> 4);
                moff[p4] = (pix0 + (long long)(half * 4 + (r >> 4)) * g.W + (r & 15)) * g.ldc + nn;
                if (Rb) { r0[p4] = *reinterpret_cast<const f32x4*>(Rb + moff[p4]); r1[p4] = *reinterpret_cast<const f32x4*>(Rb + moff[p4] + 4); }
            }
#pragma unroll
            for (int p4 = 0; p4 < 4; ++p4) {
                const int r = p4 * 16 + (tid >> 4);
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + r * R_CPITCH + c8 * 4);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + r * R_CPITCH + c8 * 4 + 16);
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * g.alpha + bv[e];
                if (Rb) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += r0[p4][e]; v[4 + e] += r1[p4][e]; }
                }
                const f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                *reinterpret_cast<f32x4*>(Cb + moff[p4]) = o0;
                *reinterpret_cast<f32x4*>(Cb + moff[p4] + 4) = o1;
                if (g.gn_part_out_d) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { gs[e] += v[e]; gq[e] += v[e] * v[e]; }
                }
            }
        }
    }
    if (g.gn_part_out_d) {                              // uniform branch (kernel argument): barriers are safe here
        __syncthreads();
        float* redw = reinterpret_cast<float*>(lds_raw);                    // [16 pixel rows][128 channels][2]; zeros from idle threads
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            redw[(((tid >> 4) * 128) + c8 + e) * 2] = gs[e];
            redw[(((tid >> 4) * 128) + c8 + e) * 2 + 1] = gq[e];
        }
        __syncthreads();
        const float* red = reinterpret_cast<const float*>(lds_raw);
        if (tid < 128) {
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


// ---------------------------------------------------------------------------------------------
// Upsampling convolution (stage1/modules/layers.py:42-53: nearest x2, then a 3x3 'same' conv) as FOUR 2x2 convolutions on the
// LOW-resolution image.  Output pixel (2 y + a, 2 x + b) sees, through its 3x3 window on the upsampled image, only the low-resolution
// pixels of rows {y - 1 + a, y + a} and columns {x - 1 + b, x + b}; taps that land on the same source pixel are added up once, at
// finalize (pack_split_up16_kernel):
//     rows:  a = 0: [w0 | w1 + w2]     a = 1: [w0 + w1 | w2]      (columns alike with b)
// The same sums as the reference, regrouped -- 4 multiply-adds per output and input channel instead of 9 (2.25x fewer MFMAs on 36 % of
// the decoder's convolution work), no approximation beyond the rounding of the summed filters, which are split into hi / lo like any
// other.  Zero padding carries over: row -1 / H of the low-resolution image is exactly what rows -1 / 2 H of the upsampled one are.
// Kernel = conv3x3_split_ring16_kernel with a workgroup = an 8 x 16 tile of LOW-resolution pixels x 128 channels of ONE phase (a, b)
// (grid.x = 4 phases x N / 128), the same (8 + 2) x (16 + 2) patch image, 4 taps per 32-channel chunk at patch offsets (a + ty, b + tx);
// with a chunk only 4 taps long the six patch pieces of the next chunk are fetched at taps 0, 1 (three each) and written at taps 2, 3,
// and the filters run ONE tap ahead (ring of 2: 32 registers) -- the second workgroup of the CU covers what that exposes.
// ---------------------------------------------------------------------------------------------
__global__ void pack_split_up16_kernel(const float* __restrict__ w, half_t* __restrict__ out, int N, int Cin, size_t total) {
    const int NC = Cin / 32, NG = (N + 31) / 32;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        size_t ch = i >> 9;
        const int plane = (int)(ch & 1); ch >>= 1;
        const int blk = (int)(ch & 1); ch >>= 1;
        const int tap = (int)(ch & 3); ch >>= 2;
        const int c = (int)(ch % NC); ch /= NC;
        const int t = (int)(ch % NG);
        const int phase = (int)(ch / NG);
        const int pa = phase >> 1, pb = phase & 1, ty = tap >> 1, tx = tap & 1;
        const int n = t * 32 + blk * 16 + (lane & 15), kin = c * 32 + 8 * (lane >> 4) + j;
        // 3x3 taps that fall on low-resolution offset ty (tx) for phase pa (pb)
        const int ky0 = pa == 0 ? (ty == 0 ? 0 : 1) : (ty == 0 ? 0 : 2), ky1 = pa == 0 ? (ty == 0 ? 0 : 2) : (ty == 0 ? 1 : 2);
        const int kx0 = pb == 0 ? (tx == 0 ? 0 : 1) : (tx == 0 ? 0 : 2), kx1 = pb == 0 ? (tx == 0 ? 0 : 2) : (tx == 0 ? 1 : 2);
        double x = 0.0;
        if (n < N)
            for (int ky = ky0; ky <= ky1; ++ky)
                for (int kx = kx0; kx <= kx1; ++kx) x += (double)w[(size_t)n * 9 * Cin + (size_t)(ky * 3 + kx) * Cin + kin];
        const half_t hi = (half_t)(float)x;
        out[i] = plane ? (half_t)(float)((x - (double)(float)hi) * 2048.0) : hi;
    }
}
// 4 phases x ceil(N / 32) channel groups x NC chunks x 4 taps x 4 KiB (+ one tap of padding: the filters run one tap ahead)
size_t split_up_elems(int N, int Cin) { return (size_t)4 * ((N + 31) / 32) * (Cin / 32) * 4 * 2048 + 2048; }
hipError_t launch_pack_split_up16(const float* w_tapmajor, half_t* out, int N, int Cin, hipStream_t st) {
    const size_t total = split_up_elems(N, Cin) - 2048;
    pack_split_up16_kernel<<<(int)std::min<size_t>((total + 255) / 256, 8192), 256, 0, st>>>(w_tapmajor, out, N, Cin, total);
    return hipGetLastError();
}

template <int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv2x2_split_up16_kernel(GemmArgs g) {
    constexpr int NI = 8, TAPS = 4;                                 // pixel blocks = tile rows; taps per chunk
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_raw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fx = lane & 15, fk = lane >> 4;
    int tile_m, tile_v;
    r_xcd_tile(tile_m, tile_v);
    const int NT = g.N / 128;                                       // real 128-channel tiles; the grid is 4 NT wide
    const int phase = tile_v / NT, pa = phase >> 1, pb = phase & 1;
    const int n0 = (tile_v - phase * NT) * 128;
    const int Hin = g.H >> 1, Win = g.W >> 1;                       // g.H / g.W: the OUTPUT
    const int tiles_x = Win / R_TX, tiles_y = Hin / R_TY;
    const int img = tile_m / (tiles_x * tiles_y);
    const int trem = tile_m - img * (tiles_x * tiles_y);
    const int ty0 = (trem / tiles_x) * R_TY, tx0 = (trem % tiles_x) * R_TX;     // low-resolution pixels
    const half_t* Abase = reinterpret_cast<const half_t*>(g.A);                 // [pixel][hi Cin | lo Cin]
    const int NC = g.Cin / 32;

    // ---- patch pieces: the (8 + 2) x (16 + 2) low-resolution neighbourhood, as in the ring kernels
    constexpr int PPW = 2 * R_PIECES / 4;
    static_assert(PPW == 6, "six pieces per wave and chunk");
    typedef int rsrc_t __attribute__((ext_vector_type(4)));
    rsrc_t img_rsrc;
    {
        const unsigned long long ib = (unsigned long long)(size_t)(Abase + (long long)img * Hin * Win * (2 * g.Cin));
        img_rsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)ib);
        img_rsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(ib >> 32) & 0xffff);      // stride 0
        img_rsrc[2] = __builtin_amdgcn_readfirstlane(Hin * Win * 2 * g.Cin * 2);               // bytes
        img_rsrc[3] = 0x00020000;                                                              // raw buffer, 32-bit data format (gfx9)
    }
    unsigned poff[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int piece = wave + 4 * u;
        const int q = piece * 16 + (lane >> 2);
        const int qy = (q * 3641) >> 16, qx = q - qy * R_PITCH;                 // q / 18 for q < 192
        const int iy = ty0 + qy - 1, ix = tx0 + qx - 1;
        const bool in = (q < R_ROWS) & ((unsigned)iy < (unsigned)Hin) & ((unsigned)ix < (unsigned)Win);
        const unsigned off = (unsigned)(((iy * Win + ix) * (2 * g.Cin) + (lane & 3) * 8) * 2);
        poff[u] = in ? off : 0x80000000u;
    }
    u32x4 pst[PPW];                                                 // the six pieces of the next chunk: loaded at taps 0 / 1, written at taps 2 / 3
    unsigned piece_base = lds_base + wave * (16 * G_PITCH) + (lane >> 2) * G_PITCH + (lane & 3) * 16;
    auto load_piece = [&](int c, int u) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(pst[u]) : "v"(poff[u % 3]), "s"(img_rsrc), "s"(c * 64 + (u / 3) * g.Cin * 2));
    };
#define HQT_STORE_PIECE(buf, u)                                                                                                \
    asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(piece_base), "v"(pst[u]),                                            \
                 "n"(((buf) * 2 + (u) / 3) * G_PLANE + 4 * ((u) % 3) * 16 * G_PITCH) : "memory")

    // this wave's filter stream: 4 KiB per tap ([block 0 hi][block 0 lo][block 1 hi][block 1 lo]), 4 taps per chunk
    const char* bfrag = reinterpret_cast<const char*>(reinterpret_cast<const half_t*>(g.Bw_up16) +
                                                      ((size_t)phase * (g.N / 32) + n0 / 32 + wave) * ((size_t)NC * TAPS * 2048));
    unsigned lane16 = lane * 16;

    f32x4 accm[NI][2], accx[NI][2];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { accm[i][j][r] = 0.0f; accx[i][j][r] = 0.0f; }
    // fragment of pixel block i at tap (ty, tx) of phase (pa, pb): patch row (i + pa + ty) * 18 + fx + pb + tx -- the phase goes into the base
    const unsigned abase = lds_base + (fx + pa * R_PITCH + pb) * G_PITCH + fk * 16;
    half8 ah[4], al[4];                                             // pixel blocks in flight: slot = block % 4
    half8 wh[2][2], wl[2][2];                                       // filter fragments [tap % 2][channel block]
#define HQT_READ_A16(ps, tapoff, i)                                                                                            \
    do {                                                                                                                       \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ah[(i) % 4]) : "v"(abase), "n"((ps) * 2 * G_PLANE + ((i) * R_PITCH + (tapoff)) * G_PITCH));            \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(al[(i) % 4]) : "v"(abase), "n"((ps) * 2 * G_PLANE + ((i) * R_PITCH + (tapoff)) * G_PITCH + G_PLANE));  \
    } while (0)
    auto load_b = [&](long long S, int slot) {
        const char* p = bfrag + S * 4096;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(wh[slot][0]) : "v"(lane16), "s"(p));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(wl[slot][0]) : "v"(lane16), "s"(p));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(wh[slot][1]) : "v"(lane16), "s"(p));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=v"(wl[slot][1]) : "v"(lane16), "s"(p));
    };

    // ---- prologue: the first patch through the piece registers, the filters of tap 0; everything lands first
#pragma unroll
    for (int u = 0; u < PPW; ++u) load_piece(0, u);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pst[0]), "+v"(pst[1]), "+v"(pst[2]), "+v"(pst[3]), "+v"(pst[4]), "+v"(pst[5]));
    HQT_STORE_PIECE(0, 0); HQT_STORE_PIECE(0, 1); HQT_STORE_PIECE(0, 2); HQT_STORE_PIECE(0, 3); HQT_STORE_PIECE(0, 4); HQT_STORE_PIECE(0, 5);
    load_b(0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    // ---- main loop: one iteration = two chunks = 8 taps of straight-line code.  Vector-memory operations in issue order per tap t:
    //      [filters of tap t + 1: 4] then, at taps 0 and 1, [three pieces].  They retire in that order, so the wait for the filters of
    //      tap t (issued at tap t - 1) leaves 3 younger loads in flight behind taps 0 / 1 and none behind taps 2 / 3 -- by then every
    //      piece has landed too, and the ds_writes at taps 2 / 3 need no wait of their own (the asm operands still carry the dependence).
#pragma unroll 1
    for (int c0 = 0; c0 < NC; c0 += 2) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int c = c0 + cc, cn = min(c + 1, NC - 1);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (ABL != 4) { HQT_READ_A16(cc, 0, 0); HQT_READ_A16(cc, 0, 1); HQT_READ_A16(cc, 0, 2); HQT_READ_A16(cc, 0, 3); }
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int s = cc * TAPS + tap, slot = s % 2, nslot = (s + 1) % 2;
                const long long S = (long long)c0 * TAPS + s;
                const int tapoff = (tap / 2) * R_PITCH + tap % 2, ntapoff = ((tap + 1) / 2) * R_PITCH + (tap + 1) % 2;
                if (ABL != 4) {
                    if (tap == 1 || tap == 2) asm volatile("s_waitcnt vmcnt(3)" : "+v"(wh[slot][0]), "+v"(wl[slot][0]), "+v"(wh[slot][1]), "+v"(wl[slot][1]));
                    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(wh[slot][0]), "+v"(wl[slot][0]), "+v"(wh[slot][1]), "+v"(wl[slot][1]));
                }
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    const int i0 = 2 * pr, i1 = 2 * pr + 1;
                    if (ABL != 4) {
                        const bool more = !(tap == TAPS - 1 && pr == 3);       // the last pair of a chunk has nothing behind it
                        if (more) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ah[i0 % 4]), "+v"(al[i0 % 4]), "+v"(ah[i1 % 4]), "+v"(al[i1 % 4]));
                        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[i0 % 4]), "+v"(al[i0 % 4]), "+v"(ah[i1 % 4]), "+v"(al[i1 % 4]));
                    }
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            accm[i0 + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[slot][j], ah[(i0 + ii) % 4], accm[i0 + ii][j], 0, 0, 0);
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            accx[i0 + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[slot][j], al[(i0 + ii) % 4], accx[i0 + ii][j], 0, 0, 0);
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            accx[i0 + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[slot][j], ah[(i0 + ii) % 4], accx[i0 + ii][j], 0, 0, 0);
                    if (ABL == 4) continue;
                    // the piece written here goes in FRONT of the fragment reads: lgkmcnt(4) of the next pair then covers exactly those reads
                    if (pr >= 1 && tap >= 2 && ABL != 2) {
                        const int u = 3 * (tap - 2) + pr - 1;
                        asm volatile("s_waitcnt vmcnt(4)" : "+v"(pst[u]));      // (already true: see above)
                        if (u == 0) HQT_STORE_PIECE(cc ^ 1, 0); else if (u == 1) HQT_STORE_PIECE(cc ^ 1, 1); else if (u == 2) HQT_STORE_PIECE(cc ^ 1, 2);
                        else if (u == 3) HQT_STORE_PIECE(cc ^ 1, 3); else if (u == 4) HQT_STORE_PIECE(cc ^ 1, 4); else HQT_STORE_PIECE(cc ^ 1, 5);
                    }
                    // refill the two slots: blocks (i0 + 4, i1 + 4) of this tap, or blocks (i0 - 4, i1 - 4) of the next tap
                    if (pr < 2) {
                        if (pr == 0) { HQT_READ_A16(cc, tapoff, 4); HQT_READ_A16(cc, tapoff, 5); }
                        else { HQT_READ_A16(cc, tapoff, 6); HQT_READ_A16(cc, tapoff, 7); }
                    } else if (tap < TAPS - 1) {
                        if (pr == 2) { HQT_READ_A16(cc, ntapoff, 0); HQT_READ_A16(cc, ntapoff, 1); }
                        else { HQT_READ_A16(cc, ntapoff, 2); HQT_READ_A16(cc, ntapoff, 3); }
                    }
                    if (pr == 0) load_b(S + 1, nslot);                  // the slot tap s - 1 released takes the filters of tap s + 1
                    if (pr >= 1 && tap < 2 && ABL != 2) load_piece(cn, 3 * tap + pr - 1);
                }
            }
        }
    }
#undef HQT_READ_A16
#undef HQT_STORE_PIECE
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // the patch buffers become the epilogue's staging area
    __builtin_amdgcn_sched_barrier(0);
    if (ABL == 1) {
        float sacc = 0.0f;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) sacc += accm[i][j][r] + accx[i][j][r];
        if (sacc == 12345.678f) reinterpret_cast<float*>(g.C)[0] = sacc;
        return;
    }
    // ---- epilogue: as the ring16 kernel, with low-resolution tile pixel (row, col) stored at output pixel (2 (ty0 + row) + pa, 2 (tx0 + col) + pb)
    char* stage = lds_raw;
    float* Cb = reinterpret_cast<float*>(g.C);
    const float* Rb = reinterpret_cast<const float*>(g.resid);
    const int c8 = (tid & 15) * 8, nn = n0 + c8;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (g.bias && nn + e < g.N) ? g.bias[nn + e] : 0.0f;
    float gs[8], gq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gs[e] = 0.0f; gq[e] = 0.0f; }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half > 0) __syncthreads();                  // the previous half has been read back
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = half * 4 + ii, r = ii * 16 + fx;      // pixel within the staged 64
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nl = wave * 32 + j * 16 + 4 * fk;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = accm[i][j][e] + accx[i][j][e] * R_INV;
                *reinterpret_cast<f32x4*>(stage + r * R_CPITCH + nl * 4) = v;
            }
        }
        __syncthreads();
        if (nn < g.N) {                                 // N % 8 == 0
            long long moff[4];
            f32x4 r0[4], r1[4];
#pragma unroll
            for (int p4 = 0; p4 < 4; ++p4) {
                const int r = p4 * 16 + (tid >> 4);
                const long long oy = 2 * (ty0 + half * 4 + (r >> 4)) + pa, ox = 2 * (tx0 + (r & 15)) + pb;
                moff[p4] = (((long long)img * g.H + oy) * g.W + ox) * g.ldc + nn;
                if (Rb) { r0[p4] = *reinterpret_cast<const f32x4*>(Rb + moff[p4]); r1[p4] = *reinterpret_cast<const f32x4*>(Rb + moff[p4] + 4); }
            }
#pragma unroll
            for (int p4 = 0; p4 < 4; ++p4) {
                const int r = p4 * 16 + (tid >> 4);
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + r * R_CPITCH + c8 * 4);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + r * R_CPITCH + c8 * 4 + 16);
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * g.alpha + bv[e];
                if (Rb) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += r0[p4][e]; v[4 + e] += r1[p4][e]; }
                }
                const f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                *reinterpret_cast<f32x4*>(Cb + moff[p4]) = o0;
                *reinterpret_cast<f32x4*>(Cb + moff[p4] + 4) = o1;
                if (g.gn_part_out_d) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { gs[e] += v[e]; gq[e] += v[e] * v[e]; }
                }
            }
        }
    }
    if (g.gn_part_out_d) {                              // uniform branch (kernel argument): barriers are safe here
        __syncthreads();
        float* redw = reinterpret_cast<float*>(lds_raw);                    // [16 pixel rows][128 channels][2]; zeros from idle threads
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            redw[(((tid >> 4) * 128) + c8 + e) * 2] = gs[e];
            redw[(((tid >> 4) * 128) + c8 + e) * 2 + 1] = gq[e];
        }
        __syncthreads();
        const float* red = reinterpret_cast<const float*>(lds_raw);
        if (tid < 128) {
            double sa = 0.0, sq = 0.0;
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) { sa += (double)red[((rg * 128) + tid) * 2]; sq += (double)red[((rg * 128) + tid) * 2 + 1]; }
            const int cpg = g.N / g.gn_out_groups;
            for (int off = cpg >> 1; off > 0; off >>= 1) { sa += __shfl_xor(sa, off, 64); sq += __shfl_xor(sq, off, 64); }
            const int ch = n0 + tid;
            if (ch < g.N && (tid & (cpg - 1)) == 0) {       // one partial per (low-resolution tile, phase): 4 tiles_x tiles_y per image
                double* pp = g.gn_part_out_d + ((((long long)img * (tiles_x * tiles_y) + trem) * 4 + phase) * g.gn_out_groups + ch / cpg) * 2;
                pp[0] = sa; pp[1] = sq;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// conv_out (C_in -> 3 channels, NCHW fp32 + clamp) on the ring16 structure: the output channels are padded to ONE 16-channel MFMA block
// (32 in the stream kernel's variant: 10x the matrix work the 3 channels need, 36 % pipe utilisation), so the four waves of a tile split
// its PIXELS: wave w owns tile rows 2 w and 2 w + 1 (2 pixel blocks x 1 channel block: 6 MFMAs per tap) and all of them fetch the same
// filter fragments (L1 hits).  Patch image, piece schedule and waits as in conv3x3_split_ring16_kernel; the fragments of the next tap are
// read one tap ahead (4 register slots: tap parity x pixel block).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void conv3x3_split_out16_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_raw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fx = lane & 15, fk = lane >> 4;
    const int tile_m = blockIdx.y;
    const int tiles_x = g.W / R_TX, tiles_y = g.H / R_TY;
    const int img = tile_m / (tiles_x * tiles_y);
    const int trem = tile_m - img * (tiles_x * tiles_y);
    const int ty0 = (trem / tiles_x) * R_TY, tx0 = (trem % tiles_x) * R_TX;
    const int Hin = g.H >> g.upsample, Win = g.W >> g.upsample;
    const half_t* Abase = reinterpret_cast<const half_t*>(g.A);
    const int NC = g.Cin / 32;

    constexpr int PPW = 2 * R_PIECES / 4;
    typedef int rsrc_t __attribute__((ext_vector_type(4)));
    rsrc_t img_rsrc;
    {
        const unsigned long long ib = (unsigned long long)(size_t)(Abase + (long long)img * Hin * Win * (2 * g.Cin));
        img_rsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)ib);
        img_rsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(ib >> 32) & 0xffff);
        img_rsrc[2] = __builtin_amdgcn_readfirstlane(Hin * Win * 2 * g.Cin * 2);
        img_rsrc[3] = 0x00020000;
    }
    unsigned poff[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int piece = wave + 4 * u;
        const int q = piece * 16 + (lane >> 2);
        const int qy = (q * 3641) >> 16, qx = q - qy * R_PITCH;
        const int iy = ty0 + qy - 1, ix = tx0 + qx - 1;
        const bool in = (q < R_ROWS) & ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
        const unsigned off = (unsigned)((((iy >> g.upsample) * Win + (ix >> g.upsample)) * (2 * g.Cin) + (lane & 3) * 8) * 2);
        poff[u] = in ? off : 0x80000000u;
    }
    u32x4 pst[2];
    unsigned piece_base = lds_base + wave * (16 * G_PITCH) + (lane >> 2) * G_PITCH + (lane & 3) * 16;
    auto load_piece = [&](int c, int u, int r) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(pst[r]) : "v"(poff[u % 3]), "s"(img_rsrc), "s"(c * 64 + (u / 3) * g.Cin * 2));
    };
#define HQT_STORE_PIECE(buf, u, r)                                                                                             \
    asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(piece_base), "v"(pst[r]),                                            \
                 "n"(((buf) * 2 + (u) / 3) * G_PLANE + 4 * ((u) % 3) * 16 * G_PITCH) : "memory")

    // one filter stream for the whole tile: 4 KiB per tap in the 16-channel-block packing, block 0 only ([hi][lo])
    const char* bfrag = reinterpret_cast<const char*>(g.Bw_frag16);
    unsigned lane16 = lane * 16;

    f32x4 accm[2], accx[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { accm[i][r] = 0.0f; accx[i][r] = 0.0f; }
    // this wave's pixel blocks = tile rows 2 wave, 2 wave + 1
    const unsigned abase = lds_base + (2 * wave * R_PITCH + fx) * G_PITCH + fk * 16;
    half8 ah[4], al[4];                                             // [tap parity * 2 + block]
    half8 wh[H_RING], wl[H_RING];
#define HQT_READ_A16(ps, tapoff, i, slot)                                                                                      \
    do {                                                                                                                       \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ah[slot]) : "v"(abase), "n"((ps) * 2 * G_PLANE + ((i) * R_PITCH + (tapoff)) * G_PITCH));            \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(al[slot]) : "v"(abase), "n"((ps) * 2 * G_PLANE + ((i) * R_PITCH + (tapoff)) * G_PITCH + G_PLANE));  \
    } while (0)
    auto load_b = [&](long long S, int slot) {
        const char* p = bfrag + S * 4096;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(wh[slot]) : "v"(lane16), "s"(p));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(wl[slot]) : "v"(lane16), "s"(p));
    };
    // loads issued after the filters of body step s (2 loads, fetched during step s - 2) and before step s begins
#define HQT_YOUNGER(s) (2 * (H_AHEAD - 1) + (h_piece_at((s) - 2) ? 1 : 0) + (h_piece_at((s) - 1) ? 1 : 0))

#pragma unroll
    for (int rnd = 0; rnd < 3; ++rnd) {
        load_piece(0, 2 * rnd, 0); load_piece(0, 2 * rnd + 1, 1);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pst[0]), "+v"(pst[1]));
        if (rnd == 0) { HQT_STORE_PIECE(0, 0, 0); HQT_STORE_PIECE(0, 1, 1); }
        else if (rnd == 1) { HQT_STORE_PIECE(0, 2, 0); HQT_STORE_PIECE(0, 3, 1); }
        else { HQT_STORE_PIECE(0, 4, 0); HQT_STORE_PIECE(0, 5, 1); }
    }
#pragma unroll
    for (int s = 0; s < H_AHEAD; ++s) load_b(s, s);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

#pragma unroll 1
    for (int c0 = 0; c0 < NC; c0 += 2) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int c = c0 + cc, cn = min(c + 1, NC - 1);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            HQT_READ_A16(cc, 0, 0, 0); HQT_READ_A16(cc, 0, 1, 1);       // tap 0 -> parity 0
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int s = cc * 9 + tap, slot = s % H_RING, nslot = (s + H_AHEAD) % H_RING;
                const long long S = (long long)c0 * 9 + s;
                const int par = tap & 1;
                const int ntapoff = ((tap + 1) / 3) * R_PITCH + (tap + 1) % 3;
                // the next tap's fragments first (read one tap ahead into the other parity's slots), then this tap's MFMAs
                if (tap < 8) {
                    if (par == 0) { HQT_READ_A16(cc, ntapoff, 0, 2); HQT_READ_A16(cc, ntapoff, 1, 3); }
                    else { HQT_READ_A16(cc, ntapoff, 0, 0); HQT_READ_A16(cc, ntapoff, 1, 1); }
                }
                asm volatile("s_waitcnt vmcnt(%2)" : "+v"(wh[slot]), "+v"(wl[slot]) : "n"(HQT_YOUNGER(s)));
                if (tap < 8) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ah[2 * par]), "+v"(al[2 * par]), "+v"(ah[2 * par + 1]), "+v"(al[2 * par + 1]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[2 * par]), "+v"(al[2 * par]), "+v"(ah[2 * par + 1]), "+v"(al[2 * par + 1]));
                accm[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[slot], ah[2 * par], accm[0], 0, 0, 0);
                accm[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[slot], ah[2 * par + 1], accm[1], 0, 0, 0);
                accx[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[slot], al[2 * par], accx[0], 0, 0, 0);
                accx[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[slot], al[2 * par + 1], accx[1], 0, 0, 0);
                accx[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[slot], ah[2 * par], accx[0], 0, 0, 0);
                accx[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[slot], ah[2 * par + 1], accx[1], 0, 0, 0);
                load_b(S + H_AHEAD, nslot);
                if (tap >= 2 && tap - 2 < PPW) {
                    if (tap - 1 < PPW) asm volatile("s_waitcnt vmcnt(5)" : "+v"(pst[tap % 2]));     // behind the piece: filters of tap + 1 (2), piece of tap - 1, filters of tap + 2 (2)
                    else asm volatile("s_waitcnt vmcnt(4)" : "+v"(pst[tap % 2]));
                    HQT_STORE_PIECE(cc ^ 1, tap - 2, tap % 2);
                }
                if (tap < PPW) load_piece(cn, tap, tap % 2);
            }
        }
    }
#undef HQT_READ_A16
#undef HQT_STORE_PIECE
#undef HQT_YOUNGER
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // ---- epilogue: D col = lane & 15 -> pixel fx of tile row 2 wave + i; row = 4 fk + r -> channel (< N <= 16)
    float* Cb = reinterpret_cast<float*>(g.C);
    const long long hw = (long long)g.H * g.W;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long long pix = (long long)(ty0 + 2 * wave + i) * g.W + tx0 + fx;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = 4 * fk + r;
            if (n >= g.N) continue;
            float v = (accm[i][r] + accx[i][r] * R_INV) * g.alpha + (g.bias ? g.bias[n] : 0.0f);
            if (g.clamp01) v = fminf(fmaxf(0.5f * v + 0.5f, 0.0f), 1.0f);
            Cb[((long long)img * g.N + n) * hw + pix] = v;
        }
    }
}

bool split_stream_ok(const GemmArgs& g) {
    static const bool off = getenv("HQT_SPLIT_STREAM") && atoi(getenv("HQT_SPLIT_STREAM")) == 0;      // A/B switch
    if (off || !g.Bw_frag) return false;
    if (g.H % R_TY != 0 || g.W % R_TX != 0 || g.Cin % 64 != 0) return false;      // an even number of 32-channel chunks (static k-tile parity)
    // the ring kernels address one image's operand planes with 32-bit byte offsets below 2^31 (buffer loads; 2^31 marks the padding ring)
    if ((long long)(g.H >> g.upsample) * (g.W >> g.upsample) * g.Cin * 4 >= (1ll << 31)) return false;
    if (g.store == STORE_NCHW) return g.N <= 32;
    return g.N % 128 == 0 && g.ldc % 8 == 0;              // whole 128-channel tiles (the packed fragments hold ceil(N / 32) n-tiles)
}
// upsampling conv as four 2x2 phase convolutions on the low-resolution image (conv2x2_split_up16_kernel)
static bool split_up_shape(const GemmArgs& g) {
    static const bool off = getenv("HQT_SPLIT_UP") && atoi(getenv("HQT_SPLIT_UP")) == 0;              // A/B switch: 0 = nine taps on the upsampled image
    return !off && g.upsample == 1 && g.Bw_up16 && g.store == STORE_ROWS && (g.H / 2) % R_TY == 0 && (g.W / 2) % R_TX == 0 && g.N % 128 == 0;
}
int split_stream_tiles_per_image(const GemmArgs& g) { return split_up_shape(g) ? 4 * (g.H / 2 / R_TY) * (g.W / 2 / R_TX) : (g.H / R_TY) * (g.W / R_TX); }
hipError_t launch_split_conv3_stream(const GemmArgs& g, hipStream_t st) {
    if (g.store == STORE_NCHW) {
        static const bool out16 = !(getenv("HQT_SPLIT_OUT16") && atoi(getenv("HQT_SPLIT_OUT16")) == 0);     // A/B switch: 0 = the 32-channel stream variant
        if (out16 && g.Bw_frag16 && g.N <= 16) conv3x3_split_out16_kernel<<<dim3(1, g.M / (R_TY * R_TX), 1), 256, G_LDS, st>>>(g);
        else conv3x3_split_stream_kernel<true, 32><<<dim3(1, g.M / (R_TY * R_TX), 1), 256, R_LDS, st>>>(g);
    }
    else if (split_up_shape(g)) {
        conv2x2_split_up16_kernel<0><<<dim3(4 * (g.N / 128), g.M / 4 / (R_TY * R_TX), 1), 256, G_LDS, st>>>(g);
    }
    else {
        static const bool ring = !(getenv("HQT_SPLIT_RING") && atoi(getenv("HQT_SPLIT_RING")) == 0);     // A/B switch: 0 = the stream kernel
        const dim3 grid((g.N + 127) / 128, g.M / (R_TY * R_TX), 1);
        static const bool ring16 = !(getenv("HQT_SPLIT_RING16") && atoi(getenv("HQT_SPLIT_RING16")) == 0);   // A/B switch: 0 = the 32x32x16 ring kernel
        if (ring && ring16 && g.Bw_frag16) conv3x3_split_ring16_kernel<0><<<grid, 256, G_LDS, st>>>(g);
        else if (ring) conv3x3_split_ring_kernel<0><<<grid, 256, G_LDS, st>>>(g);
        else conv3x3_split_stream_kernel<false, 128><<<grid, 256, R_LDS, st>>>(g);
    }
    return hipGetLastError();
}
hipError_t split_stream_configure() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_split_stream_kernel<false, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_split_ring_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_split_ring16_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2x2_split_up16_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_split_out16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_split_stream_kernel<true, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS);
}
