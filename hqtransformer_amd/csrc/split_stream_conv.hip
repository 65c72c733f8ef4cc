// SPLIT precision: the 3x3 convolutions of the HQ-VAE decoder / encoder on the matrix cores (split_kernels.h).
//
// What this file ships:
//   conv3x3_split_ring16_kernel   3x3 stride-1 'same' conv, N % 128 == 0 (every released config): the default
//   conv2x2_split_up16_kernel     nearest-x2 upsampling conv as four 2x2 phase convolutions on the low-resolution image
//   conv3x3_split_out16_kernel    conv_out (<= 16 output channels, NCHW store + clamp)
// Shapes they do not take fall back to conv3x3_split_kernel (split_conv.hip), and from there to the fp32 vector-ALU kernel.
//
// Design, arrived at over five measured generations (the superseded kernels: git history up to round 3; DESIGN.md 5.2b
// what each one taught):
//   * 4 waves per workgroup, 8 x 16 pixel tile x 128 channels, two workgroups per CU; each wave owns 128 pixels x 32 channels
//     (8 pixel blocks x 2 channel blocks of v_mfma_f32_16x16x32_f16: 128 accumulator registers, main + cross);
//   * the FILTERS never touch LDS: packed at finalize in MFMA A-operand fragment order (1-KiB chunks, [32-channel group][chunk][tap]
//     [16-channel block][hi | lo]), every wave streams the fragments of its own channels straight into registers with coalesced
//     global_load_dwordx4, a ring of three taps ahead of the MFMAs, counted s_waitcnt vmcnt;
//   * the PATCH ((8 + 2) x (16 + 2) pixels x 32 channels, both planes) is double-buffered in LDS on an 80-byte pitch (no swizzle: a
//     fragment address is base + immediate), filled through registers by buffer loads whose range check supplies the zero padding:
//     ONE workgroup barrier per 32-channel chunk;
//   * the main loop is one basic block of straight-line code with asm loads in flight across its back edge; _lib.build() audits the
//     generated ISA (csrc/audit_ring.py) and refuses to link a library whose loops the compiler has touched.
#include "split_kernels.h"
#include "split_device.h"
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

__device__ __forceinline__ void r_xcd_tile(int& tile_m, int& tile_n, int panel) {
    const int nx = gridDim.x, total = gridDim.x * gridDim.y;
    int id = blockIdx.x + nx * blockIdx.y;
    if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
    if (panel > 0) {
        // panels of `panel` pixel tiles x all n-tiles; inside a panel the pixel tile runs fastest: the workgroups an XCD holds at one
        // time walk ONE filter stream together (L2 hits) and each its own patch
        const int per = panel * nx, p = id / per, r = id - p * per;
        const int rows = min(panel, (int)gridDim.y - p * panel);
        tile_n = r / rows;
        tile_m = p * panel + (r - tile_n * rows);
        return;
    }
    tile_m = id / nx;
    tile_n = id - tile_m * nx;
}
}  // namespace

// ---------------------------------------------------------------------------------------------
// Size of a fragment-packed 3x3 filter bank (hi + lo): ceil(N / 32) channel groups x Cin / 32 chunks x 9 taps x 4 KiB
// (+ three k-tiles of padding: the kernels prefetch the filters of up to five k-steps past the end of the last n-tile's stream, and drop them)
constexpr size_t R_FRAG_PAD = 3 * 4 * 512;
size_t split_frag_elems(int N, int Cin) { return (size_t)((N + 31) / 32) * (Cin / 32) * 9 * 4 * 512 + R_FRAG_PAD; }

namespace {
// Patch rows are 64 B of data on an 80-B pitch: 8 consecutive pixels then start in 8 different 16-B bank groups (5 q mod 8), so the
// fragment reads need no XOR swizzle -- and without one a fragment address is  base(fragment) + constant(buffer, tap, plane), i.e. an
// immediate offset: ZERO vector instructions per read (a swizzled layout cost ~7 each, and vector instructions are not hidden behind
// this wave's or its SIMD neighbour's MFMAs).
constexpr int G_PITCH = 80, G_PLANE = 16 * R_PIECES * G_PITCH, G_LDS = 4 * G_PLANE;       // 15 KiB per plane, 60 KiB per workgroup
static_assert(64 * R_CPITCH <= G_LDS, "epilogue staging (64 pixels at a time) must fit in the patch buffers");
static_assert(3 * G_PLANE + 38 * G_PITCH + 32 < 65536, "ds_read immediate offsets");
}  // namespace

// ---------------------------------------------------------------------------------------------
// conv3x3_split_ring16_kernel (fifth generation): the ring structure on v_mfma_f32_16x16x32_f16.  The ring kernel keeps the matrix pipe busy 85 % of the
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
    r_xcd_tile(tile_m, tile_n, g.tile_panel);
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
                    if (pr == 0 && ABL != 5) load_b(S + H_AHEAD, nslot);            // the slot tap s - 1 released takes the filters of tap s + 2
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
    bool bad = false;
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
                if (g.out_split) {                      // uniform: the consumer is a SPLIT conv with no GroupNorm in front -- its operand planes leave from here
                    unsigned hi[4], lo[4];
                    split8_checked(v, hi, lo, bad);
                    half_t* P = reinterpret_cast<half_t*>(g.C) + 2 * moff[p4] - nn;         // pixel * 2 N + channel (ldc == N)
                    *reinterpret_cast<u32x4*>(P) = u32x4{hi[0], hi[1], hi[2], hi[3]};
                    *reinterpret_cast<u32x4*>(P + g.N) = u32x4{lo[0], lo[1], lo[2], lo[3]};
                } else {
                    const f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                    *reinterpret_cast<f32x4*>(Cb + moff[p4]) = o0;
                    *reinterpret_cast<f32x4*>(Cb + moff[p4] + 4) = o1;
                }
                if (g.gn_part_out_d) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { gs[e] += v[e]; gq[e] += v[e] * v[e]; }
                }
            }
        }
    }
    if (bad && g.range_flag) atomicOr(g.range_flag, 1);
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
    r_xcd_tile(tile_m, tile_v, g.tile_panel);
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
                    if (ABL != 5) load_b(S + 1, nslot);                               // the slot tap s - 1 released takes the filters of tap s + 1: a whole tap of MFMAs ahead
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
                        // (already landed: see above; the count only has to be loose enough not to wait for anything younger -- the
                        //  filters of the next tap, and at tap 2 the three pieces tap 1 fetched)
                        if (tap == 2) asm volatile("s_waitcnt vmcnt(7)" : "+v"(pst[u])); else asm volatile("s_waitcnt vmcnt(4)" : "+v"(pst[u]));
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

// can the conv's epilogue emit operand planes (GemmArgs::out_split)?  Only conv3x3_split_ring16_kernel does: g must take that kernel.
bool split_conv3_emits_planes(const GemmArgs& g) {
    static const bool off = getenv("HQT_SPLIT_PLANES_OUT") && atoi(getenv("HQT_SPLIT_PLANES_OUT")) == 0;        // A/B switch: 0 = fp32 tensor + operand pass
    return !off && split_stream_ok(g) && g.store == STORE_ROWS && !g.upsample && g.ldc == g.N && g.N % 8 == 0;
}
bool split_stream_ok(const GemmArgs& g) {
    if (!g.Bw_frag16) return false;
    if (g.H % R_TY != 0 || g.W % R_TX != 0 || g.Cin % 64 != 0) return false;      // an even number of 32-channel chunks (static ring slots)
    // one image's operand planes are addressed with 32-bit byte offsets below 2^31 (buffer loads; 2^31 marks the padding ring)
    if ((long long)(g.H >> g.upsample) * (g.W >> g.upsample) * g.Cin * 4 >= (1ll << 31)) return false;
    if (g.store == STORE_NCHW) return g.N <= 16;          // conv_out: one 16-channel block
    return g.N % 128 == 0 && g.ldc % 8 == 0;              // whole 128-channel tiles (the packed fragments hold ceil(N / 32) n-tiles)
}
// upsampling conv as four 2x2 phase convolutions on the low-resolution image (conv2x2_split_up16_kernel)
static bool split_up_shape(const GemmArgs& g) {
    static const bool off = getenv("HQT_SPLIT_UP") && atoi(getenv("HQT_SPLIT_UP")) == 0;              // A/B switch: 0 = nine taps on the upsampled image
    return !off && g.upsample == 1 && g.Bw_up16 && g.store == STORE_ROWS && (g.H / 2) % R_TY == 0 && (g.W / 2) % R_TX == 0 && g.N % 128 == 0;
}
int split_stream_tiles_per_image(const GemmArgs& g) { return split_up_shape(g) ? 4 * (g.H / 2 / R_TY) * (g.W / 2 / R_TX) : (g.H / R_TY) * (g.W / R_TX); }
hipError_t launch_split_conv3_stream(const GemmArgs& g, hipStream_t st) {
    if (g.store == STORE_NCHW) conv3x3_split_out16_kernel<<<dim3(1, g.M / (R_TY * R_TX), 1), 256, G_LDS, st>>>(g);
    else if (split_up_shape(g)) conv2x2_split_up16_kernel<0><<<dim3(4 * (g.N / 128), g.M / 4 / (R_TY * R_TX), 1), 256, G_LDS, st>>>(g);
    else conv3x3_split_ring16_kernel<0><<<dim3(g.N / 128, g.M / (R_TY * R_TX), 1), 256, G_LDS, st>>>(g);
    return hipGetLastError();
}
hipError_t split_stream_configure() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_split_ring16_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2x2_split_up16_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_split_out16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS);
}

// ---------------------------------------------------------------------------------------------
// norm_out -> swish -> conv_out (stage1/modules/layers.py:404-410; <= 4 output channels, NCHW fp32 + clamp) in ONE kernel, fp32 FMAs.
// conv_out is 0.26 % of the decoder's multiply-adds on the decoder's LARGEST tensor: on the matrix cores it needs the tensor as fp16
// hi / lo planes (an operand pass over 2 x 2.1 GB at batch 64) and pads 3 output channels to a 16-channel MFMA block.  Here the fp32
// tensor is read ONCE: a workgroup takes a 16 x 16 pixel tile, per 32-channel chunk it applies GroupNorm + swish while it fills the
// (16 + 2) x (16 + 2) patch in LDS (zero padding = zeros AFTER the normalisation, as nn.Conv2d pads), then every thread accumulates the
// three outputs of its pixel with packed fp32 FMAs -- filters from scalar registers (uniform addresses), patch rows on a 144-byte pitch
// (8 consecutive pixels start in 8 different 16-byte bank groups).  The loads of chunk c + 1 are in flight under the FMAs of chunk c.
// Exact fp32 products: no split, no range to check.  Bounds: 2.1 GB of HBM reads (0.42 ms at batch 64) / ~2.7 G swish evaluations.
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int O_T = 16, O_P = O_T + 2, O_CH = 32, O_PITCH = O_CH + 4;          // tile edge, patch edge, channels per chunk, floats per patch row
constexpr int O_LDS = O_P * O_P * O_PITCH * 4;                                  // 46656 B: three workgroups per CU
constexpr int O_V4 = O_P * O_P * (O_CH / 4);                                    // float4 slots of a chunk's patch (2592)
constexpr int O_PER = (O_V4 + 255) / 256;                                       // per thread (11; the last one partial)
typedef __attribute__((ext_vector_type(2))) float f32x2;
}  // namespace

template <int NO>
__global__ __launch_bounds__(256, 3) void conv_out_direct_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    float* patch = reinterpret_cast<float*>(lds_raw);
    const int tid = threadIdx.x;
    const int tiles_x = g.W / O_T, tiles = tiles_x * (g.H / O_T);
    const int img = blockIdx.x / tiles, trem = blockIdx.x - img * tiles;
    const int ty0 = (trem / tiles_x) * O_T, tx0 = (trem % tiles_x) * O_T;
    const int C = g.Cin, NC = C / O_CH;
    const float* __restrict__ X = reinterpret_cast<const float*>(g.A) + (long long)img * g.H * g.W * C;
    const float* __restrict__ Wt = reinterpret_cast<const float*>(g.Bw);        // [NO][9][C]
    // this thread's patch slots: slot = tid + 256 u -> pixel slot / 8, channel quad slot % 8 (the same quad for every u)
    const int v = tid & 7;
    int src[O_PER];                                     // element offset of the pixel inside the image (clamped: every load is unconditional)
    unsigned inside = 0;                                // bit u: slot u is a pixel of the image (else padding / past the patch: zeros)
#pragma unroll
    for (int u = 0; u < O_PER; ++u) {
        const int q = (tid + 256 * u) >> 3;
        const int qy = q / O_P, qx = q - qy * O_P;
        const int iy = ty0 + qy - 1, ix = tx0 + qx - 1;
        const bool in = (q < O_P * O_P) & ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
        src[u] = in ? (iy * g.W + ix) * C + v * 4 : v * 4;
        inside |= in ? 1u << u : 0u;
    }
    const int cpg = C / g.gn_groups;
    const float* stats = g.gn_stats + (long long)img * g.gn_groups * 2;
    f32x4 r[O_PER];
    auto fetch = [&](int c) {
#pragma unroll
        for (int u = 0; u < O_PER; ++u) r[u] = *reinterpret_cast<const f32x4*>(X + src[u] + c * O_CH);
    };
    auto fill = [&](int c) {                            // GroupNorm + swish of the fetched quads into the patch
        float a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = c * O_CH + v * 4 + j;
            const float mu = stats[(ch / cpg) * 2], rs = stats[(ch / cpg) * 2 + 1];
            a[j] = rs * g.gn_gamma[ch];
            b[j] = g.gn_beta[ch] - mu * a[j];
        }
#pragma unroll
        for (int u = 0; u < O_PER; ++u) {
            const int slot = tid + 256 * u;
            if (slot >= O_V4) continue;
            f32x4 y = {0.0f, 0.0f, 0.0f, 0.0f};
            if (inside >> u & 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = r[u][j] * a[j] + b[j];
                    if (g.gn_swish) t = t * __builtin_amdgcn_rcpf(1.0f + __expf(-t));
                    y[j] = t;
                }
            }
            *reinterpret_cast<f32x4*>(patch + (slot >> 3) * O_PITCH + v * 4) = y;
        }
    };
    const int py = tid >> 4, px = tid & 15;
    f32x2 acc[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) acc[o] = f32x2{0.0f, 0.0f};
    fetch(0);
    for (int c = 0; c < NC; ++c) {
        if (c > 0) __syncthreads();                     // the FMAs of chunk c - 1 have read the patch
        fill(c);
        __syncthreads();
        if (c + 1 < NC) fetch(c + 1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float* row = patch + ((py + tap / 3) * O_P + px + tap % 3) * O_PITCH;
            f32x4 x[O_CH / 4];
#pragma unroll
            for (int q = 0; q < O_CH / 4; ++q) x[q] = *reinterpret_cast<const f32x4*>(row + q * 4);
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                const float* w = Wt + ((long long)o * 9 + tap) * C + c * O_CH;      // uniform: scalar loads
#pragma unroll
                for (int q = 0; q < O_CH / 4; ++q) {
                    acc[o] += f32x2{x[q][0], x[q][1]} * f32x2{w[q * 4], w[q * 4 + 1]};
                    acc[o] += f32x2{x[q][2], x[q][3]} * f32x2{w[q * 4 + 2], w[q * 4 + 3]};
                }
            }
        }
    }
    float* Cb = reinterpret_cast<float*>(g.C);
    const long long hw = (long long)g.H * g.W, pix = (long long)(ty0 + py) * g.W + tx0 + px;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        float val = (acc[o][0] + acc[o][1]) * g.alpha + (g.bias ? g.bias[o] : 0.0f);
        if (g.clamp01) val = fminf(fmaxf(0.5f * val + 0.5f, 0.0f), 1.0f);
        Cb[((long long)img * NO + o) * hw + pix] = val;
    }
}

bool conv_out_direct_ok(const GemmArgs& g) {
    static const bool off = getenv("HQT_CONV_OUT_DIRECT") && atoi(getenv("HQT_CONV_OUT_DIRECT")) == 0;        // A/B switch: 0 = operand pass + conv3x3_split_out16_kernel
    return !off && g.conv_taps == 9 && g.store == STORE_NCHW && g.N >= 1 && g.N <= 4 && g.Cin % O_CH == 0 && g.H % O_T == 0 && g.W % O_T == 0 &&
           !g.upsample && !g.conv_stride2 && !g.resid && g.gn_stats && g.gn_groups > 0 && g.Cin % g.gn_groups == 0 &&
           (long long)g.H * g.W * g.Cin < (1ll << 31);
}
hipError_t launch_conv_out_direct(const GemmArgs& g, hipStream_t st) {
    const int grid = (g.M / (g.H * g.W)) * (g.H / O_T) * (g.W / O_T);
    switch (g.N) {
        case 1: conv_out_direct_kernel<1><<<grid, 256, O_LDS, st>>>(g); break;
        case 2: conv_out_direct_kernel<2><<<grid, 256, O_LDS, st>>>(g); break;
        case 3: conv_out_direct_kernel<3><<<grid, 256, O_LDS, st>>>(g); break;
        default: conv_out_direct_kernel<4><<<grid, 256, O_LDS, st>>>(g); break;
    }
    return hipGetLastError();
}
