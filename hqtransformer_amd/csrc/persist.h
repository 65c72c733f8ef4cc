// Persistent AR chain (FAST precision, up to 64 rows): one top position up to its top logits as ONE launch -- the body blocks, ln_f + sos_depth
// (hierarchical_ar.py:561,684-686), the single-key blocks of depth sub-step 0 and head_top (three code levels: the body blocks).
//
// Replaces, for a batch of up to 64 samples, the launch chain of hierarchical_ar.py:554-563 / layers.py:324-328,61-195
// (qkv -> attention -> proj -> fc1 -> fc2 per block; 5 dependent launches per block in run_block_dln) with one
// workgroup per CU that walks a PROGRAM of phases.  What the launch chain cannot do and this kernel does:
//   * every CU owns a fixed column slice of every nn.Linear, and its slice of ALL weights of the program is one
//     contiguous stream in HBM that a loader wave DMAs into an LDS ring (buffer_load ... lds), running ahead of
//     every dependency: the weights of phase p + 1 land while phase p computes and while the grid barrier waits;
//   * phases are separated by a one-hop grid barrier (8 x 8 sharded / replicated counters, write-through stores,
//     sc1 loads: MI355X_MICROARCH.md, valid forms) instead of a kernel boundary + a cold start of the weight stream;
//   * the fp32 master of the residual stream never leaves the CU that owns its columns (LDS), and the LayerNorm
//     statistics are recomputed by every consumer from the bf16 rows it loads anyway (a ones column on the matrix
//     cores for the sum, v_dot2c_f32_bf16 for the sum of squares) -- no partial-statistics arrays cross the chip;
//   * the wide-K GEMM (mlp.2, K = 4 D) is split over K across the four CUs of a "quad" (every CU streams 24 columns
//     x K / 4 and pulls a quarter of the activation panel through its L1 instead of all of it), the four fp32 partials
//     meet through a 4-arrival counter and three of the four CUs finish 8 columns each.
#pragma once
#include "common.h"
#include <vector>

enum { PP_QKV = 0, PP_KV1 = 1, PP_ATTN = 2, PP_RESID = 3, PP_GELU = 4, PP_ROWS = 5, PP_RESID_K4 = 6,
       PP_LNF = 7 };       // x <- LayerNorm(x) gamma + shift on the residual stream itself (ln_f + sos_depth between the body and the depth head, hierarchical_ar.py:561,684-686): N = K = D,
                           // PP_MAP_QUAD, A = the packed bf16 copy of x (row statistics), colsum = gamma, bias = beta + shift, out = the packed bf16 copy of the new x; no weights
enum { PP_MAP_EVEN = 0, PP_MAP_QUAD = 1, PP_MAP_K4 = 2 };

struct PersistPhase {       // one phase; the table lives in device memory and is the same for every CU
    int type;               // PP_*
    int N, K;               // GEMM phases: y[M, N] = A[M, K] W[N, K]^T
    int rot;                // PP_MAP_EVEN: rotation of the column-group -> CU map (persist_cols)
    int map;                // PP_MAP_*: PP_RESID takes PP_MAP_QUAD, PP_RESID_K4 takes PP_MAP_K4 (the same CUs own the same residual columns)
    int dln;                // 1: deferred LayerNorm on A (gamma-folded weights, colsum, folded bias: GemmArgs::ln_*)
    int act;                // ACT_* (PP_GELU)
    int cache_T;            // rows per sample of the key / value cache (PP_QKV, PP_KV1, PP_ATTN)
    int kv_row;             // PP_KV1: cache row of this step's key / value (0); PP_QKV / PP_ATTN take it from the step state
    int k4_epoch;           // PP_RESID_K4: 1-based count of such phases up to this one (the quad counters only grow inside a launch)
    const bf16_t* A;        // GEMM: A in the packed_off() layout (MB row blocks).  PP_ATTN: the query rows, bf16 [M][D]
    const float* bias;      // [N] or NULL
    const float* colsum;    // [N] (dln)
    void* out;              // PP_QKV: query rows bf16 [M][D]; PP_ATTN / PP_RESID* / PP_GELU: packed_off() bf16; PP_ROWS: fp32 [M][N]
    bf16_t* kc;             // this layer's key cache   [B][cache_T][D]
    bf16_t* vc;             // this layer's value cache
    bf16_t* vpk;            // PP_KV1: second copy of the value rows in the packed_off() layout (the attention output of a single key)
};

struct PersistArgs {
    const PersistPhase* phases;
    int n_phases;
    const char* wstream;                    // the program's weight streams, one contiguous run per CU
    const unsigned long long* cu_off;       // [ncu] byte offset of CU c's run
    unsigned* counters;                     // [64] barrier counters + [64] quad counters, one 128-byte line each (zeroed before every launch)
    unsigned* err;                          // err[0] != 0 after a launch that gave up on a barrier (results are garbage); err[1]: test hook, c + 1 = CU c never signals
                                            // its first phase, so every other CU runs into the time limit (hqt_set_switch(HQT_SWITCH_PERSIST_FAULT); 0 in production)
    float* x32;                             // fp32 master of the residual stream [M][D]: read at start, written back at the end
    float* slabs;                           // PP_RESID_K4 partials: [quad][3 groups][4 quarters][64 rows][8] fp32
    int D, M, MB, n_heads, head_dim;
    int t_base;                             // body: cache row of this step = t_base + *t_base_dev
    const int* t_base_dev;
    int write_back;
    int nt_weights;                         // non-temporal weight DMA
    int fill_s1, fill_s3;                   // loader budgets (1-KiB pieces) behind the barriers S1 / S3 of a phase (persist_default_fill)
    long long* stamps;                      // tools/micro only: [ncu][n_phases][8] wall-clock stamps of wave 0 (NULL in the product)
};

constexpr int PERSIST_RING_UNITS = 118;     // 1-KiB units of the LDS weight ring
constexpr int PERSIST_MAX_NC = 32;          // most columns of one phase on one CU
constexpr int PERSIST_COUNTER_BYTES = 128 * 128;
__host__ __device__ inline size_t persist_slab_floats(int ncu) { return (size_t)(ncu / 4) * 3 * 4 * 64 * 8; }

// Column groups (8 columns) of a phase owned by CU `cu`: groups [g0, g0 + ng), and for PP_MAP_K4 the K quarter kq.
//   PP_MAP_EVEN  G >= ncu: every CU takes G / ncu, the first G % ncu rotated CUs one more; G < ncu: rotated CU c < G takes group c
//   PP_MAP_QUAD  group g belongs to CU 4 (g / 3) + g % 3 (the CU that finishes it in a PP_RESID_K4 phase)
//   PP_MAP_K4    CU c = 4 q + kq computes groups [3 q, 3 q + 3) over K quarter kq
__host__ __device__ inline void persist_cols(int N, int rot, int map, int cu, int ncu, int* g0, int* ng, int* kq) {
    const int G = N >> 3;
    *kq = 0;
    if (map == PP_MAP_QUAD) {
        const int g = 3 * (cu >> 2) + (cu & 3);
        const bool own = (cu & 3) < 3 && g < G;
        *ng = own ? 1 : 0;
        *g0 = own ? g : 0;
        return;
    }
    if (map == PP_MAP_K4) {
        const int q = cu >> 2;
        const int lo = 3 * q, hi = (3 * q + 3 < G) ? 3 * q + 3 : G;
        *ng = hi > lo ? hi - lo : 0;
        *g0 = hi > lo ? lo : 0;
        *kq = cu & 3;
        return;
    }
    const int c = (cu + ncu - rot % ncu) % ncu;
    if (G >= ncu) {
        const int base = G / ncu, extra = G % ncu;
        *ng = base + (c < extra ? 1 : 0);
        *g0 = c * base + (c < extra ? c : extra);
    } else {
        *ng = c < G ? 1 : 0;
        *g0 = c < G ? c : 0;
    }
}
__host__ __device__ inline bool persist_is_gemm(int type) { return type != PP_ATTN; }
__host__ __device__ inline int persist_tile_units(const PersistPhase& ph, int cu, int ncu) {    // KiB of CU cu's weight tile of this phase
    if (!persist_is_gemm(ph.type) || ph.type == PP_LNF) return 0;
    int g0, ng, kq;
    persist_cols(ph.N, ph.rot, ph.map, cu, ncu, &g0, &ng, &kq);
    return ng * ((ph.map == PP_MAP_K4 ? ph.K >> 2 : ph.K) >> 6);
}

// shapes one program can run (N % 8, K % 64 -- K % 256 for the K-split phase --, at most PERSIST_MAX_NC columns and PERSIST_RING_UNITS
// KiB of weights per CU and phase, residual width D / 8 <= 3 ncu / 4, attention staging within its LDS region)
bool persist_program_ok(const std::vector<PersistPhase>& phases, int D, int M, int n_heads, int ncu);
// byte offsets of the program's weight tiles: tile_off[p * ncu + c] (absolute, from the stream base; phases without weights repeat the
// running offset), cu_off[c]; returns the stream size
size_t persist_layout(const std::vector<PersistPhase>& phases, int ncu, std::vector<unsigned long long>& cu_off, std::vector<unsigned long long>& tile_off);
// packs W (fp32 [N][K], optionally scaled per column k by gamma) into the tiles of one phase: d_tile_off = device copy of tile_off[p * ncu ..]
hipError_t launch_persist_pack(const float* w, const float* gamma, const PersistPhase& ph, int ncu, char* stream, const unsigned long long* d_tile_off, hipStream_t st);
inline void persist_default_fill(PersistArgs& a) { a.fill_s1 = 0; a.fill_s3 = 48; }
hipError_t persist_configure();
// persistent workgroups (576 threads, the whole LDS ring) one compute unit admits by the occupancy query: the grid barrier needs >= 1
int persist_blocks_per_cu();
// memset of the counters + the launch (both stream-ordered, capturable)
hipError_t launch_persist(const PersistArgs& a, int ncu, hipStream_t st);
