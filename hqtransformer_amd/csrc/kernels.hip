// Small kernels of the HQ-Transformer sampling path: embeddings, LayerNorm, KV-cache attention, the
// fused sampler, codebook gather, GroupNorm statistics.  gfx950 only (wave = 64).
#include "kernels.h"
#include <mutex>
#include "gemm_generic.h"
#include <type_traits>

// ---------------------------------------------------------------------------------------------
// reductions (256-thread workgroups = 4 waves), deterministic for a fixed launch shape
// ---------------------------------------------------------------------------------------------
template <typename T, typename Op>
__device__ __forceinline__ T wave_reduce(T v, Op op) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = op(v, __shfl_xor(v, off, 64));
    return v;
}
template <typename T, typename Op>
__device__ __forceinline__ T block_reduce(T v, Op op, T* scratch /* >= blockDim.x / 64 */) {
    v = wave_reduce(v, op);
    const int wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[wid] = v;
    __syncthreads();
    T r = scratch[0];
    for (int i = 1; i < nw; ++i) r = op(r, scratch[i]);
    return r;
}
struct OpAdd { template <typename T> __device__ T operator()(T a, T b) const { return a + b; } };
struct OpMax { template <typename T> __device__ T operator()(T a, T b) const { return a > b ? a : b; } };
struct OpMin { template <typename T> __device__ T operator()(T a, T b) const { return a < b ? a : b; } };

// Writes the bf16 copy of one fp32 row into the packed_off() layout and its (sum, sumsq) -- computed on the rounded
// values, which is what the MFMA will multiply -- for the deferred-LayerNorm GEMMs.  Called by all 256 threads.
__device__ __forceinline__ void emit_packed_row(const float* x, int m, int D, bf16_t* xpk, int pk_mb, float* parts, float* red) {
    float s = 0.0f, q = 0.0f;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        const bf16_t hb = f32_to_bf16(x[d]);
        xpk[packed_off(m, d, pk_mb)] = hb;
        const float r = bf16_to_f32(hb);
        s += r; q += r * r;
    }
    s = block_reduce(s, OpAdd(), red);
    q = block_reduce(q, OpAdd(), red);
    if (threadIdx.x == 0) { parts[2 * m] = s; parts[2 * m + 1] = q; }
}

// ---------------------------------------------------------------------------------------------
// step state
// ---------------------------------------------------------------------------------------------
__global__ void advance_step_kernel(StepState* s, int d_tbase) { s->step += 1; s->t_base += d_tbase; }
__global__ void set_step_kernel(StepState* s, int step, int t_base) { s->step = step; s->t_base = t_base; }
__global__ void set_rows_kernel(RowKey* rows, int B, unsigned long long seed, long long sample_offset) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) { rows[b].seed = seed; rows[b].global_row = sample_offset + b; }
}
hipError_t launch_set_rows(RowKey* rows, int B, uint64_t seed, int64_t sample_offset, hipStream_t st) {
    set_rows_kernel<<<(B + 255) / 256, 256, 0, st>>>(rows, B, (unsigned long long)seed, (long long)sample_offset);
    return hipGetLastError();
}
hipError_t launch_advance_step(StepState* s, int d_tbase, hipStream_t st) {
    advance_step_kernel<<<1, 1, 0, st>>>(s, d_tbase);
    return hipGetLastError();
}
hipError_t launch_set_step(StepState* s, int step, int t_base, hipStream_t st) {
    set_step_kernel<<<1, 1, 0, st>>>(s, step, t_base);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// A3/K1: input embedding of one top position (hierarchical_ar.py:493-544)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_step_kernel(EmbedArgs a) {
    __shared__ float red[4];
    const int b = blockIdx.x, D = a.D;
    const int step = a.state->step;
    float* x = a.x + (long long)b * D;
    if (step == 0) {
        const float* src = a.cond_type == 1 ? a.sos + clamp_idx(a.cond[b], a.n_classes) * (long long)D : a.sos;
        for (int d = threadIdx.x; d < D; d += blockDim.x) x[d] = src[d];
        if (a.xpk) { __syncthreads(); emit_packed_row(x, b, D, a.xpk, a.pk_mb, a.parts, red); }
        return;
    }
    const int p = step - 1;
    const long long ct = clamp_idx(a.codes_top[(long long)b * a.n_steps + p], a.V);
    long long cb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) cb[s] = clamp_idx(a.codes_bot[((long long)b * a.n_steps + p) * 4 + s], a.V);
    if (a.embedding == 1) {                      // 'reduce': channel k*4+slot of the bottom part (:522-526)
        const int Dq = D / 4;
        for (int d = threadIdx.x; d < D; d += blockDim.x) {
            const float top = a.tok_top[ct * D + d] + a.pos_top[(long long)p * D + d];
            x[d] = top + a.tok_bot[cb[d & 3] * Dq + (d >> 2)];
        }
    } else if (a.levels == 3) {                  // three levels: mean over the 21 tokens (hqtransformer.py:466-488)
        long long c2[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) c2[k] = clamp_idx(a.codes_l2[((long long)b * a.n_steps + p) * 16 + k], a.V);
        for (int d = threadIdx.x; d < D; d += blockDim.x) {
            float s = (a.tok_top[ct * D + d] + a.pos_top[(long long)p * D + d]) + a.pos_emb[d];
#pragma unroll
            for (int k = 0; k < 4; ++k) s += a.tok_bot[cb[k] * D + d] + a.pos_emb[(1 + k) * D + d];
#pragma unroll
            for (int k = 0; k < 16; ++k) s += a.tok_l2[c2[k] * D + d] + a.pos_emb[(5 + k) * D + d];
            x[d] = s / 21.0f;
        }
    } else {                                     // 'transformer1': mean over the 5 tokens (:535-544)
        for (int d = threadIdx.x; d < D; d += blockDim.x) {
            float s = (a.tok_top[ct * D + d] + a.pos_top[(long long)p * D + d]) + a.pos_emb[d];
#pragma unroll
            for (int k = 0; k < 4; ++k) s += a.tok_bot[cb[k] * D + d] + a.pos_emb[(1 + k) * D + d];
            x[d] = s / 5.0f;
        }
    }
    if (a.xpk) { __syncthreads(); emit_packed_row(x, b, D, a.xpk, a.pk_mb, a.parts, red); }
}
hipError_t launch_embed_step(const EmbedArgs& a, hipStream_t st) {
    embed_step_kernel<<<a.B, 256, 0, st>>>(a);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void embed_text_kernel(const int64_t* cond, const float* tok, const float* pos,
                                                         float* x, int T, int D, int vocab) {
    const int row = blockIdx.x, t = row % T;
    const long long id = clamp_idx(cond[row], vocab);
    for (int d = threadIdx.x; d < D; d += blockDim.x) x[(long long)row * D + d] = tok[id * D + d] + pos[(long long)t * D + d];
}
hipError_t launch_embed_text(const int64_t* cond, const float* tok, const float* pos, float* x, int B, int T, int D,
                             hipStream_t st, int vocab) {
    embed_text_kernel<<<B * T, 256, 0, st>>>(cond, tok, pos, x, T, D, vocab);
    return hipGetLastError();
}

// `tok_ld` = row length of the table: D, or 4 D for the 'reduce' head, where slot s of a code takes the D-slice s of its row
// (hqtransformer.py:108-116,532-533)
__global__ __launch_bounds__(256) void depth_embed_kernel(const int64_t* codes_top, int n_steps, const StepState* state,
                                                          const float* tok, const float* pos, float* x, int D, bf16_t* xpk,
                                                          int pk_mb, float* parts, int V, int tok_ld) {
    __shared__ float red[4];
    const int row = blockIdx.x, b = row >> 2, s = row & 3;
    const long long code = clamp_idx(codes_top[(long long)b * n_steps + state->step], V);
    const float* e = tok + code * tok_ld + (tok_ld > D ? s * D : 0);
    for (int d = threadIdx.x; d < D; d += blockDim.x) x[(long long)row * D + d] = e[d] + pos[(long long)s * D + d];
    if (xpk) { __syncthreads(); emit_packed_row(x + (long long)row * D, row, D, xpk, pk_mb, parts, red); }
}
hipError_t launch_depth_embed(const int64_t* codes_top, int n_steps, const StepState* state, const float* tok,
                              const float* pos, float* x, int B, int D, bf16_t* xpk, int pk_mb, float* parts, hipStream_t st, int V, int tok_ld) {
    depth_embed_kernel<<<B * 4, 256, 0, st>>>(codes_top, n_steps, state, tok, pos, x, D, xpk, pk_mb, parts, V, tok_ld > 0 ? tok_ld : D);
    return hipGetLastError();
}

// Level-2 tokens of the three-level head, (H1 H2 W1 W2) raster: token i carries its parent's level-1 embedding (the D-slice of child
// (H2 W2) when the table is [V, 4 D]: 'reduce'), position i and -- `tok0` non-NULL: 'add' -- the top code's embedding (hqtransformer.py:537-551)
__global__ __launch_bounds__(256) void depth_embed_l2_kernel(const int64_t* codes0, const int64_t* codes1, int n_steps, const StepState* state,
                                                             const float* tok0, const float* tok1, const float* pos, float* x, int D,
                                                             bf16_t* xpk, int pk_mb, float* parts, int V, int tok1_ld) {
    __shared__ float red[4];
    const int row = blockIdx.x, b = row >> 4, i = row & 15;
    const int parent = (i >> 3) * 2 + ((i & 3) >> 1);                  // (H1 H2 W1 W2) raster -> (H1 W1)
    const int child = ((i >> 2) & 1) * 2 + (i & 1);                    //                      -> (H2 W2)
    const long long c0 = clamp_idx(codes0[(long long)b * n_steps + state->step], V);
    const long long c1 = clamp_idx(codes1[((long long)b * n_steps + state->step) * 4 + parent], V);
    const float* e1 = tok1 + c1 * tok1_ld + (tok1_ld > D ? child * D : 0);
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float v = e1[d] + pos[(long long)i * D + d];
        if (tok0) v += tok0[c0 * D + d];
        x[(long long)row * D + d] = v;
    }
    if (xpk) { __syncthreads(); emit_packed_row(x + (long long)row * D, row, D, xpk, pk_mb, parts, red); }
}
hipError_t launch_depth_embed_l2(const int64_t* codes0, const int64_t* codes1, int n_steps, const StepState* state, const float* tok0,
                                 const float* tok1, const float* pos, float* x, int B, int D, bf16_t* xpk, int pk_mb, float* parts,
                                 hipStream_t st, int V, int tok1_ld) {
    depth_embed_l2_kernel<<<B * 16, 256, 0, st>>>(codes0, codes1, n_steps, state, tok0, tok1, pos, x, D, xpk, pk_mb, parts, V, tok1_ld > 0 ? tok1_ld : D);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void depth_embed_causal_kernel(const int64_t* codes, int stride, int slot, int n_steps, const StepState* state,
                                                                 const float* tok, const float* pos_row, float* x, int D, bf16_t* xpk,
                                                                 int pk_mb, float* parts, int V) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const long long code = clamp_idx(codes[((long long)b * n_steps + state->step) * stride + slot], V);
    for (int d = threadIdx.x; d < D; d += blockDim.x) x[(long long)b * D + d] = tok[code * D + d] + pos_row[d];
    if (xpk) { __syncthreads(); emit_packed_row(x + (long long)b * D, b, D, xpk, pk_mb, parts, red); }
}
hipError_t launch_depth_embed_causal(const int64_t* codes, int stride, int slot, int n_steps, const StepState* state, const float* tok,
                                     const float* pos_row, float* x, int B, int D, bf16_t* xpk, int pk_mb, float* parts, hipStream_t st, int V) {
    depth_embed_causal_kernel<<<B, 256, 0, st>>>(codes, stride, slot, n_steps, state, tok, pos_row, x, D, xpk, pk_mb, parts, V);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// LayerNorm (eps 1e-5), two-pass, one workgroup per row
// ---------------------------------------------------------------------------------------------
// One wave per row (4 rows per workgroup): the row lives in registers as float4 vectors (coalesced 1-KiB
// wave loads), mean and variance are two shuffle reductions, no LDS and no barrier.
template <typename TO>
__global__ __launch_bounds__(256) void layernorm_kernel(LNArgs a) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= a.M) return;
    const int D = a.D;
    const long long in_row = (long long)m * a.in_rows_per_group + a.in_row_offset;      // ln_f of the prefill reads the last token of each sample
    float* x = a.x + in_row * D;
    constexpr int MAXV = 8;                       // D <= 2048 in registers; wider rows fall back to re-reading
    float4 v[MAXV], gmv[MAXV], btv[MAXV];      // affine parameters are fetched with the row, ahead of the reductions
    float s = 0.0f;
    const int nvec = D >> 2;                      // D % 4 == 0
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = lane + i * 64;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        gmv[i] = v[i]; btv[i] = v[i];
        if (vi < nvec) {
            gmv[i] = *reinterpret_cast<const float4*>(a.gamma + vi * 4);
            btv[i] = *reinterpret_cast<const float4*>(a.beta + vi * 4);
            if (a.add) { const float4 ad = *reinterpret_cast<const float4*>(a.add + vi * 4); btv[i].x += ad.x; btv[i].y += ad.y; btv[i].z += ad.z; btv[i].w += ad.w; }
            float4 t = *reinterpret_cast<const float4*>(x + vi * 4);
            if (a.n_slabs > 0) {                  // fold in the split-K partial sums (+bias) of the previous GEMM
                if (a.slab_bias) { const float4 bb = *reinterpret_cast<const float4*>(a.slab_bias + vi * 4); t.x += bb.x; t.y += bb.y; t.z += bb.z; t.w += bb.w; }
                for (int sl = 0; sl < a.n_slabs; ++sl) {
                    const float4 p = *reinterpret_cast<const float4*>(a.slabs + ((long long)sl * a.slab_rows + in_row) * D + vi * 4);   // slabs are indexed like x
                    t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
                }
                *reinterpret_cast<float4*>(x + vi * 4) = t;
            }
            v[i] = t;
            s += (t.x + t.y) + (t.z + t.w);
        }
    }
    for (int vi = lane + MAXV * 64; vi < nvec; vi += 64) { const float4 t = *reinterpret_cast<const float4*>(x + vi * 4); s += (t.x + t.y) + (t.z + t.w); }
    const float mean = wave_reduce(s, OpAdd()) / (float)D;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (lane + i * 64 < nvec) {
            const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
    for (int vi = lane + MAXV * 64; vi < nvec; vi += 64) {
        const float4 t = *reinterpret_cast<const float4*>(x + vi * 4);
        const float dx = t.x - mean, dy = t.y - mean, dz = t.z - mean, dw = t.w - mean;
        q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    const float rstd = 1.0f / sqrtf(wave_reduce(q, OpAdd()) / (float)D + a.eps);
    float ys = 0.0f, yq = 0.0f;
    auto emit = [&](int vi, float4 t, float4 gm, float4 bt) {
        const int d = vi * 4;
        float o[4] = {(t.x - mean) * rstd * gm.x + bt.x, (t.y - mean) * rstd * gm.y + bt.y, (t.z - mean) * rstd * gm.z + bt.z,
                      (t.w - mean) * rstd * gm.w + bt.w};
        TO* dst = a.out_packed_mb ? reinterpret_cast<TO*>(a.y) + packed_off(m, d, a.out_packed_mb)      // 4 consecutive k stay contiguous
                                  : reinterpret_cast<TO*>(a.y) + (long long)m * D + d;
        if (sizeof(TO) == 2) {
            uint2 pk;
            pk.x = (unsigned)f32_to_bf16(o[0]) | ((unsigned)f32_to_bf16(o[1]) << 16);
            pk.y = (unsigned)f32_to_bf16(o[2]) | ((unsigned)f32_to_bf16(o[3]) << 16);
            *reinterpret_cast<uint2*>(dst) = pk;
        } else {
            *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        }
        if (a.ypk) {                              // deferred-LN consumers read this row as bf16 + its statistics
            uint2 pk;
            pk.x = (unsigned)f32_to_bf16(o[0]) | ((unsigned)f32_to_bf16(o[1]) << 16);
            pk.y = (unsigned)f32_to_bf16(o[2]) | ((unsigned)f32_to_bf16(o[3]) << 16);
            *reinterpret_cast<uint2*>(a.ypk + packed_off(m, d, a.ypk_mb)) = pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float r = bf16_to_f32(f32_to_bf16(o[e])); ys += r; yq += r * r; }
        }
    };
#pragma unroll
    for (int i = 0; i < MAXV; ++i) if (lane + i * 64 < nvec) emit(lane + i * 64, v[i], gmv[i], btv[i]);
    for (int vi = lane + MAXV * 64; vi < nvec; vi += 64) {
        float4 bt = *reinterpret_cast<const float4*>(a.beta + vi * 4);
        if (a.add) { const float4 ad = *reinterpret_cast<const float4*>(a.add + vi * 4); bt.x += ad.x; bt.y += ad.y; bt.z += ad.z; bt.w += ad.w; }
        emit(vi, *reinterpret_cast<const float4*>(x + vi * 4), *reinterpret_cast<const float4*>(a.gamma + vi * 4), bt);
    }
    if (a.ypk) {
        ys = wave_reduce(ys, OpAdd());
        yq = wave_reduce(yq, OpAdd());
        if (lane == 0) { a.yparts[2 * m] = ys; a.yparts[2 * m + 1] = yq; }
    }
}
hipError_t launch_layernorm(const LNArgs& a, hipStream_t st) {
    if (a.D % 4 != 0) return hipErrorInvalidValue;
    if (a.out_dtype == DT_BF16) layernorm_kernel<bf16_t><<<(a.M + 3) / 4, 256, 0, st>>>(a);
    else layernorm_kernel<float><<<(a.M + 3) / 4, 256, 0, st>>>(a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// A4/K3: KV-cache attention, one wave per (sample, head, query) -- stage2/layers.py:93-102,183-187.
// The scale 1/sqrt(hs) is applied to K before the product, as the reference does (:102).
// ---------------------------------------------------------------------------------------------
// One wave per (sample, head, query), four waves per workgroup.  A key/value row of one head (hs elements)
// is read by hs/8 adjacent lanes, 8 elements (one 16-B vector in bf16) each, so every wave load covers
// 64 / (hs/8) whole rows: QK^T is 8 FMAs per lane plus a log2(hs/8)-step shuffle reduction, PV keeps 8
// accumulators per lane and reduces over the row slots at the end.
template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void ld8<bf16_t>(const bf16_t* p, float (&v)[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = bf16_to_f32((bf16_t)(w[i] & 0xffffu)); v[2 * i + 1] = bf16_to_f32((bf16_t)(w[i] >> 16)); }
}

// One wave per (sample, head, query).  hs / 8 adjacent lanes ("chunks") cover one key / value row with 16-byte vectors,
// so a pass handles 64 / chunks rows and PB passes are fetched together (one memory round trip for K AND V of up to
// 64 keys at hs = 64).  Scores, probabilities and the output accumulator stay in registers: a group's score is
// xor-reduced over its chunk lanes (every lane of the group ends up with it), the running maximum / sum are reduced
// across the row slots, and further key groups are folded in with the online-softmax rescaling.  No LDS, no barriers.
template <typename T>
__global__ __launch_bounds__(256) void attention_kernel(AttnArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gid = blockIdx.x * 4 + wave;
    if (gid >= a.B * a.n_heads * a.Tq) return;
    long long stamp[5];
    if (a.dbg) stamp[0] = clock64();
    const int qi = gid % a.Tq;
    const int h = (gid / a.Tq) % a.n_heads;
    const int b = gid / (a.Tq * a.n_heads);
    const int hs = a.head_dim, D = a.n_heads * hs;
    const int chunks = hs >> 3;                  // lanes per row (power of two: hs in {8,16,32,64,128,256})
    const int rows_per_pass = 64 / chunks;
    const int c = lane % chunks, slot = lane / chunks;
    const T* q = reinterpret_cast<const T*>(a.q) + ((long long)(b * a.Tq + qi)) * D + h * hs + c * 8;
    const T* kc = reinterpret_cast<const T*>(a.kcache) + (long long)b * a.Tmax * D + h * hs + c * 8;
    const T* vc = reinterpret_cast<const T*>(a.vcache) + (long long)b * a.Tmax * D + h * hs + c * 8;
    float qv[8];
    ld8<T>(q, qv);                               // independent of the step state: in flight while t_base arrives
    const int tb = a.t_base + (a.t_base_dev ? *a.t_base_dev : 0);
    const int nkeys = a.causal ? tb + qi + 1 : tb + a.Tq;
    const float scale = 1.0f / sqrtf((float)hs);
    typedef typename std::conditional<sizeof(T) == 2, uint4, float4>::type raw_t;     // 16-B vector of the cache dtype
    constexpr int NRAW = sizeof(T) == 2 ? 1 : 2, PB = 8;                               // passes whose loads are issued together
    auto unpack = [](const raw_t* r, float (&f)[8]) {
        if (sizeof(T) == 2) {
            const uint4 t = *reinterpret_cast<const uint4*>(r);
            const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) { f[2 * i] = bf16_to_f32((bf16_t)(w[i] & 0xffffu)); f[2 * i + 1] = bf16_to_f32((bf16_t)(w[i] >> 16)); }
        } else {
            const float4 a0 = *reinterpret_cast<const float4*>(r), a1 = *reinterpret_cast<const float4*>(r + 1);
            f[0] = a0.x; f[1] = a0.y; f[2] = a0.z; f[3] = a0.w; f[4] = a1.x; f[5] = a1.y; f[6] = a1.z; f[7] = a1.w;
        }
    };
    float run_max = -INFINITY, run_sum = 0.0f;   // identical in every lane after each group
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
    // A group is up to PB passes fetched in one round trip; the last (often the only) group runs exactly the passes it has keys for --
    // `np` is uniform over the launch, so the switch below picks a fully unrolled body.  Skipped passes contributed exp(-inf) = 0 and
    // max(-inf) before: the results are bit-identical, and a step with few cached keys no longer pays the arithmetic of 64.
    auto group = [&](int j0, auto np_tag) {
        constexpr int NP = decltype(np_tag)::value;
        raw_t kbuf[NP][NRAW], vbuf[NP][NRAW];
#pragma unroll
        for (int p = 0; p < NP; ++p) {                                  // unconditional (clamped) loads: all in flight at once
            const long long j = min(j0 + p * rows_per_pass + slot, nkeys - 1);
            const raw_t* ks = reinterpret_cast<const raw_t*>(kc + j * D);
            const raw_t* vs = reinterpret_cast<const raw_t*>(vc + j * D);
#pragma unroll
            for (int e = 0; e < NRAW; ++e) { kbuf[p][e] = ks[e]; vbuf[p][e] = vs[e]; }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (a.dbg && j0 == 0) { stamp[1] = clock64(); __builtin_amdgcn_sched_barrier(0); }
        float sc[NP];
        float gmax = -INFINITY;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float kv[8];
            unpack(kbuf[p], kv);
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s = fmaf(qv[i], kv[i] * scale, s);            // scale on K, as layers.py:102
            for (int off = chunks >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            sc[p] = (j0 + p * rows_per_pass + slot < nkeys) ? s : -INFINITY;
            gmax = fmaxf(gmax, sc[p]);
        }
        for (int off = chunks; off < 64; off <<= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, off, 64));     // across the row slots
        if (a.dbg && j0 == 0) stamp[2] = clock64();
        const float new_max = fmaxf(run_max, gmax);                     // finite: key j0 is always valid
        const float rescale = expf(run_max - new_max);                  // 0 for the first group (run_max = -inf)
        float gsum = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] *= rescale;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const float e = expf(sc[p] - new_max);                      // 0 for masked rows
            gsum += e;
            float vv[8];
            unpack(vbuf[p], vv);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fmaf(e, vv[i], acc[i]);
        }
        for (int off = chunks; off < 64; off <<= 1) gsum += __shfl_xor(gsum, off, 64);
        run_sum = run_sum * rescale + gsum;
        run_max = new_max;
        if (a.dbg && j0 == 0) stamp[3] = clock64();
    };
    const int npass = (nkeys + rows_per_pass - 1) / rows_per_pass;
    for (int p0 = 0; p0 < npass; p0 += PB) {
        const int j0 = p0 * rows_per_pass;
        switch (min(PB, npass - p0)) {
        case 1: group(j0, std::integral_constant<int, 1>{}); break;
        case 2: group(j0, std::integral_constant<int, 2>{}); break;
        case 3: group(j0, std::integral_constant<int, 3>{}); break;
        case 4: group(j0, std::integral_constant<int, 4>{}); break;
        case 5: group(j0, std::integral_constant<int, 5>{}); break;
        case 6: group(j0, std::integral_constant<int, 6>{}); break;
        case 7: group(j0, std::integral_constant<int, 7>{}); break;
        default: group(j0, std::integral_constant<int, 8>{}); break;
        }
    }
    for (int off = chunks; off < 64; off <<= 1)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += __shfl_xor(acc[i], off, 64);
    const float inv = 1.0f / run_sum;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] *= inv;
    if (slot == 0) {
        const int row = b * a.Tq + qi, col = h * hs + c * 8;
        T* o = a.out_packed_mb ? reinterpret_cast<T*>(a.out) + packed_off(row, col, a.out_packed_mb)
                               : reinterpret_cast<T*>(a.out) + (long long)row * D + col;
        if (sizeof(T) == 2) {
            uint4 pk;
            pk.x = (unsigned)f32_to_bf16(acc[0]) | ((unsigned)f32_to_bf16(acc[1]) << 16);
            pk.y = (unsigned)f32_to_bf16(acc[2]) | ((unsigned)f32_to_bf16(acc[3]) << 16);
            pk.z = (unsigned)f32_to_bf16(acc[4]) | ((unsigned)f32_to_bf16(acc[5]) << 16);
            pk.w = (unsigned)f32_to_bf16(acc[6]) | ((unsigned)f32_to_bf16(acc[7]) << 16);
            *reinterpret_cast<uint4*>(o) = pk;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) st1<T>(o + i, acc[i]);
        }
    }
    if (a.dbg && lane == 0) {
        stamp[4] = clock64();
        long long* d = a.dbg + (long long)gid * 8;
        d[0] = wall_clock64(); d[1] = stamp[1] - stamp[0]; d[2] = stamp[2] - stamp[0]; d[3] = stamp[3] - stamp[0]; d[4] = stamp[4] - stamp[0];
    }
}
// Few queries over a short shared cache (depth sub-step 1: 4 queries x <= 5 keys; sub-steps of the three-level head): one wave per
// (sample, head) fetches the K / V rows ONCE and loops over the queries -- a quarter of the waves of the kernel above, whose cost at
// these sizes is the wave count (49 k waves at 512 samples: 50 us per launch for a few kilobytes of arithmetic).  Per query the same
// operations in the same order as attention_kernel with one group: bit-identical results.
template <typename T, int NP>
__global__ __launch_bounds__(256) void attention_fewq_kernel(AttnArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gid = blockIdx.x * 4 + wave;
    if (gid >= a.B * a.n_heads) return;
    const int h = gid % a.n_heads, b = gid / a.n_heads;
    const int hs = a.head_dim, D = a.n_heads * hs;
    const int chunks = hs >> 3, rows_per_pass = 64 / chunks;
    const int c = lane % chunks, slot = lane / chunks;
    const T* kc = reinterpret_cast<const T*>(a.kcache) + (long long)b * a.Tmax * D + h * hs + c * 8;
    const T* vc = reinterpret_cast<const T*>(a.vcache) + (long long)b * a.Tmax * D + h * hs + c * 8;
    const int tb = a.t_base + (a.t_base_dev ? *a.t_base_dev : 0);
    const int nall = tb + a.Tq;                                       // keys any query of this call can see
    const float scale = 1.0f / sqrtf((float)hs);
    typedef typename std::conditional<sizeof(T) == 2, uint4, float4>::type raw_t;
    constexpr int NRAW = sizeof(T) == 2 ? 1 : 2;
    raw_t kbuf[NP][NRAW], vbuf[NP][NRAW];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const long long j = min(p * rows_per_pass + slot, nall - 1);
        const raw_t* ks = reinterpret_cast<const raw_t*>(kc + j * D);
        const raw_t* vs = reinterpret_cast<const raw_t*>(vc + j * D);
#pragma unroll
        for (int e = 0; e < NRAW; ++e) { kbuf[p][e] = ks[e]; vbuf[p][e] = vs[e]; }
    }
    auto unpack = [](const raw_t* r, float (&f)[8]) {
        if (sizeof(T) == 2) {
            const uint4 t = *reinterpret_cast<const uint4*>(r);
            const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) { f[2 * i] = bf16_to_f32((bf16_t)(w[i] & 0xffffu)); f[2 * i + 1] = bf16_to_f32((bf16_t)(w[i] >> 16)); }
        } else {
            const float4 a0 = *reinterpret_cast<const float4*>(r), a1 = *reinterpret_cast<const float4*>(r + 1);
            f[0] = a0.x; f[1] = a0.y; f[2] = a0.z; f[3] = a0.w; f[4] = a1.x; f[5] = a1.y; f[6] = a1.z; f[7] = a1.w;
        }
    };
    for (int qi = 0; qi < a.Tq; ++qi) {
        const int nkeys = a.causal ? tb + qi + 1 : nall;
        float qv[8];
        ld8<T>(reinterpret_cast<const T*>(a.q) + ((long long)(b * a.Tq + qi)) * D + h * hs + c * 8, qv);
        float sc[NP], gmax = -INFINITY;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float kv[8];
            unpack(kbuf[p], kv);
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s = fmaf(qv[i], kv[i] * scale, s);            // scale on K, as layers.py:102
            for (int off = chunks >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            sc[p] = (p * rows_per_pass + slot < nkeys) ? s : -INFINITY;
            gmax = fmaxf(gmax, sc[p]);
        }
        for (int off = chunks; off < 64; off <<= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, off, 64));
        const float new_max = fmaxf(-INFINITY, gmax);
        const float rescale = expf(-INFINITY - new_max);                // 0: one group, as the first group of attention_kernel
        float gsum = 0.0f, acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.0f * rescale;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const float e = expf(sc[p] - new_max);
            gsum += e;
            float vv[8];
            unpack(vbuf[p], vv);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fmaf(e, vv[i], acc[i]);
        }
        for (int off = chunks; off < 64; off <<= 1) gsum += __shfl_xor(gsum, off, 64);
        const float run_sum = 0.0f * rescale + gsum;
        for (int off = chunks; off < 64; off <<= 1)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += __shfl_xor(acc[i], off, 64);
        const float inv = 1.0f / run_sum;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] *= inv;
        if (slot == 0) {
            const int row = b * a.Tq + qi, col = h * hs + c * 8;
            T* o = a.out_packed_mb ? reinterpret_cast<T*>(a.out) + packed_off(row, col, a.out_packed_mb)
                                   : reinterpret_cast<T*>(a.out) + (long long)row * D + col;
            if (sizeof(T) == 2) {
                uint4 pk;
                pk.x = (unsigned)f32_to_bf16(acc[0]) | ((unsigned)f32_to_bf16(acc[1]) << 16);
                pk.y = (unsigned)f32_to_bf16(acc[2]) | ((unsigned)f32_to_bf16(acc[3]) << 16);
                pk.z = (unsigned)f32_to_bf16(acc[4]) | ((unsigned)f32_to_bf16(acc[5]) << 16);
                pk.w = (unsigned)f32_to_bf16(acc[6]) | ((unsigned)f32_to_bf16(acc[7]) << 16);
                *reinterpret_cast<uint4*>(o) = pk;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) st1<T>(o + i, acc[i]);
            }
        }
    }
}

// Causal prefill of the text prompt (stage2/layers.py:107-111: every prompt token attends to itself and its predecessors), FAST
// precision, head size 64, up to 64 tokens: ONE wave per (sample, head) computes the whole T x T attention on the matrix cores
// instead of one wave per query re-reading the keys.
//   S^T = K Q^T   (v_mfma_f32_32x32x16_bf16; A = K rows, B = Q rows: both fragments are 16 contiguous bytes of a cache / q row, loaded
//                  straight from memory -- every K, Q, V element is fetched exactly once per (sample, head))
// leaves the scores of a query in ONE lane (column = query, the 16 registers of a tile = keys), so the causal mask, the maximum and
// the sum of the softmax are register loops plus one cross-half shuffle.  The probabilities then serve, converted to bf16 in place,
// as the B operand of
//   O^T = V^T P^T (A = V^T: element j of lane half h in k-step s is key 16 s + 8 (j >> 2) + 4 h + (j & 3), the order in which the
//                  accumulator registers hold P -- cdna_hip_programming.md, 'An accumulator tile as the next MFMA's operand')
// Key tiles above the diagonal are skipped.  Outputs leave as 8-byte pieces of a row (4 consecutive head dimensions per register quad).
template <int NT>
__global__ __launch_bounds__(256) void attention_prefill_mfma_kernel(AttnArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
    typedef __attribute__((ext_vector_type(16))) float f32x16_t;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gid = blockIdx.x * 4 + wave;
    if (gid >= a.B * a.n_heads) return;
    const int h = gid % a.n_heads, b = gid / a.n_heads;
    const int D = a.n_heads * 64, T = a.Tq;
    const int r = lane & 31, hf = lane >> 5;
    const bf16_t* qb = reinterpret_cast<const bf16_t*>(a.q) + (long long)b * T * D + h * 64;
    const bf16_t* kb = reinterpret_cast<const bf16_t*>(a.kcache) + (long long)b * a.Tmax * D + h * 64;
    const bf16_t* vb = reinterpret_cast<const bf16_t*>(a.vcache) + (long long)b * a.Tmax * D + h * 64;
    bf16x8_t kf[NT][4], qf[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const long long row = min(32 * t + r, T - 1);                 // clamped: rows beyond the prompt repeat its last token and are masked / not stored
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[t][ks] = *reinterpret_cast<const bf16x8_t*>(kb + row * D + 16 * ks + 8 * hf);
            qf[t][ks] = *reinterpret_cast<const bf16x8_t*>(qb + row * D + 16 * ks + 8 * hf);
        }
    }
    // V^T fragments: element j <- V[key(kt, s, hf, j)][32 dt + r]; all loads in flight while the scores are computed
    unsigned short vraw[2][NT][2][8];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const long long key = min(32 * kt + 16 * s2 + 8 * (j >> 2) + 4 * hf + (j & 3), T - 1);
                    vraw[dt][kt][s2][j] = vb[key * D + 32 * dt + r];
                }
    const float scale = 1.0f / sqrtf(64.0f);
    float inv_sum[NT];
    bf16x8_t pf[NT][NT][2];                                           // [key tile][query tile][k-step]: P^T as the B operand
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
        f32x16_t sc[NT];
        const int q = 32 * qt + r;
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt <= qt; ++kt) {
            f32x16_t acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt][ks], qf[qt][ks], acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = 32 * kt + (i & 3) + 8 * (i >> 2) + 4 * hf;
                const float v = key <= q ? acc[i] * scale : -INFINITY;
                acc[i] = v;
                m = fmaxf(m, v);
            }
            sc[kt] = acc;
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));                          // the other half of this query's keys; finite: key 0 is always visible
        float sum = 0.0f;
#pragma unroll
        for (int kt = 0; kt <= qt; ++kt) {
            float e[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { e[i] = __expf(sc[kt][i] - m); sum += e[i]; }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8_t f;
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = (__bf16)e[8 * s2 + j];
                pf[kt][qt][s2] = f;
            }
        }
        sum += __shfl_xor(sum, 32, 64);
        inv_sum[qt] = 1.0f / sum;
    }
    const int row_base = b * T;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        bf16x8_t vf[NT][2];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;
                u16x8_t raw;
#pragma unroll
                for (int j = 0; j < 8; ++j) raw[j] = vraw[dt][kt][s2][j];
                vf[kt][s2] = __builtin_bit_cast(bf16x8_t, raw);
            }
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) {
            f32x16_t o;
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i] = 0.0f;
#pragma unroll
            for (int kt = 0; kt <= qt; ++kt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kt][s2], pf[kt][qt][s2], o, 0, 0, 0);
            const int q = 32 * qt + r;
            if (q < T) {
                const int row = row_base + q;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int col = h * 64 + 32 * dt + 8 * g4 + 4 * hf;
                    bf16_t* dst = a.out_packed_mb ? reinterpret_cast<bf16_t*>(a.out) + packed_off(row, col, a.out_packed_mb)
                                                  : reinterpret_cast<bf16_t*>(a.out) + (long long)row * D + col;
                    uint2 pk;
                    pk.x = (unsigned)f32_to_bf16(o[4 * g4] * inv_sum[qt]) | ((unsigned)f32_to_bf16(o[4 * g4 + 1] * inv_sum[qt]) << 16);
                    pk.y = (unsigned)f32_to_bf16(o[4 * g4 + 2] * inv_sum[qt]) | ((unsigned)f32_to_bf16(o[4 * g4 + 3] * inv_sum[qt]) << 16);
                    *reinterpret_cast<uint2*>(dst) = pk;
                }
            }
        }
    }
#endif
}

// The same few-query case with head size 64 and at most 8 keys (depth sub-step 1: 4 queries x 5 keys), from 64 samples: EIGHT heads per
// wave.  A head is the 8 lanes that cover one 128-byte key / value row, and all its keys sit in that group's registers, so a wave holds
// 8 x (NK keys + NK values + Tq queries) x 16 bytes in flight instead of one head's, an eighth of the waves are dispatched (at 2048
// samples 6 144 instead of 49 152: the launch was paced by occupancy rounds of one-round-trip waves; 73.6 -> 21.2 us, 8.0 -> 5.3 us at 64
// samples: tools/micro/bench_attn), and every lane stores.  Scores, maxima and exponentials are the very operations of attention_fewq_kernel; the sums over the keys are taken in the
// order of its cross-lane tree (slot ^ 1, ^ 2, ^ 4), so the results are bit-identical.
template <typename T, int NK, int TQ>
__global__ __launch_bounds__(256) void attention_fewq8_kernel(AttnArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gid = (blockIdx.x * 4 + wave) * 8 + (lane >> 3);          // (sample, head) of this 8-lane group
    const bool live = gid < a.B * a.n_heads;
    const int g = live ? gid : a.B * a.n_heads - 1;
    const int h = g % a.n_heads, b = g / a.n_heads, c = lane & 7;
    constexpr int hs = 64;
    const int D = a.n_heads * hs;
    const T* kc = reinterpret_cast<const T*>(a.kcache) + (long long)b * a.Tmax * D + h * hs + c * 8;
    const T* vc = reinterpret_cast<const T*>(a.vcache) + (long long)b * a.Tmax * D + h * hs + c * 8;
    const T* qp = reinterpret_cast<const T*>(a.q) + (long long)b * a.Tq * D + h * hs + c * 8;
    typedef typename std::conditional<sizeof(T) == 2, uint4, float4>::type raw_t;
    constexpr int NRAW = sizeof(T) == 2 ? 1 : 2;
    raw_t kbuf[NK][NRAW], vbuf[NK][NRAW], qbuf[TQ][NRAW];
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        const raw_t* ks = reinterpret_cast<const raw_t*>(kc + (long long)j * D);
        const raw_t* vs = reinterpret_cast<const raw_t*>(vc + (long long)j * D);
#pragma unroll
        for (int e = 0; e < NRAW; ++e) { kbuf[j][e] = ks[e]; vbuf[j][e] = vs[e]; }
    }
#pragma unroll
    for (int qi = 0; qi < TQ; ++qi) {
        const raw_t* qs = reinterpret_cast<const raw_t*>(qp + (long long)min(qi, a.Tq - 1) * D);
#pragma unroll
        for (int e = 0; e < NRAW; ++e) qbuf[qi][e] = qs[e];
    }
    auto unpack = [](const raw_t* r, float (&f)[8]) {
        if (sizeof(T) == 2) {
            const uint4 t = *reinterpret_cast<const uint4*>(r);
            const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) { f[2 * i] = bf16_to_f32((bf16_t)(w[i] & 0xffffu)); f[2 * i + 1] = bf16_to_f32((bf16_t)(w[i] >> 16)); }
        } else {
            const float4 a0 = *reinterpret_cast<const float4*>(r), a1 = *reinterpret_cast<const float4*>(r + 1);
            f[0] = a0.x; f[1] = a0.y; f[2] = a0.z; f[3] = a0.w; f[4] = a1.x; f[5] = a1.y; f[6] = a1.z; f[7] = a1.w;
        }
    };
    const float scale = 0.125f;                                        // 1 / sqrt(64), exact
    float kf[NK][8];
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        unpack(kbuf[j], kf[j]);
#pragma unroll
        for (int i = 0; i < 8; ++i) kf[j][i] *= scale;                 // scale on K, as layers.py:102
    }
#pragma unroll
    for (int qi = 0; qi < TQ; ++qi) {
        if (qi >= a.Tq) break;
        const int nkeys = a.causal ? a.t_base + qi + 1 : a.t_base + a.Tq;
        float qv[8];
        unpack(qbuf[qi], qv);
        float sc[8], gmax = -INFINITY;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sc[j] = -INFINITY;
            if (j < NK) {
                float s = 0.0f;
#pragma unroll
                for (int i = 0; i < 8; ++i) s = fmaf(qv[i], kf[j][i], s);
                s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
                if (j < nkeys) sc[j] = s;
            }
            gmax = fmaxf(gmax, sc[j]);
        }
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = j < NK ? expf(sc[j] - gmax) : 0.0f;      // 0 for masked keys
        const float run_sum = ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
        const float inv = 1.0f / run_sum;
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float p[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] = 0.0f;
#pragma unroll
            for (int j = 0; j < NK; ++j) {
                float vv[8];
                unpack(vbuf[j], vv);
                p[j] = fmaf(e[j], vv[i], 0.0f);
            }
            acc[i] = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) * inv;
        }
        if (live) {
            const int row = b * a.Tq + qi, col = h * hs + c * 8;
            T* o = a.out_packed_mb ? reinterpret_cast<T*>(a.out) + packed_off(row, col, a.out_packed_mb)
                                   : reinterpret_cast<T*>(a.out) + (long long)row * D + col;
            if (sizeof(T) == 2) {
                uint4 pk;
                pk.x = (unsigned)f32_to_bf16(acc[0]) | ((unsigned)f32_to_bf16(acc[1]) << 16);
                pk.y = (unsigned)f32_to_bf16(acc[2]) | ((unsigned)f32_to_bf16(acc[3]) << 16);
                pk.z = (unsigned)f32_to_bf16(acc[4]) | ((unsigned)f32_to_bf16(acc[5]) << 16);
                pk.w = (unsigned)f32_to_bf16(acc[6]) | ((unsigned)f32_to_bf16(acc[7]) << 16);
                *reinterpret_cast<uint4*>(o) = pk;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) st1<T>(o + i, acc[i]);
            }
        }
    }
}
// One query per sample over the body's KV cache (a decode step of the AR loop), head size 64: EIGHT heads per wave, like
// attention_fewq8_kernel.  The 8 lanes of a head walk its keys in chunks of CH rows (K and V of a chunk in flight together; the bf16
// kernel fetches chunk i + 1 before it works on chunk i), scores by 8 FMAs and a 3-step shuffle inside the lane group, online softmax
// across chunks in registers, every lane stores its 8 outputs.  Against attention_kernel (one head per wave, 8 keys per pass across the
// lane groups) a launch dispatches an eighth of the waves -- with one key cached that kernel took 14 us at 512 samples and 57 us at
// 2048 for 3 / 12 MB: occupancy rounds of one-round-trip waves, under every launch of the pass -- and a key row of 8 neighbouring heads
// is one contiguous kilobyte.  The sums run over the keys in index order (attention_kernel: a tree across lane groups), so results
// differ from that kernel's in the last bits (FAST passes only: see launch_attention).
template <typename T>
__global__ __launch_bounds__(256) void attention_heads8_kernel(AttnArgs a) {
    constexpr int hs = 64, CH = sizeof(T) == 2 ? 8 : 4, NRAW = sizeof(T) == 2 ? 1 : 2;
    constexpr bool PREFETCH = sizeof(T) == 2;
    typedef typename std::conditional<sizeof(T) == 2, uint4, float4>::type raw_t;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gid = (blockIdx.x * 4 + wave) * 8 + (lane >> 3);
    const bool live = gid < a.B * a.n_heads;
    const int g = live ? gid : a.B * a.n_heads - 1;
    const int h = g % a.n_heads, b = g / a.n_heads, c = lane & 7;
    const int D = a.n_heads * hs;
    const T* kc = reinterpret_cast<const T*>(a.kcache) + (long long)b * a.Tmax * D + h * hs + c * 8;
    const T* vc = reinterpret_cast<const T*>(a.vcache) + (long long)b * a.Tmax * D + h * hs + c * 8;
    auto unpack = [](const raw_t* r, float (&f)[8]) {
        if (sizeof(T) == 2) {
            const uint4 t = *reinterpret_cast<const uint4*>(r);
            const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) { f[2 * i] = bf16_to_f32((bf16_t)(w[i] & 0xffffu)); f[2 * i + 1] = bf16_to_f32((bf16_t)(w[i] >> 16)); }
        } else {
            const float4 a0 = *reinterpret_cast<const float4*>(r), a1 = *reinterpret_cast<const float4*>(r + 1);
            f[0] = a0.x; f[1] = a0.y; f[2] = a0.z; f[3] = a0.w; f[4] = a1.x; f[5] = a1.y; f[6] = a1.z; f[7] = a1.w;
        }
    };
    raw_t qraw[NRAW];
    {
        const raw_t* qs = reinterpret_cast<const raw_t*>(reinterpret_cast<const T*>(a.q) + (long long)b * D + h * hs + c * 8);
#pragma unroll
        for (int e = 0; e < NRAW; ++e) qraw[e] = qs[e];                 // independent of the step state: in flight while t_base arrives
    }
    const int nkeys = a.t_base + (a.t_base_dev ? *a.t_base_dev : 0) + 1;       // Tq == 1: causal or not, the query sees every cached key and itself
    raw_t kb[2][CH][NRAW], vb[2][CH][NRAW];
    auto fetch = [&](int j0, int slot) {
#pragma unroll
        for (int p = 0; p < CH; ++p) {
            const long long j = min(j0 + p, nkeys - 1);                 // clamped: unconditional loads, all in flight at once
            const raw_t* ks = reinterpret_cast<const raw_t*>(kc + j * D);
            const raw_t* vs = reinterpret_cast<const raw_t*>(vc + j * D);
#pragma unroll
            for (int e = 0; e < NRAW; ++e) { kb[slot][p][e] = ks[e]; vb[slot][p][e] = vs[e]; }
        }
    };
    float qv[8];
    float run_max = -INFINITY, run_sum = 0.0f, acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
    auto work = [&](int j0, int slot) {
        float sc[CH], cmax = -INFINITY;
#pragma unroll
        for (int p = 0; p < CH; ++p) {
            float kv[8];
            unpack(kb[slot][p], kv);
            float sdot = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) sdot = fmaf(qv[i], kv[i] * 0.125f, sdot);     // scale 1 / sqrt(64) on K, as layers.py:102
            sdot += __shfl_xor(sdot, 4, 64); sdot += __shfl_xor(sdot, 2, 64); sdot += __shfl_xor(sdot, 1, 64);
            sc[p] = (j0 + p < nkeys) ? sdot : -INFINITY;
            cmax = fmaxf(cmax, sc[p]);
        }
        const float new_max = fmaxf(run_max, cmax);                     // finite: key j0 is valid
        const float rescale = expf(run_max - new_max);                  // 0 for the first chunk
        run_sum *= rescale;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] *= rescale;
#pragma unroll
        for (int p = 0; p < CH; ++p) {
            const float e = expf(sc[p] - new_max);                      // 0 for masked rows
            run_sum += e;
            float vv[8];
            unpack(vb[slot][p], vv);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fmaf(e, vv[i], acc[i]);
        }
        run_max = new_max;
    };
    fetch(0, 0);
    unpack(qraw, qv);
    if (PREFETCH) {
        for (int j0 = 0; j0 < nkeys; j0 += 2 * CH) {                    // two chunks per trip: static register slots
            if (j0 + CH < nkeys) fetch(j0 + CH, 1);
            work(j0, 0);
            if (j0 + CH >= nkeys) break;
            if (j0 + 2 * CH < nkeys) fetch(j0 + 2 * CH, 0);
            work(j0 + CH, 1);
        }
    } else {
        for (int j0 = 0; j0 < nkeys; j0 += CH) {
            if (j0 > 0) fetch(j0, 0);
            work(j0, 0);
        }
    }
    const float inv = 1.0f / run_sum;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] *= inv;
    if (live) {
        const int col = h * hs + c * 8;
        T* o = a.out_packed_mb ? reinterpret_cast<T*>(a.out) + packed_off(b, col, a.out_packed_mb)
                               : reinterpret_cast<T*>(a.out) + (long long)b * D + col;
        if (sizeof(T) == 2) {
            uint4 pk;
            pk.x = (unsigned)f32_to_bf16(acc[0]) | ((unsigned)f32_to_bf16(acc[1]) << 16);
            pk.y = (unsigned)f32_to_bf16(acc[2]) | ((unsigned)f32_to_bf16(acc[3]) << 16);
            pk.z = (unsigned)f32_to_bf16(acc[4]) | ((unsigned)f32_to_bf16(acc[5]) << 16);
            pk.w = (unsigned)f32_to_bf16(acc[6]) | ((unsigned)f32_to_bf16(acc[7]) << 16);
            *reinterpret_cast<uint4*>(o) = pk;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) st1<T>(o + i, acc[i]);
        }
    }
}
template <typename T>
static hipError_t launch_fewq8(const AttnArgs& a, hipStream_t st) {
    const int groups = a.B * a.n_heads, grid = (groups + 31) / 32;
    switch (a.t_base + a.Tq) {
    case 2: attention_fewq8_kernel<T, 2, 4><<<grid, 256, 0, st>>>(a); break;
    case 3: attention_fewq8_kernel<T, 3, 4><<<grid, 256, 0, st>>>(a); break;
    case 4: attention_fewq8_kernel<T, 4, 4><<<grid, 256, 0, st>>>(a); break;
    case 5: attention_fewq8_kernel<T, 5, 4><<<grid, 256, 0, st>>>(a); break;
    case 6: attention_fewq8_kernel<T, 6, 4><<<grid, 256, 0, st>>>(a); break;
    case 7: attention_fewq8_kernel<T, 7, 4><<<grid, 256, 0, st>>>(a); break;
    default: attention_fewq8_kernel<T, 8, 4><<<grid, 256, 0, st>>>(a); break;
    }
    return hipGetLastError();
}

hipError_t launch_attention(const AttnArgs& a, hipStream_t st) {
    const int chunks = a.head_dim / 8;
    if (a.head_dim % 8 != 0 || chunks > 64 || (chunks & (chunks - 1)) != 0) return hipErrorInvalidValue;
    // few queries over a cache known on the host to fit one pass (depth sub-steps): one wave per (sample, head), see attention_fewq_kernel
    if (a.Tq > 1 && a.Tq <= 16 && !a.t_base_dev && !a.dbg && a.t_base + a.Tq <= 2 * (64 / chunks)) {
        static const bool off = getenv("HQT_NO_FEWQ_ATTN") != nullptr;          // A/B switch
        if (!off) {
            // head size 64, <= 4 queries over <= 8 keys, 64+ samples: eight heads per wave (attention_fewq8_kernel; HQT_NO_FEWQ8=1: A/B switch)
            static const bool off8 = getenv("HQT_NO_FEWQ8") != nullptr;
            if (!off8 && a.head_dim == 64 && a.Tq <= 4 && a.t_base + a.Tq <= 8 && a.B >= 64)
                return a.dtype == DT_BF16 ? launch_fewq8<bf16_t>(a, st) : launch_fewq8<float>(a, st);
            const int g2 = (a.B * a.n_heads + 3) / 4, one = a.t_base + a.Tq <= 64 / chunks;
            if (a.dtype == DT_BF16) { if (one) attention_fewq_kernel<bf16_t, 1><<<g2, 256, 0, st>>>(a); else attention_fewq_kernel<bf16_t, 2><<<g2, 256, 0, st>>>(a); }
            else { if (one) attention_fewq_kernel<float, 1><<<g2, 256, 0, st>>>(a); else attention_fewq_kernel<float, 2><<<g2, 256, 0, st>>>(a); }
            return hipGetLastError();
        }
    }
    // causal prefill of a whole prompt (bf16, head size 64, nothing cached before it): the matrix-core kernel, one wave per (sample, head)
    if (a.causal && a.dtype == DT_BF16 && a.head_dim == 64 && a.Tq > 4 && a.Tq <= 64 && a.t_base == 0 && !a.t_base_dev && !a.dbg &&
        (a.n_heads * 64) % 8 == 0) {
        const int g2 = (a.B * a.n_heads + 3) / 4;
        if (a.Tq <= 32) attention_prefill_mfma_kernel<1><<<g2, 256, 0, st>>>(a);
        else attention_prefill_mfma_kernel<2><<<g2, 256, 0, st>>>(a);
        return hipGetLastError();
    }
    // one query per sample, head size 64 (the decode steps of a merged FAST pass): eight heads per wave (HQT_NO_HEADS8=1: A/B switch)
    static const bool no_h8 = getenv("HQT_NO_HEADS8") != nullptr;
    // FAST only, from 256 samples: below that the launch is a handful of waves and the chunk-by-chunk walk loses to attention_kernel's
    // single round trip from 32 keys on (64 samples x 64 keys: 19.1 vs 9.3 us); EXACT keeps ONE kernel for every row count, so that a
    // row's draws do not depend on the pass it sits in (hqt.h: merged steps).
    if (!no_h8 && a.Tq == 1 && a.head_dim == 64 && !a.dbg && a.dtype == DT_BF16 && a.B >= 256) {
        attention_heads8_kernel<bf16_t><<<(a.B * a.n_heads + 31) / 32, 256, 0, st>>>(a);
        return hipGetLastError();
    }
    const int grid = (a.B * a.n_heads * a.Tq + 3) / 4;
    if (a.dtype == DT_BF16) attention_kernel<bf16_t><<<grid, 256, 0, st>>>(a);
    else attention_kernel<float><<<grid, 256, 0, st>>>(a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// A7/K6: fused sampler -- temperature, top-k, softmax, top-p, argmax(p / q), code write-back.
// One 256-thread workgroup per logits row.  Restates hqvae/utils/sampling.py:12-37 and
// hierarchical_ar.py:762-785; torch.multinomial(p, 1) == argmax(p / q), q ~ Exp(1).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return __umulhi(a, b); }
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// NT threads per logits row: 1024 (16 waves, 4 per SIMD) hides the latency of the exp / log / divide / Philox chains that a
// lone 4-wave workgroup per CU exposes (25 -> ~10 us per launch at V = 8192); 256 for small vocabularies.
// FM: fast-math forms for FAST-precision calls.  The plain path (no top-k / top-p) is bound by the IEEE expf / logf / divisions of its
// 8192 entries per row, not by memory (21.5 us per 512 rows); EXACT calls keep them: their draws are the parity gate, compared bit for bit with the CPU restatement of the reference.
template <int NT, bool FM>
__global__ __launch_bounds__(NT, 8) void sampler_kernel(SamplerArgs a, int n2) {   // 8 waves per SIMD = two 1024-thread rows per CU: the plain path is 30 % slower at 7 (tools/micro/bench_sampler)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // layout: double dsum[NT]; float redf[16]; int redi[16]; unsigned hist[256]; int sel[4]; float lp[V];
    //         (top-p only) float skey[n2]; unsigned short sidx[n2]; unsigned char keep[V]
    double* dsum = reinterpret_cast<double*>(smem_raw);
    float* redf = reinterpret_cast<float*>(dsum + NT);
    int* redi = reinterpret_cast<int*>(redf + 16);
    unsigned* hist = reinterpret_cast<unsigned*>(redi + 16);
    int* sel = reinterpret_cast<int*>(hist + 256);
    float* lp = reinterpret_cast<float*>(sel + 4);
    const int V = a.V, tid = threadIdx.x;
    float* skey = lp + V;
    unsigned short* sidx = reinterpret_cast<unsigned short*>(skey + n2);
    unsigned char* keep = reinterpret_cast<unsigned char*>(sidx + n2);

    const int r = blockIdx.x, b = r / a.slots, slot = r % a.slots;
    const int step = a.state->step;
    const int draw = a.draw0 + slot;
    const float* lg = a.logits + (long long)r * V;
    const int draws = a.draws > 0 ? a.draws : 5;
    const long long nidx = (((long long)step * draws + draw) * a.B + b) * V;

    // ---- temperature (logits /= T, hierarchical_ar.py:763,779) and raw-logit dump
    const float inv_t = FM ? __builtin_amdgcn_rcpf(a.temperature) : 0.0f;
    for (int i = tid; i < V; i += NT) {
        const float v = lg[i];
        if (a.logits_out) a.logits_out[nidx + i] = v;
        lp[i] = FM ? v * inv_t : v / a.temperature;
    }
    __syncthreads();

    // ---- top-k: exact k-th largest by 4-pass radix select, keep >= threshold (sampling.py:12-19)
    if (a.top_k > 0 && a.top_k < V) {
        uint32_t prefix = 0;
        int remaining = a.top_k;
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const uint32_t himask = shift == 24 ? 0u : ~((1u << (shift + 8)) - 1u);
            for (int i = tid; i < V; i += NT) {
                const uint32_t o = f2ord(lp[i]);
                if ((o & himask) == prefix) atomicAdd(&hist[(o >> shift) & 255u], 1u);
            }
            __syncthreads();
            // the bin that holds the k-th largest = the highest bin b >= 1 with (elements in bins above b) + hist[b] >= remaining, else
            // bin 0 (what a scan from bin 255 downwards finds): suffix sums over the 256 bins by 4 waves instead of 256 dependent LDS
            // reads on one thread
            if (tid == 0) sel[0] = 0;
            unsigned h = 0, incl = 0, above = 0;
            const int lane_k = tid & 63, wave_k = tid >> 6;
            if (tid < 256) {
                h = hist[tid];
                incl = h;
                for (int off = 1; off < 64; off <<= 1) {
                    const unsigned v = (unsigned)__shfl_down((int)incl, off, 64);
                    if (lane_k + off < 64) incl += v;
                }
                if (lane_k == 0) redi[wave_k] = (int)incl;
            }
            __syncthreads();
            if (tid < 256) {
                above = incl - h;
                for (int w = wave_k + 1; w < 4; ++w) above += (unsigned)redi[w];
                if (tid >= 1 && above + h >= (unsigned)remaining) atomicMax(&sel[0], tid);
            }
            __syncthreads();
            const int bin_sel = sel[0];
            if (tid == bin_sel) sel[1] = remaining - (int)above;
            __syncthreads();
            prefix |= ((uint32_t)bin_sel) << shift;
            remaining = sel[1];
            __syncthreads();
        }
        for (int i = tid; i < V; i += NT)
            if (f2ord(lp[i]) < prefix) lp[i] = -INFINITY;
        __syncthreads();
    }

    // ---- softmax (fp32): p = exp(l - max) / sum
    float m = -INFINITY;
    for (int i = tid; i < V; i += NT) m = fmaxf(m, lp[i]);
    m = block_reduce(m, OpMax(), redf);
    float s = 0.0f;
    for (int i = tid; i < V; i += NT) { const float e = FM ? __expf(lp[i] - m) : expf(lp[i] - m); lp[i] = e; s += e; }
    s = block_reduce(s, OpAdd(), redf);
    const float inv_s = FM ? __builtin_amdgcn_rcpf(s) : 0.0f;
    for (int i = tid; i < V; i += NT) lp[i] = FM ? lp[i] * inv_s : lp[i] / s;
    __syncthreads();

    // ---- top-p (sampling.py:22-37): descending sort, prefix sums accumulated in double and rounded
    //      to fp32 per element (what torch.cumsum does on the CPU), cut after the first prefix >= p
    float renorm = 1.0f;
    const bool use_p = a.top_p > 0.0f;
    if (use_p) {
        // Only entries with p > 0 can matter (zeros -- everything top-k masked -- sort last, add nothing to the prefix sums and stay zero
        // whether kept or not), so they alone are compacted, sorted and scanned: with top_k = 2048 of V = 8192 the bitonic network
        // shrinks from 91 passes x 8 elements per thread to 66 x 2 (1.1 ms -> see profiles/ per launch at 2048 rows).  Same results bit
        // for bit: the network orders (probability desc, index asc) pairs, a total order that does not depend on where they start.
        const int per = (V + NT - 1) / NT, lane_ = tid & 63, wave_ = tid >> 6;
        int c_local = 0;
        for (int c = 0; c < per; ++c) { const int j = tid * per + c; if (j < V && lp[j] > 0.0f) ++c_local; }
        int incl = c_local;
        for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane_ >= off) incl += v; }
        if (lane_ == 63) redi[wave_] = incl;
        __syncthreads();
        int pos = incl - c_local, cnt = 0;
        for (int w = 0; w < NT / 64; ++w) { const int t = redi[w]; if (w < wave_) pos += t; cnt += t; }
        for (int c = 0; c < per; ++c) {
            const int j = tid * per + c;
            if (j < V && lp[j] > 0.0f) { skey[pos] = lp[j]; sidx[pos] = (unsigned short)j; ++pos; }
        }
        int n2e = 256;
        while (n2e < cnt) n2e <<= 1;                                 // <= n2 (cnt <= V)
        for (int i = cnt + tid; i < n2e; i += NT) { skey[i] = -1.0f; sidx[i] = (unsigned short)i; }
        __syncthreads();
        // The bitonic network with the elements in registers: element e of thread t is position e NT + t, so a partner at distance
        // j >= NT is another register of the same thread, at j < 64 a lane of the same wave (one shuffle each for key and index), and
        // only 64 <= j < NT goes through LDS with barriers: 14 of the 66 steps at 2048 entries on 1024 threads.
        constexpr int EMAX = 2;
        const int E = n2e > NT ? n2e / NT : 1;
        const bool in_regs = E <= EMAX;
        if (in_regs) {
            float kv[EMAX];
            int iv[EMAX];
#pragma unroll
            for (int e = 0; e < EMAX; ++e) {
                const int i = e * NT + tid;
                kv[e] = (e < E && i < n2e) ? skey[i] : -2.0f;
                iv[e] = (e < E && i < n2e) ? (int)sidx[i] : 0xFFFF;
            }
            auto first = [](float ka, int ia, float kb, int ib) { return ka > kb || (ka == kb && ia < ib); };      // a before b: prob desc, index asc
            for (int k2 = 2; k2 <= n2e; k2 <<= 1) {
                for (int j = k2 >> 1; j > 0; j >>= 1) {
                    if (j >= NT) {                                    // (E == 2, j == NT) partner = the thread's other register
                        const bool up = (tid & k2) == 0;              // position of register 0; bit j is clear in it
                        const bool a_first = first(kv[0], iv[0], kv[1], iv[1]);
                        if (up ? !a_first : a_first) { const float tk = kv[0]; kv[0] = kv[1]; kv[1] = tk; const int ti = iv[0]; iv[0] = iv[1]; iv[1] = ti; }
                    } else if (j >= 64) {                             // partner in another wave: through LDS
                        __syncthreads();
#pragma unroll
                        for (int e = 0; e < EMAX; ++e) { const int i = e * NT + tid; if (e < E && i < n2e) { skey[i] = kv[e]; sidx[i] = (unsigned short)iv[e]; } }
                        __syncthreads();
#pragma unroll
                        for (int e = 0; e < EMAX; ++e) {
                            const int i = e * NT + tid;
                            if (e < E && i < n2e) {
                                const float kb = skey[i ^ j]; const int ib = sidx[i ^ j];
                                const bool lower = (i & j) == 0, up = (i & k2) == 0;
                                const bool me_first = first(kv[e], iv[e], kb, ib);
                                if ((lower == up) ? !me_first : me_first) { kv[e] = kb; iv[e] = ib; }      // lower & up (or upper & down) keeps the first
                            }
                        }
                    } else {                                          // partner = lane ^ j of this wave
#pragma unroll
                        for (int e = 0; e < EMAX; ++e) {
                            const int i = e * NT + tid;
                            const float kb = __shfl_xor(kv[e], j, 64); const int ib = __shfl_xor(iv[e], j, 64);
                            const bool lower = (i & j) == 0, up = (i & k2) == 0;
                            const bool me_first = first(kv[e], iv[e], kb, ib);
                            if ((lower == up) ? !me_first : me_first) { kv[e] = kb; iv[e] = ib; }
                        }
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < EMAX; ++e) { const int i = e * NT + tid; if (e < E && i < n2e) { skey[i] = kv[e]; sidx[i] = (unsigned short)iv[e]; } }
            __syncthreads();
        }
        for (int k2 = 2; !in_regs && k2 <= n2e; k2 <<= 1) {
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < n2e; i += NT) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const float ka = skey[i], kb = skey[ixj];
                        const unsigned short ia = sidx[i], ib = sidx[ixj];
                        const bool a_first = ka > kb || (ka == kb && ia < ib);   // order: prob desc, index asc
                        const bool up = (i & k2) == 0;
                        if (up ? !a_first : a_first) { skey[i] = kb; skey[ixj] = ka; sidx[i] = ib; sidx[ixj] = ia; }
                    }
                }
                __syncthreads();
            }
        }
        const int chunk = (n2e + NT - 1) / NT;
        double local = 0.0;
        for (int c = 0; c < chunk; ++c) { const int j = tid * chunk + c; if (j < cnt) local += (double)skey[j]; }
        // exclusive prefix of the per-thread sums: shuffle scan inside each wave, then the totals of the waves in front (a 1024-step
        // serial loop on one thread before: ~9 us per row).  Everything is accumulated in double, as torch.cumsum does for an fp32 row;
        // the association differs from a strictly sequential sum only below 2^-53 relative.
        double scan = local;
        for (int off = 1; off < 64; off <<= 1) { const double v = __shfl_up(scan, off, 64); if (lane_ >= off) scan += v; }
        if (lane_ == 63) dsum[wave_] = scan;
        __syncthreads();
        double run = scan - local;
        for (int w = 0; w < wave_; ++w) run += dsum[w];
        int first = V;                                   // first sorted position whose prefix >= p
        for (int c = 0; c < chunk; ++c) {
            const int j = tid * chunk + c;
            if (j < cnt) {
                run += (double)skey[j];
                if ((float)run >= a.top_p && j < first) first = j;
            }
        }
        first = block_reduce(first, OpMin(), redi);
        for (int i = tid; i < V; i += NT) keep[i] = 0;
        __syncthreads();
        for (int j = tid; j < cnt; j += NT)
            if (j <= first) keep[sidx[j]] = 1;           // position `first` itself is kept (shifted mask)
        __syncthreads();
        float ks = 0.0f;
        for (int i = tid; i < V; i += NT) { if (!keep[i]) lp[i] = 0.0f; ks += lp[i]; }
        renorm = block_reduce(ks, OpAdd(), redf);
        __syncthreads();
    }

    // ---- draw: argmax_i p_i / q_i, lowest index on ties
    float best = -1.0f;
    int besti = 0;
    const uint64_t seed = a.rows[b].seed;                // per-call / per-row values live in device memory: the captured graph is call-independent
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const uint64_t grow = (uint64_t)a.rows[b].global_row;
    for (int i4 = tid; i4 * 4 < V; i4 += NT) {
        float q[4];
        if (a.noise) {
#pragma unroll
            for (int e = 0; e < 4; ++e) q[e] = (i4 * 4 + e < V) ? a.noise[nidx + i4 * 4 + e] : 1.0f;
        } else {
            uint32_t rnd[4];
            philox4x32_10((uint32_t)i4, (uint32_t)(step * draws + draw), (uint32_t)grow, (uint32_t)(grow >> 32), k0, k1, rnd);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float u = ((float)(rnd[e] >> 9) + 0.5f) * (1.0f / 8388608.0f);     // 23 bits + 0.5: exact in fp32, u in [2^-24, 1 - 2^-24] -- never 1.0 (q = 0)
                q[e] = FM ? -__logf(u) : -logf(u);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = i4 * 4 + e;
            if (i < V) {
                float p = lp[i];
                if (use_p) p = p / renorm;
                const float ratio = FM ? p * __builtin_amdgcn_rcpf(q[e]) : p / q[e];
                if (ratio > best) { best = ratio; besti = i; }      // ascending i per lane: first max wins
            }
        }
    }
    // reduce (value desc, index asc)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(besti, off, 64);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    __syncthreads();
    if ((tid & 63) == 0) { redf[tid >> 6] = best; redi[tid >> 6] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < NT / 64; ++w)
            if (redf[w] > best || (redf[w] == best && redi[w] < besti)) { best = redf[w]; besti = redi[w]; }
        a.out[a.out_stride > 0 ? ((long long)b * a.n_steps + step) * a.out_stride + a.out_slot : ((long long)b * a.n_steps + step) * a.slots + slot] = (int64_t)besti;
        if (a.emb_tok) sel[0] = a.emb_feed ? (int)clamp_idx(a.emb_feed[((long long)b * a.n_steps + step) * a.slots + slot], a.V) : besti;
    }
    if (!a.emb_tok) return;                              // workgroup-uniform
    // ---- fused embedding lookup: this workgroup drew the top code of sample b, so it also writes the four depth-token rows
    __syncthreads();
    const long long code = sel[0];
    const int D = a.emb_D;
#pragma unroll 1
    for (int s4 = 0; s4 < 4; ++s4) {
        const int row = b * 4 + s4;
        float rs = 0.0f, rq = 0.0f;
        for (int d = tid; d < D; d += NT) {
            const float v = a.emb_tok[code * D + d] + a.emb_pos[(long long)s4 * D + d];
            a.emb_x[(long long)row * D + d] = v;
            if (a.emb_xpk) {
                const bf16_t hb = f32_to_bf16(v);
                a.emb_xpk[packed_off(row, d, a.emb_pk_mb)] = hb;
                const float r = bf16_to_f32(hb);
                rs += r; rq += r * r;
            }
        }
        if (a.emb_xpk) {                                 // fixed-order reduction: waves, then wave 0 .. NT/64-1
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { rs += __shfl_xor(rs, off, 64); rq += __shfl_xor(rq, off, 64); }
            __syncthreads();
            if ((tid & 63) == 0) { redf[tid >> 6] = rs; dsum[tid >> 6] = (double)rq; }
            __syncthreads();
            if (tid == 0) {
                float ts = 0.0f, tq = 0.0f;
                for (int w = 0; w < NT / 64; ++w) { ts += redf[w]; tq += (float)dsum[w]; }
                a.emb_parts[2 * row] = ts; a.emb_parts[2 * row + 1] = tq;
            }
        }
    }
}

// FAST draws without a cut-off (top_k = top_p = None: the H1 harness settings): at 2048 samples per pass the bottom codes are 8192 rows
// x 8192 logits per position.  argmax_i p_i / q_i does not need p normalised, so the row never
// leaves the registers: 4 waves per row, thread t owns the float4 groups t, t + 256, ... (the Philox counter of a group is its index,
// as in sampler_kernel), one pass for the maximum, one for exp(l - max) / q and the running best, two 4-wave barriers in all -- against
// five block-wide reductions over 16 waves and four trips through a 32-KB LDS row in the general kernel (8192 rows: 239 -> 93 us, 2048
// rows: 65 -> 31 us, tools/micro/bench_sampler).  Same v_exp / v_log / v_rcp forms as sampler_kernel<.., true>; the ratio lacks the factor 1 / sum, which cannot
// change the winner except between ratios that differ in the last bit.
template <int G4>
__global__ __launch_bounds__(256) void sampler_plain_fast_kernel(SamplerArgs a) {
    __shared__ float redf[4];
    __shared__ int redi[4];
    const int V = a.V, tid = threadIdx.x;
    const int r = blockIdx.x, b = r / a.slots, slot = r % a.slots;
    const int step = a.state->step;
    const int draw = a.draw0 + slot;
    const float* lg = a.logits + (long long)r * V;
    const int draws = a.draws > 0 ? a.draws : 5;
    const long long nidx = (((long long)step * draws + draw) * a.B + b) * V;
    const float inv_t = __builtin_amdgcn_rcpf(a.temperature);
    float4 v[G4];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < G4; ++k) {
        const int i4 = tid + 256 * k;
        if (i4 * 4 < V) {                                 // V % 4 == 0
            v[k] = *reinterpret_cast<const float4*>(lg + i4 * 4);
            if (a.logits_out) *reinterpret_cast<float4*>(a.logits_out + nidx + i4 * 4) = v[k];
            v[k].x *= inv_t; v[k].y *= inv_t; v[k].z *= inv_t; v[k].w *= inv_t;
            m = fmaxf(fmaxf(m, fmaxf(v[k].x, v[k].y)), fmaxf(v[k].z, v[k].w));
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((tid & 63) == 0) redf[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    float best = -1.0f;
    int besti = 0;
    const uint64_t seed = a.rows[b].seed;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const uint64_t grow = (uint64_t)a.rows[b].global_row;
#pragma unroll
    for (int k = 0; k < G4; ++k) {
        const int i4 = tid + 256 * k;
        if (i4 * 4 < V) {
            float q[4];
            if (a.noise) {
                const float4 n4 = *reinterpret_cast<const float4*>(a.noise + nidx + i4 * 4);
                q[0] = n4.x; q[1] = n4.y; q[2] = n4.z; q[3] = n4.w;
            } else {
                uint32_t rnd[4];
                philox4x32_10((uint32_t)i4, (uint32_t)(step * draws + draw), (uint32_t)grow, (uint32_t)(grow >> 32), k0, k1, rnd);
#pragma unroll
                for (int e = 0; e < 4; ++e) q[e] = -__logf(((float)(rnd[e] >> 9) + 0.5f) * (1.0f / 8388608.0f));
            }
            const float l[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ratio = __expf(l[e] - m) * __builtin_amdgcn_rcpf(q[e]);
                if (ratio > best) { best = ratio; besti = i4 * 4 + e; }         // ascending index per thread: the first maximum wins
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(besti, off, 64);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    __syncthreads();                                      // redf has been read by everyone
    if ((tid & 63) == 0) { redf[tid >> 6] = best; redi[tid >> 6] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (redf[w] > best || (redf[w] == best && redi[w] < besti)) { best = redf[w]; besti = redi[w]; }
        a.out[a.out_stride > 0 ? ((long long)b * a.n_steps + step) * a.out_stride + a.out_slot : ((long long)b * a.n_steps + step) * a.slots + slot] = (int64_t)besti;
        if (a.emb_tok) redi[0] = a.emb_feed ? (int)clamp_idx(a.emb_feed[((long long)b * a.n_steps + step) * a.slots + slot], a.V) : besti;
    }
    if (!a.emb_tok) return;                              // workgroup-uniform
    // ---- fused embedding lookup, as in sampler_kernel: the four input rows of depth sub-step 1 for the top code just drawn.  The four rows share the token
    // embedding: one pass with every load in flight and ONE reduction for the eight row statistics (row after row -- four dependent round trips and eight
    // barriers -- this tail was ~7 of the kernel's 17 us at 64 rows); per row the same sums in the same order as before.
    __shared__ float reds[4][4], redq[4][4];
    __syncthreads();
    const long long code = redi[0];
    const int D = a.emb_D;
    float rs[4] = {0.0f, 0.0f, 0.0f, 0.0f}, rq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int d = tid; d < D; d += 256) {
        const float t = a.emb_tok[code * D + d];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int row = b * 4 + s4;
            const float x = t + a.emb_pos[(long long)s4 * D + d];
            a.emb_x[(long long)row * D + d] = x;
            if (a.emb_xpk) {
                const bf16_t hb = f32_to_bf16(x);
                a.emb_xpk[packed_off(row, d, a.emb_pk_mb)] = hb;
                const float rr = bf16_to_f32(hb);
                rs[s4] += rr; rq[s4] += rr * rr;
            }
        }
    }
    if (a.emb_xpk) {                                     // fixed-order reduction: lanes, then waves 0 .. 3
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { rs[s4] += __shfl_xor(rs[s4], off, 64); rq[s4] += __shfl_xor(rq[s4], off, 64); }
            if ((tid & 63) == 0) { reds[s4][tid >> 6] = rs[s4]; redq[s4][tid >> 6] = rq[s4]; }
        }
        __syncthreads();
        if (tid < 4) {
            const int row = b * 4 + tid;
            a.emb_parts[2 * row] = ((reds[tid][0] + reds[tid][1]) + reds[tid][2]) + reds[tid][3];
            a.emb_parts[2 * row + 1] = ((redq[tid][0] + redq[tid][1]) + redq[tid][2]) + redq[tid][3];
        }
    }
}

static size_t sampler_smem(int V, bool use_p, int& n2) {
    n2 = 256;
    while (n2 < V) n2 <<= 1;
    size_t sz = 1024 * sizeof(double) + 16 * sizeof(float) + 16 * sizeof(int) + 256 * sizeof(unsigned) + 4 * sizeof(int) +
                (size_t)V * sizeof(float);
    if (use_p) sz += (size_t)n2 * sizeof(float) + (size_t)n2 * sizeof(unsigned short) + (size_t)V;
    return (sz + 15) & ~(size_t)15;
}
// hipFuncSetAttribute applies to the CURRENT device and is only ever RAISED here: the limit is a per-device maximum over every
// (V, top_p) any handle of this process has asked for, so a handle that needs 42 KB can never lower the limit under another
// handle's (or another thread's) 99 KB launch or captured graph.
hipError_t sampler_configure(int V, bool use_top_p) {
    int n2;
    const size_t smem = sampler_smem(V, use_top_p, n2);
    if (smem > 160 * 1024 || (use_top_p && V > 65536)) return hipErrorInvalidValue;
    static std::mutex mu;
    static size_t limit[64] = {};                                   // per device ordinal
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (dev >= 0 && dev < 64 && smem <= limit[dev]) return hipSuccess;
    for (const void* k : {reinterpret_cast<const void*>(sampler_kernel<256, false>), reinterpret_cast<const void*>(sampler_kernel<256, true>),
                          reinterpret_cast<const void*>(sampler_kernel<1024, false>), reinterpret_cast<const void*>(sampler_kernel<1024, true>)}) {
        e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
    }
    if (e == hipSuccess && dev >= 0 && dev < 64) limit[dev] = smem;
    return e;
}
hipError_t launch_sampler(const SamplerArgs& a, hipStream_t st) {
    int n2;
    const size_t smem = sampler_smem(a.V, a.top_p > 0.0f, n2);
    if (smem > 160 * 1024) return hipErrorInvalidValue;               // sampler_configure(V, top_p) ran in sample_run for this call's options
    // FAST, no cut-offs, V <= 8192: the register-resident kernel (HQT_NO_PLAIN_SAMPLER=1: A/B switch)
    static const bool no_plain = getenv("HQT_NO_PLAIN_SAMPLER") != nullptr;
    if (!no_plain && a.fast_math && !(a.top_k > 0 && a.top_k < a.V) && !(a.top_p > 0.0f) && a.V % 4 == 0 && a.V <= 8192 && a.V >= 1024) {
        if (a.V <= 2048) sampler_plain_fast_kernel<2><<<a.R, 256, 0, st>>>(a);
        else if (a.V <= 4096) sampler_plain_fast_kernel<4><<<a.R, 256, 0, st>>>(a);
        else sampler_plain_fast_kernel<8><<<a.R, 256, 0, st>>>(a);
        return hipGetLastError();
    }
    if (a.V >= 4096) { if (a.fast_math) sampler_kernel<1024, true><<<a.R, 1024, smem, st>>>(a, n2); else sampler_kernel<1024, false><<<a.R, 1024, smem, st>>>(a, n2); }
    else { if (a.fast_math) sampler_kernel<256, true><<<a.R, 256, smem, st>>>(a, n2); else sampler_kernel<256, false><<<a.R, 256, smem, st>>>(a, n2); }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// A9/A10/A13: codebook gather + PixelShuffle(2) + concat, NHWC output
// (quantizer.py:179-186, generator.py:316-318,361-364; sampling_hqmodel.py:119-120)
// ---------------------------------------------------------------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void quant_gather_kernel(QuantArgs a) {
    const int pix = blockIdx.x;                      // b * r * r + Y * r + X
    const int r = a.r, rt = r / 2, E = a.E;
    const int b = pix / (r * r), Y = (pix / r) % r, X = pix % r;
    long long ct = -1, cb = -1;
    if (a.code_t) ct = clamp_idx(a.code_t[((long long)b * rt + (Y >> 1)) * rt + (X >> 1)], a.n_embed);
    if (a.code_b) {
        if (a.seq_layout) cb = a.code_b[(((long long)b * rt + (Y >> 1)) * rt + (X >> 1)) * 4 + (Y & 1) * 2 + (X & 1)];
        else cb = a.code_b[((long long)b * r + Y) * r + X];
        cb = clamp_idx(cb, a.n_embed);
    }
    TO* out = reinterpret_cast<TO*>(a.quant) + (long long)pix * 2 * E;
    const int sub = (Y & 1) * 2 + (X & 1);
    for (int c = threadIdx.x; c < 2 * E; c += blockDim.x) {
        float v = 0.0f;
        if (c < E) { if (ct >= 0) v = a.emb_t[ct * 4 * E + 4 * c + sub]; }        // out[c, 2h+i, 2w+j] = in[4c+2i+j, h, w]
        else if (cb >= 0) v = a.emb_b[cb * E + (c - E)];
        st1<TO>(out + c, v);
    }
}
hipError_t launch_quant_gather(const QuantArgs& a, hipStream_t st) {
    const int grid = a.B * a.r * a.r;
    if (a.out_dtype == DT_BF16) quant_gather_kernel<bf16_t><<<grid, 256, 0, st>>>(a);
    else quant_gather_kernel<float><<<grid, 256, 0, st>>>(a);
    return hipGetLastError();
}

template <typename TO>
__global__ __launch_bounds__(256) void quant_gather3_kernel(QuantArgs3 a) {
    const int pix = blockIdx.x;                      // b * r * r + Y * r + X
    const int r = a.r, rm = r / 2, rt = r / 4, E = a.E;
    const int b = pix / (r * r), Y = (pix / r) % r, X = pix % r;
    long long c0 = -1, c1 = -1, c2 = -1;
    if (a.code_t) c0 = a.code_t[((long long)b * rt + (Y >> 2)) * rt + (X >> 2)];      // same index in both layouts
    if (a.code_m) {
        if (a.seq_layout) c1 = a.code_m[(((long long)b * rt + (Y >> 2)) * rt + (X >> 2)) * 4 + ((Y >> 1) & 1) * 2 + ((X >> 1) & 1)];
        else c1 = a.code_m[((long long)b * rm + (Y >> 1)) * rm + (X >> 1)];
    }
    if (a.code_b) {
        if (a.seq_layout) c2 = a.code_b[(((long long)b * rt + (Y >> 2)) * rt + (X >> 2)) * 16 + (Y & 3) * 4 + (X & 3)];
        else c2 = a.code_b[((long long)b * r + Y) * r + X];
    }
    if (c0 >= 0) c0 = clamp_idx(c0, a.n_embed);
    if (c1 >= 0) c1 = clamp_idx(c1, a.n_embed);
    if (c2 >= 0) c2 = clamp_idx(c2, a.n_embed);
    TO* out = reinterpret_cast<TO*>(a.quant) + (long long)pix * E;
    const int sub = (Y & 1) * 2 + (X & 1), subm = ((Y >> 1) & 1) * 2 + ((X >> 1) & 1);
    for (int c = threadIdx.x; c < E; c += blockDim.x) {
        float v = 0.0f;                                  // reference order: (PS(PS(q0) + q1)) + q2
        if (c0 >= 0) v = a.emb0[c0 * 16 * E + 4 * (4 * c + sub) + subm];
        if (c1 >= 0) v += a.emb1[c1 * 4 * E + 4 * c + sub];
        if (c2 >= 0) v += a.emb2[c2 * E + c];
        st1<TO>(out + c, v);
    }
}
hipError_t launch_quant_gather3(const QuantArgs3& a, hipStream_t st) {
    const int grid = a.B * a.r * a.r;
    if (a.out_dtype == DT_BF16) quant_gather3_kernel<bf16_t><<<grid, 256, 0, st>>>(a);
    else quant_gather3_kernel<float><<<grid, 256, 0, st>>>(a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// GroupNorm statistics (32 groups, eps 1e-6; stage1/modules/layers.py:17-21), NHWC input.
// One workgroup per (sample, group); sums in double.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* x, float* stats, int HW, int C, int groups, float eps) {
    __shared__ double red[4];
    const int b = blockIdx.x / groups, g = blockIdx.x % groups;
    const int cpg = C / groups;
    const T* base = x + (long long)b * HW * C + g * cpg;
    const long long n = (long long)HW * cpg;
    double s = 0.0;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) s += (double)ld1<T>(base + (i / cpg) * C + (i % cpg));
    const double mean = block_reduce(s, OpAdd(), red) / (double)n;
    double v = 0.0;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) {
        const double t = (double)ld1<T>(base + (i / cpg) * C + (i % cpg)) - mean;
        v += t * t;
    }
    const double var = block_reduce(v, OpAdd(), red) / (double)n;
    if (threadIdx.x == 0) {
        stats[(long long)blockIdx.x * 2] = (float)mean;
        stats[(long long)blockIdx.x * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}
hipError_t launch_gn_stats(const void* x, int dtype, float* stats, int B, int HW, int C, int groups, float eps,
                           hipStream_t st) {
    if (dtype == DT_BF16) gn_stats_kernel<bf16_t><<<B * groups, 256, 0, st>>>((const bf16_t*)x, stats, HW, C, groups, eps);
    else gn_stats_kernel<float><<<B * groups, 256, 0, st>>>((const float*)x, stats, HW, C, groups, eps);
    return hipGetLastError();
}

// FAST variants: one pass over contiguous pixel chunks (every load a full 16-B vector), per-channel fp32
// partials in registers, per-group double partials to memory, tiny finalize.  Fixed-order reductions throughout
// (no atomics): the decode is bit-reproducible from run to run and from lane to lane.
// pixels per workgroup: HW/32 clamped to [64, 1024] -> 4..64 chunks per image
static inline int gn_chunk_pix(int HW) { int c = HW / 32; return c < 64 ? 64 : (c > 1024 ? 1024 : c); }
static inline bool gn_fixed_ok(int C) { return C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0; }
__device__ __forceinline__ void unpack8(const uint4& raw, float (&f)[8]) {
    const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = bf16_to_f32((bf16_t)(w[i] & 0xffffu)); f[2 * i + 1] = bf16_to_f32((bf16_t)(w[i] >> 16)); }
}
// thread t owns channels 8 (t % vpp) .. + 7 of pixels (t / vpp) + k rpi (vpp = C / 8 vectors per pixel, rpi = 256 / vpp)
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* x, double* partial, int HW, int C, int groups, int nchunk, int chunk_pix) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* part = reinterpret_cast<float*>(smem_raw);      // [rpi][C] sums, then [rpi][C] sums of squares; later [C] + [C] totals
    const int b = blockIdx.x / nchunk, ch = blockIdx.x % nchunk;
    const int p0 = ch * chunk_pix, p1 = min(HW, p0 + chunk_pix);
    const int vpp = C / 8, rpi = 256 / vpp;
    const int cv = threadIdx.x % vpp, pl = threadIdx.x / vpp;
    float s[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] = 0.0f; q[i] = 0.0f; }
    const T* base = x + ((long long)b * HW) * C + cv * 8;
    int p = p0 + pl;
    for (; p + 3 * rpi < p1; p += 4 * rpi) {                // four independent pixel loads in flight
        float f[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) ld8<T>(base + (long long)(p + u * rpi) * C, f[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { s[i] += f[u][i]; q[i] += f[u][i] * f[u][i]; }
        }
    }
    for (; p < p1; p += rpi) {
        float f[8];
        ld8<T>(base + (long long)p * C, f);
#pragma unroll
        for (int i = 0; i < 8; ++i) { s[i] += f[i]; q[i] += f[i] * f[i]; }
    }
    float* ps = part + (size_t)pl * C + cv * 8;
    float* pq = part + (size_t)rpi * C + (size_t)pl * C + cv * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ps[i] = s[i]; pq[i] = q[i]; }
    __syncthreads();
    float ts = 0.0f, tq = 0.0f;                             // C <= 2048 channels over 256 threads: loop, fixed order over the pixel lanes
    for (int c = threadIdx.x; c < C; c += 256) {
        ts = 0.0f; tq = 0.0f;
        for (int r = 0; r < rpi; ++r) { ts += part[(size_t)r * C + c]; tq += part[(size_t)(rpi + r) * C + c]; }
        part[(size_t)2 * rpi * C + c] = ts;
        part[(size_t)2 * rpi * C + C + c] = tq;
    }
    __syncthreads();
    const float* tot = part + (size_t)2 * rpi * C;
    const int cpg = C / groups;
    for (int g = threadIdx.x; g < groups; g += 256) {
        double a = 0.0, c2 = 0.0;
        for (int i = 0; i < cpg; ++i) { a += (double)tot[g * cpg + i]; c2 += (double)tot[C + g * cpg + i]; }
        partial[((long long)blockIdx.x * groups + g) * 2] = a;
        partial[((long long)blockIdx.x * groups + g) * 2 + 1] = c2;
    }
}
// channel counts the fixed mapping does not cover (C / 8 does not divide 256): one thread per 16-B vector, LDS float atomics
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_generic_kernel(const T* x, double* partial, int HW, int C, int groups, int nchunk, int chunk_pix) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* ssum = reinterpret_cast<float*>(smem_raw);      // [C] sum, [C] sumsq
    float* ssq = ssum + C;
    const int b = blockIdx.x / nchunk, ch = blockIdx.x % nchunk;
    const int p0 = ch * chunk_pix, p1 = min(HW, p0 + chunk_pix);
    for (int c = threadIdx.x; c < 2 * C; c += 256) ssum[c] = 0.0f;
    __syncthreads();
    const int vec_per_pix = C / 8;
    const int total = (p1 - p0) * vec_per_pix;
    const T* base = x + ((long long)b * HW + p0) * C;
    for (int v = threadIdx.x; v < total; v += 256) {
        float f[8];
        ld8<T>(base + (long long)v * 8, f);
        const int c8 = (v % vec_per_pix) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) { atomicAdd(&ssum[c8 + i], f[i]); atomicAdd(&ssq[c8 + i], f[i] * f[i]); }
    }
    __syncthreads();
    const int cpg = C / groups;
    for (int g = threadIdx.x; g < groups; g += 256) {
        double a = 0.0, c2 = 0.0;
        for (int i = 0; i < cpg; ++i) { a += (double)ssum[g * cpg + i]; c2 += (double)ssq[g * cpg + i]; }
        partial[((long long)blockIdx.x * groups + g) * 2] = a;
        partial[((long long)blockIdx.x * groups + g) * 2 + 1] = c2;
    }
}
__global__ void gn_finalize_kernel(const double* partial, float* stats, int nchunk, int groups, double count, float eps, int n) {
    const int bg = blockIdx.x * blockDim.x + threadIdx.x;     // b * groups + g
    if (bg >= n) return;
    const int b = bg / groups, g = bg % groups;
    double a = 0.0, q = 0.0;
    for (int ch = 0; ch < nchunk; ++ch) {
        a += partial[(((long long)b * nchunk + ch) * groups + g) * 2];
        q += partial[(((long long)b * nchunk + ch) * groups + g) * 2 + 1];
    }
    const double mean = a / count;
    const double var = fmax(q / count - mean * mean, 0.0);
    stats[(long long)bg * 2] = (float)mean;
    stats[(long long)bg * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}
// (mean, rstd) from the per-tile partials a halo conv left behind: one wave per (image, group); lane l sums tiles
// l, l + 64, ... in order and the lanes are combined by the fixed xor tree -> bit-reproducible, double accumulation
template <typename TP>
__global__ __launch_bounds__(64) void gn_finalize_tiles_kernel(const TP* partial, float* stats, int tiles, int groups, double count, float eps) {
    const int bg = blockIdx.x;                                  // b * groups + g
    const int b = bg / groups, g = bg % groups;
    double a = 0.0, q = 0.0;
    for (int t = threadIdx.x; t < tiles; t += 64) {
        const TP* p = partial + (((long long)b * tiles + t) * groups + g) * 2;
        a += (double)p[0];
        q += (double)p[1];
    }
    a = wave_reduce(a, OpAdd());
    q = wave_reduce(q, OpAdd());
    if (threadIdx.x == 0) {
        const double mean = a / count;
        const double var = fmax(q / count - mean * mean, 0.0);
        stats[(long long)bg * 2] = (float)mean;
        stats[(long long)bg * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}
hipError_t launch_gn_finalize_tiles(const float* partial, float* stats, int B, int tiles, int HW, int C, int groups, float eps, hipStream_t st) {
    gn_finalize_tiles_kernel<float><<<B * groups, 64, 0, st>>>(partial, stats, tiles, groups, (double)HW * (C / groups), eps);
    return hipGetLastError();
}
hipError_t launch_gn_finalize_tiles_d(const double* partial, float* stats, int B, int tiles, int HW, int C, int groups, float eps, hipStream_t st) {
    gn_finalize_tiles_kernel<double><<<B * groups, 64, 0, st>>>(partial, stats, tiles, groups, (double)HW * (C / groups), eps);
    return hipGetLastError();
}
size_t gn_stats_fast_partial_elems(int B, int HW, int C, int groups) {
    (void)C;
    const int cp = gn_chunk_pix(HW);
    return (size_t)B * ((HW + cp - 1) / cp) * groups * 2;
}
hipError_t launch_gn_stats_fast(const void* x, float* stats, double* partial, int B, int HW, int C, int groups, float eps,
                                hipStream_t st, int dtype) {
    const int cp = gn_chunk_pix(HW);
    const int nchunk = (HW + cp - 1) / cp;
    if (gn_fixed_ok(C)) {
        const int rpi = 256 / (C / 8);
        const size_t smem = (size_t)(2 * rpi + 2) * C * sizeof(float);
        if (dtype == DT_F32) gn_partial_kernel<float><<<B * nchunk, 256, smem, st>>>((const float*)x, partial, HW, C, groups, nchunk, cp);
        else gn_partial_kernel<bf16_t><<<B * nchunk, 256, smem, st>>>((const bf16_t*)x, partial, HW, C, groups, nchunk, cp);
    } else {
        if (dtype == DT_F32) gn_partial_generic_kernel<float><<<B * nchunk, 256, 2 * C * sizeof(float), st>>>((const float*)x, partial, HW, C, groups, nchunk, cp);
        else gn_partial_generic_kernel<bf16_t><<<B * nchunk, 256, 2 * C * sizeof(float), st>>>((const bf16_t*)x, partial, HW, C, groups, nchunk, cp);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int n = B * groups;
    gn_finalize_kernel<<<(n + 63) / 64, 64, 0, st>>>(partial, stats, nchunk, groups, (double)HW * (C / groups), eps, n);
    return hipGetLastError();
}

// y = swish?(gamma (x - mean) rstd + beta): a thread keeps scale/shift of its 8 channels in registers and streams
// pixels with four 16-B loads in flight (same thread <-> channel mapping as gn_partial_kernel)
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* x, bf16_t* y, const float* stats, const float* gamma,
                                                       const float* beta, int HW, int C, int groups, int swish, int nchunk, int chunk_pix) {
    const int b = blockIdx.x / nchunk, ch = blockIdx.x % nchunk;
    const int p0 = ch * chunk_pix, p1 = min(HW, p0 + chunk_pix);
    const int vpp = C / 8, rpi = 256 / vpp, cpg = C / groups;
    const int cv = threadIdx.x % vpp, pl = threadIdx.x / vpp;
    float sc[8], sh[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = cv * 8 + i;
        const float* st = stats + ((long long)b * groups + c / cpg) * 2;
        sc[i] = st[1] * gamma[c];
        sh[i] = beta[c] - st[0] * sc[i];
    }
    const long long off = ((long long)b * HW) * C + cv * 8;
    auto norm_store = [&](const uint4& raw, long long pix) {
        float f[8];
        unpack8(raw, f);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float t = fmaf(f[i], sc[i], sh[i]);
            if (swish) t = t * __builtin_amdgcn_rcpf(1.0f + __expf(-t));
            f[i] = t;
        }
        uint4 o;
        o.x = (unsigned)f32_to_bf16(f[0]) | ((unsigned)f32_to_bf16(f[1]) << 16);
        o.y = (unsigned)f32_to_bf16(f[2]) | ((unsigned)f32_to_bf16(f[3]) << 16);
        o.z = (unsigned)f32_to_bf16(f[4]) | ((unsigned)f32_to_bf16(f[5]) << 16);
        o.w = (unsigned)f32_to_bf16(f[6]) | ((unsigned)f32_to_bf16(f[7]) << 16);
        *reinterpret_cast<uint4*>(y + off + pix * C) = o;
    };
    int p = p0 + pl;
    for (; p + 3 * rpi < p1; p += 4 * rpi) {
        uint4 r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = *reinterpret_cast<const uint4*>(x + off + (long long)(p + u * rpi) * C);
#pragma unroll
        for (int u = 0; u < 4; ++u) norm_store(r[u], p + u * rpi);
    }
    for (; p < p1; p += rpi) norm_store(*reinterpret_cast<const uint4*>(x + off + (long long)p * C), p);
}
__global__ __launch_bounds__(256) void gn_apply_generic_kernel(const bf16_t* x, bf16_t* y, const float* stats, const float* gamma,
                                                               const float* beta, long long total_vec, int HW, int C, int groups, int swish) {
    const int cpg = C / groups, vec_per_pix = C / 8;
    for (long long v = blockIdx.x * (long long)blockDim.x + threadIdx.x; v < total_vec; v += (long long)gridDim.x * blockDim.x) {
        const long long pix = v / vec_per_pix;
        const int c8 = (int)(v - pix * vec_per_pix) * 8;
        const int b = (int)(pix / HW);
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(x + v * 8), f);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float* st = stats + ((long long)b * groups + (c8 + i) / cpg) * 2;
            const float scl = st[1] * gamma[c8 + i];
            float t = fmaf(f[i], scl, beta[c8 + i] - st[0] * scl);
            if (swish) t = t * __builtin_amdgcn_rcpf(1.0f + __expf(-t));
            f[i] = t;
        }
        uint4 o;
        o.x = (unsigned)f32_to_bf16(f[0]) | ((unsigned)f32_to_bf16(f[1]) << 16);
        o.y = (unsigned)f32_to_bf16(f[2]) | ((unsigned)f32_to_bf16(f[3]) << 16);
        o.z = (unsigned)f32_to_bf16(f[4]) | ((unsigned)f32_to_bf16(f[5]) << 16);
        o.w = (unsigned)f32_to_bf16(f[6]) | ((unsigned)f32_to_bf16(f[7]) << 16);
        *reinterpret_cast<uint4*>(y + v * 8) = o;
    }
}
hipError_t launch_gn_apply(const void* x, void* y, const float* stats, const float* gamma, const float* beta, int B, int HW,
                           int C, int groups, int swish, hipStream_t st) {
    if (gn_fixed_ok(C)) {
        const int cp = gn_chunk_pix(HW);
        const int nchunk = (HW + cp - 1) / cp;
        gn_apply_kernel<<<B * nchunk, 256, 0, st>>>((const bf16_t*)x, (bf16_t*)y, stats, gamma, beta, HW, C, groups, swish, nchunk, cp);
        return hipGetLastError();
    }
    const long long total_vec = (long long)B * HW * C / 8;
    const int grid = (int)std::min<long long>((total_vec + 255) / 256, 256 * 16);
    gn_apply_generic_kernel<<<grid, 256, 0, st>>>((const bf16_t*)x, (bf16_t*)y, stats, gamma, beta, total_vec, HW, C, groups, swish);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// row softmax (decoder AttnBlock, stage1/modules/layers.py:177), in place
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(T* x, int n) {
    __shared__ float red[4];
    T* row = x + (long long)blockIdx.x * n;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, ld1<T>(row + i));
    m = block_reduce(m, OpMax(), red);
    float s = 0.0f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += expf(ld1<T>(row + i) - m);
    s = block_reduce(s, OpAdd(), red);
    for (int i = threadIdx.x; i < n; i += blockDim.x) st1<T>(row + i, expf(ld1<T>(row + i) - m) / s);
}
hipError_t launch_softmax_rows(void* x, int dtype, int rows, int n, hipStream_t st) {
    if (dtype == DT_BF16) softmax_rows_kernel<bf16_t><<<rows, 256, 0, st>>>((bf16_t*)x, n);
    else softmax_rows_kernel<float><<<rows, 256, 0, st>>>((float*)x, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// HQ-VAE encode side: image layout, nearest-code search bookkeeping (the distance GEMM itself is STORE_ARGMIN)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void image_to_nhwc_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int R, int cpad) {
    const long long npix = (long long)B * R * R;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
        const long long b = p / ((long long)R * R), pix = p - b * R * R;
        for (int c = 0; c < cpad; ++c) {
            const float v = c < 3 ? img[(b * 3 + c) * R * R + pix] : 0.0f;
            st1<T>(out + p * cpad + c, v);
        }
    }
}
hipError_t launch_image_to_nhwc(const float* img, void* out, int out_dtype, int B, int R, int cpad, hipStream_t st) {
    const long long npix = (long long)B * R * R;
    const int blocks = (int)std::min<long long>((npix + 255) / 256, 4096);
    if (out_dtype == DT_BF16) image_to_nhwc_kernel<bf16_t><<<blocks, 256, 0, st>>>(img, (bf16_t*)out, B, R, cpad);
    else image_to_nhwc_kernel<float><<<blocks, 256, 0, st>>>(img, (float*)out, B, R, cpad);
    return hipGetLastError();
}

// fixed-order sum over the 256 threads of a workgroup (same result on every run)
__device__ inline float block_sum_256(float v, float* sh) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[wave] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

template <typename T>
__global__ __launch_bounds__(256) void row_sumsq_kernel(const T* __restrict__ rows, float* __restrict__ out, int M, int K) {
    __shared__ float sh[4];
    const int m = blockIdx.x;
    float acc = 0.0f;
    for (int k = threadIdx.x; k < K; k += 256) { const float v = ld1<T>(rows + (long long)m * K + k); acc += v * v; }
    acc = block_sum_256(acc, sh);
    if (threadIdx.x == 0) out[m] = acc;
}
hipError_t launch_row_sumsq(const void* rows, int dtype, float* out, int M, int K, hipStream_t st) {
    if (dtype == DT_BF16) row_sumsq_kernel<bf16_t><<<M, 256, 0, st>>>((const bf16_t*)rows, out, M, K);
    else row_sumsq_kernel<float><<<M, 256, 0, st>>>((const float*)rows, out, M, K);
    return hipGetLastError();
}

// (level row m = (b, y, x), level column cq)  ->  offset into the bottom-layout NHWC [B, r, r, E] tensors
__device__ __forceinline__ long long vq_bottom_off(int b, int y, int x, int cq, int k, int r, int E) {
    int c = cq;
    for (int t = 0; t < k; ++t) {           // PixelUnshuffle(2): channel c * 4 + 2 i + j of the coarse map = channel c at (2 y + i, 2 x + j)
        const int j = c & 1, i = (c >> 1) & 1;
        c >>= 2;
        y = 2 * y + i; x = 2 * x + j;
    }
    return (((long long)b * r + y) * r + x) * E + c;
}

template <typename T>
__global__ __launch_bounds__(256) void vq_rows_kernel(VqArgs a) {
    __shared__ float sh[4];
    const int rq = a.r >> a.k, dim = a.E << (2 * a.k);
    const int m = blockIdx.x;
    const int b = m / (rq * rq), pix = m - b * rq * rq, y = pix / rq, x = pix - y * rq;
    T* zrow = reinterpret_cast<T*>(a.z) + (long long)m * dim;
    float acc = 0.0f;
    for (int cq = threadIdx.x; cq < dim; cq += 256) {
        const long long o = vq_bottom_off(b, y, x, cq, a.k, a.r, a.E);
        float v = a.h[o];
        if (a.recon) v = v - a.recon[o];                    // generator.py:303 / 544: h - upsampled coarser quant
        if (a.resid_nchw) a.resid_nchw[((long long)b * dim + cq) * rq * rq + pix] = v;
        st1<T>(zrow + cq, v);
        const float vr = ld1<T>(zrow + cq);                  // the value the distance GEMM will read
        acc += vr * vr;
    }
    acc = block_sum_256(acc, sh);
    if (threadIdx.x == 0) a.zz[m] = acc;
}
hipError_t launch_vq_rows(const VqArgs& a, hipStream_t st) {
    const int rq = a.r >> a.k, M = a.B * rq * rq;
    if (a.z_dtype == DT_BF16) vq_rows_kernel<bf16_t><<<M, 256, 0, st>>>(a);
    else vq_rows_kernel<float><<<M, 256, 0, st>>>(a);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void vq_finish_kernel(VqArgs a) {
    __shared__ float sh[4];
    const int rq = a.r >> a.k, dim = a.E << (2 * a.k);
    const int m = blockIdx.x;
    const int b = m / (rq * rq), pix = m - b * rq * rq, y = pix / rq, x = pix - y * rq;
    const long long code = (long long)(a.best[m] & 0xffffffffull);
    if (threadIdx.x == 0) a.codes[m] = code;
    const float* e = a.emb + code * dim;
    float acc = 0.0f;
    for (int cq = threadIdx.x; cq < dim; cq += 256) {
        const long long o = vq_bottom_off(b, y, x, cq, a.k, a.r, a.E);
        const float rec = a.recon ? a.recon[o] : 0.0f;
        const float z = a.recon ? a.h[o] - rec : a.h[o];
        const float d = e[cq] - z;
        acc += d * d;                                        // quantizer.py:130 (before the straight-through form)
        const float q = z + d;                               // quantizer.py:131: z + (z_q - z).detach()
        if (a.quant_nchw) a.quant_nchw[((long long)b * dim + cq) * rq * rq + pix] = q;
        if (a.recon) a.recon[o] = q + rec;                   // generator.py:552: _quant + recons[-1], then pixel-shuffled (= this layout)
    }
    acc = block_sum_256(acc, sh);
    if (threadIdx.x == 0) a.err_rows[m] = acc;
}
hipError_t launch_vq_finish(const VqArgs& a, hipStream_t st) {
    const int rq = a.r >> a.k, M = a.B * rq * rq;
    vq_finish_kernel<<<M, 256, 0, st>>>(a);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void vq_diff_kernel(const float* __restrict__ err_rows, int M, float scale, float* __restrict__ diff) {
    __shared__ float sh[4];
    float acc = 0.0f;
    for (int m = threadIdx.x; m < M; m += 256) acc += err_rows[m];
    acc = block_sum_256(acc, sh);
    if (threadIdx.x == 0) diff[0] = acc * scale;
}
hipError_t launch_vq_diff(const float* err_rows, int M, float scale, float* diff, hipStream_t st) {
    vq_diff_kernel<<<1, 256, 0, st>>>(err_rows, M, scale, diff);
    return hipGetLastError();
}

__global__ void nhwc_to_nchw_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int hw, int C) {
    const long long n = (long long)B * hw * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / ((long long)hw * C), rem = i - b * hw * C;
        const int c = (int)(rem / hw), p = (int)(rem - (long long)c * hw);          // i indexes the NCHW output
        out[i] = in[(b * hw + p) * C + c];
    }
}
hipError_t launch_nhwc_to_nchw_f32(const float* in, float* out, int B, int hw, int C, hipStream_t st) {
    const long long n = (long long)B * hw * C;
    nhwc_to_nchw_f32_kernel<<<(int)std::min<long long>((n + 255) / 256, 4096), 256, 0, st>>>(in, out, B, hw, C);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// generic GEMM dispatch
// ---------------------------------------------------------------------------------------------
hipError_t launch_gemm_generic(const GemmArgs& g, int ta, int tb, int tc, hipStream_t st) {
    if (g.K % 16 != 0 || (g.conv_taps && g.Cin % 4 != 0)) return hipErrorInvalidValue;
    const dim3 grid((g.N + 63) / 64, (g.M + 63) / 64, g.batch > 0 ? g.batch : 1);
    const int key = ta * 4 + tb * 2 + tc;
    switch (key) {
        case 0:
            // plain fp32 nn.Linear (the AR loop): the four-quarter summation order exact_mfma_gemm_kernel shares (exact_gemm.hip)
            if (g.k_quarters && !g.conv_taps && !g.a_packed_mb && !g.a_rows_per_group && g.batch <= 1 && !g.gn_stats && g.K % 32 == 0 && (g.store == STORE_ROWS || g.store == STORE_QKV))
                gemm_tile_kernel<float, float, float, true><<<grid, 256, 0, st>>>(g);
            else gemm_tile_kernel<float, float, float><<<grid, 256, 0, st>>>(g);
            break;
        case 7: gemm_tile_kernel<bf16_t, bf16_t, bf16_t><<<grid, 256, 0, st>>>(g); break;
        case 6: gemm_tile_kernel<bf16_t, bf16_t, float><<<grid, 256, 0, st>>>(g); break;
        case 2: gemm_tile_kernel<float, bf16_t, float><<<grid, 256, 0, st>>>(g); break;
        case 3: gemm_tile_kernel<float, bf16_t, bf16_t><<<grid, 256, 0, st>>>(g); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
