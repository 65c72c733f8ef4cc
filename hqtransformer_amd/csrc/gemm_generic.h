// Generic fp32-accumulate NT GEMM / implicit-GEMM convolution on the vector ALUs.
//
// This is the EXACT-precision workhorse (fp32 in, fp32 FMA chain over k, fp32 out) and the shape
// fallback of the FAST path for the few GEMMs that are too small or too odd for the MFMA kernels.
// 64x64 output tile per 256-thread workgroup, BK = 16, 4x4 micro-tile per lane.
#pragma once
#include "common.h"

template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }

template <typename T> __device__ __forceinline__ void ld4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void ld4<float>(const float* p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void ld4<bf16_t>(const bf16_t* p, float (&v)[4]) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = bf16_to_f32((bf16_t)(t.x & 0xffff)); v[1] = bf16_to_f32((bf16_t)(t.x >> 16));
    v[2] = bf16_to_f32((bf16_t)(t.y & 0xffff)); v[3] = bf16_to_f32((bf16_t)(t.y >> 16));
}
template <typename T> __device__ __forceinline__ void st1(T* p, float v);
template <> __device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_GELU_ERF) return v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
    if (act == ACT_GELU_SIGMOID) return v / (1.0f + __expf(-1.702f * v));
    return v;
}
// The same with the activation as a compile-time constant, and the dispatcher that turns the run-time field into one: hipcc keeps
// `apply_act(v, g.act)` as two or three scalar branches PER ELEMENT of an unrolled epilogue (64 values per lane on a 128 x 128 tile:
// in-kernel stamps put 14-19 k of a workgroup's cycles there, with or without an activation selected -- round 6, tile_gemm.hip).
// dispatch_act runs `f(integral_constant<int, ACT>)`: ONE wave-uniform branch in front of the whole epilogue, three copies of its code.
template <int ACT> __device__ __forceinline__ float apply_act_c(float v) {
    if (ACT == ACT_GELU_ERF) return v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
    if (ACT == ACT_GELU_SIGMOID) return v / (1.0f + __expf(-1.702f * v));
    return v;
}
template <int V> struct ActC { static constexpr int value = V; };
template <class F> __device__ __forceinline__ void dispatch_act(int act, F&& f) {
    if (act == ACT_GELU_ERF) f(ActC<ACT_GELU_ERF>{});
    else if (act == ACT_GELU_SIGMOID) f(ActC<ACT_GELU_SIGMOID>{});
    else f(ActC<ACT_NONE>{});
}

// Loads 4 consecutive k of A row `m` (already decoded) starting at k.
template <typename TA>
struct ALoader {
    const GemmArgs& g;
    const TA* base;
    bool ok;
    int img, y, x;         // conv decode
    const TA* row;         // plain
    __device__ ALoader(const GemmArgs& g_, int m, int bz) : g(g_) {
        base = reinterpret_cast<const TA*>(g.A) + (long long)bz * g.a_batch_stride;
        ok = m < g.M;
        img = y = x = 0;
        row = base;
        if (!ok) return;
        if (g.conv_taps) {
            const int hw = g.H * g.W;
            img = m / hw;
            const int r = m - img * hw;
            y = r / g.W;
            x = r - y * g.W;
        } else {
            int ar = m;
            if (g.a_rows_per_group > 0) ar = (m / g.a_rows_per_group) * g.a_group_stride + m % g.a_rows_per_group + g.a_row_offset;
            row = base + (long long)ar * g.lda;
        }
    }
    __device__ __forceinline__ void load(int k, float (&v)[4]) const {
        v[0] = v[1] = v[2] = v[3] = 0.0f;
        if (!ok) return;
        if (!g.conv_taps) { ld4<TA>(row + k, v); return; }
        const int tap = k / g.Cin;
        const int c = k - tap * g.Cin;
        const int ks = conv_ks(g.conv_taps), s = 1 + g.conv_stride2;
        const int pad = (ks > 1 && !g.conv_nopad) ? 1 : 0;
        const int t3 = tap / ks;
        const int iy = y * s + t3 - pad, ix = x * s + (tap - t3 * ks) - pad;
        const int Hv = g.H * s, Wv = g.W * s;                              // input size as the filter sees it (after the x2 upsample)
        if (iy < 0 || iy >= Hv || ix < 0 || ix >= Wv) return;            // zero padding of the activated tensor
        const int Hin = Hv >> g.upsample, Win = Wv >> g.upsample;
        const TA* p = base + (((long long)img * Hin + (iy >> g.upsample)) * Win + (ix >> g.upsample)) * g.Cin + c;
        ld4<TA>(p, v);
        if (g.gn_stats) {
            const int cpg = g.Cin / g.gn_groups;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* st = g.gn_stats + ((long long)img * g.gn_groups + (c + i) / cpg) * 2;
                float t = (v[i] - st[0]) * st[1] * g.gn_gamma[c + i] + g.gn_beta[c + i];
                if (g.gn_swish) t = t / (1.0f + expf(-t));
                v[i] = t;
            }
        }
    }
};

// four consecutive outputs at once: 16 B (fp32) or 8 B (bf16); p must be aligned accordingly
template <typename T> __device__ __forceinline__ void st4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void st4<float>(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, const float (&v)[4]) {
    uint2 pk;
    pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
    pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
    *reinterpret_cast<uint2*>(p) = pk;
}

template <typename TC>
__device__ __forceinline__ void gemm_store(const GemmArgs& g, int bz, int m, int n, float v) {
    v *= g.alpha;
    if (g.bias) v += g.bias[n];
    v = apply_act(v, g.act);
    if (g.store == STORE_NCHW) {
        const int img = m / g.rows_per_image, pix = m - img * g.rows_per_image;
        if (g.clamp01) v = fminf(fmaxf(0.5f * v + 0.5f, 0.0f), 1.0f);
        st1<TC>(reinterpret_cast<TC*>(g.C) + ((long long)img * g.N + n) * g.rows_per_image + pix, v);
        return;
    }
    if (g.store == STORE_PACKED) {
        st1<TC>(reinterpret_cast<TC*>(g.C) + packed_off(m, n, g.c_packed_mb), v);
        return;
    }
    int orow = m;
    if (g.rows_per_group > 0) {
        orow = (m / g.rows_per_group) * g.group_stride + m % g.rows_per_group + g.row_offset;
        if (g.row_offset_dev) orow += *g.row_offset_dev;
    }
    if (g.store == STORE_QKV) {
        const int part = n / g.qkv_D, nn = n - part * g.qkv_D, which = part + g.qkv_first;
        if (which == 0) st1<TC>(reinterpret_cast<TC*>(g.C) + (long long)m * g.ldc + nn, v);
        else st1<TC>(reinterpret_cast<TC*>(which == 1 ? g.C2 : g.C3) + (long long)orow * g.ldc + nn, v);
        return;
    }
    const long long idx = (long long)bz * g.c_batch_stride + (long long)orow * g.ldc + n;
    if (g.resid) v += ld1<TC>(reinterpret_cast<const TC*>(g.resid) + idx);
    st1<TC>(reinterpret_cast<TC*>(g.C) + idx, v);
}

template <typename TA, typename TB, typename TC, bool QUARTERS = false>
__global__ __launch_bounds__(256) void gemm_tile_kernel(GemmArgs g) {
    __shared__ float As[16][68];
    __shared__ float Bs[16][68];
    const int tid = threadIdx.x, bz = blockIdx.z;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int lrow = tid >> 2, lk = (tid & 3) * 4;
    const ALoader<TA> al(g, m0 + lrow, bz);
    const int bn = n0 + lrow;
    const bool b_ok = bn < g.N;
    const TB* brow = reinterpret_cast<const TB*>(g.Bw) + (long long)bz * g.b_batch_stride + (long long)(b_ok ? bn : 0) * g.ldb;
    const int tx = tid & 15, ty = tid >> 4;
    // Summation order of an output (shared, bit for bit, with exact_gemm.hip's fp32 matrix-instruction kernel, so that an output does not
    // depend on which of the two a row count selects): K is cut into steps of 32 and the steps into four contiguous quarters (boundaries
    // (steps x s) / 4); partial s chains its quarter ascending, each 16-wide chunk in the k order [0 4 8 12 | 1 5 9 13 | 2 6 10 14 | 3 7 11 15];
    // the result is (p0 + p1) + (p2 + p3).  (Four chains instead of one: the matrix-instruction kernel gives one to each wave of a workgroup.)
    float part[4][4][4];
#pragma unroll
    for (int sp = 0; sp < 4; ++sp)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[sp][i][j] = 0.0f;

    auto chunk = [&](int k0, float (&ps)[4][4]) {
        float a4[4], b4[4];
        al.load(k0 + lk, a4);
        if (b_ok) ld4<TB>(brow + k0 + lk, b4); else b4[0] = b4[1] = b4[2] = b4[3] = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { As[lk + i][lrow] = a4[i]; Bs[lk + i][lrow] = b4[i]; }
        __syncthreads();
#pragma unroll
        for (int ki = 0; ki < 16; ++ki) {
            const int kk = (ki & 3) * 4 + (ki >> 2);
            const float4 av = *reinterpret_cast<const float4*>(&As[kk][ty * 4]);
            const float4 bv = *reinterpret_cast<const float4*>(&Bs[kk][tx * 4]);
            const float a[4] = {av.x, av.y, av.z, av.w};
            const float b[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) ps[i][j] = fmaf(a[i], b[j], ps[i][j]);
        }
        __syncthreads();
    };
    // QUARTERS: the plain fp32 nn.Linear launches of the AR loop above 256 rows (below, exact_mfma_gemm_kernel computes the same chains).  Every other
    // use -- the decoder's convolutions and attention products, bf16 fallbacks -- keeps ONE chain per output (the four partial tiles cost 178 registers
    // and a third of the EXACT decoder's speed), with the same order inside a chunk.
    if (QUARTERS) {
        const int nst = g.K >> 5;
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {
            const int k_lo = 32 * ((nst * sp) >> 2), k_hi = sp == 3 ? g.K : 32 * ((nst * (sp + 1)) >> 2);
            for (int k0 = k_lo; k0 < k_hi; k0 += 16) chunk(k0, part[sp]);
        }
    } else {
        for (int k0 = 0; k0 < g.K; k0 += 16) chunk(k0, part[0]);
    }
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (part[0][i][j] + part[1][i][j]) + (part[2][i][j] + part[3][i][j]);
    if (g.store == STORE_ARGMIN) {                   // nearest code: the tile's best (distance, index) per row, one atomic per row
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + ty * 4 + i;
            unsigned long long best = ~0ull;
            if (m < g.M) {
                const float zz = g.am_rownorm[m];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + tx * 4 + j;
                    if (n < g.N) {
                        const float d = (zz + g.am_colnorm[n]) - 2.0f * acc[i][j];       // quantizer.py:99-101, same association
                        const unsigned long long key = ((unsigned long long)float_order_key(d) << 32) | (unsigned)n;
                        best = key < best ? key : best;
                    }
                }
            }
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) {      // the 16 lanes tx = 0..15 of one ty are consecutive lanes of a wave
                const unsigned long long o = __shfl_xor(best, off);
                best = o < best ? o : best;
            }
            if (tx == 0 && m < g.M) atomicMin(g.am_best + m, best);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n < g.N) gemm_store<TC>(g, bz, m, n, acc[i][j]);
        }
    }
}
