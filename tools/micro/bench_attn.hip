// Micro-benchmark: the decode-step attention kernel at merged-pass batch sizes, cold K/V (12 layers' caches in rotation, as the body
// runs them), by number of cached keys -- achieved bytes/s against the K/V bytes the launch must read.  Also a head-major cache layout
// ([b][head][t][hs]: one (sample, head)'s keys contiguous) through the same arithmetic, to price the layout.
#include "../../hqtransformer_amd/csrc/kernels.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// experimental copy of attention_kernel<bf16_t> with explicit cache strides (elements): sample, head, key
template <int PB>
__global__ __launch_bounds__(256) void attn_strided_kernel(AttnArgs a, long long sb, long long sh, long long st_) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gid = blockIdx.x * 4 + wave;
    if (gid >= a.B * a.n_heads) return;
    const int h = gid % a.n_heads, b = gid / a.n_heads;
    const int hs = a.head_dim, D = a.n_heads * hs;
    const int chunks = hs >> 3, rows_per_pass = 64 / chunks;
    const int c = lane % chunks, slot = lane / chunks;
    const bf16_t* q = reinterpret_cast<const bf16_t*>(a.q) + (long long)b * D + h * hs + c * 8;
    const bf16_t* kc = reinterpret_cast<const bf16_t*>(a.kcache) + b * sb + h * sh + c * 8;
    const bf16_t* vc = reinterpret_cast<const bf16_t*>(a.vcache) + b * sb + h * sh + c * 8;
    float qv[8];
    ld8<bf16_t>(q, qv);
    const int nkeys = a.t_base + 1;
    const float scale = 1.0f / sqrtf((float)hs);
    float run_max = -INFINITY, run_sum = 0.0f, acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
    for (int j0 = 0; j0 < nkeys; j0 += rows_per_pass * PB) {
        uint4 kbuf[PB], vbuf[PB];
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const long long j = min(j0 + p * rows_per_pass + slot, nkeys - 1);
            kbuf[p] = *reinterpret_cast<const uint4*>(kc + j * st_);
            vbuf[p] = *reinterpret_cast<const uint4*>(vc + j * st_);
        }
        __builtin_amdgcn_sched_barrier(0);
        float sc[PB], gmax = -INFINITY;
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const unsigned w[4] = {kbuf[p].x, kbuf[p].y, kbuf[p].z, kbuf[p].w};
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) { s = fmaf(qv[2 * i], bf16_to_f32((bf16_t)(w[i] & 0xffffu)) * scale, s); s = fmaf(qv[2 * i + 1], bf16_to_f32((bf16_t)(w[i] >> 16)) * scale, s); }
            for (int off = chunks >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            sc[p] = (j0 + p * rows_per_pass + slot < nkeys) ? s : -INFINITY;
            gmax = fmaxf(gmax, sc[p]);
        }
        for (int off = chunks; off < 64; off <<= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, off, 64));
        const float new_max = fmaxf(run_max, gmax), rescale = expf(run_max - new_max);
        float gsum = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] *= rescale;
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const float e = expf(sc[p] - new_max);
            gsum += e;
            const unsigned w[4] = {vbuf[p].x, vbuf[p].y, vbuf[p].z, vbuf[p].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc[2 * i] = fmaf(e, bf16_to_f32((bf16_t)(w[i] & 0xffffu)), acc[2 * i]); acc[2 * i + 1] = fmaf(e, bf16_to_f32((bf16_t)(w[i] >> 16)), acc[2 * i + 1]); }
        }
        for (int off = chunks; off < 64; off <<= 1) gsum += __shfl_xor(gsum, off, 64);
        run_sum = run_sum * rescale + gsum;
        run_max = new_max;
    }
    for (int off = chunks; off < 64; off <<= 1)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += __shfl_xor(acc[i], off, 64);
    const float inv = 1.0f / run_sum;
    if (slot == 0) {
        bf16_t* o = reinterpret_cast<bf16_t*>(a.out) + (long long)b * D + h * hs + c * 8;
        uint4 pk;
        pk.x = (unsigned)f32_to_bf16(acc[0] * inv) | ((unsigned)f32_to_bf16(acc[1] * inv) << 16);
        pk.y = (unsigned)f32_to_bf16(acc[2] * inv) | ((unsigned)f32_to_bf16(acc[3] * inv) << 16);
        pk.z = (unsigned)f32_to_bf16(acc[4] * inv) | ((unsigned)f32_to_bf16(acc[5] * inv) << 16);
        pk.w = (unsigned)f32_to_bf16(acc[6] * inv) | ((unsigned)f32_to_bf16(acc[7] * inv) << 16);
        *reinterpret_cast<uint4*>(o) = pk;
    }
}

template <typename F>
static float graph_time(hipStream_t st, int reps, F body) {
    hipGraph_t graph; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    body();
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, st));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(graph));
    return 1000.f * ms / reps;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const int D = 1536, nh = 24, hs = 64, Tmax = 64, L = 12;
    for (int B : {64, 256, 512, 2048}) {
        bf16_t *q, *o; CK(hipMalloc(&q, (size_t)B * D * 2)); CK(hipMalloc(&o, (size_t)B * D * 2)); CK(hipMemset(q, 0, (size_t)B * D * 2));
        std::vector<bf16_t*> kc(L), vc(L);
        const size_t bytes = (size_t)B * Tmax * D * 2;
        for (int l = 0; l < L; ++l) { CK(hipMalloc(&kc[l], bytes)); CK(hipMalloc(&vc[l], bytes)); CK(hipMemset(kc[l], 0, bytes)); CK(hipMemset(vc[l], 0, bytes)); }
        for (int keys : {1, 8, 16, 32, 64}) {
            const double mb = 2.0 * B * keys * D * 2 / 1e6;
            AttnArgs at{q, nullptr, nullptr, o, B, 1, nh, hs, Tmax, keys - 1, nullptr, 1, DT_BF16, 0, nullptr};
            const float t0 = graph_time(st, 5, [&] { for (int l = 0; l < L; ++l) { AttnArgs a = at; a.kcache = kc[l]; a.vcache = vc[l]; CK(launch_attention(a, st)); } }) / L;
            auto strided = [&](auto pb, long long sb, long long sh, long long stt) {
                constexpr int PB = decltype(pb)::value;
                return graph_time(st, 5, [&] { for (int l = 0; l < L; ++l) { AttnArgs a = at; a.kcache = kc[l]; a.vcache = vc[l];
                    attn_strided_kernel<PB><<<(B * nh + 3) / 4, 256, 0, st>>>(a, sb, sh, stt); } }) / L;
            };
            const float t1 = strided(std::integral_constant<int, 8>{}, (long long)Tmax * D, hs, D);                // product layout, copy of the kernel
            const float t2 = strided(std::integral_constant<int, 8>{}, (long long)nh * Tmax * hs, (long long)Tmax * hs, hs);   // head-major
            const float t3 = strided(std::integral_constant<int, 4>{}, (long long)nh * Tmax * hs, (long long)Tmax * hs, hs);   // head-major, 32 keys per trip
            printf("B=%3d keys=%2d (%.1f MB K+V): product %.2f us (%.2f TB/s) | copy %.2f | head-major %.2f us (%.2f TB/s) | head-major PB=4 %.2f\n",
                   B, keys, mb, t0, mb / t0 / 1e6 * 1e6 / 1e6 * 1e0, t1, t2, mb / t2, t3);
        }
        {   // depth sub-step 1: 4 queries per sample over a 5-key cache (1 key cached by sub-step 0), causal
            bf16_t *q4, *o4; CK(hipMalloc(&q4, (size_t)B * 4 * D * 2)); CK(hipMalloc(&o4, (size_t)B * 4 * D * 2)); CK(hipMemset(q4, 0, (size_t)B * 4 * D * 2));
            AttnArgs at{q4, nullptr, nullptr, o4, B, 4, nh, hs, 5, 1, nullptr, 1, DT_BF16, 0, nullptr};
            const float t0 = graph_time(st, 5, [&] { for (int l = 0; l < L; ++l) { AttnArgs a = at; a.kcache = kc[l]; a.vcache = vc[l]; CK(launch_attention(a, st)); } }) / L;
            // one head per wave (attention_fewq_kernel) beside the dispatch's choice (eight heads per wave from 256 samples: attention_fewq8_kernel)
            const float t1 = graph_time(st, 5, [&] { for (int l = 0; l < L; ++l) { AttnArgs a = at; a.kcache = kc[l]; a.vcache = vc[l];
                attention_fewq_kernel<bf16_t, 1><<<(B * nh + 3) / 4, 256, 0, st>>>(a); } }) / L;
            printf("B=%4d depth sub-step 1 (4 queries x <= 5 keys): dispatch %.2f us | one head per wave (%d waves) %.2f us\n", B, t0, B * nh, t1);
            CK(hipFree(q4)); CK(hipFree(o4));
        }
        for (int l = 0; l < L; ++l) { CK(hipFree(kc[l])); CK(hipFree(vc[l])); }
        CK(hipFree(q)); CK(hipFree(o));
    }
    return 0;
}
