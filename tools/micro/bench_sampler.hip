// Micro-benchmark: the fused sampler at merged-pass row counts (V = 8192), plain / top-k / top-k + top-p.
#include "../../hqtransformer_amd/csrc/kernels.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const int V = 8192, B = 2048;
    float* logits; CK(hipMalloc(&logits, (size_t)4 * B * V * 4));
    {
        std::vector<float> h((size_t)4 * B * V);
        unsigned s = 99u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 9) - (1 << 22)) * (4.0f / (1 << 22)); }
        CK(hipMemcpy(logits, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    StepState* state; CK(hipMalloc(&state, sizeof(StepState))); CK(hipMemset(state, 0, sizeof(StepState)));
    RowKey* rows; CK(hipMalloc(&rows, B * sizeof(RowKey)));
    CK(launch_set_rows(rows, B, 1234, 0, st));
    int64_t* out; CK(hipMalloc(&out, (size_t)B * 64 * 4 * 8));
    for (int fm = 0; fm < 2; ++fm)
    for (int mode = 0; mode < 3; ++mode) {
        for (int slots : {1, 4})
        for (int B : {512, 2048}) {
            if (slots == 1 && B != 512) continue;
            SamplerArgs a{};
            a.fast_math = fm;
            a.logits = logits; a.R = B * slots; a.V = V; a.slots = slots; a.B = B; a.temperature = 1.0f;
            a.top_k = mode >= 1 ? 2048 : 0; a.top_p = mode >= 2 ? 1.0f : 0.0f; a.draw0 = slots == 1 ? 0 : 1;
            a.state = state; a.rows = rows; a.n_steps = 64; a.out = out; a.draws = 5;
            CK(sampler_configure(V, a.top_p > 0.0f));
            CK(launch_sampler(a, st)); CK(hipStreamSynchronize(st));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < 20; ++r) CK(launch_sampler(a, st));
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-22s %-10s rows %4d : %8.2f us per launch\n", mode == 0 ? "plain" : (mode == 1 ? "top_k 2048" : "top_k 2048 + top_p 1"), fm ? "fast-math" : "IEEE", a.R, 1000.f * ms / 20);
        }
    }
    return 0;
}
