// debug helper: run the stream kernel variants one by one on one shape, synchronising after each
#include "../../hqtransformer_amd/csrc/split_stream_conv.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int ABL> static void run(const GemmArgs& g, const char* what) {
    void (*k)(GemmArgs) = conv3x3_split_stream_kernel<false, 128, ABL>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS));
    printf("%s ...\n", what); fflush(stdout);
    k<<<dim3((g.N + 127) / 128, g.M / 128, 1), 256, R_LDS>>>(g);
    CK(hipDeviceSynchronize());
    printf("%s ok\n", what); fflush(stdout);
}
int main() {
    const int B = 2, res = 16, cin = 128, cout = 128;
    half_t *A, *Wfrag; float *C, *bias, *W32; void* zero;
    CK(hipMalloc(&A, (size_t)B * res * res * cin * 4)); CK(hipMalloc(&C, (size_t)B * res * res * cout * 4));
    CK(hipMalloc(&W32, (size_t)cout * 9 * cin * 4)); CK(hipMalloc(&Wfrag, split_frag_elems(cout, cin) * 2));
    CK(hipMalloc(&bias, cout * 4)); CK(hipMalloc(&zero, 256));
    CK(hipMemset(A, 0, (size_t)B * res * res * cin * 4)); CK(hipMemset(W32, 0, (size_t)cout * 9 * cin * 4)); CK(hipMemset(bias, 0, cout * 4)); CK(hipMemset(zero, 0, 256));
    CK(launch_pack_split_frag(W32, Wfrag, cout, cin, 0)); CK(hipDeviceSynchronize());
    GemmArgs g{};
    g.A = A; g.conv_taps = 9; g.H = res; g.W = res; g.Cin = cin; g.Bw_frag = Wfrag; g.ldb = 9 * cin; g.C = C; g.ldc = cout;
    g.M = B * res * res; g.N = cout; g.K = 9 * cin; g.batch = 1; g.bias = bias; g.alpha = 1.f; g.store = STORE_ROWS; g.zero_page = zero;
    run<4>(g, "ABL 4: no fragment reads / loads (DMA + MFMA + epilogue)");
    run<2>(g, "ABL 2: no patch DMA");
    run<1>(g, "ABL 1: no epilogue");
    run<0>(g, "full");
    return 0;
}
