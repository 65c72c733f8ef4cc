// compiled with -mavx512f; only called when the host reports AVX-512
#define GEMM_FN hqt_cpu_gemm_tile_avx512
#define VL 16
#define MR 4
#define NR 4
#include "gemm_impl.inc"
