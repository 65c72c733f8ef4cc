// compiled with -mavx2 -mfma (the baseline every x86-64 host of the last decade runs)
#define GEMM_FN hqt_cpu_gemm_tile_avx2
#define VL 8
#define MR 2
#define NR 4
#include "gemm_impl.inc"
