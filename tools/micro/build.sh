#!/bin/bash
# Builds the micro-benchmarks next to their sources (they #include the kernels of hqtransformer_amd/csrc directly).
#   bash tools/micro/build.sh [name ...]        default: all
set -e
cd "$(dirname "$0")"
names=("$@")
[ ${#names[@]} -eq 0 ] && names=(bench_attn bench_pack bench_chain bench_conv bench_dispatch bench_hbm bench_lanes bench_overlap bench_panel bench_sampler bench_split bench_split_gemm bench_stream bench_tile bench_fill bench_persist bench_exact launch_overhead)
for n in "${names[@]}"; do
    echo "hipcc $n"
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -o "$n" "$n.hip"
done
