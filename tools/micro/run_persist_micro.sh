#!/bin/bash
O=${O:-gpurun_out/p17}; mkdir -p $O
(
timeout 120 tools/micro/bench_persist check 128 4 2 512 3 5 1
timeout 120 tools/micro/bench_persist check 1536 24 2 0 64 31 1 | tail -4
for t in 0 31 63; do timeout 120 tools/micro/bench_persist time 1536 24 12 0 64 $t 1 | grep -v "^phase  *[1-9][0-9]* \|^phase  *[2-9] "; done
# depth-0 shaped program: 4 layers + head rows (V = 8192)
timeout 120 tools/micro/bench_persist time 1536 24 4 8192 64 0 1 | grep -v "^phase  *[1-9][0-9]* \|^phase  *[2-9] "
) > $O/log.txt 2>&1
tail -5 $O/log.txt
