#!/bin/bash
# Everything profiles/r02_* is made of, in one GPU-box call (outputs under gpurun_out/r02/; copy what is to be judged into profiles/).
#   gpurun --timeout 3000 -- 'bash tools/collect_round_evidence.sh'
set -u
O=gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PMCRUN="python3 bench.py --batch 512 --merge 1 --inflight 1 --positions 2 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-graph"
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
python tools/prof_summary.py $O/stats 60 > $O/kernel_stats_bench_default.txt
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bench_default.csv
rm -rf $O/stats
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- $PMCRUN > /dev/null 2> $O/pmc_$c.err
  python tools/pmc_summary.py $O/pmc_$c > $O/pmc_${c}_rows512_positions2.txt
  rm -rf $O/pmc_$c
done
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- $PMCRUN > /dev/null 2> $O/pmc_mfma.err
python tools/pmc_summary.py $O/pmc_mfma > $O/pmc_mfma_raw.txt
python tools/pmc_mfma.py $O/pmc_mfma_raw.txt $O/pmc_mfma_util_rows512_positions2.txt > /dev/null
rm -rf $O/pmc_mfma
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE_rows512_positions2.txt $O/pmc_WRITE_SIZE_rows512_positions2.txt $O/pmc_latest.json
timeout 600 python bench.py --sampler quality --no-cpu-baseline > $O/bench_quality_sampler.json 2>/dev/null
timeout 600 python bench.py --merge 1 --inflight 3 --no-cpu-baseline --no-roofline > $O/bench_merge1_lanes3.json 2>/dev/null
timeout 600 python bench.py --merge 4 --inflight 3 --no-cpu-baseline --no-roofline > $O/bench_merge4_lanes3.json 2>/dev/null
timeout 600 python bench.py --merge 1 --inflight 1 --steps 12 --no-cpu-baseline --no-roofline > $O/bench_serial.json 2>/dev/null
timeout 600 python bench.py --batch 256 --merge 1 --inflight 1 --steps 24 --no-cpu-baseline --no-roofline > $O/bench_batch256_inflight1.json 2>/dev/null
timeout 600 python bench.py --decode-precision fast --no-cpu-baseline --no-roofline > $O/bench_decode_fast.json 2>/dev/null
timeout 600 python bench.py --config configs/imagenet-12l-level3.yaml --steps 48 --no-cpu-baseline --no-roofline > $O/bench_level3.json 2>/dev/null
timeout 600 python bench.py --config configs/cc15m-12l-txt.yaml --steps 48 --no-cpu-baseline --no-roofline > $O/bench_text_cond.json 2>/dev/null
timeout 300 python tools/bench_decode.py --precision split > $O/decode_split_batch64.json 2>/dev/null
timeout 300 python tools/bench_decode.py --precision fast > $O/decode_fast_batch64.json 2>/dev/null
timeout 300 python tools/diag_overlap.py --rows 512 > $O/diag_overlap_rows512.json 2>/dev/null
timeout 200 tools/micro/bench_split > $O/micro_split_conv_variants.txt 2>&1
timeout 300 tools/micro/bench_stream > $O/micro_stream_gemm_variants.txt 2>&1
timeout 200 tools/micro/bench_attn > $O/micro_attention.txt 2>&1
ls -la $O
timeout 100 tools/micro/bench_sampler > $O/micro_sampler.txt 2>&1
timeout 200 tools/micro/bench_panel > $O/micro_panel_gemm_experiment.txt 2>&1
ls $O | wc -l
