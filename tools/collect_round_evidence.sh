#!/bin/bash
# Everything profiles/r06_* is made of, in one GPU-box call (outputs under gpurun_out/r06/; copy what is to be judged into profiles/).
#   gpurun --timeout 3300 -- 'bash tools/collect_round_evidence.sh'
set -u
R=r06
O=gpurun_out/$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# (the decoder runs 64-image chunks whatever the pass: its counters come from the 640-row run; the 2048-row run skips the decode -- 32 chunks of it would not fit a counter pass)
pmcrun() { echo "python3 bench.py --batch $1 --merge 1 --inflight 1 --positions 2 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-graph --no-exact-mode $([ $1 -gt 1024 ] && echo --skip-decode)"; }
# ---- the driver's own invocation FIRST (K = 20: 2 lanes x passes of 10 steps = 640 rows), plain and under the profiler: the EXACT command
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_steps20.json 2> $O/bench_driver_steps20.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/statsd -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_steps20_under_rocprof.json 2> $O/bench_driver_steps20_under_rocprof.err
python tools/prof_summary.py $O/statsd 70 > $O/kernel_stats_driver_steps20.txt
cp $(find $O/statsd -name "*kernel_stats.csv" | head -1) $O/kernel_stats_driver_steps20.csv
rm -rf $O/statsd
# ---- the default line (K = 96: 2 lanes x passes of 32 steps = 2048 rows), and the same kernels one lane at a time under the profiler
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python3 bench.py --inflight 1 --merge 32 --steps 64 --warmup 8 --no-cpu-baseline --no-exact-mode > $O/bench_one_lane_under_rocprof.json 2> $O/bench_one_lane_under_rocprof.err
python tools/prof_summary.py $O/stats1 60 > $O/kernel_stats_one_lane_merge32.txt
cp $(find $O/stats1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_one_lane_merge32.csv
rm -rf $O/stats1
# ---- counters: separate passes over a bounded run (2 positions + the decode) of the kernels AT THE ROWS OF THE TIMED PASSES: 64 (one step at a time: the persistent chain), 640 (driver) and 2048 (default)
rm -f $O/pmc_latest.json
for rows in 64 640 2048; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- $(pmcrun $rows) > /dev/null 2> $O/pmc_${c}_$rows.err
    python tools/pmc_summary.py $O/pmc_$c > $O/pmc_${c}_rows${rows}_positions2.txt
    rm -rf $O/pmc_$c
  done
  python tools/pmc_traffic.py $O/pmc_FETCH_SIZE_rows${rows}_positions2.txt $O/pmc_WRITE_SIZE_rows${rows}_positions2.txt $O/pmc_latest.json $rows
  timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- $(pmcrun $rows) > /dev/null 2> $O/pmc_mfma_$rows.err
  python tools/pmc_summary.py $O/pmc_mfma > $O/pmc_mfma_raw.txt
  python tools/pmc_mfma.py $O/pmc_mfma_raw.txt $O/pmc_mfma_util_rows${rows}_positions2.txt > /dev/null
  rm -rf $O/pmc_mfma $O/pmc_mfma_raw.txt
done
# ---- other schedules and configurations of the same build
timeout 600 python bench.py --merge 8 --inflight 3 --no-cpu-baseline --no-exact-mode > $O/bench_merge8_lanes3.json 2>/dev/null
timeout 600 python bench.py --merge 48 --inflight 2 --no-cpu-baseline --no-roofline --no-exact-mode > $O/bench_merge48_lanes2.json 2>/dev/null
timeout 600 python bench.py --merge 1 --inflight 3 --no-cpu-baseline --no-roofline --no-exact-mode > $O/bench_merge1_lanes3.json 2>/dev/null
timeout 600 python bench.py --merge 1 --inflight 1 --steps 12 --no-cpu-baseline --no-exact-mode > $O/bench_serial.json 2>/dev/null
timeout 600 python bench.py --sampler quality --no-cpu-baseline --no-exact-mode > $O/bench_quality_sampler.json 2>/dev/null
timeout 600 python bench.py --decode-precision fast --no-cpu-baseline --no-roofline --no-exact-mode > $O/bench_decode_fast.json 2>/dev/null
timeout 900 python bench.py --config configs/imagenet-12l-level3.yaml --steps 48 --no-exact-mode > $O/bench_level3.json 2>/dev/null
timeout 900 python bench.py --config configs/cc15m-12l-txt.yaml --steps 48 --no-exact-mode > $O/bench_text_cond.json 2>/dev/null
timeout 900 python bench.py --config configs/imagenet-12l-level3-top2mid2bot.yaml --steps 24 --no-cpu-baseline --no-exact-mode > $O/bench_level3_top2mid2bot.json 2>/dev/null
timeout 300 python tools/bench_decode.py --precision split > $O/decode_split_batch64.json 2>/dev/null
timeout 300 python tools/bench_decode.py --precision fast > $O/decode_fast_batch64.json 2>/dev/null
timeout 300 python tools/bench_decoder.py --precision split > $O/decoder_only_1024_split.json 2>/dev/null
timeout 300 python tools/bench_decoder.py --precision fast > $O/decoder_only_1024_fast.json 2>/dev/null
timeout 300 python tools/bench_encode.py > $O/encode_batch64.json 2>/dev/null
timeout 300 python tools/cpu_twin_probe.py > $O/cpu_twin_probe.txt 2>&1
timeout 300 python tools/diag_overlap.py --rows 512 > $O/diag_overlap_rows512.json 2>/dev/null
timeout 300 python tools/diag_overlap.py --rows 2048 --lanes 2 --passes 4 > $O/diag_overlap_rows2048.json 2>/dev/null
timeout 300 python tools/ar_pass_time.py --rows 64 512 640 1024 2048 3072 --policy 1 --breakdown > $O/ar_pass_time_by_rows.json 2>/dev/null
timeout 300 python tools/ar_pass_time.py --rows 640 --policy 1 --breakdown --by-rows > $O/ar_pass_time_rows640_per_gemm.json 2>/dev/null
for p in split exact; do timeout 400 python tools/ar_pass_time.py --rows 64 640 2048 --precision $p --reps 1 --breakdown > $O/ar_pass_time_$p.json 2>/dev/null; done
timeout 300 python tools/split_step_probe.py > $O/split_step_probe.json 2>/dev/null
# this round's A/B switches: the persistent AR chain (batch-64 like-for-like; the two class-conditional runs twice, then the text and three-level configs),
# K slices of the SPLIT AR GEMMs, conv_out in one fp32 kernel
(for v in "HQT_PERSIST=1" "HQT_PERSIST=0" "HQT_PERSIST=1" "HQT_PERSIST=0"; do echo -n "$v "; env $v timeout 300 python bench.py --merge 1 --inflight 1 --steps 12 --no-cpu-baseline --no-exact-mode --no-roofline | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['like_for_like']['phase_ms'])"; done
 for c in configs/cc15m-12l-txt.yaml configs/imagenet-12l-level3.yaml; do for v in "HQT_PERSIST=1" "HQT_PERSIST=0"; do echo -n "$c $v "; env $v timeout 400 python bench.py --config $c --merge 1 --inflight 1 --steps 8 --no-cpu-baseline --no-exact-mode --no-roofline | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['like_for_like']['phase_ms'])"; done; done) > $O/persist_ab.txt 2>/dev/null
timeout 300 python tools/diag_two_handles.py > $O/diag_two_handles.txt 2>&1
# the launch sequence of ONE batch-64 position under the profiler (graph off: one row per launch)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats64 -- python3 bench.py --merge 1 --inflight 1 --steps 4 --warmup 1 --no-cpu-baseline --no-exact-mode --no-roofline --no-graph --skip-decode > $O/bench_serial_under_rocprof.json 2> $O/bench_serial_under_rocprof.err
python tools/prof_summary.py $O/stats64 40 > $O/kernel_stats_serial_batch64_ar_only.txt
rm -rf $O/stats64
# ---- micro-benchmarks
timeout 300 tools/micro/bench_split > $O/micro_split_conv_variants.txt 2>&1
timeout 100 tools/micro/bench_split_gemm > $O/micro_split_gemm_1x1.txt 2>&1
(timeout 200 tools/micro/bench_tile check | grep -E '^check'; timeout 300 tools/micro/bench_tile) > $O/micro_tile_gemm.txt 2>&1
timeout 120 tools/micro/bench_fill > $O/micro_fill.txt 2>&1
timeout 200 tools/micro/bench_split 64 order > $O/conv_tile_order.txt 2>&1
timeout 300 tools/micro/bench_stream > $O/micro_stream_gemm.txt 2>&1
bash tools/micro/run_persist_micro.sh > /dev/null 2>&1; cp gpurun_out/p17/log.txt $O/micro_persist.txt
timeout 200 tools/micro/bench_attn > $O/micro_attention.txt 2>&1
timeout 100 tools/micro/bench_sampler > $O/micro_sampler.txt 2>&1
# ---- measured values behind every FAST gate
rm -f $O/fast_gates.txt
HQT_RECORD_GATES=$O/fast_gates.txt timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
ls -la $O | head -70
