#!/bin/bash
# Throughput of the default workload over pass sizes (--merge) and lanes (--inflight); one line per schedule.
#   gpurun --timeout 1800 -- 'bash tools/sweep_schedule.sh'
for cfg in "--merge 8 --inflight 3" "--merge 8 --inflight 2" "--merge 12 --inflight 3" "--merge 16 --inflight 2" "--merge 16 --inflight 3" "--merge 24 --inflight 2" "--merge 32 --inflight 2" "--merge 32 --inflight 3" "--merge 48 --inflight 2"; do
timeout 600 python bench.py --steps 96 --warmup 6 $cfg --no-cpu-baseline --no-roofline > gpurun_out/b.json 2>gpurun_out/b.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/b.json").read().strip().splitlines()[-1])
    print("$cfg ->", d["value"], "img/s", d["ms_per_step"], "ms/step; images in flight", d["config"]["images_in_flight_per_gpu"], "step latency ms", d["config"]["step_latency_ms"])
except Exception as e:
    print("$cfg FAILED", e); print(open("gpurun_out/b.err").read()[-1500:])
PY
done
