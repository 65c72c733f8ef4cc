#!/bin/bash
O=gpurun_out/p22; mkdir -p $O
(for v in "A=default" "HQT_X_TILE_MIN=256" "A=default" "HQT_X_TILE_MIN=256"; do echo -n "$v "; env $v timeout 300 python bench.py --merge 1 --inflight 1 --steps 12 --no-cpu-baseline --no-exact-mode --no-roofline | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['like_for_like']['phase_ms'])"; done) > $O/tile_min_ab.txt 2>&1
cat $O/tile_min_ab.txt
HQT_X_TILE_MIN=256 timeout 600 python -m pytest tests/test_gpu_persist.py tests/test_gpu_parity.py -q -x -k "persist or fast or full_benchmark" 2>&1 | tail -3
