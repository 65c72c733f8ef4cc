#!/bin/bash
O=gpurun_out/p21; mkdir -p $O
(for v in "A=default" "HQT_X_EXACT_NOXCD=1"; do echo "$v"; env $v timeout 400 python tools/ar_pass_time.py --rows 64 --precision exact --reps 1 --breakdown --by-rows; echo; done) > $O/exact_xcd_ab.txt 2>/dev/null
cat $O/exact_xcd_ab.txt | cut -c1-1500
timeout 600 python -m pytest tests/test_gpu_timed_schedule.py tests/test_gpu_dist.py -q -x 2>&1 | tail -3
