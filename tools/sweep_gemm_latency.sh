run() { python bench.py --no-cpu-baseline --no-roofline --inflight 1 --steps 6 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['serial']['phase_ms'])"; }
run default
HQT_GEMM_M64W=1,1,8,12 run M64W=1,1,8,12
HQT_GEMM_M64W=2,1,8,6 run M64W=2,1,8,6
HQT_GEMM_M256W=2,1,8,12 run M256W=2,1,8,12
HQT_GEMM_M256W=4,1,4,4 run M256W=4,1,4,4
HQT_GEMM_M256N=2,1,8,6 run M256N=2,1,8,6
HQT_GEMM_M64N=2,1,8,12 run M64N=2,1,8,12
