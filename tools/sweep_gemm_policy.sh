# A/B sweep of the streaming-GEMM tile variants per shape class (HQT_GEMM_<class>="MBW,NT,NW,U") with the default 3 lanes.
run() { python bench.py --no-cpu-baseline --no-roofline --steps 12 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['serial']['phase_ms'])"; }
export HQT_GEMM_M256N=2,2,8,6
HQT_GEMM_M256W=4,1,4,4 HQT_GEMM_M64W=2,1,8,6 run N2286+W4144+M64W2186
HQT_GEMM_M256W=4,1,4,4 HQT_GEMM_M64W=2,1,8,6 HQT_GEMM_M64N=1,1,8,12 run same+M64N1,1,8,12
HQT_GEMM_M256W=2,2,8,6 HQT_GEMM_M64W=2,1,8,6 run N2286+W2286+M64W2186
HQT_GEMM_M256W=2,1,8,6 HQT_GEMM_M64W=2,1,8,6 run N2286+W2186+M64W2186
HQT_GEMM_M64W=2,1,8,6 run N2286+M64W2186
