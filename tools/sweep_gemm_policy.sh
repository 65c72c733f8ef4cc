# A/B sweep of the streaming-GEMM tile variants per shape class (HQT_GEMM_<class>="MBW,NT,NW,U") with the default 3 lanes;
# the serial pass of the same run shows the one-lane effect (the overrides apply to both policies).  Same box, alternating.
run() { python bench.py --no-cpu-baseline --no-roofline --steps 12 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['serial']['phase_ms'])"; }
run new-default
HQT_GEMM_M64N=2,1,8,12 run old-narrow-2,1,8,12
run new-default
HQT_GEMM_M64N=2,1,8,12 run old-narrow-2,1,8,12
run new-default
HQT_GEMM_M64N=2,1,8,12 run old-narrow-2,1,8,12
