run() { python bench.py --no-cpu-baseline --no-roofline --steps 12 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['serial']['phase_ms'])"; }
HQT_GRAPH_POSITIONS=1 run G=1
HQT_GRAPH_POSITIONS=16 run G=16
HQT_GRAPH_POSITIONS=64 run G=64
HQT_GRAPH_POSITIONS=1 run G=1
HQT_GRAPH_POSITIONS=16 run G=16
HQT_GRAPH_POSITIONS=64 run G=64
