"""The bench line's contract (keys, types, internal consistency), checked on the committed line of the last round -- what the driver
parses.  bench.py itself needs the GPU; this keeps the committed evidence and the contract from drifting apart."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def latest():
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_default.json')))
    assert files, 'no committed bench line under profiles/'
    return json.load(open(files[-1]))


def test_bench_line_has_the_contract_keys():
    d = latest()
    for k, t in (('metric', str), ('value', (int, float)), ('unit', str), ('n_gpus', int), ('steps', int), ('warmup', int),
                 ('ms_per_step', (int, float)), ('higher_is_better', bool), ('scaling', str), ('dtype', str), ('data', str), ('config', dict)):
        assert isinstance(d[k], t), k
    assert 'vs_baseline' in d and d['vs_baseline'] is None          # BASELINE.md holds no published number for this metric
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['data'] == 'synthetic' and d['n_gpus'] == 1
    assert 'workload' in d['config'] and 'model' not in d['config']
    # value = images of all ranks / time: batch 64 per step
    assert abs(d['value'] - 64 * 1000.0 / d['ms_per_step']) / d['value'] < 0.01


def test_roofline_and_cpu_baseline_objects():
    d = latest()
    for r in [d['roofline']] + list(d.get('roofline_other', [])):
        assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s')
        assert abs(r['frac'] - r['achieved'] / r['peak']) < 2e-3 and 0.0 < r['frac'] <= 1.0
        assert r['traffic'] is None or r['traffic'] > 0
        assert r['launches'] > 0 and r['avg_launch_us'] > 0
    c = d['cpu_baseline']
    assert c['kind'] in ('port', 'native-port', 'reference') and c['cores'] >= 1 and c['value'] > 0 and isinstance(c['sample'], str) and c['unit'] == d['unit']
    # the per-launch figure follows from its own parts: achieved = algorithmic work per launch / average launch duration
    r = d['roofline']
    if r['bound'] == 'mfma':
        # round 3 on: achieved / frac are ALGORITHMIC, the issued matrix work sits beside them (achieved_issued / frac_issued)
        issued = r['matrix_flops_per_algorithmic_flop'] if 'achieved_issued' not in r else 1
        per_launch = r['algorithmic_flops_per_launch'] * issued
        assert abs(per_launch / (r['avg_launch_us'] * 1e-6) / 1e12 - r['achieved']) / r['achieved'] < 0.01
        if 'achieved_issued' in r:
            assert abs(r['achieved_issued'] - r['achieved'] * r['matrix_flops_per_algorithmic_flop']) / r['achieved_issued'] < 0.01
            assert abs(r['frac_issued'] - r['achieved_issued'] / r['peak']) < 2e-3
    cfg = d['config']
    if 'rows_per_pass' in cfg:                                        # the schedule in numbers (round 3 on)
        assert cfg['rows_per_pass'] % cfg['per_gpu_batch'] == 0 and cfg['images_in_flight_per_gpu'] % cfg['rows_per_pass'] == 0 and cfg['step_latency_ms'] > 0


def test_default_schedule_rule():
    """bench.py picks lanes and steps per pass from K when neither is given; explicit flags are kept."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ds = mod.default_schedule
    assert ds(96) == (2, 32) and ds(20) == (2, 10) and ds(7) == (2, 4) and ds(1) == (2, 1) and ds(1000) == (2, 32)
    assert ds(96, merge=8) == (3, 8) and ds(96, inflight=1) == (1, 8) and ds(96, 48, 2) == (2, 48)
    assert ds(48, wide=True) == (3, 16) and ds(96, wide=True) == (3, 16) and ds(20, wide=True) == (3, 7) and ds(48, 8, None, True) == (3, 8)


def _run_bench(args, env_extra):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True, env=env, timeout=300)


def test_gpus_flag_never_degrades_to_fewer_devices():
    """`bench.py --gpus N` must either run N ranks or print no line: with fewer visible GPUs than N (this container has none) it refuses
    before touching a device, and a launcher that started a different number of ranks than the flag says is refused too."""
    import torch
    if torch.cuda.device_count() >= 8:
        return
    r = _run_bench(['--gpus', '8'], {})
    assert r.returncode != 0 and 'refusing' in r.stderr and '{"metric"' not in r.stdout
    r = _run_bench(['--gpus', '8'], {'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and 'WORLD_SIZE=2' in r.stderr and '{"metric"' not in r.stdout
    r = _run_bench(['--gpus', '1'], {'WORLD_SIZE': '4', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and '{"metric"' not in r.stdout


def test_traffic_is_keyed_by_the_rows_of_the_pass():
    """`roofline.traffic` of the AR GEMM family is attached only when the committed counter pass ran at the row count the record times."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod2', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    doc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_latest.json')))
    rows = sorted(int(r) for r in doc.get('by_rows', {}))
    assert rows, 'profiles/pmc_latest.json holds no by_rows entry'
    assert mod.pmc_traffic('stream_gemm', rows[0]) == doc['by_rows'][str(rows[0])]['stream_gemm']
    assert mod.pmc_traffic('stream_gemm', 7) is None
    assert mod.pmc_traffic('decoder_conv') == doc['decoder_conv']


def test_round4_records_of_the_committed_driver_line():
    """What VERDICT r03 asked the line to carry: the schedule in the workload string, the one-step-at-a-time record under its own name, the
    bit-exact arithmetic's throughput, the gather flag, and traffic only from counters collected at the rows of the timed pass."""
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_driver_steps20.json')))
    d = json.load(open(files[-1]))
    if 'like_for_like' not in d:                      # lines of earlier rounds
        return
    cfg = d['config']
    assert d['steps'] == 20 and cfg['rows_per_pass'] == cfg['steps_per_pass'] * cfg['per_gpu_batch']
    assert f"merged into {cfg['rows_per_pass']}-row passes" in cfg['workload'] and 'like_for_like' in cfg['workload']
    lfl = d['like_for_like']
    assert lfl['images_in_flight_per_gpu'] == cfg['per_gpu_batch'] and lfl['value'] == d['serial']['value'] and lfl['value'] < d['value']
    assert abs(lfl['ms_per_step'] - (lfl['phase_ms']['ar'] + lfl['phase_ms']['decode'])) / lfl['ms_per_step'] < 0.05
    assert d['bit_exact_codes'] is False and d['gather_ok'] is None            # FAST arithmetic timed; one GPU: no collective
    em = d['exact_mode']
    for mode in ('split', 'exact'):
        assert em[mode]['like_for_like']['steps'] >= 3 and em[mode]['like_for_like']['value'] > 0
        assert em[mode]['merged']['rows_per_pass'] % cfg['per_gpu_batch'] == 0
    assert em['split']['merged']['value'] > em['exact']['merged']['value']     # the matrix-core arithmetic is the faster bit-exact one in merged passes
    gemm = [r for r in [d['roofline']] + d['roofline_other'] if 'AR GEMM family' in r['kernel']][0]
    doc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_latest.json')))
    want = doc.get('by_rows', {}).get(str(gemm['rows_per_pass']), {}).get('stream_gemm')
    # attached only from a counter pass at these rows: None when the line predates the pass; otherwise the committed pass of its own
    # evidence run or the one before it (the line reads the file committed when it ran) -- the same family, within ten per cent (round 6 changed
    # the 640-row tile geometry: 59.1 -> 55.7 MB per launch)
    assert gemm['traffic'] is None or (want is not None and abs(gemm['traffic'] - want) / want < 0.10)
    assert d['cpu_baseline']['kind'] == 'native-port' and d['cpu_baseline']['cores'] >= 1
