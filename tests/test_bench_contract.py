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
        assert abs(r['frac'] - r['achieved'] / r['peak']) < 2e-3
        assert r['traffic'] is None or r['traffic'] > 0
        assert r['launches'] > 0 and r['avg_launch_us'] > 0
    c = d['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0 and isinstance(c['sample'], str) and c['unit'] == d['unit']
    # the per-launch figure follows from its own parts: achieved = algorithmic work per launch / average launch duration
    r = d['roofline']
    if r['bound'] == 'mfma':
        per_launch = r['algorithmic_flops_per_launch'] * r['matrix_flops_per_algorithmic_flop']
        assert abs(per_launch / (r['avg_launch_us'] * 1e-6) / 1e12 - r['achieved']) / r['achieved'] < 0.01
