#!/usr/bin/env python3
"""What the GPU box's host gives a process, and how the CPU twin (oracle/cpu/hqt_cpu.cpp) scales on it: one AR position of the benchmark
model at batch 64 per OpenMP team size, and the decode of 8 images.  (bench.py's cpu_baseline picks its team size from the same sweep.)"""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    from hqtransformer_amd import synth
    from hqtransformer_amd.config import load_config
    from hqtransformer_amd.spec import stage1_spec_from_config, stage2_spec_from_config
    from oracle import hqt_cpu
    out = {'cpu_count': os.cpu_count(), 'affinity': len(os.sched_getaffinity(0))}
    for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us', '/proc/loadavg'):
        try:
            out[f] = open(f).read().strip()
        except OSError:
            pass
    try:
        ls = subprocess.run(['lscpu'], capture_output=True, text=True).stdout
        out['lscpu'] = {l.split(':')[0].strip(): l.split(':', 1)[1].strip() for l in ls.splitlines() if l.split(':')[0].strip() in
                        ('Model name', 'Socket(s)', 'Core(s) per socket', 'Thread(s) per core', 'NUMA node(s)', 'CPU(s)')}
    except OSError:
        pass
    hqt_cpu.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = load_config(os.path.join(root, 'configs', 'imagenet-12l.yaml'))
    s2, s1 = stage2_spec_from_config(cfg), stage1_spec_from_config(cfg)
    w2 = synth.stage2_weights(s2, 0, 'bench')
    cond = synth.class_ids(0, 64, 1000)
    out['ar_s_per_position'] = {}
    for n in (4, 8, 16, 24, 32, 48, 64, 96, 128):
        if n > out['affinity']:
            break
        twin = hqt_cpu.CpuTwin(s2, None, w2, threads=n)
        twin.sample(cond, 64, 1, None, seed=1)
        twin.sample(cond, 64, 2, None, seed=2)
        out['ar_s_per_position'][n] = round(twin.last_seconds / 2, 4)
        twin.close()
        print(n, out['ar_s_per_position'][n], flush=True)
        if out['ar_s_per_position'][n] > 2.0 * min(out['ar_s_per_position'].values()):
            break
    best = min(out['ar_s_per_position'], key=out['ar_s_per_position'].get)
    w1 = synth.stage1_weights(s1, 1, 'bench')
    rng = np.random.default_rng(0)
    r = s1.z_res
    out['decode_s_per_image'] = {}
    for n in sorted({best, max(4, best // 2), min(out['affinity'], best * 2)}):
        tw1 = hqt_cpu.CpuTwin(None, s1, None, w1, threads=n)
        tw1.decode_code(rng.integers(0, s1.n_embed, (8, r // 2, r // 2)), rng.integers(0, s1.n_embed, (8, r, r)))
        out['decode_s_per_image'][n] = round(tw1.last_seconds / 8, 4)
        tw1.close()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
