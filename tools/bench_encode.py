#!/usr/bin/env python3
"""Time hqt_encode (image -> codes) and the encode -> decode round trip on the ImageNet stage-1 config, with the per-kernel
breakdown of libhqt's timing slots.  GPU box only.

    python tools/bench_encode.py [--batch 64] [--iters 5] [--precision fast|exact] [--config configs/imagenet-12l.yaml]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hqtransformer_amd import synth  # noqa: E402
from hqtransformer_amd._lib import PRECISION_EXACT, PRECISION_FAST  # noqa: E402
from hqtransformer_amd.config import load_config  # noqa: E402
from hqtransformer_amd.engine import Engine  # noqa: E402
from hqtransformer_amd.spec import stage1_spec_from_config  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--precision', default='fast')
    ap.add_argument('--config', default='configs/imagenet-12l.yaml')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    s1 = stage1_spec_from_config(load_config(a.config))
    eng = Engine(None, s1, dev, a.batch)
    eng.load(stage1=synth.stage1_weights(s1, 1, 'bench', encoder=True))
    eng.finalize()
    prec = PRECISION_FAST if a.precision == 'fast' else PRECISION_EXACT
    x = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, (a.batch, 3, s1.resolution, s1.resolution)).astype(np.float32)).to(dev)
    eng.encode(x, precision=prec)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        o = eng.encode(x, precision=prec)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / a.iters
    eng.timing(True)
    eng.timing_reset()
    o = eng.encode(x, precision=prec)
    torch.cuda.synchronize()
    rep = eng.timing_report()
    eng.timing(False)
    t0 = time.perf_counter()
    for _ in range(a.iters):
        o = eng.encode(x, precision=prec)
        eng.decode(o['codes'][0], o['codes'][1], precision=prec) if len(o['codes']) == 2 else eng.decode3(o['codes'], precision=prec)
    torch.cuda.synchronize()
    rt = (time.perf_counter() - t0) * 1e3 / a.iters
    print(json.dumps({'workload': f'hq-vae encode, batch {a.batch}, {a.precision}', 'encode_ms': round(ms, 3),
                      'images_per_s': round(a.batch / ms * 1e3, 1), 'roundtrip_ms': round(rt, 3),
                      'kernel_ms': {k: [n, round(t, 3)] for k, (n, t) in sorted(rep.items(), key=lambda kv: -kv[1][1])}}))


if __name__ == '__main__':
    main()
