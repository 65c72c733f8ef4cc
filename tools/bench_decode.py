#!/usr/bin/env python3
"""Decode leg of the headline benchmark alone (configs/imagenet-12l.yaml decoder, batch 64, random codes), per precision:
wall time per batch, the per-kernel-class breakdown from libhqt's per-launch HIP-event timers, conv TFLOP/s.  GPU box only.

    python tools/bench_decode.py [--batch 64] [--iters 5] [--precision split|fast|exact ...]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hqtransformer_amd import synth  # noqa: E402
from hqtransformer_amd._lib import PRECISIONS  # noqa: E402
from hqtransformer_amd.config import load_config  # noqa: E402
from hqtransformer_amd.engine import Engine  # noqa: E402
from hqtransformer_amd.spec import stage1_spec_from_config  # noqa: E402
from tools.bench_decoder import decoder_flops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--precision', nargs='+', default=['split', 'fast'])
    ap.add_argument('--config', default=os.path.join(ROOT, 'configs', 'imagenet-12l.yaml'))
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    s1 = stage1_spec_from_config(load_config(a.config))
    eng = Engine(None, s1, dev, a.batch)
    eng.load(stage1=synth.stage1_weights(s1, 1, 'bench'))
    eng.finalize()
    r = np.random.default_rng(0)
    ct = torch.from_numpy(r.integers(0, s1.n_embed, (a.batch, s1.z_res // 2, s1.z_res // 2))).to(dev)
    cb = torch.from_numpy(r.integers(0, s1.n_embed, (a.batch, s1.z_res, s1.z_res))).to(dev)
    fl = decoder_flops(s1)
    ref = None
    for name in a.precision:
        prec = PRECISIONS[name]
        px = eng.decode(ct, cb, precision=prec, clamp01=True)
        torch.cuda.synchronize()
        again = eng.decode(ct, cb, precision=prec, clamp01=True)
        det = bool(torch.equal(px, again))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            eng.decode(ct, cb, precision=prec, clamp01=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / a.iters
        eng.timing(True)
        eng.timing_reset()
        eng.decode(ct, cb, precision=prec, clamp01=True)
        torch.cuda.synchronize()
        rep = eng.timing_report()
        eng.timing(False)
        out = {'precision': name, 'batch': a.batch, 'ms_per_batch': round(ms, 3), 'images_per_s': round(a.batch / ms * 1e3, 1),
               'flop_per_image': fl, 'tflops_algorithmic': round(fl * a.batch / ms / 1e9, 1), 'deterministic': det,
               'kernel_ms': {k: [n, round(t, 3)] for k, (n, t) in sorted(rep.items(), key=lambda kv: -kv[1][1])}}
        if ref is None:
            ref = px
        else:
            d = (px - ref).abs()
            out['vs_first'] = {'max': float(d.max()), 'mean': float(d.mean())}
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
