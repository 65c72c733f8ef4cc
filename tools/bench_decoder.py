#!/usr/bin/env python3
"""Decoder-only throughput (BASELINE.json configs[3]: the conv path under stress).  The reference has no 1024 x 1024 stage-1
config (its FFHQ model is 256 x 256), so SURVEY.md 8(d) defines one with the reference's own 5-level pattern:
Decoder(resolution=1024, ch=128, ch_mult=[1, 2, 4, 4, 4], num_res_blocks=2, attn_resolutions=[32], use_init_downsample=True,
z_channels=256) fed by random code grids top [B, 16, 16] / bottom [B, 32, 32]: 2.80 TFLOP per image.  GPU box only.

    python tools/bench_decoder.py [--batch 8] [--iters 3] [--precision split|fast|exact]
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
from hqtransformer_amd._lib import PRECISION_EXACT, PRECISION_FAST, PRECISION_SPLIT  # noqa: E402
from hqtransformer_amd.engine import Engine  # noqa: E402
from hqtransformer_amd.spec import Stage1Spec, decoder_plan  # noqa: E402


def decoder_flops(s1: Stage1Spec) -> float:
    """2 x MACs of every conv and of the attention products of Decoder.forward + post_quant_conv_b, per image."""
    f = 2.0 * s1.z_res ** 2 * (2 * s1.embed_dim) * s1.z_channels
    for l in decoder_plan(s1):
        hw = l.res * l.res
        if l.kind == 'conv3':
            f += 2.0 * hw * 9 * l.cin * l.cout
        elif l.kind == 'upconv':
            f += 2.0 * 4 * hw * 9 * l.cin * l.cout
        elif l.kind == 'res':
            f += 2.0 * hw * 9 * (l.cin * l.cout + l.cout * l.cout) + (2.0 * hw * l.cin * l.cout if l.cin != l.cout else 0.0)
        elif l.kind == 'attn':
            f += 2.0 * hw * l.cin * l.cin * 4 + 2.0 * 2 * hw * hw * l.cin
        elif l.kind == 'out':
            f += 2.0 * hw * 9 * l.cin * l.cout
    return f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--iters', type=int, default=3)
    ap.add_argument('--precision', default='fast')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    s1 = Stage1Spec(ch=128, ch_mult=[1, 2, 4, 4, 4], num_res_blocks=2, attn_resolutions=[32], resolution=1024, z_channels=256,
                    embed_dim=256, n_embed=8192, use_init_downsample=True)
    eng = Engine(None, s1, dev, a.batch)
    eng.load(stage1=synth.stage1_weights(s1, 1, 'bench'))
    eng.finalize()
    prec = {'fast': PRECISION_FAST, 'split': PRECISION_SPLIT}.get(a.precision, PRECISION_EXACT)
    r = np.random.default_rng(0)
    ct = torch.from_numpy(r.integers(0, s1.n_embed, (a.batch, s1.z_res // 2, s1.z_res // 2))).to(dev)
    cb = torch.from_numpy(r.integers(0, s1.n_embed, (a.batch, s1.z_res, s1.z_res))).to(dev)
    px = eng.decode(ct, cb, precision=prec, clamp01=True)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(px).all()) and tuple(px.shape) == (a.batch, 3, 1024, 1024)
    again = eng.decode(ct, cb, precision=prec, clamp01=True)
    deterministic = bool(torch.equal(px, again))
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
    fl = decoder_flops(s1)
    print(json.dumps({'workload': f'hq-vae decoder only, 1024x1024, ch_mult [1,2,4,4,4], codes 16x16 + 32x32, batch {a.batch}, {a.precision}',
                      'images_per_s': round(a.batch / ms * 1e3, 2), 'ms_per_batch': round(ms, 2), 'flop_per_image': fl,
                      'tflops': round(fl * a.batch / ms / 1e9, 1), 'frac_of_bf16_mfma_peak_2500': round(fl * a.batch / ms / 1e9 / 2500, 4),
                      'deterministic': deterministic, 'workspace_gb': round(eng.workspace_bytes() / 2 ** 30, 2),
                      'kernel_ms': {k: [n, round(t, 3)] for k, (n, t) in sorted(rep.items(), key=lambda kv: -kv[1][1])}}))


if __name__ == '__main__':
    main()
