#!/usr/bin/env python3
"""FAST (bf16 MFMA) against EXACT (fp32) arithmetic of the AR loop at the benchmark's own model size (ImageNet 12-layer, D = 1536):
teacher-forced logits of every draw, the KL divergence between the two sampling distributions, and how often the draw under the
same noise picks the same code.  GPU box only.

    python tools/fast_ar_error.py [--batch 16] [--positions 8]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hqtransformer_amd import synth  # noqa: E402
from hqtransformer_amd._lib import PRECISION_EXACT, PRECISION_FAST  # noqa: E402
from hqtransformer_amd.config import load_config  # noqa: E402
from hqtransformer_amd.engine import Engine  # noqa: E402
from hqtransformer_amd.spec import stage2_spec_from_config  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--positions', type=int, default=8)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    s2 = stage2_spec_from_config(load_config('configs/imagenet-12l.yaml'))
    eng = Engine(s2, None, dev, a.batch, s2.ctx_len_img)
    eng.load(stage2=synth.stage2_weights(s2, 0, 'bench'))
    eng.finalize()
    B, n, V = a.batch, a.positions, s2.vocab_top
    noise = torch.from_numpy(synth.exp_noise(4, n, B, V))
    cond = torch.from_numpy(synth.class_ids(5, B, s2.n_classes))
    ct, cb, lg_e = eng.sample(B, cond, n, precision=PRECISION_EXACT, noise=noise, return_logits=True, use_graph=False)
    ft, fb, lg_f = eng.sample(B, cond, n, precision=PRECISION_FAST, noise=noise, force_top=ct, force_bot=cb, return_logits=True, use_graph=False)
    le, lf = lg_e.double(), lg_f.double()                     # [n, 5, B, V]
    d = (le - lf).abs()
    pe, pf = torch.softmax(le, -1), torch.softmax(lf, -1)
    kl = (pe * (torch.log(pe.clamp_min(1e-300)) - torch.log(pf.clamp_min(1e-300)))).sum(-1)
    q = noise.to(dev).double()
    same = (torch.argmax(pe / q, -1) == torch.argmax(pf / q, -1)).double().mean()
    print(json.dumps({'workload': f'imagenet 12L/1536d, batch {B}, {n} positions x 5 draws, teacher-forced on the fp32 codes',
                      'logit_std': round(float(le.std()), 3), 'logit_abs_diff_max': round(float(d.max()), 4),
                      'logit_abs_diff_mean': round(float(d.mean()), 5), 'kl_exact_fast_mean_nats': float(kl.mean()),
                      'kl_max_nats': float(kl.max()), 'same_draw_under_same_noise': round(float(same), 4)}))


if __name__ == '__main__':
    main()
