#!/usr/bin/env python3
"""Fixture G12: the REFERENCE on weights with trained-like statistics (hqtransformer_amd.synth profile 'trained').

Every other fixture runs the reference on small random weights ('fixture' / 'bench' profiles): logits of std ~3 at most, residual
streams and decoder activations of order one.  Released checkpoints are unreachable offline, so this profile draws what they have
and random initialisation lacks -- LayerNorm / GroupNorm gains with outlier channels, non-zero shifts, a residual stream that grows
~20x over the body, decoder activations of |x| ~ 1e2 -- and G12 pins the oracle THERE: tiny class-conditional sampling (codes, logits)
and a tiny 64-pixel decode, produced by the reference itself in the build container.

    python tools/gen_golden_trained.py              # writes tests/golden/g12_trained.npz
    python tools/gen_golden_trained.py --calibrate  # prints the activation ranges the profile produces (no file written)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import gen_golden as G  # noqa: E402  (import shims for /root/reference, build_stage1 / build_stage2 / run_sampling)
import torch  # noqa: E402
from hqtransformer_amd import synth  # noqa: E402
from hqtransformer_amd.spec import Stage1Spec, Stage2Spec  # noqa: E402


def hook_ranges(module):
    """max |output| of every leaf module during the next forward calls"""
    seen = {}
    hs = []
    for name, m in module.named_modules():
        if len(list(m.children())) == 0:
            hs.append(m.register_forward_hook(lambda mod, inp, out, name=name: seen.__setitem__(name, max(seen.get(name, 0.0), float(out.abs().max())) if torch.is_tensor(out) else None)))
    return seen, hs


def main():
    calibrate = '--calibrate' in sys.argv
    tiny = Stage2Spec(embed_dim=128, n_layers=4, n_heads=4, n_layers_depth=4, vocab_top=512, vocab_bot=512, vocab_txt=64, ctx_len_img=64,
                      ctx_len_txt=16, n_classes=10, cond=1, embedding=0)
    m, shapes2 = G.build_stage2(tiny, seed=31, profile='trained')
    B, n = 4, 16
    noise = synth.exp_noise(32, n, B, 512)
    seen, hs = hook_ranges(m)
    ct, cb, lg, margin = G.run_sampling(m, tiny, 7, B, n, noise, (None, None), (None, None), (1.0, 1.0))
    for h in hs:
        h.remove()
    stream = max(v for k, v in seen.items() if v is not None and ('mlp.2' in k or 'attn.proj' in k or 'ln' in k))
    print(f'stage 2 (tiny, trained profile): logits std {lg.std():.2f}, max |logit| {np.abs(lg).max():.1f}, largest block output {stream:.1f}, margin {margin:.6f}')

    s1 = Stage1Spec(ch=32, ch_mult=[1, 2], num_res_blocks=2, attn_resolutions=[16], resolution=64, z_channels=32, embed_dim=16, n_embed=64)
    g, shapes1 = G.build_stage1(s1, seed=33, profile='trained')
    r = np.random.default_rng(34)
    code_t, code_b = r.integers(0, 64, (2, 8, 8)), r.integers(0, 64, (2, 16, 16))
    seen1, hs = hook_ranges(g.decoder)
    px = g.decode_code(torch.from_numpy(code_t), torch.from_numpy(code_b)).numpy()
    for h in hs:
        h.remove()
    amax = max(v for v in seen1.values() if v is not None)
    print(f'stage 1 (tiny, trained profile): largest decoder activation {amax:.1f}, pixels in [{px.min():.2f}, {px.max():.2f}]')
    if calibrate:
        from hqtransformer_amd.config import load_config
        from hqtransformer_amd.spec import stage1_spec_from_config
        big = stage1_spec_from_config(load_config(os.path.join(ROOT, 'configs', 'imagenet-12l.yaml')))
        gb, _ = G.build_stage1(big, seed=0, profile='trained')
        seenb, hs = hook_ranges(gb.decoder)
        rb = np.random.default_rng(1)
        pb = gb.decode_code(torch.from_numpy(rb.integers(0, big.n_embed, (1, 8, 8))), torch.from_numpy(rb.integers(0, big.n_embed, (1, 16, 16)))).numpy()
        print(f'stage 1 (ImageNet decoder, trained profile): largest activation {max(v for v in seenb.values() if v is not None):.1f}, pixels in [{pb.min():.2f}, {pb.max():.2f}]')
        return
    np.savez_compressed(os.path.join(G.OUT, 'g12_trained.npz'),
                        spec2=G.spec_json(tiny), weight_seed2=31, noise_seed=32, B=B, n_steps=n, codes_top=ct, codes_bot=cb, logits=lg.astype(np.float32), margin=margin,
                        spec1=G.spec_json(s1), weight_seed1=33, code_t=code_t, code_b=code_b, pixels=px.astype(np.float32), act_max=amax, stream_max=stream,
                        param_shapes2=json.dumps({k: list(v) for k, v in shapes2.items()}), param_shapes1=json.dumps({k: list(v) for k, v in shapes1.items()}))
    print('g12_trained ok')


if __name__ == '__main__':
    main()
