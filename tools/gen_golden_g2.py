#!/usr/bin/env python3
"""Fixture G2: single Block / ParallelBlock steps of the reference, with and without past K/V.

Container-only, like tools/gen_golden.py (whose import shims and model builders it reuses): imports /root/reference,
loads weights derived from numpy.default_rng by state-dict name (hqtransformer_amd/synth.py, 'fixture' profile) and
stores only inputs and outputs in tests/golden/g2_block_step.npz:
  * tiny (D = 128, 4 heads) and shape-true (D = 1536, 24 heads of 64, one layer, B = 2) configurations;
  * body Block.sample (stage2/layers.py:324-328): a 3-token causal prefix, then one cached decode step;
  * depth ParallelBlock.sample (layers.py:371-375): the no-past single token (attention == value) and the four-token
    step over one past key (all-ones mask).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (installs the import shims)
import torch  # noqa: E402

from hqtransformer_amd.spec import Stage2Spec  # noqa: E402


def kv(present):
    """reference 'present' -> (K, V) as [B*nh, T, hs] float32 arrays"""
    if isinstance(present, (list, tuple)):
        k, v = present
    else:
        k, v = present[0], present[1]
    return k.detach().numpy().astype(np.float32), v.detach().numpy().astype(np.float32)


def main():
    out = {}
    specs = {
        'tiny': Stage2Spec(embed_dim=128, n_layers=1, n_heads=4, n_layers_depth=1, vocab_top=64, vocab_bot=64, vocab_txt=64,
                           ctx_len_img=64, ctx_len_txt=16, n_classes=10, cond=1, embedding=0),
        'true': Stage2Spec(embed_dim=1536, n_layers=1, n_heads=24, n_layers_depth=1, vocab_top=64, vocab_bot=64, vocab_txt=64,
                           ctx_len_img=64, ctx_len_txt=16, n_classes=10, cond=1, embedding=0),
    }
    for name, spec in specs.items():
        model, _ = G.build_stage2(spec, 77)
        rng = np.random.default_rng(78)
        B, D = 2, spec.embed_dim
        xp = rng.standard_normal((B, 3, D)).astype(np.float32)
        xn = rng.standard_normal((B, 1, D)).astype(np.float32)
        xd0 = rng.standard_normal((B, 1, D)).astype(np.float32)
        xd1 = rng.standard_normal((B, 4, D)).astype(np.float32)
        with torch.no_grad():
            blk, dep = model.blocks[0], model.depths[0]
            yp, pres = blk.sample(torch.from_numpy(xp))
            yn, pres2 = blk.sample(torch.from_numpy(xn), layer_past=pres)
            y0, dp0 = dep.sample(torch.from_numpy(xd0))
            y1, dp1 = dep.sample(torch.from_numpy(xd1), layer_past=dp0)
        k2, v2 = kv(pres2)
        dk1, dv1 = kv(dp1)
        out.update({f'{name}_spec': G.spec_json(spec), f'{name}_xp': xp, f'{name}_xn': xn, f'{name}_yp': yp.numpy(), f'{name}_yn': yn.numpy(),
                    f'{name}_k': k2, f'{name}_v': v2, f'{name}_xd0': xd0, f'{name}_xd1': xd1, f'{name}_yd0': y0.numpy(),
                    f'{name}_yd1': y1.numpy(), f'{name}_dk': dk1, f'{name}_dv': dv1})
        print(name, 'K', k2.shape, 'depth K', dk1.shape, 'y', yn.shape)
    out['weight_seed'] = np.int64(77)
    np.savez_compressed(os.path.join(G.OUT, 'g2_block_step.npz'), **out)
    print('g2_block_step ok', os.path.getsize(os.path.join(G.OUT, 'g2_block_step.npz')), 'bytes')


if __name__ == '__main__':
    main()
