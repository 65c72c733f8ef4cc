#!/usr/bin/env python3
"""Fixture G11: state-dict keys and shapes of the reference's own stage-2 module (iHQGPT) for every released two-level config
(build container only: imports /root/reference and reads its YAML files).  The modules are built on the `meta` device, so no
weights are allocated; only names, shapes and the parameter count are stored.

    python tools/gen_released_shapes.py   ->  tests/golden/g11_released_shapes.json
"""
import copy
import json
import os
import sys
import types

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('HQT_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
_om = types.ModuleType('omegaconf')
_om.OmegaConf = object
sys.modules['omegaconf'] = _om
_hq = types.ModuleType('hqvae')
_hq.__path__ = [os.path.join(REF, 'hqvae')]
sys.modules['hqvae'] = _hq
_pkg = types.ModuleType('hqvae.models')
_pkg.__path__ = [os.path.join(REF, 'hqvae/models')]
sys.modules['hqvae.models'] = _pkg

import torch  # noqa: E402
from hqvae.models.stage2.hierarchical_ar import iHQGPT  # noqa: E402

CONFIGS = {   # ours -> the reference's
    'imagenet-12l': 'configs/master/stage2/imagenet/hqtransformer-embtrans1-soft1-layer12-top8x8.yaml',
    'imagenet-24l': 'configs/master/stage2/imagenet/hqtransformer-embtrans1-soft1-layer24-top8x8.yaml',
    'imagenet-42l': 'configs/master/stage2/imagenet/hqtransformer-embtrans1-soft1-layer42-top8x8.yaml',
    'ffhq-24l': 'configs/master/stage2/ffhq/hqtransformer-adding-soft1-layer24-top8x8.yaml',
    'cc15m-12l-txt': 'configs/master/stage2/cc15m/hqtransformer-embtrans1-soft1-layer12-top8x8-cc15m.yaml',
}


class AD(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __deepcopy__(self, memo):
        return AD(copy.deepcopy(dict(self), memo))


# Stage2Hparams defaults of hqvae/utils/config2.py:57-82 that the YAML files omit
DEFAULTS = dict(embed_dim=1536, n_layers=42, n_heads=24, n_dense_layers=42, ctx_len=None, ctx_len_img=256, ctx_len_txt=64,
                embd_pdrop=0.0, resid_pdrop=0.0, attn_pdrop=0.0, mlp_bias=True, attn_bias=True, gelu_use_approx=False,
                use_head_txt=True, n_classes=1000, causal_attn=None, embedding_type='transformer1', position_embedding='1d',
                bottom_head_type='linear', use_random_order=False, rate_random_order=1.0)


def main():
    out = {}
    for name, rel in CONFIGS.items():
        cfg = yaml.safe_load(open(os.path.join(REF, rel)))
        s2 = cfg['stage2']
        hp = AD(dict(DEFAULTS, **s2['hparams']))
        hp_dec = AD(dict(DEFAULTS, **s2['hparams_dec'])) if s2.get('hparams_dec') else None
        with torch.device('meta'):
            m = iHQGPT(s2['vocab_size_img'], s2['vocab_size_img'], s2.get('vocab_size_txt', 16384), s2.get('ratio_bot2top', 4),
                       bool(s2.get('use_cls_cond')), bool(s2.get('use_txt_cond')), s2['type'].split('/')[1], hp, hp_dec)
        sd = m.state_dict()
        out[name] = {'reference_config': rel, 'n_params': int(sum(v.numel() for v in sd.values())),
                     'shapes': {k: list(v.shape) for k, v in sd.items()}}
        print(name, out[name]['n_params'] / 1e6, len(sd))
    path = os.path.join(ROOT, 'tests', 'golden', 'g11_released_shapes.json')
    with open(path, 'w') as fp:
        json.dump(out, fp, separators=(',', ':'))
    print(path, os.path.getsize(path))


if __name__ == '__main__':
    main()
