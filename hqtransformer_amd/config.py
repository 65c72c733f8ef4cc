"""Configuration objects for the HQ-Transformer sampling path.

The reference merges an OmegaConf structured default (``hqvae/utils/config2.py:12-163``,
``get_base_config(use_default=False)``) with an experiment YAML
(``measure_throughput/__main__.py:25-31``).  ``omegaconf`` is not a dependency here: the YAML is read
with PyYAML and the defaults the YAMLs leave out are restated below (field names are the reference's
so the shipped YAML files load unchanged).  Only the keys the sampling path reads are kept.
"""
from __future__ import annotations

import copy
from typing import Any, Dict, Iterable, Optional

import yaml


class Config(dict):
    """dict with attribute access (``cfg.stage2.hparams.embed_dim``), deep-copyable, ``**``-expandable."""

    def __getattr__(self, k: str) -> Any:
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k: str, v: Any) -> None:
        self[k] = v

    def __deepcopy__(self, memo):
        return Config({k: copy.deepcopy(v, memo) for k, v in self.items()})

    @staticmethod
    def wrap(obj: Any) -> Any:
        if isinstance(obj, dict):
            return Config({k: Config.wrap(v) for k, v in obj.items()})
        if isinstance(obj, (list, tuple)):
            return [Config.wrap(v) for v in obj]
        return obj


# Defaults the YAML files omit.  config2.py:22-36 (Stage1Hparams), :39-46 (VQGAN2Hparams),
# :50-71 (Stage2Hparams), :74-82 (Stage1Config), :85-105 (Stage2Config), :12-19 (DataConfig).
_STAGE1_HPARAMS = dict(double_z=False, z_channels=256, resolution=256, in_channels=3, out_ch=3, ch=128,
                       ch_mult=[1, 1, 2, 2, 4], num_res_blocks=2, attn_resolutions=[16], pdrop=0.0,
                       use_init_downsample=False, use_mid_block=True, use_attn=True)
_STAGE1_AUX = dict(upsample=None, shared_codebook=None, bottom_start=100000000000, decoding_type='concat',
                   restart_unused_codes=None, code_levels=None)
_STAGE2_HPARAMS = dict(embed_dim=1536, n_layers=42, n_heads=24, n_dense_layers=42, ctx_len=None,
                       ctx_len_img=256, ctx_len_txt=64, embd_pdrop=0.0, resid_pdrop=0.0, attn_pdrop=0.0,
                       mlp_bias=True, attn_bias=True, gelu_use_approx=False, use_head_txt=True,
                       n_classes=None, causal_attn=None, embedding_type='baseline',
                       position_embedding='1d', bottom_head_type='linear', use_random_order=False,
                       rate_random_order=1.0)
_BASE = dict(
    dataset=dict(dataset=None, tokenizer_type='bpe16k_huggingface', context_length=64,
                 image_resolution=256, transforms='dalle-vqvae', bpe_pdrop=0.1),
    stage1=dict(type='vqgan', embed_dim=256, n_embed=16384, n_embed_levels=[8192, 8192, 8192],
                ema_update=False, hparams=dict(_STAGE1_HPARAMS), hparams_aux=None),
    stage2=dict(type='transformer1d', vocab_size_txt=16384, vocab_size_img=16384,
                vocab_sizes_img=[8192, 8192, 8192], decoding_type=None, ratio_bot2top=4,
                use_pretrained=False, use_cls_cond=None, use_txt_cond=None, weight_bottom=4.0,
                weight_txt=None, weight_img=None, gamma_focal_loss=None, temp_soft_labels=None,
                use_l2norm_logits=None, hparams=None, hparams_enc=None, hparams_dec=None),
)


def get_base_config(use_default: bool = False) -> Config:
    """Counterpart of ``hqvae.utils.config2.get_base_config`` (config2.py:162-163).

    ``use_default`` only toggles the optimizer/experiment sections in the reference; neither is on the
    sampling path, so both values return the same defaults here.
    """
    return Config.wrap(copy.deepcopy(_BASE))


def _merge(base: Any, over: Any) -> Any:
    if isinstance(base, dict) and isinstance(over, dict):
        out = dict(base)
        for k, v in over.items():
            out[k] = _merge(base.get(k), v) if k in base else v
        return out
    return copy.deepcopy(over)


def merge(base: Dict, over: Dict) -> Config:
    """Counterpart of ``OmegaConf.merge(base, experiment)`` for the keys above, including the typed
    ``Optional[Stage2Hparams] = None`` slots that pick up dataclass defaults when the YAML fills them."""
    out = _merge(dict(base), dict(over))
    s2 = out['stage2']
    for slot in ('hparams', 'hparams_enc', 'hparams_dec'):
        if s2.get(slot) is not None:
            s2[slot] = _merge(_STAGE2_HPARAMS, s2[slot])
    s1 = out['stage1']
    if s1.get('hparams_aux') is not None:
        s1['hparams_aux'] = _merge(_STAGE1_AUX, s1['hparams_aux'])
    return Config.wrap(out)


def load_config(path: str, overrides: Optional[Iterable[str]] = None) -> Config:
    """``get_base_config(False)`` merged with the YAML at ``path`` (``measure_throughput/__main__.py:25-29``)."""
    with open(path, 'r') as fp:
        exp = yaml.safe_load(fp)
    cfg = merge(get_base_config(False), exp)
    for item in overrides or ():
        set_dotted(cfg, *item.split('=', 1))
    return cfg


def set_dotted(cfg: Dict, key: str, value: str) -> None:
    node = cfg
    parts = key.split('.')
    for p in parts[:-1]:
        node = node[p]
    node[parts[-1]] = yaml.safe_load(value)


def parse_dotlist(argv: Iterable[str], defaults: Dict[str, Any]) -> Config:
    """``OmegaConf.merge(structured(Experiment), from_cli())`` for flat ``key=value`` arguments
    (``measure_throughput/__main__.py:185``): unknown keys are an error, values are YAML-typed."""
    out = dict(defaults)
    for item in argv:
        if '=' not in item:
            raise ValueError(f"expected key=value, got '{item}'")
        k, v = item.split('=', 1)
        if k not in out:
            raise KeyError(f"Key '{k}' not in '{type(defaults).__name__}'")
        val = yaml.safe_load(v)
        if out[k] is not None and not isinstance(val, type(out[k])) and not (
                isinstance(out[k], float) and isinstance(val, int)):
            val = type(out[k])(val)
        out[k] = val
    return Config.wrap(out)
