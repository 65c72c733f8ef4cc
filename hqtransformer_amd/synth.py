"""Deterministic synthetic weights, text prompts and sampler noise.

No checkpoint or dataset is reachable offline and the reference's own throughput harness runs on
random-init weights (``measure_throughput/__main__.py:25-31`` never loads a checkpoint), so every
tensor is derived from ``numpy.random.default_rng`` keyed by (seed, state-dict name).  The fixture
generator (tools/gen_golden.py) loads the same tensors into the reference, so both sides regenerate
identical weights and nothing but inputs/outputs is stored under tests/golden/.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np

from .spec import Stage1Spec, Stage2Spec, stage1_encoder_param_shapes, stage1_param_shapes, stage2_param_shapes


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def _trained_gain(r: np.random.Generator, shape, spread: float) -> np.ndarray:
    """Affine gains of a trained normalisation layer: spread around 1, never near 0, a few outlier channels several times larger."""
    g = np.clip(1.0 + spread * r.standard_normal(shape), 0.2, 3.0)
    out = r.random(shape) < 0.02
    return np.where(out, 4.0 * g, g).astype(np.float32)


def _stage2_trained(name: str, shape: Tuple[int, ...], r: np.random.Generator) -> np.ndarray:
    """'trained' profile: the statistics a trained HQ-Transformer shows and random initialisation does not -- LayerNorm gains spread around 1
    with outlier channels, non-zero shifts and biases, embeddings of unit scale, and residual writers (attn.proj, mlp.2) strong enough that
    the residual stream grows ~20x over the body (|x| of several tens): what the fp32-accurate and bf16 paths must survive on real checkpoints."""
    leaf = name.split('.')[-1]
    is_ln = ('.ln' in name or name.startswith('ln_')) and len(shape) == 1
    if is_ln:
        return _trained_gain(r, shape, 0.3) if leaf == 'weight' else (0.2 * r.standard_normal(shape)).astype(np.float32)
    if leaf == 'bias':
        return (0.1 * r.standard_normal(shape)).astype(np.float32)
    if name in ('sos_depth', 'sos'):
        return r.standard_normal(shape).astype(np.float32)
    if name.startswith('head_'):
        std = 3.0 / np.sqrt(shape[1])
    elif '.attn.proj.' in name or '.mlp.2.' in name:
        std = 3.0 / np.sqrt(shape[1])                      # residual writers
    elif '.attn.query.' in name or '.attn.key.' in name:
        std = 1.5 / np.sqrt(shape[1])                      # peaked attention
    elif '.attn.' in name or '.mlp.' in name:
        std = 1.0 / np.sqrt(shape[1])
    elif name.startswith('pos_emb'):
        std = 0.3
    else:
        std = 0.5                                          # token / class embeddings
    return (std * r.standard_normal(shape)).astype(np.float32)


def _stage1_trained(name: str, shape: Tuple[int, ...], r: np.random.Generator) -> np.ndarray:
    """'trained' profile of the HQ-VAE: GroupNorm gains / shifts drawn non-trivially, filters strong enough that the decoder's activations
    reach |x| ~ 1e2 (STAGE1_TRAINED_GAIN: calibrated with tools/gen_golden_trained.py --calibrate)."""
    leaf = name.split('.')[-1]
    if leaf == 'embedding':
        return r.standard_normal(shape).astype(np.float32)
    if '.norm' in name and len(shape) == 1:
        return _trained_gain(r, shape, 0.5) if leaf == 'weight' else (0.3 * r.standard_normal(shape)).astype(np.float32)
    if leaf == 'bias':
        return (0.1 * r.standard_normal(shape)).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    gain = STAGE1_TRAINED_GAIN if ('conv2' in name or 'proj_out' in name or 'nin_shortcut' in name or 'upsample' in name) else 1.0
    if 'conv_out' in name:
        gain = 0.3                                         # pixels stay O(1)
    return (gain * r.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)


STAGE1_TRAINED_GAIN = 1.5


def _stage2_tensor(name: str, shape: Tuple[int, ...], seed: int, profile: str) -> np.ndarray:
    r = _rng(seed, name)
    if profile == 'trained':
        return _stage2_trained(name, shape, r)
    rich = profile == 'fixture'
    leaf = name.split('.')[-1]
    is_ln = ('.ln' in name or name.startswith('ln_')) and len(shape) == 1
    if is_ln:
        if leaf == 'weight':
            return (1.0 + (0.2 * r.standard_normal(shape) if rich else 0.0) * np.ones(shape)).astype(np.float32)
        return ((0.1 * r.standard_normal(shape)) if rich else np.zeros(shape)).astype(np.float32)
    if leaf == 'bias':
        return ((0.05 * r.standard_normal(shape)) if rich else np.zeros(shape)).astype(np.float32)
    if name in ('sos_depth', 'sos'):                       # nn.Parameter(torch.randn) hierarchical_ar.py:77,156
        return r.standard_normal(shape).astype(np.float32)
    if rich:
        if name.startswith('head_'):
            std = 3.0 / np.sqrt(shape[1])                  # logits with std ~3 so top-k/top-p actually cut
        elif '.attn.' in name or '.mlp.' in name:
            std = 1.0 / np.sqrt(shape[1])
        else:
            std = 0.5                                      # embeddings
    else:
        std = 0.02                                         # hierarchical_ar.py:218-225
    return (std * r.standard_normal(shape)).astype(np.float32)


def _stage1_tensor(name: str, shape: Tuple[int, ...], seed: int, profile: str) -> np.ndarray:
    r = _rng(seed, name)
    if profile == 'trained':
        return _stage1_trained(name, shape, r)
    rich = profile == 'fixture'
    leaf = name.split('.')[-1]
    if leaf == 'embedding':                                # quantizer.py:76 randn
        return r.standard_normal(shape).astype(np.float32)
    if '.norm' in name and len(shape) == 1:
        if leaf == 'weight':
            return (1.0 + (0.2 * r.standard_normal(shape) if rich else 0.0) * np.ones(shape)).astype(np.float32)
        return ((0.1 * r.standard_normal(shape)) if rich else np.zeros(shape)).astype(np.float32)
    if leaf == 'bias':
        return (0.05 * r.standard_normal(shape)).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    return (r.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)


def stage2_weights(spec: Stage2Spec, seed: int = 0, profile: str = 'bench') -> 'OrderedDict[str, np.ndarray]':
    return OrderedDict((n, _stage2_tensor(n, s, seed, profile)) for n, s in stage2_param_shapes(spec).items())


def stage1_weights(spec: Stage1Spec, seed: int = 0, profile: str = 'bench', encoder: bool = False) -> 'OrderedDict[str, np.ndarray]':
    """Decode-side tensors; ``encoder=True`` adds the Encoder and quant_conv_b (tensors are keyed by name, so the decode-side
    values do not depend on the flag)."""
    shapes = OrderedDict(stage1_param_shapes(spec))
    if encoder:
        shapes.update(stage1_encoder_param_shapes(spec))
    return OrderedDict((n, _stage1_tensor(n, s, seed, profile)) for n, s in shapes.items())


def exp_noise(seed: int, n_steps: int, batch: int, vocab: int) -> np.ndarray:
    """Exp(1) noise ``q`` for every multinomial draw: ``[n_steps, 5, batch, vocab]`` fp32, draw order
    per position top, bot0..bot3 (``hierarchical_ar.py:762-785``).  ``torch.multinomial(p, 1)`` is
    ``argmax(p / q)``; the noise is an explicit input so runs are reproducible
    on any device."""
    q = np.random.default_rng([seed, 0x9e3779b9]).standard_exponential((n_steps, 5, batch, vocab), dtype=np.float32)
    return np.maximum(q, np.float32(1e-30))


def class_ids(seed: int, n: int, n_classes: int = 1000) -> np.ndarray:
    return np.random.default_rng([seed, 1]).integers(0, n_classes, size=n, dtype=np.int64)


def text_ids(seed: int, batch: int, ctx: int, vocab: int) -> np.ndarray:
    return np.random.default_rng([seed, 2]).integers(0, vocab, size=(batch, ctx), dtype=np.int64)
