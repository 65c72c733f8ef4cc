"""Host-side mirror of the reference's model surface for the sampling path.

``ImageGPT2(config)`` exposes ``.stage1`` / ``.stage2`` with the attributes and methods the reference's
drivers touch (``measure_throughput/__main__.py:51-113``, ``sampling_hqmodel.py:64-121,156-225``,
``hqvae/utils/sampling.py:183-192``): ``eval()``, ``to(device)``, ``load_state_dict(sd, strict=True)``,
``stage2.parameters()``, ``stage2.use_cls_cond / use_txt_cond / idx_pred / sos / tok_emb_txt / pos_emb_txt``,
``stage1.decode_code(code_t, code_b)``.  The objects hold fp32 master weights on the host and one libhqt
handle per stage on the GPU; every tensor operation runs in libhqt.so.  There is no CPU compute path:
using a model that has not been moved to a GPU raises.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Iterator, Optional

import numpy as np
import torch

from . import _lib, synth
from ._lib import PRECISION_EXACT, PRECISION_FAST, PRECISIONS
from .engine import Engine
from .spec import (COND_CLS, COND_TXT, stage2_unused, Stage1Spec, Stage2Spec, stage1_encoder_param_shapes, stage1_is_ignored, stage1_param_shapes,
                   stage1_spec_from_config, stage2_param_shapes, stage2_spec_from_config)


class _Table:
    """Stand-in for an nn.Embedding the drivers only inspect (``.weight.shape``)."""

    def __init__(self, weight: torch.Tensor):
        self.weight = weight


class _Stage:
    prefix = ''

    def __init__(self, shapes: 'OrderedDict[str, tuple]', init: Dict[str, np.ndarray]):
        self._shapes = shapes
        self._w: 'OrderedDict[str, torch.Tensor]' = OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in init.items())
        self._device = torch.device('cpu')
        self._engine: Optional[Engine] = None
        self._dirty = True

    # -- nn.Module-like surface
    def parameters(self) -> Iterator[torch.Tensor]:
        return iter(self._w.values())

    def state_dict(self) -> 'OrderedDict[str, torch.Tensor]':
        return OrderedDict(self._w)

    def eval(self):
        return self

    def to(self, device=None, **_):
        if device is not None:
            self._device = torch.device(device)
            if self._device.type == 'cuda' and self._device.index is None:
                self._device = torch.device('cuda', torch.cuda.current_device())
            self._drop_engine()
        return self

    def cuda(self, device=None):
        return self.to(torch.device('cuda', device if device is not None else torch.cuda.current_device()))

    def _drop_engine(self):
        if self._engine is not None:
            self._engine.close()                     # closes its lanes first
        self._engine = None
        self._lanes = {}

    def _lane(self, base: Engine, lane: int) -> Engine:
        """Lane 0 is the engine that owns the weights; lane k > 0 is a clone sharing them (own workspace), created on
        first use.  One lane per batch in flight (hqtransformer_amd.pipeline)."""
        if lane == 0:
            return base
        lanes = self.__dict__.setdefault('_lanes', {})
        if lane not in lanes:
            lanes[lane] = base.clone()
        return lanes[lane]

    def _ignored(self, key: str) -> bool:
        return False

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        """Same contract as nn.Module.load_state_dict (``sampling_hqmodel.py:77-79``): with strict=True a
        missing or unexpected key, or a shape mismatch, raises RuntimeError."""
        missing = [k for k in self._shapes if k not in sd]
        unexpected = [k for k in sd if k not in self._shapes and not self._ignored(k)]
        errors = []
        for k, shp in self._shapes.items():
            if k in sd and tuple(sd[k].shape) != tuple(shp):
                errors.append(f'size mismatch for {k}: checkpoint {tuple(sd[k].shape)} vs model {tuple(shp)}')
        if errors or (strict and (missing or unexpected)):
            raise RuntimeError(f'Error(s) in loading state_dict: missing={missing} unexpected={unexpected} ' + '; '.join(errors))
        for k in self._shapes:
            if k in sd:
                self._w[k] = torch.as_tensor(sd[k]).detach().to('cpu', torch.float32).contiguous().clone()
        self._drop_engine()
        return missing, unexpected

    def from_ckpt(self, path: str, strict: bool = True, ignore_keys=None, strip_prefix: int = 0) -> None:
        """``iHQGPT.from_ckpt`` (hierarchical_ar.py:880-886) / ``SimRQGAN2Generator.from_ckpt`` (generator.py:389-395): a Lightning
        checkpoint's ``state_dict`` straight into this stage.  ``strip_prefix`` characters are cut from every key first (the
        stage-1 generator's own from_ckpt cuts 10, ``'generator.'``); ``ignore_keys`` are dropped as the stage-2 method does."""
        sd = torch.load(path, map_location='cpu')['state_dict']
        sd = {k[strip_prefix:]: v for k, v in sd.items()}
        for k in (ignore_keys or []):
            del sd[k]
        self.load_state_dict(sd, strict=strict)
        print(f'{path} successfully restored..')

    def range_check(self) -> None:
        """SPLIT precision carries fp32 activations as fp16 hi / lo pairs: an activation that is NaN or beyond 65504 invalidates the call
        that met it (include/hqt.h: hqt_range_check).  Waits for the pending SPLIT calls of every lane of this stage and raises HqtError
        if one of them did; a no-op (no synchronisation) when no SPLIT call is pending."""
        if self._engine is not None:
            for e in [self._engine] + list(self.__dict__.get('_lanes', {}).values()):
                e.range_check()

    def _need_gpu(self):
        if self._device.type != 'cuda':
            raise _lib.HqtLibraryError('the model is on the CPU: call .to("cuda") first (hqtransformer_amd has no CPU compute path)')


class HQTransformerStage2(_Stage):
    """Counterpart of ``iHQGPT`` (``hqvae/models/stage2/hierarchical_ar.py:23-216``) for model_type 'parallel' and, with
    ``spec.levels == 3``, of the three-level ``HQTransformer`` ('parallel-add', 'parallel', 'parallel-reduce': ``spec.depth_decoding``; ``hqvae/models/stage2/hqtransformer.py``)."""

    def __init__(self, spec: Stage2Spec, seed: int = 0):
        super().__init__(stage2_param_shapes(spec), synth.stage2_weights(spec, seed, 'bench'))
        self.spec = spec
        self.use_cls_cond = spec.cond == COND_CLS
        self.use_txt_cond = spec.cond == COND_TXT
        self.idx_pred = spec.idx_pred
        self.ctx_len_img = spec.ctx_len_img
        self.n_layers = spec.n_layers
        self.n_layers_depth = spec.n_layers_depth
        self.model_type = 'parallel'
        self.code_level = spec.levels               # HQTransformer.code_level (hqtransformer.py:185)

    # attributes sampling.py:183-192 and the notebook read
    @property
    def sos(self):
        return _Table(self._w['sos.weight']) if self.use_cls_cond else self._w.get('sos')

    @property
    def tok_emb_txt(self):
        return _Table(self._w['tok_emb_txt.weight'])

    @property
    def pos_emb_txt(self):
        return _Table(self._w['pos_emb_txt.weight'])

    def engine(self, batch: int, n_steps: int, lane: int = 0) -> Engine:
        self._need_gpu()
        e = self._engine
        if e is None or batch > e.max_batch or n_steps > e.max_steps:
            self._drop_engine()
            e = Engine(self.spec, None, self._device, max(batch, e.max_batch if e else 0), self.spec.ctx_len_img)
            unused = stage2_unused(self.spec)
            e.load(stage2={k: v for k, v in self._w.items() if k not in unused})
            e.finalize()
            self._engine = e
        return self._lane(e, lane)


class HQVAEStage1(_Stage):
    """Counterpart of ``SimRQGAN2Generator`` (``hqvae/models/stage1/generator.py:176-395``) and, with ``spec.code_levels == 3``,
    of ``HQVAEGenerator`` (generator.py:398-615): ``decode_code`` / ``decode`` and the encode side (``encode``, ``get_codes``)."""

    def __init__(self, spec: Stage1Spec, seed: int = 0):
        shapes = OrderedDict(stage1_encoder_param_shapes(spec))
        shapes.update(stage1_param_shapes(spec))
        w = synth.stage1_weights(spec, seed, 'bench', encoder=True)
        super().__init__(shapes, OrderedDict((k, w[k]) for k in shapes))
        self.spec = spec
        # the reference decodes outside autocast, i.e. in fp32 (measure_throughput:108-111): 'split' is fp32-accurate (1e-4 pixel bar,
        # tests/test_gpu_split.py) on the matrix cores; 'exact' = fp32 FMA chains on the vector ALUs, 'fast' = bf16
        self.precision = 'split'
        self.bottom_window = 2

    def _ignored(self, key: str) -> bool:
        return stage1_is_ignored(key)

    def from_ckpt(self, path: str, strict: bool = True) -> None:       # generator.py:389-395: keys lose their first 10 characters
        super().from_ckpt(path, strict=strict, strip_prefix=10)

    def _prec(self, precision: Optional[str]) -> int:
        name = precision or self.precision
        if name not in PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(PRECISIONS)}, got {name!r}")
        return PRECISIONS[name]

    def engine(self, batch: int, lane: int = 0) -> Engine:
        self._need_gpu()
        e = self._engine
        if e is None or batch > e.max_batch:
            self._drop_engine()
            e = Engine(None, self.spec, self._device, max(batch, e.max_batch if e else 0))
            e.load(stage1=self._w)
            e.finalize()
            self._engine = e
        return self._lane(e, lane)

    def decode_code(self, code_t, code_b: Optional[torch.Tensor] = None, precision: Optional[str] = None,
                    clamp01: bool = False, lane: int = 0, check_range: bool = True) -> torch.Tensor:
        """``SimRQGAN2Generator.decode_code`` (generator.py:323-367): int64 code grids -> fp32 [B, 3, H, W],
        unclamped; either level may be None (zero quant).  ``precision``: 'exact' (fp32 FMA chains on the vector ALUs),
        'split' (fp32-accurate on the matrix cores: fp16 hi/lo operands, fp32 accumulation -- the reference decodes in fp32,
        and this meets its 1e-4 pixel bar about 6x faster than 'exact') or 'fast' (bf16 MFMA); defaults to ``self.precision``.
        ``check_range`` (SPLIT only): True = wait for the call and raise HqtError if an activation left the fp16 range (a stream
        synchronisation: the call is then NOT asynchronous); False = stay asynchronous, the caller runs ``range_check()`` itself
        once its pipeline drains (what ``decode_sequences`` always does)."""
        if isinstance(code_t, (list, tuple)):        # HQVAEGenerator.decode_code([t, m, b]) (generator.py:577-599)
            codes = list(code_t)
            ref = next(c for c in codes if c is not None)
            prec = self._prec(precision)
            eng = self.engine(int(ref.shape[0]), lane)
            px = eng.decode3(codes, precision=prec, clamp01=clamp01)
            if check_range:
                eng.range_check()
            return px
        assert code_t is not None or code_b is not None
        ref = code_t if code_t is not None else code_b
        prec = self._prec(precision)
        eng = self.engine(int(ref.shape[0]), lane)
        px = eng.decode(code_t, code_b, precision=prec, clamp01=clamp01)
        if check_range:
            eng.range_check()
        return px

    # -- encode side (generator.py:298-310, 369-370; HQVAEGenerator.encode 530-568)
    def _encode(self, x: torch.Tensor, precision: Optional[str], lane: int, check_range: bool = True, **want):
        prec = self._prec(precision)
        eng = self.engine(int(x.shape[0]), lane)
        out = eng.encode(x, precision=prec, **want)
        if check_range:                                  # SPLIT: waits for the call (see decode_code)
            eng.range_check()
        return out

    def encode(self, x: torch.Tensor, precision: Optional[str] = None, lane: int = 0, check_range: bool = True):
        """Two levels -- ``SimRQGAN2Generator.encode``: ``(quant_t, quant_b, diff_t, diff_b, (code_t, code_b, h_b))`` with the codes
        flattened as the reference returns them ([B * r_l * r_l]) and ``h_b`` the bottom quantiser's input.
        Three levels -- ``HQVAEGenerator.encode``: ``(quant, diffs, codes, resids[1:])`` with ``quant`` the summed reconstruction."""
        three = self.spec.code_levels == 3
        o = self._encode(x, precision, lane, check_range, want_quant=not three, want_resid=True, want_recon=three, want_diff=True)
        codes = [c.reshape(-1) for c in o['codes']]
        diffs = list(o['diff'].unbind(0))
        if three:
            return o['recon'], diffs, codes, o['resid'][1:]
        return o['quant'][0], o['quant'][1], diffs[0], diffs[1], (codes[0], codes[1], o['resid'][1])

    def get_codes(self, x: torch.Tensor, precision: Optional[str] = None, lane: int = 0, check_range: bool = True):
        """``SimRQGAN2Generator.get_codes`` (generator.py:369-370): ``(code_t, code_b)``, flattened; three levels: the list of codes."""
        o = self._encode(x, precision, lane, check_range)
        codes = [c.reshape(-1) for c in o['codes']]
        return codes if self.spec.code_levels == 3 else (codes[0], codes[1])

    def code_grids(self, x: torch.Tensor, precision: Optional[str] = None, lane: int = 0, check_range: bool = True):
        """The codes as grids [B, r_l, r_l], coarse -> fine: what ``decode_code`` takes back."""
        return self._encode(x, precision, lane, check_range)['codes']

    def forward(self, x: torch.Tensor, precision: Optional[str] = None, lane: int = 0, check_range: bool = True) -> torch.Tensor:
        """Reconstruction ``decode(encode(x))`` (the ``dec`` of ``SimRQGAN2Generator.forward`` in eval mode, generator.py:262-280;
        eval_stage1.py reads only this output)."""
        grids = self.code_grids(x, precision, lane, check_range=False)        # one check below covers both halves (same engine, same stream)
        if self.spec.code_levels == 3:
            return self.decode_code(list(grids), precision=precision, lane=lane, check_range=check_range)
        return self.decode_code(grids[0], grids[1], precision=precision, lane=lane, check_range=check_range)

    __call__ = forward

    def decode_sequences(self, codes_top, codes_bot: Optional[torch.Tensor] = None, precision: Optional[str] = None,
                         clamp01: bool = False, lane: int = 0) -> torch.Tensor:
        """Decode the sampler's own outputs ([B, HW], [B, HW, 4]); the two rearranges of
        sampling_hqmodel.py:119-120 are folded into the codebook-gather addressing."""
        prec = self._prec(precision)
        if isinstance(codes_top, (list, tuple)):     # three levels: [B, L], [B, L, 4], [B, L, 16]
            return self.engine(int(codes_top[0].shape[0]), lane).decode3(list(codes_top), precision=prec, clamp01=clamp01, seq_layout=True)
        return self.engine(int(codes_top.shape[0]), lane).decode(codes_top, codes_bot, precision=prec, clamp01=clamp01, seq_layout=True)


class ImageGPT2:
    """Counterpart of ``hqvae.models.ImageGPT2`` (``hqvae/models/__init__.py:92-174``): builds stage 1 and
    stage 2 from a merged config with random-init weights (the reference's harness never loads a checkpoint,
    ``measure_throughput/__main__.py:25-31``)."""

    def __init__(self, config, seed: Optional[int] = None):
        seed = int(torch.initial_seed() % (2 ** 31)) if seed is None else int(seed)
        self.config = config
        self.stage1 = HQVAEStage1(stage1_spec_from_config(config), seed + 1)
        self.stage2 = HQTransformerStage2(stage2_spec_from_config(config), seed)
        self.use_cls_cond = config.stage2.use_cls_cond
        self.use_txt_cond = config.stage2.use_txt_cond
        self.type = config.stage2.type

    def eval(self):
        return self

    def to(self, device=None, **kw):
        self.stage1.to(device)
        self.stage2.to(device)
        return self

    def cuda(self, device=None):
        self.stage1.cuda(device)
        self.stage2.cuda(device)
        return self

    def parameters(self) -> Iterator[torch.Tensor]:
        yield from self.stage1.parameters()
        yield from self.stage2.parameters()

    def state_dict(self) -> 'OrderedDict[str, torch.Tensor]':
        out = OrderedDict(('stage1.' + k, v) for k, v in self.stage1.state_dict().items())
        out.update(('stage2.' + k, v) for k, v in self.stage2.state_dict().items())
        return out

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        """Keys carry the ``stage1.`` / ``stage2.`` prefixes of the reference's checkpoints
        (``sampling_hqmodel.py:77-79``).  Encoder-side stage-1 tensors are accepted and ignored."""
        s1 = {k[len('stage1.'):]: v for k, v in sd.items() if k.startswith('stage1.')}
        s2 = {k[len('stage2.'):]: v for k, v in sd.items() if k.startswith('stage2.')}
        other = [k for k in sd if not k.startswith(('stage1.', 'stage2.'))]
        if strict and other:
            raise RuntimeError(f'Error(s) in loading state_dict: unexpected={other}')
        m1, u1 = self.stage1.load_state_dict(s1, strict)
        m2, u2 = self.stage2.load_state_dict(s2, strict)
        return ['stage1.' + k for k in m1] + ['stage2.' + k for k in m2], ['stage1.' + k for k in u1] + ['stage2.' + k for k in u2] + other
