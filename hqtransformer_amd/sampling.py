"""Counterpart of ``hqvae/utils/sampling.py`` for the HQ-Transformer path.

``sampling_ihqgpt`` keeps the reference's signature and return convention (sampling.py:164-237) and runs
the whole 64-position loop inside libhqt.so (KV cache, depth head, fused sampler, next-step embedding).
"""
from __future__ import annotations

from typing import List, Optional

import torch

from ._lib import PRECISION_EXACT, PRECISION_FAST, PRECISIONS


def _seed_from_torch() -> int:
    """The reference consumes torch's global generator through ``torch.multinomial``; the Philox seed of the
    in-kernel Exp(1) noise is drawn from that same generator so ``set_seed`` keeps runs reproducible."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def _precision(precision: Optional[str], use_fp16: bool) -> int:
    if precision is None:
        return PRECISION_FAST if use_fp16 else PRECISION_EXACT
    if precision not in PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(PRECISIONS)}, got {precision!r}")
    return PRECISIONS[precision]


@torch.no_grad()
def sampling_ihqgpt(model,
                    num_candidates: int,
                    cond,
                    top_k_top: Optional[float] = None,
                    top_p_top: Optional[float] = None,
                    top_k_bot: Optional[float] = None,
                    top_p_bot: Optional[float] = None,
                    softmax_temperature: List[float] = [1.0, 1.0],
                    is_tqdm: bool = True,
                    use_fp16: bool = True,
                    max_seq_len: int = 256,
                    model_stage1=None,
                    given_top_code: Optional[torch.LongTensor] = None,
                    noise: Optional[torch.Tensor] = None,
                    sample_offset: int = 0,
                    seed: Optional[int] = None,
                    use_graph: bool = True,
                    lane: int = 0,
                    row_seeds=None,
                    row_offsets=None,
                    precision: Optional[str] = None):
    """Returns ``(codes_top int64 [B, max_seq_len], codes_bot int64 [B, max_seq_len, 4])`` on the model's GPU.

    ``model`` is ``ImageGPT2.stage2``.  ``cond``: python int (class id, repeated for every candidate), an
    int64 tensor [B] of class ids, an int64 tensor [B, ctx_len_txt] (text; B replaces num_candidates,
    sampling.py:187-190) or anything (unconditional).  ``use_fp16=True`` -> FAST (bf16 MFMA) arithmetic,
    ``False`` -> EXACT fp32 (what the reference computes on its CPU path).  ``is_tqdm`` and ``model_stage1``
    are accepted and ignored (the latter only feeds a dead branch, hierarchical_ar.py:697-699).
    Extensions: ``noise`` fp32 [max_seq_len, 5, B, V] Exp(1) variates (draw = argmax(p/q), the multinomial
    identity) for bit-reproducible runs; ``sample_offset``/``seed`` for sharded batches; ``lane`` selects one of
    several workspaces over the same weights (one per batch in flight, see ``hqtransformer_amd.pipeline``);
    ``row_seeds`` / ``row_offsets`` (B entries each): merged steps -- row b draws what global row ``row_offsets[b]`` of a call
    seeded ``row_seeds[b]`` would draw, so several independent calls can share one pass over the weights;
    ``precision`` ('exact' | 'fast' | 'split') overrides ``use_fp16``: 'split' = the fp32 launch sequence with every nn.Linear on the
    matrix cores (fp16 hi / lo operands, three MFMAs per term, fp32 accumulation): code sequences bit-identical to 'exact' wherever the
    draw is well-conditioned, at several times its speed.

    The call is asynchronous and does not read the device's flags: 'split' passes above 256 rows SATURATE activations outside the fp16 range and
    only flag them, and a persistent FAST launch (up to 64 samples) that could not finish on a shared GPU only marks the handle -- call
    ``model.range_check()`` (``hqt_range_check``: raises ``HqtError``) once the codes are needed, as ``InflightSampler.drain`` and ``bench.py`` do.
    """
    spec = model.spec
    if model.use_txt_cond:
        cond = torch.as_tensor(cond)
        if cond.dim() != 2:
            raise ValueError('text conditioning expects cond of shape [B, ctx_len_txt]')
        B = int(cond.shape[0])
    else:
        B = int(num_candidates)
        if model.use_cls_cond:
            if isinstance(cond, int):
                cond = torch.full((B,), int(cond), dtype=torch.int64)
            else:
                cond = torch.as_tensor(cond).reshape(-1)
                if cond.numel() == 1:
                    cond = cond.repeat(B)
            if int(cond.min()) < 0 or int(cond.max()) >= spec.n_classes:
                raise IndexError('index out of range in self')          # what nn.Embedding raises in the reference
        else:
            cond = None
    force_top = None
    if given_top_code is not None:
        force_top = torch.as_tensor(given_top_code)
        if force_top.dim() == 1:
            force_top = force_top.unsqueeze(0)
        if force_top.shape[0] != B:
            force_top = force_top.repeat(B, 1)
        force_top = force_top[:, :max_seq_len]
    eng = model.engine(B, max_seq_len, lane)
    if seed is None and noise is None:
        seed = _seed_from_torch()
    return eng.sample(B, cond, max_seq_len, precision=_precision(precision, use_fp16),
                      top_k=(top_k_top, top_k_bot), top_p=(top_p_top, top_p_bot), temperature=softmax_temperature,
                      noise=noise, seed=seed or 0, sample_offset=sample_offset, force_top=force_top, use_graph=use_graph,
                      row_seeds=row_seeds, row_offsets=row_offsets)


def sampling_hqtransformer(model,
                           num_candidates: int,
                           cond,
                           top_k: Optional[List[float]] = None,
                           top_p: Optional[List[float]] = None,
                           softmax_temperature: List[float] = [1.0, 1.0, 1.0],
                           is_tqdm: bool = True,
                           use_fp16: bool = True,
                           max_seq_len: int = 256,
                           model_stage1=None,
                           noise: Optional[torch.Tensor] = None,
                           sample_offset: int = 0,
                           seed: Optional[int] = None,
                           use_graph: bool = True,
                           lane: int = 0,
                           row_seeds=None,
                           row_offsets=None,
                           precision: Optional[str] = None):
    """Counterpart of ``hqvae.utils.sampling.sampling_hqtransformer`` (sampling.py:240-307) for the three-level
    HQTransformer: returns ``[codes0 int64 [B, L], codes1 [B, L, 4], codes2 [B, L, 16]]`` on the model's GPU.
    ``top_k`` / ``top_p`` / ``softmax_temperature`` are per-level lists (None = no cut-off); ``cond`` as in
    ``sampling_ihqgpt``.  Extensions: ``noise`` fp32 [L, 21, B, V], ``seed`` / ``sample_offset``, ``lane``."""
    spec = model.spec
    if spec.levels != 3:
        raise ValueError('sampling_hqtransformer needs the three-level HQTransformer (stage2.type multilevel-hq)')
    if model.use_txt_cond:
        cond = torch.as_tensor(cond)
        if cond.dim() != 2:
            raise ValueError('text conditioning expects cond of shape [B, ctx_len_txt]')
        B = int(cond.shape[0])
    else:
        B = int(num_candidates)
        if model.use_cls_cond:
            if isinstance(cond, int):
                cond = torch.full((B,), int(cond), dtype=torch.int64)
            else:
                cond = torch.as_tensor(cond).reshape(-1)
                if cond.numel() == 1:
                    cond = cond.repeat(B)
            if int(cond.min()) < 0 or int(cond.max()) >= spec.n_classes:
                raise IndexError('index out of range in self')
        else:
            cond = None
    top_k = list(top_k) if top_k is not None else [None, None, None]
    top_p = list(top_p) if top_p is not None else [None, None, None]
    eng = model.engine(B, max_seq_len, lane)
    if seed is None and noise is None:
        seed = _seed_from_torch()
    return list(eng.sample3(B, cond, max_seq_len, precision=_precision(precision, use_fp16), top_k=top_k, top_p=top_p,
                            temperature=softmax_temperature, noise=noise, seed=seed or 0, sample_offset=sample_offset, use_graph=use_graph,
                            row_seeds=row_seeds, row_offsets=row_offsets))


def rearrange_codes3(codes: List[torch.Tensor], top_resolution: int):
    """'B (H W) -> B H W' and 'B (H W) (kerH kerW) -> B (H kerH) (W kerW)' with kerH = 2 and 4
    (``sampling_hqmodel.py:150-153``, ``measure_throughput/__main__.py:128-130``)."""
    B, K = codes[0].shape[0], top_resolution
    return (codes[0].reshape(B, K, K),
            codes[1].reshape(B, K, K, 2, 2).permute(0, 1, 3, 2, 4).reshape(B, 2 * K, 2 * K),
            codes[2].reshape(B, K, K, 4, 4).permute(0, 1, 3, 2, 4).reshape(B, 4 * K, 4 * K))


def rearrange_codes(codes_top: torch.Tensor, codes_bot: torch.Tensor, top_resolution: int):
    """'B (H W) -> B H W' and 'B (H W) (kerH kerW) -> B (H kerH) (W kerW)' with kerH = kerW = 2
    (``sampling_hqmodel.py:119-120``, ``measure_throughput/__main__.py:106-107``) as pure views."""
    B, H = codes_top.shape[0], top_resolution
    ct = codes_top.reshape(B, H, H)
    cb = codes_bot.reshape(B, H, H, 2, 2).permute(0, 1, 3, 2, 4).reshape(B, 2 * H, 2 * H)
    return ct, cb
