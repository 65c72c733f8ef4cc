"""Counterpart of the reference's text-to-image throughput harness (``measure_throughput_txt/__main__.py:83-188``).

    python -m hqtransformer_amd.measure_throughput_txt model_path=configs/cc15m-12l-txt.yaml batch_size=64

Same dot-list keys and defaults as its ``Experiment`` dataclass (:66-80: ``batch_size=50``, ``n_loop=6``, ``warmup=1``,
``top_resolution=8``, ``bot_resolution=16``, ``dataset='cc3m'``), same loop accounting (:106-165): ``n_loop`` loops of
``ceil(1000 / batch_size)`` iterations, the first ``warmup`` loops dropped, every iteration = one ``sampling_ihqgpt`` call on a
batch of prompts [B, ctx_len_txt] (``num_candidates=1`` as there -- with text conditioning the batch is the prompt count,
sampling.py:187-190) with the quality-mode sampler of that file (``top_k = 2048``, ``top_p = 1.0`` on both levels, temperature 1.0,
``use_fp16=True``) timed as "ar", then rearrange + ``stage1.decode_code`` + ``clamp(0.5 x + 0.5, 0, 1)`` timed as "decode", and the
same printed lines.

Differences, stated:
  * prompts.  The reference iterates ``CC3MTextOnly('val')`` (datasets/__init__.py:178-188), which is not available offline.
    ``prompts=synthetic`` (default) draws ids uniformly from the text vocabulary, one fresh batch per iteration, seeded;
    ``captions=<file>`` + ``tokenizer_vocab=`` / ``tokenizer_merges=`` runs real captions through the same BPE front-end
    (``hqtransformer_amd.text``), cycling over the file.
  * ``softmax_temperature``: the reference passes the float 1.0 (:135) where ``sampling_ihqgpt`` indexes a list -- its call raises
    TypeError as written; a float here means "this temperature on both levels".
  * decode: the whole batch in one call (``decode_batch=1`` restores the reference's one-image chunks, :146-150); the reference
    decodes in fp32 outside autocast, ``decode_precision=split`` (default) is fp32-accurate on the matrix cores.
  * ``inflight=N`` / ``merge=k``: several iterations in flight / merged into one pass, as in ``measure_throughput``.
"""
from __future__ import annotations

import platform
import sys
import time

import numpy as np
import torch

from .config import load_config, parse_dotlist
from .models import ImageGPT2
from .sampling import rearrange_codes, sampling_ihqgpt

EXPERIMENT_DEFAULTS = dict(f=32, model='huge', d=4, c=16384, batch_size=50, n_loop=6, warmup=1, model_path='',
                           top_resolution=8, bot_resolution=16, dataset='cc3m',
                           prompts='synthetic', captions='', tokenizer_vocab='', tokenizer_merges='',
                           top_k=2048, top_p=1.0, softmax_temperature=1.0,
                           decode_batch=0, decode_precision='split', seed=0, inflight=1, merge=1)


def iterations_per_loop(batch_size: int) -> int:
    """``n_iter_per_loop = (1000 + batch_size - 1) // batch_size`` (measure_throughput_txt/__main__.py:103)."""
    return (1000 + batch_size - 1) // batch_size


def prompt_batches(args, spec):
    """Generator of int64 [batch_size, ctx_len_txt] prompt batches: the text loader's role (:28-45, :118)."""
    B, ctx = int(args.batch_size), int(spec.ctx_len_txt)
    if args.captions:
        from . import text
        if not (args.tokenizer_vocab and args.tokenizer_merges):
            raise ValueError('captions= needs tokenizer_vocab= and tokenizer_merges= (the reference ships hqvae/tokenizers/pretrained/bpe-16k-*)')
        tok = text.build_tokenizer(args.tokenizer_vocab, args.tokenizer_merges, context_length=ctx)
        ids = text.encode(tok, text.read_captions(args.captions))
        if int(ids.max()) >= spec.vocab_txt:
            raise IndexError('a caption token id lies outside the model\'s text vocabulary')
        k = 0
        while True:
            idx = [(k + j) % ids.shape[0] for j in range(B)]
            k = (k + B) % ids.shape[0]
            yield ids[idx]
    else:
        rng = np.random.default_rng(int(args.seed) + 1)
        while True:
            yield torch.from_numpy(rng.integers(0, spec.vocab_txt, (B, ctx), dtype=np.int64))


def main(args) -> dict:
    torch.set_grad_enabled(False)
    model_ar = ImageGPT2(load_config(args.model_path))
    if not getattr(model_ar.stage2, 'use_txt_cond', False):
        raise ValueError(f'{args.model_path} is not a text-conditional model (use hqtransformer_amd.measure_throughput)')
    device = torch.device('cuda')
    model_ar = model_ar.to(device)
    model_ar.eval()
    title = f'bs{args.batch_size}, sampling loops {args.warmup + 1}-{args.n_loop}'
    print(title)
    print('python: %s, torch: %s, hip: %s, gpu: %s' % (platform.python_version(), torch.__version__, torch.version.hip,
                                                      torch.cuda.get_device_name(device)))
    ar_size = sum(p.numel() for p in model_ar.stage2.parameters()) / (10 ** 6)
    print(f'transformer size: {ar_size:.1f}M')
    batch_size = int(args.batch_size)
    n_iter_per_loop = iterations_per_loop(batch_size)
    n_loop = int(args.n_loop)
    n_pos = int(args.top_resolution) * int(args.top_resolution)
    t = args.softmax_temperature
    temperature = [float(v) for v in t] if isinstance(t, (list, tuple)) else [float(t), float(t)]
    top_k = None if args.top_k in (None, 0, 'None') else int(args.top_k)
    top_p = None if args.top_p in (None, 0, 'None') else float(args.top_p)
    sampler = dict(top_k_top=top_k, top_p_top=top_p, top_k_bot=top_k, top_p_bot=top_p, softmax_temperature=temperature)
    prompts = prompt_batches(args, model_ar.stage2.spec)

    pipe = None
    merge = max(1, int(args.merge))
    if int(args.inflight) > 1 or merge > 1:
        from .pipeline import InflightSampler
        pipe = InflightSampler(model_ar, lanes=int(args.inflight), device=device, merge=merge, record_phases=merge > 1)

    def loop(loop_idx: int):
        starts = [torch.cuda.Event(enable_timing=True) for _ in range(n_iter_per_loop)]
        middles = [torch.cuda.Event(enable_timing=True) for _ in range(n_iter_per_loop)]
        ends = [torch.cuda.Event(enable_timing=True) for _ in range(n_iter_per_loop)]
        torch.cuda.synchronize(device)
        t_begin = time.time()
        for i in range(n_iter_per_loop):
            txts = next(prompts)
            if pipe is not None:
                pipe.submit(batch_size, txts, max_seq_len=n_pos, use_fp16=True, precision=args.decode_precision, clamp01=True,
                            phase_events=None if merge > 1 else (starts[i], middles[i], ends[i]), **sampler)
                continue
            starts[i].record()
            codes_t, codes_b = sampling_ihqgpt(model_ar.stage2, cond=txts, num_candidates=1, use_fp16=True, is_tqdm=False,
                                               max_seq_len=n_pos, model_stage1=None, **sampler)
            middles[i].record()
            if args.decode_batch and int(args.decode_batch) < batch_size:
                grid_t, grid_b = rearrange_codes(codes_t, codes_b, int(args.top_resolution))
                pixels = torch.cat([model_ar.stage1.decode_code(ct, cb, precision=args.decode_precision)
                                    for ct, cb in zip(grid_t.split(int(args.decode_batch)), grid_b.split(int(args.decode_batch)))], dim=0)
                _ = (0.5 * pixels + 0.5).clamp(0, 1)
            else:      # rearranges and the clamp are folded into the decode kernels
                _ = model_ar.stage1.decode_sequences(codes_t, codes_b, precision=args.decode_precision, clamp01=True)
            ends[i].record()
        if pipe is not None:
            pipe.drain()
        torch.cuda.synchronize(device)
        wall_s = time.time() - t_begin
        model_ar.stage1.range_check()
        model_ar.stage2.range_check()          # FAST AR sampling of up to 64 rows: raises if a persistent launch gave up (hqt_range_check)
        if pipe is not None and merge > 1:
            log, pipe.phase_log = pipe.phase_log, []
            phase_s = [sum(ev[a].elapsed_time(ev[a + 1]) for ev, _ in log) / 1000 for a in (0, 1)]
        else:
            marks = (starts, middles, ends)
            phase_s = [sum(marks[a][i].elapsed_time(marks[a + 1][i]) for i in range(n_iter_per_loop)) / 1000 for a in (0, 1)]
        tag = f'{loop_idx + 1}/{n_loop}'
        print(f'{tag} | {wall_s:.1f} s/loop (ar: {phase_s[0]:.1f}, decode: {phase_s[1]:.1f})')
        images = n_iter_per_loop * batch_size
        per_image_ms = tuple(1000.0 * s / images for s in (wall_s, *phase_s))
        print(f'{tag} | {per_image_ms[0]:.1f} ms/sample (ar: {per_image_ms[1]:.1f}, decode: {per_image_ms[2]:.1f})')
        return per_image_ms

    print('-' * 80)
    kept = [loop(k) for k in range(n_loop)][int(args.warmup):]
    print('-' * 80)
    mean_ms, mean_ar_ms, mean_dec_ms = (sum(col) / len(kept) for col in zip(*kept))
    print(f'{title} | {mean_ms:.4f} ms/sample (ar: {mean_ar_ms:.4f}, decode: {mean_dec_ms:.4f})')
    print('=' * 80)
    return dict(ms_per_sample=mean_ms, ms_ar=mean_ar_ms, ms_decode=mean_dec_ms, images_per_s=1000.0 / mean_ms)


if __name__ == '__main__':
    main(parse_dotlist(sys.argv[1:], EXPERIMENT_DEFAULTS))
