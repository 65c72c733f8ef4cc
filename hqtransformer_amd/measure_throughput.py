"""Counterpart of the reference's throughput harness (``measure_throughput/__main__.py:51-180``).

    python -m hqtransformer_amd.measure_throughput model_path=configs/imagenet-12l.yaml batch_size=64

Same dot-list keys and defaults (``Experiment`` dataclass, :34-48), same loop accounting: ``n_loop`` loops of
``ceil(1000 / batch_size)`` iterations, the first ``warmup`` loops discarded, every iteration = one
``sampling_ihqgpt`` call (random class, ``top_k = top_p = None``, temperatures 1.0, ``use_fp16=True``) timed as
"ar", then code rearrange + ``stage1.decode_code`` + ``clamp(0.5 x + 0.5, 0, 1)`` timed as "decode", GPU events for
both, and the same printed lines (``ms/sample (ar: .., decode: ..)``).  Random-init weights, like the reference.
Differences, stated: the whole batch is decoded in one call instead of ``batch_size`` calls of one image
(``decode_batch=1`` restores the reference's chunking).  The reference decodes in fp32 (outside autocast, :108-113); the
default ``decode_precision=split`` is fp32-accurate on the matrix cores (pixels within 1e-4 of the fp32 result), ``exact`` runs
fp32 FMA chains on the vector ALUs, ``fast`` bf16 MFMA (faster, 0.04 max pixel error).  ``inflight=N`` (default 1 = the reference's order) keeps N
iterations in flight on N lanes (``hqtransformer_amd.pipeline``): same iterations, same accounting of the loop's wall
time; the per-phase figures then are lane times, which overlap.  ``merge=k`` executes k consecutive iterations as one pass of
k x batch_size rows (every iteration keeps its own class id and seed; ``bench.py`` picks its schedule from its step count K: 2 lanes x passes of min(32, ceil(K / 2)) steps);
the per-phase figures are then measured per pass.
"""
from __future__ import annotations

import platform
import random
import sys
import time

import torch

from .config import load_config, parse_dotlist
from .models import ImageGPT2
from .sampling import rearrange_codes, rearrange_codes3, sampling_hqtransformer, sampling_ihqgpt

EXPERIMENT_DEFAULTS = dict(f=32, model='huge', d=4, c=16384, batch_size=50, n_loop=6, warmup=1, model_path='',
                           top_resolution=8, code_levels=2, decode_batch=0, decode_precision='split', seed=0, inflight=1, merge=1)


def iterations_per_loop(batch_size: int) -> int:
    """``n_iter_per_loop = (1000 + batch_size - 1) // batch_size`` (measure_throughput/__main__.py:76)."""
    return (1000 + batch_size - 1) // batch_size


def load_model(result_path: str) -> ImageGPT2:
    return ImageGPT2(load_config(result_path))


def main(args) -> dict:
    torch.set_grad_enabled(False)
    if args.code_levels not in (2, 3):
        raise NotImplementedError('code_levels must be 2 or 3')
    random.seed(args.seed)
    model_ar = load_model(args.model_path)
    device = torch.device('cuda')
    model_ar = model_ar.to(device)
    model_ar.eval()
    title = f'bs{args.batch_size}, sampling loops {args.warmup + 1}-{args.n_loop}'
    print(title)
    print('python: %s, torch: %s, hip: %s, gpu: %s' % (platform.python_version(), torch.__version__, torch.version.hip,
                                                      torch.cuda.get_device_name(device)))
    ar_size = sum(p.numel() for p in model_ar.stage2.parameters()) / (10 ** 6)
    print(f'transformer size: {ar_size:.1f}M')
    batch_size = args.batch_size
    n_iter_per_loop = iterations_per_loop(batch_size)
    n_loop = args.n_loop

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
        for i in range(n_iter_per_loop if pipe is not None else 0):
            pipe.submit(batch_size, random.randint(0, 999), max_seq_len=args.top_resolution * args.top_resolution, use_fp16=True,
                        precision=args.decode_precision, clamp01=True, softmax_temperature=[1.0 for _ in range(args.code_levels)],
                        phase_events=None if merge > 1 else (starts[i], middles[i], ends[i]))
        if pipe is not None:
            pipe.drain()
        for i in range(n_iter_per_loop if (pipe is None and args.code_levels == 3) else 0):     # measure_throughput/__main__.py:116-138
            starts[i].record()
            codes_levels = sampling_hqtransformer(model_ar.stage2, num_candidates=batch_size, cond=random.randint(0, 999),
                                                  top_k=[None] * 3, top_p=[None] * 3, softmax_temperature=[1.0] * 3, use_fp16=True,
                                                  is_tqdm=False, max_seq_len=args.top_resolution * args.top_resolution, model_stage1=None)
            middles[i].record()
            if args.decode_batch and args.decode_batch < batch_size:
                grids = rearrange_codes3(codes_levels, args.top_resolution)
                pixels = torch.cat([model_ar.stage1.decode_code([g[j:j + args.decode_batch] for g in grids], precision=args.decode_precision)
                                    for j in range(0, batch_size, args.decode_batch)], dim=0)
                _ = (0.5 * pixels + 0.5).clamp(0, 1)
            else:
                _ = model_ar.stage1.decode_sequences(codes_levels, precision=args.decode_precision, clamp01=True)
            ends[i].record()
        for i in range(n_iter_per_loop if (pipe is None and args.code_levels == 2) else 0):
            starts[i].record()
            codes_t, codes_b = sampling_ihqgpt(model_ar.stage2, cond=random.randint(0, 999), num_candidates=batch_size,
                                               top_k_top=None, top_p_top=None, top_k_bot=None, top_p_bot=None,
                                               softmax_temperature=[1.0 for _ in range(args.code_levels)], use_fp16=True,
                                               is_tqdm=False, max_seq_len=args.top_resolution * args.top_resolution,
                                               model_stage1=None)
            middles[i].record()
            if args.decode_batch and args.decode_batch < batch_size:
                grid_t, grid_b = rearrange_codes(codes_t, codes_b, args.top_resolution)
                pixels = torch.cat([model_ar.stage1.decode_code(ct, cb, precision=args.decode_precision)
                                    for ct, cb in zip(grid_t.split(args.decode_batch), grid_b.split(args.decode_batch))], dim=0)
                _ = (0.5 * pixels + 0.5).clamp(0, 1)
            else:      # rearranges and the clamp are folded into the decode kernels
                _ = model_ar.stage1.decode_sequences(codes_t, codes_b, precision=args.decode_precision, clamp01=True)
            ends[i].record()
        torch.cuda.synchronize(device)
        wall_s = time.time() - t_begin
        model_ar.stage1.range_check()                      # SPLIT decode: raises if an activation left the fp16 range
        model_ar.stage2.range_check()          # FAST AR sampling of up to 64 rows: raises if a persistent launch gave up (hqt_range_check)
        if pipe is not None and merge > 1:                 # per pass: (AR start, AR end, decode end) on the pass's lane
            log, pipe.phase_log = pipe.phase_log, []
            phase_s = [sum(ev[a].elapsed_time(ev[a + 1]) for ev, _ in log) / 1000 for a in (0, 1)]
        else:
            marks = (starts, middles, ends)
            phase_s = [sum(marks[a][i].elapsed_time(marks[a + 1][i]) for i in range(n_iter_per_loop)) / 1000 for a in (0, 1)]
        tag = f'{loop_idx + 1}/{n_loop}'
        print(f'{tag} | {wall_s:.1f} s/loop (ar: {phase_s[0]:.1f}, decode: {phase_s[1]:.1f})')
        images = n_iter_per_loop * batch_size
        per_image_ms = tuple(1000.0 * t / images for t in (wall_s, *phase_s))        # (whole iteration, AR loop, decode) per sample
        print(f'{tag} | {per_image_ms[0]:.1f} ms/sample (ar: {per_image_ms[1]:.1f}, decode: {per_image_ms[2]:.1f})')
        return per_image_ms

    print('-' * 80)
    kept = [loop(k) for k in range(args.n_loop)][args.warmup:]                   # the first `warmup` loops are run and dropped
    print('-' * 80)
    mean_ms, mean_ar_ms, mean_dec_ms = (sum(col) / len(kept) for col in zip(*kept))
    print(f'{title} | {mean_ms:.4f} ms/sample (ar: {mean_ar_ms:.4f}, decode: {mean_dec_ms:.4f})')
    print('=' * 80)
    return dict(ms_per_sample=mean_ms, ms_ar=mean_ar_ms, ms_decode=mean_dec_ms, images_per_s=1000.0 / mean_ms)


if __name__ == '__main__':
    main(parse_dotlist(sys.argv[1:], EXPERIMENT_DEFAULTS))
