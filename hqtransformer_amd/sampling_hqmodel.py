"""Counterpart of the reference's 50k-sample driver (``sampling_hqmodel.py:24-42,156-225``).

    python -m hqtransformer_amd.sampling_hqmodel -r out_dir -m <config.yaml | result_dir | ckpt path> [--top-k 2048 ...]

Same arguments and defaults, same outputs: ``samples_({cls+1}_{batch}).pkl`` = pickle (HIGHEST_PROTOCOL) of a float32
numpy array [B, 3, H, W] in [0, 1], and ``targets_({cls+1}_{batch}).npz`` with ``targets`` int64 [B]
(sampling_hqmodel.py:217-225), so ``eval_hqmodel.py`` / ``fid_utils.py:231-258`` of the reference read them unchanged.
``-m`` may point at a YAML (random-init weights, for smoke runs), a result directory holding ``config.yaml`` and
``ckpt/last.ckpt``, or the checkpoint file itself (sampling_hqmodel.py:64-82); the legacy ``stage1`` key remap of
:45-61 is applied when the checkpoint needs it.
"""
from __future__ import annotations

import argparse
import os
import pickle

import numpy as np
import torch

from .config import load_config
from .models import ImageGPT2
from .sampling import sampling_hqtransformer, sampling_ihqgpt
from .utils import set_seed


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    p.add_argument('-r', '--result-path', type=str, required=True)
    p.add_argument('-m', '--model-path', type=str, default='', required=True)
    p.add_argument('--top-k', type=int, default=2048)
    p.add_argument('--top-p', type=float, default=1.0)
    p.add_argument('--temperature', type=float, default=1.0)
    p.add_argument('--temperature-decay', type=float, default=1.0)
    p.add_argument('--batch-size', type=int, default=50)
    p.add_argument('--code-level', type=int, default=2)
    p.add_argument('--top-resolution', type=int, default=8)
    p.add_argument('--bot-resolution', type=int, default=16)
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--num-classes', type=int, default=1000)
    p.add_argument('--samples-per-class', type=int, default=None, help='default 50000 // num_classes')
    p.add_argument('--decode-precision', choices=['split', 'exact', 'fast'], default='split',
                   help='the reference decodes in fp32: split = fp32-accurate on the matrix cores (default), exact = fp32 vector ALUs, fast = bf16')
    return p


def remap_legacy_keys(sd):
    """sampling_hqmodel.py:52-57: old checkpoints store stage-1 tensors under a 17-character prefix."""
    out = {}
    for k, v in sd.items():
        out['stage1.' + k[17:] if ('stage1' in k and not k.startswith('stage1.')) else k] = v
    return out


def read_checkpoint(path: str):
    """A checkpoint file as the reference writes them: Lightning's ``{'state_dict': ...}`` (``ckpt/last.ckpt``,
    sampling_hqmodel.py:77) or a bare state dict (``ckpt/state_dict.ckpt``, eval_stage1.py:164-166)."""
    obj = torch.load(path, map_location='cpu')
    return obj['state_dict'] if isinstance(obj, dict) and 'state_dict' in obj and not torch.is_tensor(obj['state_dict']) else obj


def load_model(model_path: str, device='cuda') -> ImageGPT2:
    """The ``-m`` forms of the reference's drivers: a result directory (``config.yaml`` + ``ckpt/state_dict.ckpt`` if present,
    else ``ckpt/last.ckpt``: eval_stage1.py:156-170, sampling_hqmodel.py:64-82), a checkpoint file inside ``<result>/ckpt/``
    (sampling_hqmodel.py:65-66), or -- no reference counterpart -- a bare YAML config (random-init weights, what
    measure_throughput builds).  Legacy ``stage1`` key prefixes are remapped (load_model_legacy, :45-61)."""
    if model_path.endswith(('.yaml', '.yml')):
        return ImageGPT2(load_config(model_path)).to(device)
    if 'ckpt' in model_path:
        config_path = os.path.join(os.path.dirname(model_path), '..', 'config.yaml')
        ckpt_path = model_path
    else:
        config_path = os.path.join(model_path, 'config.yaml')
        ckpt_path = os.path.join(model_path, 'ckpt/state_dict.ckpt')
        if not os.path.exists(ckpt_path):
            ckpt_path = os.path.join(model_path, 'ckpt/last.ckpt')
    print(ckpt_path)
    model = ImageGPT2(load_config(config_path))
    model.load_state_dict(remap_legacy_keys(read_checkpoint(ckpt_path)), strict=True)
    return model.to(device)


def load_model_legacy(result_path: str, device='cuda') -> ImageGPT2:
    """sampling_hqmodel.py:45-61: ``<result>/ckpt/last.ckpt`` whose stage-1 keys carry a 17-character legacy prefix."""
    model = ImageGPT2(load_config(os.path.join(result_path, 'config.yaml')))
    sd = torch.load(os.path.join(result_path, 'ckpt/last.ckpt'), map_location='cpu')['state_dict']
    model.load_state_dict(remap_legacy_keys(sd), strict=True)
    return model.to(device)


def save_pickle(fname, data):
    with open(fname, 'wb') as fp:
        pickle.dump(data, fp, pickle.HIGHEST_PROTOCOL)


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.code_level not in (2, 3):
        raise NotImplementedError('--code-level must be 2 or 3')
    set_seed(args.seed)
    os.makedirs(args.result_path, exist_ok=True)
    model = load_model(args.model_path).eval()
    per_class = args.samples_per_class if args.samples_per_class is not None else 50000 // args.num_classes
    n = args.batch_size
    for cls_idx in range(args.num_classes):
        for num_batches in range(per_class // n):
            targets = torch.ones(n, dtype=torch.long) * cls_idx
            temps = [args.temperature * (args.temperature_decay ** i) for i in range(args.code_level)]
            if args.code_level == 3:                 # sampling_hqmodel.py:131-153,201-214
                codes = sampling_hqtransformer(model.stage2, cond=cls_idx, num_candidates=n, top_k=[args.top_k] * 3, top_p=[args.top_p] * 3,
                                               softmax_temperature=temps, use_fp16=True, is_tqdm=False,
                                               max_seq_len=args.top_resolution * args.top_resolution, model_stage1=model.stage1)
                pixels = model.stage1.decode_sequences(codes, precision=args.decode_precision, clamp01=True)
                model.stage1.range_check()                                  # SPLIT decode: raises if an activation left the fp16 range
                model.stage2.range_check()          # FAST AR sampling of up to 64 rows: raises if a persistent launch gave up (hqt_range_check)
                save_pickle(os.path.join(args.result_path, f'samples_({cls_idx + 1}_{num_batches}).pkl'), pixels.cpu().numpy())
                np.savez(os.path.join(args.result_path, f'targets_({cls_idx + 1}_{num_batches}).npz'), targets=targets.cpu().numpy())
                continue
            codes_t, codes_b = sampling_ihqgpt(model.stage2, cond=cls_idx, num_candidates=n, top_k_top=args.top_k,
                                               top_p_top=args.top_p, top_k_bot=args.top_k, top_p_bot=args.top_p,
                                               softmax_temperature=temps, use_fp16=True, is_tqdm=False,
                                               max_seq_len=args.top_resolution * args.top_resolution, model_stage1=model.stage1)
            pixels = model.stage1.decode_sequences(codes_t, codes_b, precision=args.decode_precision, clamp01=True)
            model.stage1.range_check()                                  # SPLIT decode: raises if an activation left the fp16 range
            model.stage2.range_check()          # FAST AR sampling of up to 64 rows: raises if a persistent launch gave up (hqt_range_check)
            save_pickle(os.path.join(args.result_path, f'samples_({cls_idx + 1}_{num_batches}).pkl'), pixels.cpu().numpy())
            np.savez(os.path.join(args.result_path, f'targets_({cls_idx + 1}_{num_batches}).npz'), targets=targets.cpu().numpy())


if __name__ == '__main__':
    main()
