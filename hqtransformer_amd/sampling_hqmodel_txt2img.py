"""Counterpart of the reference's text-to-image driver (``sampling_hqmodel_txt2img.py:27-42,157-217``).

    python -m hqtransformer_amd.sampling_hqmodel_txt2img -r out_dir -m <config.yaml | result_dir | ckpt path> \
        --captions val_list.txt --tokenizer-vocab bpe-16k-vocab.json --tokenizer-merges bpe-16k-merges.txt

Same arguments and defaults (``--batch_size`` keeps the reference's underscore), same loop: captions in file order,
``batch_size`` prompts per batch, one image per prompt (``num_candidates=1``: B = number of prompts, sampling.py:187-190),
``top_k`` / ``top_p`` shared by both levels, temperatures ``T * decay^level``, decode + ``clamp(0.5 x + 0.5, 0, 1)``, and
one ``samples_({batch+1}_{batch_size}).pkl`` per batch = pickle of a float32 numpy array [B, 3, H, W] in [0, 1]
(:213-216).  The caption source replaces the hard-wired CC3M directory of the reference's ``CC3MTextOnly``
(``--captions``: its ``val_list.txt`` format or one caption per line); ``--synthetic-prompts N`` draws random token ids
instead (no tokenizer files needed: smoke runs).  The last, shorter batch is kept (the reference's DataLoader does the same).
"""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch

from . import text as T
from .sampling import sampling_ihqgpt
from .sampling_hqmodel import load_model, save_pickle
from .utils import set_seed


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    p.add_argument('-r', '--result-path', type=str, required=True)
    p.add_argument('-m', '--model-path', type=str, default='', required=True)
    p.add_argument('--top-k', type=int, default=2048)
    p.add_argument('--top-p', type=float, default=1.0)
    p.add_argument('--temperature', type=float, default=1.0)
    p.add_argument('--temperature-decay', type=float, default=1.0)
    p.add_argument('--code-level', type=int, default=2)
    p.add_argument('--batch_size', type=int, default=32)
    p.add_argument('--top-resolution', type=int, default=8)
    p.add_argument('--bot-resolution', type=int, default=16)
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--dataset', type=str, default='cc3m', choices=['cc3m'])
    # where the reference reads a fixed dataset directory and its bundled vocabulary
    p.add_argument('--captions', type=str, default=None, help='val_list.txt ("<image>\t<caption>" lines) or one caption per line')
    p.add_argument('--tokenizer-vocab', type=str, default=None)
    p.add_argument('--tokenizer-merges', type=str, default=None)
    p.add_argument('--reference-root', type=str, default=os.environ.get('HQT_REFERENCE_ROOT'),
                   help='checkout of kakaobrain/hqtransformer to take the bundled bpe-16k vocabulary from')
    p.add_argument('--synthetic-prompts', type=int, default=0, help='N random-id prompts instead of captions (smoke runs)')
    p.add_argument('--decode-precision', choices=['split', 'exact', 'fast'], default='split',
                   help='the reference decodes in fp32: split = fp32-accurate on the matrix cores (default), exact = fp32 vector ALUs, fast = bf16')
    return p


def prompt_ids(args, ctx_len: int, vocab_txt: int) -> torch.Tensor:
    if args.synthetic_prompts:
        g = torch.Generator().manual_seed(args.seed)
        return torch.randint(0, vocab_txt, (args.synthetic_prompts, ctx_len), generator=g, dtype=torch.int64)
    if not args.captions:
        raise SystemExit('give --captions FILE (with a tokenizer) or --synthetic-prompts N')
    pair = (args.tokenizer_vocab, args.tokenizer_merges) if args.tokenizer_vocab and args.tokenizer_merges else \
        T.find_reference_vocab(args.reference_root)
    if pair is None:
        raise SystemExit('no tokenizer: pass --tokenizer-vocab/--tokenizer-merges or --reference-root')
    tok = T.build_tokenizer(pair[0], pair[1], context_length=ctx_len)
    ids = T.encode(tok, T.read_captions(args.captions))
    if int(ids.max()) >= vocab_txt:
        raise SystemExit(f'token id {int(ids.max())} outside the model vocabulary ({vocab_txt})')
    return ids


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.code_level != 2:
        raise NotImplementedError('--code-level 3 (HQTransformer 3-level path) is not built yet (SURVEY.md §8f rank 1)')
    set_seed(args.seed)
    os.makedirs(args.result_path, exist_ok=True)
    model = load_model(args.model_path).eval()
    if not model.stage2.use_txt_cond:
        raise SystemExit('the model is not text-conditional (stage2.use_txt_cond)')
    spec = model.stage2.spec
    ids = prompt_ids(args, spec.ctx_len_txt, spec.vocab_txt)
    temps = [args.temperature * (args.temperature_decay ** i) for i in range(args.code_level)]
    n = args.batch_size
    for batch_idx, txts in enumerate(ids.split(n)):
        codes_t, codes_b = sampling_ihqgpt(model.stage2, cond=txts.cuda(), num_candidates=1, top_k_top=args.top_k, top_p_top=args.top_p,
                                           top_k_bot=args.top_k, top_p_bot=args.top_p, softmax_temperature=temps, use_fp16=True,
                                           is_tqdm=False, max_seq_len=args.top_resolution * args.top_resolution, model_stage1=model.stage1)
        pixels = model.stage1.decode_sequences(codes_t, codes_b, precision=args.decode_precision, clamp01=True)
        model.stage1.range_check()                                  # SPLIT decode: raises if an activation left the fp16 range
        model.stage2.range_check()          # FAST AR sampling of up to 64 rows: raises if a persistent launch gave up (hqt_range_check)
        save_pickle(os.path.join(args.result_path, f'samples_({batch_idx + 1}_{n}).pkl'), pixels.cpu().numpy().astype(np.float32))


if __name__ == '__main__':
    main()
