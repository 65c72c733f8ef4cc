#!/usr/bin/env python3
"""Fixture G10: token ids of the reference's own text front-end on a fixed set of captions (build container only: imports
/root/reference).  What is recorded is exactly what a CC3M / CC15M sample carries into `sampling_ihqgpt` as `cond`:
`create_tokenizer('bpe16k_huggingface', lowercase=True, dropout=None)` (hqvae/tokenizers/__init__.py:15-39) + `[PAD]` special token,
padding AND truncation to context_length (hqvae/datasets/__init__.py:145-151), `.encode(text).ids` (:178-188).
Only inputs (captions) and outputs (ids) are stored; the vocabulary stays in the reference checkout.

    python tools/gen_golden_tokenizer.py   ->  tests/golden/g10_tokenizer.npz
"""
import os
import sys
import types

import numpy as np

REF = os.environ.get('HQT_REFERENCE', '/root/reference')
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'g10_tokenizer.npz')

CAPTIONS = [
    'a photo of a cat',
    'A small bird sitting on a branch in the snow.',
    'the city skyline at night , reflected in the river',
    'An astronaut riding a horse on Mars, digital art',
    'portrait of a woman wearing a red hat',
    'two dogs playing with a ball on the beach',
    "children's toys scattered on a wooden floor",
    'a bowl of ramen with egg, pork and green onions',
    'VINTAGE CAR PARKED IN FRONT OF AN OLD BUILDING',
    'a close-up of a sunflower against a blue sky',
    'an oil painting of a stormy sea with a lighthouse',
    'people walking through a crowded market in the rain',
    '',
    'zzxqj wkvvp',                                                        # unknown-ish character sequences
    'a',
    'mountains',
    'the quick brown fox jumps over the lazy dog ' * 12,                  # longer than both context lengths: truncation
    'a very long caption about ' + ', '.join(f'thing number {i}' for i in range(40)),
    'café au lait & croissants – 50% off!',                              # non-ASCII, symbols
    'a photo of a cat',                                                   # duplicate on purpose
]


def main():
    sys.path.insert(0, REF)
    # the package pulls in its dataset / model stack on import: only the tokenizer factory is wanted
    pkg = types.ModuleType('hqvae')
    pkg.__path__ = [os.path.join(REF, 'hqvae')]
    sys.modules['hqvae'] = pkg
    sys.modules.setdefault('ftfy', types.ModuleType('ftfy'))              # import shim: only SimpleTokenizer (not used here) needs it
    from hqvae.tokenizers import create_tokenizer
    rec = {'captions': np.array(CAPTIONS)}
    for ctx in (64, 32):                                                  # hparams.ctx_len_txt of the released configs / the dataset default
        tok = create_tokenizer('bpe16k_huggingface', lowercase=True, dropout=None)
        tok.add_special_tokens(['[PAD]'])
        tok.enable_padding(length=ctx, pad_id=tok.token_to_id('[PAD]'))
        tok.enable_truncation(max_length=ctx)
        ids = np.array([tok.encode(t).ids for t in CAPTIONS], dtype=np.int64)
        assert ids.shape == (len(CAPTIONS), ctx)
        rec[f'ids_{ctx}'] = ids
        rec[f'pad_id_{ctx}'] = np.int64(tok.token_to_id('[PAD]'))
        rec['vocab_size'] = np.int64(tok.get_vocab_size())
    np.savez_compressed(OUT, **rec)
    print(OUT, {k: (v.shape if hasattr(v, 'shape') else v) for k, v in rec.items()})
    print(rec['ids_64'][0][:12], rec['pad_id_64'], rec['vocab_size'])


if __name__ == '__main__':
    main()
