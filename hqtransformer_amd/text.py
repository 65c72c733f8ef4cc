"""Text front-end of the text-conditional path (host side only).

Restates what the reference's caption datasets do before a prompt reaches ``sampling_ihqgpt``
(``hqvae/datasets/__init__.py:145-151,178-188``, ``hqvae/tokenizers/__init__.py:21-24``): a lower-casing character-level
BPE tokenizer (HuggingFace ``tokenizers.CharBPETokenizer``, ``unk_token='[UNK]'``) built from a vocab / merges pair, a
``[PAD]`` special token added on top, padding AND truncation to ``context_length`` ids.  The reference ships its 16k
vocabulary under ``hqvae/tokenizers/pretrained/bpe-16k-{vocab.json,merges.txt}``; this repository does not copy those files --
point ``vocab`` / ``merges`` at them (or at any compatible pair).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Tuple

import torch

REFERENCE_BPE16K = ('hqvae/tokenizers/pretrained/bpe-16k-vocab.json', 'hqvae/tokenizers/pretrained/bpe-16k-merges.txt')


def build_tokenizer(vocab: str, merges: str, context_length: int = 64):
    """``create_tokenizer('bpe16k_huggingface', lowercase=True, dropout=None)`` + the dataset's padding / truncation setup."""
    from tokenizers import CharBPETokenizer
    tok = CharBPETokenizer.from_file(vocab_filename=vocab, merges_filename=merges, unk_token='[UNK]', lowercase=True, dropout=None)
    tok.add_special_tokens(['[PAD]'])
    tok.enable_padding(length=context_length, pad_id=tok.token_to_id('[PAD]'))
    tok.enable_truncation(max_length=context_length)
    return tok


def encode(tok, texts: Iterable[str]) -> torch.Tensor:
    """int64 [n, context_length]: what ``CC3MTextOnly.__getitem__`` returns, stacked."""
    return torch.tensor([tok.encode(t).ids for t in texts], dtype=torch.int64)


def read_captions(path: str) -> List[str]:
    """``{split}_list.txt`` of the CC3M loaders (``<image path>\t<caption>`` per line) or one caption per line."""
    out = []
    with open(path, 'r') as fp:
        for line in fp:
            line = line.rstrip('\n')
            if not line.strip():
                continue
            toks = line.strip().split('\t')
            out.append(toks[1] if len(toks) == 2 else line.strip())
    return out


def find_reference_vocab(root: Optional[str]) -> Optional[Tuple[str, str]]:
    if not root:
        return None
    v, m = (os.path.join(root, p) for p in REFERENCE_BPE16K)
    return (v, m) if os.path.exists(v) and os.path.exists(m) else None
