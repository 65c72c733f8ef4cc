"""Text front-end against fixture G10 = ids the reference's own `create_tokenizer('bpe16k_huggingface')` + dataset padding /
truncation produced for 20 captions (tools/gen_golden_tokenizer.py; hqvae/tokenizers/__init__.py:15-39,
hqvae/datasets/__init__.py:145-151).  The 16k vocabulary is a data file of the reference that this repository does not copy:
the comparison runs wherever a reference checkout is present (HQT_REFERENCE or /root/reference -- the build container) and is
skipped elsewhere; the fixture's self-consistency checks run everywhere."""
import os

import numpy as np
import pytest
import torch

from hqtransformer_amd import text
from tests.helpers import load

REF = os.environ.get('HQT_REFERENCE', '/root/reference')


def vocab_pair():
    """HQT_BPE16K_VOCAB / HQT_BPE16K_MERGES name the two files directly (any host that has them: the test then RUNS, and fails if they are
    unreadable); otherwise the reference checkout under HQT_REFERENCE or /root/reference; None = neither is present."""
    v, m = os.environ.get('HQT_BPE16K_VOCAB'), os.environ.get('HQT_BPE16K_MERGES')
    if v or m:
        return (v or '', m or '')
    return text.find_reference_vocab(REF)


def test_fixture_is_self_consistent():
    fx = load('g10_tokenizer.npz')
    caps = [str(c) for c in fx['captions']]
    assert len(caps) == 20 and int(fx['vocab_size']) == 16384
    for ctx in (64, 32):
        ids, pad = fx[f'ids_{ctx}'], int(fx[f'pad_id_{ctx}'])
        assert ids.shape == (20, ctx) and ids.min() >= 0 and ids.max() < 16384
        assert (ids[caps.index('')] == pad).all()                       # the empty caption is all padding
        assert (ids[0] == ids[19]).all()                                # same caption, same ids
        assert (ids[16] != pad).all()                                   # the long caption fills the context (truncated)
    assert (fx['ids_64'][:, :32] == fx['ids_32'])[fx['ids_32'][:, -1] == int(fx['pad_id_32'])].all()   # short captions: the 32-id form is a prefix


@pytest.mark.skipif(vocab_pair() is None, reason='neither HQT_BPE16K_VOCAB / HQT_BPE16K_MERGES nor a reference checkout (bpe-16k vocabulary) is present')
def test_build_tokenizer_reproduces_the_reference_ids():
    fx = load('g10_tokenizer.npz')
    vocab, merges = vocab_pair()
    assert os.path.isfile(vocab) and os.path.isfile(merges), f'the vocabulary files named by the environment do not exist: {vocab!r}, {merges!r}'
    caps = [str(c) for c in fx['captions']]
    for ctx in (64, 32):
        tok = text.build_tokenizer(vocab, merges, context_length=ctx)
        got = text.encode(tok, caps)
        assert got.dtype == torch.int64 and tuple(got.shape) == (20, ctx)
        assert np.array_equal(got.numpy(), fx[f'ids_{ctx}']), f'token ids differ from the reference at context length {ctx}'
        assert tok.token_to_id('[PAD]') == int(fx[f'pad_id_{ctx}'])


def test_read_captions_both_formats(tmp_path):
    p = tmp_path / 'val_list.txt'
    p.write_text('images/0001.jpg\ta photo of a cat\n\nimages/0002.jpg\ttwo dogs\nplain caption line\n')
    assert text.read_captions(str(p)) == ['a photo of a cat', 'two dogs', 'plain caption line']
