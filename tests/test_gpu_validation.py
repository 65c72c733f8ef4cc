"""Caller-supplied indices and output buffers: the reference raises IndexError from nn.Embedding / F.embedding for an id outside a
table (class / text ids, teacher-forced codes, code grids); the counterpart raises the same before any launch, and the kernels
clamp such indices regardless (no fault even when a caller bypasses the Python surface)."""
import ctypes as C

import numpy as np
import pytest
import torch

from hqtransformer_amd import _lib, synth
from hqtransformer_amd._lib import PRECISION_EXACT, hqt_sample_opts
from hqtransformer_amd.engine import Engine, _ptr
from tests.helpers import load, stage1_from_fixture, stage2_from_fixture

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


@pytest.fixture(scope='module')
def engines():
    fx2, fx1 = load('g4_tiny_cls.npz'), load('g5_decode_64.npz')
    s2, w2 = stage2_from_fixture(fx2)
    s1, w1 = stage1_from_fixture(fx1)
    e2 = Engine(s2, None, DEV, 4, 8)
    e2.load(stage2=w2)
    e2.finalize()
    e1 = Engine(None, s1, DEV, 2)
    e1.load(stage1=w1)
    e1.finalize()
    return s2, e2, s1, e1, fx1


def test_out_of_range_indices_raise_index_error(engines):
    s2, e2, s1, e1, fx1 = engines
    B, n = 2, 4
    noise = torch.from_numpy(synth.exp_noise(1, n, B, s2.vocab_top))
    for bad in (torch.tensor([0, s2.n_classes]), torch.tensor([-1, 0]), torch.tensor([0, s2.n_classes], device=DEV)):
        with pytest.raises(IndexError):
            e2.sample(B, bad, n, precision=PRECISION_EXACT, noise=noise)
    ok = torch.tensor([0, s2.n_classes - 1])
    ft = torch.zeros((B, n), dtype=torch.int64)
    ft[1, 2] = s2.vocab_top
    with pytest.raises(IndexError):
        e2.sample(B, ok, n, precision=PRECISION_EXACT, noise=noise, force_top=ft)
    fb = torch.zeros((B, n, 4), dtype=torch.int64, device=DEV)
    fb[0, 0, 3] = -5
    with pytest.raises(IndexError):
        e2.sample(B, ok, n, precision=PRECISION_EXACT, noise=noise, force_bot=fb)
    ct, cb = torch.from_numpy(fx1['code_t']).clone(), torch.from_numpy(fx1['code_b']).clone()
    cb[0, 0, 0] = s1.n_embed
    with pytest.raises(IndexError):
        e1.decode(ct, cb, precision=PRECISION_EXACT)
    ct[1, 1, 1] = -1
    with pytest.raises(IndexError):
        e1.decode(ct.to(DEV), None, precision=PRECISION_EXACT)
    # the engines still work afterwards
    e2.sample(B, ok, n, precision=PRECISION_EXACT, noise=noise)
    e1.decode(torch.from_numpy(fx1['code_t']), torch.from_numpy(fx1['code_b']), precision=PRECISION_EXACT)
    torch.cuda.synchronize()


def test_out_buffers_are_validated(engines):
    s2, e2, s1, e1, fx1 = engines
    B, n = 2, 4
    ok = torch.tensor([1, 2])
    good = (torch.empty((B, n), dtype=torch.int64, device=DEV), torch.empty((B, n, 4), dtype=torch.int64, device=DEV))
    e2.sample(B, ok, n, precision=PRECISION_EXACT, seed=3, out=good)
    for bad in ((good[0][:, :2], good[1]), (good[0].int(), good[1]), (good[0].cpu(), good[1]), (good[0],)):
        with pytest.raises(ValueError):
            e2.sample(B, ok, n, precision=PRECISION_EXACT, seed=3, out=bad)
    with pytest.raises(ValueError):
        e1.decode(torch.from_numpy(fx1['code_t']), torch.from_numpy(fx1['code_b']), precision=PRECISION_EXACT,
                  out=torch.empty((2, 3, 8, 8), device=DEV))


def test_kernels_clamp_indices_that_bypass_the_python_surface(engines):
    """Straight through the C ABI with ids far outside the tables: clamped on the device (common.h: clamp_idx), no fault, and
    the result equals the run with the ids clamped by hand."""
    s2, e2, s1, e1, fx1 = engines
    B, n = 2, 3
    noise = torch.from_numpy(synth.exp_noise(2, n, B, s2.vocab_top)).to(DEV)
    lib = _lib.load()

    def run(cond, ft):
        o = hqt_sample_opts()
        o.precision, o.n_steps, o.temperature_top, o.temperature_bot = PRECISION_EXACT, n, 1.0, 1.0
        ot = torch.empty((B, n), dtype=torch.int64, device=DEV)
        ob = torch.empty((B, n, 4), dtype=torch.int64, device=DEV)
        _lib.check(lib.hqt_sample(e2.h, B, _ptr(cond), C.byref(o), _ptr(noise), _ptr(ft), None, None, _ptr(ot), _ptr(ob), None))
        torch.cuda.synchronize()
        return ot.cpu(), ob.cpu()
    wild = run(torch.tensor([10 ** 9, -7], device=DEV), torch.tensor([[5, 10 ** 12, 1], [-3, 2, 7]], device=DEV))
    tame = run(torch.tensor([s2.n_classes - 1, 0], device=DEV), torch.tensor([[5, s2.vocab_top - 1, 1], [0, 2, 7]], device=DEV))
    assert torch.equal(wild[0], tame[0]) and torch.equal(wild[1], tame[1])


def test_index_trust_dies_with_the_tensor(engines):
    """Index tensors that passed the range check are remembered (no second device round trip when sample()'s codes go into decode());
    the memory must not stay trusted once the tensor is gone: the caching allocator hands the block to the next tensor of that size
    (seen in the full suite: a bad code grid landed on a block an earlier test had validated, and decode() did not raise)."""
    import weakref
    from hqtransformer_amd import engine as E
    s2, e2, s1, e1, fx1 = engines
    cb = torch.from_numpy(fx1['code_b']).to(DEV)
    ct = torch.from_numpy(fx1['code_t']).to(DEV)
    e1.decode(ct, cb, precision=PRECISION_EXACT)                # checked once ...
    key = E.Engine._ident(ct)
    assert key in E._TRUSTED and E._TRUSTED[key][1]() is ct.untyped_storage()      # ... and trusted while it lives
    assert E._TRUSTED[key][2] <= s1.n_embed                                         # with the bound it was validated against
    bad_host = torch.from_numpy(fx1['code_t']).clone()
    bad_host[0, 0, 0] = s1.n_embed + 7
    bad = bad_host.to(DEV)
    # what a freed-and-reused block leaves behind: an entry for this address, size and version whose storage is another one
    E._TRUSTED[E.Engine._ident(bad)] = (bad._version, weakref.ref(ct.untyped_storage()), s1.n_embed)
    with pytest.raises(IndexError):
        e1.decode(bad, cb, precision=PRECISION_EXACT)
    # a tensor validated against a LARGER table is checked again for a smaller one (the entry carries its bound); views of one
    # buffer with different strides do not alias each other
    big = torch.full_like(ct, s1.n_embed + 3)
    e1._trust(big, bound=s1.n_embed + 4)                          # as if it had passed a check against a table of n_embed + 4 rows
    with pytest.raises(IndexError):
        e1.decode(big, cb, precision=PRECISION_EXACT)
    assert E.Engine._ident(ct) != E.Engine._ident(ct.transpose(1, 2))
