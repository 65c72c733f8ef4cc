"""Checkpoints on the GPU (sampling_hqmodel.py:64-82): a Lightning-style ``{'state_dict': ...}`` file is loaded into a model, the
model samples and decodes on the GPU and the result is the oracle's on THOSE weights; then a second state dict goes into the
LIVE model (engines built, lanes cloned, graphs captured) and every output must follow the new weights -- a stale engine after
``load_state_dict`` (models.py) would reproduce the first run.  tests/test_checkpoints.py covers the file forms on the CPU."""
import os
import shutil

import numpy as np
import pytest
import torch

from hqtransformer_amd import synth
from hqtransformer_amd.config import load_config
from hqtransformer_amd.models import ImageGPT2
from hqtransformer_amd.pipeline import InflightSampler
from hqtransformer_amd.sampling import sampling_ihqgpt
from hqtransformer_amd.sampling_hqmodel import load_model
from oracle import hqt_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = os.path.join(ROOT, 'configs', 'tiny-cls.yaml')


def oracle_run(model, cls, B, n, noise):
    s2, s1 = model.stage2.spec, model.stage1.spec
    w2 = {k: v.numpy() for k, v in model.stage2.state_dict().items()}
    w1 = {k: v.numpy() for k, v in model.stage1.state_dict().items()}
    ct, cb = O.OracleStage2(s2, w2).sample(np.full(B, cls), B, n, noise)
    px = O.OracleStage1(s1, w1).decode_code(*O.rearrange_codes(ct, cb, 8))
    return ct, cb, O.postprocess(px)


def gpu_run(model, cls, B, n, noise, use_graph):
    ct, cb = sampling_ihqgpt(model.stage2, num_candidates=B, cond=cls, use_fp16=False, is_tqdm=False, max_seq_len=n,
                             noise=torch.from_numpy(noise), use_graph=use_graph)
    px = model.stage1.decode_sequences(ct, cb, precision='exact', clamp01=True)
    return ct.cpu().numpy(), cb.cpu().numpy(), px.cpu().numpy()


def test_sample_from_a_loaded_checkpoint_then_reload_into_the_live_model(tmp_path):
    d = tmp_path / 'result'
    (d / 'ckpt').mkdir(parents=True)
    shutil.copy(TINY, d / 'config.yaml')
    first, second = ImageGPT2(load_config(TINY), seed=123), ImageGPT2(load_config(TINY), seed=321)
    torch.save({'state_dict': first.state_dict(), 'epoch': 3, 'global_step': 17}, os.path.join(d, 'ckpt', 'last.ckpt'))
    model = load_model(str(d), device='cuda').eval()                          # -m <result dir>, as the reference's driver
    B, n, cls = 3, 64, 4
    noise = synth.exp_noise(21, n, B, model.stage2.spec.vocab_top)
    want1 = oracle_run(first, cls, B, n, noise)
    for graph in (False, True):                                                # builds the engines, captures the position graph
        got = gpu_run(model, cls, B, n, noise, graph)
        assert (got[0] == want1[0]).all() and (got[1] == want1[1]).all(), 'codes of the loaded checkpoint differ from the oracle on its weights'
        assert np.abs(got[2] - want1[2]).max() <= 1e-4
    # lanes share the weights of the engine they were cloned from: make sure some exist before the reload
    pipe = InflightSampler(model, lanes=2, merge=1)
    pend = [pipe.submit(B, cls, seed=5 + k, max_seq_len=n, use_fp16=False, precision='exact') for k in range(2)]
    pipe.drain()
    torch.cuda.synchronize()
    lanes_before = [tuple(t.cpu().numpy() for t in r[:2]) for r in pend]             # merge = 1: submit returns (codes_top, codes_bot, pixels, event)
    old_engine = model.stage2._engine
    assert old_engine is not None
    # ---- a second set of weights into the live model
    missing, unexpected = model.load_state_dict(second.state_dict(), strict=True)
    assert not missing and not unexpected
    assert model.stage2._engine is None and model.stage1._engine is None, 'load_state_dict must drop the engines that hold the old weights'
    want2 = oracle_run(second, cls, B, n, noise)
    assert (want2[0] != want1[0]).any() and np.abs(want2[2] - want1[2]).max() > 1e-2    # the two checkpoints really disagree
    for graph in (True, False):
        got = gpu_run(model, cls, B, n, noise, graph)
        assert (got[0] == want2[0]).all() and (got[1] == want2[1]).all(), 'stale weights after load_state_dict'
        assert np.abs(got[2] - want2[2]).max() <= 1e-4
    pipe2 = InflightSampler(model, lanes=2, merge=1)
    pend = [pipe2.submit(B, cls, seed=5 + k, max_seq_len=n, use_fp16=False, precision='exact') for k in range(2)]
    pipe2.drain()
    torch.cuda.synchronize()
    for before, r in zip(lanes_before, pend):
        assert (r[0].cpu().numpy() != before[0]).any(), 'a lane still samples from the old weights'
    # per-stage restore into the live model too (hierarchical_ar.py:880-886)
    p2 = str(tmp_path / 's2.ckpt')
    torch.save({'state_dict': first.stage2.state_dict()}, p2)
    model.stage2.from_ckpt(p2, strict=True)
    ct, cb = sampling_ihqgpt(model.stage2, num_candidates=B, cond=cls, use_fp16=False, is_tqdm=False, max_seq_len=n, noise=torch.from_numpy(noise))
    assert (ct.cpu().numpy() == want1[0]).all() and (cb.cpu().numpy() == want1[1]).all()
