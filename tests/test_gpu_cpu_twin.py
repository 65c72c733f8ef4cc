"""hqt_sample / hqt_decode_seq on the GPU against their hqt_cpu_* twins (include/hqt_cpu.h) -- the C++ restatement bench.py times as
`cpu_baseline` -- with BOTH sides drawing the in-library Philox stream for the same (seed, sample_offset): EXACT code sequences
bit-identical, pixels within 1e-4 (hierarchical_ar.py:428-563,667-789; generator.py:312-367)."""
import numpy as np
import pytest
import torch

from hqtransformer_amd import synth
from hqtransformer_amd._lib import PRECISION_EXACT, PRECISION_SPLIT
from hqtransformer_amd.engine import Engine
from hqtransformer_amd.spec import Stage1Spec, Stage2Spec
from oracle import hqt_cpu

pytestmark = pytest.mark.gpu


def test_gpu_exact_equals_the_cpu_twin_under_the_same_philox_keys():
    s2 = Stage2Spec(embed_dim=128, n_layers=2, n_heads=4, n_layers_depth=2, vocab_top=512, vocab_bot=512, vocab_txt=64,
                    ctx_len_img=64, ctx_len_txt=16, n_classes=10, cond=1, embedding=0)
    s1 = Stage1Spec(ch=32, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[16], resolution=64, z_channels=32, embed_dim=16, n_embed=512)
    w2, w1 = synth.stage2_weights(s2, 11, 'fixture'), synth.stage1_weights(s1, 12, 'fixture')
    B, n = 5, 64
    cond = np.array([1, 5, 9, 0, 3])
    eng = Engine(s2, s1, torch.device('cuda:0'), B, 64)
    eng.load(stage2=w2, stage1=w1)
    eng.finalize()
    ct, cb, lg = eng.sample(B, torch.from_numpy(cond), n, precision=PRECISION_EXACT, top_k=(100, None), top_p=(0.95, None), temperature=(1.0, 0.9),
                            seed=4321, sample_offset=17, return_logits=True)
    px = eng.decode(ct, cb, precision=PRECISION_SPLIT, seq_layout=True, clamp01=True)
    torch.cuda.synchronize()
    twin = hqt_cpu.CpuTwin(s2, s1, w2, w1)
    tct, tcb, tlg = twin.sample(cond, B, n, None, (100, None), (0.95, None), (1.0, 0.9), return_logits=True, seed=4321, sample_offset=17)
    assert (ct.cpu().numpy() == tct).all() and (cb.cpu().numpy() == tcb).all()
    assert np.abs(lg.cpu().numpy() - tlg).max() <= 2e-4
    tpx = twin.decode_code(tct, tcb, clamp01=True, seq_layout=True)
    assert np.abs(px.cpu().numpy() - tpx).max() <= 1e-4
    eng.close()
